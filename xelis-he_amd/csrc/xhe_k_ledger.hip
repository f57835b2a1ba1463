// The ledger as a prefix sum (src/tx/verify.rs:301-341,347-374): segmented over (account, asset) slots by
// subtracting the global prefix at each slot's first operation.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

// ---- ledger scan -----------------------------------------------------------------------------------
// A ciphertext is two points (commitment C, handle D) whose sums never mix, so every kernel of the scan runs as two
// independent halves (blockIdx.y = 0 for C, 1 for D): half the dependent additions per thread.  All buffers keep a
// ciphertext as 64 words, C then D.
__device__ __forceinline__ void scan_apply_op(ge &acc, const dev_op &op, u32 which, const u32 *__restrict__ pts, const u32 *__restrict__ gtab) {
    if (op.kind == 2) {                               // minus amount * G: the commitment only
        if (which) return;
        u64 a = op.amount;
#pragma unroll 1
        for (u32 w = 0; w < 8 && a; w++, a >>= 8) {
            u32 b = (u32)a & 255;
            if (b) acc = ge_msub(acc, ge_niels_load(gtab + 32ull * (w * 256 + b)));
        }
    } else {
        acc = ge_madd_signed(acc, ge_niels_load(pts + 32ull * (which ? op.d_pt : op.c_pt)), op.kind == 1);
    }
}

// inclusive Hillis-Steele scan of one point per thread through LDS; returns the inclusive value of thread t
__device__ __forceinline__ ge block_scan_ge(ge v, u32 *lds /* SCAN_BLOCK * 32 words */) {
    u32 t = threadIdx.x;
    for (u32 d = 1; d < SCAN_BLOCK; d <<= 1) {
        ge_store(lds + 32 * t, v);
        __syncthreads();
        if (t >= d) { ge o = ge_load(lds + 32 * (t - d)); v = ge_add(o, v); }
        __syncthreads();
    }
    return v;
}

// phase 1: per-thread sums, block scan; writes each thread's exclusive prefix inside its block and the block total
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_local(const dev_op *__restrict__ ops, u32 n_ops, const u32 *__restrict__ pts, const u32 *__restrict__ gtab,
                                                           u32 *__restrict__ thread_excl, u32 *__restrict__ block_tot) {
    wave_priority_high();
    __shared__ u32 lds[SCAN_BLOCK * 32];
    const u32 t = threadIdx.x, gt = blockIdx.x * SCAN_BLOCK + t, which = blockIdx.y;
    ge acc = ge_identity();
    for (u32 k = 0; k < SCAN_K; k++) {
        u32 o = gt * SCAN_K + k;
        if (o < n_ops) scan_apply_op(acc, ops[o], which, pts, gtab);
    }
    ge incl = block_scan_ge(acc, lds);
    // exclusive = inclusive of the previous thread
    ge_store(lds + 32 * t, incl);
    __syncthreads();
    ge excl = t ? ge_load(lds + 32 * (t - 1)) : ge_identity();
    ge_store(thread_excl + 64ull * gt + 32 * which, excl);
    if (t == SCAN_BLOCK - 1) ge_store(block_tot + 64ull * blockIdx.x + 32 * which, incl);
}
// phase 2: one block per half turns the block totals into exclusive block prefixes (in place)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_blocks(u32 *__restrict__ block_tot, u32 n_blocks) {
    wave_priority_high();
    __shared__ u32 lds[SCAN_BLOCK * 32];
    const u32 t = threadIdx.x, which = blockIdx.y;
    u32 per = (n_blocks + SCAN_BLOCK - 1) / SCAN_BLOCK;
    u32 lo = t * per, hi = lo + per; if (hi > n_blocks) hi = n_blocks;
    ge acc = ge_identity();
    for (u32 b = lo; b < hi; b++) acc = ge_add(acc, ge_load(block_tot + 64ull * b + 32 * which));
    ge incl = block_scan_ge(acc, lds);
    ge_store(lds + 32 * t, incl);
    __syncthreads();
    ge run = t ? ge_load(lds + 32 * (t - 1)) : ge_identity();
    for (u32 b = lo; b < hi; b++) {
        ge v = ge_load(block_tot + 64ull * b + 32 * which);
        ge_store(block_tot + 64ull * b + 32 * which, run);
        run = ge_add(run, v);
    }
}
// phase 3: global inclusive prefixes at the operations that need them
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const dev_op *__restrict__ ops, u32 n_ops, const u32 *__restrict__ pts, const u32 *__restrict__ gtab,
                                                           const u32 *__restrict__ thread_excl, const u32 *__restrict__ block_excl,
                                                           const u32 *__restrict__ slot_last_op, u32 *__restrict__ slot_base, u32 *__restrict__ slot_final,
                                                           u32 *__restrict__ emit_incl) {
    wave_priority_high();
    const u32 gt = blockIdx.x * SCAN_BLOCK + threadIdx.x, which = blockIdx.y, h = 32 * which;
    if (gt * SCAN_K >= n_ops) return;
    ge acc = ge_add(ge_load(block_excl + 64ull * blockIdx.x + h), ge_load(thread_excl + 64ull * gt + h));
    for (u32 k = 0; k < SCAN_K; k++) {
        u32 o = gt * SCAN_K + k;
        if (o >= n_ops) break;
        dev_op op = ops[o];
        if (op.first) ge_store(slot_base + 64ull * op.slot + h, acc);          // prefix before the slot's initial balance
        scan_apply_op(acc, op, which, pts, gtab);
        if (op.emit != 0xffffffffu) ge_store(emit_incl + 64ull * op.emit + h, acc);
        if (slot_last_op[op.slot] == o) ge_store(slot_final + 64ull * op.slot + h, acc);
    }
}
// phase 4a: sender ciphertexts read by the equality proofs, as affine-Niels points
__global__ void __launch_bounds__(256, 2) k_scan_emit(const u32 *__restrict__ emit_incl, const u32 *__restrict__ emit_slot, const u32 *__restrict__ slot_base,
                                                      u32 n_emit, u32 emit_pt_base, u32 *__restrict__ pts, u8 *__restrict__ ok) {
    wave_priority_high();
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per point (2 per emitted ciphertext)
    if (g >= 2 * n_emit) return;
    u32 e = g >> 1, which = g & 1;
    ge incl = ge_load(emit_incl + 64ull * e + 32 * which);
    ge base = ge_load(slot_base + 64ull * emit_slot[e] + 32 * which);
    ge r = ge_sub(incl, base);
    ge_niels_store(pts + 32ull * (emit_pt_base + g), ge_to_niels_normalize(r));
    ok[emit_pt_base + g] = 1;
}
// phase 4b: final balance of every touched slot, compressed
__global__ void __launch_bounds__(256, 2) k_scan_final(const u32 *__restrict__ slot_final, const u32 *__restrict__ slot_base, u32 n_slots, u32 *__restrict__ out_words) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n_slots) return;
    u32 s = g >> 1, which = g & 1;
    ge r = ge_sub(ge_load(slot_final + 64ull * s + 32 * which), ge_load(slot_base + 64ull * s + 32 * which));
    u32 w[8]; ristretto_encode_words(w, r);
#pragma unroll
    for (int i = 0; i < 8; i++) out_words[8ull * g + i] = w[i];
}


void launch_ledger_scan(hipStream_t st, const dev_op *ops, u32 n_ops, const u32 *pts_in, const u32 *gtab, u32 *thread_excl, u32 *block_tot, const u32 *slot_last_op,
                        u32 *slot_base, u32 *slot_final, u32 *emit_incl, const u32 *emit_slot, u32 n_emit, u32 emit_pt_base, u32 *pts, u8 *ok) {
    if (!n_ops) return;
    u32 scan_threads = (n_ops + SCAN_K - 1) / SCAN_K, scan_blocks = (scan_threads + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(k_scan_local, dim3(scan_blocks, 2), dim3(SCAN_BLOCK), 0, st, ops, n_ops, pts_in, gtab, thread_excl, block_tot);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1, 2), dim3(SCAN_BLOCK), 0, st, block_tot, scan_blocks);
    hipLaunchKernelGGL(k_scan_apply, dim3(scan_blocks, 2), dim3(SCAN_BLOCK), 0, st, ops, n_ops, pts_in, gtab, thread_excl, block_tot, slot_last_op, slot_base, slot_final,
                       emit_incl);
    if (n_emit) hipLaunchKernelGGL(k_scan_emit, dim3(grid_for(2 * n_emit, 256)), dim3(256), 0, st, emit_incl, emit_slot, slot_base, n_emit, emit_pt_base, pts, ok);
}
void launch_ledger_final(hipStream_t st, const u32 *slot_final, const u32 *slot_base, u32 n_slots, u32 *final_cts) {
    if (n_slots) hipLaunchKernelGGL(k_scan_final, dim3(grid_for(2 * n_slots, 256)), dim3(256), 0, st, slot_final, slot_base, n_slots, final_cts);
}

// The ciphertexts a batch leaves in the slots it used, written over the ledger's array of all ciphertexts (64 bytes per balance,
// in the ledger view's order): what the next part of a batch verified in consecutive parts starts from.
__global__ void __launch_bounds__(256) k_ledger_carry(const u32 *__restrict__ final_cts, const u32 *__restrict__ slot_of_used, u32 n_slots, u32 *__restrict__ ledger_cts) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 16 * n_slots) return;
    ledger_cts[16ull * slot_of_used[g >> 4] + (g & 15)] = final_cts[g];
}
void launch_ledger_carry(hipStream_t st, const u32 *final_cts, const u32 *slot_of_used, u32 n_slots, u32 *ledger_cts) {
    if (n_slots) hipLaunchKernelGGL(k_ledger_carry, dim3(grid_for(16ull * n_slots, 256)), dim3(256), 0, st, final_cts, slot_of_used, n_slots, ledger_cts);
}

}  // namespace xhe
