// Point kernels: batched ristretto decode/encode, constant tables (B, H, Bulletproofs generators, fixed-base
// tables), decode of a batch's point list, and the error-precedence kernels.
#include <stdlib.h>
#include "xhe_settings.h"
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

__global__ void __launch_bounds__(256, 2) k_decode_xyzt(const u32 *__restrict__ in, u64 n, u32 *__restrict__ out, u8 *__restrict__ ok) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = in[8 * i + k];
    ge p; bool good = ristretto_decode_words(p, w);
    ge_store(out + 32 * i, p);
    ok[i] = good ? 1 : 0;
}
__global__ void __launch_bounds__(256, 2) k_decode_niels(const u32 *__restrict__ in, u64 n, u32 *__restrict__ out, u8 *__restrict__ ok) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = in[8 * i + k];
    ge p; bool good = ristretto_decode_words(p, w);
    ge_niels_store(out + 32 * i, ge_to_niels_affine(p));
    ok[i] = good ? 1 : 0;
}
__global__ void __launch_bounds__(256, 2) k_encode_xyzt(const u32 *__restrict__ in, u64 n, u32 *__restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ge p = ge_load(in + 32 * i);
    u32 w[8]; ristretto_encode_words(w, p);
#pragma unroll
    for (int k = 0; k < 8; k++) out[8 * i + k] = w[k];
}
// 32-byte little-endian integers -> canonical scalars (value mod l), in place
__global__ void __launch_bounds__(256) k_reduce_scalars(u32 *__restrict__ s, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = s[8 * i + k];
    sc x = sc_from_words(w);
    sc_to_words(w, x);
#pragma unroll
    for (int k = 0; k < 8; k++) s[8 * i + k] = w[k];
}
__global__ void k_count_bad(const u8 *__restrict__ ok, u64 n, u32 *__restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !ok[i]) atomicAdd(out, 1u);
}

// B = ristretto basepoint, H = from_uniform_bytes(SHA3-512(enc(B))) (src/elgamal.rs:16-24), as affine Niels
__global__ void k_base_points(u32 *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u8 bb[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                       0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge B; ristretto_decode(B, bb);
    ge_niels_store(out, ge_to_niels_affine(B));
    u8 h[64]; sha3_512(h, bb, 32);
    ge H = ristretto_from_uniform_bytes(h);
    ge_niels_store(out + 32, ge_to_niels_normalize(H));
}
// BulletproofGens (src/proofs.rs:20): thread per (party, G|H) walks its SHAKE256 stream, 64 generators
__global__ void __launch_bounds__(64) k_bp_generators(u32 party0, u32 n_parties, u32 cap, u32 *__restrict__ static_pts) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n_parties) return;
    u32 party = party0 + (g >> 1), which = g & 1;
    sponge sp; sp.init(136, 0x1f);
    const u8 tag[15] = {'G', 'e', 'n', 'e', 'r', 'a', 't', 'o', 'r', 's', 'C', 'h', 'a', 'i', 'n'};
    sp.absorb(tag, 15);
    u8 label[5] = {(u8)(which ? 'H' : 'G'), (u8)party, (u8)(party >> 8), (u8)(party >> 16), (u8)(party >> 24)};
    sp.absorb(label, 5);
    for (u32 i = 0; i < 64; i++) {
        u8 ub[64]; sp.squeeze(ub, 64);
        ge p = ristretto_from_uniform_bytes(ub);
        u32 idx = 2 + (which ? cap : 0) + 64 * party + i;
        ge_niels_store(static_pts + 32ull * idx, ge_to_niels_normalize(p));
    }
}


// ---- decode ----------------------------------------------------------------------------------------
// Workgroups of a decode pass at most (grid-stride beyond).  Measured, resident step: 100 000 x 8 transfers 22.2 ms at 256,
// 19.4 ms at 512, 19.0 ms at 2048; 125 000 x 1 transfer 11.4 / 10.3 / 9.8 ms; the mixed batch 11.4 / 10.8 / 11.0 ms; the parts of
// a batch verified in parts (two in flight, each with its own decode passes) are not faster with more than 512.
static constexpr u32 DECODE_MAX_BLOCKS = 2048, DECODE_MAX_BLOCKS_SHARED = 512;
// A pass over the n listed entries of decs[] (the points the ledger scan waits for, the public keys, or all the others).
// OCC = wavefronts per SIMD the kernel is compiled for: 2 (256 registers) or 4 (128 registers, a little spilling outside the
// inverse square root's squaring chain): alone the two run within 2 % of each other, but two 256-register wavefronts fill a
// SIMD's register file and keep every other kernel of the verification off it until they end.
template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_decode_list(const u8 *__restrict__ bytes, const dev_dec *__restrict__ decs, const u32 *__restrict__ list, u32 n, u32 pt_base,
                                                          u32 *__restrict__ pts, u8 *__restrict__ ok, u32 background) {
    if (background) wave_priority_mid(); else wave_priority_upper();  // the first two head the ledger scan and the signatures; the bulk only has to beat the accumulation
    for (u32 g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
        u32 i = list[g];
        u32 w[8]; load_words_unaligned(w, bytes + decs[i].off);
        ge p; bool good = ristretto_decode_words(p, w);
        ge_niels_store(pts + 32ull * (pt_base + i), ge_to_niels_affine(p));
        ok[pt_base + i] = good ? 1 : 0;
    }
}


// Fixed-base tables: tab[(base * n_windows + w) * 256 + b] = b * 2^(8w) * P_base as affine Niels, for
// n_bases points given as affine Niels records.  Thread per entry.
__global__ void __launch_bounds__(256, 2) k_build_fixed_table(const u32 *__restrict__ base_niels, u32 n_bases, u32 n_windows, u32 *__restrict__ tab) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (u64)n_bases * n_windows * 256) return;
    u32 b = (u32)g & 255, w = (u32)((g >> 8) % n_windows), base = (u32)((g >> 8) / n_windows);
    ge p = ge_from_niels_affine(ge_niels_load(base_niels + 32ull * base));
    for (u32 k = 0; k < 8 * w; k++) p = ge_double(p);
    ge r = ge_identity();
    for (int bit = 7; bit >= 0; bit--) { r = ge_double(r); if ((b >> bit) & 1) r = ge_add(r, p); }
    ge_niels_store(tab + 32ull * g, ge_to_niels_normalize(r));
}


// ---- error precedence --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_point_errors(const dev_tx *__restrict__ txs, const dev_cm *__restrict__ cms, const dev_xfer *__restrict__ xfers, u32 n_txs,
                                                      const u8 *__restrict__ ok, u32 *__restrict__ tx_err, u32 *__restrict__ rp_err) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_txs) return;
    const dev_tx d = txs[i];
    u32 key = ERRKEY_NONE;
    auto raise = [&](u32 k) { if (k < key) key = k; };
    for (u32 k = 0; k < d.n_transfers; k++) {
        const dev_xfer x = xfers[d.xfer_first + k];
        if (!ok[x.pt] || !ok[x.pt + 1] || !ok[x.pt + 2]) raise(errkey(ST_DECOMPRESS, 0, 0, XHE_ERR_DECOMPRESSION));
        if (!ok[x.dest_pt]) raise(errkey(ST_TRANSFER, k, 0, XHE_ERR_DECOMPRESSION));
        if (x.first_touch && !x.state_err && (!ok[x.slot_pt] || !ok[x.slot_pt + 1])) raise(errkey(ST_TRANSFER, k, 2, XHE_ERR_DECOMPRESSION));
        if (!ok[x.pt + 3] || !ok[x.pt + 4] || !ok[x.pt + 5]) raise(errkey(ST_TRANSFER, k, 4, XHE_ERR_CT_VALIDITY));
    }
    for (u32 k = 0; k < d.n_commitments; k++) {
        const dev_cm c = cms[d.cm_first + k];
        if (!ok[c.pt]) raise(errkey(ST_DECOMPRESS, 0, 0, XHE_ERR_DECOMPRESSION));
        if (c.first_touch && (!ok[c.slot_pt] || !ok[c.slot_pt + 1])) raise(errkey(ST_COMMITMENT, k, 1, XHE_ERR_DECOMPRESSION));
        if (!ok[c.pt + 1] || !ok[c.pt + 2] || !ok[c.pt + 3]) raise(errkey(ST_COMMITMENT, k, 3, XHE_ERR_COMMITMENT_EQ));
    }
    if (!ok[d.src_pt]) raise(errkey(ST_DECOMPRESS, 0, 0, XHE_ERR_DECOMPRESSION));
    if (key != ERRKEY_NONE) atomicMin(&tx_err[i], key);
    bool rp_bad = false;
    for (u32 k = 0; k < 4u + 2u * d.lg; k++) rp_bad |= !ok[d.rp_pt + k];
    if (rp_bad) rp_err[i] = 1;
}
// out[0..1]: (first failing tx << 32 | key) as u64 min; out[2]: first tx with a range-proof failure
// out[3]: first transaction with a scalar that is not canonical (found when the batch was uploaded), or 0xffffffff
__global__ void __launch_bounds__(256) k_verdict(const u32 *__restrict__ tx_err, const u32 *__restrict__ rp_err, u32 n_txs, unsigned long long *__restrict__ first_err,
                                                 u32 *__restrict__ first_rp, const u32 *__restrict__ noncanonical_first) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) first_rp[1] = noncanonical_first[0];
    if (i >= n_txs) return;
    u32 k = tx_err[i];
    if (k != ERRKEY_NONE) atomicMin(first_err, ((unsigned long long)i << 32) | k);
    if (rp_err[i]) atomicMin(first_rp, i);
}


__global__ void __launch_bounds__(256) k_setup(setup_jobs jobs) {
    const int j = blockIdx.y;
    u32 *dst = jobs.dst[j]; const u32 *src = jobs.src[j]; const u64 n = jobs.n_words[j]; const u32 v = jobs.value[j];
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) dst[i] = src ? src[i] : v;
    if (j == 0 && blockIdx.x == 0 && threadIdx.x < 8 && jobs.seed_dst) jobs.seed_dst[threadIdx.x] = jobs.seed[threadIdx.x];
}
void launch_setup(hipStream_t st, const setup_jobs &jobs) {
    if (jobs.n <= 0) return;
    u64 most = 1;
    for (int j = 0; j < jobs.n; j++) if (jobs.n_words[j] > most) most = jobs.n_words[j];
    unsigned bx = grid_for(most, 256 * 4); if (bx > 1024) bx = 1024; if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_setup, dim3(bx, jobs.n), dim3(256), 0, st, jobs);
}
void launch_decode_xyzt(hipStream_t st, const u32 *in, u64 n, u32 *out, u8 *ok) { if (n) hipLaunchKernelGGL(k_decode_xyzt, dim3(grid_for(n, 256)), dim3(256), 0, st, in, n, out, ok); }
void launch_decode_niels(hipStream_t st, const u32 *in, u64 n, u32 *out, u8 *ok) { if (n) hipLaunchKernelGGL(k_decode_niels, dim3(grid_for(n, 256)), dim3(256), 0, st, in, n, out, ok); }
void launch_encode_xyzt(hipStream_t st, const u32 *in, u64 n, u32 *out) { if (n) hipLaunchKernelGGL(k_encode_xyzt, dim3(grid_for(n, 256)), dim3(256), 0, st, in, n, out); }
void launch_reduce_scalars(hipStream_t st, u32 *s, u64 n) { if (n) hipLaunchKernelGGL(k_reduce_scalars, dim3(grid_for(n, 256)), dim3(256), 0, st, s, n); }
void launch_count_bad(hipStream_t st, const u8 *ok, u64 n, u32 *out) { if (n) hipLaunchKernelGGL(k_count_bad, dim3(grid_for(n, 256)), dim3(256), 0, st, ok, n, out); }
void launch_base_points(hipStream_t st, u32 *out) { hipLaunchKernelGGL(k_base_points, dim3(1), dim3(64), 0, st, out); }
void launch_bp_generators(hipStream_t st, u32 n_parties, u32 cap, u32 *static_pts) {
    if (n_parties) hipLaunchKernelGGL(k_bp_generators, dim3(grid_for(2 * n_parties, 64)), dim3(64), 0, st, 0u, n_parties, cap, static_pts);
}
void launch_build_fixed_table(hipStream_t st, const u32 *base_niels, u32 n_bases, u32 n_windows, u32 *tab) {
    if (n_bases) hipLaunchKernelGGL(k_build_fixed_table, dim3(grid_for((u64)n_bases * n_windows * 256, 256)), dim3(256), 0, st, base_niels, n_bases, n_windows, tab);
}
void launch_decode_list(hipStream_t st, const u8 *bytes, const dev_dec *decs, const u32 *list, u32 n, u32 pt_base, u32 *pts, u8 *ok, bool background, bool wide) {
    // a bounded grid (grid-stride loop): the decode is the widest kernel of a verification and runs beside the critical path,
    // whose few long-running wavefronts must find free slots the moment they are launched
    const u32 max_blocks = setting("XHE_DECODE_MAX_BLOCKS") ? (u32)strtoul(setting("XHE_DECODE_MAX_BLOCKS"), nullptr, 10) : (wide ? DECODE_MAX_BLOCKS : DECODE_MAX_BLOCKS_SHARED);
    const bool occ2 = setting("XHE_DECODE_OCC") && setting("XHE_DECODE_OCC")[0] == '2';
    u32 blocks = grid_for(n, 256);
    if (blocks > max_blocks) blocks = max_blocks;
    if (!n) return;
    if (occ2) hipLaunchKernelGGL(k_decode_list<2>, dim3(blocks), dim3(256), 0, st, bytes, decs, list, n, pt_base, pts, ok, background ? 1u : 0u);
    else hipLaunchKernelGGL(k_decode_list<4>, dim3(blocks), dim3(256), 0, st, bytes, decs, list, n, pt_base, pts, ok, background ? 1u : 0u);
}
void launch_point_errors(hipStream_t st, const dev_tx *txs, const dev_cm *cms, const dev_xfer *xfers, u32 n_txs, const u8 *ok, u32 *tx_err, u32 *rp_err) {
    if (n_txs) hipLaunchKernelGGL(k_point_errors, dim3(grid_for(n_txs, 256)), dim3(256), 0, st, txs, cms, xfers, n_txs, ok, tx_err, rp_err);
}
void launch_verdict(hipStream_t st, const u32 *tx_err, const u32 *rp_err, u32 n_txs, unsigned long long *first_err, u32 *first_rp, const u32 *noncanonical_first) {
    if (n_txs) hipLaunchKernelGGL(k_verdict, dim3(grid_for(n_txs, 256)), dim3(256), 0, st, tx_err, rp_err, n_txs, first_err, first_rp, noncanonical_first);
}

}  // namespace xhe
