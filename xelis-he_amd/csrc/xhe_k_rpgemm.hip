// Column sums of the range proofs' generator scalars as an integer matrix product on the matrix cores.
//
// Every aggregated range proof of a batch multiplies the same generators G_i, H_i (bulletproofs RangeProof::verify_multiple,
// called at /root/reference/src/tx/verify.rs:504-513), so the batch needs, per generator, the SUM over the transactions of
// that generator's scalar.  Each scalar is a sum of products over the bits of i (table above rp_prep_terms in
// xhe_k_scalars.hip), and a product over the bits of i = (hi, lo) is one factor from the low bits times one from the high
// bits.  Hence, per product kind P in {s, t, e},
//     sum over tx of P_i(tx) = sum over tx of Hi_P(tx)[hi] * Lo_P(tx)[lo]        (mod l)
// is entry (hi, lo) of the matrix product  Hi_P^T Lo_P  over the transactions: 2^h + 2^l field products per transaction
// and kind instead of 2^lg.  The matrix product itself runs on integers: every table entry (a residue below 2^253 in
// Montgomery form) is written as 32 signed base-256 digits, the digit matrices are multiplied with v_mfma_i32_32x32x32_i8
// (a 32 x 32 tile is exactly the digit-by-digit product of one (hi, lo) pair, summed over 32 transactions per
// instruction), the tile's anti-diagonals are summed (digit a of Hi times digit b of Lo weighs 256^(a+b)), and one Horner
// pass per generator turns the 63 diagonal sums back into a scalar.  The sum of integers is exact, so the result is the
// same residue the element-by-element kernel (k_rp_matrix) produces.
//
// Three kernels per pass over a chunk of transactions: k_rp_digits (tables -> digits, in the matrix instruction's fragment
// order), k_rp_gemm (split over tiles of the output, kinds and slices of the transactions), k_rp_gemm_finish.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Digit table of a chunk: [kind 0..2][entry 0..per-1: the 2^l low-bit entries, then the 2^h high-bit entries][k-block][1024],
// a k-block being 32 transactions x 32 digits in fragment order: byte 16 * (a + 32 * ((k >> 4) & 1)) + (k & 15) is digit a of
// transaction k of the block, so lane L of a wavefront reads its operand (row or column a = L & 31, sixteen transactions)
// as the 16 bytes at 16 L.
__device__ __forceinline__ u64 rpg_block_off(u32 kind, u32 per, u32 entry, u32 kb_stride, u32 kb) { return ((u64)(kind * per + entry) * kb_stride + kb) * 1024ull; }

// thread per (transaction of the chunk, kind, group of eight entries); a wavefront covers 64 consecutive transactions
__global__ void __launch_bounds__(64) k_rp_digits(const u32 *__restrict__ group_txs, u32 n_rows, u32 lg, u32 kb_stride, const u32 *__restrict__ rp_ctx,
                                                  const u32 *__restrict__ rp_err, u8 *__restrict__ dig) {
    wave_priority_mid();
    const u32 l = (lg + 1) / 2, h = lg - l, n_lo = 1u << l, per = n_lo + (1u << h), groups = per / 8;
    const u32 k = blockIdx.x * 64 + threadIdx.x;
    const u32 kind = blockIdx.y / groups, q0 = (blockIdx.y % groups) * 8;
    const bool high = q0 >= n_lo;
    const u32 bits0 = high ? q0 - n_lo : q0, base = high ? l : 0, nb = high ? h : l;
    if ((k >> 5) >= kb_stride) return;
    const u32 tx = k < n_rows ? group_txs[k] : 0;
    const bool live = k < n_rows && !rp_err[tx];
    sc e[8];
    if (live) {
        const u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * tx;
        // kind 0: s_i = coef_G prod_(b in i) U2[b];  1: coef_H3 prod_(b not in i) YU[b];  2: coef_H2 prod_(b in i) E[b] prod_(b not in i) Y2[b]
        const u32 f_set = kind == 0 ? RPC_U2 : RPC_E, f_unset = kind == 1 ? RPC_YU : RPC_Y2;
        const bool has_set = kind != 1, has_unset = kind != 0;
        sc start = high ? load_sc_limbs(ctx + 9 * (kind == 0 ? RPC_COEF_G : kind == 1 ? RPC_COEF_H3 : RPC_COEF_H2)) : sc_one();
        for (u32 b = 3; b < nb; b++) {
            const bool set = (bits0 >> b) & 1;
            if (set ? has_set : has_unset) start = sc_mul(start, load_sc_limbs(ctx + 9 * ((set ? f_set : f_unset) + base + b)));
        }
        e[0] = start;
#pragma unroll
        for (u32 b = 0; b < 3; b++) {
            sc fs = sc_one(), fu = sc_one();
            if (has_set) fs = load_sc_limbs(ctx + 9 * (f_set + base + b));
            if (has_unset) fu = load_sc_limbs(ctx + 9 * (f_unset + base + b));
#pragma unroll
            for (u32 j = 0; j < (1u << b); j++) {
                e[j + (1u << b)] = has_set ? sc_mul(e[j], fs) : e[j];
                if (has_unset) e[j] = sc_mul(e[j], fu);
            }
        }
    }
    u8 *dst = dig + ((k >> 4) & 1) * 512 + (k & 15);
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        u8 *d = dst + rpg_block_off(kind, per, q0 + j, kb_stride, k >> 5);
        if (!live) {
#pragma unroll
            for (u32 a = 0; a < 32; a++) d[16 * a] = 0;
            continue;
        }
        // every entry is a Montgomery product's output or a canonical constant: below 2^254, limbs normalised
        u32 w[8]; sc_words_from_limbs(w, sc_norm(e[j]));
        u32 carry = 0;
#pragma unroll
        for (u32 a = 0; a < 32; a++) {
            u32 v = ((w[a >> 2] >> ((a & 3) * 8)) & 255) + carry;
            carry = v >= 128 ? 1u : 0u;              // digit v - 256 and a carry into the next one
            d[16 * a] = (u8)v;
        }
    }
}

// Workgroup: an 8 x 8 tile of (hi, lo) pairs (wavefront: 4 x 4), one kind, one slice of the chunk's k-blocks.  Output: per pair
// the 63 anti-diagonal sums of its 32 x 32 digit-product tile, bins[kind][slice][hi][lo][64].
__global__ void __launch_bounds__(256) k_rp_gemm(const u8 *__restrict__ dig, u32 lg, u32 kb_stride, u32 kb_per_slice, int *__restrict__ bins) {
    __shared__ int diag[4][16][64];
    const u32 l = (lg + 1) / 2, h = lg - l, n_lo = 1u << l, n_hi = 1u << h, per = n_lo + n_hi;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 tiles_n = n_lo / 8, tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const u32 hi0 = tile_m * 8 + (wave >> 1) * 4, lo0 = tile_n * 8 + (wave & 1) * 4;
    const u32 kind = blockIdx.z, slice = blockIdx.y;
    const u32 kb0 = slice * kb_per_slice, kb1 = kb0 + kb_per_slice < kb_stride ? kb0 + kb_per_slice : kb_stride;
    const u64 entry_stride = (u64)kb_stride * 1024ull;
    const u8 *pa = dig + rpg_block_off(kind, per, n_lo + hi0, kb_stride, kb0) + 16 * lane;
    const u8 *pb = dig + rpg_block_off(kind, per, lo0, kb_stride, kb0) + 16 * lane;
    v16i acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
    v4i a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = *(const v4i *)(pa + i * entry_stride); b[i] = *(const v4i *)(pb + i * entry_stride); }
    for (u32 kb = kb0; kb < kb1; kb++) {
        v4i an[4], bn[4];
        // the next k-block's operands travel while this one's products run (the last iteration re-reads its own block)
        const u64 step = kb + 1 < kb1 ? 1024ull : 0ull;
        pa += step; pb += step;
#pragma unroll
        for (int i = 0; i < 4; i++) { an[i] = *(const v4i *)(pa + i * entry_stride); bn[i] = *(const v4i *)(pb + i * entry_stride); }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = an[i]; b[i] = bn[i]; }
    }
    // anti-diagonals: element (row a, column b) of a tile sits in lane b + 32 * ((a >> 2) & 1), register (a & 3) + 4 * (a >> 3)
    for (u32 i = lane; i < 16 * 64; i += 64) (&diag[wave][0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32 row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
                atomicAdd(&diag[wave][i * 4 + j][row + col], acc[i][j][r]);
            }
    __syncthreads();
    const u32 slices = gridDim.y;
    for (u32 t = 0; t < 16; t++) {
        const u32 hi = hi0 + (t >> 2), lo = lo0 + (t & 3);
        bins[((((u64)kind * slices + slice) * n_hi + hi) * n_lo + lo) * 64 + lane] = diag[wave][t][lane];
    }
}

// sum of rho' z over the live transactions of the group, one partial per workgroup
__global__ void __launch_bounds__(256) k_rp_rho_partial(const u32 *__restrict__ group_txs, u32 n_rows, const u32 *__restrict__ rp_ctx, const u32 *__restrict__ rp_err,
                                                        u32 *__restrict__ out) {
    wave_priority_mid();
    __shared__ u32 lds[256 * 9];
    const u32 t = threadIdx.x;
    const sc one = sc_one();
    sc acc = sc_zero(); u32 cnt = 0;
    for (u32 r = blockIdx.x * 256 + t; r < n_rows; r += gridDim.x * 256) {
        const u32 tx = group_txs[r];
        if (rp_err[tx]) continue;
        acc = sc_add(acc, load_sc_limbs(rp_ctx + 9ull * RPC_SCALARS * tx + 9 * RPC_RHO_Z));
        if (++cnt == 16) { acc = sc_mul(acc, one); cnt = 0; }
    }
    acc = sc_mul(acc, one);
    for (u32 d = 128; d >= 1; d >>= 1) {
        store_sc_limbs(lds + 9 * t, acc);
        __syncthreads();
        if (t < d) acc = sc_mul(sc_add(acc, load_sc_limbs(lds + 9 * (t + d))), one);
        __syncthreads();
    }
    if (t == 0) store_sc_limbs(out + 9ull * blockIdx.x, acc);
}

// Wavefront per generator (side 0: G_i from kind 0; side 1: H_i from kinds 1 and 2), lane d per anti-diagonal: its sum over
// slices and kinds times 256^d, the 63 terms added with wavefront shuffles, out of the doubled Montgomery factor, plus (first
// pass of a group) -+ the sum of rho' z; acc[] += that.
__global__ void __launch_bounds__(256) k_rp_gemm_finish(const int *__restrict__ bins, u32 lg, u32 slices, const u32 *__restrict__ rho_part, u32 n_rho, u32 nm_cap,
                                                        u32 *__restrict__ acc) {
    wave_priority_mid();
    const u32 l = (lg + 1) / 2, h = lg - l, n_lo = 1u << l, n_hi = 1u << h, nm = 1u << lg;
    const u32 g = blockIdx.x * 4 + (threadIdx.x >> 6), d = threadIdx.x & 63;
    if (g >= 2 * nm) return;                           // whole wavefronts leave
    const u32 side = g >= nm ? 1u : 0u, idx = g & (nm - 1), hi = idx >> l, lo = idx & (n_lo - 1);
    const sc minus_one = sc_c_MINUS_R();
    long long s = 0;
    for (u32 kind = side; kind <= 2 * side; kind++)
        for (u32 sl = 0; sl < slices; sl++) s += bins[((((u64)kind * slices + sl) * n_hi + hi) * n_lo + lo) * 64 + d];
    if (d == 63) s = 0;                                // the 64th bin is padding
    const u64 mag = s < 0 ? (u64)(-s) : (u64)s;
    sc term = sc_zero();                               // the plain integer |s| (it multiplies the residue itself, not its Montgomery form)
    term.v[0] = (u32)mag & SC_MASK29; term.v[1] = (u32)(mag >> 29) & SC_MASK29; term.v[2] = (u32)(mag >> 58);
    // 256^d in Montgomery form by square and multiply over the six bits of d
    sc pw = sc_one(), x = sc_from_u64(256);
#pragma unroll 1
    for (u32 b = 0; b < 6; b++) {
        const sc px = sc_mul(pw, x);
#pragma unroll
        for (int k = 0; k < 9; k++) pw.v[k] = ((d >> b) & 1) ? px.v[k] : pw.v[k];
        x = sc_mul(x, x);
    }
    sc v = sc_mul(term, pw);                           // |s| 256^d as a plain residue
    if (s < 0) v = sc_mul(v, minus_one);
#pragma unroll 1
    for (u32 step = 32; step >= 1; step >>= 1) {
        sc o;
#pragma unroll
        for (int k = 0; k < 9; k++) o.v[k] = __shfl_down(v.v[k], step, 64);
        v = sc_mul(sc_add(v, o), sc_one());
    }
    if (d != 0) return;
    // v = (Hi R)(Lo R) summed = R^2 * sum: one division by R leaves the sum's Montgomery form
    sc one_raw = sc_zero(); one_raw.v[0] = 1;
    v = sc_mul(v, one_raw);
    if (n_rho) {
        sc rho = sc_zero();
        for (u32 i = 0; i < n_rho; i++) rho = sc_mul(sc_add(rho, load_sc_limbs(rho_part + 9ull * i)), sc_one());
        v = sc_add(v, side ? rho : sc_mul(rho, minus_one));
    }
    u32 *dst = acc + 9ull * (side ? nm_cap + idx : idx);
    store_sc_limbs(dst, sc_norm(sc_mul(sc_add(v, load_sc_limbs(dst)), sc_one())));
}

void launch_rp_gemm(hipStream_t st, const u32 *group_txs, u32 n_rows, u32 lg, const u32 *rp_ctx, const u32 *rp_err, u8 *dig, int *bins, u32 *rho_part, u32 nm_cap, u32 *acc) {
    if (!n_rows) return;
    const rp_gemm_plan p = rp_gemm_plan_for(n_rows, lg);
    const u32 n_rho = std::min<u32>(RP_GEMM_RHO_PARTS, grid_for(n_rows, 256));
    hipLaunchKernelGGL(k_rp_rho_partial, dim3(n_rho), dim3(256), 0, st, group_txs, n_rows, rp_ctx, rp_err, rho_part);
    const u32 tiles = (p.n_hi / 8) * (p.n_lo / 8);
    for (u32 r0 = 0; r0 < n_rows; r0 += p.chunk_rows) {
        const u32 n = std::min<u32>(p.chunk_rows, n_rows - r0), kb = (n + 31) / 32, slices = (kb + p.kb_per_slice - 1) / p.kb_per_slice;
        hipLaunchKernelGGL(k_rp_digits, dim3(grid_for(32ull * kb, 64), 3 * (p.per / 8)), dim3(64), 0, st, group_txs + r0, n, lg, kb, rp_ctx, rp_err, dig);
        hipLaunchKernelGGL(k_rp_gemm, dim3(tiles, slices, 3), dim3(256), 0, st, dig, lg, kb, p.kb_per_slice, bins);
        hipLaunchKernelGGL(k_rp_gemm_finish, dim3(grid_for(2ull << lg, 4)), dim3(256), 0, st, bins, lg, slices, rho_part, r0 == 0 ? n_rho : 0u, nm_cap, acc);
    }
}

}  // namespace xhe
