// Proof scalars: sigma-proof terms (src/proofs.rs:188-208,336-358), range-proof verification scalars and
// their sums over the batch, scalars of the static generators.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

// ---- sigma proofs ----------------------------------------------------------------------------------
// gh[2p], gh[2p+1]: this proof's contributions to the scalars of G and H (Montgomery limbs)
__global__ void __launch_bounds__(256, 2) k_sigma_cm(const u8 *__restrict__ bytes, const dev_cm *__restrict__ cms, const dev_tx *__restrict__ txs, u32 n,
                                                     const u32 *__restrict__ chal, const u32 *__restrict__ seed, u32 *__restrict__ scalars,
                                                     u32 *__restrict__ term_point, u32 *__restrict__ gh) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev_cm c = cms[i];
    const dev_tx d = txs[c.tx];
    const u8 *p = bytes + c.off;
    sc ch_c = load_scalar_words(chal + 8ull * (d.chal_off + 2 * c.idx)), ch_w = load_scalar_words(chal + 8ull * (d.chal_off + 2 * c.idx + 1));
    sc z_s = load_scalar_bytes(p + 160), z_x = load_scalar_bytes(p + 192), z_r = load_scalar_bytes(p + 224);
    sc bf = prf_scalar(seed, 1, i);
    sc ww = sc_mul(ch_w, ch_w);
    sc w_bf = sc_mul(ch_w, bf), ww_bf = sc_mul(ww, bf);
    sc neg_bf = sc_neg(bf), neg_w_bf = sc_neg(w_bf), neg_ww_bf = sc_neg(ww_bf);
    u32 *s = scalars + 8ull * c.term_off;
    u32 *tp = term_point + c.term_off;
    store_scalar_words(s, sc_mul(z_s, bf));                  tp[0] = d.src_pt;          // z_s * P_source
    store_scalar_words(s + 8, neg_bf);                        tp[1] = c.pt + 1;          // -Y_0
    store_scalar_words(s + 16, sc_mul(w_bf, z_s));            tp[2] = c.src_ct_pt + 1;   // w z_s * D_source
    store_scalar_words(s + 24, sc_mul(neg_w_bf, ch_c));       tp[3] = c.src_ct_pt;       // -w c * C_source
    store_scalar_words(s + 32, neg_w_bf);                     tp[4] = c.pt + 2;          // -w * Y_1
    store_scalar_words(s + 40, sc_mul(neg_ww_bf, ch_c));      tp[5] = c.pt;              // -ww c * C_destination
    store_scalar_words(s + 48, neg_ww_bf);                    tp[6] = c.pt + 3;          // -ww * Y_2
    // G: (w + ww) z_x bf ; H: (-c + ww z_r) bf
    sc g = sc_mul(sc_add(w_bf, ww_bf), z_x);
    sc h = sc_add(sc_mul(neg_bf, ch_c), sc_mul(ww_bf, z_r));
    store_sc_limbs(gh + 18ull * i, sc_norm(g));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(h));
}
__global__ void __launch_bounds__(256, 2) k_sigma_xf(const u8 *__restrict__ bytes, const dev_xfer *__restrict__ xfers, const dev_tx *__restrict__ txs, u32 n,
                                                     const u32 *__restrict__ chal, const u32 *__restrict__ seed, u32 *__restrict__ scalars,
                                                     u32 *__restrict__ term_point, u32 *__restrict__ gh) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev_xfer x = xfers[i];
    const dev_tx d = txs[x.tx];
    const u8 *pr = bytes + x.proof_off;
    u32 cbase = d.chal_off + 2 * d.n_commitments + 2 * x.idx;
    sc ch_c = load_scalar_words(chal + 8ull * cbase), ch_w = load_scalar_words(chal + 8ull * (cbase + 1));
    sc z_r = load_scalar_bytes(pr + 96), z_x = load_scalar_bytes(pr + 128);
    sc bf = prf_scalar(seed, 2, i);
    sc w_bf = sc_mul(ch_w, bf), ww_bf = sc_mul(sc_mul(ch_w, ch_w), bf);
    sc neg_bf = sc_neg(bf), neg_w_bf = sc_neg(w_bf), neg_ww_bf = sc_neg(ww_bf);
    u32 *s = scalars + 8ull * x.term_off;
    u32 *tp = term_point + x.term_off;
    store_scalar_words(s, sc_mul(neg_bf, ch_c));              tp[0] = x.pt;              // -c * C
    store_scalar_words(s + 8, neg_bf);                        tp[1] = x.pt + 3;          // -Y_0
    store_scalar_words(s + 16, sc_mul(w_bf, z_r));            tp[2] = x.dest_pt;         // w z_r * P_dest
    store_scalar_words(s + 24, sc_mul(neg_w_bf, ch_c));       tp[3] = x.pt + 2;          // -w c * D_dest
    store_scalar_words(s + 32, neg_w_bf);                     tp[4] = x.pt + 4;          // -w * Y_1
    store_scalar_words(s + 40, sc_mul(ww_bf, z_r));           tp[5] = d.src_pt;          // ww z_r * P_source
    store_scalar_words(s + 48, sc_mul(neg_ww_bf, ch_c));      tp[6] = x.pt + 1;          // -ww c * D_source
    store_scalar_words(s + 56, neg_ww_bf);                    tp[7] = x.pt + 5;          // -ww * Y_2
    store_sc_limbs(gh + 18ull * i, sc_norm(sc_mul(z_x, bf)));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(sc_mul(z_r, bf)));
}

// sum of n scalars (Montgomery limbs, stride words apart) per residue class j = 0..classes-1:
// out[b*classes + j] = sum over i in block b's share of in[(i*classes + j)]
__device__ __forceinline__ sc sc_reduce_mont(const sc &a) { return sc_mul(a, sc_one()); }
__global__ void __launch_bounds__(256) k_sc_sum_partial(const u32 *__restrict__ in, u32 n, u32 classes, u32 *__restrict__ out) {
    __shared__ u32 lds[256 * 9];
    u32 t = threadIdx.x;
    for (u32 j = 0; j < classes; j++) {
        sc acc = sc_zero(); u32 cnt = 0;
        for (u32 i = blockIdx.x * 256 + t; i < n; i += gridDim.x * 256) {
            acc = sc_add(acc, load_sc_limbs(in + 9ull * ((u64)i * classes + j)));
            if (++cnt == 16) { acc = sc_reduce_mont(acc); cnt = 0; }
        }
        acc = sc_reduce_mont(acc);
        for (u32 d = 128; d >= 1; d >>= 1) {
            store_sc_limbs(lds + 9 * t, acc);
            __syncthreads();
            if (t < d) acc = sc_reduce_mont(sc_add(acc, load_sc_limbs(lds + 9 * (t + d))));
            __syncthreads();
        }
        if (t == 0) store_sc_limbs(out + 9ull * (blockIdx.x * classes + j), acc);
    }
}

// ---- range proofs ----------------------------------------------------------------------------------
// per-transaction context for the matrix kernel, Montgomery limbs (9 words each)
static constexpr u32 RPC_NEG_RHO_Z = 0, RPC_RHO_Z = 1, RPC_COEF_G = 2, RPC_COEF_H2 = 3, RPC_COEF_H3 = 4, RPC_U2 = 5 /* 16 */, RPC_Y2INV = 21 /* 16 */,
                     RPC_E = 37 /* 6 */, RPC_F = 43 /* 10 */;
static_assert(RPC_F + 10 == RPC_SCALARS, "context layout");

__global__ void __launch_bounds__(64) k_rp_prep(const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const dev_cm *__restrict__ cms,
                                                const dev_xfer *__restrict__ xfers, u32 n_txs, const u32 *__restrict__ chal, const u32 *__restrict__ seed,
                                                u32 *__restrict__ scalars, u32 *__restrict__ term_point, u32 *__restrict__ rp_ctx, u32 *__restrict__ gh,
                                                u32 *__restrict__ rp_err) {
    wave_priority_high();
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_txs) return;
    const dev_tx d = txs[i];
    const u32 lg = d.lg, A = d.n_commitments, T = d.n_transfers;
    const u32 n_dyn = 4 + 2 * lg + A + T;
    u32 *s = scalars + 8ull * d.rp_term_off;
    u32 *tp = term_point + d.rp_term_off;
    // term -> point map is valid whatever happens below
    for (u32 k = 0; k < 4 + 2 * lg; k++) tp[k] = d.rp_pt + k;
    for (u32 k = 0; k < A; k++) tp[4 + 2 * lg + k] = cms[d.cm_first + k].pt;
    for (u32 k = 0; k < T; k++) tp[4 + 2 * lg + A + k] = xfers[d.xfer_first + k].pt;
    bool valid = lg >= 6 && lg <= 15 && (64ull * d.m == (1ull << lg)) && d.m <= 512;
    if (!valid) {
        rp_err[i] = 1;
        for (u32 k = 0; k < 8 * n_dyn; k++) s[k] = 0;
        store_sc_limbs(gh + 18ull * i, sc_zero()); store_sc_limbs(gh + 18ull * i + 9, sc_zero());
        return;
    }
    const u8 *rp = bytes + d.rec + d.rp_off;
    const u32 *ch = chal + 8ull * (d.chal_off + 2 * A + 2 * T);
    sc y = load_scalar_words(ch), z = load_scalar_words(ch + 8), x = load_scalar_words(ch + 16), w = load_scalar_words(ch + 24);
    sc t_x = load_scalar_bytes(rp + 128), t_x_bl = load_scalar_bytes(rp + 160), e_bl = load_scalar_bytes(rp + 192);
    sc a = load_scalar_bytes(rp + 224 + 64 * lg), b = load_scalar_bytes(rp + 224 + 64 * lg + 32);
    sc rho = prf_scalar(seed, 3, i), c = prf_scalar(seed, 4, i);
    u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * i;

    // one inversion for y and all u_j: prefix products then back-substitution
    sc u[16], pre[17];
    pre[0] = y;
    for (u32 j = 0; j < lg; j++) { u[j] = load_scalar_words(ch + 32 + 8 * j); pre[j + 1] = sc_mul(pre[j], u[j]); }
    sc inv = sc_invert(pre[lg]);
    sc allinv = sc_one();
    sc rho_c = sc_mul(rho, c);
    for (int j = (int)lg - 1; j >= 0; j--) {
        sc u_inv = sc_mul(inv, pre[j]);          // 1/u_j
        inv = sc_mul(inv, u[j]);
        allinv = sc_mul(allinv, u_inv);
        sc u2 = sc_mul(u[j], u[j]), ui2 = sc_mul(u_inv, u_inv);
        store_scalar_words(s + 8 * (4 + 2 * j), sc_mul(rho, u2));            // L_j
        store_scalar_words(s + 8 * (4 + 2 * j + 1), sc_mul(rho, ui2));       // R_j
        store_sc_limbs(ctx + 9 * (RPC_U2 + (lg - 1 - j)), u2);               // bit (lg-1-j) of the index selects u_j^2
    }
    sc y_inv = inv;                               // after removing every u_j
    sc zz = sc_mul(z, z);
    store_scalar_words(s, rho);                                  // A
    store_scalar_words(s + 8, sc_mul(rho, x));                   // S
    store_scalar_words(s + 16, sc_mul(rho_c, x));                // T_1
    store_scalar_words(s + 24, sc_mul(sc_mul(rho_c, x), x));     // T_2
    {
        sc v = sc_mul(rho_c, zz);
        for (u32 k = 0; k < A + T; k++) { store_scalar_words(s + 8 * (4 + 2 * lg + k), v); v = sc_mul(v, z); }
    }
    // delta(y, z) = (z - zz) * sum_{i<nm} y^i - z^3 * (2^64 - 1) * sum_{j<m} z^j, sums as products of (1 + q^(2^k))
    sc sum_y = sc_one(), sum_z = sc_one();
    {
        sc q = y;
        for (u32 k = 0; k < lg; k++) { sum_y = sc_mul(sum_y, sc_add(sc_one(), q)); q = sc_mul(q, q); }
        q = z;
        for (u32 k = 0; k + 6 < lg; k++) { sum_z = sc_mul(sum_z, sc_add(sc_one(), q)); q = sc_mul(q, q); }
    }
    sc sum_2 = sc_from_u64(0xffffffffffffffffULL);
    sc delta = sc_sub(sc_mul(sc_sub(z, zz), sum_y), sc_mul(sc_mul(sc_mul(zz, z), sum_2), sum_z));
    sc b_scalar = sc_add(sc_mul(w, sc_sub(t_x, sc_mul(a, b))), sc_mul(c, sc_sub(delta, t_x)));
    sc bb_scalar = sc_neg(sc_add(e_bl, sc_mul(c, t_x_bl)));
    store_sc_limbs(gh + 18ull * i, sc_norm(sc_mul(rho, b_scalar)));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(sc_mul(rho, bb_scalar)));
    // context
    sc rho_z = sc_mul(rho, z);
    store_sc_limbs(ctx + 9 * RPC_RHO_Z, rho_z);
    store_sc_limbs(ctx + 9 * RPC_NEG_RHO_Z, sc_neg(rho_z));
    store_sc_limbs(ctx + 9 * RPC_COEF_G, sc_neg(sc_mul(sc_mul(rho, a), allinv)));
    store_sc_limbs(ctx + 9 * RPC_COEF_H2, sc_mul(rho, zz));
    store_sc_limbs(ctx + 9 * RPC_COEF_H3, sc_neg(sc_mul(sc_mul(rho, b), allinv)));
    {
        sc q = y_inv;                             // y^-(2^k)
        for (u32 k = 0; k < lg; k++) { store_sc_limbs(ctx + 9 * (RPC_Y2INV + k), q); q = sc_mul(q, q); }
        q = sc_add(y_inv, y_inv);                 // (2/y)^(2^k)
        for (u32 k = 0; k < 6; k++) { store_sc_limbs(ctx + 9 * (RPC_E + k), q); q = sc_mul(q, q); }
        // (z * y^-64)^(2^k): y^-64 = Y2INV[6]
        sc y64 = y_inv;
        for (u32 k = 0; k < 6; k++) y64 = sc_mul(y64, y64);
        q = sc_mul(z, y64);
        for (u32 k = 0; k + 6 < lg; k++) { store_sc_limbs(ctx + 9 * (RPC_F + k), q); q = sc_mul(q, q); }
    }
}

// thread per (tile of rows_per_thread transactions of the lg-group, i): partial column sums of the scalars
// of G_i and H_i, written as row `tile` of part[tiles][2 * 2^lg]
__global__ void __launch_bounds__(256, 2) k_rp_matrix(const u32 *__restrict__ group_txs, u32 n_rows, u32 rows_per_thread, u32 lg, const u32 *__restrict__ rp_ctx,
                                                      const u32 *__restrict__ rp_err, u32 *__restrict__ part) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 nm = 1u << lg;
    u32 tile = (u32)(g >> lg), idx = (u32)g & (nm - 1);
    u64 r0 = (u64)tile * rows_per_thread;
    if (r0 >= n_rows) return;
    u32 r1 = (u32)((r0 + rows_per_thread < n_rows) ? r0 + rows_per_thread : n_rows);
    sc sum_g = sc_zero(), sum_h = sc_zero();
    u32 pending = 0;
    for (u32 r = (u32)r0; r < r1; r++) {
        u32 tx = group_txs[r];
        if (rp_err[tx]) continue;
        const u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * tx;
        sc sprod = load_sc_limbs(ctx + 9 * RPC_COEF_G), tprod = load_sc_limbs(ctx + 9 * RPC_COEF_H3), eprod = load_sc_limbs(ctx + 9 * RPC_COEF_H2);
        for (u32 b = 0; b < lg; b++) {
            sc u2 = load_sc_limbs(ctx + 9 * (RPC_U2 + b));
            if ((idx >> b) & 1) {
                sprod = sc_mul(sprod, u2);
                tprod = sc_mul(tprod, load_sc_limbs(ctx + 9 * (RPC_Y2INV + b)));
                eprod = sc_mul(eprod, load_sc_limbs(ctx + 9 * ((b < 6) ? (RPC_E + b) : (RPC_F + b - 6))));
            } else {
                tprod = sc_mul(tprod, u2);
            }
        }
        sum_g = sc_add(sum_g, sc_add(load_sc_limbs(ctx + 9 * RPC_NEG_RHO_Z), sprod));
        sum_h = sc_add(sum_h, sc_norm(sc_add_lazy(sc_add(load_sc_limbs(ctx + 9 * RPC_RHO_Z), eprod), tprod)));
        if (++pending == 8) { sum_g = sc_reduce_mont(sum_g); sum_h = sc_reduce_mont(sum_h); pending = 0; }
    }
    store_sc_limbs(part + 9ull * ((u64)tile * 2 * nm + idx), sc_reduce_mont(sum_g));
    store_sc_limbs(part + 9ull * ((u64)tile * 2 * nm + nm + idx), sc_reduce_mont(sum_h));
}
// workgroup per column: acc[col] += sum over the rows of part[n_rows][2 * 2^lg]
__global__ void __launch_bounds__(256) k_rp_colsum(const u32 *__restrict__ mat, u32 n_rows, u32 lg, u32 nm_cap, u32 *__restrict__ acc /* 2*nm_cap scalars */) {
    __shared__ u32 lds[256 * 9];
    u32 col = blockIdx.x, t = threadIdx.x;
    u32 nm = 1u << lg;
    if (col >= 2 * nm) return;
    u32 dst = col < nm ? col : nm_cap + (col - nm);
    sc a = sc_zero(); u32 cnt = 0;
    for (u32 r = t; r < n_rows; r += 256) {
        a = sc_add(a, load_sc_limbs(mat + 9ull * ((u64)r * 2 * nm + col)));
        if (++cnt == 16) { a = sc_reduce_mont(a); cnt = 0; }
    }
    a = sc_reduce_mont(a);
    for (u32 d = 128; d >= 1; d >>= 1) {
        store_sc_limbs(lds + 9 * t, a);
        __syncthreads();
        if (t < d) a = sc_reduce_mont(sc_add(a, load_sc_limbs(lds + 9 * (t + d))));
        __syncthreads();
    }
    if (t == 0) store_sc_limbs(acc + 9ull * dst, sc_norm(sc_reduce_mont(sc_add(a, load_sc_limbs(acc + 9ull * dst)))));
}
// static terms: scalars of B, H (sums of the per-proof parts) and of G_i, H_i (column sums)
__global__ void __launch_bounds__(256) k_static_terms(const u32 *__restrict__ gh_sums /* n_parts x 2 */, u32 n_parts, const u32 *__restrict__ acc, u32 nm_cap,
                                                      u32 static_term_off, u32 *__restrict__ scalars, u32 *__restrict__ term_point) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 + 2 * nm_cap) return;
    sc v;
    if (g < 2) {
        v = sc_zero();
        for (u32 k = 0; k < n_parts; k++) v = sc_reduce_mont(sc_add(v, load_sc_limbs(gh_sums + 9ull * (2 * k + g))));
    } else v = load_sc_limbs(acc + 9ull * (g - 2));
    store_scalar_words(scalars + 8ull * (static_term_off + g), v);
    term_point[static_term_off + g] = g;
}


void launch_sigma_cm(hipStream_t st, const u8 *bytes, const dev_cm *cms, const dev_tx *txs, u32 n, const u32 *chal, const u32 *seed, u32 *scalars, u32 *term_point, u32 *gh) {
    if (n) hipLaunchKernelGGL(k_sigma_cm, dim3(grid_for(n, 256)), dim3(256), 0, st, bytes, cms, txs, n, chal, seed, scalars, term_point, gh);
}
void launch_sigma_xf(hipStream_t st, const u8 *bytes, const dev_xfer *xfers, const dev_tx *txs, u32 n, const u32 *chal, const u32 *seed, u32 *scalars, u32 *term_point,
                     u32 *gh) {
    if (n) hipLaunchKernelGGL(k_sigma_xf, dim3(grid_for(n, 256)), dim3(256), 0, st, bytes, xfers, txs, n, chal, seed, scalars, term_point, gh);
}
void launch_sc_sum_partial(hipStream_t st, u32 blocks, const u32 *in, u32 n, u32 classes, u32 *out) {
    hipLaunchKernelGGL(k_sc_sum_partial, dim3(blocks), dim3(256), 0, st, in, n, classes, out);
}
void launch_rp_prep(hipStream_t st, const u8 *bytes, const dev_tx *txs, const dev_cm *cms, const dev_xfer *xfers, u32 n_txs, const u32 *chal, const u32 *seed, u32 *scalars,
                    u32 *term_point, u32 *rp_ctx, u32 *gh, u32 *rp_err) {
    if (n_txs) hipLaunchKernelGGL(k_rp_prep, dim3(grid_for(n_txs, 64)), dim3(64), 0, st, bytes, txs, cms, xfers, n_txs, chal, seed, scalars, term_point, rp_ctx, gh, rp_err);
}
void launch_rp_matrix(hipStream_t st, const u32 *group_txs, u32 n_rows, u32 rows_per_thread, u32 lg, const u32 *rp_ctx, const u32 *rp_err, u32 *part) {
    if (!n_rows || !rows_per_thread) return;
    u64 tiles = ((u64)n_rows + rows_per_thread - 1) / rows_per_thread;
    hipLaunchKernelGGL(k_rp_matrix, dim3(grid_for(tiles << lg, 256)), dim3(256), 0, st, group_txs, n_rows, rows_per_thread, lg, rp_ctx, rp_err, part);
}
void launch_rp_colsum(hipStream_t st, const u32 *mat, u32 n_rows, u32 lg, u32 nm_cap, u32 *acc) {
    hipLaunchKernelGGL(k_rp_colsum, dim3(2u << lg), dim3(256), 0, st, mat, n_rows, lg, nm_cap, acc);
}
void launch_static_terms(hipStream_t st, const u32 *gh_sums, u32 n_parts, const u32 *acc, u32 nm_cap, u32 static_term_off, u32 *scalars, u32 *term_point) {
    hipLaunchKernelGGL(k_static_terms, dim3(grid_for(2 + 2 * nm_cap, 256)), dim3(256), 0, st, gh_sums, n_parts, acc, nm_cap, static_term_off, scalars, term_point);
}

}  // namespace xhe
