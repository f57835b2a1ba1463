// Proof scalars: sigma-proof terms (src/proofs.rs:188-208,336-358), range-proof verification scalars and
// their sums over the batch, scalars of the static generators.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

// ---- sigma proofs ----------------------------------------------------------------------------------
// A proof is three equations over points.  The reference folds them with powers of a transcript challenge w and gives the
// proof one random batch factor (src/proofs.rs:188-208, 336-358: eq0 + w eq1 + w^2 eq2, times the factor).  Here every
// equation gets a random factor of its own, of at least 128 bits (b0, b1, b2, drawn from the call's seed, independent of the proof): the
// batch sum vanishes for all factors iff every equation holds, which is what the folded form establishes too (up to the same
// negligible probability), and the commitments Y_0, Y_1, Y_2 - three of a proof's seven or eight terms - keep scalars of that
// width, about half the digits of a full one in the bucket sort and the accumulation (the width: sigma_weight_bits).  The challenge w is still derived (the transcript
// state goes on through it to the next proof); it is just not a factor.
// gh[2p], gh[2p+1]: this proof's contributions to the scalars of G and H (Montgomery limbs)
__device__ __forceinline__ void sigma_cm_terms(u32 i, const u8 *__restrict__ bytes, const dev_cm *__restrict__ cms, const dev_tx *__restrict__ txs,
                                               const u32 *__restrict__ chal, const u32 *__restrict__ seed, u32 *__restrict__ scalars,
                                               u32 *__restrict__ term_point, u32 *__restrict__ gh, u32 weight_bits) {
    const dev_cm c = cms[i];
    const dev_tx d = txs[c.tx];
    const u8 *p = bytes + c.off;
    sc ch_c = load_scalar_words(chal + 8ull * (d.chal_off + 2 * c.idx));
    sc z_s = load_scalar_bytes(p + 160), z_x = load_scalar_bytes(p + 192), z_r = load_scalar_bytes(p + 224);
    sc b0, b1, b2;
    prf_scalar_x3(b0, b1, b2, seed, 1, i, weight_bits);
    sc neg_b0 = sc_neg(b0), neg_b1 = sc_neg(b1), neg_b2 = sc_neg(b2);
    u32 *s = scalars + 8ull * c.term_off;
    u32 *tp = term_point + c.term_off;
    // eq0: z_s P_source - c H - Y_0        eq1: z_x G + z_s D_source - c C_source - Y_1        eq2: z_x G + z_r H - c C_destination - Y_2
    store_scalar_words(s, sc_mul(z_s, b0));                   tp[0] = d.src_pt;          // b0 z_s * P_source
    store_scalar_words(s + 8, neg_b0);                        tp[1] = c.pt + 1;          // -b0 * Y_0
    store_scalar_words(s + 16, sc_mul(b1, z_s));              tp[2] = c.src_ct_pt + 1;   // b1 z_s * D_source
    store_scalar_words(s + 24, sc_mul(neg_b1, ch_c));         tp[3] = c.src_ct_pt;       // -b1 c * C_source
    store_scalar_words(s + 32, neg_b1);                       tp[4] = c.pt + 2;          // -b1 * Y_1
    store_scalar_words(s + 40, sc_mul(neg_b2, ch_c));         tp[5] = c.pt;              // -b2 c * C_destination
    store_scalar_words(s + 48, neg_b2);                       tp[6] = c.pt + 3;          // -b2 * Y_2
    // G: (b1 + b2) z_x ; H: -b0 c + b2 z_r
    sc g = sc_mul(sc_add(b1, b2), z_x);
    sc h = sc_add(sc_mul(neg_b0, ch_c), sc_mul(b2, z_r));
    store_sc_limbs(gh + 18ull * i, sc_norm(g));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(h));
}
__device__ __forceinline__ void sigma_xf_terms(u32 i, const u8 *__restrict__ bytes, const dev_xfer *__restrict__ xfers, const dev_tx *__restrict__ txs,
                                               const u32 *__restrict__ chal, const u32 *__restrict__ seed, u32 *__restrict__ scalars,
                                               u32 *__restrict__ term_point, u32 *__restrict__ gh, u32 weight_bits) {
    const dev_xfer x = xfers[i];
    const dev_tx d = txs[x.tx];
    const u8 *pr = bytes + x.proof_off;
    u32 cbase = d.chal_off + 2 * d.n_commitments + 2 * x.idx;
    sc ch_c = load_scalar_words(chal + 8ull * cbase);
    sc z_r = load_scalar_bytes(pr + 96), z_x = load_scalar_bytes(pr + 128);
    sc b0, b1, b2;
    prf_scalar_x3(b0, b1, b2, seed, 2, i, weight_bits);
    sc neg_b0 = sc_neg(b0), neg_b1 = sc_neg(b1), neg_b2 = sc_neg(b2);
    u32 *s = scalars + 8ull * x.term_off;
    u32 *tp = term_point + x.term_off;
    // eq0: z_r H + z_x G - c C - Y_0        eq1: z_r P_dest - c D_dest - Y_1        eq2: z_r P_source - c D_source - Y_2
    store_scalar_words(s, sc_mul(neg_b0, ch_c));              tp[0] = x.pt;              // -b0 c * C
    store_scalar_words(s + 8, neg_b0);                        tp[1] = x.pt + 3;          // -b0 * Y_0
    store_scalar_words(s + 16, sc_mul(b1, z_r));              tp[2] = x.dest_pt;         // b1 z_r * P_dest
    store_scalar_words(s + 24, sc_mul(neg_b1, ch_c));         tp[3] = x.pt + 2;          // -b1 c * D_dest
    store_scalar_words(s + 32, neg_b1);                       tp[4] = x.pt + 4;          // -b1 * Y_1
    store_scalar_words(s + 40, sc_mul(b2, z_r));              tp[5] = d.src_pt;          // b2 z_r * P_source
    store_scalar_words(s + 48, sc_mul(neg_b2, ch_c));         tp[6] = x.pt + 1;          // -b2 c * D_source
    store_scalar_words(s + 56, neg_b2);                       tp[7] = x.pt + 5;          // -b2 * Y_2
    store_sc_limbs(gh + 18ull * i, sc_norm(sc_mul(z_x, b0)));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(sc_mul(z_r, b0)));
}

// sum of n scalars (Montgomery limbs, stride words apart) per residue class j = 0..classes-1:
// out[b*classes + j] = sum over i in block b's share of in[(i*classes + j)]
__device__ __forceinline__ sc sc_reduce_mont(const sc &a) { return sc_mul(a, sc_one()); }
__global__ void __launch_bounds__(256) k_sc_sum_partial(const u32 *__restrict__ in, u32 n, u32 classes, u32 *__restrict__ out) {
    wave_priority_mid();
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
// The verification equation of one aggregated range proof (bulletproofs RangeProof::verify_multiple, called at
// /root/reference/src/tx/verify.rs:504-513) holds iff it holds after multiplying every scalar of that proof by
// one non-zero factor.  Its inverses (u_j^-2 on R_j, s_i = prod u_j^(+-1), y^-i) all disappear under the
// factor kappa = (prod u_j)^2 * y^(nm-1), so the proof's random weight is taken as rho * kappa (as uniform as
// rho, which is drawn independently of the proof) and no scalar is ever inverted:
//   U = prod u_j,  U2[b] = u_(lg-1-b)^2,  Y2[b] = y^(2^b),  Yt = y^(nm-1) = prod Y2[b],  rho' = rho U^2 Yt
//   A: rho'   S: rho' x   T_1: rho' c x   T_2: rho' c x^2   L_j: rho' u_j^2   R_j: rho Yt prod_(k != j) u_k^2
//   V_k: rho' c z^2 z^k   B: rho' (w (t_x - a b) + c (delta - t_x))   B_blinding: -rho' (e_bl + c t_x_bl)
//   G_i: -rho' z - rho a U Yt    prod_(b in i) U2[b]
//   H_i:  rho' z - rho b U       prod_(b not in i) Y2[b] U2[b]
//               + rho z^2 U^2    prod_(b in i) E[b]  prod_(b not in i) Y2[b],   E[b] = 2^(2^b) (b < 6), z^(2^(b-6)) (b >= 6)
// per-transaction context for the matrix kernel, Montgomery limbs (9 words each)

// Two threads per proof (in different workgroups, so control flow stays uniform): `side` 0 computes the scalars of the
// L_j and R_j and the per-bit tables of the context, side 1 everything else.  Both need y^(nm-1) and prod u_j^2 and
// derive them themselves (two short chains) rather than wait for each other.
__device__ __forceinline__ void rp_prep_terms(u32 i, u32 side, const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const dev_cm *__restrict__ cms,
                                              const dev_xfer *__restrict__ xfers, const u32 *__restrict__ chal, const u32 *__restrict__ seed,
                                              u32 *__restrict__ scalars, u32 *__restrict__ term_point, u32 *__restrict__ rp_ctx, u32 *__restrict__ gh,
                                              u32 *__restrict__ rp_err) {
    const dev_tx d = txs[i];
    const u32 lg = d.lg, A = d.n_commitments, T = d.n_transfers;
    const u32 n_dyn = 4 + 2 * lg + A + T;
    u32 *s = scalars + 8ull * d.rp_term_off;
    const bool valid = lg >= 6 && lg <= 15 && (64ull * d.m == (1ull << lg)) && d.m <= 512;
    if (side == 0) {
        u32 *tp = term_point + d.rp_term_off;
        // term -> point map is valid whatever happens below
        for (u32 k = 0; k < 4 + 2 * lg; k++) tp[k] = d.rp_pt + k;
        for (u32 k = 0; k < A; k++) tp[4 + 2 * lg + k] = cms[d.cm_first + k].pt;
        for (u32 k = 0; k < T; k++) tp[4 + 2 * lg + A + k] = xfers[d.xfer_first + k].pt;
    }
    if (!valid) {
        if (side == 0) { for (u32 k = 0; k < 8 * n_dyn; k++) s[k] = 0; }
        else { rp_err[i] = 1; store_sc_limbs(gh + 18ull * i, sc_zero()); store_sc_limbs(gh + 18ull * i + 9, sc_zero()); }
        return;
    }
    const u8 *rp = bytes + d.rec + d.rp_off;
    const u32 *ch = chal + 8ull * (d.chal_off + 2 * A + 2 * T);
    const sc y = load_scalar_words(ch);
    sc rho, c;
    prf_scalar_pair(rho, c, seed, 3, i);
    u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * i;

    if (side == 0) {
        // powers of y, Yt = y^(nm-1)
        sc yt = sc_one();
        {
            sc q = y;
            for (u32 k = 0; k < lg; k++) { store_sc_limbs(ctx + 9 * (RPC_Y2 + k), q); yt = sc_mul(yt, q); q = sc_mul(q, q); }
        }
        // squares of the u_j: prefix products left to right, then suffix products on the way back
        sc u2[16], pre[17];
        pre[0] = sc_one();
        for (u32 j = 0; j < lg; j++) {
            sc u = load_scalar_words(ch + 32 + 8 * j);
            u2[j] = sc_mul(u, u);
            pre[j + 1] = sc_mul(pre[j], u2[j]);
        }
        const sc rho_yt = sc_mul(rho, yt), rho_k = sc_mul(rho_yt, pre[lg]);       // rho' = rho * kappa
        sc suffix = rho_yt;                           // rho Yt prod_(k > j) u_k^2
        for (int j = (int)lg - 1; j >= 0; j--) {
            store_scalar_words(s + 8 * (4 + 2 * j), sc_mul(rho_k, u2[j]));               // L_j
            store_scalar_words(s + 8 * (4 + 2 * j + 1), sc_mul(suffix, pre[j]));         // R_j
            suffix = sc_mul(suffix, u2[j]);
            const u32 bit = lg - 1 - (u32)j;          // bit (lg-1-j) of the index selects u_j
            store_sc_limbs(ctx + 9 * (RPC_U2 + bit), u2[j]);
            store_sc_limbs(ctx + 9 * (RPC_YU + bit), sc_mul(u2[j], load_sc_limbs(ctx + 9 * (RPC_Y2 + bit))));
        }
        return;
    }

    const sc z = load_scalar_words(ch + 8), x = load_scalar_words(ch + 16), w = load_scalar_words(ch + 24);
    const sc t_x = load_scalar_bytes(rp + 128), t_x_bl = load_scalar_bytes(rp + 160), e_bl = load_scalar_bytes(rp + 192);
    const sc a = load_scalar_bytes(rp + 224 + 64 * lg), b = load_scalar_bytes(rp + 224 + 64 * lg + 32);
    // the sum of the y^i as the product of (1 + y^(2^k)), Yt = y^(nm-1)
    sc sum_y = sc_one(), yt = sc_one();
    {
        sc q = y;
        for (u32 k = 0; k < lg; k++) { sum_y = sc_mul(sum_y, sc_add(sc_one(), q)); yt = sc_mul(yt, q); q = sc_mul(q, q); }
    }
    sc U = sc_one();
    for (u32 j = 0; j < lg; j++) U = sc_mul(U, load_scalar_words(ch + 32 + 8 * j));
    const sc UU = sc_mul(U, U);
    const sc rho_k = sc_mul(sc_mul(rho, yt), UU);     // rho' = rho * kappa
    const sc rho_kc = sc_mul(rho_k, c);
    sc zz = sc_mul(z, z);
    const sc rho_kcx = sc_mul(rho_kc, x);
    store_scalar_words(s, rho_k);                                  // A
    store_scalar_words(s + 8, sc_mul(rho_k, x));                   // S
    store_scalar_words(s + 16, rho_kcx);                           // T_1
    store_scalar_words(s + 24, sc_mul(rho_kcx, x));                // T_2
    {
        sc v = sc_mul(rho_kc, zz);
        for (u32 k = 0; k < A + T; k++) { store_scalar_words(s + 8 * (4 + 2 * lg + k), v); v = sc_mul(v, z); }
    }
    // delta(y, z) = (z - zz) * sum_{i<nm} y^i - z^3 * (2^64 - 1) * sum_{j<m} z^j, the z sum as a product of (1 + z^(2^k)) as well;
    // E[b]: 2^(2^b) for the bits inside a 64-bit value, z^(2^(b-6)) for the bits of the value's index
    sc sum_z = sc_one();
    {
        sc q = sc_from_u64(2);
        for (u32 k = 0; k < 6; k++) { store_sc_limbs(ctx + 9 * (RPC_E + k), q); q = sc_mul(q, q); }
        q = z;
        for (u32 k = 0; k + 6 < lg; k++) { store_sc_limbs(ctx + 9 * (RPC_E + 6 + k), q); sum_z = sc_mul(sum_z, sc_add(sc_one(), q)); q = sc_mul(q, q); }
    }
    sc sum_2 = sc_from_u64(0xffffffffffffffffULL);
    sc delta = sc_sub(sc_mul(sc_sub(z, zz), sum_y), sc_mul(sc_mul(sc_mul(zz, z), sum_2), sum_z));
    sc b_scalar = sc_add(sc_mul(w, sc_sub(t_x, sc_mul(a, b))), sc_mul(c, sc_sub(delta, t_x)));
    sc bb_scalar = sc_neg(sc_add(e_bl, sc_mul(c, t_x_bl)));
    store_sc_limbs(gh + 18ull * i, sc_norm(sc_mul(rho_k, b_scalar)));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(sc_mul(rho_k, bb_scalar)));
    // context
    sc rho_z = sc_mul(rho_k, z), rho_u = sc_mul(rho, U);
    store_sc_limbs(ctx + 9 * RPC_RHO_Z, rho_z);
    store_sc_limbs(ctx + 9 * RPC_NEG_RHO_Z, sc_neg(rho_z));
    store_sc_limbs(ctx + 9 * RPC_COEF_G, sc_neg(sc_mul(sc_mul(rho_u, yt), a)));
    store_sc_limbs(ctx + 9 * RPC_COEF_H2, sc_mul(sc_mul(rho, UU), zz));
    store_sc_limbs(ctx + 9 * RPC_COEF_H3, sc_neg(sc_mul(rho_u, b)));
}

// One launch for the scalars of all three proof kinds (they only need the transcript's challenges and are independent
// of each other): blocks [0, b_cm) take commitment-equality proofs, [b_cm, b_cm + b_xf) ciphertext-validity proofs, the
// rest one transaction's range proof per thread.  gh[]: per proof / per transaction the contributions to B and H.
__global__ void __launch_bounds__(64, 2) k_proof_scalars(proof_scalar_args a) {
    wave_priority_high();
    const u32 b_cm = (a.n_cm + 63) / 64, b_xf = (a.n_xf + 63) / 64;
    u32 blk = blockIdx.x;
    if (blk < b_cm) {
        u32 i = blk * 64 + threadIdx.x;
        if (i < a.n_cm) sigma_cm_terms(i, a.bytes, a.cms, a.txs, a.chal, a.seed, a.scalars, a.term_point, a.gh, a.weight_bits);
    } else if (blk < b_cm + b_xf) {
        u32 i = (blk - b_cm) * 64 + threadIdx.x;
        if (i < a.n_xf) sigma_xf_terms(i, a.bytes, a.xfers, a.txs, a.chal, a.seed, a.scalars, a.term_point, a.gh + 18ull * a.n_cm, a.weight_bits);
    } else {
        u32 r = blk - b_cm - b_xf;                     // even workgroups: side 0, odd: side 1 of the same 64 transactions
        u32 i = (r >> 1) * 64 + threadIdx.x;
        if (i < a.n_txs) rp_prep_terms(i, r & 1, a.bytes, a.txs, a.cms, a.xfers, a.chal, a.seed, a.scalars, a.term_point, a.rp_ctx, a.gh + 18ull * (a.n_cm + a.n_xf), a.rp_err);
    }
}

// Workgroup per tile of rows_per_thread transactions of the lg-group: partial column sums of the scalars of
// G_i and H_i, written as row `tile` of part[tiles][2 * 2^lg].  Per transaction and index i three products over
// the bits of i are needed (s_i, its mirror times y^-i, z^j 2^k y^-i).  A product over the bits of i is the
// product of one factor for the low bits and one for the high bits, so the workgroup first fills, per
// transaction, two small tables per product in LDS (2^l + 2^h entries, l + h = lg) and every element then costs
// three multiplications instead of 2.5 lg.  (Products: see the table above rp_prep_terms.)
static constexpr u32 RPM_THREADS = 256;
// Groups with fewer than 256 generators per side run 256 >> lg transactions side by side ("slots"), each slot
// writing its own row of part[]; larger groups give every thread several indices.
__global__ void __launch_bounds__(RPM_THREADS) k_rp_matrix(const u32 *__restrict__ group_txs, u32 n_rows, u32 rows_per_thread, u32 lg, const u32 *__restrict__ rp_ctx,
                                                            const u32 *__restrict__ rp_err, u32 *__restrict__ part) {
    wave_priority_mid();
    extern __shared__ u32 tab[];                       // [slot][product 0..2][2^l low entries | 2^h high entries][9]
    const u32 nm = 1u << lg, l = (lg + 1) / 2, h = lg - l, n_lo = 1u << l, n_hi = 1u << h, per = n_lo + n_hi;
    const u32 slots = rp_matrix_slots(lg);
    const u32 tile = blockIdx.x, t = threadIdx.x;
    const u64 r0 = (u64)tile * rows_per_thread;
    const u32 r1 = (u32)((r0 + rows_per_thread < n_rows) ? r0 + rows_per_thread : n_rows);
    const u32 my_slot = nm < RPM_THREADS ? t >> lg : 0, idx0 = nm < RPM_THREADS ? (t & (nm - 1)) : t;
    const u32 mine = nm <= RPM_THREADS ? 1 : nm / RPM_THREADS;       // indices per thread: idx0 + k * 256
    u32 *out = part + 9ull * (((u64)tile * slots + my_slot) * 2 * nm);
    const bool in_regs = mine <= 4;
    sc sum_g[4], sum_h[4];
#pragma unroll
    for (u32 k = 0; k < 4; k++) { sum_g[k] = sc_zero(); sum_h[k] = sc_zero(); }
    if (!in_regs)                                     // many indices per thread: the sums live in part[] itself
        for (u32 k = 0; k < mine; k++) { store_sc_limbs(out + 9ull * (idx0 + k * RPM_THREADS), sc_zero()); store_sc_limbs(out + 9ull * (nm + idx0 + k * RPM_THREADS), sc_zero()); }
    u32 pending = 0;
    for (u32 r = (u32)r0; r < r1; r += slots) {
        __syncthreads();                               // the previous pass's tables are no longer read
        for (u32 e = t; e < slots * 3 * per; e += RPM_THREADS) {
            const u32 slot = e / (3 * per), e3 = e % (3 * per), p = e3 / per, q = e3 % per;
            if (r + slot >= r1) continue;
            const u32 tx = group_txs[r + slot];
            if (rp_err[tx]) continue;
            const u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * tx;
            const bool high = q >= n_lo;
            const u32 bits = high ? q - n_lo : q, base = high ? l : 0, nb = high ? h : l;
            sc acc = high ? load_sc_limbs(ctx + 9 * (p == 0 ? RPC_COEF_G : p == 1 ? RPC_COEF_H3 : RPC_COEF_H2)) : sc_one();
            for (u32 k = 0; k < nb; k++) {
                const u32 b = base + k;
                const bool set = (bits >> k) & 1;
                if (p == 0) { if (set) acc = sc_mul(acc, load_sc_limbs(ctx + 9 * (RPC_U2 + b))); }
                else if (p == 1) { if (!set) acc = sc_mul(acc, load_sc_limbs(ctx + 9 * (RPC_YU + b))); }
                else acc = sc_mul(acc, load_sc_limbs(ctx + 9 * (set ? RPC_E + b : RPC_Y2 + b)));
            }
            store_sc_limbs(tab + 9 * e, acc);
        }
        __syncthreads();
        const bool live = r + my_slot < r1 && !rp_err[group_txs[r + my_slot < r1 ? r + my_slot : r]];
        if (live) {
            const u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * group_txs[r + my_slot];
            const u32 *tb = tab + 9 * (my_slot * 3 * per);
            const sc neg_rho_z = load_sc_limbs(ctx + 9 * RPC_NEG_RHO_Z), rho_z = load_sc_limbs(ctx + 9 * RPC_RHO_Z);
            auto element = [&](u32 idx, sc &g, sc &hh) {
                const u32 lo = idx & (n_lo - 1), hi = idx >> l;
                sc sprod = sc_mul(load_sc_limbs(tb + 9 * lo), load_sc_limbs(tb + 9 * (n_lo + hi)));
                sc tprod = sc_mul(load_sc_limbs(tb + 9 * (per + lo)), load_sc_limbs(tb + 9 * (per + n_lo + hi)));
                sc eprod = sc_mul(load_sc_limbs(tb + 9 * (2 * per + lo)), load_sc_limbs(tb + 9 * (2 * per + n_lo + hi)));
                g = sc_add(neg_rho_z, sprod); hh = sc_norm(sc_add_lazy(sc_add(rho_z, eprod), tprod));
            };
            if (in_regs) {
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    if (k < mine) { sc g, hh; element(idx0 + k * RPM_THREADS, g, hh); sum_g[k] = sc_add(sum_g[k], g); sum_h[k] = sc_add(sum_h[k], hh); }
                }
            } else {
                for (u32 k = 0; k < mine; k++) {
                    const u32 idx = idx0 + k * RPM_THREADS;
                    sc g, hh; element(idx, g, hh);
                    u32 *pg = out + 9ull * idx, *ph = out + 9ull * (nm + idx);
                    store_sc_limbs(pg, sc_reduce_mont(sc_add(load_sc_limbs(pg), g)));
                    store_sc_limbs(ph, sc_reduce_mont(sc_add(load_sc_limbs(ph), hh)));
                }
            }
        }
        if (in_regs && ++pending == 8) {
#pragma unroll
            for (u32 k = 0; k < 4; k++) { sum_g[k] = sc_reduce_mont(sum_g[k]); sum_h[k] = sc_reduce_mont(sum_h[k]); }
            pending = 0;
        }
    }
    if (in_regs) {
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            if (k < mine) {
                store_sc_limbs(out + 9ull * (idx0 + k * RPM_THREADS), sc_reduce_mont(sum_g[k]));
                store_sc_limbs(out + 9ull * (nm + idx0 + k * RPM_THREADS), sc_reduce_mont(sum_h[k]));
            }
        }
    }
}
// workgroup per column: acc[col] += sum over the rows of part[n_rows][2 * 2^lg]
__global__ void __launch_bounds__(256) k_rp_colsum(const u32 *__restrict__ mat, u32 n_rows, u32 lg, u32 nm_cap, u32 *__restrict__ acc /* 2*nm_cap scalars */) {
    wave_priority_mid();
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
    wave_priority_mid();
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 + 2 * nm_cap) return;
    sc v;
    if (g < 2) {
        v = sc_zero();
        for (u32 k = 0; k < n_parts; k++) {
            v = sc_add(v, load_sc_limbs(gh_sums + 9ull * (2 * k + g)));
            if ((k & 7) == 7) v = sc_reduce_mont(v);
        }
        v = sc_reduce_mont(v);
    } else v = load_sc_limbs(acc + 9ull * (g - 2));
    store_scalar_words(scalars + 8ull * (static_term_off + g), v);
    term_point[static_term_off + g] = g;
}


// ---- terms that share a point ----------------------------------------------------------------------------
// thread per pair (kept, absorbed): the kept term takes the sum of the two scalars, the absorbed one becomes zero (a zero
// scalar has no digits and never reaches the bucket sort)
__global__ void __launch_bounds__(256) k_merge_pairs(const u32 *__restrict__ pairs, u32 n, u32 *__restrict__ scalars) {
    wave_priority_high();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 keep = pairs[2 * i], drop = pairs[2 * i + 1];
    const sc sum = sc_add(load_scalar_words(scalars + 8ull * keep), load_scalar_words(scalars + 8ull * drop));
    store_scalar_words(scalars + 8ull * keep, sum);
#pragma unroll
    for (int k = 0; k < 8; k++) scalars[8ull * drop + k] = 0;
}
// wave per run of terms on one point: the first term takes the sum of the run's scalars, the others become zero
__global__ void __launch_bounds__(64) k_merge_runs(const u32 *__restrict__ run_off, u32 n_runs, const u32 *__restrict__ terms, u32 *__restrict__ scalars) {
    wave_priority_high();
    const u32 g = blockIdx.x, lane = threadIdx.x;
    if (g >= n_runs) return;
    const u32 lo = run_off[g], hi = run_off[g + 1];
    sc acc = sc_zero(); u32 cnt = 0;
    for (u32 k = lo + lane; k < hi; k += 64) {
        const u32 t = terms[k];
        acc = sc_add(acc, load_scalar_words(scalars + 8ull * t));
        if (++cnt == 8) { acc = sc_reduce_mont(acc); cnt = 0; }
        if (k != lo) {
#pragma unroll
            for (int j = 0; j < 8; j++) scalars[8ull * t + j] = 0;
        }
    }
    acc = sc_reduce_mont(acc);
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) {
        sc o;
#pragma unroll
        for (int j = 0; j < 9; j++) o.v[j] = __shfl_down(acc.v[j], d, 64);
        acc = sc_reduce_mont(sc_add(acc, o));
    }
    if (lane == 0 && hi > lo) store_scalar_words(scalars + 8ull * terms[lo], acc);
}
void launch_merge_terms(hipStream_t st, const u32 *pairs, u32 n_pairs, const u32 *run_off, u32 n_runs, const u32 *run_terms, u32 *scalars) {
    if (n_pairs) hipLaunchKernelGGL(k_merge_pairs, dim3(grid_for(n_pairs, 256)), dim3(256), 0, st, pairs, n_pairs, scalars);
    if (n_runs) hipLaunchKernelGGL(k_merge_runs, dim3(n_runs), dim3(64), 0, st, run_off, n_runs, run_terms, scalars);
}

void launch_proof_scalars(hipStream_t st, const proof_scalar_args &a) {
    u32 blocks = (a.n_cm + 63) / 64 + (a.n_xf + 63) / 64 + 2 * ((a.n_txs + 63) / 64);          // two workgroups per 64 range proofs
    if (blocks) hipLaunchKernelGGL(k_proof_scalars, dim3(blocks), dim3(64), 0, st, a);
}
void launch_sc_sum_partial(hipStream_t st, u32 blocks, const u32 *in, u32 n, u32 classes, u32 *out) {
    hipLaunchKernelGGL(k_sc_sum_partial, dim3(blocks), dim3(256), 0, st, in, n, classes, out);
}
void launch_rp_matrix(hipStream_t st, const u32 *group_txs, u32 n_rows, u32 rows_per_thread, u32 lg, const u32 *rp_ctx, const u32 *rp_err, u32 *part) {
    if (!n_rows || !rows_per_thread) return;
    u64 tiles = ((u64)n_rows + rows_per_thread - 1) / rows_per_thread;
    u32 l = (lg + 1) / 2, h = lg - l;
    u32 lds = rp_matrix_slots(lg) * 3u * ((1u << l) + (1u << h)) * 36u;
    hipLaunchKernelGGL(k_rp_matrix, dim3((u32)tiles), dim3(RPM_THREADS), lds, st, group_txs, n_rows, rows_per_thread, lg, rp_ctx, rp_err, part);
}
void launch_rp_colsum(hipStream_t st, const u32 *mat, u32 n_rows, u32 lg, u32 nm_cap, u32 *acc) {
    hipLaunchKernelGGL(k_rp_colsum, dim3(2u << lg), dim3(256), 0, st, mat, n_rows, lg, nm_cap, acc);
}
void launch_static_terms(hipStream_t st, const u32 *gh_sums, u32 n_parts, const u32 *acc, u32 nm_cap, u32 static_term_off, u32 *scalars, u32 *term_point) {
    hipLaunchKernelGGL(k_static_terms, dim3(grid_for(2 + 2 * nm_cap, 256)), dim3(256), 0, st, gh_sums, n_parts, acc, nm_cap, static_term_off, scalars, term_point);
}

}  // namespace xhe
