// Batched generation of aggregated 64-bit Bulletproofs range proofs on gfx950: the GPU side of the synthetic-batch
// generator.  It restates RangeProof::prove_multiple as the reference calls it at /root/reference/src/tx/builder.rs:525-533
// (bulletproofs 5.2.0 of the xelis fork, a dependency outside the tree: party/dealer folded into one pass, inner-product
// argument with H_factors = y^-i) and produces, for the same random draws, the bytes the CPU prover of xhe_builder.cpp
// produces.  Not part of the verification path: it creates inputs for it.
//
// Layout of the work.  Every point multiplication is by a FIXED generator (B, B_blinding, G_i, H_i), so every term is
// 32 look-ups in the context's 32 x 256 fixed-base tables and 32 mixed additions:
//   * S = s_bl*B_bl + <s_L, G> + <s_R, H> and A = a_bl*B_bl + sum(bit ? G_i : -H_i): thread per generator index;
//   * the inner-product argument never folds the generators.  After rounds 1..k-1 with challenges u_r the folded
//     generator G'_q is sum_j cG(p) G_p over the original positions p = q (mod current length), with
//     cG(p) = prod_r (bit_{lg-r}(p) ? u_r : u_r^-1) and cH(p) = y^-p prod_r (bit_{lg-r}(p) ? u_r^-1 : u_r); so round k's
//     L and R are multiscalar sums over the ORIGINAL 2 nm generators, each generator feeding exactly one of L, R with
//     the scalar (a or b entry) * c(p).  2 nm fixed-base terms per round instead of nm variable-base ones plus the folding.
// A wavefront takes 64 x PT generators that feed the same output and sums its lanes' results with shuffles.
// The transcript (Merlin) and the scalar algebra between the point stages run one thread per proof.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

// SHAKE256(seed || LE64(counter)) -> 64 bytes -> scalar: the builder's random generator (xhe_host.h, struct drbg)
__device__ __forceinline__ sc drbg_scalar(const u32 *__restrict__ seed8, u64 counter) {
    u64 st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) st[i] = (u64)seed8[2 * i] | ((u64)seed8[2 * i + 1] << 32);
    st[4] = counter; st[5] = 0x1f; st[16] = 0x8000000000000000ULL;
    keccak_f1600_regs(st);
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { w[2 * i] = (u32)st[i]; w[2 * i + 1] = (u32)(st[i] >> 32); }
    return sc_from_wide_words(w);
}

__device__ __forceinline__ sc sc_reduce_once(const sc &x) { return sc_mul(x, sc_one()); }      // any value below 2^258 -> below 1.07 l
__device__ __forceinline__ sc sc_pow_small(const sc &base, u32 e) {
    sc r = sc_one(), b = base;
    while (e) {
        if (e & 1) r = sc_mul(r, b);
        e >>= 1;
        if (e) b = sc_sq(b);
    }
    return r;
}
__device__ __forceinline__ sc sc_shfl_down(const sc &x, u32 delta) {
    sc r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = __shfl_down(x.v[i], delta, 64);
    return r;
}
// sum over the wavefront, every partial kept below 1.07 l; lane 0 holds the result
__device__ __forceinline__ sc wave_scalar_sum(sc x) {
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) x = sc_reduce_once(sc_add(x, sc_shfl_down(x, d)));
    return x;
}
__device__ __forceinline__ ge wave_point_sum(ge acc) {
    const u32 lane = threadIdx.x & 63;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) {
        ge o = ge_shfl_down(acc, d);
        ge sum = ge_add(acc, o);
        acc = ge_select(acc, sum, lane < d);
    }
    return acc;
}

// acc + k * P_base from the 32 x 256 table of P_base (entry 0 of a window is the identity: no branch on zero bytes).
// k as eight canonical words.  The record of window w + 1 is requested before the addition of window w.
__device__ __forceinline__ ge fixed_base_accumulate(ge acc, const u32 *__restrict__ gen_tab, u32 base, const u32 k[8]) {
    const u32 *tab = gen_tab + 32ull * ((u64)base * 32 * 256);
    u32 w0 = k[0], w1 = k[1], w2 = k[2], w3 = k[3], w4 = k[4], w5 = k[5], w6 = k[6], w7 = k[7];
    auto fetch = [&](u32 win, u32 byte, u32 raw[24]) {
        const uint4 *src = (const uint4 *)(tab + 32ull * (win * 256 + byte));
#pragma unroll
        for (int j = 0; j < 6; j++) { uint4 v = src[j]; raw[4 * j] = v.x; raw[4 * j + 1] = v.y; raw[4 * j + 2] = v.z; raw[4 * j + 3] = v.w; }
    };
    u32 cur[24];
    fetch(0, w0 & 255u, cur);
#pragma unroll 1
    for (u32 win = 0; win < 32; win++) {
        // shift the 256-bit scalar right by one byte
        w0 = __builtin_amdgcn_alignbit(w1, w0, 8); w1 = __builtin_amdgcn_alignbit(w2, w1, 8); w2 = __builtin_amdgcn_alignbit(w3, w2, 8);
        w3 = __builtin_amdgcn_alignbit(w4, w3, 8); w4 = __builtin_amdgcn_alignbit(w5, w4, 8); w5 = __builtin_amdgcn_alignbit(w6, w5, 8);
        w6 = __builtin_amdgcn_alignbit(w7, w6, 8); w7 >>= 8;
        u32 nxt[24];
        fetch(win < 31 ? win + 1 : 31, w0 & 255u, nxt);            // the last iteration's request is never used (byte 0 of window 31: a valid entry)
        ge_niels q; q.ypx = fe_unpack(cur); q.ymx = fe_unpack(cur + 8); q.xy2d = fe_unpack(cur + 16);
        acc = ge_madd(acc, q);
#pragma unroll
        for (int j = 0; j < 24; j++) cur[j] = nxt[j];
    }
    return acc;
}
__device__ __forceinline__ ge fixed_base_mul(const u32 *__restrict__ gen_tab, u32 base, const sc &k) {
    u32 w[8]; sc_to_words(w, k);
    return fixed_base_accumulate(ge_identity(), gen_tab, base, w);
}

// ---- per-proof state in device memory -------------------------------------------------------------------
__device__ __forceinline__ void transcript_load(transcript &t, const u32 *__restrict__ src /* RP_TSTATE_WORDS */) {
#pragma unroll 1
    for (int i = 0; i < 25; i++) t.s.st.lanes[i] = (u64)src[2 * i] | ((u64)src[2 * i + 1] << 32);
    t.s.pos = (u8)(src[50] & 255u); t.s.pos_begin = (u8)((src[50] >> 8) & 255u);
}
__device__ __forceinline__ void transcript_store(u32 *__restrict__ dst, const transcript &t) {
#pragma unroll 1
    for (int i = 0; i < 25; i++) { dst[2 * i] = (u32)t.s.st.lanes[i]; dst[2 * i + 1] = (u32)(t.s.st.lanes[i] >> 32); }
    dst[50] = (u32)t.s.pos | ((u32)t.s.pos_begin << 8);
}
__device__ __forceinline__ void words_to_bytes(u8 b[32], const u32 w[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) { b[4 * i] = (u8)w[i]; b[4 * i + 1] = (u8)(w[i] >> 8); b[4 * i + 2] = (u8)(w[i] >> 16); b[4 * i + 3] = (u8)(w[i] >> 24); }
}
// encodes p, stores the eight words into the proof and appends them to the transcript under `label`
__device__ __forceinline__ void emit_point(transcript &t, const char *label, const ge &p, u32 *__restrict__ proof_words) {
    u32 w[8]; ristretto_encode_words(w, p);
#pragma unroll
    for (int i = 0; i < 8; i++) proof_words[i] = w[i];
    u8 b[32]; words_to_bytes(b, w);
    t.append_point(label, b);
}
__device__ __forceinline__ void emit_scalar(transcript *t, const char *label, const sc &x, u32 *__restrict__ proof_words) {
    u32 w[8]; sc_to_words(w, x);
#pragma unroll
    for (int i = 0; i < 8; i++) proof_words[i] = w[i];
    if (t) { u8 b[32]; words_to_bytes(b, w); t->append_message(label, b, 32); }
}
__device__ __forceinline__ const u32 *rp_scalar(const u32 *sc_store, u32 tx, u32 slot) { return sc_store + ((u64)tx * RPS_COUNT + slot) * 9; }
__device__ __forceinline__ u32 *rp_scalar(u32 *sc_store, u32 tx, u32 slot) { return sc_store + ((u64)tx * RPS_COUNT + slot) * 9; }

// The vectors l(X) = l0 + l1 X and r(X) = r0 + r1 X at position i (value j = i / 64, bit k = i % 64):
//   l0 = a_L - z, l1 = s_L, r0 = y^i (a_R + z) + z^(2+j) 2^k, r1 = y^i s_R,  a_L = bit, a_R = bit - 1
struct rp_lr { sc l0, l1, r0, r1; };
__device__ __forceinline__ rp_lr rp_lr_terms(u32 i, const u64 *__restrict__ values, const u32 *__restrict__ seed8, u64 ctr0, const sc &y, const sc &z) {
    const u32 j = i >> 6, k = i & 63;
    const bool bit = (values[j] >> k) & 1;
    const sc one = sc_one(), zero = sc_zero();
    rp_lr o;
    o.l1 = drbg_scalar(seed8, ctr0 + 2 + 2ull * i);
    const sc s_r = drbg_scalar(seed8, ctr0 + 3 + 2ull * i);
    const sc y_i = sc_pow_small(y, i);
    const sc zzj = sc_pow_small(z, 2 + j);
    o.l0 = sc_sub(bit ? one : zero, z);
    const sc a_r_plus_z = bit ? z : sc_sub(z, one);
    o.r0 = sc_add(sc_mul(y_i, a_r_plus_z), sc_mul(zzj, sc_from_u64(1ull << k)));
    o.r1 = sc_mul(y_i, s_r);
    return o;
}

// ---- stage 1: A and S ---------------------------------------------------------------------------------
// wave (segment of 64 * pt positions, proof): partial sums of A and S over its positions
__global__ void __launch_bounds__(64) k_rp_as(rp_dims d, u32 pt, const u32 *__restrict__ gen_tab, const u32 *__restrict__ seeds, const u64 *__restrict__ ctr0,
                                              const u64 *__restrict__ values, u32 *__restrict__ part_a, u32 *__restrict__ part_s) {
    const u32 tx = blockIdx.y, seg = blockIdx.x, lane = threadIdx.x;
    const u32 *seed8 = seeds + 8ull * tx;
    const u64 c0 = ctr0[tx];
    ge acc_a = ge_identity(), acc_s = ge_identity();
    for (u32 k = 0; k < pt; k++) {
        const u32 i = (seg * pt + k) * 64 + lane;
        u32 w[8];
        sc_to_words(w, drbg_scalar(seed8, c0 + 2 + 2ull * i));
        acc_s = fixed_base_accumulate(acc_s, gen_tab, 2 + i, w);
        sc_to_words(w, drbg_scalar(seed8, c0 + 3 + 2ull * i));
        acc_s = fixed_base_accumulate(acc_s, gen_tab, 2 + d.table_cap + i, w);
        const bool bit = (values[(u64)tx * d.m + (i >> 6)] >> (i & 63)) & 1;
        // entry 1 of window 0 is the generator itself
        const u32 base = bit ? 2 + i : 2 + d.table_cap + i;
        ge_niels q = ge_niels_load(gen_tab + 32ull * ((u64)base * 32 * 256 + 1));
        acc_a = ge_madd_signed(acc_a, q, !bit);
    }
    acc_a = wave_point_sum(acc_a); acc_s = wave_point_sum(acc_s);
    if (lane == 0) {
        const u64 o = (u64)tx * gridDim.x + seg;
        ge_store(part_a + 32 * o, acc_a); ge_store(part_s + 32 * o, acc_s);
    }
}
// thread per proof: A, S into the proof and the transcript; challenges y, z
__global__ void __launch_bounds__(64) k_rp_yz(rp_dims d, u32 segs, const u32 *__restrict__ gen_tab, const u32 *__restrict__ seeds, const u64 *__restrict__ ctr0,
                                              const u32 *__restrict__ part_a, const u32 *__restrict__ part_s, u32 *__restrict__ tstate, u32 *__restrict__ sc_store,
                                              u32 *__restrict__ proofs) {
    const u32 tx = blockIdx.x * blockDim.x + threadIdx.x;
    if (tx >= d.n_txs) return;
    const u32 *seed8 = seeds + 8ull * tx;
    const u64 c0 = ctr0[tx];
    ge A = fixed_base_mul(gen_tab, 1, drbg_scalar(seed8, c0)), S = fixed_base_mul(gen_tab, 1, drbg_scalar(seed8, c0 + 1));
    for (u32 s = 0; s < segs; s++) {
        A = ge_add(A, ge_load(part_a + 32ull * ((u64)tx * segs + s)));
        S = ge_add(S, ge_load(part_s + 32ull * ((u64)tx * segs + s)));
    }
    transcript t; transcript_load(t, tstate + (u64)tx * RP_TSTATE_WORDS);
    u32 *proof = proofs + (u64)tx * 8 * (9 + 2 * d.lg);
    emit_point(t, "A", A, proof); emit_point(t, "S", S, proof + 8);
    const sc y = t.challenge_scalar("y"), z = t.challenge_scalar("z");
    store_sc_limbs(rp_scalar(sc_store, tx, RPS_Y), y); store_sc_limbs(rp_scalar(sc_store, tx, RPS_Z), z);
    store_sc_limbs(rp_scalar(sc_store, tx, RPS_YINV), sc_invert(y));
    transcript_store(tstate + (u64)tx * RP_TSTATE_WORDS, t);
}

// ---- stage 2: t(X) = <l(X), r(X)> ----------------------------------------------------------------------
// workgroup per proof: t0 = <l0, r0>, t1 = <l0, r1> + <l1, r0>, t2 = <l1, r1>
__global__ void __launch_bounds__(256) k_rp_poly(rp_dims d, const u32 *__restrict__ seeds, const u64 *__restrict__ ctr0, const u64 *__restrict__ values,
                                                 u32 *__restrict__ sc_store) {
    __shared__ u32 part[4][3][9];
    const u32 tx = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const sc y = load_sc_limbs(rp_scalar(sc_store, tx, RPS_Y)), z = load_sc_limbs(rp_scalar(sc_store, tx, RPS_Z));
    sc t0 = sc_zero(), t1 = sc_zero(), t2 = sc_zero();
    for (u32 i = threadIdx.x; i < d.nm; i += blockDim.x) {
        rp_lr v = rp_lr_terms(i, values + (u64)tx * d.m, seeds + 8ull * tx, ctr0[tx], y, z);
        t0 = sc_reduce_once(sc_add(t0, sc_mul(v.l0, v.r0)));
        t1 = sc_reduce_once(sc_add(t1, sc_add(sc_mul(v.l0, v.r1), sc_mul(v.l1, v.r0))));
        t2 = sc_reduce_once(sc_add(t2, sc_mul(v.l1, v.r1)));
    }
    t0 = wave_scalar_sum(t0); t1 = wave_scalar_sum(t1); t2 = wave_scalar_sum(t2);
    if (lane == 0) { store_sc_limbs(part[wave][0], t0); store_sc_limbs(part[wave][1], t1); store_sc_limbs(part[wave][2], t2); }
    __syncthreads();
    if (threadIdx.x < 3) {
        sc acc = load_sc_limbs(part[0][threadIdx.x]);
        for (u32 w = 1; w < (blockDim.x >> 6); w++) acc = sc_reduce_once(sc_add(acc, load_sc_limbs(part[w][threadIdx.x])));
        store_sc_limbs(rp_scalar(sc_store, tx, RPS_T0 + threadIdx.x), acc);
    }
}
// thread per proof: T_1, T_2, challenge x, t_x, t_x_blinding, e_blinding, challenge w, header of the inner-product argument
__global__ void __launch_bounds__(64) k_rp_tx(rp_dims d, const u32 *__restrict__ gen_tab, const u32 *__restrict__ seeds, const u64 *__restrict__ ctr0,
                                              const u32 *__restrict__ openings, u32 *__restrict__ tstate, u32 *__restrict__ sc_store, u32 *__restrict__ proofs) {
    const u32 tx = blockIdx.x * blockDim.x + threadIdx.x;
    if (tx >= d.n_txs) return;
    const u32 *seed8 = seeds + 8ull * tx;
    const u64 c0 = ctr0[tx];
    const sc t0 = load_sc_limbs(rp_scalar(sc_store, tx, RPS_T0)), t1 = load_sc_limbs(rp_scalar(sc_store, tx, RPS_T1)), t2 = load_sc_limbs(rp_scalar(sc_store, tx, RPS_T2));
    const sc z = load_sc_limbs(rp_scalar(sc_store, tx, RPS_Z));
    const sc t1_bl = drbg_scalar(seed8, c0 + 2 + 2ull * d.nm), t2_bl = drbg_scalar(seed8, c0 + 3 + 2ull * d.nm);
    u32 w8[8];
    sc_to_words(w8, t1); ge T1 = fixed_base_accumulate(ge_identity(), gen_tab, 0, w8);
    sc_to_words(w8, t1_bl); T1 = fixed_base_accumulate(T1, gen_tab, 1, w8);
    sc_to_words(w8, t2); ge T2 = fixed_base_accumulate(ge_identity(), gen_tab, 0, w8);
    sc_to_words(w8, t2_bl); T2 = fixed_base_accumulate(T2, gen_tab, 1, w8);
    transcript t; transcript_load(t, tstate + (u64)tx * RP_TSTATE_WORDS);
    u32 *proof = proofs + (u64)tx * 8 * (9 + 2 * d.lg);
    emit_point(t, "T_1", T1, proof + 16); emit_point(t, "T_2", T2, proof + 24);
    const sc x = t.challenge_scalar("x"), xx = sc_sq(x);
    // t0_blinding = sum_j z^(2+j) * opening_j
    sc t0_bl = sc_zero(), zp = sc_sq(z);
    for (u32 j = 0; j < d.m; j++) {
        t0_bl = sc_reduce_once(sc_add(t0_bl, sc_mul(zp, load_scalar_words(openings + 8ull * ((u64)tx * d.m + j)))));
        zp = sc_mul(zp, z);
    }
    const sc t_x = sc_add(sc_add(t0, sc_mul(t1, x)), sc_mul(t2, xx));
    const sc t_x_bl = sc_add(sc_add(t0_bl, sc_mul(t1_bl, x)), sc_mul(t2_bl, xx));
    const sc e_bl = sc_add(drbg_scalar(seed8, c0), sc_mul(drbg_scalar(seed8, c0 + 1), x));
    emit_scalar(&t, "t_x", t_x, proof + 32); emit_scalar(&t, "t_x_blinding", t_x_bl, proof + 40); emit_scalar(&t, "e_blinding", e_bl, proof + 48);
    const sc w = t.challenge_scalar("w");
    t.append_message("dom-sep", (const u8 *)"ipp v1", 6);
    t.append_u64("n", d.nm);
    store_sc_limbs(rp_scalar(sc_store, tx, RPS_X), x); store_sc_limbs(rp_scalar(sc_store, tx, RPS_W), w);
    transcript_store(tstate + (u64)tx * RP_TSTATE_WORDS, t);
}
// thread per (position, proof): a_i = l(x)_i, b_i = r(x)_i, y^-i
__global__ void __launch_bounds__(64) k_rp_ab(rp_dims d, const u32 *__restrict__ seeds, const u64 *__restrict__ ctr0, const u64 *__restrict__ values,
                                              const u32 *__restrict__ sc_store, u32 *__restrict__ vec_a, u32 *__restrict__ vec_b, u32 *__restrict__ vec_yinv) {
    const u32 tx = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.nm) return;
    const sc y = load_sc_limbs(rp_scalar(sc_store, tx, RPS_Y)), z = load_sc_limbs(rp_scalar(sc_store, tx, RPS_Z)), x = load_sc_limbs(rp_scalar(sc_store, tx, RPS_X));
    rp_lr v = rp_lr_terms(i, values + (u64)tx * d.m, seeds + 8ull * tx, ctr0[tx], y, z);
    const u64 o = 9ull * ((u64)tx * d.nm + i);
    store_sc_limbs(vec_a + o, sc_reduce_once(sc_add(v.l0, sc_mul(v.l1, x))));
    store_sc_limbs(vec_b + o, sc_reduce_once(sc_add(v.r0, sc_mul(v.r1, x))));
    store_sc_limbs(vec_yinv + o, sc_pow_small(load_sc_limbs(rp_scalar(sc_store, tx, RPS_YINV)), i));
}

// ---- stage 3: the inner-product argument, round = 1 .. lg ----------------------------------------------
// workgroup per proof: c_L = <a_lo, b_hi>, c_R = <a_hi, b_lo> over the current length 2 * len
__global__ void __launch_bounds__(256) k_rp_inner(rp_dims d, u32 len, const u32 *__restrict__ vec_a, const u32 *__restrict__ vec_b, u32 *__restrict__ sc_store) {
    __shared__ u32 part[4][2][9];
    const u32 tx = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 *a = vec_a + 9ull * (u64)tx * d.nm, *b = vec_b + 9ull * (u64)tx * d.nm;
    sc cl = sc_zero(), cr = sc_zero();
    for (u32 i = threadIdx.x; i < len; i += blockDim.x) {
        cl = sc_reduce_once(sc_add(cl, sc_mul(load_sc_limbs(a + 9ull * i), load_sc_limbs(b + 9ull * (len + i)))));
        cr = sc_reduce_once(sc_add(cr, sc_mul(load_sc_limbs(a + 9ull * (len + i)), load_sc_limbs(b + 9ull * i))));
    }
    cl = wave_scalar_sum(cl); cr = wave_scalar_sum(cr);
    if (lane == 0) { store_sc_limbs(part[wave][0], cl); store_sc_limbs(part[wave][1], cr); }
    __syncthreads();
    if (threadIdx.x < 2) {
        sc acc = load_sc_limbs(part[0][threadIdx.x]);
        for (u32 w = 1; w < (blockDim.x >> 6); w++) acc = sc_reduce_once(sc_add(acc, load_sc_limbs(part[w][threadIdx.x])));
        store_sc_limbs(rp_scalar(sc_store, tx, RPS_CL + threadIdx.x), acc);
    }
}
// wave ((class, slice), proof); class = side (G, H) x output (L, R); nm / 2 generators per class, 64 * pt per wave
__global__ void __launch_bounds__(64) k_rp_lr(rp_dims d, u32 round, u32 pt, u32 waves_per_class, const u32 *__restrict__ gen_tab, const u32 *__restrict__ vec_a,
                                              const u32 *__restrict__ vec_b, const u32 *__restrict__ vec_yinv, const u32 *__restrict__ sc_store,
                                              u32 *__restrict__ part_lr) {
    const u32 tx = blockIdx.y, lane = threadIdx.x;
    const u32 cls = blockIdx.x / waves_per_class, slice = blockIdx.x % waves_per_class;
    const u32 side = cls >> 1, out = cls & 1;
    const u32 hb = d.lg - round, len = 1u << hb, cnt = d.nm >> 1;
    // G_p feeds L when p lies in the upper half of the current vector, H_p when it lies in the lower half
    const u32 bit = side == 0 ? (out == 0 ? 1u : 0u) : (out == 0 ? 0u : 1u);
    const u32 *vec = (side == 0 ? vec_a : vec_b) + 9ull * (u64)tx * d.nm;
    ge acc = ge_identity();
    for (u32 k = 0; k < pt; k++) {
        const u32 e = (slice * pt + k) * 64 + lane;
        if (e < cnt) {
            const u32 plow = e & (len - 1), p = ((e >> hb) << (hb + 1)) | (bit << hb) | plow;
            // the vector entry paired with the generator: the other half of the current vector
            sc s = load_sc_limbs(vec + 9ull * (bit ? plow : len + plow));
            if (side) s = sc_mul(s, load_sc_limbs(vec_yinv + 9ull * ((u64)tx * d.nm + p)));
            for (u32 r = 1; r < round; r++) {
                const u32 pb = (p >> (d.lg - r)) & 1;
                const bool take_u = side == 0 ? pb != 0 : pb == 0;
                s = sc_mul(s, load_sc_limbs(rp_scalar(sc_store, tx, (take_u ? RPS_U : RPS_UINV) + r - 1)));
            }
            u32 w[8]; sc_to_words(w, s);
            acc = fixed_base_accumulate(acc, gen_tab, 2 + (side ? d.table_cap : 0) + p, w);
        }
    }
    acc = wave_point_sum(acc);
    if (lane == 0) ge_store(part_lr + 32ull * (((u64)tx * 2 + out) * (2 * waves_per_class) + side * waves_per_class + slice), acc);
}
// thread per proof: L, R into the proof and the transcript, challenge u and its inverse
__global__ void __launch_bounds__(64) k_rp_round(rp_dims d, u32 round, u32 slots, const u32 *__restrict__ gen_tab, const u32 *__restrict__ part_lr, u32 *__restrict__ tstate,
                                                 u32 *__restrict__ sc_store, u32 *__restrict__ proofs) {
    const u32 tx = blockIdx.x * blockDim.x + threadIdx.x;
    if (tx >= d.n_txs) return;
    const sc w = load_sc_limbs(rp_scalar(sc_store, tx, RPS_W));
    // c_L * Q and c_R * Q with Q = w * B
    ge L = fixed_base_mul(gen_tab, 0, sc_mul(load_sc_limbs(rp_scalar(sc_store, tx, RPS_CL)), w));
    ge R = fixed_base_mul(gen_tab, 0, sc_mul(load_sc_limbs(rp_scalar(sc_store, tx, RPS_CR)), w));
    for (u32 s = 0; s < slots; s++) {
        L = ge_add(L, ge_load(part_lr + 32ull * (((u64)tx * 2 + 0) * slots + s)));
        R = ge_add(R, ge_load(part_lr + 32ull * (((u64)tx * 2 + 1) * slots + s)));
    }
    transcript t; transcript_load(t, tstate + (u64)tx * RP_TSTATE_WORDS);
    u32 *proof = proofs + (u64)tx * 8 * (9 + 2 * d.lg) + 56 + 16 * (round - 1);
    emit_point(t, "L", L, proof); emit_point(t, "R", R, proof + 8);
    const sc u = t.challenge_scalar("u");
    store_sc_limbs(rp_scalar(sc_store, tx, RPS_U + round - 1), u);
    store_sc_limbs(rp_scalar(sc_store, tx, RPS_UINV + round - 1), sc_invert(u));
    transcript_store(tstate + (u64)tx * RP_TSTATE_WORDS, t);
}
// thread per (position below len, proof): a_i = u a_i + u^-1 a_{len+i}, b_i = u^-1 b_i + u b_{len+i}; the last round writes a, b into the proof
__global__ void __launch_bounds__(64) k_rp_fold(rp_dims d, u32 round, u32 len, u32 *__restrict__ vec_a, u32 *__restrict__ vec_b, const u32 *__restrict__ sc_store,
                                                u32 *__restrict__ proofs) {
    const u32 tx = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const sc u = load_sc_limbs(rp_scalar(sc_store, tx, RPS_U + round - 1)), ui = load_sc_limbs(rp_scalar(sc_store, tx, RPS_UINV + round - 1));
    u32 *a = vec_a + 9ull * (u64)tx * d.nm, *b = vec_b + 9ull * (u64)tx * d.nm;
    const sc na = sc_reduce_once(sc_add(sc_mul(load_sc_limbs(a + 9ull * i), u), sc_mul(ui, load_sc_limbs(a + 9ull * (len + i)))));
    const sc nb = sc_reduce_once(sc_add(sc_mul(load_sc_limbs(b + 9ull * i), ui), sc_mul(u, load_sc_limbs(b + 9ull * (len + i)))));
    store_sc_limbs(a + 9ull * i, na); store_sc_limbs(b + 9ull * i, nb);
    if (len == 1) {
        u32 *proof = proofs + (u64)tx * 8 * (9 + 2 * d.lg) + 56 + 16 * d.lg;
        emit_scalar(nullptr, "", na, proof); emit_scalar(nullptr, "", nb, proof + 8);
    }
}

void launch_range_proofs(hipStream_t st, const rp_dims &d, const u32 *gen_tab, const rp_buffers &b) {
    const u32 pt = d.nm >= 256 ? 4u : d.nm / 64, segs = d.nm / (64 * pt), tx_blocks = grid_for(d.n_txs, 64), vec_threads = d.nm < 256 ? d.nm : 256;
    hipLaunchKernelGGL(k_rp_as, dim3(segs, d.n_txs), dim3(64), 0, st, d, pt, gen_tab, b.seeds, b.ctr0, b.values, b.part_a, b.part_s);
    hipLaunchKernelGGL(k_rp_yz, dim3(tx_blocks), dim3(64), 0, st, d, segs, gen_tab, b.seeds, b.ctr0, b.part_a, b.part_s, b.tstate, b.sc_store, b.proofs);
    hipLaunchKernelGGL(k_rp_poly, dim3(d.n_txs), dim3(vec_threads), 0, st, d, b.seeds, b.ctr0, b.values, b.sc_store);
    hipLaunchKernelGGL(k_rp_tx, dim3(tx_blocks), dim3(64), 0, st, d, gen_tab, b.seeds, b.ctr0, b.openings, b.tstate, b.sc_store, b.proofs);
    hipLaunchKernelGGL(k_rp_ab, dim3(d.nm / 64, d.n_txs), dim3(64), 0, st, d, b.seeds, b.ctr0, b.values, b.sc_store, b.vec_a, b.vec_b, b.vec_yinv);
    const u32 cnt = d.nm / 2, lr_pt = cnt >= 256 ? 4u : (cnt >= 64 ? cnt / 64 : 1u), wpc = (cnt + 64 * lr_pt - 1) / (64 * lr_pt);
    for (u32 round = 1; round <= d.lg; round++) {
        const u32 len = d.nm >> round;
        hipLaunchKernelGGL(k_rp_inner, dim3(d.n_txs), dim3(len < 256 ? (len < 64 ? 64 : len) : 256), 0, st, d, len, b.vec_a, b.vec_b, b.sc_store);
        hipLaunchKernelGGL(k_rp_lr, dim3(4 * wpc, d.n_txs), dim3(64), 0, st, d, round, lr_pt, wpc, gen_tab, b.vec_a, b.vec_b, b.vec_yinv, b.sc_store, b.part_lr);
        hipLaunchKernelGGL(k_rp_round, dim3(tx_blocks), dim3(64), 0, st, d, round, 2 * wpc, gen_tab, b.part_lr, b.tstate, b.sc_store, b.proofs);
        hipLaunchKernelGGL(k_rp_fold, dim3(grid_for(len, 64), d.n_txs), dim3(64), 0, st, d, round, len, b.vec_a, b.vec_b, b.sc_store, b.proofs);
    }
}
u32 range_proof_lr_slots(u32 nm) {
    const u32 cnt = nm / 2, lr_pt = cnt >= 256 ? 4u : (cnt >= 64 ? cnt / 64 : 1u);
    return 2 * ((cnt + 64 * lr_pt - 1) / (64 * lr_pt));
}
u32 range_proof_as_segments(u32 nm) { const u32 pt = nm >= 256 ? 4u : nm / 64; return nm / (64 * pt); }

}  // namespace xhe
