// Schnorr signature verification (/root/reference/src/elgamal.rs:38-65): accept iff
// e == SHA3-512(pk || msg || enc(s*H - e*P)) mod l.  Four parts:
//   key tables     a wavefront per distinct signing key (k_sig_key_tables): the multiples 0..8 of 2^(64 j) P, j < 4 -- a key
//                  signs every transaction of its account's chain, so the 192 doublings are paid once per key (batches with
//                  few signatures per key skip this part and use k_sig_point_absorb_own_table);
//   absorb         thread per signature: absorbs the whole 72-byte blocks of pk || msg into a Keccak state
//                  held in registers (the bulk of the hashing: 19 of the 20 permutations for a one-transfer
//                  transaction) -- needs no point, so it runs beside the point part, as the other kind of workgroup of
//                  the same launch (k_sig_point_absorb);
//   point          four lanes per signature (xhe_quad.h): s*H from the 8-bit fixed-base table, e*P with signed
//                  4-bit digits over the key's table (16 rounds of 4 doublings + 4 additions), R = s*H - e*P as an XYZT record;
//   k_sig_final    thread per signature: encodes R, absorbs the message tail || enc(R) || padding, squeezes,
//                  reduces mod l and compares with e.
// A multisig co-signature hashes the 32-byte BLAKE3 digest of the pre-image instead of the message
// (/root/reference/src/tx/verify.rs:259-292): it has no whole block before R, so only the last kernel hashes.
#include <stdlib.h>
#include "xhe_settings.h"
#include "xhe_dev_common.h"
#include "xhe_kernels.h"
#include "xhe_quad.h"
#include "xhe_wide.h"

namespace xhe {

static constexpr u32 SIG_RATE_WORDS = 9;      // SHA3-512: 72-byte rate
// The signatures run in the background of the main path: their kernels stay resident with a bounded number of wavefronts
// (grid-stride beyond that) and within 128 vector registers, the size of the bulk decode's wavefronts, so that a signature
// wavefront takes the place of exactly one of the four decode wavefronts of its SIMD.  Measured on the 100 000 x 8-transfer
// batch: 165 registers / 1024 wavefronts 24.8 ms per step, 165 / 256 22.9 ms, 128 / 1024 22.8 ms, 128 / 512 22.3 ms, 128 / 256
// 22.6 ms (with fewer the signatures end after everything else).  With the other kernels as they are now that batch no longer
// cares (19.0-19.7 ms from 384 to 4096 wavefronts), but batches in which the signatures are a larger share of the work do: at
// 512 they ended long after everything else - 100 000 mixed transactions (no key tables: 1.7 of 5.3 G instructions) 17.2 ms
// against 10.9 ms at 2048, 125 000 one-transfer transactions 12.2 against 10.3 ms, 32 768 of those 3.42 against 3.04 ms.
// XHE_SIG_MAX_WAVES overrides the bound.
static constexpr u32 SIG_MAX_WAVES_DEFAULT = 2048;
static u32 sig_max_waves() {
    const u32 v = [] { const char *e = setting("XHE_SIG_MAX_WAVES"); u32 x = e ? (u32)strtoul(e, nullptr, 10) : 0; return x >= 64 && x <= 65536 ? x : SIG_MAX_WAVES_DEFAULT; }();
    return v;
}

// byte j of pk || msg (msg = the digest for a co-signature)
struct sig_stream {
    const u8 *key, *msg;
    __device__ __forceinline__ u8 at(u32 j) const { return j < 32 ? key[j] : msg[j - 32]; }
};
__device__ __forceinline__ sig_stream sig_stream_of(const dev_sig &sg, const u8 *__restrict__ bytes, const u8 *__restrict__ ms_hash) {
    sig_stream s;
    s.key = bytes + sg.key_off;
    s.msg = sg.hash_idx != 0xffffffffu ? ms_hash + 32ull * sg.hash_idx : bytes + sg.msg_off;
    return s;
}
__device__ __forceinline__ u32 sig_whole_blocks(const dev_sig &sg) { return sg.hash_idx != 0xffffffffu ? 0u : (32u + sg.msg_len) / 72u; }

// state[w * n + i]: word w of signature i's sponge after its whole blocks
__device__ __forceinline__ void sig_absorb_work(u32 block, u32 n_blocks, const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n, u64 *__restrict__ state) {
    for (u32 i = block * 64 + threadIdx.x; i < n; i += n_blocks * 64) {
    const dev_sig sg = sigs[i];
    const u32 nb = sig_whole_blocks(sg);
    const u8 *key = bytes + sg.key_off, *msg = bytes + sg.msg_off;
    u64 a[25];
#pragma unroll
    for (int w = 0; w < 25; w++) a[w] = 0;
    for (u32 b = 0; b < nb; b++) {
        // block 0 starts with the 32 key bytes; word w of block b holds the message bytes from 72 b + 8 w - 32
        const u8 *src = msg + 72ull * b;
#pragma unroll
        for (int w = 0; w < (int)SIG_RATE_WORDS; w++) {
            const u8 *p = (b == 0 && w < 4) ? key + 8 * w : src + 8 * w - 32;
            u64 word = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) word |= (u64)p[k] << (8 * k);
            a[w] ^= word;
        }
        keccak_f1600_regs(a);
    }
#pragma unroll
    for (int w = 0; w < 25; w++) state[(u64)w * n + i] = a[w];
    }
}

// ---- R = s*H - e*P on a quad of lanes ---------------------------------------------------------------
// this lane's multiplicand of a "cached" point (Y - X, Y + X, Z, 2d T), all of magnitude 1
__device__ __forceinline__ fe qd_to_cached(const qd_point &p, u32 role) {
    fe c = fe_reduce(qd_add_operand(p.c, role));
    fe t = fe_mul(c, fe_c_d2());
    fe out;
#pragma unroll
    for (int i = 0; i < 10; i++) out.v[i] = role == 3 ? t.v[i] : c.v[i];
    return out;
}
// p + q (or p - q) with q given by this lane's cached multiplicand
__device__ __forceinline__ qd_point qd_add_cached(const qd_point &p, const fe &cached, bool negative, u32 role) {
    // the negated point swaps (Y - X) with (Y + X) and negates 2d T
    fe neg = fe_negate(cached);
    fe b;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        u32 swapped = qd_pull<XHE_QP(1, 0, 2, 3)>(cached.v[i]);
        b.v[i] = negative ? (role == 3 ? neg.v[i] : swapped) : cached.v[i];
    }
    return qd_add_finish(fe_mul(qd_add_operand(p.c, role), b), role);
}

static constexpr u32 SIG_QUADS = 16;          // signatures per 64-thread workgroup

// Table of one key per wavefront (xhe_wide.h: a lane per limb, a row of lanes per coordinate): for j < SIG_KEY_POS the multiples
// 0..8 of 2^(64 j) P as the multiplicands (Y - X, Y + X, Z, 2d T) the point kernel's quads add from.  Keys sign many
// transactions of a batch (an account's whole chain), and with the table a signature costs 64 doublings + 64 additions instead
// of 256 + 72.  The 192 doublings of a key are one dependent chain, so its length in instructions is what this kernel costs:
// ~170 per doubling in this form against ~420 on a quad of lanes.  (An undecodable key left the identity in pts[];
// k_sig_final reports it.)
__global__ void __launch_bounds__(64) k_sig_key_tables(const u32 *__restrict__ sig_keys, u32 n_keys, const u32 *__restrict__ pts, u32 *__restrict__ key_tab) {
    const wd_lane L = wd_lane_init();
    const u32 d2 = wd_const_d2(L);
    // the chain is the head of the signatures' critical path and one wavefront per SIMD wide: let it issue ahead of the
    // bulk kernels it shares the SIMD with
    __builtin_amdgcn_s_setprio(3);
    for (u32 k = blockIdx.x; k < n_keys; k += gridDim.x) {
        ge_niels pn = ge_niels_load(pts + 32ull * sig_keys[k]);
        wd_point cur = wd_from_ge(ge_from_niels_affine(pn), L);
#pragma unroll 1
        for (u32 j = 0; j < SIG_KEY_POS; j++) {
            const wd_cached step = wd_to_cached(cur, L);
            wd_point m; m.v = ((L.row == 1 || L.row == 2) && L.k == 0) ? 1u : 0u;          // the identity (0, 1, 1, 0)
            u32 *dst = key_tab + ((u64)k * SIG_KEY_POS + j) * 9 * SIG_KEY_ENTRY_WORDS + 10 * L.row + L.k;
#pragma unroll 1
            for (u32 mult = 0; mult < 9; mult++) {
                u32 x, y, z, t;
                wd_rows(m.v, x, y, z, t);
                u32 td = wd_mul(t, d2, L);
                u32 sums = wd_weak_reduce(L.row == 0 ? y + (L.p2 - x) : y + x, L);
                u32 entry = L.row < 2 ? sums : L.row == 2 ? z : td;
                if (L.live) dst[mult * SIG_KEY_ENTRY_WORDS] = entry;
                if (mult < 8) m = wd_add(m, step, L);
            }
            if (j + 1 < SIG_KEY_POS) {
#pragma unroll 1
                for (u32 d = 0; d < 256 / SIG_KEY_POS; d++) cur = wd_double(cur, L);
            }
        }
    }
}

__device__ __forceinline__ void sig_point_work(u32 block, u32 n_blocks, const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n,
                                               const u32 *__restrict__ key_tab, const u32 *__restrict__ htab, u32 *__restrict__ r_out) {
    const u32 lane = threadIdx.x, role = lane & 3;
    constexpr u32 ROUNDS = 64 / SIG_KEY_POS;   // 4-bit digits per position
    for (u32 group = block; group * SIG_QUADS < n; group += n_blocks) {
    const u32 q = group * SIG_QUADS + (lane >> 2);
    const u32 i = q < n ? q : n - 1;           // spare quads repeat the last signature and store nothing
    const dev_sig sg = sigs[i];
    u32 sw[8], ew[8];
    load_words_unaligned(sw, bytes + sg.sig_off); load_words_unaligned(ew, bytes + sg.sig_off + 32);
    const u32 *tab = key_tab + (u64)sg.key_slot * SIG_KEY_TABLE_WORDS + 10 * role;
    // Q = e * P with signed radix-16 digits in [-8, 8]: digit 16 j + r of e multiplies 2^(64 j) P, so the SIG_KEY_POS positions
    // share the doublings.  The carry into digit k depends on all lower digits, so the carries are worked out first, one bit
    // per digit.
    qd_point Q = qd_identity(role);
    {
        u64 carry_bits = 0;
        u32 carry = 0;
        for (u32 k = 0; k < 64; k++) {
            carry_bits |= (u64)carry << k;
            u32 v = ((ew[k >> 3] >> ((k & 7) * 4)) & 15) + carry;
            carry = v > 8 ? 1u : 0u;
        }
#pragma unroll 1
        for (int r = (int)ROUNDS - 1; r >= 0; r--) {
            // this round's table entries are requested before the doublings
            fe c[SIG_KEY_POS]; bool negative[SIG_KEY_POS];
#pragma unroll
            for (u32 j = 0; j < SIG_KEY_POS; j++) {
                const u32 k = ROUNDS * j + (u32)r;
                u32 v = ((ew[k >> 3] >> ((k & 7) * 4)) & 15) + (u32)((carry_bits >> k) & 1);
                negative[j] = v > 8;
                u32 mag = negative[j] ? 16 - v : v;
                const u32 *src = tab + (j * 9 + mag) * SIG_KEY_ENTRY_WORDS;
#pragma unroll
                for (int l = 0; l < 10; l++) c[j].v[l] = src[l];
            }
            Q = qd_double(qd_double(qd_double(qd_double(Q, role), role), role), role);
#pragma unroll
            for (u32 j = 0; j < SIG_KEY_POS; j++) Q = qd_add_cached(Q, c[j], negative[j], role);
        }
    }
    // R = s * H (32 windows of 8 bits, fixed-base table) - Q
    qd_point R = qd_identity(role);
#pragma unroll 1
    for (u32 w = 0; w < 32; w++) {
        u32 b = (sw[w >> 2] >> ((w & 3) * 8)) & 255;
        R = qd_madd(R, htab + 32ull * (w * 256 + b), false, role);
    }
    R = qd_add_cached(R, qd_to_cached(Q, role), true, role);
    if (q < n) qd_store(r_out + 32ull * q, R, role);
    }
}

// The same with the table of multiples 0..8 of P built per signature in LDS (256 doublings + 72 additions): for batches whose
// keys sign too few transactions each for a table per key to pay.
__device__ __forceinline__ void sig_point_work_own_table(u32 block, u32 n_blocks, u32 (*tab)[10][64], const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n,
                                               const u32 *__restrict__ pts, const u32 *__restrict__ htab, u32 *__restrict__ r_out) {
    const u32 lane = threadIdx.x, role = lane & 3;
    for (u32 group = block; group * SIG_QUADS < n; group += n_blocks) {
    const u32 q = group * SIG_QUADS + (lane >> 2);
    const u32 i = q < n ? q : n - 1;           // spare quads repeat the last signature and store nothing
    const dev_sig sg = sigs[i];
    u32 sw[8], ew[8];
    load_words_unaligned(sw, bytes + sg.sig_off); load_words_unaligned(ew, bytes + sg.sig_off + 32);

    // table of k * P, k = 0..8 (an undecodable key left the identity in pts[]; the last kernel reports it)
    {
        ge_niels pn = ge_niels_load(pts + 32ull * sg.key_pt);
        qd_point P = qd_from_ge(ge_from_niels_affine(pn), role);
        // multiples 0..8 of P, each written to the table as soon as it exists (one running point instead of nine live ones
        // keeps the kernel's register count low enough to share a SIMD with two decode wavefronts)
        qd_point cur = qd_identity(role);
#pragma unroll 1
        for (int k = 0; k < 9; k++) {
            fe c = qd_to_cached(cur, role);
#pragma unroll
            for (int l = 0; l < 10; l++) tab[k][l][lane] = c.v[l];
            cur = qd_add(cur, P, role);
        }
    }
    // Q = e * P with signed radix-16 digits in [-8, 8], most significant first.  The carry into digit k depends on
    // all lower digits, so the carries are worked out first, one bit per digit.
    qd_point Q = qd_identity(role);
    {
        u64 carry_bits = 0;
        u32 carry = 0;
        for (u32 k = 0; k < 64; k++) {
            carry_bits |= (u64)carry << k;
            u32 v = ((ew[k >> 3] >> ((k & 7) * 4)) & 15) + carry;
            carry = v > 8 ? 1u : 0u;
        }
#pragma unroll 1
        for (int k = 63; k >= 0; k--) {
            Q = qd_double(qd_double(qd_double(qd_double(Q, role), role), role), role);
            u32 v = ((ew[k >> 3] >> ((k & 7) * 4)) & 15) + (u32)((carry_bits >> k) & 1);
            bool negative = v > 8;
            u32 mag = negative ? 16 - v : v;
            fe c;
#pragma unroll
            for (int l = 0; l < 10; l++) c.v[l] = tab[mag][l][lane];
            Q = qd_add_cached(Q, c, negative, role);
        }
    }
    // R = s * H (32 windows of 8 bits, fixed-base table) - Q
    qd_point R = qd_identity(role);
#pragma unroll 1
    for (u32 w = 0; w < 32; w++) {
        u32 b = (sw[w >> 2] >> ((w & 3) * 8)) & 255;
        R = qd_madd(R, htab + 32ull * (w * 256 + b), false, role);
    }
    R = qd_add_cached(R, qd_to_cached(Q, role), true, role);
    if (q < n) qd_store(r_out + 32ull * q, R, role);
    }
}

// One launch for the two independent parts: workgroups [0, point_blocks) compute the points, the others absorb the
// message blocks (both grid-stride over their share).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_sig_point_absorb(u32 point_blocks, const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n,
                                                         const u32 *__restrict__ key_tab, const u32 *__restrict__ htab, u32 *__restrict__ r_out, u64 *__restrict__ state) {
    if (blockIdx.x < point_blocks) sig_point_work(blockIdx.x, point_blocks, bytes, sigs, n, key_tab, htab, r_out);
    else sig_absorb_work(blockIdx.x - point_blocks, gridDim.x - point_blocks, bytes, sigs, n, state);
}
__global__ void __launch_bounds__(64) k_sig_point_absorb_own_table(u32 point_blocks, const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n,
                                                                   const u32 *__restrict__ pts, const u32 *__restrict__ htab, u32 *__restrict__ r_out, u64 *__restrict__ state) {
    __shared__ u32 tab[9][10][64];            // multiples 0..8 of P: [multiple][limb][lane], each lane its own cached multiplicand
    if (blockIdx.x < point_blocks) sig_point_work_own_table(blockIdx.x, point_blocks, tab, bytes, sigs, n, pts, htab, r_out);
    else sig_absorb_work(blockIdx.x - point_blocks, gridDim.x - point_blocks, bytes, sigs, n, state);
}

__global__ void __launch_bounds__(64) k_sig_final(const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n, const u64 *__restrict__ state,
                                                  const u32 *__restrict__ r_in, const u8 *__restrict__ ok, const u8 *__restrict__ ms_hash, u32 *__restrict__ tx_err) {
    __shared__ u8 tail[2 * SIG_RATE_WORDS][64][8];       // [word][lane][byte]: the last one or two blocks of each signature
    const u32 lane = threadIdx.x;
    u32 i = blockIdx.x * blockDim.x + lane;
    if (i >= n) return;
    const dev_sig sg = sigs[i];
    if (!ok[sg.key_pt]) { if (sg.key_err_key) raise_err(tx_err, sg.tx, sg.key_err_key); return; }
    u32 rw[8]; ristretto_encode_words(rw, ge_load(r_in + 32ull * i));
    // tail = rest of pk || msg, enc(R), 0x06, zeros, 0x80 at the end of the last block
    const sig_stream st = sig_stream_of(sg, bytes, ms_hash);
    const u32 total = 32 + sg.msg_len, done = 72 * sig_whole_blocks(sg), rest = total - done;     // rest < 72, or 64 for a co-signature
    const u32 blocks = (rest + 32) / 72 + 1;
#pragma unroll
    for (int w = 0; w < 2 * (int)SIG_RATE_WORDS; w++) *(u64 *)tail[w][lane] = 0;
    for (u32 j = 0; j < rest; j++) tail[j >> 3][lane][j & 7] = st.at(done + j);
    for (u32 j = 0; j < 32; j++) { u32 o = rest + j; tail[o >> 3][lane][o & 7] = (u8)(rw[j >> 2] >> ((j & 3) * 8)); }
    { u32 o = rest + 32; tail[o >> 3][lane][o & 7] ^= 0x06; o = 72 * blocks - 1; tail[o >> 3][lane][o & 7] ^= 0x80; }
    u64 a[25];
#pragma unroll
    for (int w = 0; w < 25; w++) a[w] = state[(u64)w * n + i];
#pragma unroll
    for (int w = 0; w < (int)SIG_RATE_WORDS; w++) a[w] ^= *(const u64 *)tail[w][lane];
    keccak_f1600_regs(a);
    if (blocks == 2) {
#pragma unroll
        for (int w = 0; w < (int)SIG_RATE_WORDS; w++) a[w] ^= *(const u64 *)tail[SIG_RATE_WORDS + w][lane];
        keccak_f1600_regs(a);
    }
    u32 hw[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { hw[2 * k] = (u32)a[k]; hw[2 * k + 1] = (u32)(a[k] >> 32); }
    u32 cw[8]; sc_to_words(cw, sc_from_wide_words(hw));
    u32 diff = 0;
    for (int k = 0; k < 8; k++) {
        const u8 *e = bytes + sg.sig_off + 32 + 4 * k;
        diff |= cw[k] ^ ((u32)e[0] | ((u32)e[1] << 8) | ((u32)e[2] << 16) | ((u32)e[3] << 24));
    }
    if (diff) raise_err(tx_err, sg.tx, sg.err_key);
}

void launch_sig_key_tables(hipStream_t st, const u32 *sig_keys, u32 n_keys, const u32 *pts, u32 *key_tab) {
    if (!n_keys) return;
    hipLaunchKernelGGL(k_sig_key_tables, dim3(n_keys < sig_max_waves() ? n_keys : sig_max_waves()), dim3(64), 0, st, sig_keys, n_keys, pts, key_tab);
}
void launch_sig_point_absorb(hipStream_t st, const u8 *bytes, const dev_sig *sigs, u32 n, const u32 *key_tab, const u32 *pts, const u32 *htab, u32 *r_out, u64 *state) {
    if (!n) return;
    u32 pb = grid_for(n, SIG_QUADS), ab = grid_for(n, 64);
    if (pb > sig_max_waves()) pb = sig_max_waves();
    if (ab > sig_max_waves()) ab = sig_max_waves();
    if (key_tab) hipLaunchKernelGGL(k_sig_point_absorb, dim3(pb + ab), dim3(64), 0, st, pb, bytes, sigs, n, key_tab, htab, r_out, state);
    else hipLaunchKernelGGL(k_sig_point_absorb_own_table, dim3(pb + ab), dim3(64), 0, st, pb, bytes, sigs, n, pts, htab, r_out, state);
}
void launch_sig_final(hipStream_t st, const u8 *bytes, const dev_sig *sigs, u32 n, const u64 *state, const u32 *r_in, const u8 *ok, const u8 *ms_hash, u32 *tx_err) {
    if (n) hipLaunchKernelGGL(k_sig_final, dim3(grid_for(n, 64)), dim3(64), 0, st, bytes, sigs, n, state, r_in, ok, ms_hash, tx_err);
}

}  // namespace xhe
