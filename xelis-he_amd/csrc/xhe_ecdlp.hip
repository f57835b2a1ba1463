// Balance decoding: the discrete logarithm m of a decrypted point m*G for m below 2^max_bits
// (ECDLPInstance::decode, /root/reference/src/elgamal.rs:68-92; the point comes from ElGamalSecretKey::decrypt, :140-145).
// The reference hands this to the ecdlp module of its curve25519-dalek fork (pre-computed table files); that module and its
// table format are not part of the reference's sources, so what is reproduced here is the contract - the integer, or none -
// with a table format of this library's own.
//
// Baby-step/giant-step laid out for a GPU with a large HBM:
//   * every point is first multiplied by 4 (two doublings): a decoded Ristretto point is a representative up to 4-torsion,
//     4*P is the same Edwards point for all representatives, and m*G = P  <=>  m*(4G) = 4P.  All arithmetic below is on
//     plain affine Edwards points of the prime-order subgroup, base B4 = 4G.
//   * baby table: an open-addressing hash table in HBM, 2^(b+1) slots of 8 bytes, holding (32 bits of y, j) for the affine
//     y of j*B4, 0 <= j <= 2^b (+ a few more).  y(-Q) = y(Q), so one entry answers +j and -j.
//   * giant steps: m = (2i+1) 2^b + delta, -2^b <= delta < 2^b.  Q_i = 4P - (2i+1) 2^b B4 = (4P - 2^b B4) + (-i S), S = 2^(b+1) B4,
//     is looked up for i = 0 .. 2^(max_bits-b-1) - 1.  The points -i*S are a table of affine points shared by all instances.
//   * a step is one affine addition of a table point to the instance's point, of which only y is needed:
//       y3 = (y1 y2 + x1 x2) / (1 - d x1 x2 y1 y2)        (complete on this curve: the denominator is never zero)
//     The denominators of K consecutive steps are inverted together (running products stored in a scratch array, one
//     inversion per K steps, second pass backwards): 8 field products per step + 265/K for the inversion.
//   * a hit on the 32 stored bits is a candidate, confirmed on the spot by computing m*B4 (rare: a call, not inline code).
// Work is handed out as items of 128 steps of one instance, the low steps of every instance first; an instance that has been
// found is passed over from then on (k_ec_bsgs).
#include <algorithm>
#include "xhe_settings.h"
#include <memory>
#include <vector>
#include "xhe_ctx.h"
#include "xhe_dev_common.h"
#include "xhe_quad.h"

using namespace xhe;

namespace xhe {

static constexpr u32 EC_ENTRY_WORDS = 32;                  // x[10], y[10], c[10] raw limbs, 2 unused: 128 bytes
static constexpr unsigned long long EC_EMPTY = ~0ull;
static constexpr u32 EC_MAX_PROBE = 1024;                  // slots a look-up or an insertion walks at most (the table is half empty: runs of tens of slots are already rare)
static constexpr u32 EC_CHUNK = 128;                       // steps per shared inversion (and per lane's share of the scratch array)

__device__ __forceinline__ fe ec_load_fe(const u32 *p) { fe r; for (int i = 0; i < 10; i++) r.v[i] = p[i]; return r; }
__device__ __forceinline__ void ec_store_fe(u32 *p, const fe &a) { for (int i = 0; i < 10; i++) p[i] = a.v[i]; }
// an entry through 16-byte loads (entries are 128-byte aligned): where every lane reads an entry of its own, the number of
// load instructions, each touching 64 lines, is what the address unit is busy with
__device__ __forceinline__ void ec_load_c(const u32 *e, fe &c) {
    const uint4 *q = (const uint4 *)e;
    const uint4 a = q[5], b = q[6], d = q[7];
    c.v[0] = a.x; c.v[1] = a.y; c.v[2] = a.z; c.v[3] = a.w; c.v[4] = b.x; c.v[5] = b.y; c.v[6] = b.z; c.v[7] = b.w; c.v[8] = d.x; c.v[9] = d.y;
}
__device__ __forceinline__ void ec_load_entry(const u32 *e, fe &x, fe &y, fe &c) {
    const uint4 *q = (const uint4 *)e;
    const uint4 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4];
    x.v[0] = r0.x; x.v[1] = r0.y; x.v[2] = r0.z; x.v[3] = r0.w; x.v[4] = r1.x; x.v[5] = r1.y; x.v[6] = r1.z; x.v[7] = r1.w; x.v[8] = r2.x; x.v[9] = r2.y;
    y.v[0] = r2.z; y.v[1] = r2.w; y.v[2] = r3.x; y.v[3] = r3.y; y.v[4] = r3.z; y.v[5] = r3.w; y.v[6] = r4.x; y.v[7] = r4.y; y.v[8] = r4.z; y.v[9] = r4.w;
    ec_load_c(e, c);
}
// scratch element k of this lane: word w at ((k * 10 + w) * n_lanes + lane), so a wavefront's accesses are contiguous
__device__ __forceinline__ fe ec_scratch_load(const u32 *s, u64 n_lanes, u64 lane, u32 k) {
    fe r;
#pragma unroll
    for (int w = 0; w < 10; w++) r.v[w] = s[((u64)k * 10 + w) * n_lanes + lane];
    return r;
}
__device__ __forceinline__ void ec_scratch_store(u32 *s, u64 n_lanes, u64 lane, u32 k, const fe &a) {
#pragma unroll
    for (int w = 0; w < 10; w++) s[((u64)k * 10 + w) * n_lanes + lane] = a.v[w];
}

__device__ ge ec_mul_u64(const ge &base, u64 m) {
    ge r = ge_identity();
    bool started = false;
    for (int bit = 63; bit >= 0; bit--) {
        if (started) r = ge_double(r);
        if ((m >> bit) & 1) { r = started ? ge_add(r, base) : base; started = true; }
    }
    return r;
}
__device__ ge ec_mul_words(const ge &base, const u32 k[8]) {
    ge r = ge_identity();
    bool started = false;
    for (int bit = 255; bit >= 0; bit--) {
        if (started) r = ge_double(r);
        if ((k[bit >> 5] >> (bit & 31)) & 1) { r = started ? ge_add(r, base) : base; started = true; }
    }
    return r;
}

// multiples (first + t) * base for t < count as projective X, Y, Z (raw limbs) in entries of EC_ENTRY_WORDS words
__global__ void __launch_bounds__(64) k_ec_multiples(const u32 *__restrict__ base_xyzt, u64 first, u64 count, u32 per_lane, u32 *__restrict__ out) {
    const u64 lane = (u64)blockIdx.x * 64 + threadIdx.x;
    const u64 lo = lane * per_lane;
    if (lo >= count) return;
    const u64 hi = lo + per_lane < count ? lo + per_lane : count;
    const ge base = ge_load(base_xyzt);
    ge p = ec_mul_u64(base, first + lo);
    for (u64 t = lo; t < hi; t++) {
        u32 *e = out + EC_ENTRY_WORDS * t;
        ec_store_fe(e, p.X); ec_store_fe(e + 10, p.Y); ec_store_fe(e + 20, p.Z);
        p = ge_add(p, base);
    }
}
// in place: projective entries -> affine x, y and c = x y (times d when times_d), x and c negated when negate
__global__ void __launch_bounds__(64) k_ec_normalize(u32 *__restrict__ entries, u64 count, u32 per_lane, u32 *__restrict__ scratch, u64 n_lanes, u32 negate, u32 times_d) {
    const u64 lane = (u64)blockIdx.x * 64 + threadIdx.x;
    const u64 lo = lane * per_lane;
    if (lo >= count) return;
    const u32 n = (u32)((lo + per_lane < count ? lo + per_lane : count) - lo);
    fe run = fe_one();
    for (u32 k = 0; k < n; k++) {
        run = fe_mul(run, ec_load_fe(entries + EC_ENTRY_WORDS * (lo + k) + 20));
        ec_scratch_store(scratch, n_lanes, lane, k, run);
    }
    fe inv = fe_invert(run);
    const fe d = fe_c_d();
    for (u32 k = n; k-- > 0;) {
        u32 *e = entries + EC_ENTRY_WORDS * (lo + k);
        const fe X = ec_load_fe(e), Y = ec_load_fe(e + 10), Z = ec_load_fe(e + 20);
        const fe zinv = k ? fe_mul(inv, ec_scratch_load(scratch, n_lanes, lane, k - 1)) : inv;
        inv = fe_mul(inv, Z);
        fe x = fe_mul(X, zinv), y = fe_mul(Y, zinv);
        fe c = fe_mul(x, y);
        if (times_d) c = fe_mul(c, d);
        if (negate) { x = fe_negate(x); c = fe_negate(c); }
        ec_store_fe(e, x); ec_store_fe(e + 10, y); ec_store_fe(e + 20, c);
    }
}

struct ec_bsgs_args {
    const u32 *inst;           // per instance: x, y, x*y (EC_ENTRY_WORDS words)
    u64 n_inst;
    const u32 *gt;             // table points: x, y, d*x*y
    u64 steps;                 // table points per instance
    u32 K;                     // steps per item = steps per shared inversion
    u64 n_items;               // n_inst * ceil(steps / K); item = chunk * n_inst + instance: the low steps of every instance first
    u32 *scratch; u64 n_lanes; // K * 10 words per lane; n_lanes = the whole grid
    unsigned long long *slots; u32 slot_bits;
    unsigned long long *next_item;     // the shared item counter of this launch (zero before it)
    // look-up only
    unsigned long long *amounts;       // per instance: the amount once it is confirmed, EC_EMPTY until then
    const u32 *four_p, *b4_xyzt; u32 b, max_bits;
    unsigned long long *steps_done;
    u32 *cand; u32 cand_cap; u32 *cand_count;      // MODE 1: candidates (instance, step, j) for k_ec_confirm
};

__device__ __forceinline__ void ec_key(const fe &y, u64 &slot, u32 &fp, u32 slot_bits) {
    u32 w[8]; fe_pack(w, y);
    const u64 h = ((u64)w[1] << 32 | w[0]) * 0x9e3779b97f4a7c15ull;
    slot = h >> (64 - slot_bits);
    fp = w[2] ^ (w[3] * 0x85ebca6bu);
}

// A hit on the table's 32 check bits at giant step `step` with baby step j stands for m = (2 step + 1) 2^b +- j: the one with
// m * B4 = 4 P is the amount.  Rare (one true hit per instance, a false one per 2^32 look-ups), so it is a call, not inline code.
__device__ __noinline__ void ec_confirm(const u32 *four_p, const u32 *b4_xyzt, u32 b, u32 max_bits, unsigned long long *amounts, u32 instance, u64 step, u32 j) {
    const u64 centre = (2 * step + 1) << b;
    const ge want = ge_load(four_p + 32ull * instance), base = ge_load(b4_xyzt);
    for (int sign = 0; sign < 2; sign++) {
        if (sign && j > centre) continue;
        const u64 m = sign ? centre - j : centre + j;
        if (max_bits < 64 && (m >> max_bits)) continue;
        const ge got = ec_mul_u64(base, m);
        // both are points of the prime-order subgroup: plain projective equality
        if (fe_eq(fe_mul(got.X, want.Z), fe_mul(want.X, got.Z)) && fe_eq(fe_mul(got.Y, want.Z), fe_mul(want.Y, got.Z))) atomicMin(&amounts[instance], (unsigned long long)m);
    }
}

// MODE 0: insert (fp, j) with j = instance * steps + step.  MODE 1: look up, candidates into a list that k_ec_confirm goes
// through afterwards (when every item is in flight at once anyway: a few instances, whose latency a confirmation inside a
// wavefront would double).  MODE 2: look up and confirm on the spot, so that found instances drop out while the walk runs.
// The grid is what the chip holds at once.  A lane keeps its place in the scratch array and takes one item (K steps of one
// instance) after the other from a shared counter: the lanes of a wavefront that need an item ask for as many as they are in one
// atomic, and a lane whose new item belongs to an instance that has been found meanwhile asks again - so every lane of every
// wavefront works on an unfound instance until the items run out, and an instance costs about the steps up to its amount.
// An item goes in three passes: forward (running products of the denominators into the scratch array), backward (the y of every
// sum; its table slot and check bits replace the running product in the scratch array), then the table: eight independent
// accesses in flight per lane, which is what hides the latency of a random read in a table of gigabytes.
template <int MODE>
__device__ __forceinline__ void ec_bsgs_body(const ec_bsgs_args &a) {
    const u64 lane = (u64)blockIdx.x * 256 + threadIdx.x;
    const u32 in_wave = threadIdx.x & 63;
    const u64 mask = (1ull << a.slot_bits) - 1;
    u64 done = 0;
    bool exhausted = false;
    for (;;) {
        // hand out items until every lane has one of an unfound instance, or the items are gone
        bool have = false;
        u64 instance = 0, chunk = 0;
        for (;;) {
            const unsigned long long need = __ballot(!have && !exhausted);
            if (!need) break;
            unsigned long long first = 0;
            if (in_wave == 0) first = atomicAdd(a.next_item, (unsigned long long)__popcll(need));
            first = __shfl(first, 0);
            if (!have && !exhausted) {
                const u64 item = first + __popcll(need & ((1ull << in_wave) - 1));
                if (item >= a.n_items) exhausted = true;
                else {
                    chunk = item / a.n_inst; instance = item - chunk * a.n_inst;
                    have = MODE != 2 || __atomic_load_n(&a.amounts[instance], __ATOMIC_RELAXED) == EC_EMPTY;
                }
            }
        }
        if (!__ballot(have)) break;
        const u64 base_step = chunk * a.K;
        const u32 n = have ? (u32)(a.steps - base_step < a.K ? a.steps - base_step : a.K) : 0u;
        const u32 *pi = a.inst + EC_ENTRY_WORDS * instance;
        {
            const fe cP = ec_load_fe(pi + 20);
            fe run = fe_one();
            for (u32 k = 0; k < n; k++) {
                fe ci; ec_load_c(a.gt + EC_ENTRY_WORDS * (base_step + k), ci);
                run = fe_mul(run, fe_sub(fe_one(), fe_mul(cP, ci)));
                ec_scratch_store(a.scratch, a.n_lanes, lane, k, run);
            }
            fe inv = fe_invert(run);
            const fe xP = ec_load_fe(pi), yP = ec_load_fe(pi + 10);
            for (u32 k = n; k-- > 0;) {
                fe xi, yi, ci; ec_load_entry(a.gt + EC_ENTRY_WORDS * (base_step + k), xi, yi, ci);
                const auto den = fe_sub(fe_one(), fe_mul(cP, ci));
                const fe dinv = k ? fe_mul(inv, ec_scratch_load(a.scratch, a.n_lanes, lane, k - 1)) : inv;
                inv = fe_mul(inv, den);
                const fe y3 = fe_mul(fe_add(fe_mul(yP, yi), fe_mul(xP, xi)), dinv);
                u64 slot; u32 fp;
                ec_key(y3, slot, fp, a.slot_bits);
                u32 *sk = a.scratch + ((u64)k * 10) * a.n_lanes + lane;
                sk[0] = (u32)slot; sk[a.n_lanes] = (u32)(slot >> 32); sk[2 * a.n_lanes] = fp;
            }
        }
        for (u32 k0 = 0; k0 < n; k0 += 8) {
            u64 slot[8]; u32 fp[8]; unsigned long long v[8];
#pragma unroll
            for (u32 q = 0; q < 8; q++) {
                const u32 k = k0 + q < n ? k0 + q : n - 1;
                const u32 *sk = a.scratch + ((u64)k * 10) * a.n_lanes + lane;
                slot[q] = (u64)sk[0] | ((u64)sk[a.n_lanes] << 32); fp[q] = sk[2 * a.n_lanes];
            }
            if (MODE != 0) {
#pragma unroll
                for (u32 q = 0; q < 8; q++) v[q] = a.slots[slot[q] & mask];
            }
#pragma unroll
            for (u32 q = 0; q < 8; q++) {
                if (k0 + q >= n) continue;
                const u64 step = base_step + k0 + q;
                if (MODE == 0) {
                    const unsigned long long mine = ((unsigned long long)fp[q] << 32) | (unsigned long long)(u32)(instance * a.steps + step);
                    for (u32 p = 0; p < EC_MAX_PROBE; p++) {
                        const unsigned long long old = atomicCAS(&a.slots[(slot[q] + p) & mask], EC_EMPTY, mine);
                        if (old == EC_EMPTY || old == mine) break;
                    }
                } else {
                    unsigned long long cur = v[q];
                    for (u32 p = 0; p < EC_MAX_PROBE && cur != EC_EMPTY; p++) {
                        if ((u32)(cur >> 32) == fp[q]) {
                            if (MODE == 2) ec_confirm(a.four_p, a.b4_xyzt, a.b, a.max_bits, a.amounts, (u32)instance, step, (u32)cur);
                            else {
                                const u32 at = atomicAdd(a.cand_count, 1u);
                                if (at < a.cand_cap) { u32 *c = a.cand + 4ull * at; c[0] = (u32)instance; c[1] = (u32)step; c[2] = (u32)(step >> 32); c[3] = (u32)cur; }
                            }
                        }
                        cur = a.slots[(slot[q] + p + 1) & mask];
                    }
                }
            }
        }
        done += n;
    }
    if (MODE != 0 && a.steps_done) {
#pragma unroll
        for (u32 d = 32; d >= 1; d >>= 1) done += __shfl_down((unsigned long long)done, d);
        if (in_wave == 0 && done) atomicAdd(a.steps_done, (unsigned long long)done);
    }
}

// the same walk at three or four resident wavefronts per SIMD (168 or 128 registers)
template <int MODE> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_ec_bsgs(ec_bsgs_args a) { ec_bsgs_body<MODE>(a); }
template <int MODE> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_ec_bsgs_occ3(ec_bsgs_args a) { ec_bsgs_body<MODE>(a); }

// the candidates of a MODE 1 walk, both signs of each on a thread of its own
__global__ void __launch_bounds__(64) k_ec_confirm(const u32 *__restrict__ cand, const u32 *__restrict__ cand_count, u32 cand_cap, const u32 *__restrict__ four_p,
                                                   const u32 *__restrict__ b4_xyzt, u32 b, u32 max_bits, unsigned long long *__restrict__ amounts) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    const u32 n = cand_count[0] < cand_cap ? cand_count[0] : cand_cap;
    if (t >= 2 * n) return;
    const u32 *c = cand + 4ull * (t >> 1);
    const u64 step = (u64)c[1] | ((u64)c[2] << 32), j = c[3];
    const u64 centre = (2 * step + 1) << b;
    if ((t & 1) && j > centre) return;
    const u64 m = (t & 1) ? centre - j : centre + j;
    if (max_bits < 64 && (m >> max_bits)) return;
    const ge want = ge_load(four_p + 32ull * c[0]);
    const ge got = ec_mul_u64(ge_load(b4_xyzt), m);
    if (fe_eq(fe_mul(got.X, want.Z), fe_mul(want.X, got.Z)) && fe_eq(fe_mul(got.Y, want.Z), fe_mul(want.Y, got.Z))) atomicMin(&amounts[c[0]], (unsigned long long)m);
}

// input -> 4 * point (packed XYZT, for the confirmation) and the affine start 4 * point + offset (x, y, x*y)
// mode 0: in = 32-byte Ristretto encodings.  mode 1: in = 64-byte ciphertexts (commitment, handle); point = C - s * D.
// A quad of lanes per point (xhe_quad.h): the decryption is one chain of 255 doublings per point, and a lone wavefront issues a
// doubling's eight field products one after the other - on four lanes the chain is a third as long (1.08 -> 0.4 ms for 4 096
// ciphertexts, which is all of the chip's SIMDs with one wavefront or none either way).  Lanes 0 and 1 of the quad decode C and D
// side by side.  The secret is the same for every point, so the chain's control flow is uniform.  Every lane of a quad stays
// active to the end (DPP reads need all four); quads beyond n work on the last point and store nothing.
__global__ void __launch_bounds__(256) k_ec_prepare(const u8 *__restrict__ in, u32 n, u32 mode, const u32 *__restrict__ secret_words, const u32 *__restrict__ offset_xyzt,
                                                    u32 *__restrict__ four_p, u32 *__restrict__ inst, u8 *__restrict__ status) {
    const u32 q = (blockIdx.x * 256 + threadIdx.x) >> 2, role = qd_role();
    const bool live = q < n;
    const u32 i = live ? q : n - 1;
    ge mine;
    const bool ok_mine = ristretto_decode(mine, mode == 0 ? in + 32ull * i : in + 64ull * i + 32 * (role & 1));
    const u32 okw = ok_mine ? 1u : 0u;
    const bool ok = qd_pull<XHE_QP(0, 0, 0, 0)>(okw) && qd_pull<XHE_QP(1, 1, 1, 1)>(okw);
    // C from lane 0 and D from lane 1, whole, in every lane of the quad
    ge c, d;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        c.X.v[k] = qd_pull<XHE_QP(0, 0, 0, 0)>(mine.X.v[k]); c.Y.v[k] = qd_pull<XHE_QP(0, 0, 0, 0)>(mine.Y.v[k]);
        c.Z.v[k] = qd_pull<XHE_QP(0, 0, 0, 0)>(mine.Z.v[k]); c.T.v[k] = qd_pull<XHE_QP(0, 0, 0, 0)>(mine.T.v[k]);
        d.X.v[k] = qd_pull<XHE_QP(1, 1, 1, 1)>(mine.X.v[k]); d.Y.v[k] = qd_pull<XHE_QP(1, 1, 1, 1)>(mine.Y.v[k]);
        d.Z.v[k] = qd_pull<XHE_QP(1, 1, 1, 1)>(mine.Z.v[k]); d.T.v[k] = qd_pull<XHE_QP(1, 1, 1, 1)>(mine.T.v[k]);
    }
    qd_point p = qd_from_ge(c, role);
    if (mode != 0) {
        u32 s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = secret_words[k];
        const qd_point dq = qd_from_ge(d, role);
        qd_point acc = qd_identity(role);
        for (int bit = 255; bit >= 0; bit--) {
            acc = qd_double(acc, role);
            if ((s[bit >> 5] >> (bit & 31)) & 1) acc = qd_add(acc, dq, role);
        }
        // C - s D: the negative of (X, Y, Z, T) is (-X, Y, Z, -T)
        const fe neg = fe_negate(acc.c);
        qd_point minus;
#pragma unroll
        for (int k = 0; k < 10; k++) minus.c.v[k] = (role == 0 || role == 3) ? neg.v[k] : acc.c.v[k];
        p = qd_add(p, minus, role);
    }
    if (!ok) p = qd_identity(role);
    p = qd_double(qd_double(p, role), role);
    if (live) qd_store(four_p + 32ull * q, p, role);
    const ge sum = qd_gather(qd_add(p, qd_load(offset_xyzt, role), role));
    const fe zinv = fe_invert(sum.Z);
    const fe x = fe_mul(sum.X, zinv), y = fe_mul(sum.Y, zinv);
    if (live && role == 0) {
        status[q] = ok ? 0 : 1;
        u32 *e = inst + EC_ENTRY_WORDS * (u64)q;
        ec_store_fe(e, x); ec_store_fe(e + 10, y); ec_store_fe(e + 20, fe_mul(x, y));
    }
}

__global__ void k_ec_fill64(unsigned long long *p, u64 n, unsigned long long v) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_ec_fill32(u32 *p, u32 n, u32 v) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace xhe

struct xhe_ecdlp_table {
    xhe_ctx *ctx = nullptr;
    u32 b = 0, slot_bits = 0;
    ge b4;                                   // 4 G (host copy)
    dev_buf slots, b4_dev, offset_dev;       // the table; 4 G; -(2^b) 4 G (what an instance's walk starts from, beside 4 P)
    dev_buf gt; u32 gt_max_bits = 0; u64 gt_steps = 0;         // the giant-step points of the last max_bits asked for
    dev_buf ws;                                                // one block for a call's temporaries, kept between calls (grown when a call needs more)
    std::vector<unsigned long long> host_amounts; std::vector<u8> host_status;
    xhe_ecdlp_stats last{0, 0, 0.0, 0};
    hipEvent_t ev[2] = {nullptr, nullptr};
};

namespace {

// the Ristretto basepoint G (RISTRETTO_BASEPOINT_COMPRESSED of curve25519-dalek, used at /root/reference/src/elgamal.rs:5)
ge host_basepoint() {
    const u8 bb[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                       0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge g; ristretto_decode(g, bb);
    return g;
}
// lanes of a grid that is resident at once: waves_per_simd wavefronts on every SIMD
u64 ec_grid_lanes(xhe_ctx *ctx, u32 waves_per_simd) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    return (u64)cus * 4 * waves_per_simd * 64;
}
ge host_mul_u64(const ge &base, u64 m) {
    ge r = ge_identity();
    for (int bit = 63; bit >= 0; bit--) { r = ge_double(r); if ((m >> bit) & 1) r = ge_add(r, base); }
    return r;
}

// count multiples of `base` from 0, normalised in place: entries (x, y, c) on the device
int32_t make_affine_multiples(xhe_ctx *ctx, hipStream_t st, const ge &base, u64 count, bool negate, bool times_d, dev_buf &entries) {
    dev_buf base_dev, scratch;
    struct release { dev_buf &a, &b; ~release() { a.release(); b.release(); } } guard{base_dev, scratch};
    HIPCHK(ctx, entries.reserve(count * EC_ENTRY_WORDS * 4));
    HIPCHK(ctx, base_dev.reserve(128));
    u32 packed[32]; ge_store(packed, base);
    HIPCHK(ctx, hipMemcpyAsync(base_dev.p, packed, 128, hipMemcpyHostToDevice, st));
    const u32 per_lane = count >= (1u << 16) ? 64 : 16;
    const u64 lanes = (count + per_lane - 1) / per_lane, padded = (lanes + 63) / 64 * 64;
    HIPCHK(ctx, scratch.reserve(padded * per_lane * 40));
    hipLaunchKernelGGL(k_ec_multiples, dim3((unsigned)(padded / 64)), dim3(64), 0, st, base_dev.as<u32>(), (u64)0, count, per_lane, entries.as<u32>());
    hipLaunchKernelGGL(k_ec_normalize, dim3((unsigned)(padded / 64)), dim3(64), 0, st, entries.as<u32>(), count, per_lane, scratch.as<u32>(), padded, negate ? 1u : 0u, times_d ? 1u : 0u);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));          // base_dev and scratch go back to the driver on return
    return XHE_OK;
}

}  // namespace

extern "C" {

int32_t xhe_ecdlp_table_create(xhe_ctx *ctx, uint32_t table_bits, xhe_ecdlp_table **out) {
    if (!ctx || !out) return XHE_E_ARG;
    if (table_bits < 8 || table_bits > 30) { set_error(ctx, "xhe_ecdlp_table_create: table_bits must be 8..30"); return XHE_E_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::unique_ptr<xhe_ecdlp_table> t(new xhe_ecdlp_table);
    t->ctx = ctx; t->b = table_bits; t->slot_bits = table_bits + 1;
    t->b4 = ge_double(ge_double(host_basepoint()));
    const u64 n_slots = 1ull << t->slot_bits;
    struct release_on_error { xhe_ecdlp_table *t; bool armed = true; ~release_on_error() { if (armed) { t->slots.release(); t->b4_dev.release(); t->offset_dev.release(); } } } guard{t.get()};
    HIPCHK(ctx, t->slots.reserve(n_slots * 8));
    HIPCHK(ctx, t->b4_dev.reserve(128));
    u32 packed[32]; ge_store(packed, t->b4);
    HIPCHK(ctx, hipMemcpyAsync(t->b4_dev.p, packed, 128, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, t->offset_dev.reserve(128));
    u32 packed_offset[32]; ge_store(packed_offset, ge_neg(host_mul_u64(t->b4, 1ull << table_bits)));
    HIPCHK(ctx, hipMemcpyAsync(t->offset_dev.p, packed_offset, 128, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ec_fill64, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, t->slots.as<unsigned long long>(), n_slots, EC_EMPTY);
    // j = chunk * per + k: the chunk starts (chunk * per) B4 play the instances, the points k B4 the table
    const u32 per = 1u << std::min<u32>(table_bits, 10);
    const u64 n_chunks = (1ull << table_bits) / per + 1;            // one chunk more: j = 2^b is needed, the rest of it is harmless
    dev_buf small, starts, scratch;
    struct release { dev_buf &a, &b, &c; ~release() { a.release(); b.release(); c.release(); } } temporaries{small, starts, scratch};
    if (int32_t rc = make_affine_multiples(ctx, st, t->b4, per, false, true, small)) return rc;
    if (int32_t rc = make_affine_multiples(ctx, st, host_mul_u64(t->b4, per), n_chunks, false, false, starts)) return rc;
    ec_bsgs_args a{};
    a.inst = starts.as<u32>(); a.n_inst = n_chunks; a.gt = small.as<u32>(); a.steps = per;
    a.K = std::min<u32>(per, 64);
    a.n_items = n_chunks * ((per + a.K - 1) / a.K);
    a.n_lanes = std::min<u64>((a.n_items + 255) / 256 * 256, ec_grid_lanes(ctx, 4));
    HIPCHK(ctx, scratch.reserve(a.n_lanes * a.K * 40));
    a.scratch = scratch.as<u32>(); a.slots = t->slots.as<unsigned long long>(); a.slot_bits = t->slot_bits;
    dev_buf counter;
    struct release_counter { dev_buf &c; ~release_counter() { c.release(); } } counter_guard{counter};
    HIPCHK(ctx, counter.reserve(64));
    hipLaunchKernelGGL(k_ec_fill64, dim3(1), dim3(64), 0, st, counter.as<unsigned long long>(), (u64)1, 0ull);
    a.next_item = counter.as<unsigned long long>();
    hipLaunchKernelGGL(k_ec_bsgs<0>, dim3((unsigned)(a.n_lanes / 256)), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipEventCreate(&t->ev[0])); HIPCHK(ctx, hipEventCreate(&t->ev[1]));
    guard.armed = false;
    *out = t.release();
    return XHE_OK;
}

void xhe_ecdlp_table_free(xhe_ecdlp_table *t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    t->slots.release(); t->b4_dev.release(); t->offset_dev.release(); t->gt.release(); t->ws.release();
    for (hipEvent_t e : t->ev) if (e) (void)hipEventDestroy(e);
    delete t;
}

static int32_t ecdlp_run(xhe_ecdlp_table *t, const uint8_t *in, uint64_t n, uint32_t mode, const uint8_t *secret32, uint32_t max_bits, int64_t *amounts, uint8_t *status_out) {
    if (!t || !t->ctx) return XHE_E_ARG;
    xhe_ctx *ctx = t->ctx;
    if ((!in && n) || (!amounts && n) || (mode == 1 && !secret32)) return XHE_E_ARG;
    if (max_bits < 1 || max_bits > 62) { set_error(ctx, "xhe_ecdlp_decode: max_bits must be 1..62"); return XHE_E_ARG; }
    if (n > (1ull << 24)) { set_error(ctx, "xhe_ecdlp_decode: at most 2^24 points per call"); return XHE_E_ARG; }
    const u32 b = t->b;
    const u64 steps = max_bits > b + 1 ? 1ull << (max_bits - b - 1) : 1;
    if (steps > (1ull << 24)) { set_error(ctx, "xhe_ecdlp_decode: max_bits - table_bits - 1 must not exceed 24 (use a larger table)"); return XHE_E_ARG; }
    if (!n) return XHE_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (t->gt_max_bits != max_bits) {
        // the points -i S, S = 2^(b+1) B4
        ge S = t->b4;
        for (u32 k = 0; k < b + 1; k++) S = ge_double(S);
        if (int32_t rc = make_affine_multiples(ctx, st, S, steps, true, true, t->gt)) return rc;
        t->gt_max_bits = max_bits; t->gt_steps = steps;
    }
    // the call's temporaries are 256-byte aligned ranges of the table's workspace block: no allocation from the second call on
    const u64 in_bytes = (mode ? 64ull : 32ull) * n;
    ec_bsgs_args a{};
    const char *occ_env = setting("XHE_ECDLP_OCC");
    const bool occ3 = !(occ_env && occ_env[0] == '4');
    const u64 grid_lanes = ec_grid_lanes(ctx, occ3 ? 3 : 4);
    // steps per item (= per shared inversion): 128 when that still makes at least two items per resident lane, else shorter items
    // down to 32 - a short walk (few instances, or a large table) is bound by the length of a lane's chain, not by its arithmetic
    a.K = (u32)std::min<u64>(EC_CHUNK, steps);
    while (a.K > 32 && n * ((steps + a.K - 1) / a.K) < 2 * grid_lanes) a.K >>= 1;
    a.n_items = n * ((steps + a.K - 1) / a.K);
    a.n_lanes = std::min<u64>((a.n_items + 255) / 256 * 256, grid_lanes);
    // (nearly) every item in flight at once: little can drop out early, candidates go through a list and a kernel of their own
    // instead of stalling their wavefronts with a confirmation each
    const bool listed = a.n_items <= 4 * a.n_lanes;
    const u32 cand_cap = (u32)(8 * n + 4096);
    dev_buf in_dev, secret_dev, four_p, inst, status, amt, scratch, steps_done, cand;
    {
        struct want { dev_buf *buf; u64 bytes; };
        const want wants[] = {{&in_dev, in_bytes}, {&secret_dev, 64}, {&four_p, 128 * n}, {&inst, EC_ENTRY_WORDS * 4ull * n}, {&status, n}, {&amt, 8 * n},
                              {&scratch, a.n_lanes * a.K * 40ull}, {&steps_done, 64}, {&cand, listed ? 16ull * cand_cap : 0}};
        u64 total = 0;
        for (const want &w : wants) total += (std::max<u64>(w.bytes, 64) + 255) & ~255ull;
        HIPCHK(ctx, t->ws.reserve(total));
        u64 off = 0;
        for (const want &w : wants) { const u64 sz = (std::max<u64>(w.bytes, 64) + 255) & ~255ull; w.buf->point_at(t->ws.p, off, sz); off += sz; }
    }
    HIPCHK(ctx, hipMemcpyAsync(in_dev.p, in, in_bytes, hipMemcpyHostToDevice, st));
    u32 secret_words[8] = {0};
    if (mode == 1) { for (int k = 0; k < 8; k++) secret_words[k] = (u32)secret32[4 * k] | ((u32)secret32[4 * k + 1] << 8) | ((u32)secret32[4 * k + 2] << 16) | ((u32)secret32[4 * k + 3] << 24); }
    HIPCHK(ctx, hipMemcpyAsync(secret_dev.p, secret_words, 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ec_fill64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, amt.as<unsigned long long>(), n, EC_EMPTY);
    hipLaunchKernelGGL(k_ec_fill64, dim3(1), dim3(64), 0, st, steps_done.as<unsigned long long>(), (u64)3, 0ull);     // steps walked, item counter, candidate counter
    hipLaunchKernelGGL(k_ec_prepare, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, st, in_dev.as<u8>(), (u32)n, mode, secret_dev.as<u32>(), t->offset_dev.as<u32>(), four_p.as<u32>(), inst.as<u32>(),
                       status.as<u8>());
    a.inst = inst.as<u32>(); a.n_inst = n; a.gt = t->gt.as<u32>(); a.steps = steps;
    a.scratch = scratch.as<u32>(); a.slots = t->slots.as<unsigned long long>(); a.slot_bits = t->slot_bits;
    a.amounts = amt.as<unsigned long long>(); a.four_p = four_p.as<u32>(); a.b4_xyzt = t->b4_dev.as<u32>(); a.b = b; a.max_bits = max_bits;
    a.steps_done = steps_done.as<unsigned long long>(); a.next_item = steps_done.as<unsigned long long>() + 1;
    if (listed) {
        a.cand = cand.as<u32>(); a.cand_cap = cand_cap; a.cand_count = (u32 *)(steps_done.as<unsigned long long>() + 2);
    }
    HIPCHK(ctx, hipEventRecord(t->ev[0], st));
    if (listed) {
        if (occ3) hipLaunchKernelGGL(k_ec_bsgs_occ3<1>, dim3((unsigned)(a.n_lanes / 256)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_ec_bsgs<1>, dim3((unsigned)(a.n_lanes / 256)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_ec_confirm, dim3((2 * cand_cap + 63) / 64), dim3(64), 0, st, cand.as<u32>(), a.cand_count, cand_cap, four_p.as<u32>(), t->b4_dev.as<u32>(), b, max_bits,
                           amt.as<unsigned long long>());
    } else {
        if (occ3) hipLaunchKernelGGL(k_ec_bsgs_occ3<2>, dim3((unsigned)(a.n_lanes / 256)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_ec_bsgs<2>, dim3((unsigned)(a.n_lanes / 256)), dim3(256), 0, st, a);
    }
    HIPCHK(ctx, hipEventRecord(t->ev[1], st));
    HIPCHK(ctx, hipGetLastError());
    std::vector<unsigned long long> &host = t->host_amounts; std::vector<u8> &host_status = t->host_status;
    if (host.size() < n) { host.resize(n); host_status.resize(n); }
    unsigned long long done = 0;
    HIPCHK(ctx, hipMemcpyAsync(host.data(), amt.p, 8 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(host_status.data(), status.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(&done, steps_done.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    float ms = 0; (void)hipEventElapsedTime(&ms, t->ev[0], t->ev[1]);
    t->last = {n, (uint64_t)done, (double)ms, (uint64_t)a.n_lanes};
    for (u64 i = 0; i < n; i++) {
        amounts[i] = (host[i] == EC_EMPTY || host_status[i]) ? -1 : (int64_t)host[i];
        if (status_out) status_out[i] = host_status[i];
    }
    return XHE_OK;
}

int32_t xhe_ecdlp_decode(xhe_ecdlp_table *t, const uint8_t *points32, uint64_t n, uint32_t max_bits, int64_t *amounts, uint8_t *status) {
    return ecdlp_run(t, points32, n, 0, nullptr, max_bits, amounts, status);
}
int32_t xhe_elgamal_decrypt_decode(xhe_ecdlp_table *t, const uint8_t secret32[32], const uint8_t *cts64, uint64_t n, uint32_t max_bits, int64_t *amounts, uint8_t *status) {
    return ecdlp_run(t, cts64, n, 1, secret32, max_bits, amounts, status);
}
int32_t xhe_ecdlp_last_stats(const xhe_ecdlp_table *t, xhe_ecdlp_stats *out) {
    if (!t || !out) return XHE_E_ARG;
    *out = t->last;
    return XHE_OK;
}

}  // extern "C"
