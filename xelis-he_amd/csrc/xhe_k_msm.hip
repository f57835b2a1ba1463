// Pippenger bucket multiscalar multiplication for gfx950: the kernels behind
// RistrettoPoint::vartime_multiscalar_mul (/root/reference/src/proofs.rs:49-67) and the range-proof
// mega-check (/root/reference/src/tx/verify.rs:504-513).
//
// Pipeline (every stage one launch over the whole problem):
//   digits            thread per term: signed radix-2^c digits, stored window-major as 16-bit sign|magnitude
//   tile histogram    workgroup per (window, tile of 8192..32768 terms): bucket counts in LDS, no global atomics
//   tile prefix       thread per (window, bucket): running sum over the tiles, totals per bucket
//   scan              exclusive prefix sum of the W*B totals
//   scatter           workgroup per (window, tile): positions from LDS counters seeded with
//                     offset[bucket] + prefix[tile][bucket]; term index (+sign bit) into its bucket's segment
//                     (from 200 000 terms on: the same three steps over coarse bins, then a sort inside every bin)
//   accumulate        thread per chunk of M consecutive entries of the bucket-sorted array: mixed additions of
//                     the entries' affine-Niels points; buckets that lie inside the chunk are finished there,
//                     the (at most two) buckets cut by the chunk's borders leave partial sums
//   merge             thread per bucket: adds the partial sums of a bucket cut by chunk borders; buckets
//                     spanning many chunks (the top window holds only the digit carry, so half of all terms
//                     meet in one bucket) go to a workgroup-per-bucket reduction instead
//   reduce            wave per (window, 64*L buckets): running sums per lane, then lane-weighted sums
//                     with wavefront shuffles
//   window combine    wave per window over the segment partials (same shuffle scheme)
//   horner            the 252 doublings over the window sums on one quad of lanes, compress, identity test
// Points come in as 128-byte affine-Niels records (xhe_ge.h), scalars as canonical 32-byte values.
// The work per thread of the accumulate stage is M additions whatever the distribution of the digits.
#include <stdlib.h>
#include "xhe_settings.h"
#include "xhe_dev_common.h"
#include "xhe_kernels.h"
#include "xhe_wide.h"

namespace xhe {

// window width by term count: balances n*W bucket additions against 2*W*B reduction additions and
// keeps W*B large enough to occupy the 256 CUs
msm_plan msm_make_plan(u64 n) {
    u32 c;                                           // widths dividing 252 leave no narrow top window
    if (n < 256) c = 7;
    else if (n < 4096) c = 9;
    else if (n < 200000) c = 12;
    else if (n < 1500000) c = 14;
    else c = 15;                                     // a digit is kept in 15 bits + sign
    msm_plan p;
    p.c = c; p.W = (252 + c - 1) / c; p.B = 1u << (c - 1);          // recoded scalars are below 2^252
    u32 L = p.B / 2048; if (L < 1) L = 1;          // S = B/(64L) <= 32: W*S wavefronts, at most one per SIMD
    if (p.B < 64) L = 1;
    p.L = L; p.S = (p.B + 64 * L - 1) / (64 * L);
    // entries per accumulate chunk: about half a bucket's expected load (few buckets are cut into more than a handful of
    // chunks, so the merge stays cheap), within 8..128 and small enough that the chunks still cover the 1024 SIMDs several times
    u64 entries = n * p.W, per_bucket = n / p.B;
    u32 M = 8;
    while (M < 128 && 2ull * M <= per_bucket && entries / (2ull * M) >= 8192) M *= 2;
    if (M < 16 && entries >= (1u << 16)) M = 16;
    if (M < 32 && entries >= (1u << 20)) M = 32;
    // Every thread takes one chunk, and a chunk is ~M * 2000 instructions: when the chunks need a little more than a whole
    // number of rounds of the chip's thread slots (256 CUs x 3 workgroups of 256 at this kernel's register count), the few
    // workgroups of the last round run alone at a lone wave's rate - a third of the kernel's time at the bench size.  The
    // kernel's time goes as rounds * M, so among M/2..2M the chunk length with the smallest product is taken, counting the
    // rounds with a few per cent of the slots to spare; of equal products the shorter chunk wins (more, shorter rounds: inside
    // a verification the signature kernel holds part of the slots and the rounds are not the stand-alone ones; measured on the
    // 10 000-transaction batch: M = 32 -> 0.48-0.53 ms, 34 (one round) -> 0.51 ms, 17 (two rounds) -> 0.43-0.45 ms, 12 -> 0.42 ms
    // with the merge up from 0.035 to 0.066 ms).  XHE_MSM_CHUNK overrides the choice.
    {
        const u64 slots = 256ull * 3 * 256, usable = slots - slots / 32;
        u64 best_cost = ~0ull; u32 best = M;
        for (u32 m = (M / 2 > 8 ? M / 2 : 8); m <= 2 * M; m++) {
            u64 chunks = entries / m + 1, rounds = (chunks + usable - 1) / usable;
            if (rounds > 8) continue;
            if (rounds * m < best_cost) { best_cost = rounds * m; best = m; }
        }
        p.M = best;
        if (const char *e = setting("XHE_MSM_CHUNK")) { u32 v = (u32)strtoul(e, nullptr, 10); if (v >= 1 && v <= 1024) p.M = v; }
    }
    // terms per workgroup of the counting sort: every workgroup reads and writes all B counters of its window, so a tile
    // holds at least two terms per bucket once there are enough tiles (tiles * W workgroups) to cover the chip
    p.tile = 8192;
    while (p.tile < 2 * p.B && n / (2ull * p.tile) * p.W >= 1024) p.tile *= 2;
    // Large sorts go in two passes (term index in 24 bits + 7 bits of the bucket + sign in one word): scattering straight into
    // 2^13..2^14 buckets per window writes every 4-byte entry to a line of its own, which is what bounds the one-pass form.
    p.fine_bits = (c >= 14 && n <= (1u << 24) && !setting("XHE_MSM_ONEPASS")) ? 7 : 0;
    return p;
}

static constexpr u32 SCAN_TILE = 1024;       // counters per workgroup of the bucket-offset scan
static constexpr u32 MERGE_SPAN_MAX = 8;     // a bucket cut into more chunks than this is reduced by a workgroup
static constexpr u32 SUM_WORDS = 40;         // words of a bucket or chunk sum in memory (sum_store / sum_load below)

static inline u64 msm_sort_tiles(u64 n, const msm_plan &p) { return n ? (n + p.tile - 1) / p.tile : 1; }
static inline u64 msm_chunks(u64 n, const msm_plan &p) { u64 e = n * p.W; return (e + p.M - 1) / p.M + 1; }

msm_ws_bytes msm_workspace_bytes(u64 n, const msm_plan &p) {
    u64 nb = (u64)p.W * p.B, entries = n * p.W, chunks = msm_chunks(n, p);
    msm_ws_bytes b;
    b.hist = nb * 4; b.offsets = (nb + 1) * 4; b.tile_hist = nb * msm_sort_tiles(n, p) * 4; b.sorted = (entries ? entries : 1) * 4; b.buckets = nb * (SUM_WORDS * 4ull);
    b.digits = (entries ? entries : 1) * 2;
    b.chunk_sums = chunks * 2 * (SUM_WORDS * 4ull); b.big_list = (chunks / MERGE_SPAN_MAX + 16) * 4;
    b.segs = (u64)p.W * p.S * 2 * 128 + (nb / SCAN_TILE + 1) * 4; b.wins = (u64)p.W * 128;          // segs doubles as the scan's tile sums
    b.tmp = p.fine_bits ? b.sorted : 64; b.coarse = ((u64)p.W * (p.B >> p.fine_bits) + 1) * 4 + 64;
    return b;
}

// thread per term: signed radix-2^c digits of a canonical scalar (eight words), each in
// [-2^(c-1), 2^(c-1)], serial carry from the least significant window.  A scalar above (l-1)/2 is
// replaced by l - s with every digit's sign flipped: the recoded value is below 2^251 + 2^124, so
// ceil(252/c) windows hold it with no carry out of the top one, and the digits of the top window are as
// evenly spread as those of the others (without this, half of all terms would share one carry bucket).
template <typename F>
__device__ __forceinline__ void msm_for_each_digit(const u32 s_in[8], u32 c, u32 W, F f) {
    const u32 LW[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};          // l
    const u32 HW[8] = {0x2e7ae9f6u, 0x2c09318du, 0x517bce6bu, 0x0a6f7cefu, 0u, 0u, 0u, 0x08000000u};          // (l-1)/2
    bool flip = false, decided = false;
#pragma unroll
    for (int k = 7; k >= 0; k--)
        if (!decided && s_in[k] != HW[k]) { flip = s_in[k] > HW[k]; decided = true; }
    u32 s[8];
    {
        u64 borrow = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            u64 dlt = (u64)LW[k] - s_in[k] - borrow;
            s[k] = flip ? (u32)dlt : s_in[k];
            borrow = (dlt >> 32) & 1;
        }
    }
    u32 carry = 0;
    const u32 half = 1u << (c - 1), mask = (1u << c) - 1;
    for (u32 w = 0; w < W; w++) {
        u32 bit = w * c, word = bit >> 5, off = bit & 31;
        u32 raw = 0;
        if (word < 8) {
            u64 two = (u64)s[word] | ((word + 1 < 8) ? ((u64)s[word + 1] << 32) : 0);
            raw = (u32)(two >> off) & mask;
        }
        u32 v = raw + carry;
        int d;
        if (v > half) { d = (int)v - (int)(1u << c); carry = 1; } else { d = (int)v; carry = 0; }
        f(w, flip ? -d : d);
    }
}

// digits[w * n + i]: bit 15 = sign, bits 0..14 = magnitude (0 = no contribution)
// nonzero_count (zero before the launch) receives the number of terms whose scalar is not zero: what the accumulation really works on
// once terms that share a point have been merged into one
__global__ void __launch_bounds__(256) k_msm_digits(const u32 *__restrict__ scalars, u64 n, msm_plan p, unsigned short *__restrict__ digits, u32 *__restrict__ big_count,
                                                    u32 *__restrict__ block_nonzero /*one word per workgroup, or null*/) {
    wave_priority_high();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) big_count[0] = 0;                       // the merge stage's list of oversized buckets starts empty
    u32 s[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = i < n ? scalars[8 * i + k] : 0u;
    if (block_nonzero) {                                // (statistics only; a count per workgroup, no contended counter)
        const int cnt = __syncthreads_count((s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7]) != 0);
        if (threadIdx.x == 0) block_nonzero[blockIdx.x] = (u32)cnt;
    }
    if (i >= n) return;
    msm_for_each_digit(s, p.c, p.W, [&](u32 w, int d) {
        u32 a = d < 0 ? (u32)(-d) : (u32)d;
        digits[(u64)w * n + i] = (unsigned short)(a | (d < 0 ? 0x8000u : 0u));
    });
}
// out[0] += the sum of n counts (one workgroup)
__global__ void __launch_bounds__(1024) k_msm_count_sum(const u32 *__restrict__ counts, u32 n, u32 *__restrict__ out) {
    __shared__ u32 part[16];
    u32 acc = 0;
    for (u32 i = threadIdx.x; i < n; i += 1024) acc += counts[i];
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int k = 0; k < 16; k++) t += part[k]; out[0] += t; }
}
static constexpr u32 SORT_UNROLL = 8;        // independent loads in flight per thread in the counting and scattering loops
// workgroup (tile, window): counts of the tile's digits per bucket -> tile_hist[(w * tiles + tile) * B + b]
__global__ void __launch_bounds__(512) k_msm_tile_hist(const unsigned short *__restrict__ digits, u64 n, msm_plan p, u32 tiles, u32 *__restrict__ tile_hist) {
    wave_priority_high();
    extern __shared__ u32 cnt[];
    const u32 tile = blockIdx.x, w = blockIdx.y;
    for (u32 b = threadIdx.x; b < p.B; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    u64 lo = (u64)tile * p.tile, hi = lo + p.tile < n ? lo + p.tile : n;
    // the loads of SORT_UNROLL iterations are issued together: one at a time, every iteration waits a full memory latency
    const unsigned short *dg = digits + (u64)w * n;
    for (u64 k0 = lo + threadIdx.x; k0 < hi; k0 += (u64)SORT_UNROLL * blockDim.x) {
        u32 a[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) { u64 k = k0 + (u64)j * blockDim.x; a[j] = k < hi ? (dg[k] & 0x7fffu) : 0u; }
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) if (a[j]) atomicAdd(&cnt[a[j] - 1], 1u);
    }
    __syncthreads();
    u32 *out = tile_hist + ((u64)w * tiles + tile) * p.B;
    for (u32 b = threadIdx.x; b < p.B; b += blockDim.x) out[b] = cnt[b];
}
// thread per (window, bucket): tile_hist becomes the count of the earlier tiles, hist the bucket's total
__global__ void __launch_bounds__(256) k_msm_tile_prefix(u32 *__restrict__ tile_hist, msm_plan p, u32 tiles, u32 *__restrict__ hist) {
    wave_priority_high();
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.W * p.B) return;
    u32 w = g / p.B, b = g % p.B;
    u32 run = 0;
    for (u32 t0 = 0; t0 < tiles; t0 += SORT_UNROLL) {                   // loads first: a load behind every store would wait for it
        u32 v[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) v[j] = t0 + j < tiles ? tile_hist[((u64)w * tiles + t0 + j) * p.B + b] : 0u;
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) if (t0 + j < tiles) { tile_hist[((u64)w * tiles + t0 + j) * p.B + b] = run; run += v[j]; }
    }
    hist[g] = run;
}

// Exclusive scan of `count` counters into offsets[count + 1], two launches:
// sums of 1024-counter tiles, then every tile adds the sums of the tiles before it to its local scan.
__device__ __forceinline__ u32 block_sum_256(u32 v, u32 *lds /* 4 */) {
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 r = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(256) k_msm_scan_tiles(const u32 *__restrict__ hist, u32 count, u32 *__restrict__ tile_sums) {
    wave_priority_high();
    __shared__ u32 lds[4];
    u32 i0 = blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
    u32 v = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) if (i0 + k < count) v += hist[i0 + k];
    u32 r = block_sum_256(v, lds);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_msm_scan(const u32 *__restrict__ hist, u32 count, const u32 *__restrict__ tile_sums, u32 *__restrict__ offsets) {
    wave_priority_high();
    __shared__ u32 lds[4];
    __shared__ u32 part[256];
    u32 t = threadIdx.x;
    u32 before = 0;
    for (u32 j = t; j < blockIdx.x; j += 256) before += tile_sums[j];
    u32 base = block_sum_256(before, lds);
    u32 i0 = blockIdx.x * SCAN_TILE + 4 * t;
    u32 v[4], sum = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) { v[k] = (i0 + k < count) ? hist[i0 + k] : 0; sum += v[k]; }
    part[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {           // Hillis-Steele inclusive scan
        u32 o = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    u32 run = base + part[t] - sum;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        if (i0 + k < count) offsets[i0 + k] = run;
        run += v[k];
        if (i0 + k + 1 == count) offsets[count] = run;
    }
}

// workgroup (tile, window): LDS counters start at the tile's first position inside each bucket's segment
// (entry_point, nullable: the entry names the term's point record instead of the term, see msm_launch)
__global__ void __launch_bounds__(512) k_msm_scatter(const unsigned short *__restrict__ digits, u64 n, msm_plan p, u32 tiles, const u32 *__restrict__ tile_hist,
                                                     const u32 *__restrict__ offsets, u32 *__restrict__ sorted, u64 sorted_cap, const u32 *__restrict__ entry_point, u32 n_points) {
    wave_priority_high();
    extern __shared__ u32 cnt[];
    const u32 tile = blockIdx.x, w = blockIdx.y;
    const u32 *pre = tile_hist + ((u64)w * tiles + tile) * p.B;
    const u32 *off = offsets + (u64)w * p.B;
    for (u32 b = threadIdx.x; b < p.B; b += blockDim.x) cnt[b] = off[b] + pre[b];
    __syncthreads();
    u64 lo = (u64)tile * p.tile, hi = lo + p.tile < n ? lo + p.tile : n;
    const unsigned short *dg = digits + (u64)w * n;
    for (u64 k0 = lo + threadIdx.x; k0 < hi; k0 += (u64)SORT_UNROLL * blockDim.x) {
        u32 dv[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) { u64 k = k0 + (u64)j * blockDim.x; dv[j] = k < hi ? (u32)dg[k] : 0u; }
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) {
            u32 d = dv[j], a = d & 0x7fffu;
            if (a) {
                u32 pos = atomicAdd(&cnt[a - 1], 1u);
                u32 name = (u32)(k0 + (u64)j * blockDim.x);
                if (entry_point) { name = entry_point[name]; if (name >= n_points) name = 0x7fffffffu; }
                if (pos < sorted_cap) sorted[pos] = name | ((d & 0x8000u) << 16);
            }
        }
    }
}
// ---- two-pass sort ---------------------------------------------------------------------------------------
// Pass 1 distributes the entries of a window over B >> f coarse bins (the high bits of the bucket index), pass 2 sorts
// each bin by the low f bits.  A pass-1 workgroup appends to few, long segments (its writes combine in L2); a pass-2
// workgroup owns one bin (tens of thousands of entries, read twice in order) and writes inside that bin's range only.
// Entry format between the passes: term index [23:0], low bucket bits [30:24], sign [31].
__global__ void __launch_bounds__(512) k_msm_coarse_hist(const unsigned short *__restrict__ digits, u64 n, msm_plan p, u32 tiles, u32 *__restrict__ tile_hist) {
    wave_priority_high();
    extern __shared__ u32 cnt[];
    const u32 tile = blockIdx.x, w = blockIdx.y, nb = p.B >> p.fine_bits;
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    u64 lo = (u64)tile * p.tile, hi = lo + p.tile < n ? lo + p.tile : n;
    // the loads of SORT_UNROLL iterations are issued together: one at a time, every iteration waits a full memory latency
    const unsigned short *dg = digits + (u64)w * n;
    for (u64 k0 = lo + threadIdx.x; k0 < hi; k0 += (u64)SORT_UNROLL * blockDim.x) {
        u32 a[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) { u64 k = k0 + (u64)j * blockDim.x; a[j] = k < hi ? (dg[k] & 0x7fffu) : 0u; }
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) if (a[j]) atomicAdd(&cnt[(a[j] - 1) >> p.fine_bits], 1u);
    }
    __syncthreads();
    u32 *out = tile_hist + ((u64)w * tiles + tile) * nb;
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) out[b] = cnt[b];
}
__global__ void __launch_bounds__(512) k_msm_coarse_scatter(const unsigned short *__restrict__ digits, u64 n, msm_plan p, u32 tiles, const u32 *__restrict__ tile_hist,
                                                            const u32 *__restrict__ coarse_off, u32 *__restrict__ tmp, u64 cap, const u32 *__restrict__ entry_point, u32 n_points) {
    wave_priority_high();
    extern __shared__ u32 cnt[];
    const u32 tile = blockIdx.x, w = blockIdx.y, nb = p.B >> p.fine_bits, fine_mask = (1u << p.fine_bits) - 1;
    const u32 *pre = tile_hist + ((u64)w * tiles + tile) * nb;
    const u32 *off = coarse_off + (u64)w * nb;
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) cnt[b] = off[b] + pre[b];
    __syncthreads();
    u64 lo = (u64)tile * p.tile, hi = lo + p.tile < n ? lo + p.tile : n;
    const unsigned short *dg = digits + (u64)w * n;
    for (u64 k0 = lo + threadIdx.x; k0 < hi; k0 += (u64)SORT_UNROLL * blockDim.x) {
        u32 dv[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) { u64 k = k0 + (u64)j * blockDim.x; dv[j] = k < hi ? (u32)dg[k] : 0u; }
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) {
            u32 d = dv[j], a = d & 0x7fffu;
            if (a) {
                u32 pos = atomicAdd(&cnt[(a - 1) >> p.fine_bits], 1u);
                u32 name = (u32)(k0 + (u64)j * blockDim.x);
                if (entry_point) { name = entry_point[name]; if (name >= n_points) name = 0xffffffu; }
                if (pos < cap) tmp[pos] = name | (((a - 1) & fine_mask) << 24) | ((d & 0x8000u) << 16);
            }
        }
    }
}
// One workgroup: per (window, coarse bin) the tiles' counts become the count of the earlier tiles, and the bins' totals
// are scanned into coarse_off[W * nb + 1].  A few thousand bins at most, so one launch replaces the three of the
// general prefix + scan (each launch boundary on the stream costs 10-20 us here, more than the work).
__global__ void __launch_bounds__(1024) k_msm_bin_offsets(u32 *__restrict__ tile_hist, u32 n_bins, u32 nb, u32 tiles, u32 *__restrict__ coarse_off) {
    wave_priority_high();
    __shared__ u32 wave_sum[16];
    __shared__ u32 base_sh;
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base_sh = 0;
    __syncthreads();
    for (u32 g0 = 0; g0 < n_bins; g0 += 1024) {
        u32 g = g0 + t, run = 0;
        if (g < n_bins) {
            u32 w = g / nb, b = g % nb;
            u32 *col = tile_hist + (u64)w * tiles * nb + b;
            for (u32 k0 = 0; k0 < tiles; k0 += SORT_UNROLL) {           // loads first: a load behind every store would wait for it
                u32 v[SORT_UNROLL];
#pragma unroll
                for (u32 j = 0; j < SORT_UNROLL; j++) v[j] = k0 + j < tiles ? col[(u64)(k0 + j) * nb] : 0u;
#pragma unroll
                for (u32 j = 0; j < SORT_UNROLL; j++) if (k0 + j < tiles) { col[(u64)(k0 + j) * nb] = run; run += v[j]; }
            }
        }
        u32 incl = run;                               // inclusive scan over the workgroup: inside the wave, then over the 16 wave sums
#pragma unroll
        for (u32 d = 1; d < 64; d <<= 1) { u32 o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        u32 before = base_sh;
        for (u32 k = 0; k < wave; k++) before += wave_sum[k];
        if (g < n_bins) coarse_off[g] = before + incl - run;
        __syncthreads();
        if (t == 1023) base_sh = before + incl;
        __syncthreads();
    }
    if (t == 0) coarse_off[n_bins] = base_sh;
}
// workgroup per (coarse bin, window): bucket offsets of the bin's 2^f buckets and the entries in bucket order
__global__ void __launch_bounds__(256) k_msm_fine_sort(const u32 *__restrict__ tmp, const u32 *__restrict__ coarse_off, msm_plan p, u32 *__restrict__ offsets,
                                                       u32 *__restrict__ sorted, u64 cap) {
    wave_priority_high();
    __shared__ u32 cnt[128], scan[128];
    const u32 bin = blockIdx.x, w = blockIdx.y, nb = p.B >> p.fine_bits, F = 1u << p.fine_bits, t = threadIdx.x;
    const u32 s = coarse_off[(u64)w * nb + bin], e = coarse_off[(u64)w * nb + bin + 1];
    if (t < F) cnt[t] = 0;
    __syncthreads();
    for (u32 k0 = s + t; k0 < e; k0 += SORT_UNROLL * blockDim.x) {
        u32 v[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) { u32 k = k0 + j * blockDim.x; v[j] = k < e ? tmp[k] : 0xffffffffu; }
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) if (k0 + j * blockDim.x < e) atomicAdd(&cnt[(v[j] >> 24) & 127u], 1u);
    }
    __syncthreads();
    u32 own = t < F ? cnt[t] : 0;
    if (t < F) scan[t] = own;
    __syncthreads();
    for (u32 d = 1; d < F; d <<= 1) {                 // inclusive scan of at most 128 counters
        u32 o = (t < F && t >= d) ? scan[t - d] : 0;
        __syncthreads();
        if (t < F) scan[t] += o;
        __syncthreads();
    }
    if (t < F) {
        u32 excl = scan[t] - own;
        offsets[(u64)w * p.B + (u64)bin * F + t] = s + excl;
        cnt[t] = s + excl;
    }
    if (t == 0 && bin == nb - 1 && w == p.W - 1) offsets[(u64)p.W * p.B] = e;
    __syncthreads();
    for (u32 k0 = s + t; k0 < e; k0 += SORT_UNROLL * blockDim.x) {
        u32 v[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) { u32 k = k0 + j * blockDim.x; v[j] = k < e ? tmp[k] : 0u; }
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) {
            if (k0 + j * blockDim.x < e) {
                u32 pos = atomicAdd(&cnt[(v[j] >> 24) & 127u], 1u);
                if (pos < cap) sorted[pos] = v[j] & 0x80ffffffu;
            }
        }
    }
}

// ---- the two passes with their writes staged in LDS ---------------------------------------------------
// The kernels above write every entry where its counter says, four bytes at a time to as many places as there are bins: the
// memory system combines only part of that (WRITE_SIZE: 3.2 and 2.3 times the bytes stored).  Here a workgroup takes its
// entries a few thousand at a time, orders them by bin inside LDS (count, scan, place) and then writes the LDS buffer out in
// order: consecutive entries of the buffer belong to one bin and go to consecutive addresses, whole lines at a time.
static constexpr u32 STAGE_PER_THREAD = 8;
// exclusive scan of cnt[0..nb) (nb <= 256) by the workgroup's first nb threads; returns this thread's own count and start
__device__ __forceinline__ void staged_scan(u32 *cnt, u32 *scan, u32 nb, u32 t, u32 &own, u32 &start) {
    own = t < nb ? cnt[t] : 0;
    if (t < nb) scan[t] = own;
    __syncthreads();
    for (u32 d = 1; d < nb; d <<= 1) {
        u32 o = (t < nb && t >= d) ? scan[t - d] : 0;
        __syncthreads();
        if (t < nb) scan[t] += o;
        __syncthreads();
    }
    start = t < nb ? scan[t] - own : 0;
}
__global__ void __launch_bounds__(512) k_msm_coarse_scatter_staged(const unsigned short *__restrict__ digits, u64 n, msm_plan p, u32 tiles, const u32 *__restrict__ tile_hist,
                                                                   const u32 *__restrict__ coarse_off, u32 *__restrict__ tmp, u64 cap, const u32 *__restrict__ entry_point, u32 n_points) {
    wave_priority_high();
    constexpr u32 SUB = 512 * STAGE_PER_THREAD;
    __shared__ u32 stage_val[SUB];
    __shared__ unsigned char stage_bin[SUB];
    __shared__ u32 gcur[256], lcnt[256], scan[256], delta[256];
    __shared__ u32 total_sh;
    const u32 tile = blockIdx.x, w = blockIdx.y, nb = p.B >> p.fine_bits, fine_mask = (1u << p.fine_bits) - 1, t = threadIdx.x;
    const u32 *pre = tile_hist + ((u64)w * tiles + tile) * nb;
    const u32 *off = coarse_off + (u64)w * nb;
    if (t < nb) { gcur[t] = off[t] + pre[t]; lcnt[t] = 0; }
    __syncthreads();
    const u64 lo = (u64)tile * p.tile, hi = lo + p.tile < n ? lo + p.tile : n;
    const unsigned short *dg = digits + (u64)w * n;
    for (u64 base = lo; base < hi; base += SUB) {
        u32 dv[STAGE_PER_THREAD];
#pragma unroll
        for (u32 j = 0; j < STAGE_PER_THREAD; j++) { const u64 k = base + t + (u64)j * 512; dv[j] = k < hi ? (u32)dg[k] : 0u; }
#pragma unroll
        for (u32 j = 0; j < STAGE_PER_THREAD; j++) { const u32 a = dv[j] & 0x7fffu; if (a) atomicAdd(&lcnt[(a - 1) >> p.fine_bits], 1u); }
        __syncthreads();
        u32 own, start;
        staged_scan(lcnt, scan, nb, t, own, start);
        if (t < nb) { delta[t] = gcur[t] - start; lcnt[t] = start; }          // lcnt becomes the bin's cursor inside the stage
        if (t == nb - 1) total_sh = start + own;
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < STAGE_PER_THREAD; j++) {
            const u32 d = dv[j], a = d & 0x7fffu;
            if (a) {
                const u32 bin = (a - 1) >> p.fine_bits;
                const u32 at = atomicAdd(&lcnt[bin], 1u);
                u32 name = (u32)(base + t + (u64)j * 512);
                if (entry_point) { name = entry_point[name]; if (name >= n_points) name = 0xffffffu; }
                stage_val[at] = name | (((a - 1) & fine_mask) << 24) | ((d & 0x8000u) << 16);
                stage_bin[at] = (unsigned char)bin;
            }
        }
        __syncthreads();
        const u32 total = total_sh;
        for (u32 k = t; k < total; k += 512) {
            const u32 pos = delta[stage_bin[k]] + k;               // (32-bit on purpose: delta = cursor - start may have wrapped)
            if (pos < cap) tmp[pos] = stage_val[k];
        }
        __syncthreads();
        if (t < nb) { gcur[t] += own; lcnt[t] = 0; }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_msm_fine_sort_staged(const u32 *__restrict__ tmp, const u32 *__restrict__ coarse_off, msm_plan p, u32 *__restrict__ offsets,
                                                              u32 *__restrict__ sorted, u64 cap) {
    wave_priority_high();
    constexpr u32 PER = 2 * STAGE_PER_THREAD, SUB = 256 * PER;
    __shared__ u32 stage_val[SUB];
    __shared__ unsigned char stage_b[SUB];
    __shared__ u32 cur[128], lcnt[128], scan[128], delta[128];
    __shared__ u32 total_sh;
    const u32 bin = blockIdx.x, w = blockIdx.y, nb = p.B >> p.fine_bits, F = 1u << p.fine_bits, t = threadIdx.x;
    const u32 s = coarse_off[(u64)w * nb + bin], e = coarse_off[(u64)w * nb + bin + 1];
    if (t < F) lcnt[t] = 0;
    __syncthreads();
    // the bin's buckets: counts over the whole bin, their offsets
    for (u32 k0 = s + t; k0 < e; k0 += SORT_UNROLL * blockDim.x) {
        u32 v[SORT_UNROLL];
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) { u32 k = k0 + j * blockDim.x; v[j] = k < e ? tmp[k] : 0xffffffffu; }
#pragma unroll
        for (u32 j = 0; j < SORT_UNROLL; j++) if (k0 + j * blockDim.x < e) atomicAdd(&lcnt[(v[j] >> 24) & 127u], 1u);
    }
    __syncthreads();
    {
        u32 own, start;
        staged_scan(lcnt, scan, F, t, own, start);
        if (t < F) { offsets[(u64)w * p.B + (u64)bin * F + t] = s + start; cur[t] = s + start; lcnt[t] = 0; }
        if (t == 0 && bin == nb - 1 && w == p.W - 1) offsets[(u64)p.W * p.B] = e;
        __syncthreads();
    }
    for (u32 base = s; base < e; base += SUB) {
        u32 v[PER];
#pragma unroll
        for (u32 j = 0; j < PER; j++) { const u32 k = base + t + j * 256; v[j] = k < e ? tmp[k] : 0u; }
#pragma unroll
        for (u32 j = 0; j < PER; j++) if (base + t + j * 256 < e) atomicAdd(&lcnt[(v[j] >> 24) & 127u], 1u);
        __syncthreads();
        u32 own, start;
        staged_scan(lcnt, scan, F, t, own, start);
        if (t < F) { delta[t] = cur[t] - start; lcnt[t] = start; }
        if (t == F - 1) total_sh = start + own;
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < PER; j++) {
            if (base + t + j * 256 < e) {
                const u32 b = (v[j] >> 24) & 127u;
                const u32 at = atomicAdd(&lcnt[b], 1u);
                stage_val[at] = v[j] & 0x80ffffffu;
                stage_b[at] = (unsigned char)b;
            }
        }
        __syncthreads();
        const u32 total = total_sh;
        for (u32 k = t; k < total; k += 256) {
            const u32 pos = delta[stage_b[k]] + k;
            if (pos < cap) sorted[pos] = stage_val[k];
        }
        __syncthreads();
        if (t < F) { cur[t] += own; lcnt[t] = 0; }
        __syncthreads();
    }
}

// Bucket sums and the chunks' partial sums in memory: the four coordinates as the ten limbs the arithmetic leaves them in (40
// words), not as 32 canonical bytes each.  A bucket border inside a wavefront is paid by all of its lanes, and packing a point
// costs about 400 instructions; with a few dozen entries per bucket (a batch verified in parts) nearly every iteration of a
// wavefront has a lane at a border.
__device__ __forceinline__ void sum_store(u32 *__restrict__ dst, const ge &p) {
    uint4 *d = (uint4 *)dst;
    const u32 w[40] = {p.X.v[0], p.X.v[1], p.X.v[2], p.X.v[3], p.X.v[4], p.X.v[5], p.X.v[6], p.X.v[7], p.X.v[8], p.X.v[9],
                       p.Y.v[0], p.Y.v[1], p.Y.v[2], p.Y.v[3], p.Y.v[4], p.Y.v[5], p.Y.v[6], p.Y.v[7], p.Y.v[8], p.Y.v[9],
                       p.Z.v[0], p.Z.v[1], p.Z.v[2], p.Z.v[3], p.Z.v[4], p.Z.v[5], p.Z.v[6], p.Z.v[7], p.Z.v[8], p.Z.v[9],
                       p.T.v[0], p.T.v[1], p.T.v[2], p.T.v[3], p.T.v[4], p.T.v[5], p.T.v[6], p.T.v[7], p.T.v[8], p.T.v[9]};
#pragma unroll
    for (int j = 0; j < 10; j++) d[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
__device__ __forceinline__ ge sum_load(const u32 *__restrict__ src) {
    const uint4 *s4 = (const uint4 *)src;
    u32 w[40];
#pragma unroll
    for (int j = 0; j < 10; j++) { const uint4 v = s4[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
    ge p;
#pragma unroll
    for (int i = 0; i < 10; i++) { p.X.v[i] = w[i]; p.Y.v[i] = w[10 + i]; p.Z.v[i] = w[20 + i]; p.T.v[i] = w[30 + i]; }
    return p;
}

// Thread per chunk [t*M, (t+1)*M) of the bucket-sorted entries.  term_point (nullable): term index ->
// record index in niels[]; n_points bounds the gather.  (When the sort already wrote record indices, msm_launch passes no
// map and the record count as n.)  chunk_sums[2t] = partial sum of the bucket that
// began before the chunk, chunk_sums[2t+1] = partial sum of the bucket that continues after it.
__global__ void __launch_bounds__(256, 2) k_msm_accumulate_regs(const u32 *__restrict__ niels, u64 n, const u32 *__restrict__ term_point, u64 n_points,
                                                           const u32 *__restrict__ offsets, const u32 *__restrict__ sorted, u32 n_buckets, u32 M,
                                                           u32 *__restrict__ bucket_sums, u32 *__restrict__ chunk_sums) {
    wave_priority_high();
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 total = offsets[n_buckets];
    u64 start64 = (u64)t * M;
    if (start64 >= total) return;
    u32 start = (u32)start64, stop = (total - start < M) ? total : start + M;
    // the bucket holding `start`: the largest b with offsets[b] <= start (empty buckets share an offset)
    u32 lo = 0, hi = n_buckets;
    while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (offsets[mid] <= start) lo = mid; else hi = mid; }
    u32 b = lo, bend = offsets[b + 1];
    bool began_before = offsets[b] < start;
    ge acc = ge_identity();
    // The record of entry k + 1 is requested before the addition of entry k starts: the three dependent loads
    // (sorted entry -> point index -> 96-byte record, mostly Infinity Cache hits) then overlap ~2000 instructions of arithmetic.
    // An index that does not name a point (never the case for a consistent sort) reads record 0 and is skipped.
    auto fetch = [&](u32 k, u32 &e, bool &valid, u32 raw[24]) {
        e = sorted[k];
        u32 idx = e & 0x7fffffffu;
        valid = idx < n;
        if (term_point) { idx = term_point[valid ? idx : 0]; valid = valid && idx < n_points; }
        const uint4 *src = (const uint4 *)(niels + 32ull * (valid ? idx : 0));
#pragma unroll
        for (int j = 0; j < 6; j++) { uint4 v = src[j]; raw[4 * j] = v.x; raw[4 * j + 1] = v.y; raw[4 * j + 2] = v.z; raw[4 * j + 3] = v.w; }
    };
    u32 e_cur, raw_cur[24]; bool valid_cur;
    fetch(start, e_cur, valid_cur, raw_cur);
    // The first record must have ARRIVED before the loop is entered: inside the loop the current record's registers are then
    // written by register moves only, and the compiler need not drain the loads in flight (the fresh prefetch) before every
    // addition -- with the first record's loads still pending at the loop's head it did, because the bucket-border path in
    // between may issue any number of loads of its own.
#pragma unroll
    for (int j = 0; j < 24; j++) asm volatile("" : "+v"(raw_cur[j]));
    for (u32 k = start; k < stop; k++) {
        u32 e_next = 0, raw_next[24]; bool valid_next = false;
        if (k + 1 < stop) fetch(k + 1, e_next, valid_next, raw_next);
        if (k >= bend) {                              // bucket b ended exactly at k
            sum_store(began_before ? chunk_sums + 2ull * SUM_WORDS * t : bucket_sums + (u64)SUM_WORDS * b, acc);
            acc = ge_identity(); began_before = false;
            do { b++; bend = offsets[b + 1]; } while (k >= bend);     // b < n_buckets since k < total
        }
        if (valid_cur) {
            ge_niels q; q.ypx = fe_unpack(raw_cur); q.ymx = fe_unpack(raw_cur + 8); q.xy2d = fe_unpack(raw_cur + 16);
            acc = ge_madd_signed(acc, q, (e_cur >> 31) != 0);
        }
        e_cur = e_next; valid_cur = valid_next;
#pragma unroll
        for (int j = 0; j < 24; j++) raw_cur[j] = raw_next[j];
    }
    u32 *dst;
    if (began_before) dst = chunk_sums + 2ull * SUM_WORDS * t;       // also when the bucket runs on past the chunk
    else if (bend <= stop) dst = bucket_sums + (u64)SUM_WORDS * b;   // the bucket is complete inside the chunk
    else dst = chunk_sums + 2ull * SUM_WORDS * t + SUM_WORDS;
    sum_store(dst, acc);
}

// The accumulation with the prefetched record held in LDS instead of in registers (k_msm_accumulate_regs above is the register
// form, kept for entries that name terms and as XHE_MSM_ACC_STAGED=0).  A lane's next record is requested with six LDS-DMA loads
// (global_load_lds_dwordx4: no register destination; one instruction writes 64 lanes x 16 bytes contiguously, so piece j of
// every lane's record lies at stage[wave][j][lane]) as soon as the current record has been read out of LDS, and arrives while
// the addition runs.  Two waits guard the one buffer: vmcnt(0) before the record is read (it has arrived), lgkmcnt(0) before
// the next request (every read of it has returned).  Without the 24 registers of a record in flight the kernel fits 128
// registers, four wavefronts per SIMD instead of three (5.62 -> 5.51 ms on the 100 000 x 8 batch: the stage is bound by issue
// slots, not by latency).  Entries name point records (no term -> point map in this form).
template <int A, int B>
__device__ __forceinline__ fe_t<(A > B ? A : B)> fe_pick(const fe_t<A> &a, const fe_t<B> &b, bool flag) {
    fe_t<(A > B ? A : B)> r; const u32 m = flag ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] ^ (m & (a.v[i] ^ b.v[i]));
    return r;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_msm_accumulate(const u32 *__restrict__ niels, u64 n_points, const u32 *__restrict__ offsets, const u32 *__restrict__ sorted, u32 n_buckets, u32 M,
                        u32 *__restrict__ bucket_sums, u32 *__restrict__ chunk_sums) {
    __shared__ __attribute__((aligned(16))) u32 stage[4 * 6 * 256];      // [wave][piece][lane][4 words]
    wave_priority_high();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 *wbase = stage + wave * (6 * 256);
    const u32 *mine = wbase + 4 * lane;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 total = offsets[n_buckets];
    u64 start64 = (u64)t * M;
    if (start64 >= total) return;
    u32 start = (u32)start64, stop = (total - start < M) ? total : start + M;
    u32 lo = 0, hi = n_buckets;
    while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (offsets[mid] <= start) lo = mid; else hi = mid; }
    u32 b = lo, bend = offsets[b + 1];
    bool began_before = offsets[b] < start;
    ge acc = ge_identity();
    // an index that does not name a point (never the case for a consistent sort) requests record 0 and is skipped
    auto request = [&](u32 e) -> bool {
        const u32 idx = e & 0x7fffffffu;
        const bool valid = idx < n_points;
        const u32 *src = niels + 32ull * (valid ? idx : 0);
#pragma unroll
        for (int j = 0; j < 6; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * j), (__attribute__((address_space(3))) void *)(wbase + 256 * j), 16, 0, 0);
        return valid;
    };
    auto staged = [&](u32 field) -> fe {           // 0: y + x, 1: y - x, 2: 2dxy
        u32 w[8];
        const uint4 a = *(const uint4 *)(mine + 256 * (2 * field)), c = *(const uint4 *)(mine + 256 * (2 * field + 1));
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = c.x; w[5] = c.y; w[6] = c.z; w[7] = c.w;
        return fe_unpack(w);
    };
    u32 e_cur = sorted[start];
    bool valid_cur = request(e_cur);
    for (u32 k = start; k < stop; k++) {
        const bool more = k + 1 < stop;
        const u32 e_next = more ? sorted[k + 1] : 0u;
        if (k >= bend) {                              // bucket b ended exactly at k
            sum_store(began_before ? chunk_sums + 2ull * SUM_WORDS * t : bucket_sums + (u64)SUM_WORDS * b, acc);
            acc = ge_identity(); began_before = false;
            do { b++; bend = offsets[b + 1]; } while (k >= bend);     // b < n_buckets since k < total
        }
        // the record of entry k has arrived in LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool negative = (e_cur >> 31) != 0;
        // p + q and p - q differ by which of (y + x, y - x) meets which factor and by the sign of the 2dxy product (taken on T)
        const fe q_a = staged(negative ? 0u : 1u), q_b = staged(negative ? 1u : 0u), q_c = staged(2u);
        fe A = fe_mul(fe_sub(acc.Y, acc.X), q_a);
        fe Bp = fe_mul(fe_add(acc.Y, acc.X), q_b);
        fe C = fe_mul(fe_pick(acc.T, fe_sub(fe_zero(), acc.T), negative), q_c);
        // every read of the staged record has returned (the products above may be scheduled behind this point, their operands
        // are in registers by now): the next record may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bool valid_next = false;
        if (more) valid_next = request(e_next);
        if (valid_cur) {
            auto D = fe_add(acc.Z, acc.Z);
            acc = ge_from_completed(fe_sub(Bp, A), fe_sub(D, C), fe_add(D, C), fe_add(Bp, A));
        }
        e_cur = e_next; valid_cur = valid_next;
    }
    u32 *dst;
    if (began_before) dst = chunk_sums + 2ull * SUM_WORDS * t;       // also when the bucket runs on past the chunk
    else if (bend <= stop) dst = bucket_sums + (u64)SUM_WORDS * b;   // the bucket is complete inside the chunk
    else dst = chunk_sums + 2ull * SUM_WORDS * t + SUM_WORDS;
    sum_store(dst, acc);
}

// thread per bucket: empty buckets get the identity, buckets inside one chunk are already final, the others
// add their partial sums.  big_list[0] counts the buckets deferred to k_msm_merge_big, ids from big_list[1].
__global__ void __launch_bounds__(256, 2) k_msm_merge(const u32 *__restrict__ offsets, u32 n_buckets, u32 M, const u32 *__restrict__ chunk_sums,
                                                      u32 *__restrict__ bucket_sums, u32 *__restrict__ big_list, u32 big_cap) {
    wave_priority_high();
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_buckets) return;
    u32 lo = offsets[g], hi = offsets[g + 1];
    if (lo == hi) { sum_store(bucket_sums + (u64)SUM_WORDS * g, ge_identity()); return; }
    u32 c0 = lo / M, c1 = (hi - 1) / M;
    if (c0 == c1) return;
    if (c1 - c0 > MERGE_SPAN_MAX) {
        u32 pos = atomicAdd(&big_list[0], 1u);
        if (pos < big_cap) big_list[1 + pos] = g;
        return;
    }
    ge acc = sum_load(chunk_sums + 2ull * SUM_WORDS * c0 + SUM_WORDS);        // the first chunk holds it as its trailing partial
    for (u32 c = c0 + 1; c <= c1; c++) acc = ge_add(acc, sum_load(chunk_sums + 2ull * SUM_WORDS * c));
    sum_store(bucket_sums + (u64)SUM_WORDS * g, acc);
}

// wavefront per deferred bucket (grid-stride over the list): the lanes add the chunk partials with stride 64 and fold with
// shuffles.  (A workgroup of four wavefronts per bucket left three of four lanes idle for the usual spans of some tens of chunks
// and needed four rounds of workgroups for the 2 048 buckets of the top window: 240 us against 60 us.)
__global__ void __launch_bounds__(256, 2) k_msm_merge_big(const u32 *__restrict__ offsets, u32 M, const u32 *__restrict__ chunk_sums, u32 *__restrict__ bucket_sums,
                                                          const u32 *__restrict__ big_list, u32 big_cap) {
    wave_priority_high();
    u32 count = big_list[0]; if (count > big_cap) count = big_cap;
    const u32 lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (u32 bi = wave; bi < count; bi += n_waves) {
        u32 g = big_list[1 + bi];
        u32 lo = offsets[g], hi = offsets[g + 1];
        u32 c0 = lo / M, c1 = (hi - 1) / M;
        ge acc = ge_identity();
        if (lane == 0) acc = sum_load(chunk_sums + 2ull * SUM_WORDS * c0 + SUM_WORDS);       // the first chunk holds the bucket as its trailing partial
        for (u32 c = c0 + 1 + lane; c <= c1; c += 64) acc = ge_add(acc, sum_load(chunk_sums + 2ull * SUM_WORDS * c));
#pragma unroll
        for (u32 d = 32; d >= 1; d >>= 1) {
            ge o = ge_shfl_down(acc, d);
            ge sum = ge_add(acc, o);
            acc = ge_select(acc, sum, lane < d);
        }
        if (lane == 0) sum_store(bucket_sums + (u64)SUM_WORDS * g, acc);
    }
}

// Given per-lane (T, U): returns in lane 0  sum_l U_l + mult * sum_l l * T_l, and the plain sum of T
// in t_total (lane 0).  mult is a power of two (log2_mult doublings).
__device__ __forceinline__ ge wave_weighted_sum(const ge &T, const ge &U, u32 log2_mult, ge &t_total) {
    u32 lane = threadIdx.x & 63;
    // inclusive suffix scan of T: S_l = sum_{k >= l} T_k
    ge S = T;
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        ge o = ge_shfl_down(S, d);
        ge sum = ge_add(S, o);
        S = ge_select(S, sum, lane + d < 64);
    }
    t_total = S;                                       // lane 0 holds the total
    // sum_l l*T_l = sum_{k=1..63} S_k
    ge Q = ge_select(S, ge_identity(), lane == 0);
    for (u32 k = 0; k < log2_mult; k++) Q = ge_double(Q);
    Q = ge_add(Q, U);
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) {
        ge o = ge_shfl_down(Q, d);
        ge sum = ge_add(Q, o);
        Q = ge_select(Q, sum, lane < d);
    }
    return Q;
}

// one wave per (window, segment of 64*L buckets): writes T (plain sum) and V (sum weighted by the
// position inside the segment, 1-based) to seg_out[(w*S + s)*2 + {0,1}]
__global__ void __launch_bounds__(64) k_msm_reduce_segments(const u32 *__restrict__ bucket_sums, msm_plan p, u32 *__restrict__ seg_out) {
    wave_priority_high();
    u32 w = blockIdx.x / p.S, s = blockIdx.x % p.S;
    u32 lane = threadIdx.x;
    u32 base = s * 64 * p.L + lane * p.L;             // first bucket index (0-based) of this lane
    ge run = ge_identity(), usum = ge_identity();
    for (int j = (int)p.L - 1; j >= 0; j--) {
        u32 b = base + (u32)j;
        if (b < p.B) run = ge_add(run, sum_load(bucket_sums + (u64)SUM_WORDS * ((u64)w * p.B + b)));
        usum = ge_add(usum, run);
    }
    u32 log2L = 0; while ((1u << log2L) < p.L) log2L++;
    ge total;
    ge V = wave_weighted_sum(run, usum, log2L, total);
    if (lane == 0) {
        u64 o = ((u64)w * p.S + s) * 2;
        ge_store(seg_out + 32 * o, total);
        ge_store(seg_out + 32 * (o + 1), V);
    }
}

// one wave per window over its S <= 64 segments: window sum = sum_s V_s + 64 L * sum_s s * T_s
__global__ void __launch_bounds__(64) k_msm_reduce_windows(const u32 *__restrict__ seg, msm_plan p, u32 *__restrict__ win_out) {
    wave_priority_high();
    u32 w = blockIdx.x, lane = threadIdx.x;
    ge T = ge_identity(), V = ge_identity();
    if (lane < p.S) {
        u64 o = ((u64)w * p.S + lane) * 2;
        T = ge_load(seg + 32 * o); V = ge_load(seg + 32 * (o + 1));
    }
    u32 log2m = 6; while ((1u << (log2m - 6)) < p.L) log2m++;      // 64 * L
    ge total;
    ge R = wave_weighted_sum(T, V, log2m, total);
    if (lane == 0) ge_store(win_out + 32ull * w, R);
}

// Horner over the windows, most significant first.  The 252 doublings are one dependent chain, so the whole
// wavefront shares the point (xhe_wide.h: a lane per limb): a doubling is two ten-step products.
// Writes the identity flag and, when out_words is not null, the compressed result.
// out_flag[1] receives the number of sorted entries (non-zero digits over all windows) the accumulation went through.
__global__ void __launch_bounds__(64) k_msm_horner(const u32 *__restrict__ win, msm_plan p, const u32 *__restrict__ extra_point /*may be null*/,
                                                   u32 *__restrict__ out_words /*8, may be null*/, u32 *__restrict__ out_flag, const u32 *__restrict__ entries_total) {
    wave_priority_high();
    const wd_lane L = wd_lane_init();
    wd_point R = wd_load(win + 32ull * (p.W - 1), L);
    for (int w = (int)p.W - 2; w >= 0; w--) {
        wd_cached q = wd_to_cached(wd_load(win + 32ull * w, L), L);       // off the chain: overlaps the doublings
#pragma unroll 1
        for (u32 k = 0; k < p.c; k++) R = wd_double(R, L);
        R = wd_add(R, q, L);
    }
    if (extra_point) R = wd_add(R, wd_to_cached(wd_load(extra_point, L), L), L);
    ge full = wd_gather(R);
    bool ident = ge_ristretto_is_identity(full);
    if (out_words) {
        u32 enc[8]; ristretto_encode_words(enc, full);
        if (threadIdx.x == 0) for (int i = 0; i < 8; i++) out_words[i] = enc[i];
    }
    if (threadIdx.x == 0) { out_flag[0] = ident ? 1u : 0u; out_flag[1] = entries_total[0]; }
}

// ---- fixed points: table look-ups instead of buckets ---------------------------------------------------
__device__ __forceinline__ ge block_point_sum(ge acc, u32 (*part)[40]) {
    u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) {
        ge o = ge_shfl_down(acc, d);
        ge sum = ge_add(acc, o);
        acc = ge_select(acc, sum, lane < d);
    }
    if (lane == 0 && wave > 0) {
#pragma unroll
        for (int i = 0; i < 10; i++) { part[wave - 1][i] = acc.X.v[i]; part[wave - 1][10 + i] = acc.Y.v[i]; part[wave - 1][20 + i] = acc.Z.v[i]; part[wave - 1][30 + i] = acc.T.v[i]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (u32 w = 0; w + 1 < (blockDim.x >> 6); w++) {
            ge o;
#pragma unroll
            for (int i = 0; i < 10; i++) { o.X.v[i] = part[w][i]; o.Y.v[i] = part[w][10 + i]; o.Z.v[i] = part[w][20 + i]; o.T.v[i] = part[w][30 + i]; }
            acc = ge_add(acc, o);
        }
    }
    return acc;                                       // thread 0 holds the workgroup's sum
}
// thread per (fixed point, byte of its scalar): the table entry for that byte; workgroup sums to partials[]
// scalars: B, H, G_vec[nm_cap], H_vec[nm_cap]; the tables hold B, H, G_vec[table_cap], H_vec[table_cap]
__global__ void __launch_bounds__(256, 2) k_static_eval(const u32 *__restrict__ gen_tab, const u32 *__restrict__ scalars, u32 nm_cap, u32 table_cap,
                                                        u32 *__restrict__ partials) {
    wave_priority_mid();
    __shared__ u32 part[3][40];
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    ge acc = ge_identity();
    if (g < 32 * (2 + 2 * nm_cap)) {
        u32 term = g >> 5, w = g & 31;
        u32 base = term < 2 + nm_cap ? term : term - nm_cap + table_cap;
        u32 byte = (scalars[8ull * term + (w >> 2)] >> ((w & 3) * 8)) & 255;
        if (byte) acc = ge_madd(acc, ge_niels_load(gen_tab + 32ull * (((u64)base * 32 + w) * 256 + byte)));
    }
    acc = block_point_sum(acc, part);
    if (threadIdx.x == 0) ge_store(partials + 32ull * blockIdx.x, acc);
}
__global__ void __launch_bounds__(64) k_static_sum(const u32 *__restrict__ partials, u32 n, u32 *__restrict__ out) {
    wave_priority_mid();
    u32 lane = threadIdx.x;
    ge acc = ge_identity();
    for (u32 i = lane; i < n; i += 64) acc = ge_add(acc, ge_load(partials + 32ull * i));
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) {
        ge o = ge_shfl_down(acc, d);
        ge sum = ge_add(acc, o);
        acc = ge_select(acc, sum, lane < d);
    }
    if (lane == 0) ge_store(out, acc);
}
void launch_static_eval(hipStream_t st, const u32 *gen_tab, const u32 *scalars, u32 nm_cap, u32 table_cap, u32 *partials, u32 *out) {
    u32 blocks = grid_for(32ull * (2 + 2 * nm_cap), 256);
    hipLaunchKernelGGL(k_static_eval, dim3(blocks), dim3(256), 0, st, gen_tab, scalars, nm_cap, table_cap, partials);
    hipLaunchKernelGGL(k_static_sum, dim3(1), dim3(64), 0, st, partials, blocks, out);
}

hipError_t msm_launch(hipStream_t st, const u32 *d_scalars, const u32 *d_niels, const u32 *d_term_point, u64 n_points, u64 n, const msm_plan &p,
                      const msm_ws &ws, u32 *d_result, bool want_encoding, const msm_deps &deps, msm_mark_fn mark, void *mark_user, bool count_terms) {
#define MARK(name) do { if (mark) { hipError_t e_ = mark(mark_user, name); if (e_ != hipSuccess) return e_; } } while (0)
    u64 n_buckets = (u64)p.W * p.B, sorted_cap = n * p.W ? n * p.W : 1, chunks = msm_chunks(n, p);
    // With a term -> point map, the sort's entries name the point record directly when the index fits the entry (31 bits, 24 in
    // the two-pass format): the scatter reads the map in term order (coalesced), and the accumulate stage goes from entry to
    // record without a random 4-byte gather per entry in between.
    // (XHE_MSM_TERM_ENTRIES=1 keeps term indices in the entries, the form batches of 2^24 points and more take: for tests)
    const bool direct = d_term_point && n_points < (p.fine_bits ? (1ull << 24) - 1 : (1ull << 31) - 1) && !setting("XHE_MSM_TERM_ENTRIES");
    const u32 *entry_point = direct ? d_term_point : nullptr;
    u32 big_cap = (u32)(chunks / MERGE_SPAN_MAX + 15);
    hipError_t e;
    const u32 tiles = (u32)msm_sort_tiles(n, p), lds_bytes = p.B * 4;
    unsigned short *digits = (unsigned short *)ws.digits;
    MARK("msm_histogram");
    // with count_terms, the terms with a non-zero scalar are counted into d_result[10] (the chunk sums are free until the accumulation)
    const u32 digit_blocks = n ? grid_for(n, 256) : 1;
    hipLaunchKernelGGL(k_msm_digits, dim3(digit_blocks), dim3(256), 0, st, d_scalars, n, p, digits, ws.big_list, count_terms ? ws.chunk_sums : (u32 *)nullptr);
    if (count_terms) hipLaunchKernelGGL(k_msm_count_sum, dim3(1), dim3(1024), 0, st, ws.chunk_sums, digit_blocks, d_result + 10);
    if (!p.fine_bits) {
        hipLaunchKernelGGL(k_msm_tile_hist, dim3(tiles, p.W), dim3(512), lds_bytes, st, digits, n, p, tiles, ws.tile_hist);
        hipLaunchKernelGGL(k_msm_tile_prefix, dim3(grid_for(n_buckets, 256)), dim3(256), 0, st, ws.tile_hist, p, tiles, ws.hist);
        MARK("msm_scan");
        u32 scan_tiles = (u32)((n_buckets + SCAN_TILE - 1) / SCAN_TILE);
        hipLaunchKernelGGL(k_msm_scan_tiles, dim3(scan_tiles), dim3(256), 0, st, ws.hist, (u32)n_buckets, ws.segs);          // segs is free until the reduce stage
        hipLaunchKernelGGL(k_msm_scan, dim3(scan_tiles), dim3(256), 0, st, ws.hist, (u32)n_buckets, ws.segs, ws.offsets);
        MARK("msm_scatter");
        hipLaunchKernelGGL(k_msm_scatter, dim3(tiles, p.W), dim3(512), lds_bytes, st, digits, n, p, tiles, ws.tile_hist, ws.offsets, ws.sorted, sorted_cap, entry_point, (u32)n_points);
    } else {
        msm_plan q = p; q.B = p.B >> p.fine_bits;      // the coarse bins play the buckets' part in the prefix and scan kernels
        const u32 n_bins = p.W * q.B;
        hipLaunchKernelGGL(k_msm_coarse_hist, dim3(tiles, p.W), dim3(512), q.B * 4, st, digits, n, p, tiles, ws.tile_hist);
        MARK("msm_scan");
        hipLaunchKernelGGL(k_msm_bin_offsets, dim3(1), dim3(1024), 0, st, ws.tile_hist, n_bins, q.B, tiles, ws.coarse);
        MARK("msm_scatter");
        // (XHE_MSM_SORT_STAGED=0: the forms that write every entry straight to its place, for tests and comparisons)
        const char *sort_env = setting("XHE_MSM_SORT_STAGED");
        if (sort_env && sort_env[0] == '0') {
            hipLaunchKernelGGL(k_msm_coarse_scatter, dim3(tiles, p.W), dim3(512), q.B * 4, st, digits, n, p, tiles, ws.tile_hist, ws.coarse, ws.tmp, sorted_cap, entry_point, (u32)n_points);
            hipLaunchKernelGGL(k_msm_fine_sort, dim3(q.B, p.W), dim3(256), 0, st, ws.tmp, ws.coarse, p, ws.offsets, ws.sorted, sorted_cap);
        } else {
            hipLaunchKernelGGL(k_msm_coarse_scatter_staged, dim3(tiles, p.W), dim3(512), 0, st, digits, n, p, tiles, ws.tile_hist, ws.coarse, ws.tmp, sorted_cap, entry_point, (u32)n_points);
            hipLaunchKernelGGL(k_msm_fine_sort_staged, dim3(q.B, p.W), dim3(256), 0, st, ws.tmp, ws.coarse, p, ws.offsets, ws.sorted, sorted_cap);
        }
    }
    MARK("msm_wait_points");
    for (hipEvent_t ev : deps.points_ready) if (ev && (e = hipStreamWaitEvent(st, ev, 0)) != hipSuccess) return e;
    MARK("msm_accumulate");
    // entries that name point records take the form with the prefetched record in LDS (XHE_MSM_ACC_STAGED=0: the register form)
    const char *staged_env = setting("XHE_MSM_ACC_STAGED");
    if ((direct || !d_term_point) && !(staged_env && staged_env[0] == '0'))
        hipLaunchKernelGGL(k_msm_accumulate, dim3(grid_for(chunks, 256)), dim3(256), 0, st, d_niels, direct ? n_points : n, ws.offsets, ws.sorted, (u32)n_buckets, p.M, ws.buckets, ws.chunk_sums);
    else
        hipLaunchKernelGGL(k_msm_accumulate_regs, dim3(grid_for(chunks, 256)), dim3(256), 0, st, d_niels, direct ? n_points : n, direct ? (const u32 *)nullptr : d_term_point, n_points, ws.offsets, ws.sorted, (u32)n_buckets, p.M,
                           ws.buckets, ws.chunk_sums);
    MARK("msm_merge");
    hipLaunchKernelGGL(k_msm_merge, dim3(grid_for(n_buckets, 256)), dim3(256), 0, st, ws.offsets, (u32)n_buckets, p.M, ws.chunk_sums, ws.buckets, ws.big_list, big_cap);
    hipLaunchKernelGGL(k_msm_merge_big, dim3(std::min<u32>((big_cap + 3) / 4, 1024)), dim3(256), 0, st, ws.offsets, p.M, ws.chunk_sums, ws.buckets, ws.big_list, big_cap);
    MARK("msm_reduce");
    hipLaunchKernelGGL(k_msm_reduce_segments, dim3(p.W * p.S), dim3(64), 0, st, ws.buckets, p, ws.segs);
    hipLaunchKernelGGL(k_msm_reduce_windows, dim3(p.W), dim3(64), 0, st, ws.segs, p, ws.wins);
    MARK("msm_horner");
    if (deps.extra_ready && (e = hipStreamWaitEvent(st, deps.extra_ready, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_msm_horner, dim3(1), dim3(64), 0, st, ws.wins, p, deps.extra_point, want_encoding ? d_result : (u32 *)nullptr, d_result + 8, ws.offsets + n_buckets);
#undef MARK
    return hipGetLastError();
}

}  // namespace xhe
