// Schnorr signature verification (src/elgamal.rs:38-65), thread per signature.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

__global__ void __launch_bounds__(64) k_schnorr(const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n, const u32 *__restrict__ pts,
                                                const u8 *__restrict__ ok, const u32 *__restrict__ htab, const u8 *__restrict__ ms_hash,
                                                u32 *__restrict__ tx_err) {
    wave_priority_high();
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev_sig sg = sigs[i];
    if (!ok[sg.key_pt]) { if (sg.key_err_key) raise_err(tx_err, sg.tx, sg.key_err_key); return; }
    u32 sw[8], ew[8];
    load_words_unaligned(sw, bytes + sg.sig_off); load_words_unaligned(ew, bytes + sg.sig_off + 32);
    // s * H from the fixed-base table (32 windows of 8 bits)
    ge R = ge_identity();
#pragma unroll 1
    for (u32 w = 0; w < 32; w++) {
        u32 b = (sw[w >> 2] >> ((w & 3) * 8)) & 255;
        if (b) R = ge_madd(R, ge_niels_load(htab + 32ull * (w * 256 + b)));
    }
    // e * P with 4-bit windows
    ge P = ge_from_niels_affine(ge_niels_load(pts + 32ull * sg.key_pt));
    ge tbl[16];
    tbl[0] = ge_identity(); tbl[1] = P;
#pragma unroll 1
    for (int k = 2; k < 16; k++) tbl[k] = ge_add(tbl[k - 1], P);
    ge Q = ge_identity();
#pragma unroll 1
    for (int k = 63; k >= 0; k--) {
        Q = ge_double(ge_double(ge_double(ge_double(Q))));
        u32 nib = (ew[k >> 3] >> ((k & 7) * 4)) & 15;
        Q = ge_add(Q, tbl[nib]);
    }
    R = ge_sub(R, Q);
    u32 rw[8]; ristretto_encode_words(rw, R);
    u8 rb[32];
    for (int k = 0; k < 8; k++) { rb[4 * k] = (u8)rw[k]; rb[4 * k + 1] = (u8)(rw[k] >> 8); rb[4 * k + 2] = (u8)(rw[k] >> 16); rb[4 * k + 3] = (u8)(rw[k] >> 24); }
    sponge sp; sp.init(72, 0x06);
    sp.absorb(bytes + sg.key_off, 32);
    if (sg.hash_idx != 0xffffffffu) sp.absorb(ms_hash + 32ull * sg.hash_idx, 32);
    else sp.absorb(bytes + sg.msg_off, sg.msg_len);
    sp.absorb(rb, 32);
    u8 h[64]; sp.squeeze(h, 64);
    u32 cw[8]; sc_to_words(cw, sc_from_wide_bytes(h));
    u32 diff = 0;
    for (int k = 0; k < 8; k++) diff |= cw[k] ^ ew[k];
    if (diff) raise_err(tx_err, sg.tx, sg.err_key);
}


void launch_schnorr(hipStream_t st, const u8 *bytes, const dev_sig *sigs, u32 n, const u32 *pts, const u8 *ok, const u32 *htab, const u8 *ms_hash, u32 *tx_err) {
    if (n) hipLaunchKernelGGL(k_schnorr, dim3(grid_for(n, 64)), dim3(64), 0, st, bytes, sigs, n, pts, ok, htab, ms_hash, tx_err);
}

}  // namespace xhe
