// fmd_boxcar_mfma.h -- the boxcar decimator (Demod::low_pass_complex, examples/simple_fm.rs:337-352, with rotate_90
// :276-299 and the `- 127` centring :258 folded in) as ONE small integer matrix product per wave-round, for an even
// downsample 2 DH with whole-dword windows (the tile kernel's "pair" mapping: lane l of a wave owns the ADJACENT windows
// 2 l and 2 l + 1 of the round, 8 DH contiguous bytes -- a "span").
//
// v_mfma_i32_16x16x64_i8 computes D[16 x 16] = A[16 x 64] * B[64 x 16] + C on signed bytes.  Here
//   column c (0 .. 15), row group g (0 .. 3)  <->  lane l = 16 g + c: the instruction's own output layout puts rows
//       4 g .. 4 g + 3 of column c into the four result registers of lane l, and those rows are
//       (re, im) of window 2 l and (re, im) of window 2 l + 1 -- exactly what the discriminator stage wants per lane;
//   B is the raw byte stream itself (u8 ^ 0x80 = b - 128 as s8): lane l = (c, k-quad g) supplies K bytes 16 g .. 16 g + 15 of
//       column c, and gives bytes 16 m .. 16 m + 15 of ITS OWN span for MFMA m of the round -- one 16-byte read (two
//       ds_read_b64) at one per-lane address plus an immediate, straight out of the staged tile;
//   A holds the +-1 byte weights of rotate_90 per (row, byte) -- the same FMD_W_* patterns the dot-product form uses --
//       and is block-diagonal: row group g only has weights in k-quad g (the other quads hold other lanes' spans).
// ceil(DH / 2) chained MFMAs replace 4 v_mov + 2 DH v_xor + 4 DH v_dot4 per round: at downsample 6 that is 22 VALU
// instructions (68 issue cycles) -> 8 v_xor + 2 matrix instructions that run on the matrix pipe beside the VALU.
// Not a contraction being invented: it is the same sum of D bytes per output, moved to the unit that adds bytes fastest.
#ifndef FMD_BOXCAR_MFMA_H
#define FMD_BOXCAR_MFMA_H

#include <stdint.h>

#include <vector>

#include "fmd_index.h"

#define FMD_BX_MAX_DH 7u                 /* even downsample 2 .. 14: the factors with a "pair" kernel of their own */

inline uint32_t fmd_bx_num_mfma(uint32_t dh) { return (dh + 1u) / 2u; }

// Fragment order [variant o1][mfma m][lane][16 bytes]; o1 = rotation parity of the lane's first window (wave-uniform).
// Lane (row r = lane & 15, k-quad kq = lane >> 4) holds A[r][16 kq .. 16 kq + 15] (the A layout of the 16x16x64 form).
inline std::vector<uint32_t> fmd_bx_build_amat(uint32_t dh)
{
    const uint32_t nm = fmd_bx_num_mfma(dh);
    std::vector<uint32_t> amat((size_t)2 * nm * 64 * 4, 0u);
    uint8_t* ab = reinterpret_cast<uint8_t*>(amat.data());
    static const uint32_t W[2][2] = {{FMD_W_RE_EVEN, FMD_W_RE_ODD}, {FMD_W_IM_EVEN, FMD_W_IM_ODD}};   // [component][dword parity]
    for (uint32_t o1 = 0; o1 < 2; ++o1)
        for (uint32_t m = 0; m < nm; ++m)
            for (uint32_t lane = 0; lane < 64; ++lane)
                for (uint32_t b = 0; b < 16; ++b) {
                    const uint32_t r = lane & 15u, g = r >> 2, reg = r & 3u, win = reg >> 1, comp = reg & 1u;
                    if ((lane >> 4) != g) continue;                               // block-diagonal: k-quad g <-> row group g
                    const uint32_t beta = 16u * m + b;                            // byte within the span
                    if (beta >= 8u * dh || beta / (4u * dh) != win) continue;     // beyond the span (odd DH) / the span's other window
                    const uint32_t bw = beta % (4u * dh), u = bw / 4u, bb = bw % 4u;
                    const uint32_t par = (o1 ^ (win ? (dh & 1u) : 0u) ^ (u & 1u)) & 1u;   // call-dword parity of this dword
                    ab[(((size_t)o1 * nm + m) * 64 + lane) * 16 + b] = (uint8_t)((W[comp][par] >> (8u * bb)) & 0xFFu);
                }
    return amat;
}

#endif  // FMD_BOXCAR_MFMA_H
