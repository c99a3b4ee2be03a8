// fmd_fir_common.h -- host-side construction of the banded Toeplitz tap matrix the matrix-core FIR kernels use
// (fmd_fir.hip: stand-alone operator; fmd_firdemod.hip: FIR fused with the discriminator and the resampler).
// See the header comment of fmd_fir.hip for the data path.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

struct FmdFirMfmaPlan {
    std::vector<uint32_t> amat;   // [n_pass * nku][64 lanes][4 dwords]: A fragments, K index = byte offset from the window start of output 0
    uint32_t n_pass = 0, nku = 0; // K-chunks of 64 bytes = n_pass * nku
    uint32_t digits = 2;          // tap digits (1: every |tap| <= 127, eight outputs per column; 2: four)
    std::vector<uint32_t> amat_s; // [nks][64 lanes][4 dwords]: the same matrix compressed for the 4:2 sparse matrix instruction (128-byte K chunks)
    uint32_t nks = 0;
    bool split = false;           // two digits, sparse: re and im rows in fragments of their own (eight outputs per column); `amat` is empty
    int32_t mre[2] = {0, 0}, mim[2] = {0, 0};   // additive constants (s8 domain) by window parity
};

// Two tap digits (any |tap| <= 2047): rows r = 4*i + reg (i: output within the quad, reg: re_lo, re_hi, im_lo, im_hi), taps split as
// h = 128*hi + lo with |lo| <= 64, |hi| <= 16 (both i8).  One digit (every |tap| <= 127: an 8-bit filter): rows r = 2*i + reg (i: output
// within the OCTET, reg: re, im) -- eight outputs per column instead of four, so a column group covers 128 outputs with the matrix
// instructions of 64.  rotate_90's signs (simple_fm.rs:276-299) and the re / im byte selection are folded in either way.
// Returns false when the shape does not fit the matrix-core form (decim > 64 or more than 64 K-chunks).
// Tap matrix, one entry: output i of the column, component comp (0: re, 1: im), digit selector dsel (0: the whole tap -- one-digit
// form; 1 / 2: the low / high i8 digit of h = 128 hi + lo), K byte index kb (offset from the window start of the column's output 0).
inline int fmd_fir_a_entry(const int16_t* taps, uint32_t n_taps, uint32_t decim, uint32_t i, uint32_t comp, uint32_t dsel, uint32_t kb)
{
    const int64_t rel = (int64_t)kb - 2ll * decim * i;
    if (rel < 0 || rel >= 2ll * n_taps) return 0;
    const uint32_t t = (uint32_t)rel >> 1, sg = (uint32_t)rel & 1u;
    const uint32_t phase = (t + 2u * ((decim / 2u * i) & 1u)) & 3u;
    int sign;
    if (comp == 0) sign = (phase == 0 && sg == 0) || (phase == 3 && sg == 1) ? 1
                        : (phase == 1 && sg == 1) || (phase == 2 && sg == 0) ? -1 : 0;
    else sign = (phase == 0 && sg == 1) || (phase == 1 && sg == 0) ? 1
              : (phase == 2 && sg == 1) || (phase == 3 && sg == 0) ? -1 : 0;
    const int h = taps[t];
    const int lo = ((h + 64) & 127) - 64, hi = (h - lo) / 128;              // h = 128*hi + lo, both i8
    return sign * (dsel == 0u ? h : (dsel == 1u ? lo : hi));
}

// Row r of a fragment in the two interleaved forms: two digits -> r = 4 i + (re_lo, re_hi, im_lo, im_hi); one digit -> r = 2 i + (re, im)
inline int fmd_fir_a_value(const int16_t* taps, uint32_t n_taps, uint32_t decim, uint32_t digits, uint32_t r, uint32_t kb)
{
    if (digits == 1u) return fmd_fir_a_entry(taps, n_taps, decim, r >> 1, r & 1u, 0u, kb);
    return fmd_fir_a_entry(taps, n_taps, decim, r >> 2, (r >> 1) & 1u, 1u + (r & 1u), kb);
}

// 4:2 compression of one fragment row set for v_smfmac_i32_16x16x128_i8 (layout: tools/smfmac_probe.hip, profiles/r05_smfmac_probe.txt):
// lane (row, q) holds 16 stored bytes; bytes 8 h ... 8 h + 7 cover the 16 dense K positions from 64 (q & 1) + 16 (q >> 1) + 32 h of the
// 128-byte chunk, two per group of four -- positions 0 and 3 for a re row, 1 and 2 for an im row (rotate_90 leaves every re row with
// bytes 0 / 3 and every im row with bytes 1 / 2 of each stream dword: the tap matrix IS 4:2 sparse, one fixed pattern per row).
// Round 6: the B operand of an ODD K quarter (lane quarter q = 1, 3: K positions 32 q ... 32 q + 31 of the chunk) is handed over with
// its two 16-byte halves SWAPPED -- those lanes read the second half of their 32 stream bytes first -- which takes the operand reads
// off each other's LDS banks (tools/ldsbench.py, tools/lds_model.py: the four quarters' first reads then cover every 16-byte slot of
// the 256-byte bank row once instead of the even slots twice).  The K permutation is the same for every column, so it is absorbed
// here: K position k of an odd quarter weighs the stream byte at k ^ 16.
inline uint32_t fmd_fir_kswap(uint32_t k_in_chunk) { return k_in_chunk ^ (((k_in_chunk >> 5) & 1u) << 4); }

template <typename RowFn>
inline void fmd_fir_compress_42(uint8_t* as, uint32_t nks, RowFn entry /* (row, comp&, kb) -> value, comp by reference */)
{
    for (uint32_t kc = 0; kc < nks; ++kc)
        for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t r = lane & 15u, q = lane >> 4;
            for (uint32_t sb = 0; sb < 16; ++sb) {
                const uint32_t h = sb >> 3, gl = (sb & 7u) >> 1, e = sb & 1u;
                uint32_t comp = 0;
                (void)entry(r, comp, 0u);                                     // (the row's component decides the pair of positions)
                const uint32_t pos = comp == 0u ? (e ? 3u : 0u) : (e ? 2u : 1u);
                const uint32_t kb = 128u * kc + fmd_fir_kswap(64u * (q & 1u) + 16u * (q >> 1) + 32u * h + 4u * gl + pos);
                as[((size_t)kc * 64 + lane) * 16 + sb] = (uint8_t)(int8_t)entry(r, comp, kb);
            }
        }
}

// `split` (two digits only): the sparse form with the components in fragments of their own -- rows r = 2 i + (lo, hi) of EIGHT outputs,
// one row set for re and one for im (amat_s = [re: nks chunks][im: nks chunks]) -- so that a column is eight outputs and consecutive
// accumulators sit a whole 128-byte sparse chunk apart, like the one-digit form's (fmd_firdemod.hip, fd_reg_body).
inline bool fmd_fir_build_mfma(const int16_t* taps, uint32_t n_taps, uint32_t decim, FmdFirMfmaPlan& P, uint32_t digits = 2, bool split = false)
{
    const uint32_t opc = (digits == 1u || split) ? 8u : 4u;   // outputs per column
    const uint32_t nk_tot = (2u * ((opc - 1u) * decim + n_taps) + 63u) / 64u;
    if (decim > 64 || nk_tot > 64) return false;
    P.digits = digits == 1u ? 1u : 2u;
    P.split = P.digits == 2u && split;
    P.n_pass = (nk_tot + 7u) / 8u;
    P.nku = (nk_tot + P.n_pass - 1u) / P.n_pass;
    const uint32_t chunks = P.n_pass * P.nku;
    P.amat.clear();
    if (!P.split) {
        P.amat.assign((size_t)chunks * 64 * 4, 0u);
        uint8_t* ab = reinterpret_cast<uint8_t*>(P.amat.data());
        for (uint32_t kk = 0; kk < chunks; ++kk)
            for (uint32_t lane = 0; lane < 64; ++lane)
                for (uint32_t b = 0; b < 16; ++b)
                    ab[((size_t)kk * 64 + lane) * 16 + b] =
                        (uint8_t)(int8_t)fmd_fir_a_value(taps, n_taps, decim, P.digits, lane & 15u, 64u * kk + 16u * (lane >> 4) + b);
    }
    for (int par = 0; par < 2; ++par) {
        int64_t sr = 0, si = 0;
        for (uint32_t t = 0; t < n_taps; ++t) {
            const uint32_t phase = (t + 2u * par) & 3u;
            if (phase == 0 || phase == 3) sr += taps[t];
            if (phase == 0 || phase == 1) si += taps[t];
        }
        P.mre[par] = (int32_t)sr; P.mim[par] = (int32_t)si;
    }
    P.nks = (chunks + 1u) / 2u;
    if (P.split) {
        P.amat_s.assign((size_t)2 * P.nks * 64 * 4, 0u);
        uint8_t* as = reinterpret_cast<uint8_t*>(P.amat_s.data());
        for (uint32_t c = 0; c < 2; ++c)
            fmd_fir_compress_42(as + (size_t)c * P.nks * 64 * 16, P.nks, [&](uint32_t r, uint32_t& comp, uint32_t kb) {
                comp = c;
                return fmd_fir_a_entry(taps, n_taps, decim, r >> 1, c, 1u + (r & 1u), kb);
            });
    } else {
        P.amat_s.assign((size_t)P.nks * 64 * 4, 0u);
        const uint32_t dg = P.digits;
        fmd_fir_compress_42(reinterpret_cast<uint8_t*>(P.amat_s.data()), P.nks, [&](uint32_t r, uint32_t& comp, uint32_t kb) {
            comp = dg == 1u ? r & 1u : (r >> 1) & 1u;
            return fmd_fir_a_value(taps, n_taps, decim, dg, r, kb);
        });
    }
    return true;
}
