// closed_form.cpp -- CPU CLOSED-FORM MODEL (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// Sequential emulation of exactly the decomposition the HIP kernel uses: the call is cut
// into tiles of `kt` audio samples with fmd_tile() (rtl-sdr-rs_amd/csrc/fmd_index.h), each
// tile recomputes its decimated samples straight from the raw bytes with the byte-weight
// form of rotate_90 + centring, and the Demod state is produced by the last tile only.
// tests/test_closed_form.py proves this equals the pass-by-pass oracle (fm_oracle.c, which
// follows examples/simple_fm.rs:256-426) for random chunkings, phases and tile sizes -- that
// is the specification the GPU kernel implements.  Nothing here is linked into the product.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "fmd_index.h"

namespace {

const double kPi = 3.14159265358979323846264338327950288;

// v_dot4_i32_i8 emulation: sum of products of four signed bytes.
int32_t sdot4(uint32_t a, uint32_t b, int32_t c)
{
    for (int i = 0; i < 4; i++) c += (int32_t)(int8_t)(a >> (8 * i)) * (int32_t)(int8_t)(b >> (8 * i));
    return c;
}

// Sum of rotated+centred samples n in [n0, n1) of a channel whose call starts at `chan`.
void window_sum(const uint8_t* chan, int64_t n0, int64_t n1, int32_t& re, int32_t& im)
{
    int32_t ar = 0, ai = 0;
    if (n1 > n0) {
        for (int64_t m = n0 >> 1; m <= (n1 - 1) >> 1; m++) {
            uint32_t w;
            memcpy(&w, chan + 4 * m, 4);
            w ^= 0x80808080u;
            uint32_t mask = 0xFFFFFFFFu;
            if (2 * m < n0) mask &= 0xFFFF0000u;
            if (2 * m + 1 >= n1) mask &= 0x0000FFFFu;
            const bool odd = m & 1;
            ar = sdot4(w, (odd ? 0x010000FFu : 0xFF000001u) & mask, ar);
            ai = sdot4(w, (odd ? 0x00FFFF00u : 0x00010100u) & mask, ai);
        }
        ar += fmd_const_re((int32_t)n1) - fmd_const_re((int32_t)n0);
        ai += fmd_const_im((int32_t)n1) - fmd_const_im((int32_t)n0);
    }
    re = ar; im = ai;
}

int32_t polar_f64(int32_t cr, int32_t ci)
{
    double angle = atan2((double)ci, (double)cr);
    return (int32_t)(angle / kPi * 16384.0);
}

}  // namespace

extern "C" {

// Returns number of s16 written, or -1 bad length, -2 fewer than two decimated samples,
// -3 capacity, -4 bad rates.  `st` is updated like Demod's fields.
// block_ns > 0: the buffer is nbytes / (2 * block_ns) consecutive reference calls (fmd_demod_set_block_len): the
// first decimated sample of every block takes the f64 path, index (p0 + b * block_ns) / D -- the same expression
// the kernel uses.
long fmcf_demodulate_blocks(uint32_t D, uint32_t fast, uint32_t slow, uint32_t kt, uint32_t block_ns, FmdChanState* st,
                            const uint8_t* buf, size_t nbytes, int16_t* out, size_t out_cap)
{
    if (nbytes % 8) return -1;
    if (block_ns && ((nbytes / 2) % block_ns != 0 || block_ns < 2 * D)) return -1;
    if (D == 0 || slow == 0 || fast < slow || kt == 0) return -4;
    FmdRates r;
    r.D = D; r.fast = fast; r.slow = slow;
    r.g = fmd_gcd(fast, slow); r.fr = fast / r.g; r.sr = slow / r.g; r.R = (int32_t)(fast / slow); r.kt = kt;
    if (!fmd_ranges_fit32(r, nbytes / 2)) return -6;

    const uint32_t p0 = st->prev_index, i0r = st->lpr_index_r;
    const uint32_t ns = (uint32_t)(nbytes / 2);
    const uint32_t M = fmd_num_decimated(D, p0, ns);
    if (M < 2) return -2;
    const uint32_t K = fmd_num_audio(r, i0r, M);
    if (K > out_cap) return -3;
    const uint32_t nt = fmd_num_tiles(r, K);
    const uint32_t lp_cap = fmd_tile_lp_cap(r), raw_cap = fmd_tile_raw_cap(r);
    FmdChanState nst = *st;

    for (uint32_t t = 0; t < nt; t++) {
        const FmdTile T = fmd_tile(r, p0, i0r, ns, M, K, nt, t);
        const int32_t jfirst = T.jA - 1;
        const int32_t cnt = T.jB - jfirst + 1;
        if (cnt < 1 || (uint32_t)cnt > lp_cap) return -100;                       // LDS sizing bound
        if ((uint32_t)(2 * (T.nHi - T.nLo) + 32) > raw_cap) return -101;
        std::vector<int32_t> lre(cnt), lim(cnt);
        std::vector<int16_t> d(cnt);
        for (int32_t i = 0; i < cnt; i++) {
            const int32_t j = jfirst + i;
            if (j < 0) { lre[i] = st->demod_pre_re; lim[i] = st->demod_pre_im; continue; }
            const int32_t n0 = fmd_win_begin(D, p0, j), n1 = fmd_win_end(D, p0, j);
            if (n0 < T.nLo || n1 > T.nHi) return -102;                            // tile covers its windows
            window_sum(buf, n0, n1, lre[i], lim[i]);
            if (j == 0) { lre[i] += st->lp_now_re; lim[i] += st->lp_now_im; }
        }
        for (int32_t i = 1; i < cnt; i++) {
            int32_t cr, ci;
            fmd_mul_conj(lre[i], lim[i], lre[i - 1], lim[i - 1], cr, ci);
            const uint32_t j = (uint32_t)(jfirst + i);
            bool first_of_a_call = j == 0;
            if (block_ns && j > 0) {                                              // j == (p0 + b*block_ns) / D for some b >= 1 ?
                const uint64_t lo = (uint64_t)j * D, hi = lo + D;                 // p0 + b*block_ns in [lo, hi)
                const uint64_t b = lo > p0 ? (lo - p0 + block_ns - 1) / block_ns : 1;
                first_of_a_call = b >= 1 && p0 + b * block_ns < hi && p0 + b * block_ns >= lo && b * (uint64_t)block_ns < ns;
            }
            const int32_t pcm = first_of_a_call ? polar_f64(cr, ci) : fmd_fast_atan2(ci, cr);
            d[i] = (int16_t)(uint16_t)(uint32_t)pcm;
        }
        for (uint32_t k = T.k0; k < T.k1; k++) {
            const uint32_t q = k - T.k0;
            const int32_t e = (int32_t)(T.eq + (T.er + q * r.fr) / r.sr);
            const int32_t s = q == 0 ? T.jA : (int32_t)(T.eq + (T.er + (q - 1) * r.fr) / r.sr) + 1;
            if (s < T.jA || e > T.jB) return -103;
            int32_t sum = k == 0 ? st->now_lpr : 0;
            for (int32_t j = s; j <= e; j++) sum += d[j - jfirst];
            out[k] = (int16_t)(uint16_t)(uint32_t)(sum / r.R);
        }
        if (T.last) {
            const int32_t s = K == 0 ? 0 : (int32_t)fmd_audio_end(r, i0r, K - 1) + 1;
            int32_t sum = K == 0 ? st->now_lpr : 0;
            for (int32_t j = s; j <= T.jB; j++) sum += d[j - jfirst];
            nst.now_lpr = sum;
            nst.lpr_index_r = fmd_next_lpr_index_r(r, i0r, M, K);
            nst.prev_index = fmd_next_prev_index(D, p0, ns);
            int32_t tr, ti;
            window_sum(buf, fmd_win_begin(D, p0, (int32_t)M), (int32_t)ns, tr, ti);
            nst.lp_now_re = tr; nst.lp_now_im = ti;
            nst.demod_pre_re = lre[cnt - 1]; nst.demod_pre_im = lim[cnt - 1];
        }
    }
    *st = nst;
    return (long)K;
}

long fmcf_demodulate(uint32_t D, uint32_t fast, uint32_t slow, uint32_t kt, FmdChanState* st,
                     const uint8_t* buf, size_t nbytes, int16_t* out, size_t out_cap)
{
    return fmcf_demodulate_blocks(D, fast, slow, kt, 0, st, buf, nbytes, out, out_cap);
}

// The planned tile form used by the production kernel (phase-class plans + tiling constants; one small
// division per tile, none when kt is a multiple of sr) must describe exactly the same tiles as fmd_tile().  Returns 0, or a positive code naming the first mismatch.
int fmcf_check_plan(uint32_t D, uint32_t fast, uint32_t slow, uint32_t kt, uint32_t p0, uint32_t i0r, uint32_t ns)
{
    FmdRates r;
    r.D = D; r.fast = fast; r.slow = slow;
    r.g = fmd_gcd(fast, slow); r.fr = fast / r.g; r.sr = slow / r.g; r.R = (int32_t)(fast / slow); r.kt = kt;
    if (!fmd_ranges_fit32(r, ns)) return -1;
    const FmdClassPlan P = fmd_make_plan(r, p0, i0r, ns);
    const uint32_t M = fmd_num_decimated(D, p0, ns), K = fmd_num_audio(r, i0r, M), nt = fmd_num_tiles(r, K);
    if (P.M != M || P.K != K || P.nt != nt) return 1;
    const FmdTiling tl = fmd_make_tiling(r);
    const uint32_t fa = r.fr / r.sr, fb = r.fr % r.sr;
    const float inv_sr = 1.0f / (float)r.sr;
    for (uint32_t t = 0; t < nt; t++) {
        const FmdTile A = fmd_tile(r, p0, i0r, ns, M, K, nt, t);
        const FmdTile B = fmd_tile_fast(r, P, tl, ns, t);
        if (A.k0 != B.k0 || A.k1 != B.k1) return 2;
        if (A.jA != B.jA) return 3;
        if (A.jB != B.jB) return 4;
        if (A.nLo != B.nLo || A.nHi != B.nHi) return 5;
        if (A.eq != B.eq || A.er != B.er || A.last != B.last) return 6;
        for (uint32_t q = 0; q < A.k1 - A.k0; q++) {
            const uint32_t e0 = A.eq + (A.er + q * r.fr) / r.sr;
            const uint32_t e1 = B.eq + q * fa + fmd_udiv_small(B.er + q * fb, r.sr, inv_sr);
            if (e0 != e1) return 7;
        }
    }
    return 0;
}

int32_t fmcf_fast_atan2_q(int32_t y, int32_t x) { return fmd_fast_atan2_q(y, x); }
uint32_t fmcf_udiv_small(uint32_t t, uint32_t d) { return fmd_udiv_small(t, d, 1.0f / (float)d); }
int32_t fmcf_sdiv_small(int32_t s, int32_t d) { return fmd_sdiv_small(s, d, 1.0f / (float)d); }

// Expose the scalar pieces so the tests can pin them against the oracle directly.
int32_t fmcf_fast_atan2(int32_t y, int32_t x) { return fmd_fast_atan2(y, x); }
void fmcf_window_sum(const uint8_t* chan, int64_t n0, int64_t n1, int32_t* re, int32_t* im)
{
    window_sum(chan, n0, n1, *re, *im);
}

}  // extern "C"
