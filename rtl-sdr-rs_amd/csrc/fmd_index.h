// fmd_index.h -- closed-form index algebra of the simple_fm demodulation chain.
//
// The reference (ccostes/rtl-sdr-rs, examples/simple_fm.rs:256-426) runs three sequential
// state machines over one call's buffer: the boxcar decimator (low_pass_complex :337-352),
// the discriminator (fm_demod :355-367) and the fractional boxcar resampler (low_pass_real
// :408-426).  None of their *index* state depends on sample values, so every output is a
// pure function of (call-start phases, position).  This header is that algebra; the HIP
// kernels (fmd_tile_body.h, fmd_generic_kernel.hip), the host bookkeeping (fmd_api.cpp) and the CPU closed-form
// model used by the tests (oracle/closed_form.cpp) all include it, so the tests exercise
// the very expressions the kernel uses.
//
// Notation (per channel, per call):
//   D      downsample                      p0   Demod.prev_index at call start, 0 <= p0 < D
//   fast   rate_out, slow rate_resample    i0   Demod.prev_lpr_index at call start, 0 <= i0 < fast
//   g      gcd(fast, slow); fr = fast/g, sr = slow/g, i0r = i0/g (i0 is always a multiple of g)
//   ns     complex samples in the call (= nbytes / 2)
//   M      decimated (= discriminator) samples produced this call
//   K      audio samples produced this call
//   lp[j]  j-th decimated sample of the call = (j == 0 ? lp_now : 0) + sum x[n], n in
//          [max(0, D*j - p0), D*j - p0 + D)
//   e(k)   index j of the discriminator sample that completes audio sample k
//
// All index arithmetic is unsigned 32-bit; fmd_ranges_fit32() is the host-side guard that
// makes that exact (a call that fails it is rejected with FMD_ERR_UNSUPPORTED).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FMD_HD __host__ __device__ __forceinline__
#else
#define FMD_HD inline
#endif

#define FMD_MAX_DOWNSAMPLE 128u        /* |lp| <= 128*D <= 16384 fits i16; no i32 overflow in fm_demod: the tile kernel's domain */
#define FMD_MAX_DOWNSAMPLE_WIDE 512u   /* 129 ... 512 (optimal_settings for rates down to 1957 Hz, simple_fm.rs:190): generic kernel, i32 samples */
#define FMD_MAX_RATE_REDUCED (1u << 24)

// Constants derived once per handle from DemodConfig (simple_fm.rs:179-185).
struct FmdRates {
    uint32_t D;      // downsample
    uint32_t fast;   // rate_out
    uint32_t slow;   // rate_resample
    uint32_t g;      // gcd(fast, slow)
    uint32_t fr;     // fast / g
    uint32_t sr;     // slow / g
    int32_t  R;      // (fast / slow) as i32, the divisor of simple_fm.rs:421
    uint32_t kt;     // audio samples per tile (kernel tiling, chosen by the host)
};

// Per-channel Demod state as it lives in HBM (32 bytes; simple_fm.rs:232-239).
// prev_lpr_index is stored divided by g.
struct FmdChanState {
    uint32_t prev_index;
    uint32_t lpr_index_r;
    int32_t  now_lpr;
    int32_t  lp_now_re, lp_now_im;
    int32_t  demod_pre_re, demod_pre_im;
    int32_t  reserved;
};

FMD_HD uint32_t fmd_gcd(uint32_t a, uint32_t b)
{
    while (b) { uint32_t t = a % b; a = b; b = t; }
    return a;
}

// Number of decimated samples: the decimator emits whenever prev_index reaches D (:343-349).
FMD_HD uint32_t fmd_num_decimated(uint32_t D, uint32_t p0, uint32_t ns) { return (p0 + ns) / D; }
FMD_HD uint32_t fmd_next_prev_index(uint32_t D, uint32_t p0, uint32_t ns) { return (p0 + ns) % D; }

// Number of audio samples: prev_lpr_index gains `slow` per input and emits (subtracting `fast`)
// whenever it reaches `fast` (:417-422); with slow <= fast at most one emit per input.
FMD_HD uint32_t fmd_num_audio(const FmdRates& r, uint32_t i0r, uint32_t M) { return (i0r + M * r.sr) / r.fr; }
FMD_HD uint32_t fmd_next_lpr_index_r(const FmdRates& r, uint32_t i0r, uint32_t M, uint32_t K)
{
    return i0r + M * r.sr - K * r.fr;
}

// Host guard for the 32-bit arithmetic above and in fmd_tile().
inline bool fmd_ranges_fit32(const FmdRates& r, uint64_t ns)
{
    if (ns > (1ull << 30)) return false;
    const uint64_t Mmax = (r.D - 1 + ns) / r.D;
    if (2ull * r.fr + Mmax * r.sr >= (1ull << 32)) return false;
    if ((uint64_t)(r.kt + 2) * r.fr >= (1ull << 32)) return false;
    return true;
}

// e(k): smallest j with i0 + (j+1)*slow >= (k+1)*fast  <=>  j = floor(((k+1)*fast - i0 - 1) / slow);
// in gcd-reduced units the same value is floor(((k+1)*fr - i0r - 1) / sr).
FMD_HD uint32_t fmd_audio_end(const FmdRates& r, uint32_t i0r, uint32_t k)
{
    return ((k + 1) * r.fr - i0r - 1) / r.sr;
}

// First/last input sample (exclusive end) of decimated sample j >= 0, clipped to the call.
FMD_HD int32_t fmd_win_begin(uint32_t D, uint32_t p0, int32_t j)
{
    int32_t n = (int32_t)D * j - (int32_t)p0;
    return n < 0 ? 0 : n;
}
FMD_HD int32_t fmd_win_end(uint32_t D, uint32_t p0, int32_t j) { return (int32_t)D * (j + 1) - (int32_t)p0; }

// Work decomposition of one channel-call into tiles of `kt` audio samples.
struct FmdTile {
    uint32_t k0, k1;   // audio samples [k0, k1) of this call
    int32_t  jA;       // first discriminator sample summed by this tile
    int32_t  jB;       // last discriminator sample handled (inclusive); the tile needs lp[jA-1 .. jB]
    int32_t  nLo, nHi; // input complex samples [nLo, nHi) the tile reads
    uint32_t eq, er;   // (k0+1)*fr - i0r - 1 = eq*sr + er, so that e(k0+q) = eq + (er + q*fr)/sr
    bool     last;     // also owns the tail (state update)
};

FMD_HD uint32_t fmd_num_tiles(const FmdRates& r, uint32_t K)
{
    uint32_t t = (K + r.kt - 1) / r.kt;
    return t ? t : 1u;
}

FMD_HD FmdTile fmd_tile(const FmdRates& r, uint32_t p0, uint32_t i0r, uint32_t ns, uint32_t M, uint32_t K,
                        uint32_t nt, uint32_t t)
{
    FmdTile T;
    T.last = (t + 1 == nt);
    T.k0 = t * r.kt;
    T.k1 = T.k0 + r.kt < K ? T.k0 + r.kt : K;
    if (T.k1 < T.k0) T.k1 = T.k0;
    const uint32_t a = (T.k0 + 1) * r.fr - i0r - 1;
    T.eq = a / r.sr;
    T.er = a - T.eq * r.sr;
    // jA = e(k0 - 1) + 1 ;  e(k0-1) = floor((a - fr) / sr)
    T.jA = T.k0 == 0 ? 0 : (int32_t)((a - r.fr) / r.sr) + 1;
    if (T.last) T.jB = (int32_t)M - 1;
    else        T.jB = (int32_t)(T.eq + (T.er + (T.k1 - T.k0 - 1) * r.fr) / r.sr);
    // decimated samples needed: jA-1 (predecessor) .. jB ; jA-1 == -1 is demod_pre (no input)
    T.nLo = T.jA >= 1 ? fmd_win_begin(r.D, p0, T.jA - 1) : 0;
    T.nHi = T.last ? (int32_t)ns : fmd_win_end(r.D, p0, T.jB);
    if (T.nHi < T.nLo) T.nHi = T.nLo;
    return T;
}

// ---- phase classes and the planned tile form -------------------------------------------------
// Channels that share (p0, i0r) share every index of the call.  With a0 = fr - i0r - 1 = eq0*sr + er0
// (per class and call, host) and the tiling constants below (per handle, host),
//     (t*kt + 1)*fr - i0r - 1 = a0 + t*kt*fr = (eq0 + t*Qt)*sr + (er0 + t*Rt),
// so tile t needs ONE small division, (er0 + t*Rt) / sr, and none at all when kt is a multiple of sr
// (Rt == 0: every tile of a class has the same structure shifted by t*Qt decimated samples).  jA and jB
// follow from (eq, er) by comparisons against host constants.
struct FmdClassPlan {
    uint32_t p0, i0r;   // call-start phases of the class
    uint32_t M, K, nt;  // decimated / audio samples and tiles of this call
    uint32_t eq0, er0;  // fr - i0r - 1 = eq0*sr + er0
    uint32_t pad;       // 32 bytes: FMD_MAX_CLASSES of them travel in the kernel arguments
};

struct FmdTiling {
    uint32_t Qt, Rt;    // kt*fr       = Qt*sr + Rt
    uint32_t fq, frr;   // fr          = fq*sr + frr
    uint32_t Bq, Br;    // (kt - 1)*fr = Bq*sr + Br
};

inline FmdTiling fmd_make_tiling(const FmdRates& r)
{
    FmdTiling g;
    const uint64_t a = (uint64_t)r.kt * r.fr, b = (uint64_t)(r.kt - 1) * r.fr;
    g.Qt = (uint32_t)(a / r.sr); g.Rt = (uint32_t)(a % r.sr);
    g.fq = r.fr / r.sr;          g.frr = r.fr % r.sr;
    g.Bq = (uint32_t)(b / r.sr); g.Br = (uint32_t)(b % r.sr);
    return g;
}

inline FmdClassPlan fmd_make_plan(const FmdRates& r, uint32_t p0, uint32_t i0r, uint32_t ns)
{
    FmdClassPlan P{};
    P.p0 = p0; P.i0r = i0r;
    P.M = fmd_num_decimated(r.D, p0, ns);
    P.K = fmd_num_audio(r, i0r, P.M);
    P.nt = fmd_num_tiles(r, P.K);
    const uint32_t a0 = r.fr - i0r - 1;
    P.eq0 = a0 / r.sr;
    P.er0 = a0 % r.sr;
    return P;
}

FMD_HD FmdTile fmd_tile_fast(const FmdRates& r, const FmdClassPlan& P, const FmdTiling& g, uint32_t ns, uint32_t t)
{
    FmdTile T;
    T.last = (t + 1 == P.nt);
    T.k0 = t * r.kt;
    T.k1 = T.k0 + r.kt < P.K ? T.k0 + r.kt : P.K;
    if (T.k1 < T.k0) T.k1 = T.k0;
    T.eq = t * g.Qt + P.eq0;
    T.er = P.er0;
    if (g.Rt) {                                          // t*Rt < t*kt*fr, which fmd_ranges_fit32 bounds
        const uint32_t x = P.er0 + t * g.Rt, qx = x / r.sr;
        T.eq += qx;
        T.er = x - qx * r.sr;
    }
    // jA = e(k0 - 1) + 1 = floor((eq*sr + er - fr) / sr) + 1;  jB = e(k1 - 1) = eq + (er + (kt-1)*fr) / sr
    T.jA = t == 0 ? 0 : (int32_t)(T.eq - g.fq + (T.er >= g.frr ? 1u : 0u));
    T.jB = T.last ? (int32_t)P.M - 1 : (int32_t)(T.eq + g.Bq + (T.er + g.Br >= r.sr ? 1u : 0u));
    T.nLo = T.jA >= 1 ? fmd_win_begin(r.D, P.p0, T.jA - 1) : 0;
    T.nHi = T.last ? (int32_t)ns : fmd_win_end(r.D, P.p0, T.jB);
    if (T.nHi < T.nLo) T.nHi = T.nLo;
    return T;
}

// floor(t / d) for 0 <= t < 2^24, 1 <= d < 2^24, t / d < 2^20, with inv_d = 1.0f / d: the f32
// estimate is within 1 of the quotient, one exact remainder fixes it.
FMD_HD uint32_t fmd_udiv_small(uint32_t t, uint32_t d, float inv_d)
{
    uint32_t q = (uint32_t)((float)t * inv_d);
    const int32_t rem = (int32_t)t - (int32_t)(q * d);
    // sign-mask arithmetic instead of selects (VCC-masked v_cndmask_b32 issues ~4x slower than an add on gfx950,
    // tools/valubench.hip): -1 when rem < 0, +1 when rem >= d
    q += (uint32_t)(rem >> 31);
    q -= (uint32_t)(((int32_t)d - 1 - rem) >> 31);
    return q;
}

// Truncating s / d (Rust `/` on i32, simple_fm.rs:421) for |s| < 2^24, 1 <= d < 2^24.
FMD_HD int32_t fmd_sdiv_small(int32_t s, int32_t d, float inv_d)
{
    const uint32_t m = (uint32_t)(s >> 31);              // (v ^ m) - m = m ? -v : v
    const uint32_t q = fmd_udiv_small(((uint32_t)s ^ m) - m, (uint32_t)d, inv_d);
    return (int32_t)((q ^ m) - m);
}

// Truncating s / d by a host-made reciprocal (round-up method): with sh = ceil(log2 d) and m = ceil(2^(31+sh) / d) < 2^32,
// floor(n / d) = mulhi(n, m) >> (sh - 1) for every 0 <= n < 2^24 (the error term n * (m d - 2^(31+sh)) / (d 2^(31+sh)) is
// below 2^-7 / d).  d == 1 is flagged by m == 0.  One v_mul_hi_u32 instead of two conversions and a v_mul_lo.
struct FmdMagic { uint32_t m, sh; };
inline FmdMagic fmd_make_magic(uint32_t d)
{
    FmdMagic g{0u, 0u};
    if (d <= 1u) return g;
    uint32_t s = 0; while ((1ull << s) < d) ++s;
    g.m = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
    g.sh = s - 1u;
    return g;
}
FMD_HD int32_t fmd_sdiv_magic(int32_t s, FmdMagic g)
{
    const uint32_t mk = (uint32_t)(s >> 31);
    const uint32_t n = ((uint32_t)s ^ mk) - mk;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t hi = __umulhi(n, g.m);
#else
    const uint32_t hi = (uint32_t)(((uint64_t)n * g.m) >> 32);
#endif
    const uint32_t q = g.m ? hi >> g.sh : n;
    return (int32_t)((q ^ mk) - mk);
}

// Truncating num / den for den > 0, |num / den| <= 4097 (the fast_atan2 quotient): f32 estimate
// within 1, exact wrapping-remainder fix-up.  Needs den < 2^30.
FMD_HD float fmd_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);                    // v_rcp_f32, 1 ulp
#else
    return 1.0f / x;
#endif
}

FMD_HD int32_t fmd_sdiv_q13(int32_t num, int32_t den)
{
    int32_t q = (int32_t)((float)num * fmd_rcp((float)den));
    const int32_t rem = (int32_t)((uint32_t)num - (uint32_t)q * (uint32_t)den);
    if (num >= 0) { if (rem < 0) --q; else if (rem >= den) ++q; }
    else          { if (rem > 0) ++q; else if (rem <= -den) --q; }
    return q;
}

// Upper bounds used to size LDS: decimated samples and input bytes one tile can touch.
inline uint32_t fmd_tile_lp_cap(const FmdRates& r)
{
    const uint64_t per = ((uint64_t)r.kt * r.fr + r.sr - 1) / r.sr;
    const uint64_t one = ((uint64_t)r.fr + r.sr - 1) / r.sr;
    return (uint32_t)(per + one + 3);
}
inline uint32_t fmd_tile_raw_cap(const FmdRates& r)
{
    return ((fmd_tile_lp_cap(r) + 1) * 2 * r.D + 32 + 15) & ~15u;
}

// Sum over samples n in [0, n) of the additive constants left after mapping bytes to
// t = b - 128: rotate_90 (:284-296) + `as i16 - 127` (:258) give, for n mod 4 = 0..3,
//   re = t0+1, -t3, -t4, t7+1      im = t1+1, t2+1, -t5, -t6
FMD_HD constexpr int32_t fmd_const_re(int32_t n) { return 2 * (n >> 2) + ((n & 3) >= 1 ? 1 : 0); }
FMD_HD constexpr int32_t fmd_const_im(int32_t n) { return 2 * (n >> 2) + ((n & 3) < 2 ? (n & 3) : 2); }

// Byte weights (as 4 packed signed bytes, little-endian) of one aligned dword = two complex samples;
// `odd` is the parity of the dword index within the call (rotate_90 has period 8 bytes).
#define FMD_W_RE_EVEN 0xFF000001u   /* +b0 ........ -b3 */
#define FMD_W_IM_EVEN 0x00010100u   /* ... +b1 +b2 .... */
#define FMD_W_RE_ODD  0x010000FFu   /* -b4 ........ +b7 */
#define FMD_W_IM_ODD  0x00FFFF00u   /* ... -b5 -b6 .... */

// Demod::fast_atan2 (simple_fm.rs:383-405) on wrapping 32-bit integers.  The reference widens to
// i64, multiplies by 4096 and truncates back to i32 BEFORE dividing: that is a 32-bit shift.
FMD_HD int32_t fmd_fast_atan2(int32_t y, int32_t x)
{
    if (x == 0 && y == 0) return 0;
    const uint32_t ux = (uint32_t)x;
    const uint32_t uyabs = y < 0 ? 0u - (uint32_t)y : (uint32_t)y;
    int32_t num, den, base;
    if (x >= 0) { num = (int32_t)((ux - uyabs) << 12); den = (int32_t)(ux + uyabs); base = 1 << 12; }
    else        { num = (int32_t)((ux + uyabs) << 12); den = (int32_t)(uyabs - ux); base = 3 << 12; }
    const int32_t angle = (int32_t)((uint32_t)base - (uint32_t)(num / den));
    return y < 0 ? (int32_t)(0u - (uint32_t)angle) : angle;
}

// The same function with the cheap exact division; valid while |x| + |y| < 2^30 (any downsample <= 128).
FMD_HD int32_t fmd_fast_atan2_q(int32_t y, int32_t x)
{
    const uint32_t ux = (uint32_t)x;
    const uint32_t uyabs = y < 0 ? 0u - (uint32_t)y : (uint32_t)y;
    const uint32_t dif = ux - uyabs, sum = ux + uyabs;
    const bool xpos = x >= 0;
    const int32_t num = (int32_t)((xpos ? dif : sum) << 12);
    const int32_t den = xpos ? (int32_t)sum : (int32_t)(0u - dif);
    const int32_t angle = (xpos ? (1 << 12) : (3 << 12)) - fmd_sdiv_q13(num, den);
    const int32_t res = y < 0 ? -angle : angle;
    return den == 0 ? 0 : res;
}

// a * conj(b) on Complex<i32> (num-complex 0.4), wrapping.
FMD_HD void fmd_mul_conj(int32_t ar, int32_t ai, int32_t br, int32_t bi, int32_t& cr, int32_t& ci)
{
    cr = (int32_t)((uint32_t)ar * (uint32_t)br + (uint32_t)ai * (uint32_t)bi);
    ci = (int32_t)((uint32_t)ai * (uint32_t)br - (uint32_t)ar * (uint32_t)bi);
}

