// fmd_device.h -- device-side helpers shared by the gfx950 kernels (fmd_generic_kernel.hip,
// fmd_tile_body.h, fmd_firdemod.hip).  Reference citations: examples/simple_fm.rs of ccostes/rtl-sdr-rs v0.3.1.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fmd_index.h"
#include "fmd_kernels.h"

namespace fmd_dev {

constexpr double kPi = 3.14159265358979323846264338327950288;   // std::f64::consts::PI (:17)

__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);   // v_dot4_i32_i8
}

// The FIRST dot product of an accumulation whose start value is loop-invariant (a bias, a per-phase constant).  For the
// plain intrinsic hipcc selects the two-address v_dot4c_i32_i8 and copies the start value into the accumulator first: one
// v_mov per sum per round.  With the clamp bit set it has to take the three-address VOP3P v_dot4_i32_i8, which reads the
// start value as a source; the clamp (saturation at the i32 range) never engages: |sum| <= 128 * 4 on top of a start value
// below 2^31 - 2^23.  (An inline-asm v_dot4_i32_i8 does the same but hides the dot -> VALU wait states from hipcc's hazard
// pass: wrong results at downsample 2, where the consumer follows directly.)
__device__ __forceinline__ int sdot4_init(uint32_t a, uint32_t w, int init)
{
    return __builtin_amdgcn_sdot4((int)a, (int)w, init, true);
}

// Sum of the rotated + centred complex samples n in [n0, n1) of this channel-call (rotate_90 :276-299,
// `as i16 - 127` :258, buf_to_complex :441-450, folded into signed-byte dot products), read from the
// LDS image of the raw bytes.  `wofs`: LDS dword index of the call's dword 0 (may be negative).
// Any alignment, any length: the slow, always-correct form.
__device__ __forceinline__ void lds_window_sum(const uint32_t* __restrict__ raw_w, int wofs, int n0, int n1,
                                               int& re, int& im)
{
    int ar = 0, ai = 0;
    const int m0 = n0 >> 1, m1 = (n1 - 1) >> 1;
    for (int m = m0; m <= m1; ++m) {
        const uint32_t w = raw_w[wofs + m] ^ 0x80808080u;     // u8 -> s8 (b - 128)
        uint32_t mask = 0xFFFFFFFFu;
        if (2 * m < n0) mask = 0xFFFF0000u;                   // window starts at the dword's 2nd sample
        if (2 * m + 1 >= n1) mask &= 0x0000FFFFu;             // window ends after the dword's 1st sample
        const bool odd = m & 1;
        ar = sdot4(w, (odd ? FMD_W_RE_ODD : FMD_W_RE_EVEN) & mask, ar);
        ai = sdot4(w, (odd ? FMD_W_IM_ODD : FMD_W_IM_EVEN) & mask, ai);
    }
    if (n1 > n0) {
        ar += fmd_const_re(n1) - fmd_const_re(n0);
        ai += fmd_const_im(n1) - fmd_const_im(n0);
    }
    re = ar; im = ai;
}

// Demod::polar_discriminant (:370-374) on the already-formed product c = a * conj(b): the one f64
// atan2 sample of every call (:359).  Kept out of line: it runs on one lane per channel-call.
// Exact directions first (integer decisions; libm and ocml both return the exact f64 there, so the reference's
// values are 0, +-4096, +-8192, +-12288, 16384 -- tests/test_oracle_kat.py pins that table against the host libm).
// Otherwise `guarded` reports whether the value lies within `guard` of an integer (see FmdF64Exc, fmd_kernels.h).
// The out-of-line part returns value and flag in ONE register (|value| <= 16384; bit 30 = guarded): a pointer
// out-parameter gave every kernel that calls it a 16-byte scratch frame per lane (private-segment setup on every wave
// of every launch for a path one lane per channel-call takes).
static __device__ __noinline__ int polar_f64_packed(int cr, int ci, double guard)
{
    if (ci == 0) return (cr >= 0 ? 0 : 16384) & 0x3FFFFFFF;                 // atan2(+-0, x): 0 or pi (0, 0 -> 0)
    if (cr == 0) return (ci > 0 ? 8192 : -8192) & 0x3FFFFFFF;               // +-pi/2
    if (cr == ci) return (cr > 0 ? 4096 : -12288) & 0x3FFFFFFF;             // pi/4, -3pi/4
    if (cr == -ci) return (cr > 0 ? -4096 : 12288) & 0x3FFFFFFF;            // -pi/4, 3pi/4
    const double angle = atan2((double)ci, (double)cr);
    const double v = angle / kPi * 16384.0;
    return ((int)v & 0x3FFFFFFF) | (fabs(v - rint(v)) < guard ? 0x40000000 : 0);
}

__device__ __forceinline__ int polar_f64(int cr, int ci, double guard, bool& guarded)
{
    const int r = polar_f64_packed(cr, ci, guard);
    guarded = (r & 0x40000000) != 0;
    return (int)((uint32_t)r << 2) >> 2;                     // sign-extend the 30-bit value
}

// Append the record of one guarded sample (one lane, after the tile's d16[] is complete: d16[j - jfirst] holds
// discriminator sample j for every j of the audio groups this tile owns).  (i0r, K): resampler phase / audio
// count of this channel-call.  Everything travels BY VALUE: a noinline callee that took the launch descriptor by
// reference would force the whole kernel-argument block into scratch memory on every lane (measured: 15x slower).
struct FmdExcArgs {
    uint32_t sr, fr, i0r, K, c, seq;
    int now_lpr_in, jfirst;
    const int16_t* d16;
    int16_t* out_c;          // &out[c][0]
    FmdExcBuf* exc;
    uint32_t* hflag;         // FmdLaunch::hflag
};

// "This launch has written into its report buffer": the host-mapped word fmd_demod_check reads (rare paths only).
__device__ __forceinline__ void fmd_flag_report(uint32_t* hflag)
{
    if (hflag) __hip_atomic_store(hflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (a plain store: every writer writes 1, the host clears it when the slot is idle)
}

[[maybe_unused]] static __device__ __noinline__ void exc_emit(FmdExcArgs a, int j, int cr, int ci)
{
    FmdF64Exc e{};
    e.channel = a.c; e.cr = cr; e.ci = ci;
    e.d_gpu = a.d16[j - a.jfirst];
    e.seq = a.seq;
    const uint32_t k = ((uint32_t)j * a.sr + a.i0r) / a.fr;          // the audio sample whose group contains j
    if (k < a.K) {                                                    // e(k) = ((k + 1) * fr - i0r - 1) / sr (fmd_audio_end)
        const int hi = (int)(((k + 1) * a.fr - a.i0r - 1) / a.sr), lo = k == 0 ? 0 : (int)((k * a.fr - a.i0r - 1) / a.sr) + 1;
        int sum = k == 0 ? a.now_lpr_in : 0;
        for (int jj = lo; jj <= hi; ++jj) sum += a.d16[jj - a.jfirst];
        e.k = (int)k; e.sum = sum;
        e.out_elem = (uint64_t)(uintptr_t)(a.out_c + k);
    } else {
        e.k = -1;                                                     // lies in the partial group carried in now_lpr
    }
    atomicAdd(&a.exc->guarded_total, 1u);
    const uint32_t slot = atomicAdd(&a.exc->count, 1u);
    if (slot < FMD_EXC_CAP) a.exc->rec[slot] = e; else atomicOr(&a.exc->err, FMD_DEVERR_EXC_CAP);
    fmd_flag_report(a.hflag);
}

__device__ __forceinline__ FmdExcArgs exc_args(const FmdLaunch& L, uint32_t c, uint32_t i0r, uint32_t K, int now_lpr_in,
                                               const int16_t* d16, int jfirst)
{
    FmdExcArgs a;
    a.sr = L.r.sr; a.fr = L.r.fr; a.i0r = i0r; a.K = K; a.c = c; a.seq = L.seq;
    a.now_lpr_in = now_lpr_in; a.jfirst = jfirst; a.d16 = d16;
    a.out_c = L.out + (uint64_t)c * L.out_stride; a.exc = L.exc; a.hflag = L.hflag;
    return a;
}

typedef short fmd_s2 __attribute__((ext_vector_type(2)));

// Two things tried around these helpers and withdrawn (both caught by tests/test_gpu_fuzz.py, downsample 2):
//  * the dot products as inline-asm VOP3P forms (no v_mov of the addend into a v_dot*c destination, -0.5..1.2 %):
//    on gfx950 a non-dot VALU instruction must not read a dot result for 3 wait states; hipcc inserts the
//    s_nops for its own builtins but cannot see through inline asm, so correctness hung on instruction order;
//  * bound_ctrl on the DPP moves (saves the v_mov 0 of `old`): the DPP-combine pass may then fold the move
//    into a consumer that runs under the narrower EXEC mask of the predicated stores, where a masked-off
//    neighbour lane reads as 0.
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(fmd_s2, a), __builtin_bit_cast(fmd_s2, b), 0, false);
}

// Demod::polar_discriminant_fast (:377-380) + fast_atan2 (:383-405), branch-free and without a single select, for packed
// operands (re | im << 16, components fit i16): the integer form, for downsample > 16 (and the fused FIR kernel when its
// filter gain exceeds the f32 form's range).  hipcc turns `?:` into VCC-masked v_cndmask_b32_e32, which issues ~4x slower
// than the SGPR-masked e64 form (tools/valubench.hip), hence sign-mask arithmetic throughout.
// Division: |quotient| <= 4097, so an f32 estimate is within 1 and one exact (wrapping) remainder fixes it; valid while
// |x| + |y| < 2^30, which holds for every downsample <= 128 (|lp| <= 128 * D).  Same results as fmd_fast_atan2 (tested).
// With mx / my the sign masks of x / y:  den = |x| + |y| in both branches of :390-400, the numerator is
// +-(|x| - |y|) with the sign of x, and the base angle is pi/4 + (x < 0 ? pi/2 : 0).
__device__ __forceinline__ int disc_nosel(uint32_t a, uint32_t b)
{
    const uint32_t a_sw = __builtin_amdgcn_alignbit(a, a, 16);          // (im, re)
    const uint32_t b_cj = (b & 0xFFFFu) | ((0u - (b >> 16)) << 16);      // (re, -im)
    const int cr = sdot2(a, b);                                          // ar*br + ai*bi
    const int ci = sdot2(a_sw, b_cj);                                    // ai*br - ar*bi
    const uint32_t mx = (uint32_t)(cr >> 31), my = (uint32_t)(ci >> 31);
    const uint32_t xabs = ((uint32_t)cr ^ mx) - mx, yabs = ((uint32_t)ci ^ my) - my;
    const uint32_t den = xabs + yabs, t = xabs - yabs;
    const int num = (int)(((t ^ mx) - mx) << 12);                        // the i64 product truncated to i32 (:397,399)
    const uint32_t mn = (uint32_t)(num >> 31);
    const uint32_t unum = ((uint32_t)num ^ mn) - mn;
    uint32_t q = (uint32_t)((float)unum * __builtin_amdgcn_rcpf((float)den));
    const int rem = (int)(unum - q * den);
    q += (uint32_t)(rem >> 31);                                          // estimate one too large
    q -= (uint32_t)(((int)den - 1 - rem) >> 31);                         // estimate one too small (rem >= den)
    const uint32_t qs = (q ^ mn) - mn;                                   // truncating signed quotient
    const uint32_t angle = (1u << 12) + (mx & (2u << 12)) - qs;
    const uint32_t res = (angle ^ my) - my;
    return (int)(res & (uint32_t)((int)(0u - den) >> 31));               // x == 0 && y == 0 -> 0 (:388)
}

// The same function in f32, for downsample <= 16 (FMD_DISC_F32_MAX_D): there |x|, |y| <= 2 (128 D)^2 <= 2^23 and
// |x| + |y| < 2^24, so x, y, their sum and difference are integers that f32 holds exactly.  Why: on gfx950 only add /
// sub / and / or / xor / shift-right and f32 add / sub / mul (with their free abs / neg / clamp modifiers) issue in 2
// cycles per wave; selects, compares, conversions, integer multiplies, max ... take 4 (tools/valubench.hip).  This form
// is ~100 cycles against ~124 for the integer one (-10 % on the whole launch at the reference's own rates).
//   den = |x| + |y|;  s = x >= 0 ? x - |y| : x + |y| = +-(|x| - |y|)                          (:390-400)
//   `(4096_i64 * s) as i32` keeps s mod 2^20 in [-2^19, 2^19): adding 1.5 * 2^43 (ulp 2^20) rounds s + 0.5 to a
//     multiple of 2^20 -- never a tie, s is an integer -- and subtracting it again leaves k * 2^20, k = floor(s / 2^20 + 1/2).
//     THIS step sets the limit: s + 0.5 must be representable, i.e. |s| <= 2^23 (downsample 18 fails here: an odd
//     multiple of 2^19 above 2^23 loses its + 0.5 and the big add becomes a tie)
//   q = floor(Q), Q = 4096 |sp| / den <= 4096: m = 4096 |sp| is exact (<= 2^31, a power-of-two scale); m * rcp(den) is within
//     2^-10 of Q (rcp's ulp + one rounding, 2^-22 relative), so its NEAREST integer k is floor(Q) or floor(Q) + 1, and
//     which one shows in the sign of the integer k * den - m (in (-den, 0] for floor(Q), in [1, den] for one too
//     large): ONE fma with the clamp modifier returns exactly that 0 / 1 (the product is exact inside the fma; a result
//     rounded above 2^24 keeps its sign and stays >= 1), and q = k - it.  (Rounds 1 - 3 biased the estimate low, took
//     floor() and compared the remainder with den: three instructions more.)
//   the sign of the truncating quotient is sp's, the base angle is 8192 - (+-4096) by the sign of x, the result takes
//     y's sign; (0, 0) -> den = 0 -> the final factor clamp(den + den) is 0 (:388), 1 otherwise.
// tests/test_disc_f32_model.py replays this sequence in numpy f32 with the reciprocal pushed to both ends of its
// 1-ulp band against fast_atan2 itself, for boxcar sums up to 128 * 16 and at 128 * 18 to show the limit; the GPU
// parity and fuzz tests run it on the hardware.
#define FMD_DISC_F32_MAX_D 16
// f32 -> i32 with the hardware's own rule for NaN: v_cvt_i32_f32 returns 0 for NaN (and saturates), which disc_f32_xy
// relies on for fast_atan2's `(0, 0) -> 0`.  A C++ `(int)v` of NaN is undefined behaviour (fptosi poison in LLVM: right
// only while the compiler happens not to fold it) and the saturating intrinsic expands to ten instructions on gfx950, so
// the instruction is named outright.  It is an ordinary one-cycle-class VALU conversion of a value produced by plain VALU
// code: none of the gfx950 hazards that made inline asm unsafe around the dot products (see below) applies to it.
// tests: test_near_silence, the silence cases of tests/test_gpu_fuzz.py, test_axis_aligned_full_scale.
__device__ __forceinline__ int fmd_cvt_i32_nan0(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    return v != v ? 0 : (int)v;                           // host pass: never executed, kept well-defined
#endif
}
__device__ __forceinline__ float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }   // folds into a clamp modifier
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// fast_atan2 (:383-405) of the exact f32 product (xf, yf) = a * conj(b); see above.
// NOWRAP: the caller guarantees |s| < 2^19 (downsample <= 3: 2 (128 * 3)^2 = 294 912), where `(4096 * s) as i32` cannot wrap.
// k4096: 4096.0f; straight-line callers (the register-streaming rounds) pass it in a register they keep for the whole
// function -- hipcc otherwise re-materialises it with a v_mov in front of every v_bfi.
// WRAP1: downsample 4 exactly.  There |lp| <= 512, so s = +-(|x| - |y|) lies in [-2^19, 2^19] and `(4096 * s) as i32` wraps for ONE value,
// s = +2^19 (a = b = (512, 512): the saturated constant input), which becomes -2^19; -2^19 * 4096 = -2^31 still fits.  The general form
// below spends four adds on the wrap; here clamp(s - (2^19 - 1)) is exactly 1 at that value and 0 everywhere else (s is an integer), and
// one fma takes 2^20 off: two instructions (tests/test_disc_f32_model.py::test_single_point_wrap_at_downsample_4,
// tests/test_gpu_parity.py::test_diagonal_full_scale).
template <bool NOWRAP = false, bool LO16 = false, bool WRAP1 = false>
__device__ __forceinline__ int disc_f32_xy(float xf, float yf, float k4096 = 4096.0f)
{
    const float den = __builtin_fabsf(xf) + __builtin_fabsf(yf);
    const float t = __builtin_fabsf(xf) - __builtin_fabsf(yf);
    const uint32_t sx = f2u(xf) & 0x80000000u;
    const float s = u2f(f2u(t) ^ sx);
    const float big = 13194139533312.0f;                                 // 1.5 * 2^43
    const float sp = NOWRAP ? s : WRAP1 ? __builtin_fmaf(clamp01(s - 524287.0f), -1048576.0f, s)
                                        : s - (((s + 0.5f) + big) - big);  // s mod 2^20, signed
    // (0, 0): den = 0 -> rcp = inf, m * inf = 0 * inf = NaN, and NaN runs through the rounding / the subtract (whatever the
    // clamped fma makes of it) / the sign xors to the final conversion, where v_cvt_i32_f32 turns it into 0 -- exactly
    // fast_atan2's `(0, 0) -> 0` (:388).  Three instructions fewer than guarding the reciprocal and multiplying by a
    // 0 / 1 factor (tests: test_near_silence, test_gpu_fuzz silence cases, tests/test_disc_f32_model.py for den >= 1).
    const float m = __builtin_fabsf(sp) * 4096.0f;
    const float k = __builtin_rintf(m * __builtin_amdgcn_rcpf(den));
    const float q = k - clamp01(__builtin_fmaf(k, den, -m));
    const float qs = u2f(f2u(q) ^ (f2u(sp) & 0x80000000u));
    const float base = 8192.0f - u2f(f2u(k4096) ^ sx);                   // 4096 or 12288 (:395,400)
    const float res = u2f(f2u(base - qs) ^ (f2u(yf) & 0x80000000u));
    // LO16: the caller stores the result as i16 and nothing else: adding 1.5 * 2^23 leaves the integer in the low mantissa
    // bits (two's complement, |res| <= 16384) -- a 2-cycle add where the conversion takes 4.  The NaN of (0, 0) keeps the
    // payload the hardware gave it (0x..C00000: low 16 bits zero) through the add, so it still stores 0.
    if constexpr (LO16) return (int)f2u(res + 12582912.0f);
    return fmd_cvt_i32_nan0(res);
}

// int -> f32 without v_cvt_f32_i32 (4 issue cycles): a sum accumulated ON TOP of the bit pattern of 1.5 * 2^23 -- the bias
// rides in the accumulator's initial value, a v_mov either way -- reads as the f32 12582912 + v exactly for |v| < 2^22, and
// one f32 subtract (2 cycles) takes the bias off.  Used for the window sums of the adjacent-window rounds (|lp| <= 128 * 14)
// and, with BIAS, for the complex product of packed samples up to downsample 11 (|c| <= 2 (128 * 11)^2 < 2^22).
constexpr int kSumBias = 0x4B400000;
__device__ __forceinline__ float sum_to_f32(int biased) { return u2f((uint32_t)biased) - 12582912.0f; }

template <bool BIAS = false, bool NOWRAP = false, bool LO16 = false>
__device__ __forceinline__ int disc_f32(uint32_t a, uint32_t b)
{
    const uint32_t a_sw = __builtin_amdgcn_alignbit(a, a, 16);          // (im, re)
    const uint32_t b_cj = (b & 0xFFFFu) | ((0u - (b >> 16)) << 16);      // (re, -im)
    if constexpr (BIAS) {                                                // c = a * conj(b), exact
        // (clamp bit set: the three-address v_dot2_i32_i16 with the bias as a source, no v_mov of it first -- see sdot4_init)
        const int cr = __builtin_amdgcn_sdot2(__builtin_bit_cast(fmd_s2, a), __builtin_bit_cast(fmd_s2, b), kSumBias, true);
        const int ci = __builtin_amdgcn_sdot2(__builtin_bit_cast(fmd_s2, a_sw), __builtin_bit_cast(fmd_s2, b_cj), kSumBias, true);
        return disc_f32_xy<NOWRAP, LO16>(sum_to_f32(cr), sum_to_f32(ci));
    }
    return disc_f32_xy<NOWRAP, LO16>((float)sdot2(a, b), (float)sdot2(a_sw, b_cj));
}

// The same with the samples' components already in f32 (exact integers): c = a * conj(b) by four fmas -- every product
// is below 2^22 and every sum below 2^23 for downsample <= 16, so nothing rounds -- instead of pack, swap, conjugate, two
// dot products and two conversions.
template <bool NOWRAP = false, bool LO16 = false, bool WRAP1 = false>
__device__ __forceinline__ int disc_f32_c(float ar, float ai, float br, float bi, float k4096 = 4096.0f)
{
    // A product such as 0 * -5 is -0, and (-0) + (-0) stays -0; fast_atan2 takes its signs from x < 0 / y < 0, where zero
    // is not negative, and disc_f32_xy reads sign BITS (tests/test_gpu_parity.py::test_near_silence)
    // (x too: with x = -0 the sign-bit form takes fast_atan2's "x < 0" branch, which agrees with the "x >= 0" one at x = 0 only
    //  while `(4096 * s) as i32` does not wrap: (x, y) = (-0, 2^19) -- a = (0, -768), b = (-768, 0) at downsample 6 -- would
    //  come out as 16384 instead of 8192.)
    // So the first product of each component is an fma onto +0.0: (-0) + (+0) = +0, the second fma then adds to a value
    // that is never -0, and an exact cancellation gives +0 in round-to-nearest.  Four instructions; rounds 2 - 3 had
    // mul, fma, mul, fma and `+ 0.0f` twice (session r04r: -0.3 % at downsample 6, -1.3 % at 4, -1.6 % at 2).
    // (Round 4 also measured two v_pk_fma_f32 -- t = (ar br + 0, -ar bi + 0), c = (ai bi, ai br) + t: bit-exact, equal to
    //  1 % slower than the six-instruction form: not adopted.)
    const float xf = __builtin_fmaf(ai, bi, __builtin_fmaf(ar, br, 0.0f));       // ar*br + ai*bi
    const float yf = __builtin_fmaf(ai, br, __builtin_fmaf(-ar, bi, 0.0f));      // ai*br - ar*bi
    return disc_f32_xy<NOWRAP, LO16, WRAP1>(xf, yf, k4096);
}

// Decimated samples travel packed: re in the low, im in the high 16 bits (|lp| <= 128 * D <= 16384).
__device__ __forceinline__ uint32_t pack_lp(int re, int im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }
__device__ __forceinline__ int lp_re(uint32_t p) { return (int)(int16_t)(p & 0xFFFFu); }
__device__ __forceinline__ int lp_im(uint32_t p) { return (int)(int16_t)(p >> 16); }

}  // namespace fmd_dev

