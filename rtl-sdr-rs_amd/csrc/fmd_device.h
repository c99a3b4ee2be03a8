// fmd_device.h -- device-side helpers shared by the gfx950 kernels (fmd_generic_kernel.hip,
// fmd_tile_kernel.hip).  Reference citations: examples/simple_fm.rs of ccostes/rtl-sdr-rs v0.3.1.
#ifndef FMD_DEVICE_H
#define FMD_DEVICE_H

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
// Otherwise `*guarded` reports whether the value lies within `guard` of an integer (see FmdF64Exc, fmd_kernels.h).
static __device__ __noinline__ int polar_f64(int cr, int ci, double guard, bool* guarded)
{
    *guarded = false;
    if (ci == 0) return cr >= 0 ? 0 : 16384;                 // atan2(+-0, x): 0 or pi (0, 0 -> 0)
    if (cr == 0) return ci > 0 ? 8192 : -8192;               // +-pi/2
    if (cr == ci) return cr > 0 ? 4096 : -12288;             // pi/4, -3pi/4
    if (cr == -ci) return cr > 0 ? -4096 : 12288;            // -pi/4, 3pi/4
    const double angle = atan2((double)ci, (double)cr);
    const double v = angle / kPi * 16384.0;
    *guarded = fabs(v - rint(v)) < guard;
    return (int)v;
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
};

static __device__ __noinline__ void exc_emit(FmdExcArgs a, int j, int cr, int ci)
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
}

__device__ __forceinline__ FmdExcArgs exc_args(const FmdLaunch& L, uint32_t c, uint32_t i0r, uint32_t K, int now_lpr_in,
                                               const int16_t* d16, int jfirst)
{
    FmdExcArgs a;
    a.sr = L.r.sr; a.fr = L.r.fr; a.i0r = i0r; a.K = K; a.c = c; a.seq = L.seq;
    a.now_lpr_in = now_lpr_in; a.jfirst = jfirst; a.d16 = d16;
    a.out_c = L.out + (uint64_t)c * L.out_stride; a.exc = L.exc;
    return a;
}

// Decimated samples travel packed: re in the low, im in the high 16 bits (|lp| <= 128 * D <= 16384).
__device__ __forceinline__ uint32_t pack_lp(int re, int im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }
__device__ __forceinline__ int lp_re(uint32_t p) { return (int)(int16_t)(p & 0xFFFFu); }
__device__ __forceinline__ int lp_im(uint32_t p) { return (int)(int16_t)(p >> 16); }

}  // namespace fmd_dev

#endif
