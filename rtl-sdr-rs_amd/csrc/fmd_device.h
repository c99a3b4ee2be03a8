// fmd_device.h -- device-side helpers shared by the gfx950 kernels (fmd_generic_kernel.hip,
// fmd_tile_kernel.hip).  Reference citations: examples/simple_fm.rs of ccostes/rtl-sdr-rs v0.3.1.
#ifndef FMD_DEVICE_H
#define FMD_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fmd_index.h"

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
static __device__ __noinline__ int polar_f64(int cr, int ci)
{
    const double angle = atan2((double)ci, (double)cr);
    return (int)(angle / kPi * 16384.0);
}

// Decimated samples travel packed: re in the low, im in the high 16 bits (|lp| <= 128 * D <= 16384).
__device__ __forceinline__ uint32_t pack_lp(int re, int im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }
__device__ __forceinline__ int lp_re(uint32_t p) { return (int)(int16_t)(p & 0xFFFFu); }
__device__ __forceinline__ int lp_im(uint32_t p) { return (int)(int16_t)(p >> 16); }

}  // namespace fmd_dev

#endif
