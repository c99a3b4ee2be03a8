// fmd_generic_kernel.hip -- (1) the fallback demodulation kernel that derives all tile geometry on
// the device, (2) the synthetic IQ source.  gfx950 only.
//
// The generic kernel fuses every pass of Demod::demodulate (examples/simple_fm.rs:256-269) exactly
// like the production tile kernel (fmd_tile_body.h) but makes no assumption beyond
// fmd_ranges_fit32(): any downsample, any phase per channel, any tiling.  It is what runs
// when the phase-class plans do not apply (> FMD_MAX_CLASSES distinct phases in one bank, or reduced
// rates beyond the exact-small-divide range of the tile kernel), under FMD_FORCE_GENERIC=1 (experiment build), and for
// downsample 129 ... 512 (WIDE): there |lp| <= 128 * D no longer fits 16 bits, so the decimated samples stay i32 pairs,
// and the reference's own arithmetic starts to wrap -- `a * b.conj()` on Complex<i32> (:371,378) beyond downsample 255,
// `x + yabs` / `yabs - x` in fast_atan2 (:397,399) beyond 128 -- which fmd_mul_conj / fmd_fast_atan2 reproduce as the
// wrapping 32-bit operations a release build of the reference performs.
#include "fmd_device.h"
#include "fmd_kernels.h"

namespace {

using namespace fmd_dev;

// Decimated samples in LDS: packed re | im << 16 (downsample <= 128) or an i32 pair (WIDE).
template <bool WIDE> struct LpStore;
template <> struct LpStore<false> {
    uint32_t* p;
    __device__ __forceinline__ void put(int i, int re, int im) const { p[i] = pack_lp(re, im); }
    __device__ __forceinline__ void get(int i, int& re, int& im) const { const uint32_t v = p[i]; re = lp_re(v); im = lp_im(v); }
};
template <> struct LpStore<true> {
    int2* p;
    __device__ __forceinline__ void put(int i, int re, int im) const { p[i] = make_int2(re, im); }
    __device__ __forceinline__ void get(int i, int& re, int& im) const { const int2 v = p[i]; re = v.x; im = v.y; }
};

template <bool WIDE>
__global__ void __launch_bounds__(FMD_BLOCK_THREADS) fmd_demod_generic_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* const raw_w = reinterpret_cast<uint32_t*>(smem);
    LpStore<WIDE> lp;
    lp.p = reinterpret_cast<decltype(lp.p)>(smem + L.raw_cap);
    int16_t* const d16 = reinterpret_cast<int16_t*>(smem + L.raw_cap + (WIDE ? 8u : 4u) * L.lp_cap);

    const uint32_t tid = threadIdx.x;
    const uint32_t c = blockIdx.x / L.tiles;
    const uint32_t t = blockIdx.x - c * L.tiles;
    const FmdRates r = L.r;

    // ---- per-channel call geometry (wave-uniform) ---------------------------------------
    const FmdChanState st = L.st_in[c];
    const uint32_t p0 = st.prev_index, i0r = st.lpr_index_r;
    const uint32_t M = fmd_num_decimated(r.D, p0, L.ns);
    const uint32_t K = fmd_num_audio(r, i0r, M);
    const uint32_t nt = fmd_num_tiles(r, K);
    if (t >= nt) return;
    const FmdTile T = fmd_tile(r, p0, i0r, L.ns, M, K, nt, t);
    const int jfirst = T.jA - 1;                 // lp[jfirst .. jB] are needed; jfirst == -1 -> demod_pre
    const int cnt = T.jB - jfirst + 1;

    // ---- stage the tile's raw bytes into LDS: coalesced 16-byte loads --------------------
    const uint64_t gbase = (uint64_t)(uintptr_t)L.iq + (uint64_t)c * L.chan_stride;
    const uint64_t gLo = gbase + 2ull * (uint32_t)T.nLo;
    const uint64_t gHi = gbase + 2ull * (uint32_t)T.nHi;
    const uint64_t a0 = gLo & ~15ull;
    const uint32_t nchunks = (uint32_t)((gHi - a0 + 15) >> 4);
    if ((uint32_t)cnt > L.lp_cap || nchunks * 16u > L.raw_cap) {
        if (tid == 0) { atomicOr(L.err, (uint32_t)cnt > L.lp_cap ? FMD_DEVERR_LP_CAP : FMD_DEVERR_RAW_CAP); fmd_flag_report(L.hflag); }
        return;
    }
    const uint64_t gend = (uint64_t)(uintptr_t)L.iq + L.total_bytes;
    for (uint32_t i = tid; i < nchunks; i += FMD_BLOCK_THREADS) {
        const uint64_t a = a0 + 16ull * i;
        uint4 v;
        if (a + 16 <= gend) {
            v = *reinterpret_cast<const uint4*>((uintptr_t)a);
        } else {   // the array ends in the middle of this chunk (sizes are multiples of 8)
            const uint2 h = *reinterpret_cast<const uint2*>((uintptr_t)a);
            v = make_uint4(h.x, h.y, 0u, 0u);
        }
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
    const int wofs = (int)((int64_t)(gbase - a0) >> 2);      // LDS dword index of the call's dword 0
    __syncthreads();

    // ---- low_pass_complex (:337-352): one decimated sample per lane per round ----------------
    for (int i = tid; i < cnt; i += FMD_BLOCK_THREADS) {
        const int j = jfirst + i;
        int re, im;
        if (j < 0) {
            re = st.demod_pre_re; im = st.demod_pre_im;
        } else {
            lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, j), fmd_win_end(r.D, p0, j), re, im);
            if (j == 0) { re += st.lp_now_re; im += st.lp_now_im; }
        }
        lp.put(i, re, im);
    }
    __syncthreads();

    // ---- fm_demod (:355-367): polar discriminator against the predecessor --------------------
    for (int i = tid + 1; i < cnt; i += FMD_BLOCK_THREADS) {
        int ar, ai, br, bi, cr, ci;
        lp.get(i, ar, ai); lp.get(i - 1, br, bi);
        fmd_mul_conj(ar, ai, br, bi, cr, ci);
        int pcm;
        if (jfirst + i == 0) {                               // first sample of the call (:359)
            bool g;
            pcm = polar_f64(cr, ci, L.f64_guard, g);
#ifdef FMD_EXPERIMENT
            if (g) pcm += L.f64_skew;
#endif
        } else pcm = fmd_fast_atan2(ci, cr);                 // (:362)
        d16[i] = (int16_t)pcm;
    }
    __syncthreads();

    // ---- low_pass_real (:408-426): one audio sample per lane ---------------------------------
    const uint32_t nk = T.k1 - T.k0;
    int16_t* const outc = L.out + (uint64_t)c * L.out_stride;
    for (uint32_t q = tid; q < nk; q += FMD_BLOCK_THREADS) {
        const int e = (int)(T.eq + (T.er + q * r.fr) / r.sr);
        const int s = q == 0 ? T.jA : (int)(T.eq + (T.er + (q - 1) * r.fr) / r.sr) + 1;
        int sum = (T.k0 + q == 0) ? st.now_lpr : 0;
        for (int j = s; j <= e; ++j) sum += d16[j - jfirst];
        outc[T.k0 + q] = (int16_t)(sum / r.R);
    }

    // guarded f64 sample (FmdF64Exc, fmd_kernels.h): the record needs the finished group sums
    if (jfirst < 0 && tid == 0) {
        int ar, ai, br, bi, cr, ci;
        lp.get(1, ar, ai); lp.get(0, br, bi);
        fmd_mul_conj(ar, ai, br, bi, cr, ci);
        bool g;
        (void)polar_f64(cr, ci, L.f64_guard, g);
        if (g) exc_emit(exc_args(L, c, i0r, K, st.now_lpr, d16, jfirst), 0, cr, ci);
    }

    // ---- Demod state after the call (last tile only; :232-239) --------------------------------
    if (T.last && tid == 0) {
        FmdChanState ns_;
        const int s = K == 0 ? 0 : (int)fmd_audio_end(r, i0r, K - 1) + 1;
        int sum = K == 0 ? st.now_lpr : 0;
        for (int j = s; j <= T.jB; ++j) sum += d16[j - jfirst];
        ns_.now_lpr = sum;
        ns_.lpr_index_r = fmd_next_lpr_index_r(r, i0r, M, K);
        ns_.prev_index = fmd_next_prev_index(r.D, p0, L.ns);
        int tr, ti;
        lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, (int)M), (int)L.ns, tr, ti);
        if (M == 0) { tr += st.lp_now_re; ti += st.lp_now_im; }
        ns_.lp_now_re = tr; ns_.lp_now_im = ti;
        if (M == 0) { ns_.demod_pre_re = st.demod_pre_re; ns_.demod_pre_im = st.demod_pre_im; }
        else { int lr, li; lp.get(cnt - 1, lr, li); ns_.demod_pre_re = lr; ns_.demod_pre_im = li; }
        ns_.reserved = 0;
        L.st_out[c] = ns_;
        if (L.out_len) L.out_len[c] = K;
    }
}

// ---- synthetic FM source (integer only; mirrored bit-for-bit by rtl-sdr-rs_amd/synth.py) ------
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__device__ __forceinline__ int isin_q15(uint32_t phase)
{
    const int xs = (int)phase >> 16;
    const int ax = xs < 0 ? -xs : xs;
    int y = (xs * (32768 - ax)) >> 13;
    const int ay = y < 0 ? -y : y;
    const int y2 = (y * ay) >> 15;
    y = y + (((y2 - y) * 7373) >> 15);
    return y;
}

__global__ void __launch_bounds__(256) fmd_synth_kernel(const FmdSynthLaunch S)
{
    const uint64_t quads_per_chan = S.chan_stride >> 3;          // 4 complex samples = 8 bytes per thread
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= quads_per_chan * S.n_channels) return;
    const uint32_t c = (uint32_t)(gid / quads_per_chan);
    const uint64_t quad = gid - (uint64_t)c * quads_per_chan;
    const uint64_t hc = mix64(S.seed + c + 0x9E3779B97F4A7C15ull);
    const uint32_t chan_phase = (uint32_t)hc;
    const uint32_t mod_step = (uint32_t)(0x100000000ull / S.mod_period);
    const uint32_t mod_step_c = mod_step + (c % 61u) * (mod_step >> 6);
    const uint32_t beta_q16 = (uint32_t)(((uint64_t)S.dev_q32 * S.mod_period * 10430ull) >> 32);
    const int span = 2 * (int)S.noise + 1;
    uint32_t w[2];
    for (int h = 0; h < 2; ++h) {
        uint32_t word = 0;
        for (int s = 0; s < 2; ++s) {
            const uint64_t n = S.sample_offset + quad * 4 + (uint64_t)(h * 2 + s);
            const uint32_t am = (uint32_t)(n * mod_step_c) + (uint32_t)(hc >> 32);
            const uint32_t dphi = (uint32_t)((int64_t)beta_q16 * (int64_t)isin_q15(am) * 2);
            const uint32_t theta = (uint32_t)(0xC0000000ull * n) + chan_phase + dphi;
            const int I = ((int)S.amplitude * isin_q15(theta + 0x40000000u)) >> 15;
            const int Q = ((int)S.amplitude * isin_q15(theta)) >> 15;
            const uint64_t hz = mix64(hc ^ (n * 0x9E3779B97F4A7C15ull));
            const int nI = (int)(((hz & 0xFFFFull) * (uint64_t)span) >> 16) - (int)S.noise;
            const int nQ = (int)((((hz >> 16) & 0xFFFFull) * (uint64_t)span) >> 16) - (int)S.noise;
            int bI = 127 + I + nI, bQ = 127 + Q + nQ;
            bI = bI < 0 ? 0 : (bI > 255 ? 255 : bI);
            bQ = bQ < 0 ? 0 : (bQ > 255 ? 255 : bQ);
            word |= ((uint32_t)bI | ((uint32_t)bQ << 8)) << (16 * s);
        }
        w[h] = word;
    }
    *reinterpret_cast<uint2*>(S.iq + (uint64_t)c * S.chan_stride + quad * 8) = make_uint2(w[0], w[1]);
}

}  // namespace

size_t fmd_generic_lds_bytes(const FmdLaunch& L)
{
    const size_t per = L.r.D > FMD_MAX_DOWNSAMPLE ? 8u : 4u;          // i32 pairs beyond downsample 128
    return ((size_t)L.raw_cap + per * (size_t)L.lp_cap + 2u * (size_t)L.lp_cap + 15u) & ~(size_t)15u;
}

hipError_t fmd_launch_generic(const FmdLaunch& L, hipStream_t stream, FmdKernelId* used)
{
    const size_t lds = fmd_generic_lds_bytes(L);
    const uint64_t blocks = (uint64_t)L.n_channels * L.tiles;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (L.r.D > FMD_MAX_DOWNSAMPLE) hipLaunchKernelGGL(fmd_demod_generic_kernel<true>, dim3((uint32_t)blocks), dim3(FMD_BLOCK_THREADS), lds, stream, L);
    else hipLaunchKernelGGL(fmd_demod_generic_kernel<false>, dim3((uint32_t)blocks), dim3(FMD_BLOCK_THREADS), lds, stream, L);
    if (used) { used->family = FMD_KERNEL_GENERIC; used->dh = L.r.D > FMD_MAX_DOWNSAMPLE ? 1 : 0; used->fast = 0; used->kt = L.r.kt; used->lds = (uint32_t)lds; }
    return hipGetLastError();
}

hipError_t fmd_launch_synth(const FmdSynthLaunch& S, hipStream_t stream)
{
    const uint64_t threads = (S.chan_stride >> 3) * S.n_channels;
    const uint64_t blocks = (threads + 255) / 256;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmd_synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, S);
    return hipGetLastError();
}
