// fmd_tile_kernel.hip -- the production demodulation kernel for gfx950 (CDNA4, wave64).
//
// One launch fuses every pass of Demod::demodulate (examples/simple_fm.rs:256-269):
//   rotate_90 (:276-299) + `as i16 - 127` (:258) + buf_to_complex (:441-450)
//     -> signed-byte dot products (v_dot4_i32_i8) straight off the raw u8 stream
//   low_pass_complex (:337-352)  -> per-lane window sums out of an LDS-staged tile
//   fm_demod / fast_atan2 (:355-405) incl. the one f64 atan2 sample per call (:359,370-374)
//   low_pass_real (:408-426)     -> per-lane group sums over the tile's discriminator samples
// The reference's intermediate vectors (512 KiB + 1 MiB + ... per 256 KiB call) never exist: HBM
// traffic is the u8 input once (+ a < 1 % tile halo) and the s16 output.  Memory-bound integer
// streaming: no MFMA.
//
// grid = (tiles, channels); a tile is `kt` consecutive audio samples of one channel-call.
//   * tile geometry: host-made per-phase-class plans (FmdClassPlan, fmd_index.h) -> multiply-adds;
//   * staging: LDS-DMA (global_load_lds_dwordx4), 1 KiB per wave-instruction, every load of the tile in
//     flight before the single wait; no VGPR round trip, no ds_write pass;
//   * boxcar: for an even downsample a window is DH whole dwords, 3 VALU ops per dword (xor, 2 x dot4)
//     with per-lane weight registers that already carry the rotation sign of the dword parity;
//   * predecessor sample from the neighbouring lane (DPP wave_shr:1): a wave-round is 63 new windows
//     + 1 overlap, so there is no LDS exchange and no barrier between boxcar and discriminator;
//   * discriminator: complex multiply by 2 x v_dot2_i32_i16 on packed (re, im); branch-free
//     fast_atan2 with an exact f32-reciprocal divide;
//   * resampler: one audio sample per lane from the tile's discriminator samples in LDS.
#include "fmd_device.h"
#include "fmd_kernels.h"

namespace {

using namespace fmd_dev;

typedef short fmd_s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(fmd_s2, a), __builtin_bit_cast(fmd_s2, b), 0, false);
}

// Demod::polar_discriminant_fast (:377-380) + fast_atan2 (:383-405), branch-free, for packed operands
// (re | im << 16, components fit i16).  c = a * conj(b) is returned for the f64 sample.
// Division: |quotient| <= 4097, so an f32 estimate is within 1 and one exact (wrapping) remainder fixes
// it; valid while x + |y| < 2^30, i.e. downsample <= 64.  Same results as fmd_fast_atan2 (tested).
__device__ __forceinline__ int disc_fast(uint32_t a, uint32_t b, int& cr, int& ci)
{
    const uint32_t a_sw = __builtin_amdgcn_alignbit(a, a, 16);          // (im, re)
    const uint32_t b_cj = (b & 0xFFFFu) | ((0u - (b >> 16)) << 16);      // (re, -im)
    cr = sdot2(a, b);                                                    // ar*br + ai*bi
    ci = sdot2(a_sw, b_cj);                                              // ai*br - ar*bi
    const uint32_t ux = (uint32_t)cr;
    const uint32_t yabs = (uint32_t)(ci < 0 ? -ci : ci);
    const uint32_t dif = ux - yabs, sum = ux + yabs;
    const bool xpos = cr >= 0;
    const int num = (int)((xpos ? dif : sum) << 12);                     // the i64 product truncated to i32 (:397,399)
    const uint32_t den = xpos ? sum : yabs - ux;
    const uint32_t unum = num < 0 ? 0u - (uint32_t)num : (uint32_t)num;
    uint32_t q = (uint32_t)((float)unum * __builtin_amdgcn_rcpf((float)den));
    const int rem = (int)(unum - q * den);
    q = q + (rem >= (int)den ? 1u : 0u) - (rem < 0 ? 1u : 0u);
    const int qs = num < 0 ? -(int)q : (int)q;                           // truncating signed quotient
    int angle = (xpos ? (1 << 12) : (3 << 12)) - qs;
    angle = ci < 0 ? -angle : angle;
    return den == 0u ? 0 : angle;                                        // x == 0 && y == 0 (:388)
}

__device__ __forceinline__ void lds_dma16(const unsigned char* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int DH>
__global__ void __launch_bounds__(FMD_BLOCK_THREADS) fmd_demod_tile_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t glen = L.fa + 1u;           // a resampler group spans fa or fa+1 discriminator samples
    const uint32_t d16_bytes = (2u * (L.lp_cap + glen + 1u) + 15u) & ~15u;
    uint32_t* const raw_w = reinterpret_cast<uint32_t*>(smem);
    int16_t* const d16 = reinterpret_cast<int16_t*>(smem + L.raw_cap);
    uint32_t* const last_lp = reinterpret_cast<uint32_t*>(smem + L.raw_cap + d16_bytes);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t t = blockIdx.x;
    const uint32_t c = blockIdx.z * 65535u + blockIdx.y;
    if (c >= L.n_channels) return;
    const FmdRates r = L.r;
    const uint32_t cls = L.chan_class ? L.chan_class[c] : 0u;
    const FmdClassPlan P = L.cls[cls];
    if (t >= P.nt) return;
    const FmdTile T = fmd_tile_fast(r, P, L.Qt, L.ns, t);
    const uint32_t p0 = P.p0;
    const int jfirst = T.jA - 1;               // lp[jfirst .. jB] are needed; jfirst == -1 is demod_pre
    const int cnt = T.jB - jfirst + 1;

    // ---- stage the tile's raw bytes: LDS-DMA, all loads in flight, one wait ---------------------
    const uint64_t gbase = (uint64_t)(uintptr_t)L.iq + (uint64_t)c * L.chan_stride;
    const uint64_t gLo = gbase + 2ull * (uint32_t)T.nLo;
    const uint64_t gHi = gbase + 2ull * (uint32_t)T.nHi;
    const uint64_t a0 = gLo & ~15ull;
    const uint32_t nchunks = (uint32_t)((gHi - a0 + 15) >> 4);
    if ((uint32_t)cnt > L.lp_cap || nchunks * 16u > L.raw_cap) {
        if (tid == 0) atomicOr(L.err, (uint32_t)cnt > L.lp_cap ? FMD_DEVERR_LP_CAP : FMD_DEVERR_RAW_CAP);
        return;
    }
    const uint64_t gend = (uint64_t)(uintptr_t)L.iq + L.total_bytes;
    if (a0 + 16ull * nchunks <= gend) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>((uintptr_t)a0) + 16u * tid;
        unsigned char* dst = smem + 1024u * wave;          // wave-uniform; the hardware adds lane * 16
        const uint32_t nfull = nchunks >> 8, ntail = nchunks & 255u;
        for (uint32_t l = 0; l < nfull; ++l) lds_dma16(src + 4096u * l, dst + 4096u * l);
        if (tid < ntail) lds_dma16(src + 4096u * nfull, dst + 4096u * nfull);
    } else {   // the whole array ends inside this tile's last chunk (sizes are multiples of 8)
        for (uint32_t i = tid; i < nchunks; i += FMD_BLOCK_THREADS) {
            const uint64_t a = a0 + 16ull * i;
            uint4 v;
            if (a + 16 <= gend) v = *reinterpret_cast<const uint4*>((uintptr_t)a);
            else { const uint2 h = *reinterpret_cast<const uint2*>((uintptr_t)a); v = make_uint4(h.x, h.y, 0u, 0u); }
            reinterpret_cast<uint4*>(smem)[i] = v;
        }
    }
    const int wofs = (int)((int64_t)(gbase - a0) >> 2);      // LDS dword index of the call's dword 0
    const bool fastwin = DH > 0 && (p0 & 1u) == 0u;          // windows are DH whole dwords
    const FmdChanState st = L.st_in[c];

    // Lane-constant weights of the fast window.  The window of decimated sample j starts at call dword
    // m0 = DH*j - p0/2; rotate_90's sign pattern has period 2 dwords and a wave-round advances j by an
    // even number, so the parity of m0 -- hence the weights -- is fixed per lane for the whole tile.
    constexpr int NW = FMD_BLOCK_THREADS / 64;
    const uint32_t hp = p0 >> 1;
    const int j0 = jfirst + (int)wave * 63 + (int)lane;
    const bool odd = ((((DH & 1) ? ((uint32_t)j0 ^ hp) : hp)) & 1u) != 0u;
    const uint32_t wreA = odd ? FMD_W_RE_ODD : FMD_W_RE_EVEN, wreB = odd ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
    const uint32_t wimA = odd ? FMD_W_IM_ODD : FMD_W_IM_EVEN, wimB = odd ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
    const int im0 = 2 * (odd ? DH / 2 : (DH + 1) / 2);       // +2 per call-even dword; re gets +1 per dword
    __syncthreads();

    // ---- boxcar + discriminator, 63 new decimated samples per wave-round ------------------------
    for (int base = (int)wave * 63; base < cnt; base += NW * 63) {
        const int i = base + (int)lane;                      // lane 0 re-does the previous round's last window
        const bool act = i < cnt;
        const int j = jfirst + i;
        int re = 0, im = 0;
        if (fastwin) {
            if (act) {
                const int jj = j < 1 ? 1 : j;
                const uint32_t* __restrict__ p = raw_w + (wofs + DH * jj - (int)hp);
                re = DH; im = im0;
#pragma unroll
                for (int u = 0; u < (DH > 0 ? DH : 1); ++u) {
                    const uint32_t w = p[u] ^ 0x80808080u;   // u8 -> s8 (b - 128)
                    re = sdot4(w, (u & 1) ? wreB : wreA, re);
                    im = sdot4(w, (u & 1) ? wimB : wimA, im);
                }
            }
            if (jfirst <= 0 && base == 0 && act && j <= 0) { // call start: demod_pre / the clipped first window
                if (j < 0) { re = st.demod_pre_re; im = st.demod_pre_im; }
                else {
                    lds_window_sum(raw_w, wofs, 0, fmd_win_end(r.D, p0, 0), re, im);
                    re += st.lp_now_re; im += st.lp_now_im;
                }
            }
        } else if (act) {
            if (j < 0) { re = st.demod_pre_re; im = st.demod_pre_im; }
            else {
                lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, j), fmd_win_end(r.D, p0, j), re, im);
                if (j == 0) { re += st.lp_now_re; im += st.lp_now_im; }
            }
        }
        const uint32_t pk = pack_lp(re, im);
        const uint32_t prev = wave_shr1(pk);
        if (act && lane > 0) {
            int cr, cim;
            int pcm = disc_fast(pk, prev, cr, cim);                              // (:362)
            if (jfirst < 0 && base == 0 && j == 0) pcm = polar_f64(cr, cim);     // first sample of the call (:359)
            d16[i] = (int16_t)pcm;
        }
        if (T.last && act && j == T.jB) last_lp[0] = pk;
    }
    __syncthreads();

    // ---- low_pass_real: one audio sample per lane -------------------------------------------------
    const uint32_t nk = T.k1 - T.k0;
    int16_t* const outc = L.out + (uint64_t)c * L.out_stride;
    for (uint32_t q = tid; q < nk; q += FMD_BLOCK_THREADS) {
        const int e = (int)(T.eq + q * L.fa + fmd_udiv_small(T.er + q * L.fb, r.sr, L.inv_sr));
        const int s = q == 0 ? T.jA
                             : (int)(T.eq + (q - 1) * L.fa + fmd_udiv_small(T.er + (q - 1) * L.fb, r.sr, L.inv_sr)) + 1;
        int sum = (T.k0 + q == 0) ? st.now_lpr : 0;
        const int16_t* dp = d16 + (s - jfirst);
        const int n = e - s + 1;                             // <= glen
#pragma clang loop vectorize(disable)
        for (int u = 0; u < (int)glen; ++u) { const int v = dp[u]; sum += u < n ? v : 0; }
        outc[T.k0 + q] = (int16_t)fmd_sdiv_small(sum, r.R, L.inv_R);
    }

    // ---- Demod state after the call (last tile only; :232-239) -------------------------------------
    if (T.last && tid == 0) {
        FmdChanState ns_;
        const int s = P.K == 0 ? 0 : (int)fmd_audio_end(r, P.i0r, P.K - 1) + 1;
        int sum = P.K == 0 ? st.now_lpr : 0;
        for (int jj = s; jj <= T.jB; ++jj) sum += d16[jj - jfirst];
        ns_.now_lpr = sum;
        ns_.lpr_index_r = fmd_next_lpr_index_r(r, P.i0r, P.M, P.K);
        ns_.prev_index = fmd_next_prev_index(r.D, p0, L.ns);
        int tr, ti;
        lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, (int)P.M), (int)L.ns, tr, ti);
        ns_.lp_now_re = tr; ns_.lp_now_im = ti;
        const uint32_t l = last_lp[0];                       // lp[M-1]; M >= 2 is guaranteed by the host
        ns_.demod_pre_re = lp_re(l); ns_.demod_pre_im = lp_im(l);
        ns_.reserved = 0;
        L.st_out[c] = ns_;
        if (L.out_len) L.out_len[c] = P.K;
    }
}

}  // namespace

size_t fmd_tile_lds_bytes(const FmdLaunch& L)
{
    const size_t glen = (size_t)L.fa + 1u;
    return (size_t)L.raw_cap + ((2u * ((size_t)L.lp_cap + glen + 1u) + 15u) & ~(size_t)15u) + 16u;
}

bool fmd_tile_kernel_supports(const FmdRates& r, uint32_t raw_cap)
{
    if (r.kt % r.sr != 0) return false;                                   // plans need kt*fr = Qt*sr
    if (r.D > 64) return false;                                           // disc_fast: x + |y| < 2^30
    if ((uint64_t)r.sr * (r.kt + 2) >= (1u << 24)) return false;          // fmd_udiv_small operands
    if ((uint64_t)((r.fr + r.sr - 1) / r.sr + 2) * 32768ull >= (1u << 24)) return false;   // |group sum| < 2^24
    if ((uint32_t)r.R >= (1u << 24)) return false;
    if (raw_cap > 60u * 1024u) return false;
    return true;
}

hipError_t fmd_launch_tile(const FmdLaunch& L, hipStream_t stream)
{
    const size_t lds = fmd_tile_lds_bytes(L);
    if (L.n_channels == 0 || L.tiles == 0) return hipErrorInvalidValue;
    const uint32_t gy = L.n_channels < 65535u ? L.n_channels : 65535u;
    const uint32_t gz = (L.n_channels + 65534u) / 65535u;
    const dim3 g(L.tiles, gy, gz), b(FMD_BLOCK_THREADS);
    const uint32_t dh = (L.r.D % 2 == 0) ? L.r.D / 2 : 0;
    switch (dh) {
        case 1: hipLaunchKernelGGL(fmd_demod_tile_kernel<1>, g, b, lds, stream, L); break;
        case 2: hipLaunchKernelGGL(fmd_demod_tile_kernel<2>, g, b, lds, stream, L); break;
        case 3: hipLaunchKernelGGL(fmd_demod_tile_kernel<3>, g, b, lds, stream, L); break;   // cfg-ref, D = 6
        case 4: hipLaunchKernelGGL(fmd_demod_tile_kernel<4>, g, b, lds, stream, L); break;
        case 5: hipLaunchKernelGGL(fmd_demod_tile_kernel<5>, g, b, lds, stream, L); break;   // 2.4 Msps, D = 10
        default: hipLaunchKernelGGL(fmd_demod_tile_kernel<0>, g, b, lds, stream, L); break;  // generic windows
    }
    return hipGetLastError();
}
