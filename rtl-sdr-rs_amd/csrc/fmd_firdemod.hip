// fmd_firdemod.hip -- tapped decimating FIR -> FM discriminator -> audio resampler in ONE gfx950 kernel
// (BASELINE north_star: "the FIR + demod + resample stages fused into one CDNA4 kernel").
//
// Definition (include/fmd.h, oracle/fm_oracle.h): Demod::demodulate (examples/simple_fm.rs:256-269) with
// low_pass_complex (:337-352) replaced by   lp[m] = floor( sum_t h[t] * x[M*m + t] / 2^shift )   over the rotated
// (:276-299) and centred (:258) stream; fm_demod (:355-367, f64 sample at the first output of every call) and
// low_pass_real (:408-426) are the reference's.  With h = 1...1, T = M = downsample, shift = 0 it IS the reference
// chain -- that reduction is tested bit for bit and is the operator's anchor.
//
// One workgroup = one tile of `kt` audio samples of one channel-call, geometry by the same closed-form algebra as
// the boxcar kernel (fmd_index.h: decimated samples jA-1 .. jB, audio groups by (eq, er)); per tile:
//   1. stage the raw bytes of the FIR windows of outputs max(jA-1, 0) .. jB (LDS-DMA; through registers when the
//      tile touches the history or is not 16-byte aligned),
//   2. FIR on the matrix cores exactly as fmd_fir.hip (v_mfma_i32_16x16x64_i8 over the byte stream against the
//      banded tap matrix); instead of 8 bytes per output going to HBM the normalised sample is packed to
//      re | im << 16 and kept in LDS,
//   3. discriminator per output against its predecessor (disc_nosel: |lp| <= 16384 is enforced at creation; the f64
//      sample of the call start on one lane, guarded like the boxcar kernel's): `lg` lanes per audio group, each taking
//      `ch` consecutive samples of it and adding its partial sum to the group's accumulator in LDS (decimation ratios of
//      a tapped front end are large: 52 discriminator samples per audio sample at 2.5 Msps -> 48 kHz = 14 lanes x 4),
//   4. low_pass_real: one exact small divide per audio sample.
// HBM traffic: the u8 input once + 2 bytes per AUDIO sample (the stand-alone FIR writes 8 bytes per output).
#include "../../include/fmd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "fmd_device.h"
#include "fmd_fir_common.h"
#include "fmd_host.h"
#include "fmd_internal.h"
#include "fmd_kernels.h"

namespace {

using namespace fmd_dev;

#if defined(__HIP_DEVICE_COMPILE__)
#define FMD_AS_GLOBAL __attribute__((address_space(1)))
#else
#define FMD_AS_GLOBAL
#endif

constexpr int kThreads = 256;
constexpr int kGroupsPerWave = 4;                         // a wave accumulates up to 4 column groups (64 outputs each)
constexpr uint32_t kMaxOutputs = 64u * 4u * kGroupsPerWave;   // FIR outputs one tile can form
typedef int fd_i4 __attribute__((ext_vector_type(4)));

// geometry of one tile (fmd_tile_fast), tabulated by the host; x0 = jA * sr + i0r - k0 * fr in [0, fr): discriminator sample m
// of the tile lies in its audio group (x0 + (m - jA) * sr) / fr (fmd_audio_end solved for k)
struct FdRow { int32_t jA, jB; uint32_t eq, er, x0; };
constexpr uint32_t kFdRows = 160;                               // 3.2 KB of the 4 KB of kernel arguments

// What a fresh block of fmd_firdemod_reg_kernel needs before its LDS-DMAs can go out, in the FIRST 64 bytes of the kernel
// arguments (one s_load_dwordx16, issued together with the tile's row): a block holds its LDS from dispatch to exit, so a chain
// of dependent scalar loads in front of the first DMA is idle LDS (the boxcar kernel's FmdFastGeo, fmd_kernels.h: -5 % there).
struct FdHot {
    uint64_t iq;               // device address of the input
    uint64_t stride_w;         // dwords per channel in this call
    uint64_t amat;             // device address of the tap fragments
    uint32_t n_channels, nt, kt;
    uint32_t Hw, NP, half_M, wd_first;
    uint32_t raw_bytes, use_rows, pad;
};
static_assert(sizeof(FdHot) == 64, "one s_load_dwordx16");

struct FirDemodLaunch {
    FdHot hot;                 // (must stay the first member)
    // ---- FIR (same meaning as FirLaunch in fmd_fir.hip) ----
    const uint32_t* iq;        // [C][stride_w] dwords
    uint64_t stride_w;         // dwords per channel in this call
    const uint32_t* hist_in;   // [C][Hw]
    uint32_t* hist_out;
    uint32_t Hw, NP, half_M;
    uint32_t wd_first;         // virtual dword (history ++ call) of the window of the call's first output
    uint32_t par_first;        // stream-dword parity of that window
    const uint32_t* amat;
    uint32_t n_pass, col_bytes, shift;
    int32_t mre[2], mim[2];
    uint32_t n_channels, tiles, xcd;
    // ---- demod (fmd_index.h) ----
    FmdRates r;
    FmdClassPlan P;            // M = FIR outputs of this call, K = audio samples, nt, eq0, er0 (p0 = 0)
    FmdTiling tl;
    uint32_t fa, fb;
    float inv_sr, inv_R;
    uint32_t lp_cap, raw_bytes;
    const FmdChanState* st_in;
    FmdChanState* st_out;
    int16_t* out;
    uint64_t out_stride;
    FmdExcBuf* exc;
    double f64_guard;
    uint32_t seq;
    int32_t f64_skew;
    uint32_t dbg;              // ablation bits, honoured by -DFMD_EXPERIMENT builds only
    uint32_t use_rows;         // 1: rows[tile] holds the tile's geometry
    uint32_t lg, lg_magic, ch; // discriminator pass: `lg` lanes per audio group (tid / lg = tid * lg_magic >> 16), `ch` consecutive samples per lane
    uint32_t sr_shift;         // log2(sr) when the reduced resample rate is a power of two, else 32
    uint32_t reuse;            // host only: decim / 8 selects the fragment-reuse instantiation (decim == 8, one pass; 16 in the experiment build), 0 the plain one
    uint32_t f32_disc;         // 1: |lp| <= 2048 (the boxcar's range at downsample 16): the f32 discriminator is exact (fmd_device.h)
    uint32_t reg_ng;           // host only: > 0 selects fmd_firdemod_reg_kernel<NKU, reg_ng> (discriminator out of the matrix-core result registers)
    uint32_t digits;           // host only: tap digits of the A fragments (1: fmd_firdemod_reg1_kernel)
    uint32_t sparse;           // host only: 1 = `amat` is the 4:2-compressed matrix (fmd_firdemod_regs_kernel / _reg1s_kernel)
    FmdMagic magic_fr;         // fmd_udiv_magic(x, magic_fr) == x / fr for x < 2^24
    FdRow rows[kFdRows];
};

#ifdef FMD_EXPERIMENT
#define FD_ABLATE(bit) ((L.dbg >> (bit)) & 1u)
#define FD_KNOB_PC_EVEN ((L.dbg >> 8) & 1u)                /* A/B: keep an even column pitch (FMD_DBG bit 8) */
#else
#define FD_ABLATE(bit) false
#define FD_KNOB_PC_EVEN false
#endif

__device__ __forceinline__ uint32_t virt_dword(const FirDemodLaunch& L, uint32_t c, uint32_t w)
{
    typedef const FMD_AS_GLOBAL uint32_t* gw;
    if (w < L.Hw) return ((gw)(uintptr_t)L.hist_in)[(uint64_t)c * L.Hw + w];
    uint64_t k = w - L.Hw;
    if (k >= L.stride_w) k = L.stride_w - 1;              // the zero-weighted pad sample of an odd tap count
    return ((gw)(uintptr_t)L.iq)[(uint64_t)c * L.stride_w + k];
}

__device__ __forceinline__ fd_i4 virt_chunk(const FirDemodLaunch& L, uint32_t c, uint32_t w, bool fast)
{
    fd_i4 v;
    if (fast && w >= L.Hw && (uint64_t)(w - L.Hw) + 4 <= L.stride_w) {
        typedef const FMD_AS_GLOBAL fd_i4* gq;
        v = *(gq)(uintptr_t)(L.iq + (uint64_t)c * L.stride_w + (w - L.Hw));
    } else {
        v.x = (int)virt_dword(L, c, w);     v.y = (int)virt_dword(L, c, w + 1);
        v.z = (int)virt_dword(L, c, w + 2); v.w = (int)virt_dword(L, c, w + 3);
    }
    return v;
}

__device__ __forceinline__ void dma16(const unsigned char* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2 /* nt */);
}

// Record of a guarded f64 sample (FmdF64Exc, fmd_kernels.h) for a caller that already holds the group's sum: k < 0 = the carried partial group.
static __device__ __noinline__ void exc_emit_direct(FmdExcBuf* exc, uint32_t c, uint32_t seq, int k, int sum, int d_gpu,
                                                    int cr, int ci, int16_t* out_elem)
{
    FmdF64Exc e{};
    e.channel = c; e.cr = cr; e.ci = ci; e.d_gpu = d_gpu; e.seq = seq; e.k = k; e.sum = sum;
    e.out_elem = (uint64_t)(uintptr_t)out_elem;
    atomicAdd(&exc->guarded_total, 1u);
    const uint32_t slot = atomicAdd(&exc->count, 1u);
    if (slot < FMD_EXC_CAP) exc->rec[slot] = e; else atomicOr(&exc->err, FMD_DEVERR_EXC_CAP);
}

template <int NKU, int RS>
__global__ void __launch_bounds__(kThreads, NKU <= 7 ? 8 : 6) fmd_firdemod_kernel(const FirDemodLaunch L)   // (NKU 8 spilled at 8 blocks per CU)
{
    constexpr bool REUSE = RS > 0;                           // RS = decim / 8: k-steps between a column's output groups
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // scalar: the group tests below become s_cbranch
    __builtin_amdgcn_s_setprio(3);                           // get the loads out first (see fmd_tile_body.h)
    uint32_t c, t;
    if (L.xcd == 3u) { c = blockIdx.x * gridDim.z + blockIdx.z; t = blockIdx.y; }   // grid (8, tiles, ceil(C / 8))
    else { c = blockIdx.y + 65535u * blockIdx.z; t = blockIdx.x; }
    if (c >= L.n_channels || t >= L.P.nt) return;

    const FmdRates& r = L.r;
    const FmdClassPlan& P = L.P;
    FmdTile T;
    if (L.use_rows) {                                        // tabulated by the host (fd_enqueue): no division, no comparisons
        T.last = t + 1u == P.nt;
        T.k0 = t * r.kt;
        T.k1 = T.k0 + r.kt < P.K ? T.k0 + r.kt : P.K;
        T.jA = L.rows[t].jA; T.jB = L.rows[t].jB; T.eq = L.rows[t].eq; T.er = L.rows[t].er;
        T.nLo = T.nHi = 0;
    } else T = fmd_tile_fast(r, P, L.tl, 0u, t);             // k0, k1, jA, jB, eq, er (its input range is the boxcar's: unused)
    const int jfirst = T.jA - 1, cnt = T.jB - jfirst + 1;    // lp[jfirst .. jB]; lp[-1] is demod_pre
    const uint32_t o0 = jfirst > 0 ? (uint32_t)jfirst : 0u;  // first FIR output (of this call) the tile forms
    const uint32_t no = (uint32_t)T.jB - o0 + 1u;            // FIR outputs formed
    const uint32_t w0 = L.wd_first + o0 * L.half_M;          // first virtual dword of the tile
    const uint32_t nq = ((((no - 1) * L.half_M + L.NP + 3u) >> 2) + 3u) & ~3u;   // 16-byte slots, whole 64-byte chunks
    if ((uint32_t)cnt > L.lp_cap || nq * 16u > L.raw_bytes || no > kMaxOutputs) {
        if (tid == 0) atomicOr(&L.exc->err, (uint32_t)cnt > L.lp_cap ? FMD_DEVERR_LP_CAP : FMD_DEVERR_RAW_CAP);
        return;
    }
    uint32_t* const ypk = lds + (L.raw_bytes >> 2);          // packed lp[jfirst + i], i = 0 .. cnt-1
    int* const gsum = reinterpret_cast<int*>(ypk + L.lp_cap + 1u);   // audio group sums of the tile: [nk] + the carried tail

    const uint32_t j = lane & 15u, q = lane >> 4;
    typedef const FMD_AS_GLOBAL fd_i4* gq;
    const gq amat = (gq)(uintptr_t)L.amat + lane;
    fd_i4 A[NKU];                                            // first pass' tap fragments: in flight with the data
#pragma unroll
    for (int k = 0; k < NKU; ++k) A[k] = FD_ABLATE(4) ? fd_i4{(int)lane, k, 1, 2} : amat[k * 64];

    const bool fast = ((((uintptr_t)L.iq + ((uint64_t)c * L.stride_w + (uint64_t)w0 - L.Hw) * 4u)) & 15u) == 0u;
    const bool whole = fast && w0 >= L.Hw && (uint64_t)(w0 - L.Hw) + 4ull * nq <= L.stride_w;
    fd_i4* lq = reinterpret_cast<fd_i4*>(lds);
    if (whole) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(L.iq + (uint64_t)c * L.stride_w + (w0 - L.Hw)) + 16u * tid;
        unsigned char* dst = reinterpret_cast<unsigned char*>(lds) + 1024u * wave;
        const uint32_t nfull = nq / kThreads, ntail = nq - nfull * kThreads;
        for (uint32_t l = 0; l < nfull; ++l) dma16(src + (16u * kThreads) * l, dst + (16u * kThreads) * l);
        if (tid < ntail) dma16(src + (16u * kThreads) * nfull, dst + (16u * kThreads) * nfull);
    } else {
        for (uint32_t i = tid; i < nq; i += kThreads) lq[i] = virt_chunk(L, c, w0 + 4u * i, fast);
    }
    FmdChanState st{};
    if (jfirst < 0 || T.k0 == 0 || T.last) st = L.st_in[c];
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the LDS-DMAs (and the A fragments) have landed
    __syncthreads();

    // (Measured and rejected twice: one u8 -> s8 pass over the staged bytes instead of an xor per fragment read -- the extra
    //  barrier costs what the v_xor it saves would, profiles/HISTORY.md.)
    // ---- FIR on the matrix cores (see fmd_fir.hip) -----------------------------------------------------------
    const uint8_t* lb = reinterpret_cast<const uint8_t*>(lds);
    const uint32_t groups = (no + 63u) >> 6;
    fd_i4 acc[kGroupsPerWave];
#pragma unroll
    for (int gi = 0; gi < kGroupsPerWave; ++gi) acc[gi] = fd_i4{0, 0, 0, 0};
    // Operand-fragment reuse (decim == 8 -- BASELINE config 4 --, all k-steps in one pass; decim 16 = RS 2 was measured
    // 4-6 % slower than the plain mapping and lives in the experiment build only): the tile's outputs are
    // split into 64 columns of PC consecutive outputs (16 columns per wave); a column's 4 output groups (outputs 4 jj ..
    // 4 jj + 3) start 8 * decim bytes = RS k-steps apart, so group jj at k-step s needs exactly the tap fragment
    // A[s - RS * jj]: every 16-byte operand fragment is read and sign-flipped once and feeds up to 4 accumulators (4
    // independent MFMA chains), instead of once per group -- NKU + 3 RS fragments per wave where the plain mapping reads
    // 4 NKU.  Bytes read beyond the staged range only meet zero taps or outputs that are discarded.
    uint32_t PC = (no + 63u) >> 6;                           // outputs per column
    // Column pitch and LDS banks (ds_read_b128 is served in four groups of 16 lanes, each 8 lanes of one 16-lane quarter
    // and 8 of the next -- MI355X_MICROARCH, LDS): with decim 8 the 16-byte slot of lane (j, q) is (PC * j + q) mod 16,
    // conflict-free for PC = 2 mod 4 -- 14 at BASELINE config 4 -- and an odd PC measured 88 instead of 51 conflict cycles
    // per wave at equal time; with decim 16 it is (2 PC * j + q) mod 16, conflict-free for odd PC (8 -> 9 took the
    // experiment variant from 4-6 % to 2-3 % behind the plain mapping).  The outputs a longer column adds lie beyond
    // `no` and are discarded like the tail of the last column.
    if (RS == 2 && (PC & 1u) == 0u && PC < 16u && !FD_KNOB_PC_EVEN) PC += 1u;
    if (REUSE && !FD_ABLATE(0)) {
        const uint8_t* col = lb + ((16u * wave + j) * PC) * (L.col_bytes >> 2) + 16u * q;
#pragma unroll
        for (int sft = 0; sft < NKU + 3 * RS; ++sft) {
            fd_i4 B = *reinterpret_cast<const fd_i4*>(col + 64 * sft);
            B = B ^ (int)0x80808080;                                                       // u8 -> s8
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (sft - RS * jj >= 0 && sft - RS * jj < NKU && 4u * (uint32_t)jj < PC)    // the first two at compile time
                    acc[jj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[REUSE ? sft - RS * jj : 0], B, acc[jj], 0, 0, 0);
            }
        }
    }
    for (uint32_t pass = 0; pass < L.n_pass && !FD_ABLATE(0) && !REUSE; ++pass) {
        if (pass) {
#pragma unroll
            for (int k = 0; k < NKU; ++k) A[k] = amat[(pass * NKU + k) * 64u];
        }
#pragma unroll
        for (int gi = 0; gi < kGroupsPerWave; ++gi) {
            const uint32_t g = wave + 4u * gi;
            if (g < groups) {                                // wave-uniform
                const uint8_t* col = lb + (16u * g + j) * L.col_bytes + 64u * NKU * pass;
#pragma unroll
                for (int k = 0; k < NKU; ++k) {
                    fd_i4 B = *reinterpret_cast<const fd_i4*>(col + 64 * k + 16u * q);
                    B = B ^ (int)0x80808080;                                                   // u8 -> s8
                    acc[gi] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[k], B, acc[gi], 0, 0, 0);
                }
            }
        }
    }
    // lane (j, q) holds (re_lo, re_hi, im_lo, im_hi) of the tile's output 64 g + 4 j + q: combine the tap digits, add
    // the window-parity constant, normalise, pack to re | im << 16 and keep it in LDS
    const uint32_t par_tile = (L.par_first + o0 * L.half_M) & 1u;
    const uint32_t par = (par_tile ^ (L.half_M * q)) & 1u;
    const int cre = par ? L.mre[1] : L.mre[0], cim = par ? L.mim[1] : L.mim[0];   // two scalars + a select (an indexed kernel-argument array is a VMEM load)
    const uint32_t sgn = 0u - par_tile;                      // wave-uniform sign mask: (v ^ m) - m = m ? -v : v
#pragma unroll
    for (int gi = 0; gi < kGroupsPerWave; ++gi) {
        const uint32_t g = wave + 4u * gi;
        // plain mapping: group g, column j, output q of the column; reuse mapping: column 16 wave + j, group gi of it
        const uint32_t o = REUSE ? (16u * wave + j) * PC + 4u * (uint32_t)gi + q : 64u * g + 4u * j + q;
        if ((REUSE ? 4u * (uint32_t)gi + q < PC : g < groups) && o < no) {
            const uint32_t ure = (uint32_t)acc[gi].x + ((uint32_t)acc[gi].y << 7), uim = (uint32_t)acc[gi].z + ((uint32_t)acc[gi].w << 7);
            const int re = ((int)((ure ^ sgn) - sgn) + cre) >> L.shift;          // floor(y / 2^shift)
            const int im = ((int)((uim ^ sgn) - sgn) + cim) >> L.shift;
            ypk[(int)(o0 + o) - jfirst] = __builtin_amdgcn_perm((uint32_t)im, (uint32_t)re, 0x05040100u);   // re | im << 16
        }
    }
    if (jfirst < 0 && tid == 0) ypk[0] = pack_lp(st.demod_pre_re, st.demod_pre_im);     // lp[-1]
    // audio group k of the tile (k == nk: the trailing partial group): zero its accumulator and tabulate its first and
    // last discriminator sample -- sample k0 + k ends at e = eq + k * fa + (er + k * fb) / sr and holds fa samples, or
    // fa + 1 when the remainder of that division is below fb (fmd_tile_body.h, low_pass_real): one division per
    // GROUP here instead of one per lane in the pass below
    int* const gse = gsum + (r.kt + 2u);
    for (uint32_t k = tid; k <= r.kt; k += kThreads) {
        gsum[k] = 0;
        const uint32_t x = T.er + k * L.fb;
        uint32_t u, xrem;
        if (L.sr_shift < 32u) { u = x >> L.sr_shift; xrem = x & (r.sr - 1u); }
        else { u = fmd_udiv_small(x, r.sr, L.inv_sr); xrem = x - u * r.sr; }
        int e = (int)(T.eq + k * L.fa + u);
        int s0 = e - (int)L.fa + (xrem < L.fb ? 0 : 1);
        s0 = s0 > 0 ? s0 : 0;                                // the call's first group starts at sample 0
        e = e < T.jB ? e : T.jB;                             // the carried group ends with the call
        gse[2u * k] = s0; gse[2u * k + 1u] = e;
    }
    // the channel's last tile also writes the next call's history (the raw bytes are all in global memory)
    if (T.last) {
        typedef const FMD_AS_GLOBAL uint32_t* gw;
        for (uint32_t k = tid; k < L.Hw; k += kThreads) {
            const uint64_t w = L.stride_w + k;               // virtual dword (history ++ call), < Hw + stride_w
            L.hist_out[(uint64_t)c * L.Hw + k] = w < L.Hw ? ((gw)(uintptr_t)L.hist_in)[(uint64_t)c * L.Hw + w]
                                                         : ((gw)(uintptr_t)L.iq)[(uint64_t)c * L.stride_w + (w - L.Hw)];
        }
    }
    __syncthreads();

    // ---- fm_demod (:355-367) + the sums of low_pass_real (:408-417) ----------------------------------------------
    // `lg` lanes per audio group, each taking `ch` CONSECUTIVE discriminator samples of it (chosen by the host so
    // that lanes x ch covers the longest group, fa + 1 samples, and a tile's groups fit one pass of the 256 threads): the group's first and last sample come from ONE small
    // division per lane (as in the boxcar kernel's resampler), a lane never crosses a group, and it adds its partial sum
    // to the group's accumulator in LDS with one atomic.  The resampler that is left is one divide per audio sample.
    // History: summing groups in a separate pass over an array of discriminator samples cost 115 VALU instructions per
    // wave; a lane walking 4 samples from an arbitrary start and tracking the group boundaries itself needed two exact
    // divisions per lane plus a boundary test per sample (~270 per wave in all, against ~200 here).
    const uint32_t nk = T.k1 - T.k0;
    const uint32_t ng = nk + (T.last ? 1u : 0u);             // + the trailing partial group that is carried to the next call
    int d_first = 0, cr0 = 0, ci0 = 0;
    bool any_guard = false;
    {
        const uint32_t gq0 = (tid * L.lg_magic) >> 16, lg = tid - gq0 * L.lg;           // tid / lg, tid % lg
        const uint32_t gpp = ((uint32_t)kThreads * L.lg_magic) >> 16;                    // whole groups per pass: the lanes beyond
        for (uint32_t gq = gq0 < gpp ? gq0 : ng; gq < ng && !FD_ABLATE(1); gq += gpp) { //   gpp * lg sit out
            const int s = gse[2u * gq], e = gse[2u * gq + 1u];  // the group's first and last sample (tabulated above)
            const int j0 = s + (int)(lg * L.ch);
            const int j1 = j0 + (int)L.ch - 1 < e ? j0 + (int)L.ch - 1 : e;
            if (j0 <= j1) {
                const uint32_t* yp = ypk + (j0 - jfirst);
                uint32_t prev = yp[-1];
                int part = 0;
                if (L.f32_disc && L.ch == 4u) {              // wave-uniform.  Four samples straight-line (four independent chains,
                    const int n = j1 - j0;                   // no loop control); the ones beyond the lane's run are computed on
                    const uint32_t a0 = yp[0], a1 = yp[1], a2 = yp[2], a3 = yp[3];   // whatever LDS holds there and dropped by a select
                    const float r0 = (float)lp_re(a0), i0 = (float)lp_im(a0), r1 = (float)lp_re(a1), i1 = (float)lp_im(a1);
                    const float r2 = (float)lp_re(a2), i2 = (float)lp_im(a2), r3 = (float)lp_re(a3), i3 = (float)lp_im(a3);
                    const int d0 = disc_f32_c(r0, i0, (float)lp_re(prev), (float)lp_im(prev));
                    const int d1 = disc_f32_c(r1, i1, r0, i0), d2 = disc_f32_c(r2, i2, r1, i1), d3 = disc_f32_c(r3, i3, r2, i2);
                    part = d0 + (n >= 1 ? d1 : 0) + (n >= 2 ? d2 : 0) + (n >= 3 ? d3 : 0);
                } else if (L.f32_disc) {                     // wave-uniform: components in f32, c = a * conj(b) by mul + fma
                    float pr = (float)lp_re(prev), pi = (float)lp_im(prev);
                    for (int i = 0; i <= j1 - j0; ++i) {
                        const uint32_t a = yp[i];
                        const float ar = (float)lp_re(a), ai = (float)lp_im(a);
                        part += disc_f32_c(ar, ai, pr, pi);  // (:362); the value fits i16, `as i16` changes nothing
                        pr = ar; pi = ai;
                    }
                } else {
                    for (int i = 0; i <= j1 - j0; ++i) {
                        const uint32_t a = yp[i];
                        part += (int)(int16_t)disc_nosel(a, prev);   // (:362) `as i16`, summed as i32 (:414)
                        prev = a;
                    }
                }
                if (j0 == 0) {                               // the first sample of the call takes the f64 path (:359): thread 0 only
                    const uint32_t a = yp[0], b = yp[-1];
                    fmd_mul_conj(lp_re(a), lp_im(a), lp_re(b), lp_im(b), cr0, ci0);
                    bool g;
                    int d = polar_f64(cr0, ci0, L.f64_guard, g);
#ifdef FMD_EXPERIMENT
                    if (g) d += L.f64_skew;
#endif
                    any_guard = g;
                    d_first = (int)(int16_t)d;
                    part += d_first - (int)(int16_t)disc_nosel(a, b);
                }
                atomicAdd(&gsum[gq], part);
            }
        }
    }
    __syncthreads();

    // ---- low_pass_real (:418-422): one divide per audio sample ---------------------------------------------------
    int16_t* const outc = L.out + (uint64_t)c * L.out_stride;
    for (uint32_t k = tid; k < nk && !FD_ABLATE(2); k += kThreads) {
        int sum = gsum[k];
        if (T.k0 + k == 0u) sum += st.now_lpr;               // continues the previous call's partial sum (:410-417)
        outc[T.k0 + k] = (int16_t)fmd_sdiv_small(sum, r.R, L.inv_R);
    }
    if (jfirst < 0 && tid == 0 && any_guard) {               // guarded f64 sample (FmdF64Exc, fmd_kernels.h)
        const bool in_audio = P.K > 0u;                      // sample 0 lies in audio group 0, or in the carried tail
        exc_emit_direct(L.exc, c, L.seq, in_audio ? 0 : -1, gsum[0] + st.now_lpr, d_first, cr0, ci0, outc);
    }

    // ---- state after the call (last tile; simple_fm.rs:232-239) --------------------------------------------------
    if (T.last && tid == 0) {
        FmdChanState ns_{};
        ns_.now_lpr = gsum[nk] + (P.K == 0 ? st.now_lpr : 0);       // the trailing partial group
        ns_.lpr_index_r = fmd_next_lpr_index_r(r, P.i0r, P.M, P.K);
        const uint32_t l = ypk[cnt - 1];                     // demod_pre = the last filter output (M >= 2 guaranteed by the host)
        ns_.demod_pre_re = lp_re(l); ns_.demod_pre_im = lp_im(l);
        L.st_out[c] = ns_;
    }
}

// ---- the same operator with the discriminator taken straight out of the matrix-core result registers (round 4) ---------
// fmd_firdemod_kernel above keeps the normalised FIR outputs in LDS (one packed dword each), meets at a barrier and
// runs the discriminator as a second pass over that array.  Here, for decimate 8 (BASELINE config 4), the FIR outputs
// never leave the registers the matrix instructions put them in:
//   * a wave's 16 columns hold PC = 4 NG - 2 consecutive outputs each (14, 18, 22 for NG = 4, 5, 6: a column pitch of
//     2 mod 4 outputs keeps the ds_read_b128 operand reads conflict-free, see above), lane (j, q) of the wave owns
//     outputs 4 gi + q of column j, gi < NG, in acc[gi]; consecutive waves overlap by ONE output, so every output a wave
//     owns has its predecessor inside the same wave;
//   * the predecessor of lane (j, q >= 1)'s output gi is lane (j, q - 1)'s output gi: 16 lanes down, one
//     ds_bpermute_b32 per component (a crossbar move, no LDS memory, no bank conflicts); lane (j, 0) takes lane
//     (j, 3)'s output gi - 1 -- the same move of the previous register -- and, for gi = 0, the last output of column
//     j - 1 (lane (j - 1, 1), register NG - 1);
//   * each lane then runs its NG discriminators (f32 component form), splits their sum at the one audio-group boundary
//     its outputs can straddle (a group is at least 4 NG - 3 outputs long: checked by the host, which otherwise takes the
//     kernel above) and adds the two partial sums to the groups' accumulators in LDS.
// Gone: the packed-sample array (4 bytes per output: the tile grows from 17 to 22 audio samples at 8 tiles per CU), its
// stores and 4-way-conflicting reads, one barrier, the per-lane run bookkeeping of the second pass.
// (blocks per CU by column length; the long-filter shapes NKU >= 7 and the table-less form -- calls of more than kFdRows tiles --
//  one step lower: at the bounds of the common shapes they spilled 4 ... 8 registers to scratch memory)
constexpr int fd_reg_blocks(int nku, int ng, bool rows)
{
    if (nku >= 7 && ng == 6) return 5;
    if (nku >= 7 || !rows) return ng <= 6 ? 6 : (ng <= 8 ? 4 : 3);
    return ng <= 6 ? 8 : (ng <= 8 ? 5 : 3);
}
// DIGITS = 1 (every |tap| <= 127, fmd_fir_common.h): the 16 rows of an operand fragment are (re, im) of EIGHT consecutive outputs, so
// a lane's NG outputs of its column come out of NG / 2 accumulators -- lane (j, q) holds outputs 8 g + 2q and 8 g + 2q + 1 of column j
// in acc[g] -- fed by NKU + NG - 2 operand fragments (two chunks per accumulator step); the columns, the tiles and everything behind
// the discriminators are the two-digit form's.  The odd outputs' predecessors are the lane's own even ones; the even ones' sit 16
// lanes down (lane (j, 3)'s previous accumulator for q = 0, column j - 1's last output -- lane (j - 1, 2) -- for the first).
// SP: the matrix phase on the 4:2 structured-sparse instruction (v_smfmac_i32_16x16x128_i8: a 128-byte K chunk for the cost of a dense
// 64-byte one -- 17 against 18 clocks, tools/smfmac_probe.hip).  rotate_90 leaves every re row with bytes 0 / 3 and every im row with
// bytes 1 / 2 of each stream dword: the tap matrix IS 4:2 sparse, with one fixed pattern per row (fmd_fir_common.h).  Lane (j, q)
// supplies the 32 stream bytes at 32 q of the chunk as the B operand; an accumulator's chunks are 128 bytes apart, consecutive
// accumulators 64 (two digits) or 128 (one), so fragment s = AS gi + 2 kc -- in 64-byte units from the column -- serves every
// (accumulator gi, chunk kc) on that diagonal: (NKU + 1) / 2 instead of NKU matrix instructions per accumulator.
template <int NKU, int NG, bool ROWS, int DIGITS, bool SP>
__device__ __forceinline__ void fd_reg_body(const FirDemodLaunch& L)
{
    // GEO8: a column is EIGHT outputs per accumulator step -- the one-digit forms, and the two-digit sparse form, which keeps re and
    // im in accumulators of their own (rows = (lo, hi) of eight outputs): its accumulators then sit a whole sparse chunk apart like
    // the one-digit form's, 12 fragment reads for 24 matrix instructions per wave where the interleaved rows would need 24 reads.
    constexpr bool GEO8 = DIGITS == 1 || SP;
    static_assert(!GEO8 || NG % 2 == 0, "eight outputs per accumulator step: two outputs per lane and accumulator");
    constexpr int NKS = (NKU + 1) / 2;                       // 128-byte chunks of the sparse form
    constexpr int NAF = SP ? (DIGITS == 2 ? 2 * NKS : NKS) : NKU;   // tap fragments a lane holds (two digits, sparse: re set, then im set)
    constexpr int PC = 4 * NG - 2;                           // outputs per column
    constexpr int WSTEP = 16 * PC - 1;                       // tile outputs from one wave's first column to the next wave's
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    __builtin_amdgcn_s_setprio(3);                           // get the loads out first (see fmd_tile_body.h)
    // grid (8, tiles, ceil(C / 8)) always: blockIdx.x IS the XCD, which works through its own contiguous eighth of the channels
    // (fewer than 8 channels: the surplus blocks exit at once)
    // No branch in front of the LDS-DMAs: hipcc issues a block's scalar loads where they are used and waits for each batch
    // before the next branch, so every early exit in front of the staging is one more memory round trip with the tile's
    // LDS already allocated.  A surplus block of the grid (channel count not a multiple of 8) stages channel 0's bytes like
    // any other and leaves right behind its loads; the sizing assertion is checked behind them too.
    const uint32_t t = blockIdx.y;
    const FdHot H = L.hot;
    const FmdRates& r = L.r;
    const FmdClassPlan& P = L.P;
    int jA, jB;
    uint32_t eq, er, x0;
    if constexpr (ROWS) {                                    // tabulated by the host (fd_enqueue): one more scalar load, no division
        jA = L.rows[t].jA; jB = L.rows[t].jB; eq = L.rows[t].eq; er = L.rows[t].er; x0 = L.rows[t].x0;
    } else {
        const FmdTile T = fmd_tile_fast(r, P, L.tl, 0u, t);
        jA = T.jA; jB = T.jB; eq = T.eq; er = T.er;
        x0 = (uint32_t)jA * r.sr + P.i0r - T.k0 * r.fr;      // (all three terms below 2^32: fmd_ranges_fit32)
    }
    const uint32_t gz = gridDim.z;
    // every scalar load of the prologue -- the hot block, the row, the grid size -- goes out before anything waits (left alone, the
    // scheduler issues them one behind the other's wait)
    if constexpr (ROWS) __builtin_amdgcn_sched_barrier(0);
    const uint32_t c_grid = blockIdx.x * gz + blockIdx.z;
    const bool surplus = c_grid >= H.n_channels || t >= H.nt;
    const uint32_t c = surplus ? 0u : c_grid;
    const bool last = t + 1u == H.nt;
    const int jfirst = jA - 1;                               // lp[jfirst .. jB]; lp[-1] is demod_pre
    const uint32_t o0 = jfirst > 0 ? (uint32_t)jfirst : 0u;  // first FIR output (of this call) the tile forms
    const uint32_t no = (uint32_t)jB - o0 + 1u;              // FIR outputs formed
    const uint32_t w0 = H.wd_first + o0 * H.half_M;          // first virtual dword of the tile
    const uint32_t nq_need = ((((no - 1) * H.half_M + H.NP + 3u) >> 2) + 3u) & ~3u;   // 16-byte slots, whole 64-byte chunks
    const bool oversize = nq_need * 16u > H.raw_bytes || no > (uint32_t)(64 * PC - 3);   // (cannot happen: the host sized the tiles)
    const uint32_t nq = oversize ? (H.raw_bytes >> 4) & ~3u : nq_need;
    const uint32_t* const iq_w = reinterpret_cast<const uint32_t*>((uintptr_t)H.iq);
    int* const gsum = reinterpret_cast<int*>(lds + (H.raw_bytes >> 2));   // audio group sums of the tile: [nk] + the carried tail
    int* const gse = gsum + (H.kt + 2u);                     // (first, last) discriminator sample of every group
    int* const tail = gse + 2u * (H.kt + 2u);                // the tile's last FIR output (re, im): demod_pre of the next call

    const uint32_t j = lane & 15u, q = lane >> 4;
    typedef const FMD_AS_GLOBAL fd_i4* gq;
    const bool fast = (((H.iq + ((uint64_t)c * H.stride_w + (uint64_t)w0 - H.Hw) * 4u)) & 15u) == 0u;
    const bool whole = fast && w0 >= H.Hw && (uint64_t)(w0 - H.Hw) + 4ull * nq <= H.stride_w;
    fd_i4* lq = reinterpret_cast<fd_i4*>(lds);
    // the tap fragments FIRST (five small L2-resident loads per lane: behind the DMAs they queue behind 30 KB of staging
    // traffic and every wave waits for them at the barrier -- measured on the stand-alone FIR kernel: 2 % per call)
    const gq amat = (gq)(uintptr_t)H.amat + lane;
    fd_i4 A[NAF];
#pragma unroll
    for (int k = 0; k < NAF; ++k) A[k] = amat[k * 64];
    if (whole) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(iq_w + (uint64_t)c * H.stride_w + (w0 - H.Hw)) + 16u * tid;
        unsigned char* dst = reinterpret_cast<unsigned char*>(lds) + 1024u * wave;
        const uint32_t nfull = nq / kThreads, ntail = nq - nfull * kThreads;
        for (uint32_t l = 0; l < nfull; ++l) dma16(src + (16u * kThreads) * l, dst + (16u * kThreads) * l);
        if (tid < ntail) dma16(src + (16u * kThreads) * nfull, dst + (16u * kThreads) * nfull);
    } else {
        for (uint32_t i = tid; i < nq; i += kThreads) lq[i] = virt_chunk(L, c, w0 + 4u * i, fast);
    }
    if (surplus || oversize) {                               // (rare) leave -- behind the staging loads: they write this block's LDS
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (oversize && !surplus && tid == 0) atomicOr(&L.exc->err, FMD_DEVERR_RAW_CAP);
        return;
    }
    const uint32_t k0 = t * H.kt, k1 = k0 + H.kt < P.K ? k0 + H.kt : P.K;
    FmdChanState st{};
    if (jfirst < 0 || k0 == 0 || last) st = L.st_in[c];
    // under the load latency: audio group k of the tile (k == nk: the trailing partial group) -- zero its accumulator and
    // tabulate its first and last discriminator sample (see fmd_firdemod_kernel)
    for (uint32_t k = tid; k <= H.kt && !FD_ABLATE(22); k += kThreads) {   // (ablation 22: no group table)
        gsum[k] = 0;
        const uint32_t x = er + k * L.fb;
        uint32_t u, xrem;
        if (L.sr_shift < 32u) { u = x >> L.sr_shift; xrem = x & (r.sr - 1u); }
        else { u = fmd_udiv_small(x, r.sr, L.inv_sr); xrem = x - u * r.sr; }
        int e = (int)(eq + k * L.fa + u);
        int s0 = e - (int)L.fa + (xrem < L.fb ? 0 : 1);
        s0 = s0 > 0 ? s0 : 0;                                // the call's first group starts at sample 0
        e = e < jB ? e : jB;                                 // the carried group ends with the call
        gse[2u * k] = s0; gse[2u * k + 1u] = e;
    }
    // the channel's last tile also writes the next call's history (the raw bytes are all in global memory)
    if (last) {
        typedef const FMD_AS_GLOBAL uint32_t* gw;
        for (uint32_t k = tid; k < L.Hw; k += kThreads) {
            const uint64_t w = L.stride_w + k;               // virtual dword (history ++ call), < Hw + stride_w
            L.hist_out[(uint64_t)c * L.Hw + k] = w < L.Hw ? ((gw)(uintptr_t)L.hist_in)[(uint64_t)c * L.Hw + w]
                                                         : ((gw)(uintptr_t)L.iq)[(uint64_t)c * L.stride_w + (w - L.Hw)];
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the LDS-DMAs (and the A fragments) have landed
    __syncthreads();
    if (FD_ABLATE(23)) {                                     // ablation: staging skeleton only (prologue, DMAs, barrier, one store)
        if (tid == 0) L.out[(uint64_t)c * L.out_stride + t * H.kt] = (int16_t)(lds[t & 63u] + (uint32_t)A[0].x);
        return;
    }

    // ---- FIR on the matrix cores: NKU + NG - 1 operand fragments feed NG accumulators (fragment reuse, see above) ----------
    const uint32_t tcol = wave * (uint32_t)WSTEP + j * (uint32_t)PC;     // tile output index of the column's first output
    if (FD_ABLATE(27)) __builtin_amdgcn_s_sleep(16);                     // pacing probes: 1024 clocks behind the staging barrier ...
    if (FD_ABLATE(24)) __builtin_amdgcn_s_setprio(3);                    // knob: the matrix phase's dependent LDS -> xor -> MFMA chains at raised priority
    if (FD_ABLATE(25)) __builtin_amdgcn_s_setprio(1);
    constexpr int NA = GEO8 ? NG / 2 : NG;                   // accumulators per lane (and component, in the two-digit sparse form)
    constexpr int AS = GEO8 ? 2 : 1;                         // K-chunks from one accumulator's column to the next (8 / 4 outputs of 16 bytes)
    constexpr int NACC = (SP && DIGITS == 2) ? 2 * NA : NA;  // (two digits, sparse: acc[0 .. NA) re, acc[NA .. 2 NA) im)
    fd_i4 acc[NACC];
#pragma unroll
    for (int gi = 0; gi < NACC; ++gi) acc[gi] = fd_i4{0, 0, 0, 0};   // (all zero: the first matrix instruction takes the inline constant, no v_mov)
    if constexpr (SP) {
        typedef int fd_i8 __attribute__((ext_vector_type(8)));
        // (round 6: the lanes of the ODD K quarters read the second half of their 32 bytes first -- fmd_fir_kswap, fmd_fir_common.h: the
        //  tap fragments are permuted accordingly.  With every quarter reading its first half first, the 16 lanes the LDS serves
        //  together hit the even 16-byte slots of the bank row twice: SQ_LDS_BANK_CONFLICT was a third of the LDS-active cycles of
        //  this kernel in round 5; tools/ldsbench.py: 1.65 x a conflict-free read, 0.94 x with the swap)
        const uint8_t* col = reinterpret_cast<const uint8_t*>(lds) + 16u * tcol + 32u * q;   // the lane's 32 bytes of a 128-byte chunk
        const uint32_t h0 = FD_ABLATE(29) ? 0u : 16u * (q & 1u);                            // (probe, experiment build: round 5's order -- wrong audio, same work)
        // (re rows keep bytes 0 and 3 of every dword: index pairs (0, 3); im rows bytes 1 and 2: (1, 2))
        const uint32_t row = lane & 15u;
        const int idx = ((DIGITS == 1 ? row & 1u : (row >> 1) & 1u) != 0u) ? (int)0x99999999u : (int)0xCCCCCCCCu;
#pragma unroll
        for (int s = 0; s <= AS * (NA - 1) + 2 * (NKS - 1); s += (AS == 2 ? 2 : 1)) {
            if (FD_ABLATE(16)) continue;
            const fd_i4 b0 = *reinterpret_cast<const fd_i4*>(col + h0 + 64 * s) ^ (int)0x80808080;     // u8 -> s8
            const fd_i4 b1 = *reinterpret_cast<const fd_i4*>(col + (h0 ^ 16u) + 64 * s) ^ (int)0x80808080;
            const fd_i8 B = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int kc = 0; kc < NKS; ++kc)
                if ((s - 2 * kc) >= 0 && (s - 2 * kc) % AS == 0 && (s - 2 * kc) / AS < NA) {
                    const int g = (s - 2 * kc) / AS;
                    if constexpr (DIGITS == 2) {             // one fragment, both components: every re row keeps bytes 0 / 3, every im row 1 / 2
                        acc[g] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(A[kc], B, acc[g], (int)0xCCCCCCCCu, 0, 0);
                        acc[NA + g] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(A[NKS + kc], B, acc[NA + g], (int)0x99999999u, 0, 0);
                    } else {
                        acc[g] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(A[kc], B, acc[g], idx, 0, 0);
                    }
                }
        }
    } else {
        const uint8_t* col = reinterpret_cast<const uint8_t*>(lds) + 16u * tcol + 16u * q;   // decimate 8: 16 bytes per output
#pragma unroll
        for (int sft = 0; sft < NKU + AS * (NA - 1); ++sft) {
            if (FD_ABLATE(16)) continue;                                                    // ablation: no operand reads, no matrix instructions
            fd_i4 B = *reinterpret_cast<const fd_i4*>(col + 64 * sft);
            B = B ^ (int)0x80808080;                                                       // u8 -> s8
#pragma unroll
            for (int jj = 0; jj < NA; ++jj)
                if (sft - AS * jj >= 0 && sft - AS * jj < NKU) {
                    if (!FD_ABLATE(17)) acc[jj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[sft - AS * jj], B, acc[jj], 0, 0, 0);
                    else acc[jj].x ^= B.x + A[sft - AS * jj].y;                            // ablation: operand reads kept, every matrix instruction replaced by two vector ones
                }
        }
    }
    // lane (j, q) holds (re_lo, re_hi, im_lo, im_hi) of tile output tcol + 4 gi + q in acc[gi]: combine the tap digits, add
    // the window-parity constant, normalise -- and keep the components as exact f32 integers
    // Decimate 8: every window starts at an even stream dword (4 o dwords from the call's first one, itself even), so the
    // window parity is 0 for every output of every call: no sign flip of the matrix-core results (the general kernel
    // negates them for odd parities) and one pair of additive constants (scalars).
    const int cre = L.mre[0], cim = L.mim[0];
    if (FD_ABLATE(24) || FD_ABLATE(25)) __builtin_amdgcn_s_setprio(0);
    if (FD_ABLATE(26)) __builtin_amdgcn_s_setprio(2);                    // knob: the discriminator phase at raised priority instead
    // fr[k], fi[k]: the lane's k-th output of its column, at column index CI(k) = 4 k + q (two digits) or 8 (k / 2) + 2 q + k % 2 (one)
    float fr[NG], fi[NG];
    if constexpr (SP && DIGITS == 2) {
#pragma unroll
        for (int g = 0; g < NA; ++g) {                       // (lo, hi) of two outputs per accumulator, re and im in accumulators of their own
            fr[2 * g] = (float)((int)((uint32_t)acc[g].x + ((uint32_t)acc[g].y << 7) + (uint32_t)cre) >> L.shift);
            fr[2 * g + 1] = (float)((int)((uint32_t)acc[g].z + ((uint32_t)acc[g].w << 7) + (uint32_t)cre) >> L.shift);
            fi[2 * g] = (float)((int)((uint32_t)acc[NA + g].x + ((uint32_t)acc[NA + g].y << 7) + (uint32_t)cim) >> L.shift);
            fi[2 * g + 1] = (float)((int)((uint32_t)acc[NA + g].z + ((uint32_t)acc[NA + g].w << 7) + (uint32_t)cim) >> L.shift);
        }
    } else if constexpr (DIGITS == 2) {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (FD_ABLATE(18)) { fr[gi] = u2f((uint32_t)acc[gi].x & 0x3F800FFFu); fi[gi] = u2f((uint32_t)acc[gi].z & 0x3F800FFFu); continue; }   // ablation: no digit combine / shift / conversion
            const int re = (int)((uint32_t)acc[gi].x + ((uint32_t)acc[gi].y << 7) + (uint32_t)cre) >> L.shift;     // floor(y / 2^shift)
            const int im = (int)((uint32_t)acc[gi].z + ((uint32_t)acc[gi].w << 7) + (uint32_t)cim) >> L.shift;
            fr[gi] = (float)re; fi[gi] = (float)im;
        }
    } else {
#pragma unroll
        for (int g = 0; g < NA; ++g) {                       // (re, im) of two outputs per accumulator: nothing to combine
            fr[2 * g] = (float)((int)((uint32_t)acc[g].x + (uint32_t)cre) >> L.shift);
            fi[2 * g] = (float)((int)((uint32_t)acc[g].y + (uint32_t)cim) >> L.shift);
            fr[2 * g + 1] = (float)((int)((uint32_t)acc[g].z + (uint32_t)cre) >> L.shift);
            fi[2 * g + 1] = (float)((int)((uint32_t)acc[g].w + (uint32_t)cim) >> L.shift);
        }
    }
    if (last) {                                              // block-uniform, one tile per channel: the output that becomes demod_pre
        if constexpr (!GEO8) {
            const int dl = (int)no - 1 - (int)(tcol + q);    // (tiles overlap by one output: two lanes may hold it, with the same value)
            if (dl >= 0 && (dl & 3) == 0 && (dl >> 2) < (q < 2u ? NG : NG - 1)) {
#pragma unroll
                for (int gi = 0; gi < NG; ++gi)
                    if ((dl >> 2) == gi) { tail[0] = (int)fr[gi]; tail[1] = (int)fi[gi]; }
            }
        } else {
            const int dlc = (int)no - 1 - (int)tcol;         // the tile's last output as a column index of this lane's column
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int ci = 8 * (k >> 1) + 2 * (int)q + (k & 1);
                if (ci == dlc && ci < PC) { tail[0] = (int)fr[k]; tail[1] = (int)fi[k]; }
            }
        }
    }
    // predecessors: 16 lanes down (q - 1, same column); lane (j, 0) takes the previous register of lane (j, 3), and for its
    // first output the last output of column j - 1 (lane (j - 1, 1), register NG - 1: PC - 1 = 4 (NG - 1) + 1)
    const int down = (int)(((lane + 48u) & 63u) << 2);       // byte address of lane - 16 (mod 64)
    const int left = (int)((16u + ((j + 15u) & 15u)) << 2);  // byte address of lane (j - 1, 1)
    // (one digit: only the lane's EVEN outputs have their predecessor in another lane -- the odd output of the same accumulator 16
    //  lanes down; column j - 1's last output, index PC - 1 = 8 (NG / 2 - 1) + 2 * 2 + 1, is lane (j - 1, 2)'s last one)
    const int left1 = (int)((32u + ((j + 15u) & 15u)) << 2);
    float pr[NA], pi[NA];
#pragma unroll
    for (int gi = 0; gi < NA; ++gi) {
        const int src = GEO8 ? 2 * gi + 1 : gi;
        if (FD_ABLATE(19)) { pr[gi] = fi[src]; pi[gi] = fr[src]; continue; }                // ablation: no predecessor moves
        pr[gi] = u2f((uint32_t)__builtin_amdgcn_ds_bpermute(down, (int)f2u(fr[src])));
        pi[gi] = u2f((uint32_t)__builtin_amdgcn_ds_bpermute(down, (int)f2u(fi[src])));
    }
    const float xr = FD_ABLATE(19) ? fi[0] : u2f((uint32_t)__builtin_amdgcn_ds_bpermute(GEO8 ? left1 : left, (int)f2u(fr[NG - 1])));
    const float xi = FD_ABLATE(19) ? fr[0] : u2f((uint32_t)__builtin_amdgcn_ds_bpermute(GEO8 ? left1 : left, (int)f2u(fi[NG - 1])));
    const bool q0 = q == 0u;
    // (every lane takes part in every move: ds_bpermute returns 0 for a source lane that is masked off, so the row-0 lanes
    //  cannot fetch their previous-register values under an EXEC mask of their own -- they select instead)
    // ---- fm_demod (:355-367) per output, summed per audio group (:408-417) -------------------------------------------
    const uint32_t tmin = jfirst < 0 ? 0u : 1u;              // tile output 0 is only a predecessor (except at the call start)
    const uint32_t t_first = tcol + (GEO8 ? 2u * q : q);     // the lane's first tile output; sample index m = o0 + t
    // audio group of the lane's first output and that group's last sample
    const int dm = (int)(o0 + t_first) - jA;                 // >= -1
    const uint32_t kq = (uint32_t)fmd_sdiv_magic((int)(x0 + (uint32_t)(dm > 0 ? dm : 0) * r.sr), L.magic_fr);
    const uint32_t kqc = kq <= H.kt ? kq : H.kt;             // (lanes beyond the tile: any valid row)
    const int e_lo = gse[2u * kqc + 1u];
    // the lane's outputs gi < g_hi lie inside the tile (and are real rows of the column: PC = 4 NG - 2), those gi < g_split in
    // the first audio group; its output 0 is skipped where it is only a predecessor (the tile's first output away from the
    // call start, the one output a wave shares with its predecessor)
    const int g_max = q < 2u ? NG : NG - 1;
    const int g_in = ((int)no - (int)t_first + 3) >> 2;
    const int g_hi = g_in < g_max ? g_in : g_max;
    const int g_split = ((e_lo - (int)(o0 + t_first)) >> 2) + 1;
    const bool skip0 = q0 && j == 0u && (wave > 0u || tmin != 0u);
    int sum_all = 0, sum_lo = 0;
    int d_first = 0, cr0 = 0, ci0 = 0;
    bool any_guard = false;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        float br, bi;
        if constexpr (!GEO8) {
            br = q0 ? (gi == 0 ? xr : pr[gi - 1]) : pr[gi];
            bi = q0 ? (gi == 0 ? xi : pi[gi - 1]) : pi[gi];
        } else if (gi & 1) {                                 // the odd output of an accumulator: its predecessor is the lane's own even one
            br = fr[gi - 1]; bi = fi[gi - 1];
        } else {
            br = q0 ? (gi == 0 ? xr : pr[gi / 2 - 1]) : pr[gi / 2];
            bi = q0 ? (gi == 0 ? xi : pi[gi / 2 - 1]) : pi[gi / 2];
        }
        int d = FD_ABLATE(20) ? (int)(f2u(fr[gi]) ^ f2u(bi)) + (int)f2u(br) : disc_f32_c(fr[gi], fi[gi], br, bi);   // (:362); the value fits i16, `as i16` changes nothing (ablation 20: no discriminator)
        if (gi == 0 && jfirst < 0 && tid == 0) {             // the first sample of the call takes the f64 path (:359) against demod_pre
            fmd_mul_conj((int)fr[0], (int)fi[0], st.demod_pre_re, st.demod_pre_im, cr0, ci0);
            bool g;
            d = polar_f64(cr0, ci0, L.f64_guard, g);
#ifdef FMD_EXPERIMENT
            if (g) d += L.f64_skew;
#endif
            any_guard = g;
            d = (int)(int16_t)d;
            d_first = d;
        }
        int dv, dlo;
        if constexpr (!GEO8) {
            dv = (gi < g_hi && !(gi == 0 && skip0)) ? d : 0;
            dlo = gi < g_split ? dv : 0;
        } else {                                             // the lane's outputs are not equally spaced: each by its own column index
            const int ci = 8 * (gi >> 1) + 2 * (int)q + (gi & 1);
            const int tt = (int)tcol + ci;                   // tile output index; sample index m = o0 + tt
            dv = (ci < PC && tt < (int)no && !(gi == 0 && skip0)) ? d : 0;
            dlo = (int)o0 + tt <= e_lo ? dv : 0;
        }
        sum_all += dv;
        sum_lo += dlo;
    }
    // (a lane whose outputs are all beyond the tile or not owned adds zeros: harmless)
    if (FD_ABLATE(21)) { if (sum_all == 0x7fffffff) gsum[0] = sum_lo; }                   // ablation: no group sums in LDS
    else {
        if (kq <= H.kt) atomicAdd(&gsum[kqc], sum_lo);
        if (kq + 1u <= H.kt && sum_all != sum_lo) atomicAdd(&gsum[kq + 1u], sum_all - sum_lo);
    }
    __syncthreads();

    // ---- low_pass_real (:418-422): one divide per audio sample ---------------------------------------------------
    if (FD_ABLATE(28)) __builtin_amdgcn_s_sleep(16);         // ... or in front of the divide pass
    const uint32_t nk = k1 - k0;
    int16_t* const outc = L.out + (uint64_t)c * L.out_stride;
    for (uint32_t k = tid; k < nk; k += kThreads) {
        int sum = gsum[k];
        if (k0 + k == 0u) sum += st.now_lpr;                 // continues the previous call's partial sum (:410-417)
        outc[k0 + k] = (int16_t)fmd_sdiv_small(sum, r.R, L.inv_R);
    }
    if (jfirst < 0 && tid == 0 && any_guard) {               // guarded f64 sample (FmdF64Exc, fmd_kernels.h)
        const bool in_audio = P.K > 0u;                      // sample 0 lies in audio group 0, or in the carried tail
        exc_emit_direct(L.exc, c, L.seq, in_audio ? 0 : -1, gsum[0] + st.now_lpr, d_first, cr0, ci0, outc);
    }
    // ---- state after the call (last tile; simple_fm.rs:232-239) --------------------------------------------------
    if (last && tid == 0) {
        FmdChanState ns_{};
        ns_.now_lpr = gsum[nk] + (P.K == 0 ? st.now_lpr : 0);       // the trailing partial group
        ns_.lpr_index_r = fmd_next_lpr_index_r(r, P.i0r, P.M, P.K);
        ns_.demod_pre_re = tail[0]; ns_.demod_pre_im = tail[1];     // demod_pre = the last filter output (M >= 2 guaranteed by the host)
        L.st_out[c] = ns_;
    }
}

template <int NKU, int NG, bool ROWS>
__global__ void __launch_bounds__(kThreads, fd_reg_blocks(NKU, NG, ROWS)) fmd_firdemod_reg_kernel(const FirDemodLaunch L)
{
    fd_reg_body<NKU, NG, ROWS, 2, false>(L);
}

// the one-digit form (every |tap| <= 127): see fd_reg_body
template <int NKU, int NG, bool ROWS>
__global__ void __launch_bounds__(kThreads, fd_reg_blocks(NKU, NG, ROWS)) fmd_firdemod_reg1_kernel(const FirDemodLaunch L)
{
    fd_reg_body<NKU, NG, ROWS, 1, false>(L);
}

// both with the matrix phase on the 4:2 sparse instruction (NKU stays the DENSE chunk count of the shape: the kernel derives its own)
// (two digits, sparse: two tap-fragment sets and two accumulator sets -- NG = 6 takes the 6 blocks per CU its LDS budget leaves anyway)
template <int NKU, int NG, bool ROWS>
__global__ void __launch_bounds__(kThreads, NG == 6 && fd_reg_blocks(NKU, NG, ROWS) > 6 ? 6 : fd_reg_blocks(NKU, NG, ROWS)) fmd_firdemod_regs_kernel(const FirDemodLaunch L)
{
    fd_reg_body<NKU, NG, ROWS, 2, true>(L);
}

template <int NKU, int NG, bool ROWS>
__global__ void __launch_bounds__(kThreads, fd_reg_blocks(NKU, NG, ROWS)) fmd_firdemod_reg1s_kernel(const FirDemodLaunch L)
{
    fd_reg_body<NKU, NG, ROWS, 1, true>(L);
}

template <int NKU>
void launch(const FirDemodLaunch& L, dim3 g, size_t lds, hipStream_t s)
{
#define FD_REGX(K, N) if (L.use_rows) hipLaunchKernelGGL((K<NKU, N, true>), g, dim3(kThreads), lds, s, L); \
                      else hipLaunchKernelGGL((K<NKU, N, false>), g, dim3(kThreads), lds, s, L)
    if (L.reg_ng && L.digits == 1u) {                       // (even column parameters only: the host falls back to two digits otherwise)
        if (L.sparse) {
            if (L.reg_ng == 4u) { FD_REGX(fmd_firdemod_reg1s_kernel, 4); }
            else if (L.reg_ng == 6u) { FD_REGX(fmd_firdemod_reg1s_kernel, 6); }
            else { FD_REGX(fmd_firdemod_reg1s_kernel, 8); }
            return;
        }
#ifdef FMD_EXPERIMENT
        if (L.reg_ng == 4u) { FD_REGX(fmd_firdemod_reg1_kernel, 4); }
        else if (L.reg_ng == 6u) { FD_REGX(fmd_firdemod_reg1_kernel, 6); }
        else { FD_REGX(fmd_firdemod_reg1_kernel, 8); }
        return;
#endif
    }
    if (L.reg_ng && L.sparse) {                             // two digits, re / im split (even column parameters)
        if (L.reg_ng == 4u) { FD_REGX(fmd_firdemod_regs_kernel, 4); }
        else if (L.reg_ng == 6u) { FD_REGX(fmd_firdemod_regs_kernel, 6); }
        else { FD_REGX(fmd_firdemod_regs_kernel, 8); }
        return;
    }
#undef FD_REGX
#define FD_REG(N) if (L.use_rows) hipLaunchKernelGGL((fmd_firdemod_reg_kernel<NKU, N, true>), g, dim3(kThreads), lds, s, L); \
                  else hipLaunchKernelGGL((fmd_firdemod_reg_kernel<NKU, N, false>), g, dim3(kThreads), lds, s, L)
    if (L.reg_ng == 4u) { FD_REG(4); }
    else if (L.reg_ng == 5u) { FD_REG(5); }
    else if (L.reg_ng == 6u) { FD_REG(6); }
    else if (L.reg_ng == 7u) { FD_REG(7); }
    else if (L.reg_ng == 8u) { FD_REG(8); }
#ifdef FMD_EXPERIMENT
    else if (L.reg_ng == 10u) { FD_REG(10); }
    else if (L.reg_ng == 12u) { FD_REG(12); }
#endif
#undef FD_REG
    else if (L.reuse == 1u) hipLaunchKernelGGL((fmd_firdemod_kernel<NKU, 1>), g, dim3(kThreads), lds, s, L);
#ifdef FMD_EXPERIMENT
    else if (L.reuse == 2u) hipLaunchKernelGGL((fmd_firdemod_kernel<NKU, 2>), g, dim3(kThreads), lds, s, L);   // FMD_FD_REUSE16: measured 4-6 % slower than the plain mapping
#endif
    else hipLaunchKernelGGL((fmd_firdemod_kernel<NKU, 0>), g, dim3(kThreads), lds, s, L);
}

#define FD_TRY(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            char m_[256];                                                                   \
            snprintf(m_, sizeof m_, "%s failed: %s", #expr, hipGetErrorString(e_));         \
            fmd_internal_set_err(m_);                                                       \
            return e_ == hipErrorOutOfMemory ? FMD_ERR_NOMEM : FMD_ERR_HIP;                 \
        }                                                                                   \
    } while (0)

#define FD_ON_DEVICE(dev)                                                                   \
    FmdDeviceGuard dev_guard_(dev);                                                         \
    if (dev_guard_.error() != hipSuccess) { fmd_internal_set_err("hipSetDevice failed"); return FMD_ERR_HIP; }

}  // namespace

struct fmd_firdemod {
    uint32_t T = 0, M = 0, C = 0, NP = 0, Hw = 0, shift = 0;
    uint32_t lp_bound = 0;                                // ceil(128 * sum|taps| / 2^shift): the largest |lp| component
    int device = 0;
    uint64_t pos = 0;                                     // samples consumed per channel
    FmdFirMfmaPlan plan;
    uint32_t* d_amat = nullptr;
    uint32_t* d_hist[2] = {nullptr, nullptr};
    FmdChanState* d_state[2] = {nullptr, nullptr};
    int cur = 0;
    FmdRates r{};
    uint32_t i0r = 0;                                     // resampler phase of every channel (host mirror)
    uint32_t last_K = 0;
    uint32_t taps_hash = 0;                               // FNV-1a of the taps: a checkpoint names the filter it belongs to
    uint32_t lp_cap = 0, raw_bytes = 0;
    FmdExcBuf* d_exc = nullptr;
    uint32_t* h_head = nullptr;                           // page-locked copy of the report head (fmd_firdemod_check, see fmd_demod_check)
    double f64_guard = 0x1p-20;
    int32_t f64_skew = 0;
    uint32_t seq = 0;
    uint64_t f64_guarded = 0, f64_patched = 0;
    FmdStreamOrder order;
    uint32_t dbg = 0;                                     // ablation bits (FMD_DBG, experiment build)
    bool no_rows = false;                                 // FMD_FD_ROWS=0: geometry on the device (A/B)
    bool reuse16 = false;                                 // FMD_FD_REUSE16 (experiment build): fragment reuse at decim 16 too
    bool no_reuse = false, int_disc = false;              // FMD_FD_NOREUSE / FMD_FD_INT_DISC: plain MFMA mapping / integer discriminator (A/B), read at creation
    int last_rows = -1;                                   // the most recent launch's `use_rows` (-1: none yet) -- part of the kernel's name
    uint32_t reg_ng = 0;                                  // > 0: fmd_firdemod_reg_kernel with this many output groups per column (decimate 8, f32 discriminator)
    bool sparse = false;                                  // the register form's matrix phase on v_smfmac (d_amat holds the compressed matrix)
    size_t lds_budget = 20480;                            // LDS per tile (8 tiles per CU); FMD_FD_LDS (tuning)
    hipStream_t stream = nullptr;
    uint8_t* d_iq = nullptr; size_t d_iq_cap = 0;
    int16_t* d_out = nullptr; size_t d_out_cap = 0;
};

namespace {

void fd_counts(const fmd_firdemod* f, uint64_t ns, uint64_t* m0, uint64_t* m1)
{
    const uint64_t S = f->pos, T = f->T, M = f->M;
    *m0 = S >= T ? (S - T) / M + 1 : 0;
    *m1 = S + ns >= T ? (S + ns - T) / M + 1 : 0;
}

// Discriminator pass: `lg` lanes per audio group, `ch` consecutive samples per lane, lg * ch >= fa + 1 (a group holds fa
// or fa + 1 samples).  The smallest ch >= 4 whose kt + 1 groups (the tile's audio samples + the carried partial group)
// fit one pass of the 256 threads; one pass is not required (the kernel loops), only cheaper.
void fd_lanes(uint32_t fa, uint32_t kt, uint32_t* lg, uint32_t* magic, uint32_t* ch)
{
    uint32_t c = 4, l = (fa + 1u + c - 1u) / c;
    while (l > 1u && (uint64_t)(kt + 1u) * l > (uint32_t)kThreads && c < fa + 1u) { ++c; l = (fa + 1u + c - 1u) / c; }
    if (l > (uint32_t)kThreads) { l = kThreads; c = (fa + 1u + l - 1u) / l; }
    uint32_t m = 65536u / l + 1u;
    for (uint32_t t = 0; t <= (uint32_t)kThreads; ++t)
        if (((t * m) >> 16) != t / l) { m = 0; break; }
    if (!m) {                                             // never seen; keep the arithmetic exact anyway: a power of two
        l = 1; while (l * c < fa + 1u && l < (uint32_t)kThreads) l *= 2;
        c = (fa + 1u + l - 1u) / l; m = 65536u / l;
    }
    *lg = l; *magic = m; *ch = c;
}

// LDS per tile for `kt` audio samples: raw bytes of the windows + packed samples + discriminator samples
bool fd_sizes(const fmd_firdemod* f, uint32_t kt, uint32_t* lp_cap, uint32_t* raw_bytes, size_t* lds)
{
    FmdRates r = f->r; r.kt = kt;
    uint32_t cap = fmd_tile_lp_cap(r);
    const uint32_t half_M = f->M / 2;
    if (f->reg_ng) {
        // Register form.  Its calls are tiled with the trailing partial audio group counted as a group (fd_enqueue: nt =
        // K / kt + 1), so no tile holds more than kt groups -- at most ceil(kt fr / sr) discriminator samples + the
        // predecessor -- where fmd_tile_lp_cap budgets a whole further group for the last tile: 22 instead of 20 audio
        // samples per tile at BASELINE config 4.  No packed-sample array; the columns' operand reads bound the raw region.
        cap = (uint32_t)(((uint64_t)kt * r.fr + r.sr - 1) / r.sr) + 3u;
        const uint64_t staged = (((((uint64_t)(cap - 1) * half_M + f->NP + 3) / 4) + 3) & ~(uint64_t)3) * 16;
        const uint32_t pc = 4u * f->reg_ng - 2u;
        if (cap > 64u * pc - 3u) return false;
        const uint64_t touched = 16ull * (63u * pc - 3u) + 64ull * (f->plan.nku + f->reg_ng);   // (the sparse form's 128-byte chunks: one 64-byte unit further for an odd chunk count)
        const uint64_t raw = ((staged > touched ? staged : touched) + 15) & ~(uint64_t)15;
        const uint64_t total = raw + 12ull * (kt + 2) + 8 + 16;               // + group sums, the (first, last) table, the tail sample
        if (total > 60 * 1024) return false;
        *lp_cap = cap; *raw_bytes = (uint32_t)raw; *lds = (size_t)total;
        return true;
    }
    if (cap > kMaxOutputs) return false;
    const uint64_t staged = (((((uint64_t)(cap - 1) * half_M + f->NP + 3) / 4) + 3) & ~(uint64_t)3) * 16;
    const uint64_t touched = (uint64_t)16 * ((cap + 63) / 64) * (8u * f->M) + (uint64_t)64 * f->plan.n_pass * f->plan.nku;
    const uint64_t raw = ((staged > touched ? staged : touched) + 15) & ~(uint64_t)15;
    const uint64_t total = raw + 4ull * (cap + 1) + 12ull * (kt + 2) + 16;   // + group sums and the (first, last) table
    if (total > 60 * 1024) return false;
    *lp_cap = cap; *raw_bytes = (uint32_t)raw; *lds = (size_t)total;
    return true;
}

int fd_enqueue(fmd_firdemod* f, const void* d_iq, size_t nbytes, void* d_out, size_t out_cap, size_t* n_each, hipStream_t stream)
{
    if (nbytes % 8 != 0) { fmd_internal_set_err("nbytes % 8 != 0 (simple_fm.rs:286 would panic)"); return FMD_ERR_BAD_LENGTH; }
    if (nbytes == 0 || nbytes > (1ull << 31)) { fmd_internal_set_err("nbytes out of range"); return FMD_ERR_UNSUPPORTED; }
    if (((uintptr_t)d_iq & 3u) != 0 || ((uintptr_t)d_out & 1u) != 0) { fmd_internal_set_err("misaligned device buffer"); return FMD_ERR_INVALID_ARG; }
    const uint64_t ns = nbytes / 2;
    uint64_t m0, m1;
    fd_counts(f, ns, &m0, &m1);
    const uint64_t Mdec = m1 - m0;
    if (Mdec < 2) { fmd_internal_set_err("the call yields fewer than 2 filter outputs (simple_fm.rs:356 asserts > 1)"); return FMD_ERR_TOO_SHORT; }
    FmdRates r = f->r;
    if (!fmd_ranges_fit32(r, Mdec * r.D)) { fmd_internal_set_err("call exceeds the 32-bit index range for these rates"); return FMD_ERR_UNSUPPORTED; }
    FirDemodLaunch L{};
    L.P = fmd_make_plan(r, 0u, f->i0r, 0u);
    L.P.M = (uint32_t)Mdec;
    L.P.K = fmd_num_audio(r, f->i0r, L.P.M);
    L.P.nt = f->reg_ng ? L.P.K / r.kt + 1u : fmd_num_tiles(r, L.P.K);   // register form: the trailing partial group counts as a group (fd_sizes)
    if (L.P.K > out_cap) { fmd_internal_set_err("out_cap too small"); return FMD_ERR_CAPACITY; }
    L.iq = static_cast<const uint32_t*>(d_iq);
    L.stride_w = nbytes / 4;
    L.hist_in = f->d_hist[f->cur]; L.hist_out = f->d_hist[f->cur ^ 1];
    L.Hw = f->Hw; L.NP = f->NP; L.half_M = f->M / 2;
    const uint64_t vs0 = f->M * m0 + 2ull * f->Hw - f->pos;      // virtual sample index of the first window (even)
    L.wd_first = (uint32_t)(vs0 / 2);
    L.par_first = (uint32_t)((f->M * m0 / 2) & 1u);
    L.amat = f->d_amat; L.n_pass = f->plan.n_pass; L.col_bytes = 8u * f->M; L.shift = f->shift;
    for (int p = 0; p < 2; ++p) { L.mre[p] = f->plan.mre[p]; L.mim[p] = f->plan.mim[p]; }
    L.n_channels = f->C; L.tiles = L.P.nt;
    L.r = r; L.tl = fmd_make_tiling(r);
    L.fa = r.fr / r.sr; L.fb = r.fr % r.sr;
    L.inv_sr = 1.0f / (float)r.sr; L.inv_R = 1.0f / (float)r.R;
    L.lp_cap = f->lp_cap; L.raw_bytes = f->raw_bytes;
    L.st_in = f->d_state[f->cur]; L.st_out = f->d_state[f->cur ^ 1];
    L.out = static_cast<int16_t*>(d_out); L.out_stride = out_cap;
    L.exc = f->d_exc; L.f64_guard = f->f64_guard; L.seq = f->seq + 1; L.f64_skew = f->f64_skew;
    L.dbg = f->dbg;
    fd_lanes(L.fa, r.kt, &L.lg, &L.lg_magic, &L.ch);
    L.reg_ng = f->reg_ng;
    L.digits = f->plan.digits;
    L.sparse = f->sparse ? 1u : 0u;
    L.magic_fr = fmd_make_magic(r.fr);
    L.reuse = (f->M == 8u || (f->M == 16u && f->reuse16)) && f->plan.n_pass == 1u && !f->no_reuse ? f->M / 8u : 0u;
    L.f32_disc = f->lp_bound <= 2048u && !f->int_disc ? 1u : 0u;
    L.sr_shift = 32u;
    if ((r.sr & (r.sr - 1u)) == 0u) { L.sr_shift = 0u; while ((1u << L.sr_shift) < r.sr) ++L.sr_shift; }
    L.use_rows = 0u;
    if (L.P.nt <= kFdRows && !f->no_rows) {
        for (uint32_t t = 0; t < L.P.nt; ++t) {
            const FmdTile T = fmd_tile_fast(r, L.P, L.tl, 0u, t);
            L.rows[t] = FdRow{T.jA, T.jB, T.eq, T.er, (uint32_t)((uint64_t)T.jA * r.sr + f->i0r - (uint64_t)T.k0 * r.fr)};
        }
        L.use_rows = 1u;
    }
    uint32_t lc, rb; size_t lds;
    if (!fd_sizes(f, r.kt, &lc, &rb, &lds)) { fmd_internal_set_err("tile sizing failed"); return FMD_ERR_UNSUPPORTED; }
    FD_TRY(f->order.before(stream));
    const uint32_t per = (f->C + 7u) / 8u;
    dim3 g(L.tiles, f->C < 65535u ? f->C : 65535u, (f->C + 65534u) / 65535u);
    if (f->C >= 8u && L.tiles <= 65535u && per <= 65535u) { g = dim3(8u, L.tiles, per); L.xcd = 3u; }
    if (L.reg_ng) {                                       // the register form always runs the (8, tiles, ceil(C / 8)) grid
        if (L.tiles > 65535u || per > 65535u) { fmd_internal_set_err("call too large for the register-form grid"); return FMD_ERR_UNSUPPORTED; }
        g = dim3(8u, L.tiles, per); L.xcd = 3u;
        FdHot& H = L.hot;
        H.iq = (uint64_t)(uintptr_t)L.iq; H.stride_w = L.stride_w; H.amat = (uint64_t)(uintptr_t)L.amat;
        H.n_channels = L.n_channels; H.nt = L.P.nt; H.kt = r.kt;
        H.Hw = L.Hw; H.NP = L.NP; H.half_M = L.half_M; H.wd_first = L.wd_first;
        H.raw_bytes = L.raw_bytes; H.use_rows = L.use_rows;
    }
    // (cannot happen -- fmd_firdemod_new keeps a one-digit register plan or a split plan only together with `sparse` in
    //  the shipped build -- but launch() would hand a one-digit tap matrix to the two-digit kernel, or run on an empty dense
    //  matrix: silently wrong audio.  Refuse instead.)
    {
#ifdef FMD_EXPERIMENT
        const bool dense_one_digit_kernel = true;
#else
        const bool dense_one_digit_kernel = false;
#endif
        if ((f->reg_ng && f->plan.digits == 1u && !f->sparse && !dense_one_digit_kernel) || (f->plan.split && !f->sparse)) {
            fmd_internal_set_err("internal: tap-matrix form and kernel form disagree");
            return FMD_ERR_UNSUPPORTED;
        }
    }
    switch (f->plan.nku) {
        case 1: launch<1>(L, g, lds, stream); break;
        case 2: launch<2>(L, g, lds, stream); break;
        case 3: launch<3>(L, g, lds, stream); break;
        case 4: launch<4>(L, g, lds, stream); break;
        case 5: launch<5>(L, g, lds, stream); break;
        case 6: launch<6>(L, g, lds, stream); break;
        case 7: launch<7>(L, g, lds, stream); break;
        default: launch<8>(L, g, lds, stream); break;
    }
    FD_TRY(hipGetLastError());
    (void)f->order.after(stream);
    f->seq += 1;
    f->cur ^= 1;
    f->i0r = fmd_next_lpr_index_r(r, f->i0r, L.P.M, L.P.K);
    f->pos += ns;
    f->last_K = L.P.K;
    f->last_rows = (int)L.use_rows;
    if (n_each) *n_each = L.P.K;
    return FMD_OK;
}

}  // namespace

extern "C" {

size_t fmd_firdemod_out_cap(uint32_t decim, uint32_t rate_out, uint32_t rate_resample, size_t nbytes)
{
    if (!decim || !rate_out) return 0;
    const uint64_t M = nbytes / 2 / decim + 2;
    return (size_t)((M * rate_resample + rate_out - 1) / rate_out + 1);
}

int fmd_firdemod_new(const int16_t* taps, uint32_t n_taps, uint32_t decim, uint32_t shift, uint32_t rate_out,
                     uint32_t rate_resample, const fmd_device_config* dev, fmd_firdemod** out)
{
    if (!taps || !dev || !out || dev->n_channels == 0) { fmd_internal_set_err("null / empty argument"); return FMD_ERR_INVALID_ARG; }
    *out = nullptr;
    if (n_taps == 0 || n_taps > 1024 || decim == 0 || decim % 2 != 0 || decim > 64 || shift > 24) {
        fmd_internal_set_err("need 1 <= n_taps <= 1024, an even 2 <= decim <= 64 and shift <= 24");
        return FMD_ERR_UNSUPPORTED;
    }
    if (rate_resample == 0 || rate_out < rate_resample) {
        fmd_internal_set_err("need rate_out >= rate_resample >= 1 (simple_fm.rs:421 divides by rate_out / rate_resample)");
        return FMD_ERR_BAD_RATES;
    }
    uint64_t sum_abs = 0;
    for (uint32_t t = 0; t < n_taps; ++t) {
        if (taps[t] > 2047 || taps[t] < -2047) { fmd_internal_set_err("|tap| > 2047"); return FMD_ERR_UNSUPPORTED; }
        sum_abs += (uint64_t)(taps[t] < 0 ? -taps[t] : taps[t]);
    }
    // |lp| <= 128 * sum|h| / 2^shift must stay within the discriminator's range (the boxcar's bound at downsample 128)
    if (((128ull * sum_abs) >> shift) > 16384ull) {
        fmd_internal_set_err("filter gain too large for the discriminator: need (128 * sum|taps|) >> shift <= 16384");
        return FMD_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fmd_internal_set_err("no HIP device (this library has no CPU path)"); return FMD_ERR_NO_DEVICE; }
    int device = dev->device_id;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    hipDeviceProp_t prop;
    if (device >= ndev || hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fmd_internal_set_err("device is not a gfx950");
        return FMD_ERR_NO_DEVICE;
    }
    fmd_firdemod* f = new (std::nothrow) fmd_firdemod();
    if (!f) return FMD_ERR_NOMEM;
    f->T = n_taps; f->M = decim; f->C = dev->n_channels; f->device = device; f->shift = shift;
    f->taps_hash = 2166136261u;
    for (uint32_t t = 0; t < n_taps; ++t) {
        f->taps_hash = (f->taps_hash ^ (uint32_t)(taps[t] & 0xFF)) * 16777619u;
        f->taps_hash = (f->taps_hash ^ (uint32_t)((taps[t] >> 8) & 0xFF)) * 16777619u;
    }
    // (rounded UP: lp = floor(y / 2^shift) of a negative y reaches -ceil(128 sum|h| / 2^shift) when every tap is negative, and the
    //  f32 discriminator's exactness argument -- fmd_device.h: |s| <= 2^23 -- needs the true bound, not the truncated one)
    f->lp_bound = (uint32_t)((128ull * sum_abs + ((1ull << shift) - 1ull)) >> shift);
    f->NP = ((n_taps + 1) / 2 + 3u) & ~3u;
    const uint32_t H = n_taps - 1, Hp = H + (H & 1u);
    f->Hw = Hp / 2;
    if (!fmd_fir_build_mfma(taps, n_taps, decim, f->plan)) { delete f; fmd_internal_set_err("shape does not fit the matrix-core form"); return FMD_ERR_UNSUPPORTED; }
    FmdRates& r = f->r;
    r.D = decim; r.fast = rate_out; r.slow = rate_resample;
    r.g = fmd_gcd(r.fast, r.slow); r.fr = r.fast / r.g; r.sr = r.slow / r.g;
    r.R = (int32_t)(r.fast / r.slow);
    const uint64_t fa = r.fr / r.sr;
    if (r.fr > FMD_MAX_RATE_REDUCED || (fa + 2) * 32768ull >= (1u << 24) || (uint32_t)r.R >= (1u << 24)) {
        delete f; fmd_internal_set_err("rate ratio outside the exact-small-divide range"); return FMD_ERR_UNSUPPORTED;
    }
    // tiling: the most audio samples per tile that fit (<= 1024 filter outputs, ~20 KB of LDS -> 8 tiles per CU)
    uint32_t best = 0; size_t lds = 0;
    // knobs: -DFMD_EXPERIMENT builds only (fmd_host.h); constants in the shipped library
    const uint32_t kt_env = fmd_knob_u32("FMD_FD_KT", 0);
    f->lds_budget = (size_t)fmd_knob_u32("FMD_FD_LDS", (uint32_t)f->lds_budget);
    const bool lds_knob = fmd_knob("FMD_FD_LDS") != nullptr;
    f->no_rows = fmd_knob_u32("FMD_FD_ROWS", 1) == 0u;
    f->no_reuse = fmd_knob("FMD_FD_NOREUSE") != nullptr;
    f->reuse16 = fmd_knob("FMD_FD_REUSE16") != nullptr;
    f->int_disc = fmd_knob("FMD_FD_INT_DISC") != nullptr;
    f->dbg = fmd_knob_u32("FMD_DBG", 0);
    // decimate 8 with the f32 discriminator and audio groups of at least 16 filter outputs: the register form, with the longest
    // columns the audio groups admit (a lane's 4 NG - 3 consecutive outputs must not straddle more than one group boundary) up to
    // NG = 8 -- columns of 30 outputs, 36 audio samples per tile, 5 tiles per CU at BASELINE config 4: measured best of NG = 4 ... 12
    // (profiles/r04_experiments.md section 5).  FMD_FD_REG (experiment build): 0 = off, 4 ... 12 = that NG.
    f->reg_ng = 0;
    if (decim == 8u && f->plan.n_pass == 1u && f->lp_bound <= 2048u && !f->no_reuse && !f->int_disc) {
        uint32_t ng = fmd_knob_u32("FMD_FD_REG", (uint32_t)(fa / 4 < 8 ? fa / 4 : 8));
#ifndef FMD_EXPERIMENT
        if (ng > 8u) ng = 8u;                                 // (9 ... 12 are instantiated in the experiment build only)
#endif
        if (ng >= 4u && ng <= 12u && ng != 9u && ng != 11u && fa >= 4ull * ng && (uint64_t)r.sr * (64u * (4u * ng - 2u)) < (1u << 24)) f->reg_ng = ng;
        // tiles per CU that go with the column length: 8 (20 KB) up to 18 outputs per column, then 6, 5, 5, 4, 3 -- in whole LDS
        // allocation granules (1280 bytes on gfx950, 128 per CU: profiles/r05_lds_granule.jsonl), so that a tile a few bytes over
        // does not cost a CU one of its blocks.  The full columns of NG = 6 / 7 need 23.3 / 27.6 KB: one granule more than 7 / 6
        // blocks per CU leave, so they get the budget of 6 / 5 (rounds 4 - 5 gave them 23 400 / 27 300 bytes: the blocks per CU of
        // the larger budget with the tile of the smaller one; session r05as: whole-granule budgets BELOW the columns' need are
        // far worse -- the planner then cuts the tile until it fits)
        static const uint32_t budget[13] = {0, 0, 0, 0, 20480, 20480, 26880, 32000, 32000, 32000, 40960, 40960, 53760};
        if (f->reg_ng && !lds_knob) f->lds_budget = budget[f->reg_ng];
        // an 8-bit filter (every |tap| <= 127) in the register form with an even column parameter: one digit per tap, (re, im) of eight
        // outputs per operand fragment -- NG / 2 accumulators and 24 instead of 40 matrix instructions per wave at config 4's shape
        // (fmd_firdemod_reg1_kernel; FMD_FD_DIGITS=2, experiment build: keep two digits for an A/B)
#ifndef FMD_FD_SPARSE_DEFAULT
#define FMD_FD_SPARSE_DEFAULT 1
#endif
        const bool sparse_on = fmd_knob_u32("FMD_FD_SPARSE", FMD_FD_SPARSE_DEFAULT) != 0u;
        bool small = true;
        for (uint32_t t = 0; t < n_taps; ++t) if (taps[t] > 127 || taps[t] < -127) small = false;
        // (an 8-bit filter whose audio groups admit an ODD column parameter -- 5 or 7 -- takes the even one below it: the one-digit
        //  sparse form with slightly shorter columns beats two dense digits by more than the columns cost)
        // (12-bit filters keep the odd parameter: session r05bl, 5 -> 4 sparse +6 ... +7 %, 7 -> 6 sparse -1 %)
        // (the one-digit plan is built FIRST -- with eight outputs per column it needs n_taps <= 200 to fit one K pass,
        //  the register form admits 232 -- and the column parameter is lowered only when that plan exists: a filter of 201 ... 232
        //  8-bit taps keeps its odd parameter and the dense two-digit kernel with the full columns)
        FmdFirMfmaPlan one;
        const bool one_fits = small && fmd_knob_u32("FMD_FD_DIGITS", 0) != 2u && fmd_fir_build_mfma(taps, n_taps, decim, one, 1u) && one.n_pass == 1u;
        if (one_fits && sparse_on && (f->reg_ng == 5u || f->reg_ng == 7u) && fmd_knob("FMD_FD_REG") == nullptr) {
            f->reg_ng -= 1u;
            if (!lds_knob) f->lds_budget = budget[f->reg_ng];
        }
        const bool even_ng = f->reg_ng == 4u || f->reg_ng == 6u || f->reg_ng == 8u;
#ifdef FMD_EXPERIMENT
        const bool dense_one_digit = true;                    // (fmd_firdemod_reg1_kernel: instantiated in the experiment build only)
#else
        const bool dense_one_digit = false;
#endif
        if (one_fits && even_ng && (sparse_on || dense_one_digit)) f->plan = one;
        // The register form's matrix phase runs on the 4:2 sparse matrix instruction when the column parameter is even: one digit -- 12
        // instead of 24 matrix instructions per wave at config 4's shape for the same 12 operand reads (-4.3 %); two digits -- re and
        // im in accumulators of their own (fmd_fir_common.h `split`), 24 instead of 40 for the same 12 reads.  (Two digits with the
        // components interleaved in one accumulator, 64 bytes = half a sparse chunk apart, needed 24 reads: +4.5 %, dropped.)
        // FMD_FD_SPARSE=0 (experiment build): the dense instruction.
        f->sparse = even_ng && sparse_on;
        if (f->sparse && f->plan.digits == 2u) {
            FmdFirMfmaPlan sp;
            if (fmd_fir_build_mfma(taps, n_taps, decim, sp, 2u, true) && sp.n_pass == 1u) f->plan = sp;
            else f->sparse = false;
        }
    }
    for (uint32_t kt = 1; kt <= 1024; ++kt) {
        if ((uint64_t)r.sr * (kt + 2) >= (1u << 24)) break;
        uint32_t lc, rb; size_t l;
        if (!fd_sizes(f, kt, &lc, &rb, &l)) break;
        if (l > f->lds_budget && best) break;
        best = kt;
        if (kt_env && kt == kt_env) break;
    }
    if (!best) { delete f; fmd_internal_set_err("one audio sample does not fit a tile: rate_out / rate_resample x decim too large"); return FMD_ERR_UNSUPPORTED; }
    r.kt = best;
    if (!fd_sizes(f, r.kt, &f->lp_cap, &f->raw_bytes, &lds)) { delete f; return FMD_ERR_UNSUPPORTED; }
    if (const char* g = fmd_knob("FMD_F64_GUARD_LOG2")) f->f64_guard = ldexp(1.0, atoi(g));   // experiment build only
    f->f64_skew = fmd_knob_i32("FMD_F64_SKEW", 0);

    auto fail = [&](const char* what) { fmd_internal_set_err(what); fmd_firdemod_free(f); return FMD_ERR_HIP; };
    FmdDeviceGuard guard(device);
    if (guard.error() != hipSuccess) return fail("hipSetDevice");
    const std::vector<uint32_t>& amat = f->sparse ? f->plan.amat_s : f->plan.amat;
    if (hipMalloc(&f->d_amat, amat.size() * 4) != hipSuccess) return fail("hipMalloc(tap matrix)");
    if (hipMemcpy(f->d_amat, amat.data(), amat.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy(tap matrix)");
    const size_t hb = (size_t)f->C * (f->Hw ? f->Hw : 1) * 4, sb = (size_t)f->C * sizeof(FmdChanState);
    for (int i = 0; i < 2; ++i) {
        if (hipMalloc(&f->d_hist[i], hb) != hipSuccess || hipMemset(f->d_hist[i], 0, hb) != hipSuccess) return fail("hipMalloc(history)");
        if (hipMalloc(&f->d_state[i], sb) != hipSuccess || hipMemset(f->d_state[i], 0, sb) != hipSuccess) return fail("hipMalloc(state)");
    }
    if (hipMalloc(&f->d_exc, sizeof(FmdExcBuf)) != hipSuccess || hipMemset(f->d_exc, 0, sizeof(FmdExcBuf)) != hipSuccess) return fail("hipMalloc(reports)");
    if (hipHostMalloc(reinterpret_cast<void**>(&f->h_head), 16, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc(report head)");
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate");
    if (hipDeviceSynchronize() != hipSuccess) return fail("hipDeviceSynchronize");
    *out = f;
    return FMD_OK;
}

void fmd_firdemod_free(fmd_firdemod* f)
{
    if (!f) return;
    FmdDeviceGuard guard(f->device);
    (void)hipDeviceSynchronize();
    f->order.destroy();
    if (f->d_amat) (void)hipFree(f->d_amat);
    for (int i = 0; i < 2; ++i) { if (f->d_hist[i]) (void)hipFree(f->d_hist[i]); if (f->d_state[i]) (void)hipFree(f->d_state[i]); }
    if (f->d_exc) (void)hipFree(f->d_exc);
    if (f->h_head) (void)hipHostFree(f->h_head);
    if (f->d_iq) (void)hipFree(f->d_iq);
    if (f->d_out) (void)hipFree(f->d_out);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

int fmd_firdemod_reset(fmd_firdemod* f)
{
    if (!f) return FMD_ERR_INVALID_ARG;
    FD_ON_DEVICE(f->device);
    FD_TRY(hipDeviceSynchronize());
    const size_t hb = (size_t)f->C * (f->Hw ? f->Hw : 1) * 4, sb = (size_t)f->C * sizeof(FmdChanState);
    for (int i = 0; i < 2; ++i) { FD_TRY(hipMemset(f->d_hist[i], 0, hb)); FD_TRY(hipMemset(f->d_state[i], 0, sb)); }
    FD_TRY(hipMemset(f->d_exc, 0, 16));
    FD_TRY(hipDeviceSynchronize());
    f->pos = 0; f->cur = 0; f->i0r = 0; f->last_K = 0;
    f->order.reset();
    return FMD_OK;
}

int fmd_firdemod_demodulate_device(fmd_firdemod* f, const void* d_iq, size_t nbytes, void* d_out, size_t out_cap,
                                   size_t* out_len_each, void* stream)
{
    if (!f || !d_iq || !d_out) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    FD_ON_DEVICE(f->device);
    return fd_enqueue(f, d_iq, nbytes, d_out, out_cap, out_len_each, static_cast<hipStream_t>(stream));
}

int fmd_firdemod_check(fmd_firdemod* f)
{
    if (!f) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    FD_ON_DEVICE(f->device);
    if (f->order.have_last && f->h_head) {                   // one stream synchronisation in the common case (see fmd_demod_check)
        f->h_head[0] = f->h_head[1] = ~0u;
        hipError_t e = hipMemcpyAsync(f->h_head, f->d_exc, 16, hipMemcpyDeviceToHost, f->order.last);
        if (e == hipSuccess) e = hipStreamSynchronize(f->order.last);
        if (e == hipSuccess && f->h_head[0] == 0u && f->h_head[1] == 0u) return FMD_OK;
        if (e != hipSuccess) (void)hipGetLastError();
    }
    FD_TRY(hipDeviceSynchronize());
    return fmd_internal_resolve_exc(f->d_exc, f->r.R, f->seq, f->seq, f->d_state[f->cur], nullptr, 0, &f->f64_guarded, &f->f64_patched);
}

int fmd_firdemod_demodulate_batch(fmd_firdemod* f, const uint8_t* iq, size_t nbytes, int16_t* out, size_t out_cap,
                                  size_t* out_len)
{
    if (!f || !iq || !out || !out_len) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    FD_ON_DEVICE(f->device);
    if (nbytes % 8 != 0) { fmd_internal_set_err("nbytes % 8 != 0"); return FMD_ERR_BAD_LENGTH; }
    const size_t in_bytes = nbytes * (size_t)f->C, out_elems = out_cap * (size_t)f->C;
    if (in_bytes > f->d_iq_cap) {
        if (f->d_iq) { FD_TRY(hipFree(f->d_iq)); f->d_iq = nullptr; f->d_iq_cap = 0; }
        FD_TRY(hipMalloc(&f->d_iq, in_bytes ? in_bytes : 1));
        f->d_iq_cap = in_bytes;
    }
    if (out_elems > f->d_out_cap) {
        if (f->d_out) { FD_TRY(hipFree(f->d_out)); f->d_out = nullptr; f->d_out_cap = 0; }
        FD_TRY(hipMalloc(&f->d_out, (out_elems ? out_elems : 1) * sizeof(int16_t)));
        f->d_out_cap = out_elems;
    }
    FD_TRY(hipMemcpyAsync(f->d_iq, iq, in_bytes, hipMemcpyHostToDevice, f->stream));
    size_t n = 0;
    int rc = fd_enqueue(f, f->d_iq, nbytes, f->d_out, out_cap, &n, f->stream);
    if (rc) return rc;
    if (n) FD_TRY(hipMemcpyAsync(out, f->d_out, out_elems * sizeof(int16_t), hipMemcpyDeviceToHost, f->stream));
    FD_TRY(hipStreamSynchronize(f->stream));
    for (uint32_t c = 0; c < f->C; ++c) out_len[c] = n;
    return fmd_internal_resolve_exc(f->d_exc, f->r.R, f->seq, f->seq, f->d_state[f->cur], out, out_cap, &f->f64_guarded, &f->f64_patched);
}

int fmd_firdemod_get_state(fmd_firdemod* f, uint32_t channel, fmd_demod_state* state)
{
    if (!f || !state || channel >= f->C) { fmd_internal_set_err("bad argument"); return FMD_ERR_INVALID_ARG; }
    FD_ON_DEVICE(f->device);
    FD_TRY(hipDeviceSynchronize());
    int rc = fmd_internal_resolve_exc(f->d_exc, f->r.R, f->seq, f->seq, f->d_state[f->cur], nullptr, 0, &f->f64_guarded, &f->f64_patched);
    if (rc) return rc;
    FmdChanState s;
    FD_TRY(hipMemcpy(&s, f->d_state[f->cur] + channel, sizeof(s), hipMemcpyDeviceToHost));
    memset(state, 0, sizeof(*state));
    state->now_lpr = s.now_lpr;
    state->prev_lpr_index = (int32_t)(s.lpr_index_r * f->r.g);
    state->demod_pre_re = s.demod_pre_re; state->demod_pre_im = s.demod_pre_im;
    return FMD_OK;
}

// ---- checkpoint / resume of a whole bank ---------------------------------------------------------------------------
// What one call hands to the next: the sample position and the resampler phase (shared by all channels), and per
// channel the resampler accumulator + last filter output (FmdChanState) and the FIR history (the last T - 1 samples).
namespace {
struct FdCkptHeader {
    uint32_t magic, version;
    uint32_t T, M, C, Hw, shift, taps_hash, fast, slow, i0r, last_K;
    uint64_t pos;
    uint64_t checksum;                                    // FNV-1a 64 of the header (this field as 0) followed by the payload
};
static_assert(sizeof(FdCkptHeader) == 64, "blob layout (little-endian, no padding)");
constexpr uint32_t kFdCkptMagic = 0x4B434446u;            // "FDCK"
constexpr uint32_t kFdCkptVersion = 2u;                   // 2: checksum (round 5; version 1 never left the repository)
uint64_t fd_fnv64(uint64_t h, const void* data, size_t n)
{
    const uint8_t* p = static_cast<const uint8_t*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001B3ull; }
    return h;
}
uint64_t fd_ckpt_checksum(FdCkptHeader h, const void* payload, size_t n)
{
    h.checksum = 0;
    return fd_fnv64(fd_fnv64(0xCBF29CE484222325ull, &h, sizeof(h)), payload, n);
}
size_t fd_ckpt_size(const fmd_firdemod* f)
{
    return sizeof(FdCkptHeader) + (size_t)f->C * sizeof(FmdChanState) + (size_t)f->C * f->Hw * 4;
}
}  // namespace

size_t fmd_firdemod_checkpoint_size(const fmd_firdemod* f) { return f ? fd_ckpt_size(f) : 0; }

int fmd_firdemod_checkpoint(fmd_firdemod* f, void* blob, size_t cap)
{
    if (!f || !blob) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (cap < fd_ckpt_size(f)) { fmd_internal_set_err("checkpoint buffer smaller than fmd_firdemod_checkpoint_size"); return FMD_ERR_CAPACITY; }
    FD_ON_DEVICE(f->device);
    FD_TRY(hipDeviceSynchronize());
    int rc = fmd_internal_resolve_exc(f->d_exc, f->r.R, f->seq, f->seq, f->d_state[f->cur], nullptr, 0, &f->f64_guarded, &f->f64_patched);
    if (rc) return rc;
    FdCkptHeader h{kFdCkptMagic, kFdCkptVersion, f->T, f->M, f->C, f->Hw, f->shift, f->taps_hash, f->r.fast, f->r.slow, f->i0r, f->last_K, f->pos, 0ull};
    uint8_t* const payload = static_cast<uint8_t*>(blob) + sizeof(h);
    uint8_t* p = payload;
    FD_TRY(hipMemcpy(p, f->d_state[f->cur], (size_t)f->C * sizeof(FmdChanState), hipMemcpyDeviceToHost));
    p += (size_t)f->C * sizeof(FmdChanState);
    if (f->Hw) FD_TRY(hipMemcpy(p, f->d_hist[f->cur], (size_t)f->C * f->Hw * 4, hipMemcpyDeviceToHost));
    h.checksum = fd_ckpt_checksum(h, payload, fd_ckpt_size(f) - sizeof(h));
    memcpy(blob, &h, sizeof(h));
    return FMD_OK;
}

int fmd_firdemod_resume(fmd_firdemod* f, const void* blob, size_t size)
{
    if (!f || !blob) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    FdCkptHeader h;
    if (size < sizeof(h)) { fmd_internal_set_err("checkpoint truncated"); return FMD_ERR_BAD_STATE; }
    memcpy(&h, blob, sizeof(h));
    if (h.magic != kFdCkptMagic || h.version != kFdCkptVersion) { fmd_internal_set_err("not a fused-bank checkpoint (magic / version)"); return FMD_ERR_BAD_STATE; }
    if (h.T != f->T || h.M != f->M || h.C != f->C || h.Hw != f->Hw || h.shift != f->shift || h.taps_hash != f->taps_hash ||
        h.fast != f->r.fast || h.slow != f->r.slow) {
        fmd_internal_set_err("checkpoint was taken from a bank with other taps, decimation, shift, rates or channel count");
        return FMD_ERR_BAD_STATE;
    }
    if (size != fd_ckpt_size(f)) { fmd_internal_set_err("checkpoint size does not match its header"); return FMD_ERR_BAD_STATE; }
    const uint8_t* const payload = static_cast<const uint8_t*>(blob) + sizeof(h);
    if (h.checksum != fd_ckpt_checksum(h, payload, size - sizeof(h))) { fmd_internal_set_err("checkpoint damaged (checksum)"); return FMD_ERR_BAD_STATE; }
    // A blob with a valid checksum can still be hand-made: every field the kernels take on trust is range-checked, as
    // fmd_demod_set_state does for a boxcar bank.  |demod_pre| beyond the filter's own bound would take the f32 discriminator
    // out of the range in which it is exact (silently wrong audio, no status); |now_lpr| is at most one full audio group of
    // discriminator samples.
    if (h.i0r >= f->r.fr) { fmd_internal_set_err("checkpoint holds a resampler phase outside [0, rate_out / g)"); return FMD_ERR_BAD_STATE; }
    const int64_t lim_pre = f->lp_bound, lim_lpr = 16384ll * ((f->r.fr + f->r.sr - 1) / f->r.sr);
    auto within = [](int64_t v, int64_t lim) { return v >= -lim && v <= lim; };
    for (uint32_t c = 0; c < f->C; ++c) {
        FmdChanState s; memcpy(&s, payload + (size_t)c * sizeof(FmdChanState), sizeof(s));
        if (s.lpr_index_r != h.i0r) { fmd_internal_set_err("checkpoint channel disagrees with the bank's resampler phase"); return FMD_ERR_BAD_STATE; }
        if (s.prev_index != 0u || s.lp_now_re != 0 || s.lp_now_im != 0 || s.reserved != 0 || !within(s.demod_pre_re, lim_pre) ||
            !within(s.demod_pre_im, lim_pre) || !within(s.now_lpr, lim_lpr)) {
            fmd_internal_set_err("checkpoint channel state not reachable by this bank (demod_pre / now_lpr out of range)");
            return FMD_ERR_BAD_STATE;
        }
    }
    FD_ON_DEVICE(f->device);
    FD_TRY(hipDeviceSynchronize());
    // Both arrays go into the INACTIVE halves of the ping-pong; `cur` flips only once both copies have succeeded, so a
    // failing second copy cannot leave the channels' state restored and their filter history not.
    const int nxt = f->cur ^ 1;
    FD_TRY(hipMemcpy(f->d_state[nxt], payload, (size_t)f->C * sizeof(FmdChanState), hipMemcpyHostToDevice));
    if (f->Hw) FD_TRY(hipMemcpy(f->d_hist[nxt], payload + (size_t)f->C * sizeof(FmdChanState), (size_t)f->C * f->Hw * 4, hipMemcpyHostToDevice));
    FD_TRY(hipMemset(f->d_exc, 0, 16));
    FD_TRY(hipDeviceSynchronize());
    f->cur = nxt;
    f->pos = h.pos; f->i0r = h.i0r; f->last_K = h.last_K;
    return FMD_OK;
}

int fmd_firdemod_f64_stats(const fmd_firdemod* f, uint64_t* guarded, uint64_t* patched)
{
    if (!f) return FMD_ERR_INVALID_ARG;
    if (guarded) *guarded = f->f64_guarded;
    if (patched) *patched = f->f64_patched;
    return FMD_OK;
}

int fmd_firdemod_kernel_name(const fmd_firdemod* f, char* name, size_t cap)
{
    if (!f || !name || cap == 0) return FMD_ERR_INVALID_ARG;
    const uint32_t nku = f->plan.nku < 8u ? f->plan.nku : 8u;
    const bool reuse = f->M == 8u && f->plan.n_pass == 1u && !f->no_reuse;
    // The kernels live in this file's anonymous namespace, and the register form carries a third template argument -- whether
    // the launch had a per-tile table (every call of at most kFdRows tiles: a 2 MiB config-4 buffer has 26) -- so the name is
    // that of the most recent launch; before the first one, of a launch with a table.
    const bool rows = f->last_rows < 0 ? !f->no_rows : f->last_rows != 0;
    const int n = f->reg_ng ? snprintf(name, cap, "(anonymous namespace)::fmd_firdemod_reg%s%s_kernel<%u, %u, %s>", f->plan.digits == 1u ? "1" : "", f->sparse ? "s" : "", nku, f->reg_ng, rows ? "true" : "false")
                            : snprintf(name, cap, "(anonymous namespace)::fmd_firdemod_kernel<%u, %u>", nku, reuse ? 1u : 0u);
    return n < 0 || (size_t)n >= cap ? FMD_ERR_CAPACITY : FMD_OK;
}

int fmd_firdemod_tiling(const fmd_firdemod* f, uint32_t* audio_per_tile, uint32_t* lds_bytes)
{
    if (!f) return FMD_ERR_INVALID_ARG;
    uint32_t lc, rb; size_t lds = 0;
    (void)fd_sizes(f, f->r.kt, &lc, &rb, &lds);
    if (audio_per_tile) *audio_per_tile = f->r.kt;
    if (lds_bytes) *lds_bytes = (uint32_t)lds;
    return FMD_OK;
}

}  // extern "C"
