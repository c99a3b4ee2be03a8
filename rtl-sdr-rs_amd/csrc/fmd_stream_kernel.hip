// fmd_stream_kernel.hip -- register-streaming form of the fused demodulation kernel (gfx950, wave64).
//
// Same chain and same arithmetic as fmd_tile_kernel.hip (Demod::demodulate, examples/simple_fm.rs:256-269),
// different data movement: NO LDS staging and NO workgroup barriers.  Every wave owns a run of consecutive
// "rounds" of one channel-call.  A round is `kt` audio samples (kt a multiple of sr, so every round of a phase
// class has the same shape: Qt = kt*fr/sr new decimated samples, Qt <= 127) and is processed entirely inside
// one wave:
//   * each lane loads ITS OWN two decimation windows (window i and i + 64; D even => DH whole dwords each)
//     straight from HBM into VGPRs as dwordx4 + dword(s).  Lanes are 2*D bytes apart, so a wave-instruction
//     covers a contiguous 64*2*D-byte span: every 128-byte line is fetched once (measured 6.2-6.3 TB/s for this
//     shape, tools/membench.hip "window regs");
//   * the NEXT round's loads are issued before the current round is computed (two named register sets,
//     ping-pong), so a wave always has HBM requests in flight while it computes -- the bytes in flight are
//     held by VGPRs, not by LDS tiles;
//   * boxcar = xor + 2 x v_dot4_i32_i8 per dword; predecessor via DPP; discriminator as in the tile kernel;
//   * resampler: the discriminator samples of up to 64/kt rounds collect in a per-wave LDS strip (wave-private:
//     ordering only, no barrier), then one audio sample per lane.
// The first round of a call (demod_pre, clipped first window, the f64 sample :359) and the last one (tail,
// next Demod state) run the same code with clamped window indices plus a one-lane patch.
#include "fmd_device.h"
#include "fmd_kernels.h"

namespace {

using namespace fmd_dev;

#if defined(__HIP_DEVICE_COMPILE__)
#define FMD_AS_GLOBAL __attribute__((address_space(1)))
#define FMD_AS_CONSTANT __attribute__((address_space(4)))
#else
#define FMD_AS_GLOBAL
#define FMD_AS_CONSTANT
#endif

#define FMD_STRIP 640          /* int16 slots of the per-wave discriminator strip (4 rounds of <= 127 + tail + over-read) */

typedef short fmd_s2 __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_shr1_old(uint32_t old, uint32_t v)   // lane 0 keeps old[0]
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v)                     // lane 0 <- lane 63
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(fmd_s2, a), __builtin_bit_cast(fmd_s2, b), 0, false);
}
__device__ __forceinline__ uint32_t pack_lp_perm(int re, int im)
{
    return __builtin_amdgcn_perm((uint32_t)im, (uint32_t)re, 0x05040100u);
}

// polar_discriminant_fast + fast_atan2 (:377-405), branch-free; see fmd_tile_kernel.hip for the derivation.
__device__ __forceinline__ int disc_fast(uint32_t a, uint32_t b)
{
    const uint32_t a_sw = __builtin_amdgcn_alignbit(a, a, 16);
    const uint32_t b_cj = (b & 0xFFFFu) | ((0u - (b >> 16)) << 16);
    const int cr = sdot2(a, b);
    const int ci = sdot2(a_sw, b_cj);
    const uint32_t ux = (uint32_t)cr;
    const uint32_t my = (uint32_t)(ci >> 31);
    const uint32_t yabs = ((uint32_t)ci ^ my) - my;
    const uint32_t dif = ux - yabs, sum = ux + yabs;
    const bool xpos = cr >= 0;
    const int num = (int)((xpos ? dif : sum) << 12);
    const uint32_t den = xpos ? sum : yabs - ux;
    const uint32_t mn = (uint32_t)(num >> 31);
    const uint32_t unum = ((uint32_t)num ^ mn) - mn;
    uint32_t q = (uint32_t)((float)unum * __builtin_amdgcn_rcpf((float)den));
    const int rem = (int)(unum - q * den);
    q = q + (rem >= (int)den ? 1u : 0u) - (rem < 0 ? 1u : 0u);
    const uint32_t qs = (q ^ mn) - mn;
    const uint32_t angle = (xpos ? (1u << 12) : (3u << 12)) - qs;
    const uint32_t res = (angle ^ my) - my;
    return den == 0u ? 0 : (int)res;
}

// One lane's window: DH dwords (DH <= 8).
template <int DH>
struct Win { uint32_t w[DH]; };

typedef const FMD_AS_GLOBAL unsigned char* gbytes_t;

template <int DH>
__device__ __forceinline__ Win<DH> load_window(gbytes_t chan, uint32_t byte_off)
{
    Win<DH> W;
    const gbytes_t p = chan + byte_off;
    if constexpr (DH >= 4) {
        const v4u x = *(const FMD_AS_GLOBAL v4u*)p;
        W.w[0] = x.x; W.w[1] = x.y; W.w[2] = x.z; W.w[3] = x.w;
#pragma unroll
        for (int u = 4; u < DH; ++u) W.w[u] = *(const FMD_AS_GLOBAL uint32_t*)(p + 4 * u);
    } else {
#pragma unroll
        for (int u = 0; u < DH; ++u) W.w[u] = *(const FMD_AS_GLOBAL uint32_t*)(p + 4 * u);
    }
    return W;
}

// General window sum straight from global memory (the rare one-lane patches only).
__device__ __forceinline__ void global_window_sum(gbytes_t chan, int n0, int n1, int& re, int& im)
{
    int ar = 0, ai = 0;
    for (int m = n0 >> 1; n1 > n0 && m <= ((n1 - 1) >> 1); ++m) {
        const uint32_t w = *(const FMD_AS_GLOBAL uint32_t*)(chan + 4u * (uint32_t)m) ^ 0x80808080u;
        uint32_t mask = 0xFFFFFFFFu;
        if (2 * m < n0) mask = 0xFFFF0000u;
        if (2 * m + 1 >= n1) mask &= 0x0000FFFFu;
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

// Geometry of one round (wave-uniform).
struct Round {
    int jfirst;      // decimated index of window slot i = 0
    int hi;          // last valid slot (cnt - 1)
    int lo;          // first slot whose window is a plain whole-dword window (j >= 1, or j == 0 with p0 == 0)
    uint32_t k0, nk; // audio samples [k0, k0 + nk)
    int ebase;       // e(k0 + q) - jfirst = ebase + eoff(q)
    int s0;          // slot where audio sample k0 starts
    bool last;
};

__device__ __forceinline__ Round round_setup(const FmdLaunch& L, const FmdClassPlan& P, uint32_t t)
{
    const FmdTile T = fmd_tile_fast(L.r, P, L.tl, L.ns, t);
    Round R;
    R.jfirst = T.jA - 1;
    R.hi = T.jB - R.jfirst;
    R.lo = R.jfirst >= 1 ? 0 : ((P.p0 == 0 ? 0 : 1) - R.jfirst);   // skip j < 0 (and the clipped j == 0)
    if (R.lo < 0) R.lo = 0;
    R.k0 = T.k0; R.nk = T.k1 - T.k0;
    R.ebase = (int)T.eq - R.jfirst;
    R.s0 = T.jA - R.jfirst;
    R.last = T.last;
    return R;
}

template <int DH>
__global__ void __launch_bounds__(FMD_BLOCK_THREADS) fmd_demod_stream_kernel(const FmdLaunch L)
{
    constexpr int DW = DH > 0 ? DH : 1;
    __shared__ int16_t d16_all[FMD_BLOCK_THREADS / 64][FMD_STRIP];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    int16_t* const d16 = d16_all[wave];

    const uint32_t c = blockIdx.z * 65535u + blockIdx.y;
    if (c >= L.n_channels) return;
    const uint32_t cls = L.chan_class ? L.chan_class[c] : 0u;
    const FmdClassPlan P = L.cls[cls];
    const uint32_t span = blockIdx.x * (FMD_BLOCK_THREADS / 64) + wave;      // this wave's run of rounds
    const uint32_t t0 = span * L.rounds_per_wave;
    if (t0 >= P.nt) return;
    const uint32_t t1 = t0 + L.rounds_per_wave < P.nt ? t0 + L.rounds_per_wave : P.nt;

    const FmdRates& r = L.r;
    const uint32_t p0 = P.p0, hp = p0 >> 1;
    const gbytes_t chan = (gbytes_t)(uintptr_t)((uint64_t)(uintptr_t)L.iq + (uint64_t)c * L.chan_stride);
    const int byte_base = -2 * (int)p0;                       // window j starts at byte 2*D*j - 2*p0 of the channel
    const int step = 2 * (int)r.D;
    const uint32_t group_rounds = L.group_rounds;             // rounds whose audio is produced together (<= 64 / kt)

    // per-lane resampler constants: e(k0 + q) = jfirst + ebase + eoff(q), for the first round of a group
    const uint32_t q = lane;
    const int eoff = (int)(q * L.fa + fmd_udiv_small(P.er0 + q * L.fb, r.sr, L.inv_sr));
    const int eoff_prev = q ? (int)((q - 1) * L.fa + fmd_udiv_small(P.er0 + (q - 1) * L.fb, r.sr, L.inv_sr)) : 0;
    const int glen = (int)L.fa + 1;
    int16_t* const outc = L.out + (uint64_t)c * L.out_stride;

    FmdChanState st{};                                        // only the spans that touch the call's ends need it
    if (t0 == 0 || t1 == P.nt) {
        typedef const FMD_AS_CONSTANT FmdChanState* cptr_t;
        st = *((cptr_t)(uintptr_t)L.st_in + c);
    }

    // window slots of this lane in a round: i1 = lane, i2 = lane + 64, clamped into [lo, hi]
    auto slot_off = [&](const Round& RR, int i) -> uint32_t {
        const int ic = i < RR.lo ? RR.lo : (i > RR.hi ? RR.hi : i);
        return (uint32_t)(byte_base + step * (RR.jfirst + ic));
    };

    // state of the current audio group
    Round G{};                                                // geometry of the group's first round
    uint32_t g_rounds = 0, g_nk = 0;

    // One register set: the boxcar consumes the two windows first thing in a round, and the NEXT round's
    // loads are issued right behind it, so they are in flight during the discriminator + resampler (most of
    // the round) without a second set of registers.
    Round Rn = round_setup(L, P, t0);
    Win<DW> A = load_window<DW>(chan, slot_off(Rn, (int)lane));
    Win<DW> B = load_window<DW>(chan, slot_off(Rn, (int)lane + 64));
    for (uint32_t t = t0; t < t1; ++t) {
        const Round Rc = Rn;
        const bool close_group = t + 1 >= t1;
        if (g_rounds == 0) { G = Rc; g_nk = 0; }
        const int sbase = Rc.jfirst - G.jfirst;               // strip index of this round's slot 0
        const int i1 = (int)lane, i2 = i1 + 64;
        const bool odd = ((((DH & 1) ? ((uint32_t)(Rc.jfirst + i1) ^ hp) : hp)) & 1u) != 0u;   // i2 = i1 + 64: same parity
        const uint32_t wreA = odd ? FMD_W_RE_ODD : FMD_W_RE_EVEN, wreB = odd ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
        const uint32_t wimA = odd ? FMD_W_IM_ODD : FMD_W_IM_EVEN, wimB = odd ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
        int re1 = DH, im1 = 2 * (odd ? DH / 2 : (DH + 1) / 2), re2 = re1, im2 = im1;
#pragma unroll
        for (int u = 0; u < DW; ++u) {
            const uint32_t wa = A.w[u] ^ 0x80808080u, wb = B.w[u] ^ 0x80808080u;   // u8 -> s8 (b - 128)
            re1 = sdot4(wa, (u & 1) ? wreB : wreA, re1);
            im1 = sdot4(wa, (u & 1) ? wimB : wimA, im1);
            re2 = sdot4(wb, (u & 1) ? wreB : wreA, re2);
            im2 = sdot4(wb, (u & 1) ? wimB : wimA, im2);
        }
        if (t + 1 < t1) {                                    // next round's windows: in flight for the rest of this round
            Rn = round_setup(L, P, t + 1);
            A = load_window<DW>(chan, slot_off(Rn, (int)lane));
            B = load_window<DW>(chan, slot_off(Rn, (int)lane + 64));
        }
        const uint32_t pk1 = pack_lp_perm(re1, im1), pk2 = pack_lp_perm(re2, im2);
        const uint32_t prev1 = wave_shr1(pk1);
        const uint32_t prev2 = wave_shr1_old(wave_ror1(pk1), pk2);
        const int d1 = disc_fast(pk1, prev1), d2 = disc_fast(pk2, prev2);                 // (:362)
        if (lane > 0 && i1 <= Rc.hi) d16[sbase + i1] = (int16_t)d1;
        if (i2 <= Rc.hi) d16[sbase + i2] = (int16_t)d2;

        // call start: lp[-1] = demod_pre, lp[0] = clipped window + lp_now, d[0] on the f64 path (:359)
        if (Rc.jfirst <= 0 && lane == 0) {
            int r0, i0, r1, i1w, cr, ci;
            global_window_sum(chan, 0, fmd_win_end(r.D, p0, 0), r0, i0);
            r0 += st.lp_now_re; i0 += st.lp_now_im;
            global_window_sum(chan, fmd_win_begin(r.D, p0, 1), fmd_win_end(r.D, p0, 1), r1, i1w);
            if (Rc.jfirst < 0) {
                fmd_mul_conj(r0, i0, st.demod_pre_re, st.demod_pre_im, cr, ci);
                d16[sbase + 1] = (int16_t)polar_f64(cr, ci);
            }
            fmd_mul_conj(r1, i1w, r0, i0, cr, ci);
            d16[sbase + 1 - Rc.jfirst] = (int16_t)fmd_fast_atan2(ci, cr);
        }
        ++g_rounds; g_nk += Rc.nk;
        if (!close_group && g_rounds < group_rounds && !Rc.last) continue;

        // ---- resampler for the whole group: one audio sample per lane -------------------------------
        __builtin_amdgcn_wave_barrier();                     // the strip is wave-private: ordering only
        if (q < g_nk) {
            const int e = G.ebase + eoff;
            const int s = q == 0 ? G.s0 : G.ebase + eoff_prev + 1;
            int sum = (G.k0 + q == 0) ? st.now_lpr : 0;
            const int16_t* dp = d16 + s;
            const int n = e - s + 1;
#pragma clang loop vectorize(disable)
            for (int u = 0; u < glen; ++u) { const int v = dp[u]; sum += u < n ? v : 0; }
            outc[G.k0 + q] = (int16_t)fmd_sdiv_small(sum, r.R, L.inv_R);
        }
        // ---- Demod state after the call (:232-239) ------------------------------------------------
        if (Rc.last && lane == 0) {
            FmdChanState ns_;
            const int jB = (int)P.M - 1;
            const int s = P.K == 0 ? 0 : (int)fmd_audio_end(r, P.i0r, P.K - 1) + 1;
            int sum = P.K == 0 ? st.now_lpr : 0;
            for (int jj = s; jj <= jB; ++jj) sum += d16[jj - G.jfirst];
            ns_.now_lpr = sum;
            ns_.lpr_index_r = fmd_next_lpr_index_r(r, P.i0r, P.M, P.K);
            ns_.prev_index = fmd_next_prev_index(r.D, p0, L.ns);
            int tr, ti, pr, pi;
            global_window_sum(chan, fmd_win_begin(r.D, p0, (int)P.M), (int)L.ns, tr, ti);
            ns_.lp_now_re = tr; ns_.lp_now_im = ti;
            global_window_sum(chan, fmd_win_begin(r.D, p0, jB), fmd_win_end(r.D, p0, jB), pr, pi);   // M >= 2: jB >= 1
            ns_.demod_pre_re = pr; ns_.demod_pre_im = pi;
            ns_.reserved = 0;
            L.st_out[c] = ns_;
            if (L.out_len) L.out_len[c] = P.K;
        }
        __builtin_amdgcn_wave_barrier();
        g_rounds = 0;
    }
}

template <int DH>
void launch_stream(const FmdLaunch& L, dim3 g, hipStream_t stream)
{
    hipLaunchKernelGGL(fmd_demod_stream_kernel<DH>, g, dim3(FMD_BLOCK_THREADS), 0, stream, L);
}

}  // namespace

// The streaming kernel needs: whole-dword windows (even downsample; even phases are checked per call by the
// caller), a round of at most 127 new decimated samples plus the call's tail, at most 64 audio samples per
// group of rounds (one per lane), and the same exact-division ranges as the tile kernel.
bool fmd_stream_kernel_supports(const FmdRates& r)
{
    if (r.D % 2 != 0 || r.D > 16) return false;                           // DH <= 8 dwords held per window
    if (r.kt % r.sr != 0 || r.kt > 64) return false;
    const uint64_t Qt = (uint64_t)r.kt * r.fr / r.sr;
    const uint64_t tail = (r.fr + r.sr - 1) / r.sr;
    if (Qt + tail > 128) return false;                                    // slots 1 .. 127 hold the round + tail
    if ((uint64_t)r.sr * (64 + 2) >= (1u << 24)) return false;
    if ((uint64_t)(tail + 2) * 32768ull >= (1u << 24)) return false;
    if ((uint32_t)r.R >= (1u << 24)) return false;
    return true;
}

// Largest round (audio samples) the streaming kernel can take for these rates, 0 if none.
uint32_t fmd_stream_round_kt(const FmdRates& r0)
{
    FmdRates r = r0;
    uint32_t best = 0;
    for (uint32_t kt = r.sr; kt <= 64; kt += r.sr) {
        r.kt = kt;
        if (fmd_stream_kernel_supports(r)) best = kt; else if (best) break;
    }
    return best;
}

// Rounds whose audio samples are produced together: as many as fit 64 lanes and the strip.
uint32_t fmd_stream_group_rounds(const FmdRates& r)
{
    const uint64_t Qt = (uint64_t)r.kt * r.fr / r.sr;
    const uint64_t tail = (r.fr + r.sr - 1) / r.sr;
    uint32_t g = 64u / r.kt;
    while (g > 1 && g * Qt + 2 * tail + 16 > FMD_STRIP) --g;
    return g ? g : 1u;
}

hipError_t fmd_launch_stream(const FmdLaunch& L, hipStream_t stream)
{
    if (L.n_channels == 0 || L.tiles == 0 || L.rounds_per_wave == 0 || L.group_rounds == 0) return hipErrorInvalidValue;
    const uint32_t spans = (L.tiles + L.rounds_per_wave - 1) / L.rounds_per_wave;
    const uint32_t bx = (spans + FMD_BLOCK_THREADS / 64 - 1) / (FMD_BLOCK_THREADS / 64);
    const uint32_t gy = L.n_channels < 65535u ? L.n_channels : 65535u;
    const uint32_t gz = (L.n_channels + 65534u) / 65535u;
    const dim3 g(bx, gy, gz);
    switch (L.r.D / 2) {
        case 1: launch_stream<1>(L, g, stream); break;
        case 2: launch_stream<2>(L, g, stream); break;
        case 3: launch_stream<3>(L, g, stream); break;   // cfg-ref, D = 6
        case 4: launch_stream<4>(L, g, stream); break;
        case 5: launch_stream<5>(L, g, stream); break;   // 2.4 Msps, D = 10
        case 6: launch_stream<6>(L, g, stream); break;
        case 7: launch_stream<7>(L, g, stream); break;
        case 8: launch_stream<8>(L, g, stream); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
