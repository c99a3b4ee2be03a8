// fmd_tile_kernel.hip -- the production demodulation kernels for gfx950 (CDNA4, wave64).
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
// A tile is `kt` consecutive audio samples of one channel-call; its geometry comes from host-made
// per-phase-class plans (FmdClassPlan, fmd_index.h), so the device only does multiply-adds.
//   tile_body (shared):
//     * boxcar: for an even downsample a window is DH whole dwords, 3 VALU ops per dword (xor, 2 x dot4)
//       with per-lane weight registers that already carry the rotation sign of the dword parity;
//     * predecessor sample from the neighbouring lane (DPP wave_shr:1): a wave-round is 63 new windows
//       + 1 overlap, so there is no LDS exchange and no barrier between boxcar and discriminator;
//     * discriminator: complex multiply by 2 x v_dot2_i32_i16 on packed (re, im); branch-free
//       fast_atan2 with an exact f32-reciprocal divide;
//     * resampler: one audio sample per lane from the tile's discriminator samples in LDS.
//   fmd_demod_persist_kernel (default): grid = CUs x resident blocks; each block walks tiles
//     lin, lin + G, ... and keeps the NEXT tile's 16-byte loads in flight in registers while it
//     computes the current one (issue early / write late), so HBM requests never stop during compute.
//   fmd_demod_tile_kernel: one block per tile, LDS-DMA staging (global_load_lds_dwordx4); any tile size.
#include "fmd_device.h"
#include "fmd_kernels.h"

namespace {

using namespace fmd_dev;

// Ablation switches exist only in -DFMD_EXPERIMENT tuning builds (libfmd_hip_exp.so); the shipped
// library compiles them to `false`.
#ifdef FMD_EXPERIMENT
#define FMD_ABLATE(bit) ((L.dbg >> (bit)) & 1u)
#else
#define FMD_ABLATE(bit) false
#endif

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

// Where a tile's bytes are and where they land in LDS (all wave-uniform).
struct TileCtx {
    FmdTile T;
    uint64_t a0;        // 16-byte aligned global address of the first staged chunk
    uint32_t nchunks;   // 16-byte chunks staged
    int wofs;           // LDS dword index of the call's dword 0 (may be negative)
    int jfirst, cnt;    // decimated samples jfirst .. jfirst + cnt - 1 are formed (jfirst == -1: demod_pre)
    uint32_t c, cls;
    bool valid;         // false: empty grid slot (t >= the class's tile count)
    bool whole;         // every staged chunk lies inside the input array
};

__device__ __forceinline__ TileCtx tile_setup(const FmdLaunch& L, uint32_t c, uint32_t t)
{
    TileCtx X;
    X.c = c;
    X.cls = L.chan_class ? L.chan_class[c] : 0u;
    const FmdClassPlan& P = L.cls[X.cls];
    X.valid = t < P.nt;
    X.T = fmd_tile_fast(L.r, P, L.Qt, L.ns, X.valid ? t : 0u);
    X.jfirst = X.T.jA - 1;
    X.cnt = X.T.jB - X.jfirst + 1;
    const uint64_t gbase = (uint64_t)(uintptr_t)L.iq + (uint64_t)c * L.chan_stride;
    const uint64_t gLo = gbase + 2ull * (uint32_t)X.T.nLo;
    const uint64_t gHi = gbase + 2ull * (uint32_t)X.T.nHi;
    X.a0 = gLo & ~15ull;
    X.nchunks = (uint32_t)((gHi - X.a0 + 15) >> 4);
    X.wofs = (int)((int64_t)(gbase - X.a0) >> 2);
    X.whole = X.a0 + 16ull * X.nchunks <= (uint64_t)(uintptr_t)L.iq + L.total_bytes;
    return X;
}

// Synchronous staging for the one tile whose last chunk crosses the end of the input array
// (array sizes are multiples of 8, chunks of 16).
__device__ __forceinline__ void stage_slow(const FmdLaunch& L, const TileCtx& X, unsigned char* smem, uint32_t tid)
{
    const uint64_t gend = (uint64_t)(uintptr_t)L.iq + L.total_bytes;
    for (uint32_t i = tid; i < X.nchunks; i += FMD_BLOCK_THREADS) {
        const uint64_t a = X.a0 + 16ull * i;
        uint4 v;
        if (a + 16 <= gend) v = *reinterpret_cast<const uint4*>((uintptr_t)a);
        else { const uint2 h = *reinterpret_cast<const uint2*>((uintptr_t)a); v = make_uint4(h.x, h.y, 0u, 0u); }
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
}

// Everything after the tile's bytes are visible in LDS.  Contains one __syncthreads().
template <int DH>
__device__ __forceinline__ void tile_body(const FmdLaunch& L, const TileCtx& X, unsigned char* smem)
{
    const FmdRates& r = L.r;
    const FmdClassPlan& P = L.cls[X.cls];
    const FmdTile& T = X.T;
    const uint32_t glen = L.fa + 1u;           // a resampler group spans fa or fa+1 discriminator samples
    const uint32_t d16_bytes = (2u * (L.lp_cap + glen + 1u) + 15u) & ~15u;
    const uint32_t* const raw_w = reinterpret_cast<const uint32_t*>(smem);
    int16_t* const d16 = reinterpret_cast<int16_t*>(smem + L.raw_cap);
    uint32_t* const last_lp = reinterpret_cast<uint32_t*>(smem + L.raw_cap + d16_bytes);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t p0 = P.p0, c = X.c;
    const int jfirst = X.jfirst, cnt = X.cnt, wofs = X.wofs;
    const bool fastwin = DH > 0 && (p0 & 1u) == 0u;          // windows are DH whole dwords
    FmdChanState st{};
    if (jfirst <= 0 || T.k0 == 0 || T.last) st = L.st_in[c]; // only call-start and call-end tiles need the state

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

    // ---- boxcar + discriminator, 63 new decimated samples per wave-round ------------------------
    for (int base = (int)wave * 63; base < cnt; base += NW * 63) {
        const int i = base + (int)lane;                      // lane 0 re-does the previous round's last window
        const bool act = i < cnt;
        const int j = jfirst + i;
        int re = 0, im = 0;
        if (FMD_ABLATE(1)) { re = (int)lane; im = j & 255; }              // ablation: no window
        else if (fastwin) {
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
            int pcm;
            if (FMD_ABLATE(0)) { pcm = (int)(pk ^ prev); cr = 1; cim = 0; }     // ablation: no discriminator
            else pcm = disc_fast(pk, prev, cr, cim);                             // (:362)
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
        if (FMD_ABLATE(2)) { outc[T.k0 + q] = d16[q + 1]; continue; }           // ablation: no resampler
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

__device__ __forceinline__ bool tile_fits(const FmdLaunch& L, const TileCtx& X, uint32_t tid)
{
    if ((uint32_t)X.cnt > L.lp_cap || X.nchunks * 16u > L.raw_cap) {
        if (tid == 0) atomicOr(L.err, (uint32_t)X.cnt > L.lp_cap ? FMD_DEVERR_LP_CAP : FMD_DEVERR_RAW_CAP);
        return false;
    }
    return true;
}

// ---- one block per tile, LDS-DMA staging ------------------------------------------------------------
template <int DH>
__global__ void __launch_bounds__(FMD_BLOCK_THREADS) fmd_demod_tile_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const uint32_t c = blockIdx.z * 65535u + blockIdx.y;
    if (c >= L.n_channels) return;
    const TileCtx X = tile_setup(L, c, blockIdx.x);
    if (!X.valid || !tile_fits(L, X, tid)) return;
    if (X.whole) {
        // global_load_lds_dwordx4: 1 KiB per wave-instruction straight into the tile image, destination =
        // wave-uniform base (M0) + lane * 16; no VGPR round trip, no ds_write pass, one wait for all.
        const unsigned char* src = reinterpret_cast<const unsigned char*>((uintptr_t)X.a0) + 16u * tid;
        unsigned char* dst = smem + 1024u * wave;
        const uint32_t nfull = X.nchunks >> 8, ntail = X.nchunks & 255u;
        for (uint32_t l = 0; l < nfull; ++l) lds_dma16(src + 4096u * l, dst + 4096u * l);
        if (tid < ntail) lds_dma16(src + 4096u * nfull, dst + 4096u * nfull);
    } else {
        stage_slow(L, X, smem, tid);
    }
    if (FMD_ABLATE(3)) {                                     // ablation: staging skeleton only
        __syncthreads();
        if (tid == 0) L.out[(uint64_t)c * L.out_stride + X.T.k0] = (int16_t)reinterpret_cast<uint32_t*>(smem)[blockIdx.x & 63u];
        return;
    }
    __syncthreads();
    tile_body<DH>(L, X, smem);
}

// ---- persistent blocks, next tile's loads in flight during compute ---------------------------------------
// FMD_PERSIST_LOADS (= 5) x 16 B per lane as NAMED members: hipcc keeps these in VGPRs, whereas a
// `uint4 v[5]` that is conditionally (re)defined across loop iterations is demoted to scratch.
struct TileRegs { uint4 a, b, c, d, e; };
static_assert(FMD_PERSIST_LOADS == 5, "TileRegs holds five 16-byte chunks per lane");

__device__ __forceinline__ TileRegs issue_loads(const TileCtx& X, uint32_t tid)
{
    const uint4* src = reinterpret_cast<const uint4*>((uintptr_t)X.a0);
    const uint32_t lastc = X.nchunks - 1u;                   // surplus lanes re-read the last chunk (in bounds)
    TileRegs v;
    uint32_t i;
    i = tid;                            v.a = src[i < lastc ? i : lastc];
    i = tid + 1u * FMD_BLOCK_THREADS;   v.b = src[i < lastc ? i : lastc];
    i = tid + 2u * FMD_BLOCK_THREADS;   v.c = src[i < lastc ? i : lastc];
    i = tid + 3u * FMD_BLOCK_THREADS;   v.d = src[i < lastc ? i : lastc];
    i = tid + 4u * FMD_BLOCK_THREADS;   v.e = src[i < lastc ? i : lastc];
    return v;
}

__device__ __forceinline__ void write_loads(const TileCtx& X, const TileRegs& v, unsigned char* smem, uint32_t tid)
{
    uint4* dst = reinterpret_cast<uint4*>(smem) + tid;
    const uint32_t n = X.nchunks;
    if (tid < n) dst[0] = v.a;
    if (tid + 1u * FMD_BLOCK_THREADS < n) dst[1 * FMD_BLOCK_THREADS] = v.b;
    if (tid + 2u * FMD_BLOCK_THREADS < n) dst[2 * FMD_BLOCK_THREADS] = v.c;
    if (tid + 3u * FMD_BLOCK_THREADS < n) dst[3 * FMD_BLOCK_THREADS] = v.d;
    if (tid + 4u * FMD_BLOCK_THREADS < n) dst[4 * FMD_BLOCK_THREADS] = v.e;
}

template <int DH>
__global__ void __launch_bounds__(FMD_BLOCK_THREADS) fmd_demod_persist_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t total = L.tiles * L.n_channels, G = gridDim.x;
    uint32_t lin = blockIdx.x;
    if (lin >= total) return;
    const uint32_t dq = G / L.tiles, dr = G - dq * L.tiles;   // (c, t) step of one grid stride
    uint32_t c = lin / L.tiles, t = lin - c * L.tiles;

    constexpr uint32_t kMaxChunks = FMD_PERSIST_LOADS * FMD_BLOCK_THREADS;
    TileRegs v{};
    {
        const TileCtx first = tile_setup(L, c, t);
        if (first.valid && first.whole && first.nchunks <= kMaxChunks) v = issue_loads(first, tid);
    }
    for (;;) {
        // Only (c, t) and the prefetched registers are carried across iterations; the tile context is a few
        // scalar multiply-adds and is recomputed rather than kept live.
        const TileCtx cur = tile_setup(L, c, t);
        if (cur.valid && !tile_fits(L, cur, tid)) return;
        lin += G; c += dq; t += dr;
        if (t >= L.tiles) { t -= L.tiles; ++c; }
        const bool more = lin < total;
        if (cur.valid) {
            if (cur.whole && cur.nchunks <= kMaxChunks) write_loads(cur, v, smem, tid);   // loads issued a tile ago
            else stage_slow(L, cur, smem, tid);
        }
        if (more) {
            const TileCtx nxt = tile_setup(L, c, t);
            if (nxt.valid && nxt.whole && nxt.nchunks <= kMaxChunks) v = issue_loads(nxt, tid);   // in flight during compute
        }
        if (cur.valid) {
            __syncthreads();
            if (!FMD_ABLATE(3)) tile_body<DH>(L, cur, smem);
            __syncthreads();                                          // LDS free for the next tile
        }
        if (!more) break;
    }
}

template <int DH>
void launch_one(const FmdLaunch& L, dim3 g, size_t lds, hipStream_t stream)
{
    hipLaunchKernelGGL(fmd_demod_tile_kernel<DH>, g, dim3(FMD_BLOCK_THREADS), lds, stream, L);
}

template <int DH>
int persist_blocks_per_cu(size_t lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fmd_demod_persist_kernel<DH>, FMD_BLOCK_THREADS, lds) !=
        hipSuccess) nb = 0;
    return nb;
}

template <int DH>
void launch_persist(const FmdLaunch& L, uint32_t blocks, size_t lds, hipStream_t stream)
{
    hipLaunchKernelGGL(fmd_demod_persist_kernel<DH>, dim3(blocks), dim3(FMD_BLOCK_THREADS), lds, stream, L);
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

int fmd_persist_blocks_per_cu(const FmdLaunch& L)
{
    const size_t lds = fmd_tile_lds_bytes(L);
    const uint32_t dh = (L.r.D % 2 == 0) ? L.r.D / 2 : 0;
    switch (dh) {
        case 1: return persist_blocks_per_cu<1>(lds);
        case 2: return persist_blocks_per_cu<2>(lds);
        case 3: return persist_blocks_per_cu<3>(lds);
        case 4: return persist_blocks_per_cu<4>(lds);
        case 5: return persist_blocks_per_cu<5>(lds);
        default: return persist_blocks_per_cu<0>(lds);
    }
}

hipError_t fmd_launch_tile(const FmdLaunch& L, hipStream_t stream)
{
    const size_t lds = fmd_tile_lds_bytes(L);
    if (L.n_channels == 0 || L.tiles == 0) return hipErrorInvalidValue;
    const uint32_t dh = (L.r.D % 2 == 0) ? L.r.D / 2 : 0;
    if (L.persist_blocks) {
        const uint64_t total = (uint64_t)L.tiles * L.n_channels;
        if (total > 0xFFFFFFFFull) return hipErrorInvalidValue;
        const uint32_t blocks = total < L.persist_blocks ? (uint32_t)total : L.persist_blocks;
        switch (dh) {
            case 1: launch_persist<1>(L, blocks, lds, stream); break;
            case 2: launch_persist<2>(L, blocks, lds, stream); break;
            case 3: launch_persist<3>(L, blocks, lds, stream); break;   // cfg-ref, D = 6
            case 4: launch_persist<4>(L, blocks, lds, stream); break;
            case 5: launch_persist<5>(L, blocks, lds, stream); break;   // 2.4 Msps, D = 10
            default: launch_persist<0>(L, blocks, lds, stream); break;  // generic windows
        }
        return hipGetLastError();
    }
    const uint32_t gy = L.n_channels < 65535u ? L.n_channels : 65535u;
    const uint32_t gz = (L.n_channels + 65534u) / 65535u;
    const dim3 g(L.tiles, gy, gz);
    switch (dh) {
        case 1: launch_one<1>(L, g, lds, stream); break;
        case 2: launch_one<2>(L, g, lds, stream); break;
        case 3: launch_one<3>(L, g, lds, stream); break;
        case 4: launch_one<4>(L, g, lds, stream); break;
        case 5: launch_one<5>(L, g, lds, stream); break;
        default: launch_one<0>(L, g, lds, stream); break;
    }
    return hipGetLastError();
}
