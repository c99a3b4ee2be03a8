// fmd_tile_body.h -- device code of the production demodulation kernels for gfx950 (CDNA4, wave64); the kernels are
// instantiated in fmd_tile_lds_*.hip (LDS-DMA staging, one kernel per downsample factor) and fmd_tile_stream.hip
// (register streaming, downsample 2 and 4), and launched by fmd_tile_launch.hip.
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
// per-phase-class plans and per-handle tiling constants (FmdClassPlan, FmdTiling, fmd_index.h), so the
// device does multiply-adds and at most one small division per tile.
//   tile_body (shared):
//     * boxcar: for an even downsample <= 14 a window is DH whole dwords, 3 VALU ops per dword (xor, 2 x dot4)
//       with weight registers that already carry the rotation sign of the dword parity; any other
//       downsample / phase runs the same loop with the window's half dwords masked out of the weights;
//     * predecessor sample from the neighbouring lane (DPP wave_shr:1): a wave-round is 127 new windows
//       + 1 overlap, two per lane, so there is no LDS exchange and no barrier between boxcar and discriminator;
//     * discriminator: a * conj(b) in exact f32 (downsample <= 16) or by 2 x v_dot2_i32_i16 on packed (re, im);
//       branch-free fast_atan2 with an exact reciprocal divide;
//     * resampler: one audio sample per lane from the tile's discriminator samples in LDS.
#pragma once

#include "fmd_device.h"
#include "fmd_kernels.h"

#include <type_traits>

namespace fmd_tk {

using namespace fmd_dev;

constexpr int kThreads = FMD_BLOCK_THREADS;     // one 256-thread block (4 waves) per tile
constexpr int NW = kThreads / 64;
// New decimated samples per wave-round: 128 windows minus the overlap.  A block's round stride NW * RS windows keeps a
// lane's rotation phase fixed (a lane's windows are an even number of samples apart).
constexpr int RS = 127;

// Ablation switches exist only in -DFMD_EXPERIMENT tuning builds (libfmd_hip_exp.so); the shipped
// library compiles them to `false`.
#ifdef FMD_EXPERIMENT
#define FMD_ABLATE(bit) ((L.dbg >> (bit)) & 1u)
#define FMD_F64_SKEW (L.f64_skew)      /* test hook: make the kernel's value of a guarded f64 sample wrong on purpose */
#else
#define FMD_ABLATE(bit) false
#define FMD_F64_SKEW 0
#endif
// Explicit address spaces exist only in the device pass (the host pass parses the same bodies).
#if defined(__HIP_DEVICE_COMPILE__)
#define FMD_AS_GLOBAL __attribute__((address_space(1)))
#define FMD_AS_CONSTANT __attribute__((address_space(4)))
#define FMD_AS_LDS __attribute__((address_space(3)))
#else
#define FMD_AS_GLOBAL
#define FMD_AS_CONSTANT
#define FMD_AS_LDS
#endif

// Opens a region that only `special` tiles enter (TileCtx::special_bits, wave-uniform).  The flag word goes through an empty
// asm at every site: the test is then s_and + s_cbranch_scc on the spot -- hipcc otherwise keeps ONE 64-bit mask of the
// condition alive across the whole body and folds `if (special) if (cond && tid == 0)` back into a vector condition, which puts
// the mask bookkeeping (and the scalar loads of `cond`) on every tile's path again.
__device__ __forceinline__ bool fmd_special(uint32_t bits, bool rich)
{
    if (!rich) return true;                                  // the other prologues test every condition themselves
    asm volatile("" : "+s"(bits));
    return bits != 0u;
}
#define FMD_SPECIAL_ONLY(X) if (fmd_special((X).special_bits, (X).rich))

// Cache policy of the staging loads (aux of global_load_lds): 2 = nt (read once, do not keep).  Settled-clock A/B at the
// bench configuration: nt -0.8 % against the default policy (0), equal elsewhere; on some boxes the default is 15 % slower
// (profiles/HISTORY.md).
constexpr int kDmaAux = 2;

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// lane l <- v[l-1] for l >= 1; lane 0 keeps old[0] (no source lane, bound_ctrl off).
__device__ __forceinline__ uint32_t wave_shr1_old(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// wave_shr:1 with a register that is DEAD at the call site as `old` (lane 0 keeps it: only for results lane 0 never uses):
// the destination is tied to `old`, so the v_mov 0 of wave_shr1 disappears.  bound_ctrl stays off (see fmd_device.h).
__device__ __forceinline__ uint32_t wave_shr1_dead(uint32_t dead, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)dead, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// lane l <- v[(l + 63) % 64]: lane 0 receives lane 63.
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
}
// The same into a register that is dead at the call site (every lane is written: `old` only names the destination).
__device__ __forceinline__ uint32_t wave_ror1_dead(uint32_t dead, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)dead, (int)v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
}

// Sum of one audio group: FA samples plus one optional (low_pass_real, :411-415).
// (hipcc merges the FA + 1 int16_t loads into ONE ds_read_b64 / b96 / b128 at a 2-byte aligned address.  Round 4 measured
//  the alternative -- aligned dword pairs summed by v_dot2_i32_i16 against 0 / 1 half-word weights -- 0.8 ... 1.3 % SLOWER at
//  downsample 4, 6, 7 and 18 % slower at downsample 1, where the resampler dominates: profiles/r04_experiments.md.)
template <int FA>
__device__ __forceinline__ int group_sum(const int16_t* __restrict__ dp, bool extra)
{
    int sum = 0;
#pragma unroll
    for (int i = 0; i < FA; ++i) sum += dp[i];
    const int v = dp[FA];
    return sum + (extra ? v : 0);
}

// (re, im) -> re | im << 16 in one v_perm_b32.
__device__ __forceinline__ uint32_t pack_lp_perm(int re, int im)
{
    return __builtin_amdgcn_perm((uint32_t)im, (uint32_t)re, 0x05040100u);
}

__device__ __forceinline__ void lds_dma16(const unsigned char* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, kDmaAux);
}

// Where a tile's bytes are and where they land in LDS (all wave-uniform).
struct TileCtx {
    FmdTile T;
    uint64_t gbase;     // global address of the channel-call's first byte
    uint64_t a0;        // 16-byte aligned global address of the first staged chunk
    uint32_t nchunks;   // 16-byte chunks staged
    int wofs;           // LDS dword index of the call's dword 0 (may be negative)
    int jfirst, cnt;    // decimated samples jfirst .. jfirst + cnt - 1 are formed (jfirst == -1: demod_pre)
    uint32_t c, cls;
    bool valid;         // false: empty grid slot (t >= the class's tile count)
    bool whole;         // every staged chunk lies inside the input array
    // Table form (FAST == 2, fast_ctx<2>): the row's ready-made values.  `rich` is a compile-time fact of each kernel
    // instantiation after inlining, so every `X.rich ? row value : expression` below folds to one side.
    bool rich = false;
    int wbase = 0, s00 = 0;     // FmdTileRow::wbase, ::s00
    uint32_t par = 0, nk = 0;   // FmdTileRow::par, ::nk
    bool need_state = false;    // FMD_ROW_STATE
    // Does the tile have ANY of the one-lane jobs around the rounds and the resampler -- call-start patch, block_len boundaries,
    // first audio sample, guard records, state epilogue?  A common tile has none, and in the table form it learns so from ONE
    // flag test per group of them (each `cond && tid == 0` region costs ~5 scalar instructions of mask bookkeeping whether or
    // not it is entered; round 4 counted ~75 per wave for them).  The other prologues test every condition as before.
    uint32_t special_bits = 1u;
};

__device__ __forceinline__ TileCtx tile_setup(const FmdLaunch& L, uint32_t c, uint32_t t)
{
    TileCtx X;
    X.c = c;
    X.cls = L.chan_class ? L.chan_class[c] : 0u;
    const FmdClassPlan& P = L.cls[X.cls];
    X.valid = t < P.nt;
    X.T = fmd_tile_fast(L.r, P, L.tl, L.ns, X.valid ? t : 0u);
    X.jfirst = X.T.jA - 1;
    X.cnt = X.T.jB - X.jfirst + 1;
    const uint64_t gbase = (uint64_t)(uintptr_t)L.iq + (uint64_t)c * L.chan_stride;
    X.gbase = gbase;
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
    for (uint32_t i = tid; i < X.nchunks; i += kThreads) {
        const uint64_t a = X.a0 + 16ull * i;
        uint4 v;
        if (a + 16 <= gend) v = *reinterpret_cast<const uint4*>((uintptr_t)a);
        else { const uint2 h = *reinterpret_cast<const uint2*>((uintptr_t)a); v = make_uint4(h.x, h.y, 0u, 0u); }
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
}

// Rare path (one lane): walk the tile's f64 samples again -- the call-start sample and, in block_len mode, every
// reference-call boundary inside the tile -- and append a record for each guarded one.  Recomputing the products
// here keeps the common path free of bookkeeping (no LDS list, no live registers).  Inlined on purpose (it reads the
// launch descriptor, which must stay in SGPRs); only polar_f64 and exc_emit, which take plain values, are calls.
__device__ __forceinline__ void tile_exc_flush(const FmdLaunch& L, const TileCtx& X, const FmdChanState& st,
                                               const uint32_t* raw_w, const int16_t* d16)
{
    const FmdRates& r = L.r;
    const FmdClassPlan& P = L.cls[X.cls];
    const FmdTile& T = X.T;
    const uint32_t p0 = P.p0;
    const int jfirst = X.jfirst, wofs = X.wofs;
    bool g;
    if (jfirst < 0) {
        int r0, i0, cr, ci;
        lds_window_sum(raw_w, wofs, 0, fmd_win_end(r.D, p0, 0), r0, i0);
        r0 += st.lp_now_re; i0 += st.lp_now_im;
        fmd_mul_conj(r0, i0, st.demod_pre_re, st.demod_pre_im, cr, ci);
        (void)polar_f64(cr, ci, L.f64_guard, g);
        if (g) exc_emit(exc_args(L, X.c, P.i0r, P.K, st.now_lpr, d16, jfirst), 0, cr, ci);
    }
    if (L.block_ns) {
        const uint32_t D = r.D, nb = L.block_ns;
        const uint32_t lo = (uint32_t)(T.jA > 1 ? T.jA : 1) * D;
        uint32_t b = lo > p0 ? (lo - p0 + nb - 1u) / nb : 1u;
        if (b == 0u) b = 1u;
        for (; (uint64_t)b * nb < L.ns; ++b) {
            const int j = (int)((p0 + b * nb) / D);
            if (j > T.jB) break;
            if (j < T.jA) continue;
            int ar, ai, br, bi, cr, ci;
            lds_window_sum(raw_w, wofs, fmd_win_begin(D, p0, j), fmd_win_end(D, p0, j), ar, ai);
            lds_window_sum(raw_w, wofs, fmd_win_begin(D, p0, j - 1), fmd_win_end(D, p0, j - 1), br, bi);
            if (j - 1 == 0) { br += st.lp_now_re; bi += st.lp_now_im; }
            fmd_mul_conj(ar, ai, br, bi, cr, ci);
            (void)polar_f64(cr, ci, L.f64_guard, g);
            if (g) exc_emit(exc_args(L, X.c, P.i0r, P.K, st.now_lpr, d16, jfirst), j, cr, ci);
        }
    }
}

// Masked-window rounds with the dword count known at compile time (odd downsample 3 ... 15, 14, or an even one at an
// odd boxcar phase): a window of D samples covers NDW dwords of which the first and / or the last counts only
// half.  Which half is fixed per lane (a lane's windows are an even number of samples apart), so the masks fold
// into 2 * NDW per-lane weight registers and the body is the same 3 VALU instructions per dword as the whole-dword
// loop -- no run-time trip counts, no per-dword branches.  (Round 1 ran these through the run-time loop below:
// downsample 7 sat at 57 % of the HBM spec against 70 % for 6 and 8.)
template <int NDW, bool BIAS, bool NOWRAP>
__device__ __forceinline__ void masked_rounds(const uint32_t* __restrict__ raw_w, int16_t* __restrict__ d16, int wofs, int s00,
                                              int D, int cnt, uint32_t lane, uint32_t wave, uint32_t wreA, uint32_t wreB,
                                              uint32_t wimA, uint32_t wimB, uint32_t mf, uint32_t ml, int cre, int cim, bool smallD)
{
    uint32_t wr[NDW], wi[NDW];
#pragma unroll
    for (int u = 0; u < NDW; ++u) {
        const uint32_t m = (u == 0 ? mf : 0xFFFFFFFFu) & (u == NDW - 1 ? ml : 0xFFFFFFFFu);
        wr[u] = ((u & 1) ? wreB : wreA) & m;
        wi[u] = ((u & 1) ? wimB : wimA) & m;
    }
    const int last = cnt - 1;
    // A lane's window moves D * NW * RS samples = D * NW * RS / 2 dwords per round (NW * RS is even), and its second window
    // lies 64 D samples = 32 D dwords behind the first: one pointer, advanced by a constant, and a constant offset.
    // (`ds_read2_b32` carries dword offsets up to 255: beyond downsample 7 the second window gets a pointer of its own.)
    const uint32_t* __restrict__ pa = raw_w + (uint32_t)(wofs + ((s00 + D * ((int)wave * RS + (int)lane)) >> 1));
    const bool far = 32 * D + NDW > 256;
    const uint32_t* __restrict__ pb_far = pa + 32 * D;
    // One wave-round; FULL: all 128 windows of the round lie inside the tile, so the stores need no per-lane range test.
    auto round = [&](int base, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int i1 = base + (int)lane, i2 = i1 + 64;
        const uint32_t* __restrict__ pb = far ? pb_far : pa + 32 * D;
        int re1 = cre, im1 = cim, re2 = cre, im2 = cim;
        uint32_t dead1 = 0, dead2 = 0;                       // the last sign-flipped dwords: dead after the dot products
#pragma unroll
        for (int u = 0; u < NDW; ++u) {
            const uint32_t wa = pa[u] ^ 0x80808080u, wb = pb[u] ^ 0x80808080u;           // u8 -> s8 (b - 128)
            if (u == 0) {
                re1 = sdot4_init(wa, wr[0], cre); im1 = sdot4_init(wa, wi[0], cim);
                re2 = sdot4_init(wb, wr[0], cre); im2 = sdot4_init(wb, wi[0], cim);
            } else {
                re1 = sdot4(wa, wr[u], re1); im1 = sdot4(wa, wi[u], im1);
                re2 = sdot4(wb, wr[u], re2); im2 = sdot4(wb, wi[u], im2);
            }
            dead1 = wa; dead2 = wb;
        }
        const uint32_t pk1 = pack_lp_perm(re1, im1), pk2 = pack_lp_perm(re2, im2);
        const uint32_t prev1 = wave_shr1_dead(dead1, pk1);                                  // lane 0's is never used
        const uint32_t prev2 = wave_shr1_old(wave_ror1_dead(dead2, pk1), pk2);               // lane 0 <- first window of lane 63
        if (lane > 0 && (FULL || i1 < cnt)) d16[i1] = (int16_t)(smallD ? disc_f32<BIAS, NOWRAP, true>(pk1, prev1) : disc_nosel(pk1, prev1));   // smallD: wave-uniform
        if (FULL || i2 < cnt) d16[i2] = (int16_t)(smallD ? disc_f32<BIAS, NOWRAP, true>(pk2, prev2) : disc_nosel(pk2, prev2));
        pa += D * (NW * RS / 2);
        if (far) pb_far += D * (NW * RS / 2);
    };
    int base = (int)wave * RS;
    for (; base + 128 <= cnt; base += NW * RS) round(base, std::true_type{});
    for (; base < last; base += NW * RS) round(base, std::false_type{});
}

// Downsample 1: a "window" is one sample, half a dword.  Lane l of a round owns the ADJACENT samples base + 2 l and
// base + 2 l + 1 -- one dword of the tile image, or the high half of one and the low half of the next (`v_alignbit`) when the
// round starts at an odd sample; which of the two is the same for every lane and round of a wave (a round advances by an even
// number of samples).  The rotation phase of a lane's samples is (s + 2 l) mod 4, so the byte weights and the additive
// constants (+1 where the weight is +1: the `255 - x` negation has none) are per-lane registers, set up once per wave; the
// rest is the adjacent-window form of the even factors: components in f32 off biased sums, the second sample's predecessor
// the lane's own first one, the first one's from lane l - 1 by DPP.  (The masked-window rounds did this factor with two
// dword reads, two packs and three DPP moves per lane-pair and the packed complex product: 31 instead of 26 vector
// instructions per sample.)
__device__ __forceinline__ void d1_pair_rounds(const uint32_t* __restrict__ raw_w, int16_t* __restrict__ d16, int wofs, int s00, int cnt,
                                               uint32_t lane, uint32_t wave)
{
    const int last = cnt - 1;
    int base = (int)wave * RS;
    const int s_w = s00 + base;                              // call sample of window `base` (wave-uniform; may be negative in tile 0)
    const bool odd = (s_w & 1) != 0;                         // wave-uniform
    const uint32_t ph1 = (uint32_t)(s_w + 2 * (int)lane) & 3u;               // the second sample's phase is (ph1 + 1) & 3
    // 16-bit weight pairs (I byte, Q byte) by phase: re = +I, -Q, -I, +Q (0x0001, 0xFF00, 0x00FF, 0x0100); im = +Q, +I, -Q, -I
    // (0x0100, 0x0001, 0xFF00, 0x00FF) -- as 8-byte tables these are the dword weights themselves (EVEN, ODD), and one v_perm_b32
    // picks a phase's pair (round 6: it was a chain of compares and selects per weight).  The second sample's tables are rotated by
    // one phase and land in the upper half.  The additive constants (+1 where the weight is +1) come out of byte tables the same way.
    const uint32_t selL = ph1 * 0x0202u + 0x0C0C0100u;                       // bytes (2 ph, 2 ph + 1, zero, zero)
    const uint32_t selH = (selL << 16) | 0x0C0Cu;                            // bytes (zero, zero, 2 ph, 2 ph + 1)
    const uint32_t selB = ph1 | 0x07060500u;                                 // byte 0: table entry ph; bytes 1 ... 3: kSumBias's
    const uint32_t r1 = __builtin_amdgcn_perm(FMD_W_RE_ODD, FMD_W_RE_EVEN, selL), m1 = __builtin_amdgcn_perm(FMD_W_IM_ODD, FMD_W_IM_EVEN, selL);
    // phase ph1 + 1: entries (1, 2, 3, 0) = 0xFF00, 0x00FF, 0x0100, 0x0001 (re), 0x0001, 0xFF00, 0x00FF, 0x0100 (im)
    const uint32_t r2 = __builtin_amdgcn_perm(0x00010100u, 0x00FFFF00u, selH), m2 = __builtin_amdgcn_perm(0x010000FFu, 0xFF000001u, selH);
    static_assert((kSumBias & 0xFF) == 0, "the table byte replaces the bias's low byte");
    // re gets +1 at phases 0 and 3, im at phases 0 and 1; the second sample's tables rotated by one phase
    const int bre1 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, 0x01000001u, selB), bim1 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, 0x00000101u, selB);
    const int bre2 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, 0x01010000u, selB), bim2 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, 0x01000001u, selB);
    const uint32_t* __restrict__ pa = raw_w + (uint32_t)(wofs + (s_w >> 1) + (int)lane);
    auto round = [&](int b, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int i1 = b + 2 * (int)lane, i2 = i1 + 1;
        uint32_t w = pa[0];
        if (odd) w = __builtin_amdgcn_alignbit(pa[1], w, 16);    // (high half of pa[0], low half of pa[1])
        w ^= 0x80808080u;                                    // u8 -> s8 (b - 128)
        const int re1 = sdot4_init(w, r1, bre1), im1 = sdot4_init(w, m1, bim1);
        const int re2 = sdot4_init(w, r2, bre2), im2 = sdot4_init(w, m2, bim2);
        const float ar1 = sum_to_f32(re1), ai1 = sum_to_f32(im1), ar2 = sum_to_f32(re2), ai2 = sum_to_f32(im2);
        const float br1 = u2f(wave_shr1_dead(w, f2u(ar2))), bi1 = u2f(wave_shr1(f2u(ai2)));   // second sample of lane l - 1
        if (lane > 0 && (FULL || i1 < cnt)) d16[i1] = (int16_t)disc_f32_c<true, true>(ar1, ai1, br1, bi1);   // (no i32 wrap at this factor)
        if (FULL || i2 < cnt) d16[i2] = (int16_t)disc_f32_c<true, true>(ar2, ai2, ar1, ai1);
        pa += NW * RS / 2;
    };
    for (; base + 128 <= cnt; base += NW * RS) round(base, std::true_type{});
    for (; base < last; base += NW * RS) round(base, std::false_type{});
}

// Odd downsample 3 ... 15 in the adjacent-window form.  Lane l of a round owns windows base + 2 l and base + 2 l + 1: 2 D
// consecutive samples.  Where the pair starts at an even sample that is D whole dwords, the middle one shared -- its low half
// closes the first window, its high half opens the second; at an odd sample it is D + 1 dwords with half a dword at either
// end.  Which of the two is the same for every lane and round of a wave (a lane's pairs are 2 D samples apart, a round 508 D),
// and so is the rotation phase up to the lane's parity ((s + 2 D l) mod 4): the byte weights -- the A / B pair of an aligned
// dword, two masked copies for the half dwords -- and the additive constants are per-lane registers set up once per wave.
// From there on it is the even factors' loop: biased sums read as f32, the second window's predecessor the lane's own
// first, the first one's from lane l - 1 by DPP, the complex product as four fmas.  (The masked-window rounds this replaces
// for these factors read D + 1 dwords per lane-pair too, but pack each sum, move three values by DPP and take the packed
// complex product: 81 against 75 vector instructions per round at downsample 5; the lane stride of D dwords is odd, so the
// LDS reads are conflict-free.)
template <int D, bool NOWRAP>
__device__ __forceinline__ void odd_pair_rounds(const uint32_t* __restrict__ raw_w, int16_t* __restrict__ d16, int wofs, int s00, int cnt,
                                                uint32_t lane, uint32_t wave)
{
    static_assert((D & 1) == 1 && D >= 3 && D <= 15, "odd factors with the f32 discriminator");
    constexpr int H = (D - 1) / 2;                           // whole dwords per window
    const int last = cnt - 1;
    int base = (int)wave * RS;
    const int s_w = s00 + D * base;                          // call sample where window `base` starts (wave-uniform)
    const bool odd = (s_w & 1) != 0;                         // wave-uniform: the pair starts in the high half of a dword
    // The lane's first window starts at call sample s_l = s_w + 2 D lane.  D is odd, so everything per-lane below depends on the
    // lane's PARITY only: s_l = s_w + 2 (lane & 1) (mod 4), and (s_l >> 1) = (s_w >> 1) + D lane has the parity of (s_w >> 1) + lane.
    // Round 6: the set-up written for exactly that -- one sign mask and xors for the byte weights, the additive constants out of
    // compile-time byte tables by v_perm_b32 -- instead of per-lane phase arithmetic, compares and selects: ~25 instead of ~75 vector
    // instructions per wave, a fifth of what a wave of the downsample-5 kernel issued (profiles/r06_experiments.md 1).
    // dword 0 of the lane's span is call dword (s_l >> 1): its weights are the EVEN pair when that index is even
    const uint32_t mk = (uint32_t)__builtin_amdgcn_sbfe((int)(lane ^ (uint32_t)(s_w >> 1)), 0, 1);       // all ones: an odd call dword
    constexpr uint32_t XRE = FMD_W_RE_EVEN ^ FMD_W_RE_ODD, XIM = FMD_W_IM_EVEN ^ FMD_W_IM_ODD;
    const uint32_t wrA = FMD_W_RE_EVEN ^ (mk & XRE), wrB = wrA ^ XRE;   // dword u: A for even u, B for odd u
    const uint32_t wiA = FMD_W_IM_EVEN ^ (mk & XIM), wiB = wiA ^ XIM;
    // the half dwords: even start -> dword H (low half: window 1, high half: window 2), weights A or B by H's parity; odd start ->
    // dword 0 (high half, window 1: A) and dword D (low half, window 2: B, D is odd).  `odd` is wave-uniform: scalar masks / flips.
    const uint32_t hm1 = odd ? 0xFFFF0000u : 0x0000FFFFu, hm2 = ~hm1;
    const bool f1 = !odd && (H & 1) != 0, f2 = odd || (H & 1) != 0;          // the half dword's weights are the B pair
    const uint32_t r1h = (wrA ^ (f1 ? XRE : 0u)) & hm1, i1h = (wiA ^ (f1 ? XIM : 0u)) & hm1;
    const uint32_t r2h = (wrA ^ (f2 ? XRE : 0u)) & hm2, i2h = (wiA ^ (f2 ? XIM : 0u)) & hm2;
    // additive constants of a window that starts at rotation phase sm: fmd_const(sm + D) - fmd_const(sm), one byte each (<= 9);
    // window 1 starts at sm1 = s_l & 3, window 2 at (sm1 + D) & 3 -- its tables are rotated accordingly, so ONE selector serves all
    // four: byte 0 picks the table entry, bytes 1 ... 3 the upper bytes of kSumBias
    constexpr auto tbl = [](bool im, int k) constexpr {
        uint32_t t = 0;
        for (int sm = 0; sm < 4; ++sm) {
            const int s = (sm + k) & 3;
            t |= (uint32_t)(im ? fmd_const_im(s + D) - fmd_const_im(s) : fmd_const_re(s + D) - fmd_const_re(s)) << (8 * sm);
        }
        return t;
    };
    constexpr uint32_t TRE1 = tbl(false, 0), TIM1 = tbl(true, 0), TRE2 = tbl(false, D), TIM2 = tbl(true, D);
    static_assert((kSumBias & 0xFF) == 0, "the table byte replaces the bias's low byte");
    const uint32_t sel = ((((uint32_t)s_w) ^ (lane << 1)) & 3u) | 0x07060500u;
    const int bre1 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, TRE1, sel), bim1 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, TIM1, sel);
    const int bre2 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, TRE2, sel), bim2 = (int)__builtin_amdgcn_perm((uint32_t)kSumBias, TIM2, sel);
    const uint32_t* __restrict__ pa = raw_w + (uint32_t)(wofs + (s_w >> 1)) + (uint32_t)D * lane;
    auto round = [&](int b, auto full_c, auto odd_c) {
        constexpr bool FULL = decltype(full_c)::value, ODD = decltype(odd_c)::value;
        const int i1 = b + 2 * (int)lane, i2 = i1 + 1;
        int re1, im1, re2, im2;
        uint32_t dead1 = 0, dead2 = 0;
        if constexpr (!ODD) {
            // dwords 0 .. H-1: window 1; dword H: both (halves); dwords H+1 .. D-1: window 2
            {
                const uint32_t w = pa[H] ^ 0x80808080u;
                re1 = sdot4_init(w, r1h, bre1); im1 = sdot4_init(w, i1h, bim1);
                re2 = sdot4_init(w, r2h, bre2); im2 = sdot4_init(w, i2h, bim2);
                dead1 = w;
            }
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const uint32_t wa = pa[u] ^ 0x80808080u, wb = pa[H + 1 + u] ^ 0x80808080u;
                re1 = sdot4(wa, (u & 1) ? wrB : wrA, re1); im1 = sdot4(wa, (u & 1) ? wiB : wiA, im1);
                re2 = sdot4(wb, ((H + 1 + u) & 1) ? wrB : wrA, re2); im2 = sdot4(wb, ((H + 1 + u) & 1) ? wiB : wiA, im2);
                dead2 = wb;
            }
        } else {
            // dword 0 (high half) + dwords 1 .. H: window 1; dwords H+1 .. D-1 + dword D (low half): window 2
            {
                const uint32_t wa = pa[0] ^ 0x80808080u, wb = pa[D] ^ 0x80808080u;
                re1 = sdot4_init(wa, r1h, bre1); im1 = sdot4_init(wa, i1h, bim1);
                re2 = sdot4_init(wb, r2h, bre2); im2 = sdot4_init(wb, i2h, bim2);
                dead1 = wa; dead2 = wb;
            }
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const uint32_t wa = pa[1 + u] ^ 0x80808080u, wb = pa[H + 1 + u] ^ 0x80808080u;
                re1 = sdot4(wa, ((1 + u) & 1) ? wrB : wrA, re1); im1 = sdot4(wa, ((1 + u) & 1) ? wiB : wiA, im1);
                re2 = sdot4(wb, ((H + 1 + u) & 1) ? wrB : wrA, re2); im2 = sdot4(wb, ((H + 1 + u) & 1) ? wiB : wiA, im2);
            }
        }
        const float ar1 = sum_to_f32(re1), ai1 = sum_to_f32(im1), ar2 = sum_to_f32(re2), ai2 = sum_to_f32(im2);
        const float br1 = u2f(wave_shr1_dead(dead1, f2u(ar2))), bi1 = u2f(wave_shr1_dead(dead2, f2u(ai2)));   // second window of lane l - 1
        if (lane > 0 && (FULL || i1 < cnt)) d16[i1] = (int16_t)disc_f32_c<NOWRAP, true>(ar1, ai1, br1, bi1);
        if (FULL || i2 < cnt) d16[i2] = (int16_t)disc_f32_c<NOWRAP, true>(ar2, ai2, ar1, ai1);
        pa += D * (NW * RS / 2);                             // 508 windows = 508 D samples = 254 D dwords
    };
    if (odd) {
        for (; base + 128 <= cnt; base += NW * RS) round(base, std::true_type{}, std::true_type{});
        for (; base < last; base += NW * RS) round(base, std::false_type{}, std::true_type{});
    } else {
        for (; base + 128 <= cnt; base += NW * RS) round(base, std::true_type{}, std::false_type{});
        for (; base < last; base += NW * RS) round(base, std::false_type{}, std::false_type{});
    }
}

// ---- register-streaming rounds (fmd_demod_stream_kernel) ------------------------------------------------------------
// The adjacent-window rounds with the raw bytes going global memory -> registers, never through LDS: lane l of a round
// owns windows base + 2 l and base + 2 l + 1, 8 DH CONTIGUOUS bytes of the channel, and the 64 lanes of a wave read
// 512 DH contiguous bytes per round with one or two 16-byte loads per lane.  A wave keeps P rounds of loads in flight in
// a ring of register sets (the loop body appears P times, so the ring is indexed at compile time) and computes round k
// while rounds k + 1 ... k + P - 1 are on their way: load and compute overlap per ROUND inside every wave, instead of
// per TILE across the 8 blocks a CU can hold.  Bytes in flight per CU: 32 waves x P x 512 DH (196 KB at downsample 6, P =
// 4) against ~86 KB for 8 LDS tiles of which some are always computing.  LDS only holds the discriminator samples, so a
// tile can be several times larger (fewer prologues / resampler passes / halos per byte).  Loads beyond the wave's last
// round are issued against the channel's first bytes (one cache line for the whole wave): the count of outstanding
// loads stays a compile-time constant, which is what lets hipcc wait with vmcnt(N > 0).
// Used for downsample 2 and 4 ONLY, where a lane's span is one 8- / 16-byte load and a wave's load instruction reads
// 512 / 1024 contiguous bytes: +4 ... 11 % over the LDS-DMA kernel (D=4: 65 -> 73 % of the HBM spec).  At 6, 10, 12 the
// span is 24 / 40 / 48 bytes -- 16-byte loads at that stride touch 1.5 - 3 x the cache lines -- and the kernel measured
// 25 - 70 % SLOWER; a variant with one window per load and lanes 64 windows apart (perfectly coalesced dwordx3 / dwordx4
// loads at downsample 6 / 8) measured 2 / 6 % slower than LDS-DMA (profiles/r03_experiments.md).
template <int NDW>
__device__ __forceinline__ void stream_load(const unsigned char* __restrict__ p, uint32_t (&w)[NDW])
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef const FMD_AS_GLOBAL u32x4* g4;
    typedef const FMD_AS_GLOBAL u32x2* g2;
    typedef const FMD_AS_GLOBAL uint32_t* g1;
    int u = 0;
#pragma unroll
    for (; u + 4 <= NDW; u += 4) {
        const u32x4 v = __builtin_nontemporal_load((g4)(uintptr_t)(p + 4 * u));
        w[u] = v.x; w[u + 1] = v.y; w[u + 2] = v.z; w[u + 3] = v.w;
    }
    if constexpr (NDW % 4 >= 2) {
        const u32x2 v = __builtin_nontemporal_load((g2)(uintptr_t)(p + 4 * (NDW & ~3)));
        w[NDW & ~3] = v.x; w[(NDW & ~3) + 1] = v.y;
    }
    if constexpr (NDW % 2 == 1) w[NDW - 1] = __builtin_nontemporal_load((g1)(uintptr_t)(p + 4 * (NDW - 1)));
}

template <int DH>
__device__ __forceinline__ void stream_pair_rounds(const unsigned char* __restrict__ chan, uint32_t nbytes, int16_t* __restrict__ d16,
                                                   int jfirst, uint32_t hp, int cnt, uint32_t lane, uint32_t wave, [[maybe_unused]] uint32_t dbg)
{
    constexpr int NDW = 2 * DH;
    // rounds in flight per wave (register budget: 64 VGPRs; the kernel runs downsample 2 and 4 -- 2 / 4 dwords per lane and round, 32 / 46
    // VGPRs at 8 rounds: session r05bk, 3 / 4 / 5 / 6 / 8 rounds: +0.9 ... 0 / 0 / -0.3 ... -1.0 / -0.3 ... -1.2 / -0.4 ... -1.7 %)
    constexpr int P0 = NDW <= 4 ? 8 : (NDW <= 6 ? 4 : (NDW <= 10 ? 3 : 2));
    // ONE-SHOT waves (round 6, downsample 4): no more rounds per wave than register sets -- every load of the wave is issued up front,
    // nothing is refilled, the wave is short-lived like a block of the LDS-DMA kernels (what tools/membench shows for plain reads:
    // one-shot tiles 6.3 - 7.0 TB/s, a grid-stride stream 5.3 - 5.6)
    constexpr bool ONESHOT = (int)FMD_STREAM_MAX_ROUNDS(2u * DH) <= P0;
    constexpr int P = ONESHOT ? (int)FMD_STREAM_MAX_ROUNDS(2u * DH) : P0;
    wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const int last = cnt - 1;
    constexpr int RSTEP = NW * RS;
    int base = (int)wave * RS;
    constexpr int32_t STRIDE = 4 * DH * RSTEP;               // bytes from one round of a wave to its next
    if (base >= last) return;
    const int jw = jfirst + base;
    const bool o1 = ((((DH & 1) ? ((uint32_t)jw ^ hp) : hp)) & 1u) != 0u, o2 = o1 != ((DH & 1) != 0);   // wave-uniform (see tile_body)
    const uint32_t r1A = o1 ? FMD_W_RE_ODD : FMD_W_RE_EVEN, r1B = o1 ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
    const uint32_t m1A = o1 ? FMD_W_IM_ODD : FMD_W_IM_EVEN, m1B = o1 ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
    const uint32_t r2A = o2 ? FMD_W_RE_ODD : FMD_W_RE_EVEN, r2B = o2 ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
    const uint32_t m2A = o2 ? FMD_W_IM_ODD : FMD_W_IM_EVEN, m2B = o2 ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
    const int c1 = 2 * (o1 ? DH / 2 : (DH + 1) / 2), c2 = 2 * (o2 ? DH / 2 : (DH + 1) / 2);
    // byte offset of the lane's span within the channel-call, round by round; spans outside the call (the window before
    // the call's first sample in tile 0, surplus lanes of the last round) are clamped into it: their results are patched
    // (call start) or never stored
    int32_t off = 4 * (DH * (jfirst + base + 2 * (int)lane) - (int)hp);
    const int32_t maxoff = (int32_t)nbytes - 4 * NDW;
    int fbase = base;                                        // round the next fetch belongs to
    auto fetch = [&](uint32_t (&w)[NDW]) {
        const int32_t lim = fbase < last ? maxoff : 0;       // wave-uniform: beyond the last round -> the channel's first bytes
        int32_t o;                                           // clamp(off, 0, lim) as ONE instruction (hipcc: min, compare, VCC select)
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o) : "v"(off), "s"(lim));
        stream_load<NDW>(chan + (uint32_t)o, w);
        off += STRIDE; fbase += RSTEP;
    };
    // Per lane: where its two samples of round 0 go, and how many samples are left from there.  Round k is then a compile-time
    // offset (the rounds are unrolled) and ONE compare against a literal per store instead of index arithmetic per round.
    int16_t* const dl = d16 + base + 2 * (int)lane;
    // the sums' start values in registers of their own for the whole (straight-line) function: left to itself hipcc
    // re-materialises them with a v_mov in every round
    int bre = kSumBias + DH, bim1 = kSumBias + c1, bim2 = kSumBias + c2;
    float k4096 = 4096.0f;
    asm volatile("" : "+v"(bre), "+v"(bim1), "+v"(bim2), "+v"(k4096));
    const int rem = cnt - base - 2 * (int)lane;              // sample i1 of round k is inside the tile iff k * NW * RS < rem
    auto round = [&](const uint32_t (&w)[NDW], auto kc) {
        constexpr int KO = decltype(kc)::value * RSTEP;
#ifdef FMD_EXPERIMENT
        if ((dbg >> 24) & 1u) {                                  // ablation (experiment build): the loads and one store per round, no arithmetic -- the kernel's skeleton
            uint32_t x = w[0];
#pragma unroll
            for (int u = 1; u < NDW; ++u) x ^= w[u];
            if (KO < rem) dl[KO] = (int16_t)x;
            return;
        }
#endif
        int re1 = bre, im1 = bim1, re2 = bre, im2 = bim2;
        uint32_t dead1 = 0, dead2 = 0;
#pragma unroll
        for (int u = 0; u < DH; ++u) {
            const uint32_t wa = w[u] ^ 0x80808080u, wb = w[u + DH] ^ 0x80808080u;    // u8 -> s8 (b - 128)
            if (u == 0) {
                re1 = sdot4_init(wa, r1A, re1); im1 = sdot4_init(wa, m1A, im1);
                re2 = sdot4_init(wb, r2A, re2); im2 = sdot4_init(wb, m2A, im2);
            } else {
                re1 = sdot4(wa, (u & 1) ? r1B : r1A, re1);
                im1 = sdot4(wa, (u & 1) ? m1B : m1A, im1);
                re2 = sdot4(wb, (u & 1) ? r2B : r2A, re2);
                im2 = sdot4(wb, (u & 1) ? m2B : m2A, im2);
            }
            dead1 = wa; dead2 = wb;
        }
        const float ar1 = sum_to_f32(re1), ai1 = sum_to_f32(im1), ar2 = sum_to_f32(re2), ai2 = sum_to_f32(im2);
        const float br1 = u2f(wave_shr1_dead(dead1, f2u(ar2))), bi1 = u2f(wave_shr1_dead(dead2, f2u(ai2)));   // second window of lane l - 1
        // (the discriminators sit INSIDE the predicated stores: two separately masked instruction streams, the form hipcc
        //  built by itself while the conversion at their end was a plain cast it could sink -- and the faster one)
        if (lane > 0 && KO < rem) dl[KO] = (int16_t)disc_f32_c<DH == 1, true, DH == 2>(ar1, ai1, br1, bi1, k4096);
        if (KO + 1 < rem) dl[KO + 1] = (int16_t)disc_f32_c<DH == 1, true, DH == 2>(ar2, ai2, ar1, ai1, k4096);
    };
    // Straight-line code for up to FMD_STREAM_MAX_ROUNDS(downsample) rounds per wave (the host sizes the tiles accordingly): in a
    // loop hipcc's wait-count pass gives up at the back edge and waits for EVERY outstanding load (vmcnt(0)) once per trip,
    // which stalls the whole ring; unrolled, every round waits for exactly its own register set.
    uint32_t w[P][NDW];
#pragma unroll
    for (int p = 0; p < P; ++p) fetch(w[p]);
    constexpr int MAXR = (int)FMD_STREAM_MAX_ROUNDS(2u * DH);
    if constexpr (ONESHOT) {
        static_assert(P == MAXR && MAXR <= 16, "one-shot form");
#define FMD_SR1(K) if constexpr ((K) < MAXR) { round(w[(K) % P], std::integral_constant<int, (K)>{}); \
                       if constexpr ((K) + 1 < MAXR) { base += RSTEP; if (base >= last) return; } }
        FMD_SR1(0) FMD_SR1(1) FMD_SR1(2) FMD_SR1(3) FMD_SR1(4) FMD_SR1(5) FMD_SR1(6) FMD_SR1(7)
        FMD_SR1(8) FMD_SR1(9) FMD_SR1(10) FMD_SR1(11) FMD_SR1(12) FMD_SR1(13) FMD_SR1(14) FMD_SR1(15)
#undef FMD_SR1
        return;
    } else {
    static_assert(MAXR == 12 || MAXR == 16, "the unrolled rounds below");
#define FMD_SR(K) round(w[(K) % P], std::integral_constant<int, (K)>{}); base += RSTEP; if (base >= last) return; fetch(w[(K) % P])
    FMD_SR(0); FMD_SR(1); FMD_SR(2); FMD_SR(3); FMD_SR(4); FMD_SR(5); FMD_SR(6); FMD_SR(7); FMD_SR(8); FMD_SR(9); FMD_SR(10);
    if constexpr (MAXR == 16) {
        FMD_SR(11); FMD_SR(12); FMD_SR(13); FMD_SR(14);
        round(w[15 % P], std::integral_constant<int, 15>{});
    } else {
        round(w[11 % P], std::integral_constant<int, 11>{});
    }
#undef FMD_SR
    }
}

// Everything after the tile's bytes are visible in LDS.  Contains one __syncthreads().
// STREAM (fmd_demod_stream_kernel): the raw bytes are not staged -- the rounds read them from global memory straight into
// registers (stream_pair_rounds), LDS holds the discriminator samples only, and the few one-lane window sums of the
// call-start patch / the guard record / the state update read the channel in global memory through the same helper
// (raw_w = the channel-call's first dword, wofs = 0).
template <int DH, bool STREAM = false>
__device__ __forceinline__ void tile_body(const FmdLaunch& L, const TileCtx& X, unsigned char* smem)
{
    const FmdRates& r = L.r;
    const FmdClassPlan& P = L.cls[X.cls];
    const FmdTile& T = X.T;
    const uint32_t* const raw_w = STREAM ? reinterpret_cast<const uint32_t*>((uintptr_t)X.gbase) : reinterpret_cast<const uint32_t*>(smem);
    int16_t* const d16 = reinterpret_cast<int16_t*>(smem + (STREAM ? 0u : X.rich ? L.rg.raw_cap : L.raw_cap));

    const uint32_t tid = threadIdx.x;
    // (the wave index as a SCALAR: the round loops' trip control then runs on the scalar unit -- s_cmp / s_cbranch -- instead of
    //  a per-lane compare and an EXEC-mask update per round)
    const uint32_t lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t p0 = X.rich ? L.rg.p0 : P.p0, c = X.c;
    const int jfirst = X.jfirst, cnt = X.cnt, wofs = STREAM ? 0 : X.wofs;
    // DH >= 8 (downsample 16, 32, 64 with kernels of their own): whole-dword windows too, but a multiple of 4 dwords long --
    // those take the wrap-around walk below, with the window length a compile-time constant
    const bool fastwin = DH > 0 && DH < 8 && (p0 & 1u) == 0u;   // windows are DH whole dwords
    // Only call-start and call-end tiles need the channel's state, each in one-lane regions of its own.  st_in is read-only
    // for the whole launch (st_out is the other buffer): constant address space -> one s_load_dwordx8 on the scalar path, which
    // does not touch vmcnt.  Table form: the load sits INSIDE each of those regions (chan_state() at their top) -- loaded once
    // up front under `if (needed)`, the five values had to be zeroed on the common path (five s_mov per wave of every tile)
    // for the regions no common tile enters.
    typedef const FMD_AS_CONSTANT FmdChanState* cptr_t;
    FmdChanState st0{};
    if (!X.rich && (jfirst <= 0 || T.k0 == 0 || T.last)) st0 = *((cptr_t)(uintptr_t)L.st_in + c);
    auto chan_state = [&]() -> FmdChanState {
        if (X.rich) return *((cptr_t)(uintptr_t)L.rg.st_in + c);
        return st0;
    };

    // Lane-constant weights of the fast window.  The window of decimated sample j starts at call dword
    // m0 = DH*j - p0/2; rotate_90's sign pattern has period 2 dwords, a lane's two windows are 64 samples
    // apart and a wave-round advances by an even number, so the parity of m0 -- hence the weights -- is
    // fixed per lane for the whole tile.
    const uint32_t hp = p0 >> 1;
    const int j0 = jfirst + (int)wave * RS + (int)lane;
    const bool odd = ((((DH & 1) ? ((uint32_t)j0 ^ hp) : hp)) & 1u) != 0u;
    const uint32_t wreA = odd ? FMD_W_RE_ODD : FMD_W_RE_EVEN, wreB = odd ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
    const uint32_t wimA = odd ? FMD_W_IM_ODD : FMD_W_IM_EVEN, wimB = odd ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
    const int im0 = 2 * (odd ? DH / 2 : (DH + 1) / 2);       // +2 per call-even dword; re gets +1 per dword
    const int last = cnt - 1;

    // ---- boxcar + discriminator ------------------------------------------------------------------
    // A wave-round is 128 windows, two per lane (i and i + 64: two independent dependency chains the
    // scheduler interleaves), producing 127 new discriminator samples; lane 0's first window repeats the
    // previous round's last one and only serves as predecessor.
    // DH == 4 (downsample 8): lanes 8 dwords apart are a 4-way conflict for dword reads, which costs more than the pair
    // form saves -- unless the windows are 16-byte aligned in LDS (channel buffers aligned, boxcar phase 0: the usual
    // case), where one ds_read_b128 fetches a whole window: then the pair form is taken with two such reads per lane.
    const bool dh4_aligned = DH == 4 && (((wofs - (int)hp) & 3) == 0);     // block-uniform
    if constexpr (STREAM) {
        // (the host only selects this kernel for an even downsample at an even boxcar phase: whole-dword windows)
        if constexpr (DH == 1 || DH == 2)
            stream_pair_rounds<DH>(reinterpret_cast<const unsigned char*>((uintptr_t)X.gbase), 2u * L.ns, d16, jfirst, hp, cnt, lane, wave, L.dbg);
    } else if (fastwin && (DH != 4 || dh4_aligned)) {
        // Whole-dword windows, f32 discriminator (downsample 2 ... 10).  Lane l takes the ADJACENT windows i = base + 2l
        // and i + 1: the second window's predecessor is the lane's own first one, and only the first one's comes from
        // the neighbour (the second window of lane l - 1; lane 0's first window is the round's overlap and is not
        // stored).  With the components kept in f32 the product a * conj(b) is two multiplies and two fmas -- no pack,
        // swap, conjugate, dot products or conversions of the product -- and half the DPP traffic.  A lane's windows are
        // 2 DH dwords apart (a 2-way LDS bank conflict for odd DH: cheaper than what it saves); the rotation parity of
        // the first window is the same for every lane of a wave, the second window's differs by DH.
        const int wbase = X.rich ? X.wbase : wofs - (int)hp + DH * jfirst;      // LDS dword index of window i is wbase + DH * i
        const int jw = jfirst + (int)wave * RS;
        // (RS is odd: the parity of jfirst + wave * RS is the row's parity bit flipped by the wave's)
        const bool o1 = X.rich ? (((X.par ^ ((DH & 1) ? wave : 0u)) & 1u) != 0u) : (((((DH & 1) ? ((uint32_t)jw ^ hp) : hp)) & 1u) != 0u);
        const bool o2 = o1 != ((DH & 1) != 0);
        const uint32_t r1A = o1 ? FMD_W_RE_ODD : FMD_W_RE_EVEN, r1B = o1 ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
        const uint32_t m1A = o1 ? FMD_W_IM_ODD : FMD_W_IM_EVEN, m1B = o1 ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
        const uint32_t r2A = o2 ? FMD_W_RE_ODD : FMD_W_RE_EVEN, r2B = o2 ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
        const uint32_t m2A = o2 ? FMD_W_IM_ODD : FMD_W_IM_EVEN, m2B = o2 ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
        const int c1 = 2 * (o1 ? DH / 2 : (DH + 1) / 2), c2 = 2 * (o2 ? DH / 2 : (DH + 1) / 2);
        // One wave-round; FULL: all 128 windows of the round lie inside the tile (every round of a wave but its last), so the
        // stores need no per-lane range test.
        auto pair_round = [&](int base, auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            const int i1 = base + 2 * (int)lane, i2 = i1 + 1;
            const uint32_t* __restrict__ pa = raw_w + (uint32_t)(wbase + DH * i1);
            int re1 = kSumBias + DH, im1 = kSumBias + c1, re2 = kSumBias + DH, im2 = kSumBias + c2;
            uint32_t dead1 = 0, dead2 = 0;                   // the last sign-flipped dwords: dead after the dot products
            if constexpr (DH == 4) {                         // (aligned: checked above) one 16-byte read per window
                const uint4 va = *reinterpret_cast<const uint4*>(pa), vb = *reinterpret_cast<const uint4*>(pa + 4);
                const uint32_t a4[4] = {va.x, va.y, va.z, va.w}, b4[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t wa = a4[u] ^ 0x80808080u, wb = b4[u] ^ 0x80808080u;    // u8 -> s8 (b - 128)
                    if (u == 0) {
                        re1 = sdot4_init(wa, r1A, re1); im1 = sdot4_init(wa, m1A, im1);
                        re2 = sdot4_init(wb, r2A, re2); im2 = sdot4_init(wb, m2A, im2);
                    } else {
                        re1 = sdot4(wa, (u & 1) ? r1B : r1A, re1);
                        im1 = sdot4(wa, (u & 1) ? m1B : m1A, im1);
                        re2 = sdot4(wb, (u & 1) ? r2B : r2A, re2);
                        im2 = sdot4(wb, (u & 1) ? m2B : m2A, im2);
                    }
                    dead1 = wa; dead2 = wb;
                }
            } else {
#pragma unroll
                for (int u = 0; u < (DH > 0 ? DH : 1); ++u) {
                    const uint32_t wa = pa[u] ^ 0x80808080u, wb = pa[u + DH] ^ 0x80808080u;   // u8 -> s8 (b - 128)
                    if (u == 0) {
                        re1 = sdot4_init(wa, r1A, re1); im1 = sdot4_init(wa, m1A, im1);
                        re2 = sdot4_init(wb, r2A, re2); im2 = sdot4_init(wb, m2A, im2);
                    } else {
                        re1 = sdot4(wa, (u & 1) ? r1B : r1A, re1);
                        im1 = sdot4(wa, (u & 1) ? m1B : m1A, im1);
                        re2 = sdot4(wb, (u & 1) ? r2B : r2A, re2);
                        im2 = sdot4(wb, (u & 1) ? m2B : m2A, im2);
                    }
                    dead1 = wa; dead2 = wb;
                }
            }
            const float ar1 = sum_to_f32(re1), ai1 = sum_to_f32(im1), ar2 = sum_to_f32(re2), ai2 = sum_to_f32(im2);
            const float br1 = u2f(wave_shr1_dead(dead1, f2u(ar2))), bi1 = u2f(wave_shr1_dead(dead2, f2u(ai2)));   // second window of lane l - 1
            // (:362); DH == 1 is downsample 2: no i32 wrap to emulate.  The discriminators sit INSIDE the predicated stores
            // (see stream_pair_rounds)
            if (lane > 0 && (FULL || i1 < cnt)) d16[i1] = (int16_t)disc_f32_c<DH == 1, true, DH == 2>(ar1, ai1, br1, bi1);
            if (FULL || i2 < cnt) d16[i2] = (int16_t)disc_f32_c<DH == 1, true, DH == 2>(ar2, ai2, ar1, ai1);
        };
        int base = (int)wave * RS;
        const int full_to = cnt - 128;                       // (the bound as ONE scalar: `base + 128 <= cnt` cost an add per round)
        for (; base <= full_to && !FMD_ABLATE(6); base += NW * RS) pair_round(base, std::true_type{});
        for (; base < last && !FMD_ABLATE(6); base += NW * RS) pair_round(base, std::false_type{});
    } else if (fastwin) {
        // Hot loop: no branches, no special cases.  Lanes whose window lies outside the tile (the two
        // call-start samples of tile 0, surplus lanes of the last round) read whatever LDS holds there --
        // out-of-range DS reads return 0 -- and their results are either not stored or patched below.
        const int wbase = X.rich ? X.wbase : wofs - (int)hp + DH * jfirst;      // LDS dword index of window i is wbase + DH * i
        for (int base = (int)wave * RS; base < last && !FMD_ABLATE(6); base += NW * RS) {
            const int i1 = base + (int)lane, i2 = i1 + 64;
            int re1 = DH, im1 = im0, re2 = DH, im2 = im0;
            if (FMD_ABLATE(1)) { re1 = (int)lane; im1 = i1 & 255; re2 = im1; im2 = re1; }   // ablation: no window
            else {
                const uint32_t* __restrict__ pa = raw_w + (uint32_t)(wbase + DH * i1);
                const uint32_t* __restrict__ pb = raw_w + (uint32_t)(wbase + DH * i2);
#pragma unroll
                for (int u = 0; u < (DH > 0 ? DH : 1); ++u) {
                    const uint32_t wa = pa[u] ^ 0x80808080u, wb = pb[u] ^ 0x80808080u;   // u8 -> s8 (b - 128)
                    if (u == 0) {
                        re1 = sdot4_init(wa, wreA, DH); im1 = sdot4_init(wa, wimA, im0);
                        re2 = sdot4_init(wb, wreA, DH); im2 = sdot4_init(wb, wimA, im0);
                    } else {
                        re1 = sdot4(wa, (u & 1) ? wreB : wreA, re1);
                        im1 = sdot4(wa, (u & 1) ? wimB : wimA, im1);
                        re2 = sdot4(wb, (u & 1) ? wreB : wreA, re2);
                        im2 = sdot4(wb, (u & 1) ? wimB : wimA, im2);
                    }
                }
            }
            const uint32_t pk1 = pack_lp_perm(re1, im1), pk2 = pack_lp_perm(re2, im2);
            const uint32_t prev1 = wave_shr1(pk1);           // lane l <- first window of lane l-1
            const uint32_t prev2 = wave_shr1_old(wave_ror1(pk1), pk2);   // lane l <- second of l-1; lane 0 <- first of 63
            // (Measured and rejected in round 2: storing the full rounds without predication -- lane 0 to a dummy slot --
            //  so that both discriminators run as one interleaved stream: +2 % at downsample 6 / 10, +7 % at 7.  The
            //  two separately masked regions the compiler builds here are the faster form.)
            // (:362); whole-dword windows: downsample <= 14 (<= FMD_DISC_F32_MAX_D)
            constexpr bool kBias = DH > 0 && 2 * DH <= 11;       // (this branch: downsample 8 at an unaligned tile)
            if (lane > 0 && i1 < cnt) d16[i1] = (int16_t)(FMD_ABLATE(0) ? (int)(pk1 ^ prev1) : disc_f32<kBias, false, true>(pk1, prev1));
            if (i2 < cnt) d16[i2] = (int16_t)(FMD_ABLATE(0) ? (int)(pk2 ^ prev2) : disc_f32<kBias, false, true>(pk2, prev2));
        }
    } else if (DH == -1) {
        d1_pair_rounds(raw_w, d16, wofs, X.rich ? X.s00 : jfirst - (int)p0, cnt, lane, wave);
    } else if (DH < -1 && DH >= -15) {
        if constexpr (DH < -1 && DH >= -15) odd_pair_rounds<-DH, (-DH <= 3)>(raw_w, d16, wofs, X.rich ? X.s00 : -DH * jfirst - (int)p0, cnt, lane, wave);
    } else {
        // Any downsample, any phase: a window of D samples starting at call sample s covers the dwords
        // s/2 .. (s + D - 1)/2; the half dwords at its ends are masked out of the byte weights.  s mod 4 (the
        // rotation phase), hence weights, masks and the additive constants, are again fixed per lane: a lane's
        // windows are 64 and 4*127 samples apart.  The dword count is the same for every window of the call.
        // DH > 0 instantiations only ever run downsample 2 DH (here: at an odd boxcar phase, i.e. DH + 1 dwords per
        // window): telling the compiler so prunes every other window length, the wrap-around walk and the general loop
        // from those kernels (the launch's hot loop is the same; the kernel around it shrinks to a third)
        const int D = DH > 0 ? 2 * DH : DH < 0 ? -DH : (int)r.D;             // DH < 0: an odd downsample -DH with a kernel of its own
        const int s00 = X.rich ? X.s00 : D * jfirst - (int)p0;   // start sample of window i is s00 + D*i
        const int sl = s00 + D * ((int)wave * RS + (int)lane);
        const uint32_t sm = (uint32_t)sl & 3u;               // two's complement: right for the clipped windows too
        const bool podd = ((sl >> 1) & 1) != 0;              // call-dword parity of the first dword
        const uint32_t wreA = podd ? FMD_W_RE_ODD : FMD_W_RE_EVEN, wreB = podd ? FMD_W_RE_EVEN : FMD_W_RE_ODD;
        const uint32_t wimA = podd ? FMD_W_IM_ODD : FMD_W_IM_EVEN, wimB = podd ? FMD_W_IM_EVEN : FMD_W_IM_ODD;
        const int ndw = DH > 0 && DH < 8 ? DH + 1                            // (0 < DH < 8 gets here with an odd phase only)
                                         : (D & 1) ? (D + 1) / 2 : D / 2 + (int)(p0 & 1u);
        const uint32_t mf = (sm & 1u) ? 0xFFFF0000u : 0xFFFFFFFFu;
        const uint32_t ml = ((sm + (uint32_t)D) & 1u) ? 0x0000FFFFu : 0xFFFFFFFFu;
        const bool lastB = ((ndw - 1) & 1) != 0;
        const uint32_t wreF = wreA & mf & (ndw == 1 ? ml : 0xFFFFFFFFu), wimF = wimA & mf & (ndw == 1 ? ml : 0xFFFFFFFFu);
        const uint32_t wreL = (lastB ? wreB : wreA) & ml, wimL = (lastB ? wimB : wimA) & ml;
        const int cre = fmd_const_re((int)sm + D) - fmd_const_re((int)sm);
        const int cim = fmd_const_im((int)sm + D) - fmd_const_im((int)sm);
        // Whole-dword windows an even number of dwords long (downsample 12, 16, ... with an even phase): lanes
        // are ndw dwords apart, so up to 32 of them would hit one LDS bank (PMC at downsample 64: 94 % of the LDS
        // cycles were bank conflicts).  Each lane therefore walks its window from a different even offset and
        // wraps; an even rotation keeps the A, B weight order.
        // (A window length of 2 mod 4 dwords -- downsample 12, 20, 28 -- is only a 2-way conflict: cheaper to take than
        //  to rotate around; those run the plain loops.)
        const bool rotate = (D & 1) == 0 && (p0 & 1u) == 0u && (ndw & 3) == 0 && ndw >= 4;
        const bool smallD = D <= FMD_DISC_F32_MAX_D;
        uint32_t rot0 = 0;
        if (rotate) {
            const uint32_t low = (uint32_t)ndw & (0u - (uint32_t)ndw);                   // lanes 32 / g apart share a bank,
            const uint32_t g = low < 32u ? low : 32u;                                    // g = gcd(ndw, 32)
            rot0 = 2u * ((((lane & 31u) * g) >> 5) % ((uint32_t)ndw >> 1));
        }
        // compile-time dword counts (see masked_rounds); anything else runs the general loop below
        // (products of the packed samples stay below 2^22 up to downsample 11: the biased int -> f32 form of fmd_device.h)
        // (downsample <= 3: |s| <= 2 (128 * 3)^2 < 2^19, `(4096 * s) as i32` cannot wrap: four instructions fewer per sample)
        constexpr bool kBiasOk = (DH < 0 && -DH <= 11) || (DH > 0 && 2 * DH <= 11);
        constexpr bool kNoWrap = (DH < 0 && -DH <= 3) || DH == 1;
#define FMD_MASKED(N) case N: masked_rounds<N, kBiasOk, kNoWrap>(raw_w, d16, wofs, s00, D, cnt, lane, wave, wreA, wreB, wimA, wimB, mf, ml, cre, cim, smallD); break
        bool done = false;
        // (the catch-all kernel, DH == 0, only sees downsample >= 16 once every smaller factor has a kernel of its own:
        //  windows of 9 dwords and more, none of the compile-time counts below)
        if (!rotate && DH != 0) {
            done = true;
            switch (ndw) {                                   // wave-uniform
                FMD_MASKED(1); FMD_MASKED(2); FMD_MASKED(3); FMD_MASKED(4); FMD_MASKED(5); FMD_MASKED(6); FMD_MASKED(7); FMD_MASKED(8);
                default: done = false;
            }
        }
#undef FMD_MASKED
        for (int base = (int)wave * RS; base < last && !done; base += NW * RS) {
            const int i1 = base + (int)lane, i2 = i1 + 64;
            const uint32_t* __restrict__ pa = raw_w + (uint32_t)(wofs + ((s00 + D * i1) >> 1));
            const uint32_t* __restrict__ pb = raw_w + (uint32_t)(wofs + ((s00 + D * i2) >> 1));
            if (rotate) {
                int re1 = cre, im1 = cim, re2 = cre, im2 = cim;
                uint32_t o = rot0;
                for (int v = 0; v < ndw; v += 2) {
                    uint32_t wa = pa[o] ^ 0x80808080u, wb = pb[o] ^ 0x80808080u;         // u8 -> s8 (b - 128)
                    re1 = sdot4(wa, wreA, re1); im1 = sdot4(wa, wimA, im1);
                    re2 = sdot4(wb, wreA, re2); im2 = sdot4(wb, wimA, im2);
                    wa = pa[o + 1] ^ 0x80808080u; wb = pb[o + 1] ^ 0x80808080u;
                    re1 = sdot4(wa, wreB, re1); im1 = sdot4(wa, wimB, im1);
                    re2 = sdot4(wb, wreB, re2); im2 = sdot4(wb, wimB, im2);
                    // wrap: a mask where the window length is a compile-time power of two (downsample 16, 32, 64), sign-mask
                    // arithmetic otherwise -- as `o == ndw ? 0 : o` it was a compare and a VCC-masked select per two dwords
                    o += 2u;
                    if constexpr (DH >= 8 && (DH & (DH - 1)) == 0) o &= (uint32_t)(DH - 1);     // (here ndw == DH: see `rotate`)
                    else o &= (uint32_t)((int)(o - (uint32_t)ndw) >> 31);                   // o < ndw: keep; o == ndw: 0
                }
                const uint32_t pk1 = pack_lp_perm(re1, im1), pk2 = pack_lp_perm(re2, im2);
                const uint32_t prev1 = wave_shr1(pk1);
                const uint32_t prev2 = wave_shr1_old(wave_ror1(pk1), pk2);
                if (lane > 0 && i1 < cnt) d16[i1] = (int16_t)(smallD ? disc_f32(pk1, prev1) : disc_nosel(pk1, prev1));   // smallD: wave-uniform
                if (i2 < cnt) d16[i2] = (int16_t)(smallD ? disc_f32(pk2, prev2) : disc_nosel(pk2, prev2));
                continue;
            }
            uint32_t wa = pa[0] ^ 0x80808080u, wb = pb[0] ^ 0x80808080u;                 // u8 -> s8 (b - 128)
            int re1 = sdot4(wa, wreF, cre), im1 = sdot4(wa, wimF, cim);
            int re2 = sdot4(wb, wreF, cre), im2 = sdot4(wb, wimF, cim);
            int u = 1;
            for (; u + 2 < ndw; u += 2) {                    // whole dwords, weights alternate B, A
                wa = pa[u] ^ 0x80808080u; wb = pb[u] ^ 0x80808080u;
                re1 = sdot4(wa, wreB, re1); im1 = sdot4(wa, wimB, im1);
                re2 = sdot4(wb, wreB, re2); im2 = sdot4(wb, wimB, im2);
                wa = pa[u + 1] ^ 0x80808080u; wb = pb[u + 1] ^ 0x80808080u;
                re1 = sdot4(wa, wreA, re1); im1 = sdot4(wa, wimA, im1);
                re2 = sdot4(wb, wreA, re2); im2 = sdot4(wb, wimA, im2);
            }
            if (u + 1 < ndw) {
                wa = pa[u] ^ 0x80808080u; wb = pb[u] ^ 0x80808080u;
                re1 = sdot4(wa, wreB, re1); im1 = sdot4(wa, wimB, im1);
                re2 = sdot4(wb, wreB, re2); im2 = sdot4(wb, wimB, im2);
                ++u;
            }
            if (u < ndw) {                                   // last dword (possibly half)
                wa = pa[u] ^ 0x80808080u; wb = pb[u] ^ 0x80808080u;
                re1 = sdot4(wa, wreL, re1); im1 = sdot4(wa, wimL, im1);
                re2 = sdot4(wb, wreL, re2); im2 = sdot4(wb, wimL, im2);
            }
            const uint32_t pk1 = pack_lp_perm(re1, im1), pk2 = pack_lp_perm(re2, im2);
            const uint32_t prev1 = wave_shr1(pk1);
            const uint32_t prev2 = wave_shr1_old(wave_ror1(pk1), pk2);
            if (lane > 0 && i1 < cnt) d16[i1] = (int16_t)(smallD ? disc_f32(pk1, prev1) : disc_nosel(pk1, prev1));   // smallD: wave-uniform
            if (i2 < cnt) d16[i2] = (int16_t)(smallD ? disc_f32(pk2, prev2) : disc_nosel(pk2, prev2));
        }
    }
    // Call start (at most once per channel-call, one lane): lp[-1] is demod_pre, lp[0] is the clipped first
    // window plus lp_now, d[0] takes the f64 path (:359); d[0] and d[1] are rewritten with them.  Same
    // wave as the loop's own stores to these entries, so program order makes the patch win.
    bool any_guard = false;                                  // a guarded f64 sample in this tile (FmdF64Exc, fmd_kernels.h)
    FMD_SPECIAL_ONLY(X) if (jfirst <= 0 && tid == 0) {
        // the launch's mailbox post (FmdLaunch::mbox): ANY block of this launch runs after the previous launch has completed;
        // this region is entered by the first tile of every channel-call only, and channel 0's does the post
        // (ONE relaxed 8-byte store -- sequence number in the low word, "its report buffer is not empty" in the high one: nothing to order,
        //  so no release fence; a system-scope release writes the XCD's L2 back, once per launch: session r06c measured the two-word
        //  form with a release 3.3 % over the bare launches, and +0.5 % on cfg-ref's plain launches)
        if (c == 0u && L.mbox) {
            const uint32_t busy = L.exc_prev ? ((L.exc_prev->err != 0u ? 1u : 0u) | (L.exc_prev->count != 0u ? 2u : 0u)) : 0u;
            __hip_atomic_store(reinterpret_cast<uint64_t*>(L.mbox), (uint64_t)(L.seq - 1u) | ((uint64_t)busy << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const FmdChanState st = chan_state();
        int r0, i0, r1, i1w, cr, ci;
        lds_window_sum(raw_w, wofs, 0, fmd_win_end(r.D, p0, 0), r0, i0);
        r0 += st.lp_now_re; i0 += st.lp_now_im;
        lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, 1), fmd_win_end(r.D, p0, 1), r1, i1w);
        if (jfirst < 0) {
            fmd_mul_conj(r0, i0, st.demod_pre_re, st.demod_pre_im, cr, ci);
            bool g;
            const int v = polar_f64(cr, ci, L.f64_guard, g);
            d16[1] = (int16_t)(v + (g ? FMD_F64_SKEW : 0));
            any_guard = g;
        }
        fmd_mul_conj(r1, i1w, r0, i0, cr, ci);
        d16[1 - jfirst] = (int16_t)fmd_fast_atan2(ci, cr);
    }
#ifdef FMD_EXPERIMENT
    if constexpr (STREAM) {                                  // timeline probe of the streaming kernel: the end of this wave's rounds
        if (FMD_ABLATE(29) && lane == 0u) reinterpret_cast<uint32_t*>(smem)[(L.lp_cap * 2u + 64u) / 4u + wave] = (uint32_t)__builtin_readcyclecounter();
    }
#endif
    if (!FMD_ABLATE(20)) __syncthreads();                    // (probe, experiment build: what the barrier in front of the resampler pass costs)
    if constexpr (STREAM && (DH == 1 || DH == 2)) {
        // The call's LAST decimated sample: a lane's span is two windows, and when the call ends after a lane's FIRST
        // window the span runs past the channel-call -- stream_pair_rounds clamps its load into the call, which shifts the
        // lane's bytes.  At most that one sample per channel-call can be affected; one lane redoes it from global memory
        // (after the barrier: another wave may have stored it) and a second barrier orders the resampler behind it.
        FMD_SPECIAL_ONLY(X) if (T.last) {                    // block-uniform
            if (tid == 0) {
                const FmdChanState st = chan_state();
                const int j = (int)P.M - 1;                  // >= 1: the host guarantees M >= 2
                int ar, ai, br, bi, cr, ci;
                lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, j), fmd_win_end(r.D, p0, j), ar, ai);
                lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, j - 1), fmd_win_end(r.D, p0, j - 1), br, bi);
                if (j - 1 == 0) { br += st.lp_now_re; bi += st.lp_now_im; }
                fmd_mul_conj(ar, ai, br, bi, cr, ci);
                if (j >= 2 || jfirst > 0) d16[j - jfirst] = (int16_t)fmd_fast_atan2(ci, cr);   // (j == 1 in tile 0 is the call-start patch's)
            }
            __syncthreads();
        }
    }
    // Several reference calls in one launch (fmd_demod_set_block_len): reference call b >= 1 starts at sample
    // b * block_ns, and its first decimated sample -- index (p0 + b * block_ns) / D of this launch -- takes the
    // f64 path against its predecessor (:359).  A tile owns discriminator samples jA .. jB; at most a few
    // boundaries fall into one, and one lane redoes them -- after the barrier (another wave's loop wrote the entry)
    // and with a second one before the resampler reads it.
    FMD_SPECIAL_ONLY(X) if (L.block_ns && tid == 0) {
        const FmdChanState st = chan_state();
        const uint32_t D = r.D, nb = L.block_ns;
        const uint32_t lo = (uint32_t)(T.jA > 1 ? T.jA : 1) * D;             // first sample count that can complete sample jA
        uint32_t b = lo > p0 ? (lo - p0 + nb - 1u) / nb : 1u;                 // smallest b with (p0 + b*nb) / D >= max(jA, 1)
        if (b == 0u) b = 1u;
        for (; (uint64_t)b * nb < L.ns; ++b) {
            const int j = (int)((p0 + b * nb) / D);
            if (j > T.jB) break;
            if (j < T.jA) continue;
            int ar, ai, br, bi, cr, ci;
            lds_window_sum(raw_w, wofs, fmd_win_begin(D, p0, j), fmd_win_end(D, p0, j), ar, ai);
            lds_window_sum(raw_w, wofs, fmd_win_begin(D, p0, j - 1), fmd_win_end(D, p0, j - 1), br, bi);
            if (j - 1 == 0) { br += st.lp_now_re; bi += st.lp_now_im; }
            fmd_mul_conj(ar, ai, br, bi, cr, ci);
            bool g;
            const int v = polar_f64(cr, ci, L.f64_guard, g);
            d16[j - jfirst] = (int16_t)(v + (g ? FMD_F64_SKEW : 0));
            any_guard |= g;
        }
    }
    FMD_SPECIAL_ONLY(X) if (L.block_ns) __syncthreads();

    // a block that is nearly done holds 20 KB of LDS for nothing: let its last phase win issue arbitration
    // (A/B, two interleaved rounds: config 3 0.1620 -> 0.1607 ms, the reference's rates equal)
    if (FMD_ABLATE(15)) __builtin_amdgcn_s_sleep(16);        // pacing probes: 1024 / 512 / 256 clocks between the rounds and the resampler pass
    if (FMD_ABLATE(16)) __builtin_amdgcn_s_sleep(8);
    if (FMD_ABLATE(17)) __builtin_amdgcn_s_sleep(4);
    __builtin_amdgcn_s_setprio(3);
    // ---- low_pass_real: one audio sample per lane -------------------------------------------------
    // Audio sample k0 + q ends at decimated sample e = eq + q*fa + (er + q*fb) / sr; it sums fa samples, or
    // fa + 1 when the remainder of that one division is below fb (then the previous group ended one sample
    // earlier) -- so one small division gives both ends.  Only the first group of a call can be shorter
    // (it continues the previous call's partial sum, :410-417): one lane redoes it below.
    const uint32_t nk = FMD_ABLATE(7) ? 0u : X.rich ? X.nk : T.k1 - T.k0;
    // (table form: a 32-bit product -- the host admits it only while the whole output array is below 4 GiB)
    // (L.out itself, not the copy of its address in L.rg: a pointer that is a kernel argument is known to be global memory, one
    //  rebuilt from an integer would make every store a flat_store)
    int16_t* const outc = X.rich ? L.out + c * L.rg.out_stride : L.out + (uint64_t)c * L.out_stride;
    // The pass is one loop per group length, chosen by ONE block-uniform switch in front of it: with the switch inside the
    // loop (its exit is per lane) hipcc's structurizer turned the dispatch into forty scalar flag moves and tests per pass --
    // and the scalar unit is what these kernels run out of (profiles/r04_experiments.md 28).
    // FA = 0: any group length (run-time loop); FA = -1: rate_out == rate_resample (fr == sr == 1, divisor 1), every decimated
    // sample IS an audio sample -- a copy (at downsample 1, 48 k -> 48 k, the general pass was 60 % of the kernel's vector
    // instructions).
    auto resample = [&](auto fa_c) {
        constexpr int FA = decltype(fa_c)::value;
        uint32_t q_lo = 0u, q_hi = 0u;                       // audio samples q_lo <= q < q_hi are done by the 16-byte copy below
        if constexpr (FA < 0) {
            // rate_out == rate_resample: the pass is a COPY of the tile's discriminator samples.  Round 6: eight samples per lane -- one
            // 16-byte LDS read (2-byte aligned: the hardware takes it), one ALIGNED 16-byte store -- instead of one: at downsample 1,
            // 48 k -> 48 k, the sample-per-lane copy was a quarter of the kernel's vector instructions (20 passes per tile, now 2.5):
            // -12 % on the launch.  The samples in front of the first 16-byte boundary of the output row and behind the last one take
            // the sample-per-lane loop (unaligned 16-byte stores measured +0.9 % at downsample 6, where the memory side is the bound).
            typedef short fmd_s8 __attribute__((ext_vector_type(8)));
            typedef short fmd_s8u __attribute__((ext_vector_type(8), aligned(2)));
            const int s0 = (int)(T.eq + T.er) - jfirst;      // d16 index of the tile's audio sample 0
            const uint32_t head = (uint32_t)((0u - (uint32_t)(uintptr_t)(outc + T.k0)) & 15u) >> 1;     // samples up to the boundary (block-uniform)
            if (s0 >= 0 && (int)(T.eq + T.er) >= 0 && nk >= head + 8u && !FMD_ABLATE(2) && !FMD_ABLATE(21)) {
                q_lo = head; q_hi = head + ((nk - head) & ~7u);
#pragma clang loop unroll(disable)
                for (uint32_t q = q_lo + 8u * tid; q < q_hi; q += 8u * kThreads)
                    *reinterpret_cast<fmd_s8*>(outc + T.k0 + q) = *reinterpret_cast<const fmd_s8u*>(d16 + s0 + (int)q);
            }
        }
        // (the samples the 16-byte copy left: q < q_lo -- fewer than 8, one lane each -- and everything from q_hi on)
        const uint32_t q_end = tid < q_lo ? q_lo : nk;
#pragma clang loop unroll(disable) vectorize(disable)
        for (uint32_t q = tid < q_lo ? tid : tid - q_lo + q_hi; q < q_end; q += kThreads) {
            if (FMD_ABLATE(2)) { outc[T.k0 + q] = d16[q + 1]; continue; }       // ablation: no resampler
            if (FMD_ABLATE(21)) { outc[T.k0 + q] = (int16_t)q; continue; }      // ... and no LDS read either: the bare store
            if constexpr (FA < 0) {
                const int s = (int)(T.eq + T.er + q);
                outc[T.k0 + q] = d16[(s > 0 ? s : 0) - jfirst];
            } else {
                const uint32_t x = T.er + q * L.fb;
                // sr (the reduced resample rate) is a power of two at the reference's rates (170 k -> 32 k: 16) and at the
                // bench configuration (240 k -> 32 k: 2): shift and mask instead of the exact small divide (wave-uniform)
                uint32_t u, xrem;
                if (L.sr_shift < 32u) { u = x >> L.sr_shift; xrem = x & (r.sr - 1u); }
                else { u = fmd_udiv_small(x, r.sr, L.inv_sr); xrem = x - u * r.sr; }
                const bool extra = xrem < L.fb;
                const int e = (int)(T.eq + q * L.fa + u);
                int s = e - (int)L.fa + (extra ? 0 : 1);
                s = s > 0 ? s : 0;                               // the call's first group starts at 0
                const int16_t* dp = d16 + (s - jfirst);
                int sum;
                if constexpr (FA > 0) sum = group_sum<FA>(dp, extra);          // FA unconditional terms + one optional
                else {
                    sum = 0;
#pragma clang loop vectorize(disable)
                    for (int i = 0; i < (int)L.fa; ++i) sum += dp[i];
                    const int v = dp[L.fa];
                    sum += extra ? v : 0;
                }
                const int16_t av = (int16_t)fmd_sdiv_magic(sum, L.magic_R);
                if (FMD_ABLATE(18)) __builtin_nontemporal_store(av, &outc[T.k0 + q]);      // store-policy probe (experiment build): nt
                else if (FMD_ABLATE(19)) asm volatile("" :: "v"((int)av));                // ... everything but the store itself
                else if (FMD_ABLATE(22)) L.out[(c & 63u) * 256u + (q & 255u)] = av;       // ... stores that never leave the L2 (32 KB of addresses)
                else outc[T.k0 + q] = av;
                if (FMD_ABLATE(23)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ... does the end of the wave wait for the store anyway?
            }
        }
    };
    // (an explicit decision tree, every leaf its own call: as a `switch` with a shared default hipcc's structurizer threaded
    //  three 64-bit "which case ran" flags through all of it -- set, tested and cleared on every tile's path)
#define FMD_RS(N) resample(std::integral_constant<int, N>{})
    const uint32_t fa = L.fa;                                // block-uniform; >= 1 (rate_out >= rate_resample)
    if (r.fr == 1u) FMD_RS(-1);
    else if (fa <= 4u) { if (fa <= 2u) { if (fa == 1u) FMD_RS(1); else FMD_RS(2); } else { if (fa == 3u) FMD_RS(3); else FMD_RS(4); } }
    else if (fa <= 8u) { if (fa <= 6u) { if (fa == 5u) FMD_RS(5); else FMD_RS(6); } else { if (fa == 7u) FMD_RS(7); else FMD_RS(8); } }
    else FMD_RS(0);
#undef FMD_RS
    FMD_SPECIAL_ONLY(X) if (T.k0 == 0 && tid == 0 && nk > 0) {   // same lane as the loop's store to outc[0]: this one wins
        const FmdChanState st = chan_state();
        const int e = (int)(T.eq + fmd_udiv_small(T.er, r.sr, L.inv_sr));
        int sum = st.now_lpr;
        for (int jj = 0; jj <= e; ++jj) sum += d16[jj - jfirst];
        outc[0] = (int16_t)fmd_sdiv_magic(sum, L.magic_R);
    }

    // Guarded f64 samples (rare: within 2^-20 of an integer): their records need the finished group sums.
    FMD_SPECIAL_ONLY(X) if ((jfirst < 0 || L.block_ns) && tid == 0 && any_guard) tile_exc_flush(L, X, chan_state(), raw_w, d16);

    // ---- Demod state after the call (last tile only; :232-239) -------------------------------------
    FMD_SPECIAL_ONLY(X) if (T.last && tid == 0 && !FMD_ABLATE(5)) {
        const FmdChanState st = chan_state();
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
        int pr, pi;                                          // demod_pre = lp[M-1]; M >= 2 is guaranteed by the host
        lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, (int)P.M - 1), fmd_win_end(r.D, p0, (int)P.M - 1), pr, pi);
        ns_.demod_pre_re = pr; ns_.demod_pre_im = pi;
        ns_.reserved = 0;
        L.st_out[c] = ns_;
        if (L.out_len) L.out_len[c] = P.K;
    }
}

__device__ __forceinline__ bool tile_fits(const FmdLaunch& L, const TileCtx& X, uint32_t tid)
{
    if ((uint32_t)X.cnt > L.lp_cap || X.nchunks * 16u > L.raw_cap) {
        if (tid == 0) { atomicOr(L.err, (uint32_t)X.cnt > L.lp_cap ? FMD_DEVERR_LP_CAP : FMD_DEVERR_RAW_CAP); fmd_flag_report(L.hflag); }
        return false;
    }
    return true;
}

// Fast geometry (FmdFastGeo, fmd_kernels.h): where the tile's bytes are, from the first 64 bytes of the kernel
// arguments alone -- everything a fresh block needs before its DMAs can go out.
struct FastAddr {
    uint32_t c, t;
    uint32_t lo2, hi2;       // byte range [lo2, hi2) of the channel-call the tile reads
    uint64_t gbase, a0;
    uint32_t nchunks;
    bool whole;
};

// Row t of the table.  (Indexing is all hipcc may see: taking the address of a member of the by-value kernel-argument struct --
// to form a 32-bit row offset by hand -- makes it copy the whole 3 KB argument block to scratch memory first.)
__device__ __forceinline__ const FmdTileRow* fast_row(const FmdLaunch& L, uint32_t t) { return &L.rows[t]; }

template <int FAST>
__device__ __forceinline__ FastAddr fast_addr(const FmdLaunch& L)
{
    const FmdFastGeo& g = L.fg;
    FastAddr A;
    A.t = blockIdx.y;
    if constexpr (FAST == 2) {                               // the tile's row of the table: the staged range ready-made
        // ONE scalar-memory round trip for everything a block needs up to its resampler: the geometry block (first 64 bytes of the
        // kernel arguments) and the tile's WHOLE row (one 64-byte line).  The empty asm statement makes every field "used" here,
        // so all the loads go out together and are waited for once; left to itself hipcc fetches each field where it is first
        // used -- the block's channel behind one wait, the staged range behind a second, `s00` / `wbase` and the boxcar phase behind
        // the staging barrier: one more round trip at the head of every wave's compute phase (session r05g: the rows that run
        // closest to their staging skeleton lost 1 - 2 % to it).
        const FmdTileRow& R = *fast_row(L, A.t);
        const FmdRowGeo& G = L.rg;
        asm volatile("" :: "s"(G.iq), "s"(G.chan_stride), "s"(G.n_channels), "s"(G.per), "s"(G.raw_cap), "s"(G.p0), "s"(G.out_stride),
                     "s"(R.lo2a), "s"(R.nchunks), "s"(R.wofs), "s"(R.jfirst), "s"(R.cnt), "s"(R.eq), "s"(R.er), "s"(R.k0), "s"(R.nk), "s"(R.jB),
                     "s"(R.flags), "s"(R.wbase), "s"(R.s00), "s"(R.par));
        __builtin_amdgcn_sched_barrier(0);
        A.c = blockIdx.x * G.per + blockIdx.z;               // grid (8, tiles, per): blockIdx.x is the XCD
        A.lo2 = R.lo2a; A.hi2 = 0u;
        A.gbase = G.iq + (uint64_t)A.c * (uint32_t)G.chan_stride;            // (a call is below 2^31 bytes per channel)
        A.a0 = A.gbase + R.lo2a;
        A.nchunks = R.nchunks;
        A.whole = true;
        return A;
    } else {
        A.c = blockIdx.x * g.per + blockIdx.z;
        const int32_t base = (int32_t)(A.t * g.step2);
        const int32_t lo = base + g.lo_off2, hi = base + g.hi_off2;
        A.lo2 = (uint32_t)(lo > 0 ? lo : 0);
        A.hi2 = A.t + 1u == g.nt ? g.ns2 : (uint32_t)hi;
    }
    // (the host admits the fast prologues only for 16-byte aligned channel bases and call lengths -- fmd_fast_geometry --
    //  so the staged range is 32-bit arithmetic on the row and always lies inside the buffer)
    A.gbase = g.iq + (uint64_t)A.c * g.chan_stride;
    const uint32_t lo2a = A.lo2 & ~15u;
    A.a0 = A.gbase + lo2a;
    A.nchunks = (A.hi2 - lo2a + 15u) >> 4;
    A.whole = true;
    return A;
}

// The rest of the tile context, computed while the DMAs are in flight.
template <int FAST>
__device__ __forceinline__ TileCtx fast_ctx(const FmdLaunch& L, const FastAddr& A)
{
    const FmdFastGeo& g = L.fg;
    const FmdClassPlan& P = L.cls[0];
    TileCtx X;
    X.c = A.c; X.cls = 0u; X.valid = true; X.whole = A.whole;
    X.a0 = A.a0; X.nchunks = A.nchunks; X.gbase = A.gbase;
    FmdTile& T = X.T;
    const uint32_t t = A.t;
    if constexpr (FAST == 2) {                               // everything from the row: no index arithmetic
        const FmdTileRow& R = *fast_row(L, t);
        X.rich = true;
        X.wofs = R.wofs; X.jfirst = R.jfirst; X.cnt = (int)R.cnt;
        X.wbase = R.wbase; X.s00 = R.s00; X.par = R.par; X.nk = R.nk;
        X.need_state = (R.flags & FMD_ROW_STATE) != 0u;
        X.special_bits = R.flags & (FMD_ROW_STATE | FMD_ROW_BLOCKS);
        T.last = (R.flags & FMD_ROW_LAST) != 0u;
        T.k0 = R.k0; T.k1 = R.k0 + R.nk; T.eq = R.eq; T.er = R.er; T.jA = R.jA; T.jB = R.jB;
        T.nLo = 0; T.nHi = 0;                                // (only the general prologue's staging reads them)
        return X;
    }
    X.wofs = -(int)((A.lo2 & ~15u) >> 2);
    T.last = t + 1u == g.nt;
    T.k0 = t * L.r.kt;
    T.k1 = T.k0 + L.r.kt < P.K ? T.k0 + L.r.kt : P.K;
    {
        T.eq = t * g.Qt + P.eq0;
        T.er = P.er0;
        const int32_t ja = (int32_t)(t * g.Qt) + g.jA_off;
        T.jA = ja > 0 ? ja : 0;
        T.jB = T.last ? (int32_t)P.M - 1 : (int32_t)(t * g.Qt) + g.jB_off;
    }
    T.nLo = (int32_t)(A.lo2 >> 1); T.nHi = (int32_t)(A.hi2 >> 1);
    X.jfirst = T.jA - 1;
    X.cnt = T.jB - X.jfirst + 1;
    return X;
}

__device__ __forceinline__ void issue_dma(uint64_t a0, uint32_t nchunks, unsigned char* smem, uint32_t tid)
{
    // global_load_lds_dwordx4: 1 KiB per wave-instruction straight into the tile image, destination =
    // wave-uniform base (M0) + lane * 16; no VGPR round trip, no ds_write pass, one wait for all.
    const unsigned char* src = reinterpret_cast<const unsigned char*>((uintptr_t)a0) + 16u * tid;
    // (the wave index as a scalar: the LDS destination -- M0 -- then advances by an s_add per DMA instead of a
    //  v_readfirstlane + s_mov of a per-lane expression)
    unsigned char* dst = smem + 1024u * (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t nfull = nchunks / kThreads, ntail = nchunks - nfull * kThreads;
#pragma unroll 1
    for (uint32_t l = 0; l < nfull; ++l) { lds_dma16(src, dst); src += 16u * kThreads; dst += 16u * kThreads; }
    if (tid < ntail) lds_dma16(src, dst);
}

// ---- one block per tile, LDS-DMA staging ------------------------------------------------------------
template <int DH, int FAST>
__global__ void __launch_bounds__(kThreads) fmd_demod_tile_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    // A freshly dispatched block's only job is to get its loads out: let it win issue arbitration against the
    // computing waves of the other resident blocks until the DMAs are queued (measured -1.8 % launch time).
    __builtin_amdgcn_s_setprio(3);
#ifdef FMD_EXPERIMENT
    // Timeline probe (ablation bit 29; tools/timeline.py): every wave records its entry, the end of its staging barrier and its end
    // (shader-clock counter) with the hardware slot it ran in, behind the per-channel counts of the caller's `out_len` array.
    const bool tl_on = FMD_ABLATE(29);
    uint64_t tl_t0 = 0, tl_t1 = 0;
    if (tl_on) tl_t0 = __builtin_readcyclecounter();
#endif
    if constexpr (FAST) {
        // One phase class, tiles that repeat exactly (kt * fr % sr == 0), XCD-aware grid (8, tiles, ceil(C / 8)):
        // the byte range of the tile is a multiply-add of the first 64 bytes of the kernel arguments.  The host has
        // checked the LDS sizing of every tile of this launch (fmd_fast_geometry), so there is nothing to assert.
        const FastAddr A = fast_addr<FAST>(L);
        if (A.c >= L.fg.n_channels) return;
        if (FMD_ABLATE(9)) __builtin_amdgcn_s_sleep(2);      // pacing probes (experiment build): 128 / 512 clocks in front of the DMAs
        if (FMD_ABLATE(10)) __builtin_amdgcn_s_sleep(8);
        if (!FMD_ABLATE(4)) issue_dma(A.a0, A.nchunks, smem, tid);
        if constexpr (FAST == 2) {
            // ... and what the resampler starts with, fetched under the DMAs' latency (waited for at the staging barrier, which waits
            // for scalar loads anyway) instead of behind a wait of its own between the rounds and the resampler pass
            asm volatile("" :: "s"(L.out), "s"(L.fa), "s"(L.fb), "s"(L.sr_shift), "s"(L.r.sr), "s"(L.r.fr), "s"(L.magic_R.m), "s"(L.magic_R.sh), "s"(L.inv_sr));
        }
        const TileCtx X = fast_ctx<FAST>(L, A);              // scalar work under the load latency
        if (FMD_ABLATE(3)) {                                 // ablation: staging skeleton only
            __syncthreads();
            if (tid == 0) L.out[(uint64_t)X.c * L.out_stride + X.T.k0] = (int16_t)reinterpret_cast<uint32_t*>(smem)[blockIdx.y & 63u];
            return;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
        if (FMD_ABLATE(11)) __builtin_amdgcn_s_sleep(4);     // ... 256 / 1024 clocks behind the staging barrier
        if (FMD_ABLATE(12)) __builtin_amdgcn_s_sleep(16);
        if (FMD_ABLATE(13)) __builtin_amdgcn_s_sleep(32);
        if (FMD_ABLATE(14)) __builtin_amdgcn_s_sleep(64);
#ifdef FMD_EXPERIMENT
        if (tl_on) tl_t1 = __builtin_readcyclecounter();
#endif
        tile_body<DH>(L, X, smem);
#ifdef FMD_EXPERIMENT
        if (tl_on && (tid & 63u) == 0u && L.out_len) {
            const uint64_t t2 = __builtin_readcyclecounter();
            const uint32_t lin = (blockIdx.z * gridDim.y + blockIdx.y) * 8u + blockIdx.x;
            uint32_t* rec = L.out_len + ((L.n_channels + 1023u) & ~1023u) + 8u * (4u * lin + (tid >> 6));
            rec[0] = (uint32_t)tl_t0; rec[1] = (uint32_t)(tl_t0 >> 32); rec[2] = (uint32_t)(tl_t1 - tl_t0); rec[3] = (uint32_t)(t2 - tl_t0);
            rec[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            rec[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
            rec[6] = X.c; rec[7] = blockIdx.y | (blockIdx.y << 16);  // (tile, block of the channel-call)
        }
#endif
        return;
    }
    uint32_t c = blockIdx.z * 65535u + blockIdx.y, tix = blockIdx.x;
    // XCD-aware block -> (channel, tile) mapping.  Workgroups go to the 8 XCDs (each with its own L2) round-robin
    // in dispatch order, so with the plain mapping the 15 tiles of one channel's 256 KiB are spread over all XCDs.
    // Here XCD k works through its own contiguous eighth of the channels, tile after tile: neighbouring tiles
    // share their halo in one L2 and every XCD streams one contiguous region.  A/B on four boxes: 0 ... -3 %
    // per call, never slower.  Channels beyond the last multiple of 8 keep the plain mapping.
    if (L.xcd_swizzle == 3u) {                               // grid (8, tiles, ceil(C / 8)): see fmd_launch_tile
        c = blockIdx.x * gridDim.z + blockIdx.z;
        tix = blockIdx.y;
    } else if (L.xcd_swizzle && c < (L.n_channels & ~7u)) {
        const uint32_t lin = blockIdx.x + L.tiles * c, xcd = lin & 7u, idx = lin >> 3, q = idx / L.tiles;
        c = L.xcd_swizzle == 1u ? q * 8u + xcd : xcd * (L.n_channels >> 3) + q;
        tix = idx - q * L.tiles;
    }
    if (c >= L.n_channels) return;
    const TileCtx X = tile_setup(L, c, tix);
    if (!X.valid || !tile_fits(L, X, tid)) return;
    if (FMD_ABLATE(4)) {                                     // ablation: no loads at all (compute on LDS garbage)
    } else if (X.whole) {
        issue_dma(X.a0, X.nchunks, smem, tid);
    } else {
        stage_slow(L, X, smem, tid);
    }
    if (FMD_ABLATE(3)) {                                     // ablation: staging skeleton only
        __syncthreads();
        if (tid == 0) L.out[(uint64_t)c * L.out_stride + X.T.k0] = (int16_t)reinterpret_cast<uint32_t*>(smem)[blockIdx.x & 63u];
        return;
    }
    // The LDS-DMAs count on vmcnt.  hipcc already waits for them at the barrier (it knows the builtin writes LDS);
    // the explicit s_waitcnt vmcnt(0) states the requirement instead of relying on that.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    // (Measured and rejected: touching the lines of a tile 512..3584 dispatch slots ahead to pre-warm
    //  L2 / Infinity Cache made the launch 4..40 % SLOWER -- the stream is bandwidth-, not latency-bound.)
    __builtin_amdgcn_s_setprio(0);
    tile_body<DH>(L, X, smem);
}

// ---- register-streaming form: no staging, no staging barrier (see stream_pair_rounds) -------------------------------
// (Round 6 measured this kernel with ONE wave per block -- no block barrier in front of the resampler pass, four times the dispatch
//  granularity: +1.3 ... +5 % at downsample 4 / 2, profiles/r06_experiments.md 2; the instantiation is deleted.)
template <int DH, int FAST>
__global__ void __launch_bounds__(256, 8) fmd_demod_stream_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef FMD_EXPERIMENT
    // Timeline probe (ablation bit 29; tools/timeline.py), as in the tile kernel: entry, end of the rounds (= in front of the block
    // barrier: the "staging" column of the tool is this kernel's whole load + round phase), end of the wave.
    const bool tl_on = FMD_ABLATE(29);
    uint64_t tl_t0 = 0;
    if (tl_on) tl_t0 = __builtin_readcyclecounter();
#endif
    const FastAddr A = fast_addr<FAST>(L);
    if (A.c >= L.fg.n_channels) return;
    const TileCtx X = fast_ctx<FAST>(L, A);
    tile_body<DH, true>(L, X, smem);
#ifdef FMD_EXPERIMENT
    if (tl_on && (threadIdx.x & 63u) == 0u && L.out_len) {
        const uint64_t t2 = __builtin_readcyclecounter();
        const uint32_t lin = (blockIdx.z * gridDim.y + blockIdx.y) * 8u + blockIdx.x;
        uint32_t* rec = L.out_len + ((L.n_channels + 1023u) & ~1023u) + 8u * (4u * lin + (threadIdx.x >> 6));
        const uint32_t t1 = reinterpret_cast<const uint32_t*>(smem)[(L.lp_cap * 2u + 64u) / 4u + (threadIdx.x >> 6)];   // left there by tile_body
        rec[0] = (uint32_t)tl_t0; rec[1] = (uint32_t)(tl_t0 >> 32); rec[2] = t1 - (uint32_t)tl_t0; rec[3] = (uint32_t)(t2 - tl_t0);
        rec[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
        rec[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
        rec[6] = X.c; rec[7] = blockIdx.y | (blockIdx.y << 16);
    }
#endif
}

// Host side of the instantiations: launch_lds<DH> / launch_stream<DH> are defined here and explicitly instantiated in
// fmd_tile_lds_*.hip / fmd_tile_stream.hip (one translation unit per group of downsample factors: `make -j` builds them
// side by side); fmd_tile_launch.hip only sees the declarations.
template <int DH>
void launch_lds(const FmdLaunch& L, dim3 g, size_t lds, hipStream_t stream)
{
    if (L.fast == 1u) hipLaunchKernelGGL((fmd_demod_tile_kernel<DH, 1>), g, dim3(kThreads), lds, stream, L);
    else if (L.fast == 2u) hipLaunchKernelGGL((fmd_demod_tile_kernel<DH, 2>), g, dim3(kThreads), lds, stream, L);
    else hipLaunchKernelGGL((fmd_demod_tile_kernel<DH, 0>), g, dim3(kThreads), lds, stream, L);
}

template <int DH>
void launch_stream(const FmdLaunch& L, dim3 g, size_t lds, hipStream_t stream)
{
    if (L.fast == 2u) hipLaunchKernelGGL((fmd_demod_stream_kernel<DH, 2>), g, dim3(kThreads), lds, stream, L);
    else hipLaunchKernelGGL((fmd_demod_stream_kernel<DH, 1>), g, dim3(kThreads), lds, stream, L);
}

}  // namespace fmd_tk

