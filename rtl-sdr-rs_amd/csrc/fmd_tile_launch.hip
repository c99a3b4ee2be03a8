// fmd_tile_launch.hip -- host side of the production demodulation kernels: LDS sizing, the fast-prologue geometry of a
// launch and the choice of the kernel instantiation (device code: fmd_tile_body.h; instantiations: fmd_tile_lds_*.hip,
// fmd_tile_stream.hip).
#include "fmd_kernels.h"

namespace fmd_tk {
template <int DH> void launch_lds(const FmdLaunch& L, dim3 g, size_t lds, hipStream_t stream);      // fmd_tile_lds_*.hip
template <int DH> void launch_stream(const FmdLaunch& L, dim3 g, size_t lds, hipStream_t stream);   // fmd_tile_stream.hip
}  // namespace fmd_tk

size_t fmd_tile_lds_bytes(const FmdLaunch& L)
{
    const size_t glen = (size_t)L.fa + 1u;
#ifdef FMD_EXPERIMENT
    if (L.stream) return ((2u * ((size_t)L.lp_cap + glen + 1u) + 15u) & ~(size_t)15u) + 16u + 96u;   // + the timeline probe's four words behind the samples (fmd_demod_stream_kernel)
#endif
    return (L.stream ? 0u : (size_t)L.raw_cap) + ((2u * ((size_t)L.lp_cap + glen + 1u) + 15u) & ~(size_t)15u) + 16u;
}

bool fmd_tile_kernel_supports(const FmdRates& r, uint32_t raw_cap)
{
    // disc_nosel needs |x| + |y| < 2^30: |lp| <= 128*D, so |x| + |y| < 4 * (128*D)^2 <= 2^30 up to D = 128
    if (r.D > FMD_MAX_DOWNSAMPLE) return false;
    if ((uint64_t)r.sr * (r.kt + 2) >= (1u << 24)) return false;          // fmd_udiv_small operands
    if ((uint64_t)((r.fr + r.sr - 1) / r.sr + 2) * 32768ull >= (1u << 24)) return false;   // |group sum| < 2^24
    if ((uint32_t)r.R >= (1u << 24)) return false;
    if (raw_cap > 60u * 1024u) return false;
    return true;
}

// Fills L.fg (and L.rows) when the launch qualifies for a fast prologue: one phase class, every tile
// of the launch within the LDS sizing (checked here, once, instead of by every block).  Returns the mode: 1 = closed
// form (tiles repeat exactly: kt * fr % sr == 0), 2 = per-tile table (any rates, at most FMD_FAST_ROWS tiles), 0 = none.
static uint32_t fmd_fast_geometry(FmdLaunch& L, uint32_t per)
{
    if (!L.fast || L.chan_class) return 0u;                     // L.fast on entry: allowed (FMD_FAST != 0 in the experiment build)
    const FmdRates& r = L.r;
    const FmdClassPlan& P = L.cls[0];
    if (P.nt != L.tiles || P.nt == 0u) return 0u;
    const FmdTiling& tl = L.tl;
    const uint64_t ns2 = 2ull * L.ns;
    if (ns2 >= (1ull << 31)) return 0u;
    // The fast prologues take every channel base and the call length as multiples of 16 bytes: a tile's staged range is then
    // 32-bit arithmetic on its row and never runs past the buffer (the scalar unit is the kernels' co-bottleneck:
    // profiles/r04_experiments.md 28).  Anything else takes the general prologue.
    if (((uint64_t)(uintptr_t)L.iq | L.chan_stride | ns2) & 15u) return 0u;
    // the table whenever it fits (round 4: ~1 % faster than the closed form even where both apply; round 5: the row carries the
    // whole tile context); FMD_FAST=1 keeps the closed form for A/B.  The table form addresses the output array with 32-bit
    // products.
    if (P.nt <= FMD_FAST_ROWS && (tl.Rt != 0u || L.fast != 1u) && (uint64_t)L.n_channels * L.out_stride * 2ull < (1ull << 32)) {
        const uint32_t hp = P.p0 >> 1, dhalf = r.D >> 1;
        for (uint32_t t = 0; t < P.nt; ++t) {
            const FmdTile T = fmd_tile_fast(r, P, tl, L.ns, t);
            if ((uint64_t)(T.jB - T.jA + 2) > L.lp_cap || (!L.stream && 2ull * (uint64_t)(T.nHi - T.nLo) + 30u > L.raw_cap)) return 0u;
            FmdTileRow R{};
            const uint32_t lo2 = 2u * (uint32_t)T.nLo, hi2 = 2u * (uint32_t)T.nHi;
            R.lo2a = lo2 & ~15u;
            R.nchunks = (hi2 - R.lo2a + 15u) >> 4;
            R.wofs = -(int32_t)(R.lo2a >> 2);
            R.jfirst = T.jA - 1;
            R.cnt = (uint32_t)(T.jB - R.jfirst + 1);
            R.eq = T.eq; R.er = T.er; R.k0 = T.k0; R.nk = T.k1 - T.k0; R.jA = T.jA; R.jB = T.jB;
            R.flags = (T.last ? FMD_ROW_LAST : 0u) | ((R.jfirst <= 0 || T.k0 == 0u || T.last) ? FMD_ROW_STATE : 0u) | (L.block_ns ? FMD_ROW_BLOCKS : 0u);
            R.wbase = R.wofs - (int32_t)hp + (int32_t)dhalf * R.jfirst;
            R.s00 = (int32_t)r.D * R.jfirst - (int32_t)P.p0;
            R.par = (((dhalf & 1u) ? (uint32_t)R.jfirst : 0u) ^ hp) & 1u;
            L.rows[t] = R;
        }
        FmdRowGeo& q = L.rg;
        q = FmdRowGeo{};
        q.iq = (uint64_t)(uintptr_t)L.iq; q.out = (uint64_t)(uintptr_t)L.out; q.chan_stride = L.chan_stride;
        q.n_channels = L.n_channels; q.per = per; q.out_stride = (uint32_t)L.out_stride; q.raw_cap = L.raw_cap;
        q.p0 = P.p0; q.nt = P.nt; q.st_in = (uint64_t)(uintptr_t)L.st_in;
        return 2u;
    }
    FmdFastGeo& g = L.fg;
    g.iq = (uint64_t)(uintptr_t)L.iq; g.iq_end = g.iq + L.total_bytes; g.chan_stride = L.chan_stride;
    g.n_channels = L.n_channels; g.per = per; g.nt = P.nt; g.ns2 = (uint32_t)ns2; g.Qt = tl.Qt;
    if (tl.Rt != 0u) return 0u;
    const int64_t jA_off = (int64_t)P.eq0 - tl.fq + (P.er0 >= tl.frr ? 1 : 0);
    const int64_t jB_off = (int64_t)P.eq0 + tl.Bq + (P.er0 + tl.Br >= r.sr ? 1 : 0);
    const int64_t lo_off2 = 2 * ((int64_t)r.D * (jA_off - 1) - P.p0), hi_off2 = 2 * ((int64_t)r.D * (jB_off + 1) - P.p0);
    const uint64_t step2 = 2ull * r.D * tl.Qt;
    if (jA_off > 0 || step2 * P.nt + (uint64_t)(hi_off2 > 0 ? hi_off2 : 0) >= (1ull << 31)) return 0u;
    // every tile: the same expressions as fmd_tile_fast (tests/test_plan_and_divides.py proves that one), plus the LDS sizing
    for (uint32_t t = 0; t < P.nt; ++t) {
        const FmdTile T = fmd_tile_fast(r, P, tl, L.ns, t);
        const int64_t ja = (int64_t)t * tl.Qt + jA_off, lo = (int64_t)t * step2 + lo_off2;
        const int64_t jA = ja > 0 ? ja : 0, jB = T.last ? (int64_t)P.M - 1 : (int64_t)t * tl.Qt + jB_off;
        const int64_t nLo2 = lo > 0 ? lo : 0, nHi2 = T.last ? (int64_t)ns2 : (int64_t)t * step2 + hi_off2;
        if (jA != T.jA || jB != T.jB || nLo2 != 2ll * T.nLo || nHi2 != 2ll * T.nHi) return 0u;
        if ((uint64_t)(jB - jA + 2) > L.lp_cap || (!L.stream && (uint64_t)(nHi2 - nLo2) + 30u > L.raw_cap)) return 0u;
    }
    g.step2 = (uint32_t)step2; g.lo_off2 = (int32_t)lo_off2; g.hi_off2 = (int32_t)hi_off2;
    g.jA_off = (int32_t)jA_off; g.jB_off = (int32_t)jB_off;
    return 1u;
}

hipError_t fmd_launch_tile(const FmdLaunch& L, hipStream_t stream, FmdKernelId* used)
{
    if (L.n_channels == 0 || L.tiles == 0) return hipErrorInvalidValue;
    const int dh = (L.r.D % 2 == 0) ? (int)(L.r.D / 2) : -(int)L.r.D;
    uint32_t gy = L.n_channels < 65535u ? L.n_channels : 65535u;
    uint32_t gz = (L.n_channels + 65534u) / 65535u;
    dim3 g(L.tiles, gy, gz);
    FmdLaunch K = L;
    // XCD-aware mapping without index arithmetic: grid (8, tiles, ceil(C / 8)), x fastest in dispatch order, so
    // blockIdx.x IS the XCD and channel = x * gridDim.z + z (blocks of channels >= C exit at once).
    const uint32_t per = (L.n_channels + 7u) / 8u;
    if (L.xcd_swizzle && L.n_channels >= 8u && L.tiles <= 65535u && per <= 65535u) {
        g = dim3(8u, L.tiles, per);
        K.xcd_swizzle = 3u;
        K.fast = fmd_fast_geometry(K, per);
    } else K.fast = 0u;
    // the streaming kernel has the table / closed-form prologue only, and its tiles do not fit the LDS kernel: the caller
    // plans the call again with the LDS tiling
    if (K.stream && !K.fast) return hipErrorNotSupported;
    const size_t lds = fmd_tile_lds_bytes(K);                // (after K.stream is final: the streaming form stages nothing)
    int inst = 0;                                            // the instantiation that runs: DH (catch-all: 0)
    using namespace fmd_tk;
    if (K.stream) {
        inst = dh;
        if (dh == 1) launch_stream<1>(K, g, lds, stream);
        else if (dh == 2) launch_stream<2>(K, g, lds, stream);
        else return hipErrorNotSupported;
    } else {
        // One kernel per downsample factor (DH = half the factor, DH < 0 = the odd factor -DH): with the factor a
        // compile-time constant each one holds a single window loop and gets its own register allocation (one kernel with
        // every window length in it measured 2 ... 8 % slower on each).  Everything else (33 ... 127 without 64) runs the
        // catch-all instantiation 0.
#define FMD_CASE(N) case N: launch_lds<N>(K, g, lds, stream); inst = N; break
        switch (dh) {
            FMD_CASE(1); FMD_CASE(2); FMD_CASE(3); FMD_CASE(4); FMD_CASE(5); FMD_CASE(6); FMD_CASE(7);        // fmd_tile_lds_even.hip
            FMD_CASE(8); FMD_CASE(9); FMD_CASE(10); FMD_CASE(11); FMD_CASE(12); FMD_CASE(13); FMD_CASE(14);   // fmd_tile_lds_wide.hip
            FMD_CASE(15); FMD_CASE(16); FMD_CASE(32); FMD_CASE(64);
            FMD_CASE(-1); FMD_CASE(-3); FMD_CASE(-5); FMD_CASE(-7); FMD_CASE(-9); FMD_CASE(-11); FMD_CASE(-13); FMD_CASE(-15);   // fmd_tile_lds_odd.hip
            FMD_CASE(-17); FMD_CASE(-19); FMD_CASE(-21); FMD_CASE(-23); FMD_CASE(-25); FMD_CASE(-27); FMD_CASE(-29); FMD_CASE(-31);
            default: launch_lds<0>(K, g, lds, stream); inst = 0; break;
        }
#undef FMD_CASE
    }
    if (used) {
        used->family = K.stream ? FMD_KERNEL_STREAM : FMD_KERNEL_TILE;
        used->dh = (int16_t)inst; used->fast = (uint8_t)K.fast; used->kt = K.r.kt; used->lds = (uint32_t)lds;
    }
    return hipGetLastError();
}
