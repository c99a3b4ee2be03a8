// fmd_api.cpp -- the C ABI of include/fmd.h over the HIP kernels (compiled with hipcc).
//
// Host bookkeeping only: validation (the reference's panics become status codes), the phase
// classes that make output counts and tile geometry known without a device round trip,
// double-buffered per-channel Demod state in HBM, staging for the host-buffer entry points.
// There is deliberately no CPU implementation of the path here.
#include "../../include/fmd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <new>
#include <utility>
#include <vector>

#include "fmd_host.h"
#include "fmd_index.h"
#include "fmd_internal.h"
#include "fmd_kernels.h"

namespace {

thread_local char g_err[512] = "";

void set_err(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Work on the handle's device; the caller's current device is restored on every exit path (fmd_host.h).
#define ON_DEVICE(dev)                                                                       \
    FmdDeviceGuard dev_guard_(dev);                                                          \
    if (dev_guard_.error() != hipSuccess) {                                                  \
        set_err("hipSetDevice(%d) failed: %s", (dev), hipGetErrorString(dev_guard_.error()));  \
        return FMD_ERR_HIP;                                                                  \
    }

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? FMD_ERR_NOMEM : FMD_ERR_HIP;                  \
        }                                                                                    \
    } while (0)

// Channels sharing the data-independent phases (prev_index, prev_lpr_index / g).  Channels created
// together and fed equal-sized buffers stay in one class forever; only set_state splits them.
struct PhaseClass {
    uint32_t p0 = 0, i0r = 0;
    uint32_t count = 0;      // channels in the class
    uint32_t last_K = 0;     // audio samples of the most recent call
};

}  // namespace

struct fmd_demod {
    fmd_demod_config cfg{};
    FmdRates r{};
    uint32_t C = 0;
    int device = 0;
    uint32_t lp_cap = 0, raw_cap = 0;
    uint32_t tiling_plan = 0;             // which plan chose the LDS tile: 0 caller's (FMD_KT), 1 the 20 KB budget, 2 the 15.5 ... 17.3 KB window (choose_tiling)
    // register-streaming kernel (fmd_demod_stream_kernel: even downsample <= 16, whole-dword windows): its own, larger tiling
    bool stream_ok = false;
    FmdRates rs{};                        // r with the streaming kernel's audio samples per tile
    uint32_t lp_cap_s = 0;
    uint32_t block_ns = 0;                // fmd_demod_set_block_len: samples per reference call inside one launch
    bool force_generic = false;
    uint32_t xcd_swizzle = 2;             // block -> (channel, tile) mapping (FMD_XCD, experiment build)
    uint32_t dbg = 0;                     // ablation bits (FMD_DBG, experiment build)
    int n_cus = 0;                        // compute units of the device
    uint32_t allow_fast = 2;              // FMD_FAST: 0 general prologue only, 1 closed form only, 2 (default) table, else closed form (A/B)
    // Round 6: everything a launch leaves behind lives in a RING OF THREE (kRing), indexed by the launch's sequence number: the state
    // it wrote, its report buffer (device error word + guarded f64 samples, fmd_kernels.h), its mailbox word and its launch record.
    // That is what lets fmd_demod_check_behind settle launch n - 2 (or n - 1) while the newer ones run: the records of the last three
    // launches stay apart, and the state launch s wrote is still there while launches s + 1 and s + 2 are in flight (a ping-pong of
    // two would have launch s + 2 overwrite it).
    static constexpr uint32_t kRing = 3;
    static constexpr uint32_t kFlagWord = 8;          // h_mbox[kFlagWord + s % kRing]: launch s has written into its report buffer (FmdLaunch::hflag)
    FmdChanState* d_state[kRing] = {nullptr, nullptr, nullptr};
    uint32_t cur = 0;                     // index of the state the NEXT launch reads (= what the newest launch wrote)
    FmdExcBuf* d_exc = nullptr;           // [kRing]: launch seq reports into d_exc[seq % kRing]
    uint32_t* h_head = nullptr;           // page-locked copy of the report buffers' first 16 bytes (fmd_demod_check reads them behind ONE stream synchronisation)
    FmdExcBuf* h_recs = nullptr;          // page-locked copy of ONE report buffer (the light settle path of fmd_demod_check_behind)
    uint32_t* h_mbox = nullptr;           // host-mapped mailbox, kRing 8-byte words: word[s % kRing] = (s | flags << 32), posted by launch s + 1 when it starts
    uint32_t* d_mbox = nullptr;           // ... its device address
    // what the last three launches were: enough to patch their output buffers and, for the one case that needs it -- a guarded sample
    // of launch s in the carried partial sum that launch s + 1 has already consumed -- to run the launches behind s again
    struct Pending {
        uint32_t seq = 0;
        bool valid = false, settled = true;
        bool posts = false;               // the launch posts its predecessor's completion (tile / streaming kernels, the handle's own buffers)
        bool generic = false;             // it ran fmd_demod_generic_kernel (how to launch it again)
        uint32_t st_out = 0;              // index of the state buffer it wrote
        hipStream_t stream = nullptr;
        FmdLaunch L{};
    } pend[kRing];
    FmdExcBuf* exc_override = nullptr;    // fmd_internal_set_report_buffer (pipelined sink: one buffer per in-flight launch)
    double f64_guard = 0x1p-20;           // fixed in the shipped library; FMD_F64_GUARD_LOG2 in the experiment build (tests widen it to exercise the patch path)
    int32_t f64_skew = 0;                 // FMD_F64_SKEW, honoured by -DFMD_EXPERIMENT builds only
    uint32_t seq = 0;                     // launches enqueued
    uint64_t f64_guarded = 0, f64_patched = 0;
    FmdStreamOrder order;                 // cross-stream ordering of consecutive launches (fmd_host.h)
    std::vector<PhaseClass> classes;      // host mirror of the phases
    std::vector<uint32_t> chan_class;     // [C] index into classes
    FmdKernelId last_kernel;              // what the most recent launch ran (fmd_demod_last_kernel, fmd_demod_tiling)
    uint8_t* d_chan_class = nullptr;      // device copy, valid while 1 < classes <= FMD_MAX_CLASSES
    bool d_chan_class_dirty = true;
    hipStream_t stream = nullptr;         // used by the host-buffer entry points
    uint8_t* d_iq = nullptr;  size_t d_iq_cap = 0;
    int16_t* d_out = nullptr; size_t d_out_cap = 0;
};

namespace {

int choose_tiling(fmd_demod* d, uint32_t kt_req)
{
    FmdRates& r = d->r;
    uint32_t kt = kt_req;
    d->tiling_plan = 0;
    if (kt == 0) {
        // Largest tiles that keep 8 blocks per CU resident (160 KiB LDS / 8), then, among those, the tiling
        // with the least issue work per input byte.  The work of one tile is quantised: a wave handles whole
        // rounds of 127 decimated samples (4 waves share the rounds) and the resampler runs in passes of one
        // audio sample per thread, so e.g. 272 audio samples per tile (a second pass for 16 of them) measured
        // 10 % slower than 256 at the reference's own rates.  Weights are instruction counts of the kernel.
        const uint32_t waves = FMD_BLOCK_THREADS / 64u;
        const uint32_t dh = (r.D + 1u) / 2u;                                 // dwords per window
        const double c_round = 64.0 + 12.0 * dh, c_audio = 70.0 + 6.0 * (double)(r.fr / r.sr), c_fixed = 90.0;   // per wave: prologue, staging, barriers, epilogue
        // Two LDS budgets (round 5, profiles/r05_experiments.md section 9).  Rows whose vector work per tile would keep a CU's four
        // SIMDs busy for less than ~85 % of the time its share of the HBM bandwidth needs for the tile's bytes (downsample 10, 12,
        // 16 ...: ~3 clocks per wave-instruction on 4 SIMDs against ~13 bytes per clock and CU) run 1.5 - 2.8 % faster on ~17 KB tiles
        // than on the 20 KB that 8 blocks per CU admit -- since the scalar diet made a tile's fixed costs cheap; the vector-bound
        // rows (downsample <= 8, the odd factors up to 9) still want the largest tile.  So: plan with the large budget, price the
        // result, and plan again with the small one if the row is on the memory side.
        // plan(lo, hi): the tiling with the least issue work per input byte among those whose LDS need lies in (lo, hi]; 0 = none
        // The figures of the ONE part this planner is tuned for (MI355X): 256 CUs x 4 SIMDs, 2.4 GHz, 8 TB/s, ~3 shader clocks per
        // wave-instruction; the LDS windows are whole allocation granules of 1280 bytes (named, not inlined).
        constexpr double kLdsBudget8 = 20480.0;              // 160 KiB / 8 blocks per CU: 16 granules
        constexpr double kLdsSmallLo = 15500.0, kLdsSmallHi = 17300.0;   // the memory-side rows' window (profiles/r05_experiments.md 9)
        constexpr double kClocksPerInstr = 3.0, kSimdsPerCu = 4.0;
        constexpr double kBytesPerClockCu = 8.0e12 / 256.0 / 2.4e9;      // HBM spec over the CUs at the shader clock: ~13 bytes
        constexpr double kVectorShareForSmallTiles = 0.85;   // below this vector-side share of the memory time the row is "on the memory side"
        auto plan = [&](double lo, double hi, double* per_byte_out) -> uint32_t {
            double best = 0.0;
            uint32_t pick = 0;
            for (uint32_t k = 1; k <= 8192u; ++k) {          // (1024 until the end of round 2: at rate ratios near 1 that left most of the LDS unused)
                r.kt = k;
                if ((uint64_t)r.sr * (k + 2) >= (1u << 24)) break;
                const uint32_t lp = fmd_tile_lp_cap(r), raw = fmd_tile_raw_cap(r);
                const double lds = (double)raw + 2.0 * (lp + r.fr / r.sr + 2) + 32.0;
                if (lds > hi && (k > 1 || lo > 0.0)) break;  // (the smallest tile is exempt from the bound only where there is no other plan)
                if (lds <= lo) continue;
                const uint64_t cnt = ((uint64_t)k * r.fr + r.sr - 1) / r.sr + 1;        // decimated samples formed
                const uint64_t rounds = (cnt + 126) / 127, per_wave = (rounds + waves - 1) / waves;
                // What counts is the wave-instructions a tile costs in all -- its rounds, the resampler once per wave that has an
                // audio sample to form, the per-wave prologue / epilogue -- plus a share of the round slots that stay idle when the
                // rounds do not divide by the waves (the tile's LDS is held until its busiest wave is done).  Checked against tiling
                // sweeps at downsample 4, 5, 6, 10, 13, 14, 32, 64 (e.g. 64: 27 audio samples = one full round, not 31 = a second,
                // nearly empty one: -7 %).
                const uint64_t wave_passes = (k + 63) / 64;
                const double idle = (double)(per_wave * waves - rounds);
                const double work = ((double)rounds + 0.35 * idle) * c_round + (double)wave_passes * c_audio + (double)waves * c_fixed;
                const double per_byte = work / (2.0 * r.D * (double)cnt);
                if (pick == 0 || per_byte < best) { best = per_byte; pick = k; }
            }
            if (per_byte_out) *per_byte_out = best;
            return pick;
        };
        double per_byte = 0.0;
        kt = plan(0.0, kLdsBudget8, &per_byte);
        if (kt == 0) kt = 1;
        d->tiling_plan = 1;
        // per_byte = wave-instructions per input byte of that tiling: x 3 clocks / 4 SIMDs = vector clocks per byte and CU; the memory
        // side: 1 / 13 clocks per byte and CU (8 TB/s over 256 CUs at 2.4 GHz)
        if (per_byte * (kClocksPerInstr / kSimdsPerCu) * kBytesPerClockCu < kVectorShareForSmallTiles) {
            const uint32_t k2 = plan(kLdsSmallLo, kLdsSmallHi, nullptr);
            if (k2) { kt = k2; d->tiling_plan = 2; }
        }
    }
    r.kt = kt;
    d->lp_cap = fmd_tile_lp_cap(r);
    d->raw_cap = fmd_tile_raw_cap(r);
    const size_t lds = (size_t)d->raw_cap + (r.D > FMD_MAX_DOWNSAMPLE ? 10u : 6u) * (size_t)d->lp_cap + 32;   // i32 pairs beyond downsample 128
    if (lds > 64 * 1024) {
        set_err("tile needs %zu bytes of LDS (kt=%u): rate_out/rate_resample x downsample too large", lds, kt);
        return FMD_ERR_UNSUPPORTED;
    }
    return FMD_OK;
}

void reset_classes(fmd_demod* d)
{
    d->classes.assign(1, PhaseClass{});
    d->classes[0].count = d->C;
    d->chan_class.assign(d->C, 0u);
    d->d_chan_class_dirty = true;
}

// Re-cluster after one channel's phases changed.
void regroup(fmd_demod* d, uint32_t channel, uint32_t p0, uint32_t i0r)
{
    std::vector<std::pair<uint32_t, uint32_t>> ph(d->C);
    for (uint32_t c = 0; c < d->C; ++c) ph[c] = {d->classes[d->chan_class[c]].p0, d->classes[d->chan_class[c]].i0r};
    std::vector<uint32_t> lastK(d->C);
    for (uint32_t c = 0; c < d->C; ++c) lastK[c] = d->classes[d->chan_class[c]].last_K;
    ph[channel] = {p0, i0r};
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> ids;
    d->classes.clear();
    for (uint32_t c = 0; c < d->C; ++c) {
        auto it = ids.find(ph[c]);
        if (it == ids.end()) {
            it = ids.emplace(ph[c], (uint32_t)d->classes.size()).first;
            PhaseClass pc; pc.p0 = ph[c].first; pc.i0r = ph[c].second; pc.last_K = lastK[c];
            d->classes.push_back(pc);
        }
        d->chan_class[c] = it->second;
        d->classes[it->second].count++;
    }
    d->d_chan_class_dirty = true;
}

bool tile_kernel_ok(const fmd_demod* d)
{
    return !d->force_generic && d->classes.size() <= FMD_MAX_CLASSES && fmd_tile_kernel_supports(d->r, d->raw_cap);
}

// Validates a call; fills one plan per class and the grid's tiles-per-channel.
int plan_call(const fmd_demod* d, const FmdRates& r, size_t nbytes, size_t out_cap, std::vector<FmdClassPlan>& plans, uint32_t* tiles)
{
    if (nbytes % 8 != 0) { set_err("nbytes %% 8 != 0 (simple_fm.rs:286 would panic)"); return FMD_ERR_BAD_LENGTH; }
    const uint64_t ns = nbytes / 2;
    if (d->block_ns) {
        if (ns % d->block_ns != 0) {
            set_err("nbytes %zu is not a multiple of the block length %u set by fmd_demod_set_block_len", nbytes, 2u * d->block_ns);
            return FMD_ERR_BAD_LENGTH;
        }
        if (!tile_kernel_ok(d)) {
            set_err("several reference calls per launch need the tile kernel (<= %d phase classes)", FMD_MAX_CLASSES);
            return FMD_ERR_UNSUPPORTED;
        }
    }
    if (!fmd_ranges_fit32(r, ns)) {
        set_err("call of %zu bytes exceeds the 32-bit index range for these rates", nbytes);
        return FMD_ERR_UNSUPPORTED;
    }
    uint32_t tmax = 1;
    plans.resize(d->classes.size());
    for (size_t k = 0; k < d->classes.size(); ++k) {
        const FmdClassPlan P = fmd_make_plan(r, d->classes[k].p0, d->classes[k].i0r, (uint32_t)ns);
        if (P.M < 2) { set_err("%u decimated samples (simple_fm.rs:356 asserts > 1)", P.M); return FMD_ERR_TOO_SHORT; }
        if (P.K > out_cap) { set_err("a channel produces %u samples, out_cap %zu", P.K, out_cap); return FMD_ERR_CAPACITY; }
        plans[k] = P;
        if (P.nt > tmax) tmax = P.nt;
    }
    *tiles = tmax;
    return FMD_OK;
}

void advance_classes(fmd_demod* d, size_t nbytes, const std::vector<FmdClassPlan>& plans)
{
    const uint32_t ns = (uint32_t)(nbytes / 2);
    bool merged = false;
    for (size_t k = 0; k < d->classes.size(); ++k) {
        PhaseClass& pc = d->classes[k];
        pc.i0r = fmd_next_lpr_index_r(d->r, pc.i0r, plans[k].M, plans[k].K);
        pc.p0 = fmd_next_prev_index(d->r.D, pc.p0, ns);
        pc.last_K = plans[k].K;
        for (size_t m = 0; m < k; ++m) merged |= (d->classes[m].p0 == pc.p0 && d->classes[m].i0r == pc.i0r);
    }
    (void)merged;   // classes that converge stay separate; that only costs a plan slot
}

int enqueue(fmd_demod* d, const void* d_iq, size_t nbytes, void* d_out, size_t out_cap, void* d_out_len,
            hipStream_t stream, bool allow_stream = true)
{
    std::vector<FmdClassPlan> plans;
    uint32_t tiles = 1;
    // the register-streaming kernel: one phase class at an even boxcar phase (whole-dword windows), one reference call per launch
    const bool stream_now = allow_stream && d->stream_ok && d->C >= 8u && d->block_ns == 0u && d->classes.size() == 1 && (d->classes[0].p0 & 1u) == 0u &&
                            nbytes >= 64u * d->r.D && !d->force_generic;
    const FmdRates& rr = stream_now ? d->rs : d->r;
    int rc = plan_call(d, rr, nbytes, out_cap, plans, &tiles);
    if (rc) return rc;
    if (((uintptr_t)d_iq & 15u) != 0) { set_err("d_iq must be 16-byte aligned"); return FMD_ERR_INVALID_ARG; }
    FmdLaunch L{};
    L.iq = static_cast<const uint8_t*>(d_iq);
    L.chan_stride = nbytes;
    L.total_bytes = (uint64_t)nbytes * d->C;
    L.r = rr;
    L.stream = stream_now ? 1u : 0u;
    L.ns = (uint32_t)(nbytes / 2);
    L.block_ns = d->block_ns;
    L.xcd_swizzle = d->xcd_swizzle;               // 2 (default): contiguous eighth of the channels per XCD; 0: plain
    L.n_channels = d->C;
    L.tiles = tiles;
    L.lp_cap = stream_now ? d->lp_cap_s : d->lp_cap;
    L.raw_cap = d->raw_cap;
    L.st_in = d->d_state[d->cur];
    L.st_out = d->d_state[(d->cur + 1u) % fmd_demod::kRing];
    L.out = static_cast<int16_t*>(d_out);
    L.out_stride = out_cap;
    L.out_len = static_cast<uint32_t*>(d_out_len);
    L.exc = d->exc_override ? d->exc_override : d->d_exc + (d->seq + 1u) % fmd_demod::kRing;
    L.err = &L.exc->err;
    L.exc_prev = d->exc_override ? nullptr : d->d_exc + d->seq % fmd_demod::kRing;          // the previous launch's buffer ...
    L.mbox = d->exc_override ? nullptr : d->d_mbox + 2u * (d->seq % fmd_demod::kRing);       // ... and its mailbox word
    L.hflag = d->exc_override || !d->d_mbox ? nullptr : d->d_mbox + fmd_demod::kFlagWord + (d->seq + 1u) % fmd_demod::kRing;    // this launch's "has reported" word
    L.f64_guard = d->f64_guard;
    L.seq = d->seq + 1;
#ifdef FMD_EXPERIMENT
    L.dbg = d->dbg;
    L.f64_skew = d->f64_skew;
#endif
    {   // consecutive launches on different streams: the new stream waits for the previous launch (state ping-pong)
        const hipError_t eo = d->order.before(stream);
        if (eo != hipSuccess) { set_err("stream ordering failed: %s", hipGetErrorString(eo)); return FMD_ERR_HIP; }
    }
    if (tile_kernel_ok(d)) {
        const FmdRates& r = rr;
        L.tl = fmd_make_tiling(r);
        L.Qt = L.tl.Qt;
        L.fa = r.fr / r.sr; L.fb = r.fr % r.sr;
        L.inv_sr = 1.0f / (float)r.sr; L.inv_R = 1.0f / (float)r.R;
        L.magic_R = fmd_make_magic((uint32_t)r.R);
        L.sr_shift = 32u;
        if ((r.sr & (r.sr - 1u)) == 0u) { L.sr_shift = 0u; while ((1u << L.sr_shift) < r.sr) ++L.sr_shift; }
        for (size_t k = 0; k < plans.size(); ++k) L.cls[k] = plans[k];
        L.chan_class = nullptr;
        if (d->classes.size() > 1) {
            if (d->d_chan_class_dirty) {
                std::vector<uint8_t> ids(d->C);
                for (uint32_t c = 0; c < d->C; ++c) ids[c] = (uint8_t)d->chan_class[c];
                if (!d->d_chan_class) HIP_TRY(hipMalloc(&d->d_chan_class, d->C));
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipMemcpy(d->d_chan_class, ids.data(), d->C, hipMemcpyHostToDevice));
                d->d_chan_class_dirty = false;
            }
            L.chan_class = d->d_chan_class;
        }
        L.fast = d->allow_fast;               // fmd_launch_tile decides (fmd_fast_geometry)
        const hipError_t le = fmd_launch_tile(L, stream, &d->last_kernel);
        if (le == hipErrorNotSupported && stream_now)        // no table / closed-form geometry for this call: the LDS kernel with its own tiling
            return enqueue(d, d_iq, nbytes, d_out, out_cap, d_out_len, stream, false);
        HIP_TRY(le);
    } else {
        HIP_TRY(fmd_launch_generic(L, stream, &d->last_kernel));
    }
    const hipError_t e_after = d->order.after(stream);       // (event mode: records the launch's event; reported below, after the bookkeeping)
    d->seq += 1;
    {
        fmd_demod::Pending& pd = d->pend[d->seq % fmd_demod::kRing];
        pd.seq = d->seq; pd.valid = true; pd.settled = false; pd.stream = stream; pd.L = L;
        pd.generic = d->last_kernel.family == FMD_KERNEL_GENERIC;
        pd.posts = L.mbox != nullptr && !pd.generic;
        pd.st_out = (d->cur + 1u) % fmd_demod::kRing;
    }
    d->cur = (d->cur + 1u) % fmd_demod::kRing;
    advance_classes(d, nbytes, plans);
    HIP_TRY(e_after);
    return FMD_OK;
}

// Demod::polar_discriminant (simple_fm.rs:370-374) with the HOST libm -- the function the reference itself calls.
int host_polar_f64(int cr, int ci)
{
    const double angle = atan2((double)ci, (double)cr);
    return (int)(angle / 3.14159265358979323846264338327950288 * 16384.0);
}

int resolve_device_reports(fmd_demod* d, int16_t* host_out, size_t host_cap);

}  // namespace

// Error text for the other translation units of the library (fmd_fir.hip).
void fmd_internal_set_err(const char* msg) { set_err("%s", msg); }

void fmd_internal_set_report_buffer(fmd_demod* d, FmdExcBuf* buf) { d->exc_override = buf; }

FmdHandleView fmd_internal_view(fmd_demod* d)
{
    FmdHandleView v;
    v.R = d->r.R; v.seq = d->seq; v.state_cur = d->d_state[d->cur];
    v.guarded = &d->f64_guarded; v.patched = &d->f64_patched; v.device = d->device;
    return v;
}

// After a handle's work has completed: surface the device error word and re-evaluate the guarded f64 samples
// (FmdF64Exc) with the host libm.  A sample whose host value differs from the kernel's is patched -- in `host_out`
// ([C][host_cap], the caller's copy of the most recent launch's output) when given, else in the device buffer the
// launch wrote -- together with the carried partial sum (d_state_cur[channel].now_lpr) when it lies in the trailing
// group.  host_seq: the launch `host_out` is a copy of; state_seq: the launch whose state d_state_cur holds (the most
// recent one).  Shared by fmd_demod, fmd_firdemod and the pipelined sink.
int fmd_internal_resolve_exc(FmdExcBuf* d_exc, int32_t R, uint32_t host_seq, uint32_t state_seq, FmdChanState* d_state_cur,
                             int16_t* host_out, size_t host_cap, uint64_t* guarded, uint64_t* patched)
{
    uint32_t head[4] = {0, 0, 0, 0};                        // err, count, guarded_total, pad
    HIP_TRY(hipMemcpy(head, d_exc, sizeof(head), hipMemcpyDeviceToHost));
    if (head[0] & ~FMD_DEVERR_EXC_CAP) { set_err("device-side sizing assertion failed (bits 0x%x)", head[0]); return FMD_ERR_HIP; }
    if (head[1] == 0) return FMD_OK;
    const uint32_t n = head[1] < FMD_EXC_CAP ? head[1] : FMD_EXC_CAP;
    std::vector<FmdF64Exc> recs(n);
    HIP_TRY(hipMemcpy(recs.data(), d_exc->rec, n * sizeof(FmdF64Exc), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(d_exc, 0, 16));
    *guarded += head[1];
    int rc = FMD_OK;
    // several guarded samples may share one audio group (block_len mode): their corrections add up
    std::map<std::pair<uint32_t, uint64_t>, std::pair<int64_t, const FmdF64Exc*>> groups;   // (seq, out_elem) -> (delta, record)
    for (const FmdF64Exc& e : recs) {
        const int16_t want = (int16_t)host_polar_f64(e.cr, e.ci), have = (int16_t)e.d_gpu;
        if (want == have) continue;
        *patched += 1;
        const int delta = (int)want - (int)have;
        if (e.k >= 0) {
            auto& g = groups[{e.seq, e.out_elem}];
            g.first += delta; g.second = &e;
        } else if (e.seq == state_seq) {
            int32_t now = 0;
            int32_t* p = &d_state_cur[e.channel].now_lpr;
            HIP_TRY(hipMemcpy(&now, p, sizeof(now), hipMemcpyDeviceToHost));
            now += delta;
            HIP_TRY(hipMemcpy(p, &now, sizeof(now), hipMemcpyHostToDevice));
        } else {
            set_err("a guarded f64 sample of launch %u (channel %u) lies in the carried partial sum and a later launch "
                    "has already consumed it: call the handle's check function after every *_device launch", e.seq, e.channel);
            rc = FMD_ERR_HIP;
        }
    }
    for (const auto& kv : groups) {
        const FmdF64Exc& e = *kv.second.second;
        const int16_t fixed = (int16_t)((e.sum + (int)kv.second.first) / R);              // low_pass_real, simple_fm.rs:421
        if (host_out && e.seq == host_seq) host_out[(size_t)e.channel * host_cap + (size_t)e.k] = fixed;
        else if (e.seq == state_seq) HIP_TRY(hipMemcpy((void*)(uintptr_t)e.out_elem, &fixed, sizeof(fixed), hipMemcpyHostToDevice));
        else {
            // Nothing ties the address saved in an OLDER launch's record to memory that still holds that launch's audio:
            // the caller may have reused the buffer for a later launch (a write would corrupt newer audio) or freed it.
            // Only the most recent launch's buffer is ever written; for anything older the caller is told.
            set_err("a guarded f64 sample of launch %u (channel %u, audio sample %d) needs the host-libm value, but later "
                    "launches have been enqueued since: call the handle's check function after every *_device launch",
                    e.seq, e.channel, e.k);
            rc = FMD_ERR_HIP;
        }
    }
    if (head[0] & FMD_DEVERR_EXC_CAP) {
        set_err("more than %u guarded f64 samples since the last check: some were not re-evaluated", FMD_EXC_CAP);
        rc = FMD_ERR_HIP;
    }
    return rc;
}

namespace {

// Run launch `pd` again (same arguments, same stream): see settle_launch.
int replay_launch(fmd_demod* d, fmd_demod::Pending& pd)
{
    FmdKernelId used;
    // (the caller has synchronised the device; event mode: the handle's own stream -- the caller's may be gone)
    hipStream_t rs = d->order.event_mode ? d->stream : pd.stream;
    if (pd.generic) HIP_TRY(fmd_launch_generic(pd.L, rs, &used));
    else HIP_TRY(fmd_launch_tile(pd.L, rs, &used));
    HIP_TRY(hipStreamSynchronize(rs));
    if (d->order.event_mode && pd.seq == d->seq) HIP_TRY(hipEventRecord(d->order.ev, rs));   // the newest launch's event is now the second run's
    return FMD_OK;
}

// Settle the records launch `s` left in its report buffer (the device is idle: the caller synchronised).  A guarded sample whose
// host-libm value differs from the kernel's is patched where it went: an audio sample in the output buffer launch s wrote (or, for
// the newest launch, in the caller's host copy), a sample of the trailing group in the state launch s wrote -- and if a newer launch
// has consumed that state meanwhile, *carried is set: the caller runs the launches behind s again (their inputs -- the callers' input
// buffers, the state ring -- are untouched until they are settled: the contract of fmd_demod_check_behind).  Records of any other
// launch found in the buffer belong to a launch three or more back that was never settled: reported, not touched.
// The report buffer of ring slot s % kRing has just been cleared (everything that could write it is idle or belongs to other slots).
static inline void clear_report_flag(fmd_demod* d, uint32_t s)
{
    if (d->h_mbox) __atomic_store_n(d->h_mbox + fmd_demod::kFlagWord + s % fmd_demod::kRing, 0u, __ATOMIC_RELAXED);
}

int settle_launch(fmd_demod* d, uint32_t s, int16_t* host_out, size_t host_cap, bool* carried)
{
    FmdExcBuf* const buf = d->d_exc + s % fmd_demod::kRing;
    const fmd_demod::Pending& P = d->pend[s % fmd_demod::kRing];
    uint32_t head[4] = {0, 0, 0, 0};                        // err, count, guarded_total, pad
    HIP_TRY(hipMemcpy(head, buf, sizeof(head), hipMemcpyDeviceToHost));
    if (head[0] & ~FMD_DEVERR_EXC_CAP) { set_err("device-side sizing assertion failed (bits 0x%x)", head[0]); return FMD_ERR_HIP; }
    if (head[1] == 0) return FMD_OK;
    const uint32_t n = head[1] < FMD_EXC_CAP ? head[1] : FMD_EXC_CAP, newest = d->seq;
    std::vector<FmdF64Exc> recs(n);
    HIP_TRY(hipMemcpy(recs.data(), buf->rec, n * sizeof(FmdF64Exc), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(buf, 0, 16));
    clear_report_flag(d, s);
    d->f64_guarded += head[1];
    int rc = FMD_OK;
    // several guarded samples may share one audio group (block_len mode): their corrections add up
    std::map<uint64_t, std::pair<int64_t, const FmdF64Exc*>> groups;       // out_elem -> (delta, record)
    for (const FmdF64Exc& e : recs) {
        const int16_t want = (int16_t)host_polar_f64(e.cr, e.ci), have = (int16_t)e.d_gpu;
        if (want == have) continue;
        d->f64_patched += 1;
        if (e.seq != s) {
            // Nothing ties a record of an older launch to memory that still holds that launch's audio or state: its buffers may have
            // been reused.  Only the last three launches are ever written; for anything older the caller is told.
            set_err("a guarded f64 sample of launch %u (channel %u) needs the host-libm value, but three or more launches have been "
                    "enqueued since: call fmd_demod_check / fmd_demod_check_behind for every *_device launch", e.seq, e.channel);
            rc = FMD_ERR_HIP;
            continue;
        }
        const int delta = (int)want - (int)have;
        if (e.k >= 0) {
            auto& g = groups[e.out_elem];
            g.first += delta; g.second = &e;
            continue;
        }
        int32_t now = 0;                                     // the carried partial sum: in the state launch s WROTE
        int32_t* p = &d->d_state[P.st_out][e.channel].now_lpr;
        HIP_TRY(hipMemcpy(&now, p, sizeof(now), hipMemcpyDeviceToHost));
        now += delta;
        HIP_TRY(hipMemcpy(p, &now, sizeof(now), hipMemcpyHostToDevice));
        if (s != newest) *carried = true;
    }
    for (const auto& kv : groups) {
        const FmdF64Exc& e = *kv.second.second;
        const int16_t fixed = (int16_t)((e.sum + (int)kv.second.first) / d->r.R);         // low_pass_real, simple_fm.rs:421
        if (host_out && s == newest) host_out[(size_t)e.channel * host_cap + (size_t)e.k] = fixed;
        else HIP_TRY(hipMemcpy((void*)(uintptr_t)e.out_elem, &fixed, sizeof(fixed), hipMemcpyHostToDevice));
    }
    if (head[0] & FMD_DEVERR_EXC_CAP) {
        set_err("more than %u guarded f64 samples in one report buffer since the last check: some were not re-evaluated", FMD_EXC_CAP);
        rc = FMD_ERR_HIP;
    }
    return rc;
}

// The LIGHT settle path of fmd_demod_check_behind: launch `s` has completed (its successor posted so) and reported records, newer
// launches are still running.  A guarded sample is, almost always, one whose host-libm value AGREES with the kernel's (the guard band
// only marks where two faithful atan2 implementations COULD differ) -- settling it must not cost the pipeline a drain.  So: copy the
// records on the handle's own (non-blocking) stream, evaluate them on the host, and write the audio samples that need a patch into
// launch s's output buffer -- nothing the running launches touch.  Only a record that corrects the CARRIED partial sum (the state a
// newer launch has already read) or a device assertion needs everything idle: *need_heavy, with the buffer left as it was.
int settle_launch_light(fmd_demod* d, uint32_t s, bool* need_heavy)
{
    *need_heavy = false;
    FmdExcBuf* const buf = d->d_exc + s % fmd_demod::kRing;
    FmdExcBuf* const h = d->h_recs;
    constexpr uint32_t kFirst = 8;                          // the head and the first few records travel in one copy
    HIP_TRY(hipMemcpyAsync(h, buf, offsetof(FmdExcBuf, rec) + kFirst * sizeof(FmdF64Exc), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (h->err != 0u || h->count > FMD_EXC_CAP) { *need_heavy = true; return FMD_OK; }
    const uint32_t n = h->count;
    if (n > kFirst) {
        HIP_TRY(hipMemcpyAsync(h->rec + kFirst, buf->rec + kFirst, (n - kFirst) * sizeof(FmdF64Exc), hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
    }
    std::map<uint64_t, std::pair<int64_t, const FmdF64Exc*>> groups;       // out_elem -> (delta, record)
    uint64_t patched = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const FmdF64Exc& e = h->rec[i];
        const int16_t want = (int16_t)host_polar_f64(e.cr, e.ci), have = (int16_t)e.d_gpu;
        if (want == have) continue;
        if (e.seq != s || e.k < 0) { *need_heavy = true; return FMD_OK; }     // an older launch's record, or the carried sum: everything idle first
        auto& g = groups[e.out_elem];
        g.first += (int)want - (int)have; g.second = &e;
        patched += 1;
    }
    for (const auto& kv : groups) {
        const FmdF64Exc& e = *kv.second.second;
        const int16_t fixed = (int16_t)((e.sum + (int)kv.second.first) / d->r.R);         // low_pass_real, simple_fm.rs:421
        HIP_TRY(hipMemcpyAsync((void*)(uintptr_t)e.out_elem, &fixed, sizeof(fixed), hipMemcpyHostToDevice, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));            // (`fixed` is a stack variable)
    }
    HIP_TRY(hipMemsetAsync(buf, 0, 16, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    clear_report_flag(d, s);
    d->f64_guarded += n;
    d->f64_patched += patched;
    return FMD_OK;
}

// Everything the handle has enqueued has completed (the caller synchronised): settle every launch of the ring that is not settled
// yet, the oldest first.  host_out: the caller's host copy of the NEWEST launch's output, or nullptr.
int resolve_device_reports(fmd_demod* d, int16_t* host_out, size_t host_cap)
{
    if (d->exc_override)                                     // (the pipelined sink owns its report buffers and settles them itself)
        return fmd_internal_resolve_exc(d->d_exc, d->r.R, d->seq, d->seq, d->d_state[d->cur], host_out, host_cap, &d->f64_guarded, &d->f64_patched);
    const uint32_t newest = d->seq;
    int rc = FMD_OK;
    for (uint32_t back = fmd_demod::kRing; back-- > 0;) {
        if (newest < back + 1u) continue;
        const uint32_t s = newest - back;
        fmd_demod::Pending& P = d->pend[s % fmd_demod::kRing];
        if (!P.valid || P.seq != s || P.settled) continue;
        bool carried = false;
        const int r1 = settle_launch(d, s, host_out, host_cap, &carried);
        P.settled = true;
        if (r1 && !rc) rc = r1;
        if (r1 == FMD_OK && carried) {
            // the launches behind s ran on a carried sum that has just been corrected: drop what they reported and run them again,
            // oldest first; the loop then settles them like any other launch
            for (uint32_t t = s + 1u; t <= newest; ++t) {
                fmd_demod::Pending& T = d->pend[t % fmd_demod::kRing];
                if (!T.valid || T.seq != t || T.settled) { set_err("launch %u ran on a carried sum that launch %u's check has corrected, but it has been settled (delivered) already: settle launches in order", t, s); return FMD_ERR_HIP; }
                if (host_out && t == newest) { set_err("internal: replay with a host copy"); return FMD_ERR_HIP; }   // (host entry points settle every call: unreachable)
                HIP_TRY(hipMemset(d->d_exc + t % fmd_demod::kRing, 0, 16));
                clear_report_flag(d, t);
                const int r2 = replay_launch(d, T);
                if (r2) return r2;
                T.settled = false;
            }
        }
    }
    return rc;
}
}  // namespace

extern "C" {

int fmd_version(void) { return FMD_VERSION_MAJOR * 1000 + FMD_VERSION_MINOR; }

const char* fmd_last_error(void) { return g_err; }

const char* fmd_strerror(int status)
{
    switch (status) {
        case FMD_OK: return "ok";
        case FMD_ERR_INVALID_ARG: return "invalid argument";
        case FMD_ERR_BAD_LENGTH: return "buffer length is not a multiple of 8 bytes";
        case FMD_ERR_TOO_SHORT: return "buffer yields fewer than two decimated samples";
        case FMD_ERR_BAD_RATES: return "rate_out / rate_resample / downsample invalid";
        case FMD_ERR_CAPACITY: return "output capacity too small";
        case FMD_ERR_UNSUPPORTED: return "configuration outside the supported domain";
        case FMD_ERR_BAD_STATE: return "state is not reachable by a Demod";
        case FMD_ERR_NO_DEVICE: return "no usable gfx950 device";
        case FMD_ERR_HIP: return "HIP runtime error";
        case FMD_ERR_NOMEM: return "out of memory";
        case FMD_ERR_IO: return "rtl_tcp source: socket error";
        default: return "unknown status";
    }
}

int fmd_device_count(int* count)
{
    if (!count) return FMD_ERR_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; set_err("hipGetDeviceCount: %s", hipGetErrorString(e)); return FMD_ERR_NO_DEVICE; }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    *count = ok;
    return FMD_OK;
}

int fmd_optimal_settings(uint32_t freq, uint32_t rate, uint32_t rate_resample, fmd_radio_config* radio,
                         fmd_demod_config* demod)
{
    if (rate == 0) { set_err("rate == 0 (simple_fm.rs:190 divides by it)"); return FMD_ERR_BAD_RATES; }
    const uint32_t downsample = 1000000u / rate + 1;          // simple_fm.rs:190
    const uint32_t capture_rate = downsample * rate;          // :192
    const uint32_t capture_freq = freq + capture_rate / 4;    // :195
    uint32_t output_scale = (1u << 15) / (128u * downsample); // :197
    if (output_scale < 1) output_scale = 1;                   // :198-200
    if (radio) { radio->capture_freq = capture_freq; radio->capture_rate = capture_rate; }
    if (demod) {
        demod->rate_in = rate; demod->rate_out = rate; demod->rate_resample = rate_resample;   // :207-209
        demod->downsample = downsample; demod->output_scale = output_scale;                    // :210-211
    }
    return FMD_OK;
}

size_t fmd_out_cap(const fmd_demod_config* config, size_t nbytes)
{
    if (!config || config->downsample == 0 || config->rate_out == 0) return 0;
    const uint64_t M = (nbytes / 2 + config->downsample - 1) / config->downsample + 1;
    return (size_t)((M * config->rate_resample + config->rate_out - 1) / config->rate_out + 1);
}

int fmd_demod_new(const fmd_demod_config* config, const fmd_device_config* dev, fmd_demod** out)
{
    if (!config || !dev || !out) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    *out = nullptr;
    if (dev->n_channels == 0) { set_err("n_channels == 0"); return FMD_ERR_INVALID_ARG; }
    if (config->downsample == 0 || config->rate_resample == 0 || config->rate_out < config->rate_resample) {
        set_err("need downsample >= 1 and rate_out >= rate_resample >= 1 (simple_fm.rs:421 divides by rate_out/rate_resample)");
        return FMD_ERR_BAD_RATES;
    }
    if (config->downsample > FMD_MAX_DOWNSAMPLE_WIDE) {
        set_err("downsample %u > %u unsupported", config->downsample, FMD_MAX_DOWNSAMPLE_WIDE);
        return FMD_ERR_UNSUPPORTED;
    }
    fmd_demod* d = new (std::nothrow) fmd_demod();
    if (!d) return FMD_ERR_NOMEM;
    d->cfg = *config;
    d->C = dev->n_channels;
    FmdRates& r = d->r;
    r.D = config->downsample; r.fast = config->rate_out; r.slow = config->rate_resample;
    r.g = fmd_gcd(r.fast, r.slow); r.fr = r.fast / r.g; r.sr = r.slow / r.g;
    r.R = (int32_t)(r.fast / r.slow);
    if (r.fr > FMD_MAX_RATE_REDUCED) {
        set_err("rate_out / gcd = %u > 2^24 unsupported", r.fr);
        delete d; return FMD_ERR_UNSUPPORTED;
    }
    // knobs: -DFMD_EXPERIMENT builds only (fmd_host.h); the shipped library takes the defaults, read once here
    d->force_generic = fmd_knob_u32("FMD_FORCE_GENERIC", 0) != 0;
    d->xcd_swizzle = fmd_knob_u32("FMD_XCD", 2);
    d->dbg = fmd_knob_u32("FMD_DBG", 0);
    const uint32_t kt_env = fmd_knob_u32("FMD_KT", 0);
    d->allow_fast = fmd_knob_u32("FMD_FAST", 2);
    int rc = choose_tiling(d, kt_env);
    if (rc) { delete d; return rc; }
    // Register-streaming kernel (fmd_demod_stream_kernel): downsample 2 and 4, where a lane's two adjacent windows are ONE
    // 8- / 16-byte load and the 64 lanes of a wave read 512 / 1024 contiguous bytes per round (at 6, 10, 12 a lane's span
    // is 24 / 40 / 48 bytes: strided 16-byte loads touch 1.5 - 3 x the cache lines and measured 25 - 70 % SLOWER than the
    // LDS-DMA kernel).  LDS only holds the discriminator samples there, so the tile is sized for the work per block: the
    // least wave-instructions per audio sample under the kernel's quantisation -- whole rounds of 127 decimated samples
    // per wave (at most FMD_STREAM_MAX_ROUNDS(downsample), straight-line code), the busiest of the 4 waves sets the pace, one
    // resampler pass per 256 audio samples, a fixed cost per tile.
    d->stream_ok = false;
    if ((r.D == 2u || r.D == 4u) && fmd_knob_u32("FMD_STREAM", 1) != 0u) {   // (knob: A/B in the experiment build)
        FmdRates rs = r;
        uint32_t kts = fmd_knob_u32("FMD_KT_STREAM", 0);
        const uint32_t nw = 4u;                              // waves per block (one wave per tile: measured in round 6, slower, deleted)
        const uint32_t cap_cnt = nw * 127u * FMD_STREAM_MAX_ROUNDS(r.D) - 8u;
        if (kts == 0u) {
            double best = 0.0;
            for (uint32_t k = 64; k <= 4096u; k += 32u) {
                rs.kt = k;
                if (fmd_tile_lp_cap(rs) > cap_cnt || (uint64_t)rs.sr * (k + 2) >= (1u << 24)) break;
                const uint64_t cnt = ((uint64_t)k * r.fr + r.sr - 1) / r.sr + 1;
                const uint64_t rounds = (cnt + 126) / 127, per_wave = (rounds + nw - 1) / nw, passes = (k + 64 * nw - 1) / (64 * nw);
                // (the fixed cost of a tile: downsample 2 -- twice the rounds per byte, half the bytes per round -- measured best at the
                //  largest tile its rounds admit, 384 instead of 256 audio samples at 500 k -> 32 k: -2.5 %; downsample 4 at the model's
                //  choice: profiles/r05_experiments.md 15)
                const double fixed = r.D == 2u ? 250.0 : 150.0;
                const double work = (double)per_wave * nw * 100.0 + (double)passes * nw * (70.0 + 6.0 * (double)(r.fr / r.sr)) + nw * fixed;
                const double per_audio = work / (double)k;
                if (kts == 0u || per_audio < best) { best = per_audio; kts = k; }
            }
            if (kts == 0u) kts = r.kt;
        }
        rs.kt = kts;
        while (rs.kt > 1u && fmd_tile_lp_cap(rs) > cap_cnt) rs.kt -= 1u;
        const uint32_t lp = fmd_tile_lp_cap(rs);
        const bool fits = (uint64_t)rs.sr * (rs.kt + 2) < (1u << 24) && 2ull * (lp + r.fr / r.sr + 3) + 48 <= 60u * 1024u;
        if (fits && fmd_tile_kernel_supports(rs, 0u)) { d->rs = rs; d->lp_cap_s = lp; d->stream_ok = true; }
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_err("no HIP device (this library has no CPU path)");
        delete d; return FMD_ERR_NO_DEVICE;
    }
    int device = dev->device_id;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) { set_err("device_id %d out of range (%d devices)", device, ndev); delete d; return FMD_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err("device %d is not gfx950", device);
        delete d; return FMD_ERR_NO_DEVICE;
    }
    d->device = device;
    d->n_cus = prop.multiProcessorCount;
    reset_classes(d);

    auto fail = [&](hipError_t e, const char* what) {
        set_err("%s: %s", what, hipGetErrorString(e));
        fmd_demod_free(d);
        return e == hipErrorOutOfMemory ? FMD_ERR_NOMEM : FMD_ERR_HIP;
    };
    hipError_t e;
    FmdDeviceGuard guard(device);                             // the caller's current device comes back on return
    if ((e = guard.error()) != hipSuccess) return fail(e, "hipSetDevice");
    if (const char* g = fmd_knob("FMD_F64_GUARD_LOG2")) d->f64_guard = ldexp(1.0, atoi(g));   // experiment build only: the
    d->f64_skew = fmd_knob_i32("FMD_F64_SKEW", 0);                                              // shipped guard band is 2^-20, fixed
    const size_t sbytes = sizeof(FmdChanState) * (size_t)d->C;
    for (uint32_t i = 0; i < fmd_demod::kRing; ++i) {
        if ((e = hipMalloc(&d->d_state[i], sbytes)) != hipSuccess) return fail(e, "hipMalloc(state)");
        if ((e = hipMemset(d->d_state[i], 0, sbytes)) != hipSuccess) return fail(e, "hipMemset(state)");
    }
    if ((e = hipMalloc(&d->d_exc, fmd_demod::kRing * sizeof(FmdExcBuf))) != hipSuccess) return fail(e, "hipMalloc(reports)");
    if ((e = hipMemset(d->d_exc, 0, fmd_demod::kRing * sizeof(FmdExcBuf))) != hipSuccess) return fail(e, "hipMemset(reports)");
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&d->h_head), 16 * fmd_demod::kRing, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc(report head)");
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&d->h_recs), sizeof(FmdExcBuf), hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc(report copy)");
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&d->h_mbox), 64, hipHostMallocMapped)) != hipSuccess) return fail(e, "hipHostMalloc(mailbox)");
    memset(d->h_mbox, 0, 64);
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d->d_mbox), d->h_mbox, 0)) != hipSuccess) return fail(e, "hipHostGetDevicePointer(mailbox)");
    if ((e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "hipDeviceSynchronize");
    *out = d;
    return FMD_OK;
}

void fmd_demod_free(fmd_demod* d)
{
    if (!d) return;
    FmdDeviceGuard guard(d->device);
    (void)hipDeviceSynchronize();
    for (uint32_t i = 0; i < fmd_demod::kRing; ++i) if (d->d_state[i]) (void)hipFree(d->d_state[i]);
    if (d->d_exc) (void)hipFree(d->d_exc);
    if (d->h_head) (void)hipHostFree(d->h_head);
    if (d->h_mbox) (void)hipHostFree(d->h_mbox);
    if (d->h_recs) (void)hipHostFree(d->h_recs);
    d->order.destroy();
    if (d->d_chan_class) (void)hipFree(d->d_chan_class);
    if (d->d_iq) (void)hipFree(d->d_iq);
    if (d->d_out) (void)hipFree(d->d_out);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

int fmd_demod_reset(fmd_demod* d)
{
    if (!d) return FMD_ERR_INVALID_ARG;
    ON_DEVICE(d->device);
    HIP_TRY(hipDeviceSynchronize());
    const size_t sbytes = sizeof(FmdChanState) * (size_t)d->C;
    for (uint32_t i = 0; i < fmd_demod::kRing; ++i) {
        HIP_TRY(hipMemset(d->d_state[i], 0, sbytes));
        HIP_TRY(hipMemset(d->d_exc + i, 0, 16));
        clear_report_flag(d, i);
    }
    HIP_TRY(hipDeviceSynchronize());
    for (uint32_t i = 0; i < fmd_demod::kRing; ++i) d->pend[i] = fmd_demod::Pending{};
    d->cur = 0;
    d->order.reset();
    reset_classes(d);
    return FMD_OK;
}

int fmd_demod_demodulate_device(fmd_demod* d, const void* d_iq, size_t nbytes, void* d_out, size_t out_cap,
                                void* d_out_len, void* stream)
{
    if (!d || !d_iq || !d_out) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    ON_DEVICE(d->device);
    return enqueue(d, d_iq, nbytes, d_out, out_cap, d_out_len, static_cast<hipStream_t>(stream));
}

int fmd_demod_demodulate_batch(fmd_demod* d, const uint8_t* iq, size_t nbytes, int16_t* out, size_t out_cap,
                               size_t* out_len)
{
    if (!d || !iq || !out || !out_len) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    ON_DEVICE(d->device);
    int rc;
    {   // validate before touching the staging buffers
        std::vector<FmdClassPlan> plans; uint32_t tiles;
        rc = plan_call(d, d->r, nbytes, out_cap, plans, &tiles);
        if (rc) return rc;
    }
    const size_t in_bytes = nbytes * (size_t)d->C;
    const size_t out_elems = out_cap * (size_t)d->C;
    if (in_bytes > d->d_iq_cap) {
        if (d->d_iq) { HIP_TRY(hipFree(d->d_iq)); d->d_iq = nullptr; d->d_iq_cap = 0; }
        HIP_TRY(hipMalloc(&d->d_iq, in_bytes));
        d->d_iq_cap = in_bytes;
    }
    if (out_elems > d->d_out_cap) {
        if (d->d_out) { HIP_TRY(hipFree(d->d_out)); d->d_out = nullptr; d->d_out_cap = 0; }
        HIP_TRY(hipMalloc(&d->d_out, (out_elems ? out_elems : 1) * sizeof(int16_t)));
        d->d_out_cap = out_elems;
    }
    HIP_TRY(hipMemcpyAsync(d->d_iq, iq, in_bytes, hipMemcpyHostToDevice, d->stream));
    rc = enqueue(d, d->d_iq, nbytes, d->d_out, out_cap, nullptr, d->stream);
    if (rc) return rc;
    uint32_t kmax = 0;
    for (uint32_t c = 0; c < d->C; ++c) {
        const uint32_t K = d->classes[d->chan_class[c]].last_K;
        out_len[c] = K;
        if (K > kmax) kmax = K;
    }
    if (kmax && (size_t)kmax * 2 >= out_cap) {
        // rows are (nearly) full: one linear copy of the whole block beats a strided one by an order of magnitude
        HIP_TRY(hipMemcpyAsync(out, d->d_out, out_elems * sizeof(int16_t), hipMemcpyDeviceToHost, d->stream));
    } else if (kmax) {
        HIP_TRY(hipMemcpy2DAsync(out, out_cap * sizeof(int16_t), d->d_out, out_cap * sizeof(int16_t),
                                 (size_t)kmax * sizeof(int16_t), d->C, hipMemcpyDeviceToHost, d->stream));
    }
    const bool head = d->h_head && !d->exc_override;           // the report head rides behind the output copy (see fmd_demod_check)
    bool older_clean = true;                                  // (a caller may mix the host and the _device entry points)
    for (uint32_t i = 0; i < fmd_demod::kRing; ++i) if (i != d->seq % fmd_demod::kRing && d->pend[i].valid && !d->pend[i].settled) older_clean = false;
    if (head) { d->h_head[0] = d->h_head[1] = ~0u; HIP_TRY(hipMemcpyAsync(d->h_head, d->d_exc + d->seq % fmd_demod::kRing, 16, hipMemcpyDeviceToHost, d->stream)); }
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (head && older_clean && d->h_head[0] == 0u && d->h_head[1] == 0u) { d->pend[d->seq % fmd_demod::kRing].settled = true; return FMD_OK; }
    if (!older_clean) HIP_TRY(hipDeviceSynchronize());
    return resolve_device_reports(d, out, out_cap);   // device assertions + guarded f64 samples (patched in `out`)
}

int fmd_demod_demodulate(fmd_demod* d, const uint8_t* iq, size_t nbytes, int16_t* out, size_t out_cap,
                         size_t* out_len)
{
    if (!d) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (d->C != 1) { set_err("fmd_demod_demodulate needs a 1-channel handle (this one has %u)", d->C); return FMD_ERR_INVALID_ARG; }
    return fmd_demod_demodulate_batch(d, iq, nbytes, out, out_cap, out_len);
}

int fmd_host_alloc(size_t nbytes, void** ptr)
{
    if (!ptr || nbytes == 0) { set_err("null pointer or zero size"); return FMD_ERR_INVALID_ARG; }
    *ptr = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_err("no HIP device (this library has no CPU path)");
        return FMD_ERR_NO_DEVICE;
    }
    // portable: usable by every device of the node (one handle per GPU may share a reader thread's buffers)
    HIP_TRY(hipHostMalloc(ptr, nbytes, hipHostMallocPortable));
    return FMD_OK;
}

int fmd_host_free(void* ptr)
{
    if (!ptr) return FMD_OK;
    HIP_TRY(hipHostFree(ptr));
    return FMD_OK;
}

int fmd_demod_set_block_len(fmd_demod* d, size_t block_bytes)
{
    if (!d) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (block_bytes == 0) { d->block_ns = 0; return FMD_OK; }
    if (block_bytes % 8 != 0) { set_err("block_bytes %% 8 != 0 (simple_fm.rs:286 would panic on every block)"); return FMD_ERR_BAD_LENGTH; }
    if (block_bytes / 2 < 2ull * d->r.D) {
        set_err("a block of %zu bytes yields fewer than 2 decimated samples at downsample %u (simple_fm.rs:356)", block_bytes, d->r.D);
        return FMD_ERR_TOO_SHORT;
    }
    if (block_bytes / 2 > (1ull << 30)) { set_err("block_bytes out of range"); return FMD_ERR_UNSUPPORTED; }
    d->block_ns = (uint32_t)(block_bytes / 2);
    return FMD_OK;
}

int fmd_demod_last_out_len(const fmd_demod* d, size_t* out_len)
{
    if (!d || !out_len) return FMD_ERR_INVALID_ARG;
    for (uint32_t c = 0; c < d->C; ++c) out_len[c] = d->classes[d->chan_class[c]].last_K;
    return FMD_OK;
}

int fmd_demod_check(fmd_demod* d)
{
    if (!d) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    ON_DEVICE(d->device);
    // The common case -- no device assertion, no guarded f64 sample -- costs one synchronisation with the handle's most recent launch
    // (the library orders a handle's launches across streams itself, so that launch's completion is the handle's; event mode: its
    // event, the caller's stream is not touched) and three reads of host memory: a launch that writes a record or an error bit into
    // its report buffer also sets its host-mapped word (FmdLaunch::hflag), so nothing is copied unless something was reported
    // (rounds 2 - 5 copied the buffer's head behind the launch: +4.4 us per call, profiles/r06_experiments.md 11).
    if (d->order.have_last && d->h_mbox && !d->exc_override) {
        const hipError_t e = d->order.wait_last();
        if (e == hipSuccess) {
            auto flag = [&](uint32_t i) { return __atomic_load_n(d->h_mbox + fmd_demod::kFlagWord + i % fmd_demod::kRing, __ATOMIC_ACQUIRE); };
            // a launch that DID report: its records are evaluated here, oldest launch first (settle_launch_light -- the usual guarded
            // sample agrees with the host's value and costs two small copies); anything else takes the device-wide path below
            bool heavy = false;
            const uint32_t newest = d->seq;
            for (uint32_t back = fmd_demod::kRing; back-- > 0 && !heavy;) {
                if (newest < back + 1u) continue;
                const uint32_t s = newest - back;
                fmd_demod::Pending& P = d->pend[s % fmd_demod::kRing];
                if (!P.valid || P.seq != s || P.settled) continue;
                if (flag(s) != 0u) {
                    if (!d->h_recs) { heavy = true; break; }
                    const int rl = settle_launch_light(d, s, &heavy);
                    if (rl) return rl;
                    if (heavy) break;
                }
                P.settled = true;
            }
            for (uint32_t i = 0; i < fmd_demod::kRing; ++i) heavy = heavy || flag(i) != 0u;       // (a sticky error bit, a slot without a launch record)
            if (!heavy) return FMD_OK;
        } else {
            (void)hipGetLastError();
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    return resolve_device_reports(d, nullptr, 0);
}

int fmd_demod_check_behind(fmd_demod* d, uint32_t back)
{
    if (!d) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (back == 0u) return fmd_demod_check(d);
    if (back >= fmd_demod::kRing) { set_err("fmd_demod_check_behind: back must be 0, 1 or 2"); return FMD_ERR_INVALID_ARG; }
    ON_DEVICE(d->device);
    const uint32_t newest = d->seq;
    if (newest < back + 1u) return FMD_OK;                   // nothing that far back yet
    const uint32_t target = newest - back;
    fmd_demod::Pending& pt = d->pend[target % fmd_demod::kRing];
    if (!pt.valid || pt.seq != target || pt.settled) return FMD_OK;
    // Launch target + 1 posts the completion of launch `target` (and whether its report buffer is empty) when its first tile runs:
    // spin on the host-mapped word.  No post is coming if that launch is not one that posts (generic kernel, a caller-owned report
    // buffer); and launches are settled IN ORDER (a correction of an older launch's carried sum runs the newer ones again: they must
    // not have been delivered): with an older launch still unsettled this is fmd_demod_check.
    const fmd_demod::Pending& pnext = d->pend[(target + 1u) % fmd_demod::kRing];
    bool in_order = pnext.valid && pnext.seq == target + 1u && pnext.posts && d->h_mbox != nullptr;
    for (uint32_t i = 0; i < fmd_demod::kRing; ++i) if (d->pend[i].valid && !d->pend[i].settled && (int32_t)(d->pend[i].seq - target) < 0) in_order = false;
    if (!in_order) return fmd_demod_check(d);
    const uint64_t* const m = reinterpret_cast<const uint64_t*>(d->h_mbox) + target % fmd_demod::kRing;
    // (the wait is a plain spin on the host-mapped word; only after a millisecond without the post does it start asking the runtime
    //  -- once per further millisecond -- whether the newest launch is still running at all: a failed launch would never post)
    uint64_t spins = 0, word;
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    double next_query_s = 1.0e-3;
    while ((int32_t)((uint32_t)(word = __atomic_load_n(m, __ATOMIC_ACQUIRE)) - target) < 0) {
        if ((++spins & 0xFFu) == 0u) {
            struct timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            const double waited = (double)(ts.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts.tv_nsec - ts0.tv_nsec);
            if (waited >= next_query_s) {
                next_query_s = waited + 1.0e-3;
                const hipError_t q = d->order.query_last();
                if (q == hipSuccess) {                       // the newest launch has completed too -- the post is there, or never comes
                    if ((int32_t)((uint32_t)(word = __atomic_load_n(m, __ATOMIC_ACQUIRE)) - target) < 0) return fmd_demod_check(d);
                    break;
                }
                if (q != hipErrorNotReady) { set_err("querying the newest launch: %s", hipGetErrorString(q)); (void)hipGetLastError(); return FMD_ERR_HIP; }
            }
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    if ((uint32_t)word == target && (word >> 32) == 0u) { pt.settled = true; return FMD_OK; }   // the common case: nothing to settle
    // Something was reported by launch `target`: guarded f64 samples (settled without waiting for the newer launches unless one of them
    // corrects the carried sum: settle_launch_light), or a device assertion.
    if ((uint32_t)word == target && d->h_recs) {
        bool need_heavy = false;
        const int rl = settle_launch_light(d, target, &need_heavy);
        if (rl) return rl;
        if (!need_heavy) { pt.settled = true; return FMD_OK; }
    }
    // The rare path waits for the newer launches as well and settles the ring in order (settle_launch: patches go into the output
    // buffer of the launch that produced them -- all three are still the caller's to keep -- and a corrected carried sum re-runs the
    // launches behind it).
    HIP_TRY(hipDeviceSynchronize());
    return resolve_device_reports(d, nullptr, 0);
}

int fmd_demod_check_prev(fmd_demod* d) { return fmd_demod_check_behind(d, 1u); }

int fmd_demod_set_event_ordering(fmd_demod* d, int on)
{
    if (!d) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    ON_DEVICE(d->device);
    HIP_TRY(hipDeviceSynchronize());                         // the switch happens between launches, with nothing in flight
    {
        const int rr = resolve_device_reports(d, nullptr, 0);
        if (rr) return rr;
    }
    if (on) HIP_TRY(d->order.ensure_event());
    d->order.event_mode = on != 0;
    d->order.reset();
    return FMD_OK;
}

int fmd_demod_f64_stats(const fmd_demod* d, uint64_t* guarded, uint64_t* patched)
{
    if (!d) return FMD_ERR_INVALID_ARG;
    if (guarded) *guarded = d->f64_guarded;
    if (patched) *patched = d->f64_patched;
    return FMD_OK;
}

int fmd_demod_get_state(fmd_demod* d, uint32_t channel, fmd_demod_state* state)
{
    if (!d || !state || channel >= d->C) { set_err("bad argument"); return FMD_ERR_INVALID_ARG; }
    ON_DEVICE(d->device);
    HIP_TRY(hipDeviceSynchronize());
    int rc = resolve_device_reports(d, nullptr, 0);
    if (rc) return rc;
    FmdChanState s;
    HIP_TRY(hipMemcpy(&s, d->d_state[d->cur] + channel, sizeof(s), hipMemcpyDeviceToHost));
    state->prev_index = s.prev_index;
    state->now_lpr = s.now_lpr;
    state->prev_lpr_index = (int32_t)(s.lpr_index_r * d->r.g);
    state->lp_now_re = s.lp_now_re; state->lp_now_im = s.lp_now_im;
    state->demod_pre_re = s.demod_pre_re; state->demod_pre_im = s.demod_pre_im;
    return FMD_OK;
}

int fmd_demod_set_state(fmd_demod* d, uint32_t channel, const fmd_demod_state* state)
{
    if (!d || !state || channel >= d->C) { set_err("bad argument"); return FMD_ERR_INVALID_ARG; }
    const FmdRates& r = d->r;
    const int64_t lim_lp = 128ll * state->prev_index, lim_pre = 128ll * r.D;
    // |d| <= 16384 per discriminator sample -- except beyond downsample 128, where the reference's i32 products wrap and
    // `pcm as i16` (simple_fm.rs:362) can be anything in +-32768: a state the handle itself produced must be restorable
    const int64_t lim_lpr = (r.D > FMD_MAX_DOWNSAMPLE ? 32768ll : 16384ll) * ((r.fr + r.sr - 1) / r.sr);
    auto within = [](int64_t v, int64_t lim) { return v >= -lim && v <= lim; };
    if (state->prev_index >= r.D || state->prev_lpr_index < 0 || (uint32_t)state->prev_lpr_index >= r.fast ||
        (uint32_t)state->prev_lpr_index % r.g != 0 || !within(state->lp_now_re, lim_lp) ||
        !within(state->lp_now_im, lim_lp) || !within(state->demod_pre_re, lim_pre) ||
        !within(state->demod_pre_im, lim_pre) || !within(state->now_lpr, lim_lpr)) {
        set_err("state not reachable from Demod::new by demodulate calls");
        return FMD_ERR_BAD_STATE;
    }
    ON_DEVICE(d->device);
    HIP_TRY(hipDeviceSynchronize());
    {
        const int rr = resolve_device_reports(d, nullptr, 0);   // a pending patch of the carried sum comes first
        if (rr) return rr;
    }
    FmdChanState s{};
    s.prev_index = state->prev_index;
    s.lpr_index_r = (uint32_t)state->prev_lpr_index / r.g;
    s.now_lpr = state->now_lpr;
    s.lp_now_re = state->lp_now_re; s.lp_now_im = state->lp_now_im;
    s.demod_pre_re = state->demod_pre_re; s.demod_pre_im = state->demod_pre_im;
    HIP_TRY(hipMemcpy(d->d_state[d->cur] + channel, &s, sizeof(s), hipMemcpyHostToDevice));
    const PhaseClass& pc = d->classes[d->chan_class[channel]];
    if (pc.p0 != s.prev_index || pc.i0r != s.lpr_index_r) regroup(d, channel, s.prev_index, s.lpr_index_r);
    return FMD_OK;
}

int fmd_demod_tiling(const fmd_demod* d, uint32_t* audio_per_tile, uint32_t* lds_bytes, uint32_t* block_threads)
{
    if (!d) return FMD_ERR_INVALID_ARG;
    if (block_threads) *block_threads = FMD_BLOCK_THREADS;
    if (d->last_kernel.family != FMD_KERNEL_NONE) {          // what the most recent launch actually ran
        if (audio_per_tile) *audio_per_tile = d->last_kernel.kt;
        if (lds_bytes) *lds_bytes = d->last_kernel.lds;
        return FMD_OK;
    }
    // before the first launch: what a bank in one phase class, fed whole read_sync buffers, will run
    const bool streaming = d->stream_ok && d->C >= 8u && d->block_ns == 0u && tile_kernel_ok(d);   // what a bank in one phase class runs
    if (audio_per_tile) *audio_per_tile = streaming ? d->rs.kt : d->r.kt;
    if (lds_bytes) {
        FmdLaunch L{}; L.raw_cap = d->raw_cap; L.lp_cap = streaming ? d->lp_cap_s : d->lp_cap; L.fa = d->r.fr / d->r.sr; L.r = d->r;
        L.stream = streaming ? 1u : 0u;
        *lds_bytes = (uint32_t)(tile_kernel_ok(d) ? fmd_tile_lds_bytes(L) : fmd_generic_lds_bytes(L));
    }
    return FMD_OK;
}

int fmd_demod_tiling_plan(const fmd_demod* d) { return d ? (int)d->tiling_plan : FMD_ERR_INVALID_ARG; }

int fmd_demod_last_kernel(const fmd_demod* d, char* name, size_t cap)
{
    if (!d || !name || cap == 0) return FMD_ERR_INVALID_ARG;
    const FmdKernelId& k = d->last_kernel;
    int n = 0;
    switch (k.family) {                                      // the names rocprofv3 --kernel-trace prints for these launches
        case FMD_KERNEL_TILE:    n = snprintf(name, cap, "fmd_tk::fmd_demod_tile_kernel<%d, %d>", (int)k.dh, (int)k.fast); break;
        case FMD_KERNEL_STREAM:  n = snprintf(name, cap, "fmd_tk::fmd_demod_stream_kernel<%d, %d>", (int)k.dh, (int)k.fast); break;
        case FMD_KERNEL_GENERIC: n = snprintf(name, cap, "fmd_demod_generic_kernel<%s>", k.dh ? "true" : "false"); break;
        default:                 n = snprintf(name, cap, "%s", ""); break;
    }
    return n < 0 || (size_t)n >= cap ? FMD_ERR_CAPACITY : FMD_OK;
}

int fmd_demod_set_tiling(fmd_demod* d, uint32_t audio_per_tile)
{
    if (!d) return FMD_ERR_INVALID_ARG;
    const FmdRates saved = d->r; const uint32_t lc = d->lp_cap, rc_ = d->raw_cap;
    int rc = choose_tiling(d, audio_per_tile);
    if (rc) { d->r = saved; d->lp_cap = lc; d->raw_cap = rc_; }
    else if (audio_per_tile) d->stream_ok = false;           // an explicit tile size means the LDS kernel with exactly that tile
    return rc;
}

int fmd_synth_fill_device(int device_id, void* d_iq, uint32_t n_channels, size_t nbytes, uint64_t sample_offset,
                          const fmd_synth_params* p, void* stream)
{
    if (!d_iq || !p || n_channels == 0) { set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (nbytes % 8 != 0) { set_err("nbytes %% 8 != 0"); return FMD_ERR_BAD_LENGTH; }
    if (p->amplitude > 120 || p->noise > 64 || p->mod_period < 2) { set_err("synth parameters out of range"); return FMD_ERR_INVALID_ARG; }
    int dev_now = device_id;
    if (dev_now < 0 && hipGetDevice(&dev_now) != hipSuccess) dev_now = 0;
    ON_DEVICE(dev_now);
    if (nbytes == 0) return FMD_OK;
    FmdSynthLaunch S{};
    S.iq = static_cast<uint8_t*>(d_iq);
    S.chan_stride = nbytes; S.n_channels = n_channels; S.sample_offset = sample_offset;
    S.seed = p->seed; S.amplitude = p->amplitude; S.noise = p->noise; S.dev_q32 = p->dev_q32; S.mod_period = p->mod_period;
    HIP_TRY(fmd_launch_synth(S, static_cast<hipStream_t>(stream)));
    return FMD_OK;
}

}  // extern "C"
