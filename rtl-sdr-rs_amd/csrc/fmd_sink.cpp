// fmd_sink.cpp -- pipelined, multi-GPU sink for read_sync buffers (include/fmd.h, fmd_sink_*).
//
// NEW SURFACE, labelled as such: the reference has no asynchronous reader (no `read_async`).  What
// it does have is the hand-off this mirrors -- `receive()` fills a buffer with RtlSdr::read_sync and sends it down an
// unbounded mpsc channel, `process()` takes it, demodulates and calls output() (examples/simple_fm.rs:55-60,
// 114-127, 150-156).  Here the channel is a ring of `depth` page-locked slots and the consumer is one or more GPUs:
//
//     caller:  acquire slot -> fill it ([n_channels][nbytes], what read_sync wrote) -> submit          (never blocks on the GPU
//     device d of D (channels [lo_d, hi_d), simple_fm.rs:137: one Demod per stream, no cross-channel term):          unless the ring is full)
//         copy-in stream :  H2D of the slot's rows lo_d..hi_d          -> event
//         compute stream :  wait, fmd_demod_demodulate_device           -> event        (state carries launch to launch)
//         copy-out stream:  wait, D2H of the audio rows                 -> event
//     completion (in submit order, from acquire / poll / drain): settle the slot's guarded f64 samples against the
//     host libm, then callback(user, seq, audio [n_channels][out_cap], out_len [n_channels]).
//
// H2D of buffer n+1, the kernel of buffer n and D2H of buffer n-1 overlap; D devices work on their channel ranges
// concurrently, driven by ONE host thread (everything is asynchronous), with no collective.  Each in-flight launch
// has its own device report buffer (fmd_internal_set_report_buffer), so settling slot n never races with the kernels
// of slots n+1 ... behind it.
#include "../../include/fmd.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "fmd_host.h"
#include "fmd_internal.h"

namespace {

struct DevPart {
    int device = 0;
    uint32_t lo = 0, hi = 0;              // channel range
    fmd_demod* demod = nullptr;
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
};

struct SlotDev {
    uint8_t* d_iq = nullptr;
    int16_t* d_out = nullptr;
    FmdExcBuf* d_exc = nullptr;
    uint32_t* h_head = nullptr;           // page-locked copy of the report buffer's header (err, count, ...), part of the pipeline
    hipEvent_t e_in = nullptr, e_k = nullptr, e_done = nullptr;
    uint32_t launch_seq = 0;              // the handle's sequence number of this slot's launch
};

struct Slot {
    uint8_t* h_iq = nullptr;              // page-locked [C][nbytes]
    int16_t* h_out = nullptr;             // page-locked [C][out_cap]
    std::vector<SlotDev> dev;
    std::vector<size_t> out_len;          // [C]
    uint64_t seq = 0;
    int state = 0;                        // 0 free, 1 acquired (being filled), 2 in flight
};

#define SK_TRY(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            char m_[256];                                                                   \
            snprintf(m_, sizeof m_, "%s failed: %s", #expr, hipGetErrorString(e_));         \
            fmd_internal_set_err(m_);                                                       \
            return e_ == hipErrorOutOfMemory ? FMD_ERR_NOMEM : FMD_ERR_HIP;                 \
        }                                                                                   \
    } while (0)

}  // namespace

struct fmd_sink {
    fmd_demod_config cfg{};
    uint32_t C = 0;
    size_t nbytes = 0, out_cap = 0;
    std::vector<DevPart> parts;
    std::vector<Slot> slots;
    uint32_t head = 0;                    // next slot to acquire
    uint32_t tail = 0;                    // oldest slot in flight
    uint32_t in_flight = 0;
    uint64_t next_seq = 0;
    fmd_sink_callback cb = nullptr;
    void* user = nullptr;
    int acquired = -1;
    int poisoned = FMD_OK;                // first error of a submit that had already touched a device: terminal (see fmd_sink_submit)
    char poison_msg[256] = "";
};

namespace {

// Finish the oldest in-flight slot: wait (or test) its D2H events, settle, deliver.  Returns 1 when a slot was
// delivered, 0 when `block` is false and it is not ready yet, < 0 on error.
int complete_oldest(fmd_sink* s, bool block)
{
    if (s->in_flight == 0) return 0;
    Slot& sl = s->slots[s->tail];
    for (size_t k = 0; k < s->parts.size(); ++k) {
        FmdDeviceGuard guard(s->parts[k].device);
        if (block) SK_TRY(hipEventSynchronize(sl.dev[k].e_done));
        else {
            const hipError_t q = hipEventQuery(sl.dev[k].e_done);
            if (q == hipErrorNotReady) return 0;
            SK_TRY(q);
        }
    }
    int status = FMD_OK;
    for (size_t k = 0; k < s->parts.size(); ++k) {
        DevPart& p = s->parts[k];
        FmdDeviceGuard guard(p.device);
        // the header travelled with the audio: nothing reported (the common case) -> no further device access, so
        // completing slot n never queues a blocking copy behind the transfers of the slots after it
        if (sl.dev[k].h_head[0] == 0u && sl.dev[k].h_head[1] == 0u) continue;
        const FmdHandleView v = fmd_internal_view(p.demod);
        // rows of this device inside the slot's host audio block; the slot's own report buffer
        const int rc = fmd_internal_resolve_exc(sl.dev[k].d_exc, v.R, sl.dev[k].launch_seq, v.seq, v.state_cur,
                                                sl.h_out + (size_t)p.lo * s->out_cap, s->out_cap, v.guarded, v.patched);
        if (rc != FMD_OK && status == FMD_OK) status = rc;
    }
    sl.state = 0;
    s->tail = (s->tail + 1) % (uint32_t)s->slots.size();
    s->in_flight -= 1;
    if (s->cb) s->cb(s->user, sl.seq, sl.h_out, sl.out_len.data(), s->out_cap, status);
    return status == FMD_OK ? 1 : status;
}

}  // namespace

extern "C" {

int fmd_sink_new(const fmd_demod_config* config, uint32_t n_channels, const int32_t* device_ids, uint32_t n_devices,
                 size_t nbytes, uint32_t depth, fmd_sink_callback callback, void* user, fmd_sink** out)
{
    if (!config || !out || n_channels == 0 || depth == 0 || depth > 64) { fmd_internal_set_err("bad argument (need n_channels >= 1, 1 <= depth <= 64)"); return FMD_ERR_INVALID_ARG; }
    *out = nullptr;
    if (nbytes == 0 || nbytes % 8 != 0) { fmd_internal_set_err("nbytes % 8 != 0 (simple_fm.rs:286 would panic)"); return FMD_ERR_BAD_LENGTH; }
    if (n_devices == 0 || (device_ids == nullptr && n_devices != 1) || n_devices > n_channels) {
        fmd_internal_set_err("need 1 <= n_devices <= n_channels and a device_ids array"); return FMD_ERR_INVALID_ARG;
    }
    fmd_sink* s = new (std::nothrow) fmd_sink();
    if (!s) return FMD_ERR_NOMEM;
    s->cfg = *config; s->C = n_channels; s->nbytes = nbytes; s->cb = callback; s->user = user;
    s->out_cap = fmd_out_cap(config, nbytes);
    if (s->out_cap == 0) { delete s; fmd_internal_set_err("bad configuration"); return FMD_ERR_BAD_RATES; }
    int rc = FMD_OK;
    s->parts.resize(n_devices);
    const uint32_t base = n_channels / n_devices, extra = n_channels % n_devices;       // contiguous blocks, like shard.channel_range
    uint32_t lo = 0;
    for (uint32_t k = 0; k < n_devices && rc == FMD_OK; ++k) {
        DevPart& p = s->parts[k];
        p.lo = lo; p.hi = lo + base + (k < extra ? 1u : 0u); lo = p.hi;
        fmd_device_config dc{p.hi - p.lo, device_ids ? device_ids[k] : -1, 0u};
        rc = fmd_demod_new(config, &dc, &p.demod);
        if (rc != FMD_OK) break;
        p.device = fmd_internal_view(p.demod).device;
        FmdDeviceGuard guard(p.device);
        if (hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&p.s_k, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking) != hipSuccess) { fmd_internal_set_err("hipStreamCreate failed"); rc = FMD_ERR_HIP; }
    }
    s->slots.resize(depth);
    for (uint32_t i = 0; i < depth && rc == FMD_OK; ++i) {
        Slot& sl = s->slots[i];
        sl.out_len.assign(n_channels, 0);
        sl.dev.resize(n_devices);
        if (hipHostMalloc((void**)&sl.h_iq, nbytes * (size_t)n_channels, hipHostMallocPortable) != hipSuccess ||
            hipHostMalloc((void**)&sl.h_out, s->out_cap * (size_t)n_channels * sizeof(int16_t), hipHostMallocPortable) != hipSuccess) {
            fmd_internal_set_err("hipHostMalloc failed"); rc = FMD_ERR_NOMEM; break;
        }
        for (uint32_t k = 0; k < n_devices && rc == FMD_OK; ++k) {
            DevPart& p = s->parts[k];
            SlotDev& sd = sl.dev[k];
            FmdDeviceGuard guard(p.device);
            const size_t nc = p.hi - p.lo;
            if (hipMalloc((void**)&sd.d_iq, nbytes * nc) != hipSuccess || hipMalloc((void**)&sd.d_out, s->out_cap * nc * sizeof(int16_t)) != hipSuccess ||
                hipMalloc((void**)&sd.d_exc, sizeof(FmdExcBuf)) != hipSuccess || hipMemset(sd.d_exc, 0, sizeof(FmdExcBuf)) != hipSuccess) {
                fmd_internal_set_err("hipMalloc failed"); rc = FMD_ERR_NOMEM; break;
            }
            if (hipHostMalloc((void**)&sd.h_head, 16, hipHostMallocPortable) != hipSuccess) { fmd_internal_set_err("hipHostMalloc failed"); rc = FMD_ERR_NOMEM; break; }
            memset(sd.h_head, 0, 16);
            if (hipEventCreateWithFlags(&sd.e_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sd.e_k, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&sd.e_done, hipEventDisableTiming) != hipSuccess) { fmd_internal_set_err("hipEventCreate failed"); rc = FMD_ERR_HIP; }
        }
    }
    if (rc == FMD_OK) {
        for (DevPart& p : s->parts) { FmdDeviceGuard guard(p.device); if (hipDeviceSynchronize() != hipSuccess) rc = FMD_ERR_HIP; }
    }
    if (rc != FMD_OK) { fmd_sink_free(s); return rc; }
    *out = s;
    return FMD_OK;
}

void fmd_sink_free(fmd_sink* s)
{
    if (!s) return;
    for (DevPart& p : s->parts) { if (p.demod) { FmdDeviceGuard guard(p.device); (void)hipDeviceSynchronize(); } }
    for (Slot& sl : s->slots) {
        for (size_t k = 0; k < sl.dev.size() && k < s->parts.size(); ++k) {
            FmdDeviceGuard guard(s->parts[k].device);
            SlotDev& sd = sl.dev[k];
            if (sd.d_iq) (void)hipFree(sd.d_iq);
            if (sd.d_out) (void)hipFree(sd.d_out);
            if (sd.d_exc) (void)hipFree(sd.d_exc);
            if (sd.h_head) (void)hipHostFree(sd.h_head);
            if (sd.e_in) (void)hipEventDestroy(sd.e_in);
            if (sd.e_k) (void)hipEventDestroy(sd.e_k);
            if (sd.e_done) (void)hipEventDestroy(sd.e_done);
        }
        if (sl.h_iq) (void)hipHostFree(sl.h_iq);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
    }
    for (DevPart& p : s->parts) {
        if (!p.demod) continue;
        {
            FmdDeviceGuard guard(p.device);
            if (p.s_in) (void)hipStreamDestroy(p.s_in);
            if (p.s_k) (void)hipStreamDestroy(p.s_k);
            if (p.s_out) (void)hipStreamDestroy(p.s_out);
        }
        fmd_demod_free(p.demod);
    }
    delete s;
}

int fmd_sink_acquire(fmd_sink* s, uint8_t** iq)
{
    if (!s || !iq) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (s->poisoned != FMD_OK) { fmd_internal_set_err(s->poison_msg); return s->poisoned; }
    if (s->acquired >= 0) { fmd_internal_set_err("a slot is already acquired: submit it first"); return FMD_ERR_INVALID_ARG; }
    // deliver whatever has finished; if the ring is full, wait for the oldest launch
    for (;;) {
        const int r = complete_oldest(s, s->in_flight == s->slots.size());
        if (r < 0) return r;
        if (r == 0) break;
    }
    Slot& sl = s->slots[s->head];
    sl.state = 1;
    s->acquired = (int)s->head;
    *iq = sl.h_iq;
    return FMD_OK;
}

// receive() (simple_fm.rs:89-132) for a bank of rtl_tcp streams: one slot per call, all sockets behind one poll().
int fmd_sink_fill_from_rtltcp(fmd_sink* s, fmd_rtltcp* const* sources, uint32_t n_sources, uint32_t* n_short)
{
    if (!s || !sources || !n_short) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    *n_short = 0;
    if (n_sources != s->C) { fmd_internal_set_err("one rtl_tcp source per channel of the sink"); return FMD_ERR_INVALID_ARG; }
    uint8_t* iq = nullptr;
    int rc = fmd_sink_acquire(s, &iq);
    if (rc != FMD_OK) return rc;
    std::vector<size_t> got(n_sources, 0);
    rc = fmd_rtltcp_read_many(sources, n_sources, iq, s->nbytes, s->nbytes, got.data());
    uint32_t shorts = 0;
    for (uint32_t c = 0; c < n_sources; ++c) shorts += got[c] < s->nbytes ? 1u : 0u;
    if (rc != FMD_OK || shorts) {                            // socket error / timeout, or "samples lost": nothing is submitted
        (void)fmd_sink_release(s);
        *n_short = shorts;
        return rc;
    }
    return fmd_sink_submit(s);
}

int fmd_sink_pump_rtltcp(fmd_sink* s, fmd_rtltcp* const* sources, uint32_t n_sources, uint64_t max_buffers, uint64_t* n_submitted)
{
    if (!n_submitted) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    *n_submitted = 0;
    int rc = FMD_OK;
    while (max_buffers == 0 || *n_submitted < max_buffers) {
        uint32_t shorts = 0;
        rc = fmd_sink_fill_from_rtltcp(s, sources, n_sources, &shorts);
        if (rc != FMD_OK || shorts) break;
        *n_submitted += 1;
    }
    const int rd = fmd_sink_drain(s);                        // everything submitted is delivered, also after an error
    return rc != FMD_OK ? rc : rd;
}

// Give an acquired slot back without submitting it (a short read ends the run, simple_fm.rs:122-125): the sink stays usable.
int fmd_sink_release(fmd_sink* s)
{
    if (!s) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (s->acquired < 0) { fmd_internal_set_err("no slot acquired"); return FMD_ERR_INVALID_ARG; }
    s->slots[(size_t)s->acquired].state = 0;
    s->acquired = -1;
    return FMD_OK;
}

}  // extern "C"

namespace {

// One device part of a submit.  Returns the first error; everything before it stays enqueued.
int submit_part(fmd_sink* s, Slot& sl, size_t k)
{
    DevPart& p = s->parts[k];
    SlotDev& sd = sl.dev[k];
    FmdDeviceGuard guard(p.device);
    const size_t nc = p.hi - p.lo;
    {
        SK_TRY(hipMemcpyAsync(sd.d_iq, sl.h_iq + (size_t)p.lo * s->nbytes, s->nbytes * nc, hipMemcpyHostToDevice, p.s_in));
        SK_TRY(hipEventRecord(sd.e_in, p.s_in));
        SK_TRY(hipStreamWaitEvent(p.s_k, sd.e_in, 0));
        fmd_internal_set_report_buffer(p.demod, sd.d_exc);
        const int rc = fmd_demod_demodulate_device(p.demod, sd.d_iq, s->nbytes, sd.d_out, s->out_cap, nullptr, p.s_k);
        fmd_internal_set_report_buffer(p.demod, nullptr);
        if (rc != FMD_OK) return rc;
        sd.launch_seq = fmd_internal_view(p.demod).seq;
        SK_TRY(hipEventRecord(sd.e_k, p.s_k));
        SK_TRY(hipStreamWaitEvent(p.s_out, sd.e_k, 0));
        SK_TRY(hipMemcpyAsync(sl.h_out + (size_t)p.lo * s->out_cap, sd.d_out, s->out_cap * nc * sizeof(int16_t), hipMemcpyDeviceToHost, p.s_out));
        SK_TRY(hipMemcpyAsync(sd.h_head, sd.d_exc, 16, hipMemcpyDeviceToHost, p.s_out));
        SK_TRY(hipEventRecord(sd.e_done, p.s_out));
    }
    return fmd_demod_last_out_len(p.demod, sl.out_len.data() + p.lo);
}

}  // namespace

extern "C" {

// A submit is all-or-nothing per buffer only up to its first enqueue: once one device part has advanced its Demod state
// (or a copy out of the page-locked slot is in flight), a failure on a later part cannot be rolled back -- feeding the
// buffer again would run it through part 0 twice and the audio would silently diverge from the reference's.  Such a
// failure is therefore TERMINAL for the sink: everything already enqueued is waited for (so the caller may reuse the
// slot's memory), the slot is dropped, and this and every later acquire / submit return the same error; poll / drain
// still deliver the buffers submitted before it, then return it too.  fmd_sink_free is the only way on.
int fmd_sink_submit(fmd_sink* s)
{
    if (!s) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (s->poisoned != FMD_OK) { fmd_internal_set_err(s->poison_msg); return s->poisoned; }
    if (s->acquired < 0) { fmd_internal_set_err("no slot acquired"); return FMD_ERR_INVALID_ARG; }
    Slot& sl = s->slots[(size_t)s->acquired];
    for (size_t k = 0; k < s->parts.size(); ++k) {
        const int rc = submit_part(s, sl, k);
        if (rc == FMD_OK) continue;
        snprintf(s->poison_msg, sizeof s->poison_msg, "sink failed in submit (device part %zu of %zu): %s", k, s->parts.size(), fmd_last_error());
        for (DevPart& p : s->parts) { FmdDeviceGuard guard(p.device); (void)hipDeviceSynchronize(); }   // nothing reads the slot any more
        s->poisoned = rc;
        sl.state = 0;
        s->acquired = -1;
        fmd_internal_set_err(s->poison_msg);
        return rc;
    }
    sl.seq = s->next_seq++;
    sl.state = 2;
    s->acquired = -1;
    s->head = (s->head + 1) % (uint32_t)s->slots.size();
    s->in_flight += 1;
    return FMD_OK;
}

int fmd_sink_poll(fmd_sink* s)
{
    if (!s) return FMD_ERR_INVALID_ARG;
    int n = 0;
    for (;;) {
        const int r = complete_oldest(s, false);
        if (r < 0) return r;
        if (r == 0) break;
        ++n;
    }
    if (s->poisoned != FMD_OK && s->in_flight == 0) { fmd_internal_set_err(s->poison_msg); return s->poisoned; }
    return n;
}

int fmd_sink_drain(fmd_sink* s)
{
    if (!s) return FMD_ERR_INVALID_ARG;
    while (s->in_flight) {
        const int r = complete_oldest(s, true);
        if (r < 0) return r;
    }
    if (s->poisoned != FMD_OK) { fmd_internal_set_err(s->poison_msg); return s->poisoned; }
    return FMD_OK;
}

int fmd_sink_info(const fmd_sink* s, size_t* out_cap, uint32_t* n_devices, uint32_t* in_flight)
{
    if (!s) return FMD_ERR_INVALID_ARG;
    if (out_cap) *out_cap = s->out_cap;
    if (n_devices) *n_devices = (uint32_t)s->parts.size();
    if (in_flight) *in_flight = s->in_flight;
    return FMD_OK;
}

int fmd_sink_f64_stats(const fmd_sink* s, uint64_t* guarded, uint64_t* patched)
{
    if (!s) return FMD_ERR_INVALID_ARG;
    uint64_t g = 0, p = 0;
    for (const DevPart& part : s->parts) {                   // the parts' Demod banks hold the counters (fmd_internal_resolve_exc)
        uint64_t gi = 0, pi = 0;
        const int rc = fmd_demod_f64_stats(part.demod, &gi, &pi);
        if (rc != FMD_OK) return rc;
        g += gi; p += pi;
    }
    if (guarded) *guarded = g;
    if (patched) *patched = p;
    return FMD_OK;
}

}  // extern "C"
