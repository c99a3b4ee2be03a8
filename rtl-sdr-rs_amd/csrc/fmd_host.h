// fmd_host.h -- host-side helpers shared by the translation units of libfmd_hip.so (fmd_api.cpp, fmd_fir.hip,
// fmd_firdemod.hip, fmd_sink.cpp).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

// Tuning and test knobs (kernel variants, tilings, the width of the f64 guard band, ablation bits) exist ONLY in the
// -DFMD_EXPERIMENT build (libfmd_hip_exp.so, loaded through FMD_LIB by the tests and the A/B tools).  The shipped
// library reads no environment variable at all: fmd_knob() is a constant there, so every knob folds to its default
// and no stray variable can swap a kernel or narrow the guard band the bit-exactness argument rests on.
#ifdef FMD_EXPERIMENT
inline const char* fmd_knob(const char* name) { const char* s = getenv(name); return (s && *s) ? s : nullptr; }
#else
inline const char* fmd_knob(const char*) { return nullptr; }
#endif
inline uint32_t fmd_knob_u32(const char* name, uint32_t dflt)
{
    const char* s = fmd_knob(name);
    return s ? (uint32_t)strtoul(s, nullptr, 10) : dflt;
}
inline int32_t fmd_knob_i32(const char* name, int32_t dflt)
{
    const char* s = fmd_knob(name);
    return s ? (int32_t)strtol(s, nullptr, 10) : dflt;
}

// Error text for fmd_last_error() (thread-local, lives in fmd_api.cpp).
void fmd_internal_set_err(const char* msg);

// Every entry point works on its handle's device and puts the caller's current device back on every exit path:
// a process that shares the HIP runtime with other code (torch, several handles on several GPUs) must not find
// its later allocations and launches silently redirected.
class FmdDeviceGuard {
public:
    explicit FmdDeviceGuard(int device) : prev_(-1), err_(hipSuccess)
    {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (prev_ != device) err_ = hipSetDevice(device); else prev_ = -1;   // nothing to restore
    }
    ~FmdDeviceGuard() { if (prev_ >= 0) (void)hipSetDevice(prev_); }
    hipError_t error() const { return err_; }
    FmdDeviceGuard(const FmdDeviceGuard&) = delete;
    FmdDeviceGuard& operator=(const FmdDeviceGuard&) = delete;
private:
    int prev_;
    hipError_t err_;
};

// Launches of one handle are ordered by the stream they go to.  A handle's double-buffered state (st_in / st_out,
// FIR history) makes launch n+1 read what launch n wrote, so when consecutive calls use DIFFERENT streams the new
// stream first waits for everything the handle enqueued on the previous one.  Free in the common case (same stream).
// LIFETIME RULE (include/fmd.h): the stream of a handle's most recent `_device` call must stay alive until the handle's next
// `_device` call or completion point (`*_check`, `get_state`, a host entry) has returned -- those are the only places the
// remembered handle is used again.  Round 5 measured the alternative that needs no such rule -- an event recorded behind
// EVERY launch, later waits on the event only: +2.4 ... +3.2 % per launch at the headline and the reference's rates (the
// record is a barrier packet between back-to-back kernels), +8 % on launch + check per buffer (profiles/r05_experiments.md).
struct FmdStreamOrder {
    hipStream_t last = nullptr;
    bool have_last = false;
    hipEvent_t ev = nullptr;
    // EVENT MODE (round 6, opt-in per handle: fmd_demod_set_event_ordering): an event is recorded behind EVERY launch and every later
    // wait -- the next launch's stream, the completion points -- goes to the EVENT; the caller's stream handle is never used again
    // after the enqueue call has returned, so the lifetime rule above does not apply (a stream from a pool that destroys it at will
    // is fine).  Costs the record: +2 - 3 % per launch at the headline rates (profiles/r05_experiments.md 1), which is why it is not
    // the default.
    bool event_mode = false;

    hipError_t ensure_event()
    {
        return ev ? hipSuccess : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    }
    // Call before enqueueing on `stream`.
    hipError_t before(hipStream_t stream)
    {
        if (!have_last) return hipSuccess;
        if (event_mode) return hipStreamWaitEvent(stream, ev, 0);      // (recorded by after(): always there once have_last is set)
        if (stream == last) return hipSuccess;
        hipError_t e = ensure_event();
        if (e != hipSuccess) return e;
        e = hipEventRecord(ev, last);                      // everything submitted to the old stream so far
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipDeviceSynchronize(); }   // be safe
        return e;
    }
    hipError_t after(hipStream_t stream)
    {
        last = stream; have_last = true;
        if (!event_mode) return hipSuccess;
        hipError_t e = ensure_event();
        if (e == hipSuccess) e = hipEventRecord(ev, stream);
        return e;
    }
    // Wait for the handle's most recent launch (completion points).
    hipError_t wait_last()
    {
        if (!have_last) return hipSuccess;
        return event_mode ? hipEventSynchronize(ev) : hipStreamSynchronize(last);
    }
    // hipSuccess: the most recent launch has completed; hipErrorNotReady: still running.
    hipError_t query_last()
    {
        if (!have_last) return hipSuccess;
        return event_mode ? hipEventQuery(ev) : hipStreamQuery(last);
    }
    void reset() { have_last = false; }
    void destroy() { if (ev) (void)hipEventDestroy(ev); ev = nullptr; have_last = false; }
};
