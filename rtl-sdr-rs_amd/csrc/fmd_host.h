// fmd_host.h -- host-side helpers shared by the translation units of libfmd_hip.so (fmd_api.cpp, fmd_fir.hip,
// fmd_firdemod.hip, fmd_sink.cpp).  Not part of the C ABI.
#ifndef FMD_HOST_H
#define FMD_HOST_H

#include <hip/hip_runtime.h>

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
struct FmdStreamOrder {
    hipStream_t last = nullptr;
    bool have_last = false;
    hipEvent_t ev = nullptr;

    // Call before enqueueing on `stream`.
    hipError_t before(hipStream_t stream)
    {
        if (!have_last || stream == last) return hipSuccess;
        if (!ev) { hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (e != hipSuccess) return e; }
        hipError_t e = hipEventRecord(ev, last);           // everything submitted to the old stream so far
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipDeviceSynchronize(); }   // old stream gone: be safe
        return e;
    }
    void after(hipStream_t stream) { last = stream; have_last = true; }
    void reset() { have_last = false; }
    void destroy() { if (ev) (void)hipEventDestroy(ev); ev = nullptr; have_last = false; }
};

#endif  // FMD_HOST_H
