// fmd_internal.h -- what the other translation units of libfmd_hip.so may ask of a fmd_demod handle beyond the C
// ABI (fmd_api.cpp owns the struct).  Not part of the boundary.
#pragma once

#include "../../include/fmd.h"
#include "fmd_kernels.h"

// Launches enqueued after this call report device assertions / guarded f64 samples into `buf` (device memory, zeroed
// by the caller) instead of the handle's own buffer; nullptr switches back.  Lets a pipelined caller keep one report
// buffer per in-flight launch, so that settling one launch never races with the kernels behind it.
void fmd_internal_set_report_buffer(fmd_demod* d, FmdExcBuf* buf);

struct FmdHandleView {
    int32_t R;                 // rate_out / rate_resample (simple_fm.rs:421)
    uint32_t seq;              // launches enqueued so far
    FmdChanState* state_cur;   // device state the NEXT launch will read
    uint64_t* guarded;         // f64 statistics of the handle
    uint64_t* patched;
    int device;
};
FmdHandleView fmd_internal_view(fmd_demod* d);

// Shared settle step (fmd_api.cpp): see fmd_demod_check in include/fmd.h.
int fmd_internal_resolve_exc(FmdExcBuf* d_exc, int32_t R, uint32_t host_seq, uint32_t state_seq, FmdChanState* d_state_cur,
                             int16_t* host_out, size_t host_cap, uint64_t* guarded, uint64_t* patched);

