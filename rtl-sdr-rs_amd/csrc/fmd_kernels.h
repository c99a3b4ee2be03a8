// fmd_kernels.h -- launch descriptors shared by fmd_kernels.hip and fmd_api.cpp.
#ifndef FMD_KERNELS_H
#define FMD_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fmd_index.h"

#define FMD_BLOCK_THREADS 256

// Device-side error bits (FmdLaunch::err), all "cannot happen" conditions.
#define FMD_DEVERR_LP_CAP  1u
#define FMD_DEVERR_RAW_CAP 2u

struct FmdLaunch {
    const uint8_t* iq;        // [n_channels][chan_stride] interleaved u8 IQ, 16-byte aligned base
    uint64_t chan_stride;     // bytes per channel (= nbytes of the call)
    uint64_t total_bytes;     // n_channels * chan_stride
    FmdRates r;
    uint32_t ns;              // complex samples per channel this call
    uint32_t n_channels;
    uint32_t tiles;           // grid tiles per channel (>= every channel's own tile count)
    uint32_t lp_cap;          // LDS sizing: decimated samples per tile
    uint32_t raw_cap;         // LDS sizing: raw bytes per tile (multiple of 16)
    const FmdChanState* st_in;
    FmdChanState* st_out;
    int16_t* out;             // [n_channels][out_stride]
    uint64_t out_stride;      // samples
    uint32_t* out_len;        // [n_channels] or nullptr
    uint32_t* err;            // device error word
};

struct FmdSynthLaunch {
    uint8_t* iq;
    uint64_t chan_stride;     // bytes per channel
    uint32_t n_channels;
    uint64_t sample_offset;   // index of the first complex sample of this buffer in the stream
    uint64_t seed;
    uint32_t amplitude, noise, dev_q32, mod_period;
};

size_t fmd_demod_lds_bytes(const FmdLaunch& L);
hipError_t fmd_launch_demod(const FmdLaunch& L, hipStream_t stream);
hipError_t fmd_launch_synth(const FmdSynthLaunch& S, hipStream_t stream);

#endif
