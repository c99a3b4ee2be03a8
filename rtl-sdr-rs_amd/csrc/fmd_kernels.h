// fmd_kernels.h -- launch descriptors shared by fmd_kernels.hip and fmd_api.cpp.
#ifndef FMD_KERNELS_H
#define FMD_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fmd_index.h"

#define FMD_BLOCK_THREADS 256
#ifndef FMD_MAX_CLASSES
#define FMD_MAX_CLASSES 16       /* phase classes one launch can carry (32-byte plans in the kernel arguments) */
#endif
#define FMD_TILE_MAX_LOADS 8      /* 16-byte chunks per thread the tile kernel can stage */

// Device-side error bits (FmdLaunch::err), all "cannot happen" conditions.
#define FMD_DEVERR_LP_CAP  1u
#define FMD_DEVERR_RAW_CAP 2u

struct FmdLaunch {
    const uint8_t* iq;        // [n_channels][chan_stride] interleaved u8 IQ, 16-byte aligned base
    uint64_t chan_stride;     // bytes per channel (= nbytes of the call)
    uint64_t total_bytes;     // n_channels * chan_stride
    FmdRates r;
    uint32_t ns;              // complex samples per channel this call
    uint32_t xcd_swizzle;     // tile kernel: 0 plain, 1 / 2 XCD-aware block -> (channel, tile) mapping (FMD_XCD, default 2)
    uint32_t block_ns;        // > 0: the call is ns / block_ns consecutive reference calls of block_ns samples each
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
    uint32_t dbg;             // ablation bits, honoured only by -DFMD_EXPERIMENT builds (tuning; never shipped)
    uint32_t persist_blocks;  // > 0: persistent kernel with this many blocks; 0: one block per tile
    uint32_t block_threads;   // one-block-per-tile kernel: 64, 128 or 256 (default) threads
    uint32_t rounds_per_wave; // streaming kernel: consecutive rounds (tiles of kt audio samples) one wave walks
    uint32_t group_rounds;    // streaming kernel: rounds whose audio samples are produced together (<= 64 / kt)
    // ---- tile kernel only (phase-class plans; see fmd_index.h) ----
    uint32_t Qt;              // decimated samples per full tile = kt * fr / sr (== tl.Qt)
    FmdTiling tl;             // tiling constants of fmd_tile_fast
    uint32_t fa, fb;          // fr = fa * sr + fb
    float    inv_sr, inv_R;
    const uint8_t* chan_class;// [n_channels] class id, or nullptr when every channel is class 0
    FmdClassPlan cls[FMD_MAX_CLASSES];
};

struct FmdSynthLaunch {
    uint8_t* iq;
    uint64_t chan_stride;     // bytes per channel
    uint32_t n_channels;
    uint64_t sample_offset;   // index of the first complex sample of this buffer in the stream
    uint64_t seed;
    uint32_t amplitude, noise, dev_q32, mod_period;
};

// Can the division-free tile kernel run this configuration?  (Otherwise the generic kernel does.)
bool fmd_tile_kernel_supports(const FmdRates& r, uint32_t raw_cap);

// The persistent kernel prefetches a whole tile in registers: FMD_PERSIST_LOADS x 16 B per lane.
#define FMD_PERSIST_LOADS 5
inline bool fmd_persist_supports(uint32_t raw_cap) { return raw_cap <= 16u * FMD_PERSIST_LOADS * FMD_BLOCK_THREADS; }
int fmd_persist_blocks_per_cu(const FmdLaunch& L);   // occupancy of the persistent kernel for this config

size_t fmd_generic_lds_bytes(const FmdLaunch& L);
size_t fmd_tile_lds_bytes(const FmdLaunch& L);
hipError_t fmd_launch_generic(const FmdLaunch& L, hipStream_t stream);
hipError_t fmd_launch_tile(const FmdLaunch& L, hipStream_t stream);
// Register-streaming kernel (fmd_stream_kernel.hip): no LDS staging, a wave walks consecutive rounds.
bool fmd_stream_kernel_supports(const FmdRates& r);       // for r.kt = the round size
uint32_t fmd_stream_round_kt(const FmdRates& r);          // largest supported round size, 0 = none
uint32_t fmd_stream_group_rounds(const FmdRates& r);      // rounds per audio group for r.kt
hipError_t fmd_launch_stream(const FmdLaunch& L, hipStream_t stream);
hipError_t fmd_launch_synth(const FmdSynthLaunch& S, hipStream_t stream);

#endif
