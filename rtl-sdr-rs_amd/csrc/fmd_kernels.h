// fmd_kernels.h -- launch descriptors shared by the kernels (fmd_tile_body.h, fmd_generic_kernel.hip) and fmd_api.cpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fmd_index.h"

#define FMD_BLOCK_THREADS 256
#define FMD_MAX_CLASSES 16       /* phase classes one launch can carry (32-byte plans in the kernel arguments) */
/* register-streaming kernel: wave-rounds of 127 decimated samples per wave and tile (straight-line code).  Round 6 (second lever on
 * this kernel, profiles/r06_experiments.md 3): 16 instead of 12 rounds -- fewer ramps, barriers and resampler tails per byte -- is 1.0 ... 2.1 %
 * faster at downsample 2 and -0.8 ... +3.4 % at downsample 4 (where the planner's larger tile is not always the better one): 16 at
 * downsample 2 only. */
// Rounds per wave of the register-streaming kernels (straight-line code; the host sizes the tiles accordingly).  Downsample 4: 8 --
// ONE-SHOT waves, every load of a wave's rounds issued up front (round 6: -2 ... -5 % against 12 rounds with 8 loads in flight and
// refills, profiles/r06_experiments.md 12); downsample 2: 16 with 8 in flight (all 16 up front: +7 %; 8 one-shot: +12 %).
#define FMD_STREAM_MAX_ROUNDS(D) ((D) == 2u ? 16u : 8u)
#define FMD_TILE_MAX_LOADS 8      /* 16-byte chunks per thread the tile kernel can stage */

// Device-side error bits (FmdLaunch::err), all "cannot happen" conditions.
#define FMD_DEVERR_LP_CAP  1u
#define FMD_DEVERR_RAW_CAP 2u
#define FMD_DEVERR_EXC_CAP 4u     /* more guarded f64 samples than the exception buffer holds */

// ---- guarded f64 sample (Demod::polar_discriminant, simple_fm.rs:370-374) ------------------------------------
// The one f64 sample of every reference call is `(atan2(im, re) / PI * 16384.0) as i32`: the reference takes
// atan2 from the system libm, the kernel from ocml.  Two faithful atan2 implementations (a few ulp each) can
// only truncate differently when angle / PI * 16384 lies within ~2^-35 of an integer.  The kernel therefore
//   * decides the 8 axis / diagonal directions (and 0, 0) with integers (both libraries are exact there),
//   * and for every other sample whose value lies within f64_guard (default 2^-20) of an integer appends a
//     record here; the host re-evaluates those few with ITS libm -- the function the reference calls -- and
//     patches the audio sample (or the carried partial sum) that contains it.  Outside the guard band the two
//     results are provably equal, so the s16 output never depends on ocml's last bits.
#define FMD_EXC_CAP 1024u
struct FmdF64Exc {
    uint32_t channel;
    int32_t  k;          // audio sample of this launch that contains the f64 sample; -1: the trailing partial group (now_lpr)
    int32_t  cr, ci;     // the product a * conj(b) the sample is the angle of
    int32_t  d_gpu;      // what the kernel used
    int32_t  sum;        // k >= 0: the group's sum including d_gpu (audio = sum / R)
    uint32_t seq;        // launch sequence number of the handle
    uint32_t pad;
    uint64_t out_elem;   // device address of out[channel][k]
    uint64_t pad2;
};
struct FmdExcBuf {
    uint32_t err;        // FMD_DEVERR_* bits
    uint32_t count;      // records appended (may exceed FMD_EXC_CAP: then FMD_DEVERR_EXC_CAP is set too)
    uint32_t guarded_total, pad;
    FmdF64Exc rec[FMD_EXC_CAP];
};

// Fast tile geometry, closed form (calls of more than FMD_FAST_ROWS tiles per channel: long block_len launches; everything
// else takes the table below).  When a bank is one phase class and the tile length is a multiple of the reduced resample
// rate (kt * fr % sr == 0: true for the reference's rates and for the 2.4 Msps configuration), tile t of every
// channel is tile 0 shifted by t * Qt decimated samples, so where a tile's bytes are is a multiply-add of host
// constants.  They sit in the FIRST 64 bytes of the kernel arguments: a fresh block fetches them with one scalar
// load and issues its LDS-DMAs a few dozen scalar instructions later (the general prologue walks through a dozen
// dependent scalar loads first; a block holds its LDS the whole time).
struct FmdFastGeo {
    uint64_t iq;              // device address of the input
    uint64_t iq_end;          // iq + n_channels * chan_stride
    uint64_t chan_stride;     // bytes
    uint32_t n_channels;
    uint32_t per;             // ceil(n_channels / 8): gridDim.z of the XCD-aware grid (8, tiles, per)
    uint32_t nt;              // tiles per channel
    uint32_t step2;           // 2 * D * Qt: input bytes from one tile to the next
    int32_t  lo_off2;         // first byte of tile t = max(0, t * step2 + lo_off2)
    int32_t  hi_off2;         // end byte of tile t   = t * step2 + hi_off2 (the last tile ends at ns2)
    uint32_t ns2;             // bytes per channel-call
    uint32_t Qt;              // decimated samples per tile step
    int32_t  jA_off, jB_off;  // jA = max(0, t * Qt + jA_off); jB = t * Qt + jB_off (last tile: M - 1)
};
static_assert(sizeof(FmdFastGeo) == 64, "one s_load_dwordx16");

// The same per tile as a table -- what every launch of at most FMD_FAST_ROWS tiles per channel-call runs (a read_sync buffer
// always is), whatever the rates.  Round 5: the row carries EVERYTHING a block derives from its tile index -- staged range,
// LDS offsets, sample counts, resampler start, flags -- so that a block does one 64-byte scalar load and no index arithmetic
// (the scalar unit is the kernels' co-bottleneck: profiles/r04_experiments.md 28; the round-4 row held six numbers and the
// block rebuilt the rest with ~60 scalar instructions per wave).
struct FmdTileRow {
    uint32_t lo2a;            // first staged byte of the channel-call (multiple of 16)
    uint32_t nchunks;         // 16-byte chunks staged
    int32_t  wofs;            // LDS dword index of the call's dword 0: -(lo2a / 4)
    int32_t  jfirst;          // jA - 1: decimated samples jfirst .. jB are formed (-1: demod_pre)
    uint32_t cnt;             // jB - jfirst + 1
    uint32_t eq, er;          // (k0 + 1) fr - i0r - 1 = eq sr + er
    uint32_t k0, nk;          // audio samples [k0, k0 + nk) of the call
    int32_t  jA, jB;          // decimated samples the tile owns
    uint32_t flags;           // FMD_ROW_*
    int32_t  wbase;           // whole-dword windows: LDS dword index of the tile's window 0 = wofs - p0 / 2 + (D / 2) jfirst
    int32_t  s00;             // any window: call sample where the tile's window 0 starts = D jfirst - p0
    uint32_t par;             // rotation parity of window 0: (D / 2 odd ? jfirst : 0) ^ (p0 / 2), bit 0
    uint32_t pad;
};
static_assert(sizeof(FmdTileRow) == 64, "one s_load_dwordx16");
#define FMD_ROW_LAST  1u      /* the channel-call's last tile: writes the next state */
#define FMD_ROW_STATE 2u      /* the tile reads the channel's state (call start, first audio sample, last tile) */
#define FMD_ROW_BLOCKS 4u     /* the launch carries several reference calls (fmd_demod_set_block_len): every row of such a launch */
#define FMD_FAST_ROWS 32

// First 64 bytes of the kernel arguments in the table form (FmdLaunch::fast == 2): what a fresh block needs beside its row.
struct FmdRowGeo {
    uint64_t iq;              // device address of the input
    uint64_t out;             // device address of the output
    uint64_t chan_stride;     // input bytes per channel
    uint32_t n_channels;      // (same offsets as FmdFastGeo::n_channels / per)
    uint32_t per;
    uint32_t out_stride;      // output samples per channel (the host admits the form only while n_channels * out_stride * 2 < 2^32)
    uint32_t raw_cap;         // LDS bytes of the tile image: the discriminator samples sit behind it
    uint32_t p0;              // boxcar phase of the launch's one phase class
    uint32_t nt;              // tiles per channel-call
    uint64_t st_in;           // channel states read by this launch
    uint32_t pad[2];
};
static_assert(sizeof(FmdRowGeo) == 64, "one s_load_dwordx16");
// the kernels' early exit reads n_channels / per through whichever member of the union is active (fmd_tile_body.h)
static_assert(offsetof(FmdRowGeo, n_channels) == offsetof(FmdFastGeo, n_channels) && offsetof(FmdRowGeo, per) == offsetof(FmdFastGeo, per),
              "FmdRowGeo and FmdFastGeo keep n_channels / per at the same offsets");

struct FmdLaunch {
    union {                   // (must stay the first member)
        FmdFastGeo fg;        // fast == 1
        FmdRowGeo rg;         // fast == 2
    };
    uint32_t fast;            // 0: general prologue; 1: closed-form geometry (fg alone); 2: rg + rows[tile]
    uint32_t pad0[15];        // rows start on a 64-byte line of the kernel arguments
    FmdTileRow rows[FMD_FAST_ROWS];
    const uint8_t* iq;        // [n_channels][chan_stride] interleaved u8 IQ, 16-byte aligned base
    uint64_t chan_stride;     // bytes per channel (= nbytes of the call)
    uint64_t total_bytes;     // n_channels * chan_stride
    FmdRates r;
    uint32_t ns;              // complex samples per channel this call
    uint32_t xcd_swizzle;     // tile kernel: 0 plain; 1 / 2 XCD-aware block -> (channel, tile) mapping by index arithmetic (FMD_XCD);
                              // 3 (what runs by default, set by fmd_launch_tile): the same mapping through the grid shape (8, tiles, ceil(C/8))
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
    uint32_t* err;            // device error word (= &exc->err)
    FmdExcBuf* exc;           // guarded f64 samples of this launch (the handle keeps one buffer per launch parity)
    // Round 6, the pipelined completion point (fmd_demod_check_prev): a launch starts when its predecessor has COMPLETED (same
    // stream, or ordered by the library), so the first tile of channel 0 posts "launch seq - 1 is done" together with the head of
    // the predecessor's report buffer into host-mapped memory -- the host learns that buffer n - 1 is final while buffer n runs,
    // without an event or a copy between the two kernels (an event behind every launch costs 2 - 3 % per launch, round 5).
    const FmdExcBuf* exc_prev; // the predecessor's report buffer, or nullptr
    uint32_t* mbox;           // host-mapped, 8-byte aligned: low word = seq of the launch known complete, high word = 1 | 2 if its report buffer holds an error / records; or nullptr
    uint32_t* hflag;          // host-mapped word of THIS launch's ring slot, or nullptr: set (a system-scope store of 1, on the rare paths only) whenever the launch writes a
                              // record or an error bit into `exc` -- fmd_demod_check reads it after its stream synchronisation instead of copying the buffer's head
    double    f64_guard;      // half-width of the guard band around integers (see above)
    uint32_t  seq;            // launch sequence number (FmdF64Exc::seq)
    int32_t   f64_skew;       // -DFMD_EXPERIMENT builds only: added to the kernel's value of guarded samples (patch-path test)
    uint32_t dbg;             // ablation bits, honoured only by -DFMD_EXPERIMENT builds (tuning; never shipped)
    uint32_t stream;          // 1: register-streaming kernel (fmd_demod_stream_kernel): no LDS staging, raw_cap unused
    // ---- tile kernel only (phase-class plans; see fmd_index.h) ----
    uint32_t Qt;              // decimated samples per full tile = kt * fr / sr (== tl.Qt)
    FmdTiling tl;             // tiling constants of fmd_tile_fast
    uint32_t fa, fb;          // fr = fa * sr + fb
    float    inv_sr, inv_R;
    uint32_t sr_shift;        // log2(sr) when sr is a power of two, else 32
    FmdMagic magic_R;         // fmd_sdiv_magic(sum, magic_R) == sum / R for |sum| < 2^24
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

size_t fmd_generic_lds_bytes(const FmdLaunch& L);
size_t fmd_tile_lds_bytes(const FmdLaunch& L);
// Which kernel a launch ran (fmd_demod_last_kernel): family, instantiation, prologue form, tiling.
#define FMD_KERNEL_NONE    0
#define FMD_KERNEL_TILE    1     /* fmd_tk::fmd_demod_tile_kernel<dh, fast>: LDS-DMA staging */
#define FMD_KERNEL_STREAM  2     /* fmd_tk::fmd_demod_stream_kernel<dh, fast>: register streaming */
#define FMD_KERNEL_GENERIC 3     /* fmd_demod_generic_kernel<wide> (dh = 1: wide) */
struct FmdKernelId {
    uint8_t  family = FMD_KERNEL_NONE;
    uint8_t  fast = 0;        // 0 general prologue, 1 closed form, 2 per-tile table
    int16_t  dh = 0;          // template argument: half the (even) downsample, minus the odd one, 0 = catch-all
    uint32_t kt = 0;          // audio samples per tile
    uint32_t lds = 0;         // dynamic LDS bytes per block
};
hipError_t fmd_launch_generic(const FmdLaunch& L, hipStream_t stream, FmdKernelId* used);
hipError_t fmd_launch_tile(const FmdLaunch& L, hipStream_t stream, FmdKernelId* used);
hipError_t fmd_launch_synth(const FmdSynthLaunch& S, hipStream_t stream);

