/*
 * fmd.h -- C ABI of the MI355X-native FM demodulation path (libfmd_hip.so).
 *
 * Drop-in boundary for ONE path of ccostes/rtl-sdr-rs v0.3.1: the `Demod` chain of
 * examples/simple_fm.rs (u8 IQ -> rotate_90 -> centre -> boxcar decimate -> polar
 * discriminator -> fractional boxcar resampler -> s16).  Every entry point cites the
 * reference interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Threading: a handle is like the reference's `&mut Demod` -- one caller at a time.
 * Use one handle per host thread / per GPU.  Consecutive launches of one handle may go to different streams:
 * the library orders them (the new stream waits for the handle's previous launch); a stream handed to a
 * *_device entry point must stay alive until the handle's next call or fmd_demod_check.  Every entry point
 * leaves the caller's current HIP device as it found it.  All functions return FMD_OK (0) or a negative
 * fmd_status; nothing panics or aborts where the reference would.
 *
 * There is NO CPU fallback in this library: without a usable gfx950 device fmd_demod_new
 * fails with FMD_ERR_NO_DEVICE / FMD_ERR_HIP.
 */
#ifndef FMD_H
#define FMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMD_VERSION_MAJOR 0
#define FMD_VERSION_MINOR 3

/* DEFAULT_BUF_LENGTH, src/lib.rs:25 -- the buffer size RtlSdr::read_sync callers use. */
#define FMD_DEFAULT_BUF_LENGTH (16 * 16384)

typedef enum fmd_status {
    FMD_OK = 0,
    FMD_ERR_INVALID_ARG   = -1,  /* NULL pointer, zero channels, ...                                */
    FMD_ERR_BAD_LENGTH    = -2,  /* nbytes % 8 != 0 -- reference: index panic, simple_fm.rs:286     */
    FMD_ERR_TOO_SHORT     = -3,  /* < 2 decimated samples -- reference: assert!, simple_fm.rs:356   */
    FMD_ERR_BAD_RATES     = -4,  /* rate_out < rate_resample or 0 -- reference: div by zero, :421   */
    FMD_ERR_CAPACITY      = -5,  /* out_cap smaller than the number of samples produced             */
    FMD_ERR_UNSUPPORTED   = -6,  /* configuration outside this implementation's documented domain   */
    FMD_ERR_BAD_STATE     = -7,  /* fmd_demod_set_state with a state no Demod can reach             */
    FMD_ERR_NO_DEVICE     = -8,  /* no gfx950 device / bad device_id                                */
    FMD_ERR_HIP           = -9,  /* a HIP runtime call failed; see fmd_last_error()                 */
    FMD_ERR_NOMEM         = -10,
    FMD_ERR_IO            = -11  /* rtl_tcp source: connect / handshake / socket error, see fmd_last_error()  */
} fmd_status;

/* struct RadioConfig, simple_fm.rs:173-176 */
typedef struct fmd_radio_config {
    uint32_t capture_freq;
    uint32_t capture_rate;
} fmd_radio_config;

/* struct DemodConfig, simple_fm.rs:179-185 (same field order). */
typedef struct fmd_demod_config {
    uint32_t rate_in;        /* stored, unused in arithmetic (reference: :180, logged :139)       */
    uint32_t rate_out;       /* "fast" rate of the audio resampler, :412                          */
    uint32_t rate_resample;  /* "slow" rate of the audio resampler, :411                          */
    uint32_t downsample;     /* boxcar length / decimation, :343                                  */
    uint32_t output_scale;   /* computed, unused in arithmetic (reference: :184,197-200)          */
} fmd_demod_config;

/* The mutable fields of struct Demod, simple_fm.rs:232-239.  This is the resumable state. */
typedef struct fmd_demod_state {
    uint32_t prev_index;      /* boxcar phase, 0 .. downsample-1                                  */
    int32_t  now_lpr;         /* partial audio sum                                                */
    int32_t  prev_lpr_index;  /* resampler phase, 0 .. rate_out-1                                 */
    int32_t  lp_now_re, lp_now_im;       /* partial boxcar sum                                    */
    int32_t  demod_pre_re, demod_pre_im; /* last decimated sample of the previous call            */
} fmd_demod_state;

/* Placement of a Demod bank on the machine (no reference counterpart: the reference is
 * one Demod on one CPU thread, simple_fm.rs:137). */
typedef struct fmd_device_config {
    uint32_t n_channels;  /* independent IQ streams (one reference `Demod` each); >= 1           */
    int32_t  device_id;   /* HIP device ordinal; -1 = current device                             */
    uint32_t flags;       /* reserved, 0                                                          */
} fmd_device_config;

typedef struct fmd_demod fmd_demod;   /* opaque; owns device buffers + per-channel state */

/* ---- configuration ------------------------------------------------------------------- */

/* optimal_settings(freq, rate), simple_fm.rs:189-214.  rate_resample is the reference's
 * RATE_RESAMPLE const (:27, 32000).  FMD_ERR_BAD_RATES where the reference divides by zero. */
int fmd_optimal_settings(uint32_t freq, uint32_t rate, uint32_t rate_resample,
                         fmd_radio_config *radio, fmd_demod_config *demod);

/* ---- lifecycle ------------------------------------------------------------------------ */

/* Demod::new(config), simple_fm.rs:243-252, for a bank of n_channels independent streams,
 * all with zeroed state.  downsample 1 ... 512 (FMD_ERR_UNSUPPORTED beyond).  From 129 up the
 * reference's own i32 arithmetic wraps at full-scale input (reproduced bit for bit); where a
 * release build of the reference would PANIC -- a zero divisor inside fast_atan2 after such a
 * wrap, reachable from downsample 305 with particular full-scale inputs -- the audio sample
 * that contains the discriminator value is unspecified (no status is raised). */
int fmd_demod_new(const fmd_demod_config *config, const fmd_device_config *dev, fmd_demod **out);

/* Drop for Demod. NULL is a no-op. */
void fmd_demod_free(fmd_demod *d);

/* Zero every channel's state (== dropping and re-creating the Demods). */
int fmd_demod_reset(fmd_demod *d);

/* ---- the hot path --------------------------------------------------------------------- */

/* Demod::demodulate(&mut self, buf: Vec<u8>) -> Vec<i16>, simple_fm.rs:256-269, for a
 * single-channel handle and HOST buffers: `iq` is exactly what RtlSdr::read_sync
 * (src/lib.rs:153) filled -- interleaved offset-binary u8 I,Q,I,Q...  Writes *out_len
 * samples to out (capacity out_cap samples; see fmd_out_cap).  The input is not modified. */
int fmd_demod_demodulate(fmd_demod *d, const uint8_t *iq, size_t nbytes,
                         int16_t *out, size_t out_cap, size_t *out_len);

/* The same for every channel of the bank, HOST buffers:
 *   iq      [n_channels][nbytes]   channel-major, contiguous
 *   out     [n_channels][out_cap]
 *   out_len [n_channels]
 * Elements of a row beyond its out_len (up to out_cap) are unspecified after the call. */
int fmd_demod_demodulate_batch(fmd_demod *d, const uint8_t *iq, size_t nbytes,
                               int16_t *out, size_t out_cap, size_t *out_len);

/* The same with DEVICE-resident buffers (the measured path): d_iq / d_out / d_out_len are
 * device pointers on the handle's GPU with the layouts above (d_iq 16-byte aligned;
 * d_out_len uint32, may be NULL).  Enqueues on `stream` (a hipStream_t; NULL = the
 * default stream) and returns without synchronising; the per-channel counts are also
 * available on the host, without a sync, from fmd_demod_last_out_len.
 * Stream lifetime: the library orders a handle's consecutive launches itself, also across different
 * streams, and fmd_demod_check completes behind the most recent launch ON ITS STREAM -- so the `stream`
 * of a handle's most recent _device call must stay alive until the handle's next _device call or
 * completion point (fmd_demod_check, _get_state, _set_state, a host entry point) has returned.  The
 * same rule holds for fmd_fir_filter_device and fmd_firdemod_demodulate_device.
 * Fast path: nbytes % 16 == 0 (every read_sync buffer: DEFAULT_BUF_LENGTH = 262144) and a bank whose
 * channels share one phase (banks fed equal-length buffers always do) take the kernels' short
 * prologues.  nbytes % 16 == 8 is legal (the reference only demands % 8, :284) and bit-exact, but runs the
 * general prologue -- a few percent slower, and for downsample 2 / 4 the LDS kernel instead of the
 * register-streaming one; fmd_demod_last_kernel reports which kernel a launch ran. */
int fmd_demod_demodulate_device(fmd_demod *d, const void *d_iq, size_t nbytes,
                                void *d_out, size_t out_cap, void *d_out_len, void *stream);

/* Several reference calls per launch.  Demod::demodulate takes the f64 atan2 path for the first
 * decimated sample of EVERY call (simple_fm.rs:359), so the audio of a stream depends on where
 * the caller cut it into buffers.  After fmd_demod_set_block_len(d, block_bytes) every
 * demodulate_* call is treated as the concatenation of nbytes / block_bytes consecutive
 * reference calls of block_bytes each (nbytes must be a multiple): one launch then returns
 * exactly the concatenated audio -- and leaves exactly the state -- of feeding the reference
 * those blocks one by one, e.g. 256 x DEFAULT_BUF_LENGTH in one 64 MiB launch.  block_bytes
 * % 8 == 0 and >= 4 * downsample bytes (every block yields >= 2 decimated samples, :356);
 * 0 (the default) switches it off: one call = one reference call. */
int fmd_demod_set_block_len(fmd_demod *d, size_t block_bytes);

/* Completion + verification point for the DEVICE entry point: waits for everything this handle has enqueued,
 * returns FMD_ERR_HIP if a device-side sizing assertion fired (fmd_last_error() has the bits), and settles the
 * one f64 sample of every reference call (Demod::polar_discriminant, simple_fm.rs:359,370-374 -- `atan2` from
 * the system libm in the reference): the kernel decides the axis / diagonal directions with integers and reports
 * every other sample whose value lies within 2^-20 of an integer; those few are re-evaluated here with the HOST
 * libm (the function the reference calls) and, if the truncated value differs, the audio sample (in the device
 * output buffer the launch wrote, which must still be allocated) or the carried partial sum is patched.
 * Outside that band the two results are provably equal.  The HOST entry points and fmd_demod_get_state do this
 * themselves before returning; after fmd_demod_demodulate_device call it before reading the output.  Only the output
 * buffers of the THREE most recent launches are ever written by a patch (they must still be allocated; round 5: the most recent
 * one only); a sample of an OLDER launch that turns out to need one (probability ~2^-36 per reference call) makes this function
 * return FMD_ERR_HIP instead of touching memory the caller may have reused: for the strict bit-exactness guarantee call it --
 * or fmd_demod_check_behind, which does not serialise -- for every launch. */
int fmd_demod_check(fmd_demod *d);

/* The same completion point ONE or TWO LAUNCHES BACK (round 6): with launches 1 ... n enqueued through fmd_demod_demodulate_device,
 * fmd_demod_check_behind(d, back) waits until launch n - back has completed -- not for the newer ones -- and settles ITS f64 samples,
 * so that the reference's cadence
 *     loop { buf = read_sync(); audio = demod.demodulate(buf); output(audio); }        (simple_fm.rs:150-156)
 * runs as  enqueue(buf n); fmd_demod_check_behind(d, 2); output(audio n - 2);  with the GPU never idle between launches and the
 * strict bit-exactness guarantee intact (fmd_demod_check after every launch serialises host and device: ~8 % at the headline
 * configuration; this cadence: 1.00 - 1.01 x the bare launches, extra.check_pipelined of the bench line).  back = 2 keeps a whole
 * launch queued behind the running one, which absorbs a late host; back = 1 keeps none; back = 0 is fmd_demod_check.  How: launch
 * s + 1 cannot start before launch s has completed, so its first tile posts "s is done" together with whether s reported anything
 * into host-mapped memory; everything a launch leaves behind -- its report buffer, the state it wrote, its launch record -- lives in
 * a ring of three.  A launch that DID report guarded samples is settled while the newer ones run as well (its records are copied and
 * its audio samples patched on the handle's own stream); only a correction of the sum it carried into the next launch, or a device
 * assertion, waits for everything.
 * Contract: until a launch has been settled by this function or by fmd_demod_check, (a) its OUTPUT buffer stays allocated and
 * unread -- a patch goes into the buffer of the launch that produced the sample, for the last THREE launches -- and (b) its INPUT
 * buffer stays unmodified: in the one case where a launch's corrected sample lies in the partial sum it carried into the next launch
 * (probability ~2^-36 per reference call), the launches behind it are run again on the corrected state, inside this call.  Launches
 * are settled IN ORDER (an older unsettled launch turns the call into fmd_demod_check).  Returns FMD_OK at once when fewer than
 * back + 1 launches are outstanding; falls back to fmd_demod_check where no post is coming (generic kernel).  Call fmd_demod_check
 * after the LAST launch of a run.  (fmd_firdemod / the pipelined sink have their own completion points.) */
int fmd_demod_check_behind(fmd_demod *d, uint32_t back);
/* fmd_demod_check_behind(d, 1). */
int fmd_demod_check_prev(fmd_demod *d);

/* EVENT ORDERING (opt-in, round 6): with on != 0 the handle records an event behind every launch and every later wait -- the next
 * launch on another stream, fmd_demod_check / _check_prev / _get_state -- goes to that EVENT: a `stream` handed to
 * fmd_demod_demodulate_device is never used again once that call has returned, so the stream lifetime rule above does not apply (for
 * callers whose streams come from a pool that may destroy them).  Costs the record: +2 - 3 % per launch at the headline rates, which
 * is why it is not the default.  Synchronises the device; call it between launches. */
int fmd_demod_set_event_ordering(fmd_demod *d, int on);

/* Diagnostics of the above: f64 samples that fell into the guard band / whose value had to be patched. */
int fmd_demod_f64_stats(const fmd_demod *d, uint64_t *guarded, uint64_t *patched);

/* Per-channel sample counts of the most recent demodulate_* call (host bookkeeping). */
int fmd_demod_last_out_len(const fmd_demod *d, size_t *out_len /* [n_channels] */);

/* Upper bound of samples one call can produce per channel for nbytes of input. */
size_t fmd_out_cap(const fmd_demod_config *config, size_t nbytes);

/* Page-locked host buffers for the HOST entry points above.  The reference reads into a pageable
 * heap buffer and sends a copy of it per block (simple_fm.rs:114,127 `buf.to_vec()`); a binding that takes
 * its read buffers from here instead lets the copy engine DMA straight out of them (no staging
 * copy through the runtime's bounce buffers).  Plain pageable pointers keep working. */
int fmd_host_alloc(size_t nbytes, void **ptr);
int fmd_host_free(void *ptr);

/* ---- state (checkpoint / resume; simple_fm.rs:232-239) -------------------------------- */
int fmd_demod_get_state(fmd_demod *d, uint32_t channel, fmd_demod_state *state);
int fmd_demod_set_state(fmd_demod *d, uint32_t channel, const fmd_demod_state *state);

/* ---- synthetic IQ source (stands in for the absent capture.bin) -------- */

typedef struct fmd_synth_params {
    uint64_t seed;          /* channel c uses seed + c                                            */
    uint32_t amplitude;     /* carrier amplitude in LSB, <= 120                                   */
    uint32_t noise;         /* uniform noise in [-noise, +noise] LSB                              */
    uint32_t dev_q32;       /* peak frequency deviation, cycles/sample in Q32                     */
    uint32_t mod_period;    /* triangle modulation period in samples (even, >= 2)                 */
} fmd_synth_params;

/* Fill d_iq [n_channels][nbytes] (device pointer) with a deterministic integer-only FM
 * signal sitting at -Fs/4 (what offset tuning, simple_fm.rs:194-195, delivers).
 * rtl-sdr-rs_amd/synth.py generates identical bytes with numpy. */
int fmd_synth_fill_device(int device_id, void *d_iq, uint32_t n_channels, size_t nbytes,
                          uint64_t sample_offset, const fmd_synth_params *p, void *stream);

/* ---- generalised tapped decimating FIR (SURVEY 8a row G', BASELINE config 4) ----------- */
/* NOT a reference interface: the reference's only tapped FIR lives in the RTL2832U chip
 * (src/rtlsdr.rs:525-558).  Definition (also oracle/fm_oracle.h):
 *     y[m] = sum_{t < n_taps} taps[t] * x[decim * m + t]
 * over the stream x[n] of rotated (rotate_90, simple_fm.rs:276-299) and centred (`- 127`, :258)
 * complex samples of one channel, counted from the first sample fed after fmd_fir_new /
 * fmd_fir_reset; y[m] is produced by the call in which x[decim*m + n_taps - 1] arrives.
 * With taps = 1...1 and n_taps == decim == downsample it is Demod::low_pass_complex (:337-352).
 * decim must be even (whole-dword windows), 1 <= n_taps <= 1024, |taps| <= 2047. */
typedef struct fmd_fir fmd_fir;
int fmd_fir_new(const int16_t *taps, uint32_t n_taps, uint32_t decim, const fmd_device_config *dev,
                fmd_fir **out);
void fmd_fir_free(fmd_fir *f);
int fmd_fir_reset(fmd_fir *f);
/* Which form the handle runs (introspection for tests and bench lines): 1 = matrix cores with one i8 digit per tap (every
 * |tap| <= 127: eight outputs per operand column), 2 = two digits (|tap| <= 2047: four), 0 = the vector-pipe kernel
 * (decim > 64 or a filter too long for the matrix-core form). */
int fmd_fir_tap_digits(const fmd_fir *f);
/* Name of the kernel this handle launches, as `rocprofv3 --kernel-trace` prints it (see fmd_demod_last_kernel). */
int fmd_fir_kernel_name(const fmd_fir *f, char *name, size_t cap);
/* Complex outputs one call of nbytes can produce per channel (upper bound). */
size_t fmd_fir_out_cap(uint32_t n_taps, uint32_t decim, size_t nbytes);
/* HOST buffers: iq [n_channels][nbytes]; out [n_channels][out_cap][2] int32 (re, im);
 * out_len [n_channels] complex samples written. */
int fmd_fir_filter_batch(fmd_fir *f, const uint8_t *iq, size_t nbytes, int32_t *out, size_t out_cap,
                         size_t *out_len);
/* DEVICE buffers with the same layouts, enqueued on `stream` without synchronising; the
 * per-channel count (identical for all channels) is returned in *out_len_each. */
int fmd_fir_filter_device(fmd_fir *f, const void *d_iq, size_t nbytes, void *d_out, size_t out_cap,
                          size_t *out_len_each, void *stream);

/* ---- tapped FIR -> discriminator -> resampler in one kernel (BASELINE north_star: "FIR + demod + resample fused") -- */
/* NOT a reference interface either.  Definition: Demod::demodulate (simple_fm.rs:256-269) with low_pass_complex
 * (:337-352) replaced by the tapped decimating FIR above, normalised by a right shift,
 *     lp[m] = floor( sum_{t < n_taps} taps[t] * x[decim * m + t] / 2^shift ),
 * followed by the reference's own fm_demod (:355-367, the f64 sample at the first filter output of every call) and
 * low_pass_real (:408-426, rate_out -> rate_resample).  With taps = 1...1, n_taps == decim == downsample, shift == 0
 * it returns exactly what fmd_demod_* (and the oracle of the reference chain) return -- tested bit for bit; that is
 * its anchor.  (An 8-bit filter -- every |tap| <= 127 -- takes a form of the matrix-core kernels with one i8 digit per tap and
 * eight outputs per operand column, here and in fmd_fir_*: same results, fewer matrix instructions.)
 * Domain: decim even and <= 64, 1 <= n_taps <= 1024, |taps| <= 2047, (128 * sum|taps|) >> shift <= 16384 (so that
 * |lp| stays in the discriminator's range, the boxcar's bound at downsample 128), shift <= 24.  A shift that brings
 * (128 * sum|taps|) >> shift down to 2048 -- the boxcar's range at downsample 16 -- selects the kernel's f32 form of
 * the discriminator (same results, ~8 % faster); the Python mirror's auto_shift() picks that one by default.
 * A call that yields fewer than 2 filter outputs returns FMD_ERR_TOO_SHORT (assert at :356) and changes nothing.
 * All channels of a bank advance together (equal-sized buffers), so there is no per-channel set_state; the bank as a
 * whole is saved and restored with fmd_firdemod_checkpoint / fmd_firdemod_resume. */
typedef struct fmd_firdemod fmd_firdemod;
int fmd_firdemod_new(const int16_t *taps, uint32_t n_taps, uint32_t decim, uint32_t shift, uint32_t rate_out,
                     uint32_t rate_resample, const fmd_device_config *dev, fmd_firdemod **out);
void fmd_firdemod_free(fmd_firdemod *f);
int fmd_firdemod_reset(fmd_firdemod *f);
/* Audio samples one call of nbytes can produce per channel (upper bound). */
size_t fmd_firdemod_out_cap(uint32_t decim, uint32_t rate_out, uint32_t rate_resample, size_t nbytes);
/* HOST buffers: iq [n_channels][nbytes]; out [n_channels][out_cap] s16; out_len [n_channels]. */
int fmd_firdemod_demodulate_batch(fmd_firdemod *f, const uint8_t *iq, size_t nbytes, int16_t *out, size_t out_cap,
                                  size_t *out_len);
/* DEVICE buffers, enqueued on `stream` without synchronising; the per-channel count (identical for all channels)
 * is returned in *out_len_each.  fmd_firdemod_check is its completion / verification point (see fmd_demod_check). */
int fmd_firdemod_demodulate_device(fmd_firdemod *f, const void *d_iq, size_t nbytes, void *d_out, size_t out_cap,
                                   size_t *out_len_each, void *stream);
int fmd_firdemod_check(fmd_firdemod *f);
/* demod_pre, now_lpr, prev_lpr_index of one channel (prev_index / lp_now are 0: the FIR owns the decimation). */
int fmd_firdemod_get_state(fmd_firdemod *f, uint32_t channel, fmd_demod_state *state);
/* Checkpoint / resume of the WHOLE bank (its channels advance together, so the unit is the bank, not a channel): the
 * sample position, the resampler phase, and per channel now_lpr, demod_pre and the filter's history (the last
 * n_taps - 1 samples).  A bank resumed from a checkpoint continues bit for bit as the one it was taken from would
 * have; the blob is host memory, little-endian, and names the filter (taps, decim, shift, rates, channel count) it
 * belongs to -- fmd_firdemod_resume on any other bank, or on a damaged blob, returns FMD_ERR_BAD_STATE and changes
 * nothing.  Both calls synchronise the device first. */
size_t fmd_firdemod_checkpoint_size(const fmd_firdemod *f);
int fmd_firdemod_checkpoint(fmd_firdemod *f, void *blob, size_t cap);
int fmd_firdemod_resume(fmd_firdemod *f, const void *blob, size_t size);
int fmd_firdemod_f64_stats(const fmd_firdemod *f, uint64_t *guarded, uint64_t *patched);
int fmd_firdemod_tiling(const fmd_firdemod *f, uint32_t *audio_per_tile, uint32_t *lds_bytes);
/* Name of the kernel this handle launches, as `rocprofv3 --kernel-trace` prints it (see fmd_demod_last_kernel). */
int fmd_firdemod_kernel_name(const fmd_firdemod *f, char *name, size_t cap);

/* ---- pipelined, multi-GPU sink for read_sync buffers ------------------------------------------------------- */
/* NEW SURFACE (the reference has no asynchronous reader).  It mirrors the hand-off the example
 * does have: receive() fills a buffer with RtlSdr::read_sync (src/lib.rs:153) and sends it down an mpsc channel,
 * process() demodulates it and calls output() (examples/simple_fm.rs:55-60,114-127,150-156).  Here the channel
 * is a ring of `depth` page-locked slots and the consumer is one Demod bank per GPU: channels are split into
 * contiguous ranges over `device_ids` (one Demod per stream, :137 -- no communication between devices); host ->
 * device copy of buffer n+1, the kernel of buffer n and the device -> host copy of buffer n-1 overlap.
 *   fmd_sink_acquire: the next slot to fill, [n_channels][nbytes] channel-major (blocks only when all `depth`
 *                     slots are in flight: then the oldest is completed first);
 *   fmd_sink_submit : enqueue it on every device and return without waiting;
 *   fmd_sink_release: give the acquired slot back unsubmitted (a short read_sync ends the run, simple_fm.rs:122-125);
 *   completion, in submission order, from inside acquire / poll / drain on the caller's thread:
 *       callback(user, seq, audio [n_channels][out_cap], out_len [n_channels], out_cap, status)
 *     -- `audio` is valid during the callback only; status FMD_OK or the first error of that buffer.
 * Results are exactly those of feeding the same buffers to fmd_demod_demodulate_batch one by one, with ONE stated
 * exception: a buffer so short that it yields no audio sample at all carries its f64 sample (simple_fm.rs:359) in the
 * partial sum handed to the next buffer; with depth > 1 the next launch may already be enqueued when that sample turns
 * out to need the host-libm correction (probability ~2^-36 per buffer, see fmd_demod_check) -- that buffer is then
 * delivered with status FMD_ERR_HIP instead of silently different audio.  read_sync-sized buffers never get there.
 * A submit that fails after it has touched a device cannot be rolled back (the Demod state of the parts before the
 * failing one has advanced): it is terminal for the sink -- this and every later acquire / submit return the same
 * error, poll / drain still deliver what was submitted before and then return it too. */
typedef struct fmd_sink fmd_sink;
typedef void (*fmd_sink_callback)(void *user, uint64_t seq, const int16_t *audio, const size_t *out_len,
                                  size_t out_cap, int status);
int fmd_sink_new(const fmd_demod_config *config, uint32_t n_channels, const int32_t *device_ids, uint32_t n_devices,
                 size_t nbytes, uint32_t depth, fmd_sink_callback callback, void *user, fmd_sink **out);
void fmd_sink_free(fmd_sink *s);
int fmd_sink_acquire(fmd_sink *s, uint8_t **iq);
int fmd_sink_submit(fmd_sink *s);
int fmd_sink_release(fmd_sink *s);
int fmd_sink_poll(fmd_sink *s);     /* deliver what has finished; returns the number of buffers delivered or < 0 */
int fmd_sink_drain(fmd_sink *s);    /* wait for and deliver everything in flight */
int fmd_sink_info(const fmd_sink *s, size_t *out_cap, uint32_t *n_devices, uint32_t *in_flight);
/* f64 samples that fell into the guard band / that the host libm corrected, summed over the device parts (fmd_demod_f64_stats). */
int fmd_sink_f64_stats(const fmd_sink *s, uint64_t *guarded, uint64_t *patched);

/* ---- rtl_tcp client-side IQ source ---------------------------------------------------- */
/* The reference ships the rtl_tcp SERVER (examples/rtl_tcp.rs); this is the matching client, so that a dongle on another
 * host feeds the sinks above without USB code here.  Wire format: a 12-byte handshake "RTL0" + tuner type + tuner gain
 * count, both u32 big-endian (send_handshake, examples/rtl_tcp.rs:691-697); then raw interleaved u8 IQ exactly as
 * RtlSdr::read_sync delivered it (sender_loop, :609-631); commands are 5 bytes, opcode + big-endian 32-bit parameter
 * (command_loop, :639-678).  Host code only (no GPU needed).
 *   fmd_rtltcp_open     : connect (timeout_ms, 0 = 10 s) and read the handshake; FMD_ERR_IO when it is not "RTL0";
 *   fmd_rtltcp_read_sync: RtlSdr::read_sync (src/lib.rs:153) -- fill buf, *n_read = bytes written; FEWER than nbytes
 *                         means the stream ended, which the reference's callers treat as "samples lost"
 *                         (examples/simple_fm.rs:122); a socket error or a timeout returns FMD_ERR_IO with
 *                         *n_read = the bytes that did arrive (the stream keeps its I/Q byte alignment);
 *   fmd_rtltcp_command  : one command; a negative (i32) parameter travels as its two's complement. */
typedef struct fmd_rtltcp fmd_rtltcp;
#define FMD_RTLTCP_SET_FREQUENCY       0x01   /* opcodes of command_loop, examples/rtl_tcp.rs:659-675 */
#define FMD_RTLTCP_SET_SAMPLE_RATE     0x02
#define FMD_RTLTCP_SET_GAIN_MODE       0x03
#define FMD_RTLTCP_SET_GAIN            0x04
#define FMD_RTLTCP_SET_FREQ_CORRECTION 0x05
#define FMD_RTLTCP_SET_IF_GAIN         0x06
#define FMD_RTLTCP_SET_TEST_MODE       0x07
#define FMD_RTLTCP_SET_AGC_MODE        0x08
#define FMD_RTLTCP_SET_DIRECT_SAMPLING 0x09
#define FMD_RTLTCP_SET_OFFSET_TUNING   0x0a
#define FMD_RTLTCP_SET_RTL_XTAL        0x0b
#define FMD_RTLTCP_SET_TUNER_XTAL      0x0c
#define FMD_RTLTCP_SET_GAIN_BY_INDEX   0x0d
#define FMD_RTLTCP_SET_BIAS_TEE        0x0e
int fmd_rtltcp_open(const char *host, uint16_t port, uint32_t timeout_ms, fmd_rtltcp **out);
void fmd_rtltcp_close(fmd_rtltcp *s);
int fmd_rtltcp_info(const fmd_rtltcp *s, uint32_t *tuner_type, uint32_t *gain_count);
int fmd_rtltcp_read_sync(fmd_rtltcp *s, uint8_t *buf, size_t nbytes, size_t *n_read);
/* read_sync for MANY sources behind one poll(): row c (row_stride bytes apart, the first nbytes of it) is filled from
 * sources[c], n_read[c] = bytes written to it.  FMD_OK with n_read[c] < nbytes: that stream ended early; FMD_ERR_IO: a
 * socket error, or no byte on any unfinished stream within the smallest timeout of the sources (n_read as far as it
 * got).  The receive() loop of simple_fm.rs:100-132 for a bank of streams. */
int fmd_rtltcp_read_many(fmd_rtltcp *const *sources, uint32_t n, uint8_t *base, size_t row_stride, size_t nbytes,
                         size_t *n_read);
int fmd_rtltcp_command(fmd_rtltcp *s, uint8_t opcode, uint32_t param);
/* receive() (simple_fm.rs:89-132) for a bank of rtl_tcp streams, below the binding: acquire the next slot, fill row c from
 * sources[c] (n_sources must equal the sink's n_channels; fmd_rtltcp_read_many: ONE poll() loop over all sockets) and
 * submit it.  *n_short = sources that ended before their row was full: when > 0 the slot has been released unsubmitted
 * and the run is over ("samples lost", :122-125) -- the sink stays usable, drain it to get what was submitted before.
 * fmd_sink_pump_rtltcp repeats that until a short read or max_buffers (0 = no limit) and then drains;
 * *n_submitted = buffers that went to the GPUs. */
int fmd_sink_fill_from_rtltcp(fmd_sink *s, fmd_rtltcp *const *sources, uint32_t n_sources, uint32_t *n_short);
int fmd_sink_pump_rtltcp(fmd_sink *s, fmd_rtltcp *const *sources, uint32_t n_sources, uint64_t max_buffers,
                         uint64_t *n_submitted);

/* ---- diagnostics ---------------------------------------------------------------------- */
const char *fmd_strerror(int status);
const char *fmd_last_error(void);          /* thread-local detail of the last failure          */
int fmd_device_count(int *count);          /* gfx950 devices visible to HIP                    */
int fmd_version(void);                     /* FMD_VERSION_MAJOR * 1000 + FMD_VERSION_MINOR     */
/* Kernel tiling (for benchmarks / DESIGN.md bookkeeping): of the handle's most recent launch,
 * or, before the first one, what a bank fed whole read_sync buffers will run. */
int fmd_demod_tiling(const fmd_demod *d, uint32_t *audio_per_tile, uint32_t *lds_bytes,
                     uint32_t *block_threads);
/* Which plan chose the LDS kernels' tile (diagnostics for measured rows): 0 = the caller (fmd_demod_set_tiling), 1 = the largest
 * tile that keeps 8 blocks per CU resident (20 KB), 2 = the 15.5 ... 17.3 KB window of the rows on the memory side; negative on
 * a null handle.  (The register-streaming kernel of downsample 2 / 4 has a tiling of its own: fmd_demod_tiling reports it.) */
int fmd_demod_tiling_plan(const fmd_demod *d);
/* Name of the kernel the handle's most recent launch ran, as `rocprofv3 --kernel-trace` prints
 * it (e.g. "fmd_tk::fmd_demod_tile_kernel<5, 2>"); "" before the first launch.  bench.py
 * quotes it in `roofline.kernel` instead of a constant. */
int fmd_demod_last_kernel(const fmd_demod *d, char *name, size_t cap);
/* Override the tiling (audio samples per workgroup tile; 0 = automatic).  Results never
 * depend on it; it exists for tuning sweeps. */
int fmd_demod_set_tiling(fmd_demod *d, uint32_t audio_per_tile);

#ifdef __cplusplus
}
#endif
#endif /* FMD_H */
