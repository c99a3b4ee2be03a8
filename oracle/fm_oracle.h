/*
 * fm_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the FM demodulation chain of ccostes/rtl-sdr-rs
 * `examples/simple_fm.rs` (crate v0.3.1).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The shipped HIP path
 * (rtl-sdr-rs_amd/csrc) never includes, links or calls anything in oracle/.
 *
 * Parity pinning: the three known-answer tests the reference holds for this path
 * (examples/simple_fm.rs:466-555, vectors taken from osmocom rtl_fm) are committed
 * as tests/golden/ref_kat_*.json and checked by tests/test_oracle_kat.py.
 * NOT pinned by any reference test (the reference has none): rotate_90, the -127
 * centring, call-boundary state carry, the f64 atan2 sample with a non-zero
 * predecessor, the i32 wrap in fast_atan2.  For those this file is the definition;
 * it is cross-checked against an independent pure-Python restatement in
 * tests/pyref.py.  The reference itself (Rust) cannot be built in this image
 * (no rustc/cargo) -- see DESIGN.md "Oracle".
 *
 * Integer semantics are those of a Rust *release* build (two's-complement wrap,
 * `as` casts truncate); compile with -fwrapv.
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* num_complex::Complex<i32> (Cargo.toml:29; used at simple_fm.rs:237-238,340,371,378) */
typedef struct { int32_t re, im; } fmo_cplx;

/* struct RadioConfig, simple_fm.rs:173-176 */
typedef struct { uint32_t capture_freq, capture_rate; } fmo_radio_config;

/* struct DemodConfig, simple_fm.rs:179-185 */
typedef struct {
    uint32_t rate_in;
    uint32_t rate_out;
    uint32_t rate_resample;
    uint32_t downsample;
    uint32_t output_scale;
} fmo_demod_config;

/* struct Demod, simple_fm.rs:232-239 */
typedef struct {
    fmo_demod_config config;
    size_t  prev_index;
    int32_t now_lpr;
    int32_t prev_lpr_index;
    fmo_cplx lp_now;
    fmo_cplx demod_pre;
} fmo_demod;

/* optimal_settings, simple_fm.rs:189-214.  The reference fills rate_in/rate_out/rate_resample
 * from the file-level consts SAMPLE_RATE/RATE_RESAMPLE (:207-209); at its only call site
 * (:48) `rate == SAMPLE_RATE`, so passing them as arguments is the same function there.
 * Returns 0, or -1 where the reference would panic (rate == 0: division by zero). */
int fmo_optimal_settings(uint32_t freq, uint32_t rate, uint32_t rate_resample,
                         fmo_radio_config *radio, fmo_demod_config *demod);

/* Demod::new, simple_fm.rs:243-252 */
void fmo_demod_new(fmo_demod *d, const fmo_demod_config *config);

/* Demod::rotate_90 scalar branch, simple_fm.rs:282-298.  len % 8 must be 0 (the
 * reference indexes out of bounds and panics otherwise) -> returns -1. */
int fmo_rotate_90(uint8_t *buf, size_t len);

/* centring closure simple_fm.rs:258 */
void fmo_center(const uint8_t *buf, size_t len, int16_t *out);

/* buf_to_complex simple_fm.rs:441-450; returns element count = len/2 */
size_t fmo_buf_to_complex(const int16_t *buf, size_t len, fmo_cplx *out);

/* Demod::low_pass_complex simple_fm.rs:337-352; out must hold (len + D-1)/D + 1 */
size_t fmo_low_pass_complex(fmo_demod *d, const fmo_cplx *buf, size_t len, fmo_cplx *out);

/* Demod::polar_discriminant simple_fm.rs:370-374 */
int32_t fmo_polar_discriminant(fmo_cplx a, fmo_cplx b);
/* Demod::polar_discriminant_fast simple_fm.rs:377-380 */
int32_t fmo_polar_discriminant_fast(fmo_cplx a, fmo_cplx b);
/* Demod::fast_atan2 simple_fm.rs:383-405 */
int32_t fmo_fast_atan2(int32_t y, int32_t x);
/* Number of samples so far at which the reference's `/` would have panicked (zero divisor after the i32 products of
 * `a * b.conj()` wrapped: downsample >= 305 with full-scale input only); the oracle returns 0 for them. */
long fmo_would_panic(void);

/* Demod::fm_demod simple_fm.rs:355-367; returns count, or -1 where the reference
 * asserts (len <= 1, :356). */
long fmo_fm_demod(fmo_demod *d, const fmo_cplx *buf, size_t len, int16_t *out);

/* Demod::low_pass_real simple_fm.rs:408-426; out must hold len entries.  Returns count,
 * or -1 where the reference would panic on an emit (rate_out / rate_resample == 0). */
long fmo_low_pass_real(fmo_demod *d, const int16_t *buf, size_t len, int16_t *out);

/* Demod::demodulate simple_fm.rs:256-269: six passes and the same intermediate
 * vectors (heap-allocated per call like the reference).  Does not modify `buf`.
 * Returns the number of s16 written to out (capacity out_cap), or
 *   -1  len % 8 != 0                     (reference: index panic, :286)
 *   -2  fewer than 2 decimated samples   (reference: assert, :356)
 *   -3  out_cap too small
 *   -4  rate_out < rate_resample with an emit (reference: divide by zero, :421)
 *   -5  allocation failure */
long fmo_demodulate(fmo_demod *d, const uint8_t *buf, size_t len, int16_t *out, size_t out_cap);

/* File mode of simple_fm (main(), :65-84) restricted to the COMPLETE 262144-byte blocks of the
 * input (the reference ignores the read count and never terminates at
 * EOF).  block_len is DEFAULT_BUF_LENGTH (src/lib.rs:25) in the reference.  Returns total s16
 * written or a negative fmo_demodulate code. */
long fmo_file_mode(fmo_demod *d, const uint8_t *data, size_t len, size_t block_len,
                   int16_t *out, size_t out_cap);

/* CPU-baseline helper for bench.py: n_channels independent Demod instances, each fed
 * `calls` consecutive blocks of block_len bytes taken from iq[c][call][block_len]
 * (channel stride = calls*block_len), statically partitioned over n_threads pthreads.
 * Writes per-channel output counts of the LAST call to out_len (may be NULL) and returns
 * elapsed wall seconds (CLOCK_MONOTONIC) of the demodulation only, or < 0 on error. */
double fmo_bench_batch(const fmo_demod_config *config, const uint8_t *iq, size_t n_channels,
                       size_t calls, size_t block_len, int n_threads, uint64_t *checksum,
                       uint32_t *out_len);

/* ---- Row G' of SURVEY 8a: generalised tapped decimating FIR (BASELINE config 4) -------------------
 * NOT IN THE REFERENCE (its only tapped FIR is programmed into the RTL2832U, src/rtlsdr.rs:525-558, and
 * never runs on the host); this is the builder's definition, "parity unpinned" by construction:
 *     y[m] = sum_{t < n_taps} taps[t] * x[decim * m + t]
 * over the stream x[n] of rotated + centred complex samples (rotate_90 :276-299, `- 127` :258) counted from
 * the first sample ever fed; y[m] is emitted by the call in which x[decim*m + n_taps - 1] arrives.  With
 * taps = 1...1, n_taps = decim = D it is exactly Demod::low_pass_complex (:337-352) -- tested. */
typedef struct fmo_fir fmo_fir;
fmo_fir *fmo_fir_new(const int16_t *taps, uint32_t n_taps, uint32_t decim);
void fmo_fir_free(fmo_fir *f);
/* Returns the number of complex outputs written (capacity out_cap), -1 len % 8, -3 capacity. */
long fmo_fir_filter(fmo_fir *f, const uint8_t *buf, size_t len, fmo_cplx *out, size_t out_cap);

/* ---- the tapped FIR in place of the boxcar inside Demod::demodulate (BASELINE north_star: "the FIR + demod +
 * resample stages fused").  NOT IN THE REFERENCE; the definition is the composition of the functions above:
 *     lowpassed = floor(fmo_fir_filter(buf) / 2^shift)   -- instead of low_pass_complex, simple_fm.rs:261
 *     fmo_fm_demod(lowpassed)  (:264, f64 sample at the first output of every call, demod_pre carried)
 *     fmo_low_pass_real(...)   (:267, rate_out -> rate_resample)
 * `shift` is the usual fixed-point normalisation of an integer-tap filter (0 = none).  With taps = 1...1,
 * n_taps = decim = D and shift = 0 it IS fmo_demodulate -- tested: that reduction is its anchor to the reference.
 * Return codes as fmo_demodulate (-2: the call yields fewer than 2 filter outputs, assert at :356). */
typedef struct fmo_firdemod fmo_firdemod;
fmo_firdemod *fmo_firdemod_new(const int16_t *taps, uint32_t n_taps, uint32_t decim, uint32_t shift,
                               uint32_t rate_out, uint32_t rate_resample);
void fmo_firdemod_free(fmo_firdemod *f);
long fmo_firdemod_demodulate(fmo_firdemod *f, const uint8_t *buf, size_t len, int16_t *out, size_t out_cap);
void fmo_firdemod_state(const fmo_firdemod *f, fmo_demod *out);
int fmo_firdemod_batch(fmo_firdemod **fs, const uint8_t *iq, size_t n_channels, size_t len, int16_t *out,
                       size_t out_cap, uint32_t *out_len, int n_threads);

/* firs[c] fed iq[c] (channel-major iq[n_channels][len], out[n_channels][out_cap]) over n_threads pthreads: the
 * config-4 check at its real size (256 channels x 2 MiB).  Returns 0 or the first negative fmo_fir_filter code. */
int fmo_fir_filter_batch(fmo_fir **firs, const uint8_t *iq, size_t n_channels, size_t len, fmo_cplx *out,
                         size_t out_cap, uint32_t *out_len, int n_threads);

/* Test helper: demods[c].demodulate(iq[c]) for c in [0, n_channels), channel-major buffers
 * iq[n_channels][len], out[n_channels][out_cap], out_len[n_channels], over n_threads pthreads.
 * Returns 0 or the first negative fmo_demodulate code. */
int fmo_demodulate_batch(fmo_demod *demods, const uint8_t *iq, size_t n_channels, size_t len,
                         int16_t *out, size_t out_cap, uint32_t *out_len, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
