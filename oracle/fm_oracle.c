/*
 * fm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See fm_oracle.h.
 *
 * Pass-by-pass restatement of examples/simple_fm.rs (ccostes/rtl-sdr-rs v0.3.1); every
 * function cites the lines it follows.  Build: gcc -O2 -fwrapv (see oracle/Makefile).
 */
#include "fm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* std::f64::consts::PI (simple_fm.rs:17,373) */
static const double FMO_PI = 3.14159265358979323846264338327950288;

/* ---- optimal_settings, simple_fm.rs:189-214 ------------------------------------------ */
int fmo_optimal_settings(uint32_t freq, uint32_t rate, uint32_t rate_resample,
                         fmo_radio_config *radio, fmo_demod_config *demod)
{
    if (rate == 0) return -1;                           /* :190 would divide by zero */
    uint32_t downsample = (1000000u / rate) + 1;        /* :190 */
    uint32_t capture_rate = downsample * rate;          /* :192 */
    uint32_t capture_freq = freq + capture_rate / 4;    /* :195 offset tuning */
    uint32_t output_scale = (1u << 15) / (128u * downsample); /* :197 */
    if (output_scale < 1) output_scale = 1;             /* :198-200 */
    if (radio) {
        radio->capture_freq = capture_freq;             /* :203 */
        radio->capture_rate = capture_rate;             /* :204 */
    }
    if (demod) {
        demod->rate_in = rate;                          /* :207 (SAMPLE_RATE) */
        demod->rate_out = rate;                         /* :208 (SAMPLE_RATE) */
        demod->rate_resample = rate_resample;           /* :209 (RATE_RESAMPLE) */
        demod->downsample = downsample;                 /* :210 */
        demod->output_scale = output_scale;             /* :211 */
    }
    return 0;
}

/* ---- Demod::new, simple_fm.rs:243-252 ------------------------------------------------- */
void fmo_demod_new(fmo_demod *d, const fmo_demod_config *config)
{
    d->config = *config;
    d->prev_index = 0;
    d->now_lpr = 0;
    d->prev_lpr_index = 0;
    d->lp_now.re = 0; d->lp_now.im = 0;
    d->demod_pre.re = 0; d->demod_pre.im = 0;
}

/* ---- Demod::rotate_90 (scalar cfg branch), simple_fm.rs:282-298 ------------------------ */
int fmo_rotate_90(uint8_t *buf, size_t len)
{
    if (len % 8 != 0) return -1;
    uint8_t tmp;
    for (size_t i = 0; i < len; i += 8) {               /* :284 */
        /* uint8_t negation = 255 - x */
        tmp = (uint8_t)(255 - buf[i + 3]);              /* :286 */
        buf[i + 3] = buf[i + 2];                        /* :287 */
        buf[i + 2] = tmp;                               /* :288 */

        buf[i + 4] = (uint8_t)(255 - buf[i + 4]);       /* :290 */
        buf[i + 5] = (uint8_t)(255 - buf[i + 5]);       /* :291 */

        tmp = (uint8_t)(255 - buf[i + 6]);              /* :293 */
        buf[i + 6] = buf[i + 7];                        /* :294 */
        buf[i + 7] = tmp;                               /* :295 */
    }
    return 0;
}

/* ---- `*val as i16 - 127`, simple_fm.rs:258 --------------------------------------------- */
void fmo_center(const uint8_t *buf, size_t len, int16_t *out)
{
    for (size_t i = 0; i < len; i++) out[i] = (int16_t)((int16_t)buf[i] - 127);
}

/* ---- buf_to_complex, simple_fm.rs:441-450: windows(2).step_by(2) ------------------------ */
size_t fmo_buf_to_complex(const int16_t *buf, size_t len, fmo_cplx *out)
{
    size_t n = 0;
    for (size_t i = 0; i + 1 < len; i += 2) {           /* an odd trailing element is dropped */
        out[n].re = (int32_t)buf[i];
        out[n].im = (int32_t)buf[i + 1];
        n++;
    }
    return n;
}

/* ---- Demod::low_pass_complex, simple_fm.rs:337-352 -------------------------------------- */
size_t fmo_low_pass_complex(fmo_demod *d, const fmo_cplx *buf, size_t len, fmo_cplx *out)
{
    size_t n = 0;
    for (size_t orig = 0; orig < len; orig++) {
        d->lp_now.re += buf[orig].re;                   /* :340 */
        d->lp_now.im += buf[orig].im;

        d->prev_index += 1;                             /* :342 */
        if (d->prev_index < (size_t)d->config.downsample) continue; /* :343 */

        out[n++] = d->lp_now;                           /* :347 */
        d->lp_now.re = 0; d->lp_now.im = 0;             /* :348 */
        d->prev_index = 0;                              /* :349 */
    }
    return n;
}

/* num_complex 0.4 `a * b.conj()` on Complex<i32>: (ar*br - ai*(-bi), ar*(-bi) + ai*br),
 * plain (wrapping in release) i32 arithmetic. */
static fmo_cplx mul_conj(fmo_cplx a, fmo_cplx b)
{
    fmo_cplx c;
    int32_t cbi = -b.im;
    c.re = a.re * b.re - a.im * cbi;
    c.im = a.re * cbi + a.im * b.re;
    return c;
}

/* ---- Demod::polar_discriminant, simple_fm.rs:370-374 ------------------------------------ */
int32_t fmo_polar_discriminant(fmo_cplx a, fmo_cplx b)
{
    fmo_cplx c = mul_conj(a, b);                        /* :371 */
    double angle = atan2((double)c.im, (double)c.re);   /* :372 f64::atan2 -> libm */
    double v = angle / FMO_PI * (double)(1 << 14);      /* :373 */
    /* Rust `as i32`: truncate toward zero, saturate, NaN -> 0.  |v| <= 16384 here. */
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}

/* ---- Demod::fast_atan2, simple_fm.rs:383-405 -------------------------------------------- */
static long g_would_panic = 0;     /* samples at which the reference's integer division would panic (see below) */
long fmo_would_panic(void) { return __atomic_load_n(&g_would_panic, __ATOMIC_RELAXED); }

int32_t fmo_fast_atan2(int32_t y, int32_t x)
{
    /* Pre-scaled for i16: pi = 1 << 14 */
    const int32_t pi4 = 1 << 12;                        /* :386 */
    const int32_t pi34 = 3 * (1 << 12);                 /* :387 */
    if (x == 0 && y == 0) return 0;                     /* :388-390 */
    int32_t yabs = y;
    if (yabs < 0) yabs = -yabs;                         /* :391-394 */
    int32_t angle;
    if (x >= 0) {
        /* :397  (pi4 as i64 * (x - yabs) as i64) as i32 / (x + yabs):
         * i32 subtract, widen, i64 multiply, TRUNCATE to i32, then i32 divide. */
        int32_t num = (int32_t)(uint32_t)((int64_t)pi4 * (int64_t)(int32_t)(x - yabs));
        int32_t den = x + yabs;
        if (den == 0 || (den == -1 && num == INT32_MIN)) { __atomic_fetch_add(&g_would_panic, 1, __ATOMIC_RELAXED); return 0; }
        angle = pi4 - num / den;
    } else {
        int32_t num = (int32_t)(uint32_t)((int64_t)pi4 * (int64_t)(int32_t)(x + yabs));
        int32_t den = yabs - x;
        /* Rust's `/` panics on a zero divisor (and on MIN / -1) in release builds too.  Only reachable once the i32
         * products of `a * b.conj()` wrap -- downsample >= 305 with full-scale input, e.g. a = (32768, 32768),
         * b = (65536, 0): x = y = 2^31 -> i32::MIN, yabs = MIN, yabs - x = 0.  The oracle counts such samples
         * (fmo_would_panic) instead of dying; the tests assert the count stays 0 for their inputs. */
        if (den == 0 || (den == -1 && num == INT32_MIN)) { __atomic_fetch_add(&g_would_panic, 1, __ATOMIC_RELAXED); return 0; }
        angle = pi34 - num / den;                       /* :399 */
    }
    if (y < 0) return -angle;                           /* :401-403 */
    return angle;
}

/* ---- Demod::polar_discriminant_fast, simple_fm.rs:377-380 ------------------------------- */
int32_t fmo_polar_discriminant_fast(fmo_cplx a, fmo_cplx b)
{
    fmo_cplx c = mul_conj(a, b);
    return fmo_fast_atan2(c.im, c.re);
}

/* ---- Demod::fm_demod, simple_fm.rs:355-367 ---------------------------------------------- */
long fmo_fm_demod(fmo_demod *d, const fmo_cplx *buf, size_t len, int16_t *out)
{
    if (!(len > 1)) return -1;                          /* :356 assert */
    int32_t pcm = fmo_polar_discriminant(buf[0], d->demod_pre); /* :359 */
    out[0] = (int16_t)(uint16_t)(uint32_t)pcm;          /* :360 `as i16` wraps */
    for (size_t i = 1; i < len; i++) {
        pcm = fmo_polar_discriminant_fast(buf[i], buf[i - 1]);  /* :362 */
        out[i] = (int16_t)(uint16_t)(uint32_t)pcm;      /* :363 */
    }
    d->demod_pre = buf[len - 1];                        /* :365 */
    return (long)len;
}

/* ---- Demod::low_pass_real, simple_fm.rs:408-426 ----------------------------------------- */
long fmo_low_pass_real(fmo_demod *d, const int16_t *buf, size_t len, int16_t *out)
{
    long n = 0;
    /* Simple square-window FIR */
    uint32_t slow = d->config.rate_resample;            /* :411 */
    uint32_t fast = d->config.rate_out;                 /* :412 */
    size_t i = 0;
    while (i < len) {
        d->now_lpr += (int32_t)buf[i];                  /* :415 */
        i += 1;
        d->prev_lpr_index += (int32_t)slow;             /* :417 */
        if (d->prev_lpr_index < (int32_t)fast) continue;/* :418 */
        if (slow == 0 || (int32_t)(fast / slow) == 0) return -1;  /* :421 would panic */
        out[n++] = (int16_t)(uint16_t)(uint32_t)(d->now_lpr / (int32_t)(fast / slow)); /* :421 */
        d->prev_lpr_index -= (int32_t)fast;             /* :422 */
        d->now_lpr = 0;                                 /* :423 */
    }
    return n;
}

/* A push-grown vector, like the `vec![]` + push of :338,:357,:409 (amortised doubling). */
typedef struct { void *p; size_t len, cap, esz; } growvec;
static int gv_push(growvec *v, const void *e)
{
    if (v->len == v->cap) {
        size_t ncap = v->cap ? v->cap * 2 : 4;
        void *np = realloc(v->p, ncap * v->esz);
        if (!np) return -1;
        v->p = np; v->cap = ncap;
    }
    memcpy((char *)v->p + v->len * v->esz, e, v->esz);
    v->len++;
    return 0;
}

/* ---- Demod::demodulate, simple_fm.rs:256-269 -------------------------------------------- */
long fmo_demodulate(fmo_demod *d, const uint8_t *buf_in, size_t len, int16_t *out, size_t out_cap)
{
    if (len % 8 != 0) return -1;
    long rc = -5;
    uint8_t *buf = NULL; int16_t *buf_signed = NULL; fmo_cplx *complex_ = NULL;
    int16_t *demodulated = NULL;
    growvec lowpassed = {NULL, 0, 0, sizeof(fmo_cplx)};

    buf = (uint8_t *)malloc(len ? len : 1);             /* the owned Vec<u8> argument */
    if (!buf) goto done;
    memcpy(buf, buf_in, len);
    fmo_rotate_90(buf, len);                            /* :257 */

    buf_signed = (int16_t *)malloc((len ? len : 1) * sizeof(int16_t));
    if (!buf_signed) goto done;
    fmo_center(buf, len, buf_signed);                   /* :258 */

    complex_ = (fmo_cplx *)malloc((len / 2 + 1) * sizeof(fmo_cplx));
    if (!complex_) goto done;
    size_t nc = fmo_buf_to_complex(buf_signed, len, complex_);  /* :259 */

    /* :261 low-pass filter to downsample to our desired sample rate (push-grown result) */
    for (size_t orig = 0; orig < nc; orig++) {
        d->lp_now.re += complex_[orig].re;
        d->lp_now.im += complex_[orig].im;
        d->prev_index += 1;
        if (d->prev_index < (size_t)d->config.downsample) continue;
        if (gv_push(&lowpassed, &d->lp_now)) goto done;
        d->lp_now.re = 0; d->lp_now.im = 0;
        d->prev_index = 0;
    }

    /* :264 Demodulate FM signal */
    if (!(lowpassed.len > 1)) { rc = -2; goto done; }
    demodulated = (int16_t *)malloc(lowpassed.len * sizeof(int16_t));
    if (!demodulated) goto done;
    fmo_fm_demod(d, (const fmo_cplx *)lowpassed.p, lowpassed.len, demodulated);

    /* :267 Resample and return result */
    {
        int16_t *res = (int16_t *)malloc(lowpassed.len * sizeof(int16_t));
        if (!res) goto done;
        long n = fmo_low_pass_real(d, demodulated, lowpassed.len, res);
        if (n < 0) rc = -4;
        else if ((size_t)n > out_cap) rc = -3;
        else { memcpy(out, res, (size_t)n * sizeof(int16_t)); rc = n; }
        free(res);
    }
done:
    free(buf); free(buf_signed); free(complex_); free(demodulated); free(lowpassed.p);
    return rc;
}

/* ---- simple_fm file mode over complete blocks, simple_fm.rs:65-84 ----------------------- */
long fmo_file_mode(fmo_demod *d, const uint8_t *data, size_t len, size_t block_len,
                   int16_t *out, size_t out_cap)
{
    long total = 0;
    if (block_len == 0) return -1;
    for (size_t off = 0; off + block_len <= len; off += block_len) {
        long n = fmo_demodulate(d, data + off, block_len, out + total, out_cap - (size_t)total);
        if (n < 0) return n;
        total += n;
    }
    return total;
}

/* ---- cpu_baseline helper -------------------------------------------------------------- */
typedef struct {
    const fmo_demod_config *config; const uint8_t *iq;
    size_t c0, c1, calls, block_len; uint64_t checksum; uint32_t *out_len; int err;
} bench_job;

static void *bench_worker(void *arg)
{
    bench_job *j = (bench_job *)arg;
    size_t cap = j->block_len / 2 + 16;
    int16_t *out = (int16_t *)malloc(cap * sizeof(int16_t));
    uint64_t h = 0;
    for (size_t c = j->c0; c < j->c1; c++) {
        fmo_demod d;
        fmo_demod_new(&d, j->config);
        long n = 0;
        for (size_t k = 0; k < j->calls; k++) {
            n = fmo_demodulate(&d, j->iq + (c * j->calls + k) * j->block_len, j->block_len, out, cap);
            if (n < 0) { j->err = (int)n; break; }
            for (long i = 0; i < n; i++)
                h = h * 1099511628211ull + (uint16_t)out[i] + 0x9e3779b97f4a7c15ull * (c + 1);
        }
        if (j->out_len && n >= 0) j->out_len[c] = (uint32_t)n;
    }
    j->checksum = h;
    free(out);
    return NULL;
}

double fmo_bench_batch(const fmo_demod_config *config, const uint8_t *iq, size_t n_channels,
                       size_t calls, size_t block_len, int n_threads, uint64_t *checksum,
                       uint32_t *out_len)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_channels) n_threads = (int)(n_channels ? n_channels : 1);
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    bench_job *jobs = (bench_job *)calloc((size_t)n_threads, sizeof(bench_job));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; t++) {
        jobs[t].config = config; jobs[t].iq = iq; jobs[t].calls = calls;
        jobs[t].block_len = block_len; jobs[t].out_len = out_len;
        jobs[t].c0 = n_channels * (size_t)t / (size_t)n_threads;
        jobs[t].c1 = n_channels * (size_t)(t + 1) / (size_t)n_threads;
        if (n_threads == 1) bench_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
    }
    uint64_t h = 0; int err = 0;
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        h ^= jobs[t].checksum;
        if (jobs[t].err) err = jobs[t].err;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (checksum) *checksum = h;
    free(th); free(jobs);
    if (err) return (double)err;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- batch test helper ------------------------------------------------------------------ */
typedef struct {
    fmo_demod *demods; const uint8_t *iq; size_t c0, c1, len; int16_t *out; size_t out_cap;
    uint32_t *out_len; int err;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *j = (batch_job *)arg;
    for (size_t c = j->c0; c < j->c1; c++) {
        long n = fmo_demodulate(&j->demods[c], j->iq + c * j->len, j->len, j->out + c * j->out_cap, j->out_cap);
        if (n < 0) { if (!j->err) j->err = (int)n; continue; }
        j->out_len[c] = (uint32_t)n;
    }
    return NULL;
}

int fmo_demodulate_batch(fmo_demod *demods, const uint8_t *iq, size_t n_channels, size_t len,
                         int16_t *out, size_t out_cap, uint32_t *out_len, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_channels) n_threads = (int)(n_channels ? n_channels : 1);
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    batch_job *jobs = (batch_job *)calloc((size_t)n_threads, sizeof(batch_job));
    for (int t = 0; t < n_threads; t++) {
        batch_job b = {demods, iq, n_channels * (size_t)t / (size_t)n_threads,
                       n_channels * (size_t)(t + 1) / (size_t)n_threads, len, out, out_cap, out_len, 0};
        jobs[t] = b;
        if (n_threads == 1) batch_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    int err = 0;
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        if (jobs[t].err && !err) err = jobs[t].err;
    }
    free(th); free(jobs);
    return err;
}

/* ---- Row G': generalised tapped decimating FIR (no reference symbol; see fm_oracle.h) ------------- */
struct fmo_fir {
    int32_t *taps; uint32_t n_taps, decim;
    fmo_cplx *hist;       /* the last n_taps - 1 stream samples */
    uint64_t pos;         /* samples consumed so far */
};

fmo_fir *fmo_fir_new(const int16_t *taps, uint32_t n_taps, uint32_t decim)
{
    if (!taps || n_taps == 0 || decim == 0) return NULL;
    fmo_fir *f = (fmo_fir *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->taps = (int32_t *)malloc(n_taps * sizeof(int32_t));
    f->hist = (fmo_cplx *)calloc(n_taps, sizeof(fmo_cplx));
    if (!f->taps || !f->hist) { fmo_fir_free(f); return NULL; }
    for (uint32_t t = 0; t < n_taps; t++) f->taps[t] = taps[t];
    f->n_taps = n_taps; f->decim = decim; f->pos = 0;
    return f;
}

void fmo_fir_free(fmo_fir *f)
{
    if (!f) return;
    free(f->taps); free(f->hist); free(f);
}

long fmo_fir_filter(fmo_fir *f, const uint8_t *buf_in, size_t len, fmo_cplx *out, size_t out_cap)
{
    if (len % 8 != 0) return -1;
    const size_t ns = len / 2, H = f->n_taps - 1;
    uint8_t *buf = (uint8_t *)malloc(len ? len : 1);
    int16_t *sig = (int16_t *)malloc((len ? len : 1) * sizeof(int16_t));
    fmo_cplx *v = (fmo_cplx *)malloc((H + ns + 1) * sizeof(fmo_cplx));   /* history ++ this call */
    long n = -5;
    if (!buf || !sig || !v) goto done;
    memcpy(buf, buf_in, len);
    fmo_rotate_90(buf, len);                       /* the rotation phase restarts per call; calls are multiples of */
    fmo_center(buf, len, sig);                     /* 4 samples, so it equals the stream position mod 4            */
    memcpy(v, f->hist, H * sizeof(fmo_cplx));
    fmo_buf_to_complex(sig, len, v + H);
    {
        const uint64_t S = f->pos, T = f->n_taps, M = f->decim;
        const uint64_t m0 = S >= T ? (S - T) / M + 1 : 0;
        const uint64_t m1 = S + ns >= T ? (S + ns - T) / M + 1 : 0;
        n = 0;
        for (uint64_t m = m0; m < m1; m++) {
            if ((size_t)n >= out_cap) { n = -3; goto done; }
            const size_t base = (size_t)(M * m + H - S);         /* virtual index of stream sample M*m */
            int32_t re = 0, im = 0;
            for (uint32_t t = 0; t < f->n_taps; t++) { re += f->taps[t] * v[base + t].re; im += f->taps[t] * v[base + t].im; }
            out[n].re = re; out[n].im = im; n++;
        }
        memmove(f->hist, v + ns, H * sizeof(fmo_cplx));          /* last H samples of history ++ call */
        f->pos = S + ns;
    }
done:
    free(buf); free(sig); free(v);
    return n;
}

/* ---- threaded FIR over a bank (test helper for BASELINE configs[3] at its real size) ----------- */
typedef struct {
    fmo_fir **firs; const uint8_t *iq; size_t c0, c1, len; fmo_cplx *out; size_t out_cap; uint32_t *out_len; int err;
} fir_job;

static void *fir_worker(void *arg)
{
    fir_job *j = (fir_job *)arg;
    for (size_t c = j->c0; c < j->c1; c++) {
        long n = fmo_fir_filter(j->firs[c], j->iq + c * j->len, j->len, j->out + c * j->out_cap, j->out_cap);
        if (n < 0) { if (!j->err) j->err = (int)n; continue; }
        j->out_len[c] = (uint32_t)n;
    }
    return NULL;
}

int fmo_fir_filter_batch(fmo_fir **firs, const uint8_t *iq, size_t n_channels, size_t len, fmo_cplx *out,
                         size_t out_cap, uint32_t *out_len, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_channels) n_threads = (int)(n_channels ? n_channels : 1);
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    fir_job *jobs = (fir_job *)calloc((size_t)n_threads, sizeof(fir_job));
    for (int t = 0; t < n_threads; t++) {
        fir_job b = {firs, iq, n_channels * (size_t)t / (size_t)n_threads, n_channels * (size_t)(t + 1) / (size_t)n_threads,
                     len, out, out_cap, out_len, 0};
        jobs[t] = b;
        if (n_threads == 1) fir_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, fir_worker, &jobs[t]);
    }
    int err = 0;
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        if (jobs[t].err && !err) err = jobs[t].err;
    }
    free(th); free(jobs);
    return err;
}

/* ---- tapped FIR in place of the boxcar: FIR -> fm_demod -> low_pass_real (BASELINE north_star's "FIR + demod +
 * resample"; see fm_oracle.h).  Composition of the functions above; nothing new is computed here. */
struct fmo_firdemod {
    fmo_fir *fir;
    fmo_demod d;            /* demod_pre, now_lpr, prev_lpr_index (prev_index / lp_now stay 0: the FIR owns the decimation) */
    uint32_t shift;
};

fmo_firdemod *fmo_firdemod_new(const int16_t *taps, uint32_t n_taps, uint32_t decim, uint32_t shift,
                               uint32_t rate_out, uint32_t rate_resample)
{
    fmo_firdemod *f = (fmo_firdemod *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->fir = fmo_fir_new(taps, n_taps, decim);
    if (!f->fir || shift > 31) { fmo_firdemod_free(f); return NULL; }
    fmo_demod_config cfg = {rate_out, rate_out, rate_resample, decim, 1};
    fmo_demod_new(&f->d, &cfg);
    f->shift = shift;
    return f;
}

void fmo_firdemod_free(fmo_firdemod *f)
{
    if (!f) return;
    fmo_fir_free(f->fir);
    free(f);
}

long fmo_firdemod_demodulate(fmo_firdemod *f, const uint8_t *buf, size_t len, int16_t *out, size_t out_cap)
{
    if (len % 8 != 0) return -1;
    const size_t cap = len / 2 / f->fir->decim + 2;
    fmo_cplx *lowpassed = (fmo_cplx *)malloc(cap * sizeof(fmo_cplx));
    int16_t *demodulated = (int16_t *)malloc(cap * sizeof(int16_t));
    int16_t *res = (int16_t *)malloc(cap * sizeof(int16_t));
    long rc = -5;
    if (!lowpassed || !demodulated || !res) goto done;
    {
        const long n = fmo_fir_filter(f->fir, buf, len, lowpassed, cap);       /* in place of low_pass_complex, :261 */
        if (n < 0) { rc = n; goto done; }
        for (long i = 0; i < n; i++) {                                         /* fixed-point normalisation: floor(y / 2^shift) */
            lowpassed[i].re = (int32_t)((int64_t)lowpassed[i].re >> f->shift);
            lowpassed[i].im = (int32_t)((int64_t)lowpassed[i].im >> f->shift);
        }
        if (!(n > 1)) { rc = -2; goto done; }                                  /* assert!(buf.len() > 1), :356 */
        fmo_fm_demod(&f->d, lowpassed, (size_t)n, demodulated);                /* :264 */
        const long k = fmo_low_pass_real(&f->d, demodulated, (size_t)n, res);  /* :267 */
        if (k < 0) rc = -4;
        else if ((size_t)k > out_cap) rc = -3;
        else { memcpy(out, res, (size_t)k * sizeof(int16_t)); rc = k; }
    }
done:
    free(lowpassed); free(demodulated); free(res);
    return rc;
}

void fmo_firdemod_state(const fmo_firdemod *f, fmo_demod *out) { *out = f->d; }

typedef struct {
    fmo_firdemod **fs; const uint8_t *iq; size_t c0, c1, len; int16_t *out; size_t out_cap; uint32_t *out_len; int err;
} firdemod_job;

static void *firdemod_worker(void *arg)
{
    firdemod_job *j = (firdemod_job *)arg;
    for (size_t c = j->c0; c < j->c1; c++) {
        long n = fmo_firdemod_demodulate(j->fs[c], j->iq + c * j->len, j->len, j->out + c * j->out_cap, j->out_cap);
        if (n < 0) { if (!j->err) j->err = (int)n; continue; }
        j->out_len[c] = (uint32_t)n;
    }
    return NULL;
}

int fmo_firdemod_batch(fmo_firdemod **fs, const uint8_t *iq, size_t n_channels, size_t len, int16_t *out,
                       size_t out_cap, uint32_t *out_len, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_channels) n_threads = (int)(n_channels ? n_channels : 1);
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    firdemod_job *jobs = (firdemod_job *)calloc((size_t)n_threads, sizeof(firdemod_job));
    for (int t = 0; t < n_threads; t++) {
        firdemod_job b = {fs, iq, n_channels * (size_t)t / (size_t)n_threads, n_channels * (size_t)(t + 1) / (size_t)n_threads,
                          len, out, out_cap, out_len, 0};
        jobs[t] = b;
        if (n_threads == 1) firdemod_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, firdemod_worker, &jobs[t]);
    }
    int err = 0;
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        if (jobs[t].err && !err) err = jobs[t].err;
    }
    free(th); free(jobs);
    return err;
}
