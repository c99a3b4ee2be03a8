/* c_abi_pump.c -- plain C11 consumer of the many-stream receive() of include/fmd.h (TEST INFRASTRUCTURE).
 *
 * receive() + process() + output() of examples/simple_fm.rs:89-170 for a bank of rtl_tcp streams, entirely below the
 * binding: N fmd_rtltcp sources -> fmd_sink_pump_rtltcp (one poll() loop fills each slot's rows, submit, repeat until
 * a stream runs short: "samples lost", :122-125) -> the completion callback appends every delivered buffer to a file as
 *     u64 seq, u32 n_channels, then per channel: u32 out_len, out_len x s16
 * which tests/test_c_abi.py compares with the oracle.  Usage: c_abi_pump <out.bin> <nbytes> <port>...
 * Exit code 0 on success, 2 when the library reports no usable device (the product has no CPU path). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fmd.h"

static FILE *g_out;
static uint32_t g_channels;
static int g_bad;

static void on_audio(void *user, uint64_t seq, const int16_t *audio, const size_t *out_len, size_t out_cap, int status)
{
    (void)user;
    if (status != FMD_OK) { g_bad = status; return; }
    fwrite(&seq, sizeof seq, 1, g_out);
    fwrite(&g_channels, sizeof g_channels, 1, g_out);
    for (uint32_t c = 0; c < g_channels; ++c) {
        const uint32_t n = (uint32_t)out_len[c];
        fwrite(&n, sizeof n, 1, g_out);
        fwrite(audio + (size_t)c * out_cap, sizeof(int16_t), n, g_out);
    }
}

static int fail(const char *what, int rc)
{
    fprintf(stderr, "%s: %s (%d): %s\n", what, fmd_strerror(rc), rc, fmd_last_error());
    return rc == FMD_ERR_NO_DEVICE ? 2 : 1;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s out.bin nbytes port...\n", argv[0]); return 1; }
    const size_t nbytes = (size_t)strtoul(argv[2], NULL, 10);
    const uint32_t n = (uint32_t)(argc - 3);
    fmd_radio_config radio;
    fmd_demod_config cfg;
    int rc = fmd_optimal_settings(94900000u, 170000u, 32000u, &radio, &cfg);     /* simple_fm.rs:48 */
    if (rc != FMD_OK) return fail("fmd_optimal_settings", rc);
    fmd_rtltcp **src = calloc(n, sizeof *src);
    if (!src) return 1;
    for (uint32_t c = 0; c < n; ++c) {
        rc = fmd_rtltcp_open("127.0.0.1", (uint16_t)atoi(argv[3 + c]), 10000u, &src[c]);
        if (rc != FMD_OK) return fail("fmd_rtltcp_open", rc);
        rc = fmd_rtltcp_command(src[c], FMD_RTLTCP_SET_FREQUENCY, radio.capture_freq);     /* config_sdr, :217-229 */
        if (rc == FMD_OK) rc = fmd_rtltcp_command(src[c], FMD_RTLTCP_SET_SAMPLE_RATE, radio.capture_rate);
        if (rc != FMD_OK) return fail("fmd_rtltcp_command", rc);
    }
    g_out = fopen(argv[1], "wb");
    if (!g_out) return 1;
    g_channels = n;
    const int32_t dev0 = 0;
    fmd_sink *sink = NULL;
    rc = fmd_sink_new(&cfg, n, &dev0, 1u, nbytes, 3u, on_audio, NULL, &sink);
    if (rc != FMD_OK) return fail("fmd_sink_new", rc);
    uint64_t submitted = 0;
    rc = fmd_sink_pump_rtltcp(sink, (fmd_rtltcp *const *)src, n, 0u, &submitted);  /* until a stream runs short */
    if (rc != FMD_OK) return fail("fmd_sink_pump_rtltcp", rc);
    /* wrong source count: an argument error, nothing acquired */
    uint32_t shorts = 0;
    if (n > 1 && fmd_sink_fill_from_rtltcp(sink, (fmd_rtltcp *const *)src, n - 1, &shorts) != FMD_ERR_INVALID_ARG) { fprintf(stderr, "source count not checked\n"); return 1; }
    /* the streams have ended: one more fill reports every source short and leaves the sink usable */
    rc = fmd_sink_fill_from_rtltcp(sink, (fmd_rtltcp *const *)src, n, &shorts);
    if (rc != FMD_OK || shorts != n) { fprintf(stderr, "fill after the end: rc %d, %u short\n", rc, (unsigned)shorts); return 1; }
    uint8_t *slot = NULL;
    if (fmd_sink_acquire(sink, &slot) != FMD_OK || fmd_sink_release(sink) != FMD_OK) { fprintf(stderr, "sink not usable after a short read\n"); return 1; }
    fmd_sink_free(sink);
    fclose(g_out);
    for (uint32_t c = 0; c < n; ++c) fmd_rtltcp_close(src[c]);
    free(src);
    if (g_bad) return fail("completion callback", g_bad);
    printf("submitted %lu\n", (unsigned long)submitted);
    return 0;
}
