/* c_abi_smoke.c -- a plain C11 consumer of include/fmd.h (TEST INFRASTRUCTURE).
 *
 * Proves the boundary is a C ABI, not a C++-only header: compiled with `gcc -std=c11 -pedantic -Wall -Wextra
 * -Werror` and linked against libfmd_hip.so by tests/test_c_abi.py.  It is the call sequence a cgo / Rust FFI
 * / JNI binding would make in place of Demod::new + Demod::demodulate (examples/simple_fm.rs:243,256):
 *   optimal_settings -> new -> demodulate (one DEFAULT_BUF_LENGTH block, twice) -> get_state -> free.
 * Usage: c_abi_smoke <iq.bin> <audio.s16> <state.txt> [port]   (the Python test compares both files with the oracle)
 * With a port the IQ bytes do not come from the file but from an rtl_tcp server on 127.0.0.1 (fmd_rtltcp_*: handshake,
 * the four commands of config_sdr -- examples/simple_fm.rs:217-229 --, two read_sync-shaped reads and the short read
 * at the end of the stream).
 * Exit code 0 on success, 2 when the library reports no usable device (the product has no CPU path).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fmd.h"

static int fail(const char *what, int rc)
{
    fprintf(stderr, "%s: %s (%d): %s\n", what, fmd_strerror(rc), rc, fmd_last_error());
    return rc == FMD_ERR_NO_DEVICE ? 2 : 1;
}

int main(int argc, char **argv)
{
    if (argc != 4 && argc != 5) { fprintf(stderr, "usage: %s iq.bin audio.s16 state.txt [rtl_tcp port]\n", argv[0]); return 1; }
    fmd_radio_config radio;
    fmd_demod_config cfg;
    int rc = fmd_optimal_settings(94900000u, 170000u, 32000u, &radio, &cfg);     /* simple_fm.rs:48 */
    if (rc != FMD_OK) return fail("fmd_optimal_settings", rc);
    if (cfg.downsample != 6u || radio.capture_rate != 1020000u) { fprintf(stderr, "optimal_settings mismatch\n"); return 1; }

    const size_t n = FMD_DEFAULT_BUF_LENGTH;
    uint8_t *iq = malloc(2 * n);
    if (!iq) return 1;
    if (argc == 5) {                                                              /* receive(), simple_fm.rs:89-132, over rtl_tcp */
        fmd_rtltcp *src = NULL;
        uint32_t tuner = 0, gains = 0;
        size_t got = 0;
        rc = fmd_rtltcp_open("127.0.0.1", (uint16_t)atoi(argv[4]), 5000u, &src);
        if (rc != FMD_OK) return fail("fmd_rtltcp_open", rc);
        if (fmd_rtltcp_info(src, &tuner, &gains) != FMD_OK || tuner != 5u || gains != 29u) { fprintf(stderr, "handshake mismatch\n"); return 1; }
        if (fmd_rtltcp_command(src, FMD_RTLTCP_SET_GAIN_MODE, 0u) != FMD_OK || fmd_rtltcp_command(src, FMD_RTLTCP_SET_BIAS_TEE, 0u) != FMD_OK ||
            fmd_rtltcp_command(src, FMD_RTLTCP_SET_FREQUENCY, radio.capture_freq) != FMD_OK ||
            fmd_rtltcp_command(src, FMD_RTLTCP_SET_SAMPLE_RATE, radio.capture_rate) != FMD_OK) return fail("fmd_rtltcp_command", FMD_ERR_IO);
        for (int call = 0; call < 2; ++call) {
            rc = fmd_rtltcp_read_sync(src, iq + (size_t)call * n, n, &got);
            if (rc != FMD_OK || got != n) { fprintf(stderr, "read_sync: rc %d, %lu bytes\n", rc, (unsigned long)got); return 1; }
        }
        uint8_t tail[64];
        rc = fmd_rtltcp_read_sync(src, tail, sizeof tail, &got);                  /* 40 bytes left: a short read, not an error */
        if (rc != FMD_OK || got != 40u) { fprintf(stderr, "short read: rc %d, %lu bytes\n", rc, (unsigned long)got); return 1; }
        fmd_rtltcp_close(src);
    } else {
        FILE *f = fopen(argv[1], "rb");
        if (!f || fread(iq, 1, 2 * n, f) != 2 * n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
        fclose(f);
    }

    fmd_device_config dev;
    memset(&dev, 0, sizeof dev);
    dev.n_channels = 1; dev.device_id = -1;
    fmd_demod *d = NULL;
    rc = fmd_demod_new(&cfg, &dev, &d);
    if (rc != FMD_OK) return fail("fmd_demod_new", rc);

    const size_t cap = fmd_out_cap(&cfg, n);
    int16_t *out = malloc(cap * sizeof *out);
    FILE *fo = fopen(argv[2], "wb");
    if (!out || !fo) return 1;
    for (int call = 0; call < 2; ++call) {
        size_t got = 0;
        rc = fmd_demod_demodulate(d, iq + (size_t)call * n, n, out, cap, &got);
        if (rc != FMD_OK) return fail("fmd_demod_demodulate", rc);
        if (fwrite(out, sizeof *out, got, fo) != got) return 1;                   /* output(), simple_fm.rs:430-438 */
    }
    fclose(fo);
    /* the reference panics on these; the C ABI returns codes */
    size_t got = 0;
    if (fmd_demod_demodulate(d, iq, 12, out, cap, &got) != FMD_ERR_BAD_LENGTH) { fprintf(stderr, "len %% 8 not rejected\n"); return 1; }
    if (fmd_demod_demodulate(d, iq, 8, out, cap, &got) != FMD_ERR_TOO_SHORT) { fprintf(stderr, "short buffer not rejected\n"); return 1; }

    fmd_demod_state st;
    rc = fmd_demod_get_state(d, 0u, &st);
    if (rc != FMD_OK) return fail("fmd_demod_get_state", rc);
    rc = fmd_demod_check(d);
    if (rc != FMD_OK) return fail("fmd_demod_check", rc);
    FILE *fs = fopen(argv[3], "w");
    if (!fs) return 1;
    fprintf(fs, "%u %d %d %d %d %d %d\n", (unsigned)st.prev_index, (int)st.now_lpr, (int)st.prev_lpr_index,
            (int)st.lp_now_re, (int)st.lp_now_im, (int)st.demod_pre_re, (int)st.demod_pre_im);
    fclose(fs);
    fmd_demod_free(d);
    free(out); free(iq);
    return 0;
}
