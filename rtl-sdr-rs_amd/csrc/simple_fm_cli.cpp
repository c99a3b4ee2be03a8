// simple_fm_cli.cpp -- file mode of the reference's simple_fm example on the GPU path.
//
// Mirrors main() of examples/simple_fm.rs with READ_FROM_FILE = true (:65-84): read DEFAULT_BUF_LENGTH
// (src/lib.rs:25) byte blocks of interleaved u8 IQ from a file (or stdin with "-"), demodulate, write raw
// s16 mono at the resample rate to stdout -- so the documented pipeline still works:
//     simple_fm_gpu capture.bin | play -r 32k -t raw -e s -b 16 -c 1 -V1 -      (readme.md:13,17)
// Defaults are the example's constants: FREQUENCY 94.9 MHz (unused here), SAMPLE_RATE 170 kHz, RATE_RESAMPLE 32 kHz
// (:25-27) through optimal_settings (:48,189-214).
//
// Several input files: one channel per file in ONE bank (one Demod per stream, :137), block-synchronous; audio of
// file k goes to <prefix>.<k>.s16 (-o prefix, default "audio") and the run ends with the shortest file.  With -g N the
// same goes through the pipelined multi-GPU sink (fmd_sink_*): channels split over N devices, byte-identical output.
//
// Live mode, -t host:port: the example with READ_FROM_FILE = false (:51-64, receive() :89-132, process() :135-170) with
// the dongle behind an rtl_tcp server (the reference's own examples/rtl_tcp.rs): config_sdr's settings (:217-229) go out
// as rtl_tcp commands, DEFAULT_BUF_LENGTH blocks come back through fm::RtlTcpSource::read_sync, a short read ends the run
// with the example's "samples lost" message (:122-125).
//
// EOF policy (the reference ignores the read count and never terminates at EOF): only COMPLETE
// blocks are demodulated; a trailing partial block is dropped with a note on stderr.  Logging goes to stderr
// because stdout carries audio (:37-38).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "demod.hpp"

// one channel per file through the pipelined sink (-g N): the channels are split over N GPUs (device k % visible), the
// file reads of block n+1 overlap the transfers and the kernel of block n
static int run_sink(const std::vector<const char*>& paths, const char* prefix, uint32_t freq, uint32_t rate, uint32_t resample, int gpus)
{
    const size_t C = paths.size(), N = fm::DEFAULT_BUF_LENGTH;
    std::vector<FILE*> in(C, nullptr), out(C, nullptr);
    int rc = 0;
    try {
        for (size_t c = 0; c < C; ++c) {
            if (!(in[c] = fopen(paths[c], "rb"))) { perror(paths[c]); throw 2; }
            char name[4096];
            snprintf(name, sizeof name, "%s.%zu.s16", prefix, c);
            if (!(out[c] = fopen(name, "wb"))) { perror(name); throw 2; }
        }
        int visible = 0;
        fm::check(fmd_device_count(&visible));
        if (visible < 1) throw fm::Error(FMD_ERR_NO_DEVICE);
        if ((size_t)gpus > C) gpus = (int)C;
        std::vector<int32_t> ids;
        for (int k = 0; k < gpus; ++k) ids.push_back(k % visible);
        const auto settings = fm::optimal_settings(freq, rate, resample);
        fm::Sink sink(settings.second, (uint32_t)C, N, ids, 3,
                      [&](uint64_t, uint32_t c, const int16_t* a, size_t n) { if (n) fwrite(a, sizeof(int16_t), n, out[c]); });
        size_t loops = 0;
        for (;; ++loops) {
            uint8_t* slot = sink.acquire();
            bool full = true;
            for (size_t c = 0; c < C && full; ++c) full = fread(slot + c * N, 1, N, in[c]) == N;
            if (!full) { sink.release(); break; }               // the shortest file ends the run; partial blocks dropped
            sink.submit();
        }
        sink.drain();
        fprintf(stderr, "%zu channels x %zu blocks on %d device part(s)\n", C, loops, gpus);
    } catch (const fm::Error& e) {
        fprintf(stderr, "error: %s\n", e.what());
        rc = 1;
    } catch (int code) {
        rc = code;
    }
    for (size_t c = 0; c < C; ++c) { if (in[c]) fclose(in[c]); if (out[c]) { fflush(out[c]); fclose(out[c]); } }
    return rc;
}

// one channel per file, all channels in one bank
static int run_bank(const std::vector<const char*>& paths, const char* prefix, uint32_t freq, uint32_t rate, uint32_t resample)
{
    const size_t C = paths.size(), N = fm::DEFAULT_BUF_LENGTH;
    std::vector<FILE*> in(C, nullptr), out(C, nullptr);
    int rc = 0;
    try {
        for (size_t c = 0; c < C; ++c) {
            if (!(in[c] = fopen(paths[c], "rb"))) { perror(paths[c]); throw 2; }
            char name[4096];
            snprintf(name, sizeof name, "%s.%zu.s16", prefix, c);
            if (!(out[c] = fopen(name, "wb"))) { perror(name); throw 2; }
        }
        const auto settings = fm::optimal_settings(freq, rate, resample);
        fm::DemodBank bank(settings.second, (uint32_t)C);
        std::vector<uint8_t> buf(C * N);
        size_t loops = 0;
        for (;; ++loops) {
            bool full = true;
            for (size_t c = 0; c < C && full; ++c) full = fread(buf.data() + c * N, 1, N, in[c]) == N;
            if (!full) break;                                   // the shortest file ends the run; partial blocks dropped
            const auto audio = bank.demodulate(buf.data(), N);
            for (size_t c = 0; c < C; ++c) fm::output(audio[c], out[c]);
        }
        fprintf(stderr, "%zu channels x %zu blocks\n", C, loops);
    } catch (const fm::Error& e) {
        fprintf(stderr, "error: %s\n", e.what());
        rc = 1;
    } catch (int code) {
        rc = code;
    }
    for (size_t c = 0; c < C; ++c) { if (in[c]) fclose(in[c]); if (out[c]) fclose(out[c]); }
    return rc;
}

// receive() + process() of the example over rtl_tcp (-t host:port)
static int run_rtl_tcp(const char* hostport, uint32_t freq, uint32_t rate, uint32_t resample, size_t max_blocks)
{
    // host, host:port, [v6-literal] or [v6-literal]:port (a bare IPv6 literal holds colons of its own); port 1 ... 65535
    std::string host(hostport), portstr;
    uint16_t port = 1234;                                    // rtl_tcp's default
    if (!host.empty() && host[0] == '[') {
        const size_t close = host.find(']');
        if (close == std::string::npos) { fprintf(stderr, "-t %s: missing ']'\n", hostport); return 2; }
        if (close + 1 < host.size()) {
            if (host[close + 1] != ':') { fprintf(stderr, "-t %s: expected ':port' after ']'\n", hostport); return 2; }
            portstr = host.substr(close + 2);
        }
        host = host.substr(1, close - 1);
    } else if (std::count(host.begin(), host.end(), ':') == 1) {
        const size_t colon = host.find(':');
        portstr = host.substr(colon + 1);
        host.resize(colon);
    }                                                        // (several colons without brackets: an IPv6 literal, default port)
    if (!portstr.empty()) {
        char* end = nullptr;
        const unsigned long v = strtoul(portstr.c_str(), &end, 10);
        if (*end != '\0' || v < 1 || v > 65535) { fprintf(stderr, "-t %s: port must be 1 ... 65535\n", hostport); return 2; }
        port = (uint16_t)v;
    }
    if (host.empty()) { fprintf(stderr, "-t %s: empty host\n", hostport); return 2; }
    try {
        const auto settings = fm::optimal_settings(freq, rate, resample);           // :48
        const fm::DemodConfig& dc = settings.second;
        fm::RtlTcpSource sdr(host, port);
        fprintf(stderr, "rtl_tcp %s:%u tuner type %u, %u gains\n", host.c_str(), (unsigned)port, sdr.tuner_type(), sdr.gain_count());
        sdr.set_tuner_gain_auto();                                                   // config_sdr, :217-229
        sdr.set_bias_tee(false);
        sdr.set_center_freq(settings.first.capture_freq);
        sdr.set_sample_rate(settings.first.capture_rate);
        fprintf(stderr, "Oversampling input by: %ux\n", dc.downsample);            // :138
        fprintf(stderr, "Output at %u Hz\n", dc.rate_in);                          // :139
        fm::Demod demod(dc);                                                         // :137
        std::vector<uint8_t> buf(fm::DEFAULT_BUF_LENGTH);
        size_t loops = 0;
        std::chrono::duration<double> total(0);
        while (max_blocks == 0 || loops < max_blocks) {
            const size_t n = sdr.read_sync(buf.data(), buf.size());                  // :116
            if (n < buf.size()) {                                                    // :122-125
                fprintf(stderr, "Short read (%zu bytes), samples lost, exiting!\n", n);
                break;
            }
            const auto t0 = std::chrono::steady_clock::now();
            const std::vector<int16_t> audio = demod.demodulate(buf);               // :153
            total += std::chrono::steady_clock::now() - t0;
            fm::output(audio);                                                       // :156
            ++loops;
        }
        if (loops) fprintf(stderr, "Average processing time: %.2fms (%zu loops)\n", 1e3 * total.count() / (double)loops, loops);   // :162-168
    } catch (const fm::Error& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv)
{
    uint32_t rate = 170000, resample = 32000, freq = 94900000;
    size_t per_launch = 1;                                   // -b: reference blocks handed to the GPU per launch
    int gpus = 0;                                            // -g N: several files through the pipelined sink on N GPUs
    const char* path = nullptr;
    const char* prefix = "audio";
    const char* rtl_tcp = nullptr;                           // -t host:port: live mode over rtl_tcp
    size_t max_blocks = 0;                                   // -n: stop after this many blocks (live mode; 0 = until the stream ends)
    std::vector<const char*> paths;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-s") && i + 1 < argc) rate = (uint32_t)strtoul(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "-r") && i + 1 < argc) resample = (uint32_t)strtoul(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "-f") && i + 1 < argc) freq = (uint32_t)strtoul(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) prefix = argv[++i];
        else if (!strcmp(argv[i], "-g") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) rtl_tcp = argv[++i];
        else if (!strcmp(argv[i], "-n") && i + 1 < argc) max_blocks = strtoul(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "-b") && i + 1 < argc) { per_launch = strtoul(argv[++i], nullptr, 10); if (!per_launch) per_launch = 1; }
        else if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
            fprintf(stderr, "usage: %s [-f freq_hz] [-s sample_rate_hz] [-r resample_hz] [-b blocks_per_launch] <capture.bin | ->\n"
                            "       %s [-s ...] [-r ...] [-o prefix] [-g n_gpus] <a.bin> <b.bin> ...   (one channel per file)\n"
                            "       %s [-f freq_hz] [-s ...] [-r ...] [-n blocks] -t host:port            (live: IQ from an rtl_tcp server)\n", argv[0], argv[0], argv[0]);
            return 0;
        } else paths.push_back(argv[i]);
    }
    if (rtl_tcp) return run_rtl_tcp(rtl_tcp, freq, rate, resample, max_blocks);
    if (paths.empty()) { fprintf(stderr, "missing input file (use - for stdin)\n"); return 2; }
    if (paths.size() > 1 && gpus > 0) return run_sink(paths, prefix, freq, rate, resample, gpus);
    if (paths.size() > 1) return run_bank(paths, prefix, freq, rate, resample);
    path = paths[0];
    FILE* in = strcmp(path, "-") ? fopen(path, "rb") : stdin;
    if (!in) { perror(path); return 2; }
    try {
        const auto settings = fm::optimal_settings(freq, rate, resample);
        const fm::DemodConfig& dc = settings.second;
        fprintf(stderr, "Oversampling input by: %ux\n", dc.downsample);             // simple_fm.rs:138
        fprintf(stderr, "Output at %u Hz\n", dc.rate_in);                           // :139
        fprintf(stderr, "Output scale: %u\n", dc.output_scale);                     // :140
        fprintf(stderr, "capture_rate: %u capture_freq: %u\n", settings.first.capture_rate, settings.first.capture_freq);
        fm::Demod demod(dc);
        // -b N: N blocks per launch with the result of N single calls (the f64 sample at every block start, :359)
        if (per_launch > 1) demod.set_block_len(fm::DEFAULT_BUF_LENGTH);
        std::vector<uint8_t> buf(fm::DEFAULT_BUF_LENGTH * per_launch);
        size_t fill = 0, loops = 0;
        std::chrono::duration<double> total(0);
        for (;;) {
            const size_t n = fread(buf.data() + fill, 1, buf.size() - fill, in);
            fill += n;
            if (fill < buf.size()) {
                if (n == 0) break;      // EOF (or error) before a complete block
                continue;
            }
            const auto t0 = std::chrono::steady_clock::now();
            const std::vector<int16_t> audio = demod.demodulate(buf);              // :80
            total += std::chrono::steady_clock::now() - t0;
            fm::output(audio);                                                      // :82
            loops += per_launch;
            fill = 0;
        }
        if (per_launch > 1 && fill >= fm::DEFAULT_BUF_LENGTH) {                     // complete blocks of a partly filled launch
            const size_t whole = fill / fm::DEFAULT_BUF_LENGTH * fm::DEFAULT_BUF_LENGTH;
            fm::output(demod.demodulate(buf.data(), whole));
            loops += whole / fm::DEFAULT_BUF_LENGTH;
            fill -= whole;
        }
        if (fill) fprintf(stderr, "dropped %zu trailing bytes (not a complete %zu-byte block)\n", fill, (size_t)fm::DEFAULT_BUF_LENGTH);
        if (loops)                                                                  // :162-168
            fprintf(stderr, "Average processing time: %.2fms (%zu loops)\n", 1e3 * total.count() / (double)loops, loops);
    } catch (const fm::Error& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
