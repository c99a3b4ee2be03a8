// rtltcp_harness.cpp -- drives the rtl_tcp client of the C ABI (fmd_rtltcp_*, rtl-sdr-rs_amd/csrc/fmd_rtltcp.cpp) from a
// plain host program so that the client can be built with -fsanitize=address,undefined and pointed at hostile servers
// (tests/test_sanitizers.py).  TEST INFRASTRUCTURE: it supplies the one symbol the client takes from the library proper
// (the error-text sink of fmd_last_error) and prints what every call returned, one line each:
//   open <status> [tuner gains]      read <status> <n_read> <sum of the bytes>      cmd <status>
// usage: rtltcp_harness <port> <timeout_ms> <nbytes per read> <reads> [<opcode> <param>]...
//        rtltcp_harness many <timeout_ms> <nbytes per row> <reads> <port>...      (fmd_rtltcp_read_many over all the ports:
//                       many <status> <rows full> <rows short> <sum of all bytes read>)
//        rtltcp_harness nval <timeout_ms> <nbytes per row> <port> <port>       (two sources; the second one's descriptor is closed
//                       behind the library's back before fmd_rtltcp_read_many: poll() reports POLLNVAL -> nval <status>)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <unistd.h>

#include "../../include/fmd.h"

static std::string g_err;
void fmd_internal_set_err(const char* msg) { g_err = msg ? msg : ""; }

static int many(int argc, char** argv)
{
    const uint32_t timeout_ms = (uint32_t)strtoul(argv[2], nullptr, 10);
    const size_t nbytes = (size_t)strtoul(argv[3], nullptr, 10);
    const int reads = atoi(argv[4]);
    const uint32_t n = (uint32_t)(argc - 5);
    std::vector<fmd_rtltcp*> src(n, nullptr);
    for (uint32_t c = 0; c < n; ++c)
        if (fmd_rtltcp_open("127.0.0.1", (uint16_t)atoi(argv[5 + c]), timeout_ms, &src[c]) != FMD_OK) { printf("open %u failed %s\n", c, g_err.c_str()); return 0; }
    std::vector<uint8_t> buf((size_t)n * nbytes);           // rows exactly nbytes apart: a row that overran would hit its neighbour / the end
    std::vector<size_t> got(n);
    for (int r = 0; r < reads; ++r) {
        const int st = fmd_rtltcp_read_many(src.data(), n, buf.data(), nbytes, nbytes, got.data());
        unsigned long sum = 0; uint32_t full = 0, shorts = 0;
        for (uint32_t c = 0; c < n; ++c) {
            for (size_t k = 0; k < got[c]; ++k) sum += buf[(size_t)c * nbytes + k];
            if (got[c] == nbytes) ++full; else ++shorts;
        }
        printf("many %d %u %u %lu\n", st, full, shorts, sum);
        if (st != FMD_OK || shorts) break;
    }
    for (uint32_t c = 0; c < n; ++c) fmd_rtltcp_close(src[c]);
    return 0;
}

static int nval(int argc, char** argv)
{
    if (argc < 6) return 2;
    const uint32_t timeout_ms = (uint32_t)strtoul(argv[2], nullptr, 10);
    const size_t nbytes = (size_t)strtoul(argv[3], nullptr, 10);
    fmd_rtltcp* src[2] = {nullptr, nullptr};
    if (fmd_rtltcp_open("127.0.0.1", (uint16_t)atoi(argv[4]), timeout_ms, &src[0]) != FMD_OK) { printf("open 0 failed %s\n", g_err.c_str()); return 0; }
    const int probe = dup(0);                               // the lowest free descriptor: the one the next socket() will get
    close(probe);
    if (fmd_rtltcp_open("127.0.0.1", (uint16_t)atoi(argv[5]), timeout_ms, &src[1]) != FMD_OK) { printf("open 1 failed %s\n", g_err.c_str()); return 0; }
    close(probe);                                           // the second source's socket, closed behind the library's back
    std::vector<uint8_t> buf(2 * nbytes);
    size_t got[2] = {0, 0};
    const int st = fmd_rtltcp_read_many(src, 2, buf.data(), nbytes, nbytes, got);
    printf("nval %d %zu %s\n", st, got[1], g_err.c_str());
    fmd_rtltcp_close(src[0]);
    fmd_rtltcp_close(src[1]);                                // (close() of an already closed descriptor: EBADF, harmless)
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    if (std::string(argv[1]) == "many") return many(argc, argv);
    if (std::string(argv[1]) == "nval") return nval(argc, argv);
    const uint16_t port = (uint16_t)atoi(argv[1]);
    const uint32_t timeout_ms = (uint32_t)strtoul(argv[2], nullptr, 10);
    const size_t nbytes = (size_t)strtoul(argv[3], nullptr, 10);
    const int reads = atoi(argv[4]);
    fmd_rtltcp* s = nullptr;
    const int rc = fmd_rtltcp_open("127.0.0.1", port, timeout_ms, &s);
    if (rc != FMD_OK) { printf("open %d %s\n", rc, g_err.c_str()); return s ? 3 : 0; }   // a failed open must leave *out null
    uint32_t tt = 0, gc = 0;
    (void)fmd_rtltcp_info(s, &tt, &gc);
    printf("open 0 %u %u\n", tt, gc);
    for (int i = 5; i + 1 < argc; i += 2)
        printf("cmd %d\n", fmd_rtltcp_command(s, (uint8_t)strtoul(argv[i], nullptr, 0), (uint32_t)strtoul(argv[i + 1], nullptr, 0)));
    // the buffer is exactly nbytes long: a client that wrote one byte too many would trip AddressSanitizer here
    std::vector<uint8_t> buf(nbytes ? nbytes : 1);
    for (int r = 0; r < reads; ++r) {
        size_t n = (size_t)-1;
        const int st = fmd_rtltcp_read_sync(s, buf.data(), nbytes, &n);
        unsigned long sum = 0;
        for (size_t k = 0; k < n && k < nbytes; ++k) sum += buf[k];
        printf("read %d %zu %lu\n", st, n, sum);
        if (st != FMD_OK || n < nbytes) break;               // error or end of stream ("samples lost", simple_fm.rs:122)
    }
    fmd_rtltcp_close(s);
    fmd_rtltcp_close(nullptr);                               // documented no-op
    return 0;
}
