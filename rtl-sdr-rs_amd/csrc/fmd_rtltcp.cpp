// fmd_rtltcp.cpp -- client side of the rtl_tcp wire protocol behind the C ABI (include/fmd.h, fmd_rtltcp_*).
//
// The reference ships the SERVER (examples/rtl_tcp.rs): after accept it sends a 12-byte handshake -- "RTL0", tuner type
// (u32 big-endian), tuner gain count (u32 big-endian) (send_handshake, :691-697) --, then streams the raw interleaved u8
// IQ exactly as RtlSdr::read_sync delivered it (sender_loop, :609-631) and accepts 5-byte commands: one opcode byte + a
// big-endian u32 / i32 parameter (command_loop, :633-689, opcodes 0x01 ... 0x0e).  This is the matching client, shaped
// like RtlSdr::read_sync (src/lib.rs:153): fill the caller's buffer, report the bytes written -- fewer than asked means
// the stream ended, which every caller of the reference treats as "samples lost" (examples/simple_fm.rs:122).  A dongle
// on another host thereby feeds the GPU sink with no USB code here.  Host code only: sockets and bytes, no arithmetic,
// no HIP header (it works on a box without a GPU, and builds with plain g++ -fsanitize=address,undefined:
// tests/test_sanitizers.py runs it against hostile servers that way).
#include "../../include/fmd.h"

#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>

#include <climits>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

void fmd_internal_set_err(const char* msg);                  // thread-local text behind fmd_last_error() (fmd_api.cpp)

struct fmd_rtltcp {
    int fd = -1;
    uint32_t tuner_type = 0, gain_count = 0;
    int timeout_ms = 10000;
};

namespace {

void err(const char* what)
{
    char m[256];
    snprintf(m, sizeof m, "rtl_tcp: %s: %s", what, strerror(errno));
    fmd_internal_set_err(m);
}

// Wait until `fd` is readable / writable or the timeout expires.  1 ready, 0 timeout, -1 error.
int wait_fd(int fd, short events, int timeout_ms)
{
    struct pollfd p;
    p.fd = fd; p.events = events; p.revents = 0;
    for (;;) {
        const int r = poll(&p, 1, timeout_ms);
        if (r < 0 && errno == EINTR) continue;
        return r < 0 ? -1 : (r == 0 ? 0 : 1);
    }
}

// Up to n bytes; stops early at end of stream.  Returns bytes read, or -1 on error / timeout (then *partial, when
// given, holds the bytes that did arrive, so a caller can keep the I/Q byte alignment of the stream).
long read_upto(fmd_rtltcp* s, uint8_t* buf, size_t n, size_t* partial = nullptr)
{
    size_t got = 0;
    struct Keep { size_t* p; const size_t& g; ~Keep() { if (p) *p = g; } } keep{partial, got};
    while (got < n) {
        const int w = wait_fd(s->fd, POLLIN, s->timeout_ms);
        if (w == 0) { errno = ETIMEDOUT; return -1; }
        if (w < 0) return -1;
        const ssize_t r = recv(s->fd, buf + got, n - got, 0);
        if (r < 0) { if (errno == EINTR || errno == EAGAIN) continue; return -1; }
        if (r == 0) break;                                   // orderly end of stream
        got += (size_t)r;
    }
    return (long)got;
}

}  // namespace

extern "C" {

int fmd_rtltcp_open(const char* host, uint16_t port, uint32_t timeout_ms, fmd_rtltcp** out)
{
    if (!host || !out) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    *out = nullptr;
    char portstr[16];
    snprintf(portstr, sizeof portstr, "%u", (unsigned)port);
    struct addrinfo hints;
    memset(&hints, 0, sizeof hints);
    hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM;
    struct addrinfo* res = nullptr;
    const int gai = getaddrinfo(host, portstr, &hints, &res);
    if (gai != 0 || !res) {
        char m[256];
        snprintf(m, sizeof m, "rtl_tcp: cannot resolve %s: %s", host, gai_strerror(gai));
        fmd_internal_set_err(m);
        return FMD_ERR_IO;
    }
    fmd_rtltcp* s = new (std::nothrow) fmd_rtltcp();
    if (!s) { freeaddrinfo(res); return FMD_ERR_NOMEM; }
    s->timeout_ms = timeout_ms == 0 ? 10000 : (timeout_ms > (uint32_t)INT_MAX ? INT_MAX : (int)timeout_ms);   // (a negative poll() timeout waits forever)
    int last_errno = ECONNREFUSED;
    for (struct addrinfo* a = res; a && s->fd < 0; a = a->ai_next) {
        const int fd = socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (fd < 0) { last_errno = errno; continue; }
        const int fl = fcntl(fd, F_GETFL, 0);
        (void)fcntl(fd, F_SETFL, fl | O_NONBLOCK);           // connect with a timeout
        int rc = connect(fd, a->ai_addr, a->ai_addrlen);
        if (rc < 0 && errno == EINPROGRESS) {
            rc = -1;
            if (wait_fd(fd, POLLOUT, s->timeout_ms) == 1) {
                int soerr = 0; socklen_t sl = sizeof soerr;
                if (getsockopt(fd, SOL_SOCKET, SO_ERROR, &soerr, &sl) == 0 && soerr == 0) rc = 0; else errno = soerr ? soerr : errno;
            } else errno = ETIMEDOUT;
        }
        if (rc < 0) { last_errno = errno; close(fd); continue; }
        (void)fcntl(fd, F_SETFL, fl);                        // blocking again: reads wait in poll()
        const int one = 1;
        (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        s->fd = fd;
    }
    freeaddrinfo(res);
    if (s->fd < 0) { errno = last_errno; err("connect"); delete s; return FMD_ERR_IO; }
    uint8_t hs[12];                                          // send_handshake, examples/rtl_tcp.rs:691-697
    const long n = read_upto(s, hs, sizeof hs);
    if (n != (long)sizeof hs || memcmp(hs, "RTL0", 4) != 0) {
        if (n < 0) err("handshake"); else fmd_internal_set_err("rtl_tcp: not an rtl_tcp handshake (need 12 bytes starting with \"RTL0\")");
        fmd_rtltcp_close(s);
        return FMD_ERR_IO;
    }
    s->tuner_type = ((uint32_t)hs[4] << 24) | ((uint32_t)hs[5] << 16) | ((uint32_t)hs[6] << 8) | hs[7];
    s->gain_count = ((uint32_t)hs[8] << 24) | ((uint32_t)hs[9] << 16) | ((uint32_t)hs[10] << 8) | hs[11];
    *out = s;
    return FMD_OK;
}

void fmd_rtltcp_close(fmd_rtltcp* s)
{
    if (!s) return;
    if (s->fd >= 0) close(s->fd);
    delete s;
}

int fmd_rtltcp_info(const fmd_rtltcp* s, uint32_t* tuner_type, uint32_t* gain_count)
{
    if (!s) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    if (tuner_type) *tuner_type = s->tuner_type;
    if (gain_count) *gain_count = s->gain_count;
    return FMD_OK;
}

int fmd_rtltcp_read_sync(fmd_rtltcp* s, uint8_t* buf, size_t nbytes, size_t* n_read)
{
    if (!s || !buf || !n_read) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    *n_read = 0;
    const long n = read_upto(s, buf, nbytes, n_read);       // on error *n_read = the bytes that did arrive
    if (n < 0) { err("read"); return FMD_ERR_IO; }
    *n_read = (size_t)n;                                     // < nbytes: the stream ended ("samples lost", simple_fm.rs:122)
    return FMD_OK;
}

// receive() of the example (simple_fm.rs:100-132) for MANY streams at once: row c of a [n][row_stride] buffer is filled from
// sources[c], all sockets behind ONE poll() -- every source is read as far as its bytes have arrived, so n slow streams cost
// one wait, not n (n sequential blocking read_sync calls serialise the network latency of every stream).
int fmd_rtltcp_read_many(fmd_rtltcp* const* sources, uint32_t n, uint8_t* base, size_t row_stride, size_t nbytes, size_t* n_read)
{
    if (!sources || !base || !n_read || n == 0 || row_stride < nbytes) { fmd_internal_set_err("bad argument"); return FMD_ERR_INVALID_ARG; }
    int timeout_ms = INT_MAX;
    for (uint32_t c = 0; c < n; ++c) {
        if (!sources[c]) { fmd_internal_set_err("null source"); return FMD_ERR_INVALID_ARG; }
        n_read[c] = 0;
        if (sources[c]->timeout_ms < timeout_ms) timeout_ms = sources[c]->timeout_ms;
    }
    if (nbytes == 0) return FMD_OK;
    std::vector<struct pollfd> pf(n);
    std::vector<uint32_t> who(n);
    std::vector<uint8_t> done(n, 0);                         // 1: row full, 2: stream ended early
    uint32_t open_rows = n;
    while (open_rows) {
        nfds_t m = 0;
        for (uint32_t c = 0; c < n; ++c)
            if (!done[c]) { pf[m].fd = sources[c]->fd; pf[m].events = POLLIN; pf[m].revents = 0; who[m] = c; ++m; }
        const int r = poll(pf.data(), m, timeout_ms);
        if (r < 0) { if (errno == EINTR) continue; err("poll"); return FMD_ERR_IO; }
        if (r == 0) { errno = ETIMEDOUT; err("read"); return FMD_ERR_IO; }   // no byte on ANY open stream for a whole timeout
        for (nfds_t k = 0; k < m; ++k) {
            // POLLNVAL (the descriptor is not open: a source closed behind the library's back) makes poll() return at once on
            // every pass without ever delivering a byte or an end of stream -- without this test the loop would spin forever
            if (pf[k].revents & POLLNVAL) { errno = EBADF; err("read"); return FMD_ERR_IO; }
            if (!(pf[k].revents & (POLLIN | POLLHUP | POLLERR))) continue;
            const uint32_t c = who[k];
            const ssize_t g = recv(pf[k].fd, base + (size_t)c * row_stride + n_read[c], nbytes - n_read[c], MSG_DONTWAIT);
            if (g < 0) { if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue; err("read"); return FMD_ERR_IO; }
            if (g == 0) { done[c] = 2; --open_rows; continue; }          // orderly end of this stream: "samples lost" for the caller
            n_read[c] += (size_t)g;
            if (n_read[c] == nbytes) { done[c] = 1; --open_rows; }
        }
    }
    return FMD_OK;
}

int fmd_rtltcp_command(fmd_rtltcp* s, uint8_t opcode, uint32_t param)
{
    if (!s) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    const uint8_t msg[5] = {opcode, (uint8_t)(param >> 24), (uint8_t)(param >> 16), (uint8_t)(param >> 8), (uint8_t)param};   // :653-657
    size_t sent = 0;
    while (sent < sizeof msg) {                              // the timeout covers a server that stopped reading, too
        const int w = wait_fd(s->fd, POLLOUT, s->timeout_ms);
        if (w <= 0) { if (w == 0) errno = ETIMEDOUT; err("command"); return FMD_ERR_IO; }
        const ssize_t r = send(s->fd, msg + sent, sizeof msg - sent, MSG_NOSIGNAL | MSG_DONTWAIT);
        if (r < 0) { if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue; err("command"); return FMD_ERR_IO; }
        sent += (size_t)r;
    }
    return FMD_OK;
}

}  // extern "C"
