// demod.hpp -- C++ host-side mirror of the reference's interface for this path, over the C ABI.
//
// The reference is Rust (ccostes/rtl-sdr-rs v0.3.1); this image has no rustc, so the host side above
// include/fmd.h is C++ with the reference's names and argument meaning (examples/simple_fm.rs):
//   optimal_settings(freq, rate)              :189-214  -> std::pair<RadioConfig, DemodConfig>
//   Demod::new(config)                        :243-252  -> fm::Demod(config)
//   Demod::demodulate(&mut self, Vec<u8>)     :256-269  -> fm::Demod::demodulate(const std::vector<uint8_t>&)
//   output(Vec<i16>)                          :430-438  -> fm::output(const std::vector<int16_t>&, FILE*)
// Where the reference panics (len % 8, < 2 decimated samples, rate_out < rate_resample) these throw
// fm::Error carrying the fmd_status.  Header-only; link with libfmd_hip.so.
#pragma once

#include <cstdint>
#include <cstdio>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fmd.h"

namespace fm {

using RadioConfig = fmd_radio_config;   // simple_fm.rs:173-176
using DemodConfig = fmd_demod_config;   // simple_fm.rs:179-185

constexpr size_t DEFAULT_BUF_LENGTH = FMD_DEFAULT_BUF_LENGTH;   // src/lib.rs:25

struct Error : std::runtime_error {
    int status;
    explicit Error(int s)
        : std::runtime_error(std::string(fmd_strerror(s)) + ": " + fmd_last_error()), status(s) {}
};

inline void check(int status) { if (status != FMD_OK) throw Error(status); }

// optimal_settings(freq, rate), simple_fm.rs:189-214 (rate_resample = RATE_RESAMPLE, :27).
inline std::pair<RadioConfig, DemodConfig> optimal_settings(uint32_t freq, uint32_t rate, uint32_t rate_resample = 32000)
{
    RadioConfig r{};
    DemodConfig d{};
    check(fmd_optimal_settings(freq, rate, rate_resample, &r, &d));
    return {r, d};
}

// struct Demod + impl, simple_fm.rs:232-269: one IQ stream, state carried across calls.
class Demod {
public:
    explicit Demod(const DemodConfig& config, int device_id = -1) : config_(config)
    {
        fmd_device_config dev{1u, device_id, 0u};
        check(fmd_demod_new(&config_, &dev, &h_));
    }
    ~Demod() { fmd_demod_free(h_); }
    Demod(const Demod&) = delete;
    Demod& operator=(const Demod&) = delete;

    // demodulate(&mut self, buf: Vec<u8>) -> Vec<i16>
    std::vector<int16_t> demodulate(const std::vector<uint8_t>& buf) { return demodulate(buf.data(), buf.size()); }
    std::vector<int16_t> demodulate(const uint8_t* buf, size_t len)
    {
        std::vector<int16_t> out(fmd_out_cap(&config_, len) + 1);
        size_t n = 0;
        check(fmd_demod_demodulate(h_, buf, len, out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }

    fmd_demod_state state()
    {
        fmd_demod_state s{};
        check(fmd_demod_get_state(h_, 0, &s));
        return s;
    }
    void set_state(const fmd_demod_state& s) { check(fmd_demod_set_state(h_, 0, &s)); }
    // treat every buffer as consecutive reference calls of block_bytes each (0 = off); see fmd_demod_set_block_len
    void set_block_len(size_t block_bytes) { check(fmd_demod_set_block_len(h_, block_bytes)); }
    const DemodConfig& config() const { return config_; }

private:
    DemodConfig config_;
    fmd_demod* h_ = nullptr;
};

// n independent Demods on one GPU behind one handle (one `Demod` per stream, simple_fm.rs:137): `iq` holds
// n_channels equal-sized read_sync buffers back to back.
class DemodBank {
public:
    DemodBank(const DemodConfig& config, uint32_t n_channels, int device_id = -1) : config_(config), n_(n_channels)
    {
        fmd_device_config dev{n_channels, device_id, 0u};
        check(fmd_demod_new(&config_, &dev, &h_));
    }
    ~DemodBank() { fmd_demod_free(h_); }
    DemodBank(const DemodBank&) = delete;
    DemodBank& operator=(const DemodBank&) = delete;

    // out[c] = Demod::demodulate(channel c's buffer); `len` bytes per channel
    std::vector<std::vector<int16_t>> demodulate(const uint8_t* iq, size_t len)
    {
        const size_t cap = fmd_out_cap(&config_, len) + 1;
        std::vector<int16_t> flat(cap * n_);
        std::vector<size_t> lens(n_);
        check(fmd_demod_demodulate_batch(h_, iq, len, flat.data(), cap, lens.data()));
        std::vector<std::vector<int16_t>> out(n_);
        for (uint32_t c = 0; c < n_; ++c) out[c].assign(flat.begin() + c * cap, flat.begin() + c * cap + lens[c]);
        return out;
    }
    uint32_t channels() const { return n_; }
    const DemodConfig& config() const { return config_; }

private:
    DemodConfig config_;
    uint32_t n_;
    fmd_demod* h_ = nullptr;
};

// The receive -> mpsc -> process -> output hand-off of the example (simple_fm.rs:55-60,114-127,150-156) with the GPU(s)
// as consumer: fmd_sink_* (new surface, see include/fmd.h).  `on_audio(seq, channel, samples, n)` is output() per
// channel, called in submission order from inside acquire() / drain() on the caller's thread.
class Sink {
public:
    using Callback = std::function<void(uint64_t seq, uint32_t channel, const int16_t* samples, size_t n)>;

    Sink(const DemodConfig& config, uint32_t n_channels, size_t nbytes, const std::vector<int32_t>& device_ids,
         uint32_t depth, Callback on_audio)
        : n_(n_channels), nbytes_(nbytes), cb_(std::move(on_audio))
    {
        check(fmd_sink_new(&config, n_channels, device_ids.data(), (uint32_t)device_ids.size(), nbytes, depth, &Sink::trampoline,
                           this, &h_));
    }
    ~Sink() { fmd_sink_free(h_); }
    Sink(const Sink&) = delete;
    Sink& operator=(const Sink&) = delete;

    // the next slot to fill: n_channels buffers of nbytes back to back (page-locked; read_sync writes straight into it)
    uint8_t* acquire() { uint8_t* p = nullptr; check(fmd_sink_acquire(h_, &p)); return p; }
    void submit() { check(fmd_sink_submit(h_)); }
    void release() { check(fmd_sink_release(h_)); }      // give the acquired slot back unsubmitted (short read, simple_fm.rs:122-125)
    void drain() { check(fmd_sink_drain(h_)); if (status_ != FMD_OK) throw Error(status_); }
    uint32_t channels() const { return n_; }
    size_t nbytes() const { return nbytes_; }

private:
    static void trampoline(void* user, uint64_t seq, const int16_t* audio, const size_t* out_len, size_t out_cap, int status)
    {
        Sink* self = static_cast<Sink*>(user);
        if (status != FMD_OK) { if (self->status_ == FMD_OK) self->status_ = status; return; }
        for (uint32_t c = 0; c < self->n_; ++c) self->cb_(seq, c, audio + (size_t)c * out_cap, out_len[c]);
    }
    uint32_t n_;
    size_t nbytes_;
    Callback cb_;
    int status_ = FMD_OK;
    fmd_sink* h_ = nullptr;
};

// An rtl_tcp server (the reference's examples/rtl_tcp.rs) as the producer: `read_sync` has the shape of
// RtlSdr::read_sync (src/lib.rs:153) -- bytes written, fewer than asked = the stream ended ("samples lost",
// simple_fm.rs:122).  Wire format and opcodes: include/fmd.h, fmd_rtltcp_*.
class RtlTcpSource {
public:
    RtlTcpSource(const std::string& host, uint16_t port, uint32_t timeout_ms = 10000)
    {
        check(fmd_rtltcp_open(host.c_str(), port, timeout_ms, &h_));
        check(fmd_rtltcp_info(h_, &tuner_type_, &gain_count_));
    }
    ~RtlTcpSource() { fmd_rtltcp_close(h_); }
    RtlTcpSource(const RtlTcpSource&) = delete;
    RtlTcpSource& operator=(const RtlTcpSource&) = delete;

    size_t read_sync(uint8_t* buf, size_t nbytes) { size_t n = 0; check(fmd_rtltcp_read_sync(h_, buf, nbytes, &n)); return n; }
    void command(uint8_t opcode, uint32_t param) { check(fmd_rtltcp_command(h_, opcode, param)); }
    // config_sdr of the example (simple_fm.rs:217-229), by name
    void set_tuner_gain_auto() { command(FMD_RTLTCP_SET_GAIN_MODE, 0); }
    void set_bias_tee(bool on) { command(FMD_RTLTCP_SET_BIAS_TEE, on ? 1u : 0u); }
    void set_center_freq(uint32_t hz) { command(FMD_RTLTCP_SET_FREQUENCY, hz); }
    void set_sample_rate(uint32_t hz) { command(FMD_RTLTCP_SET_SAMPLE_RATE, hz); }
    uint32_t tuner_type() const { return tuner_type_; }
    uint32_t gain_count() const { return gain_count_; }

private:
    fmd_rtltcp* h_ = nullptr;
    uint32_t tuner_type_ = 0, gain_count_ = 0;
};

// output(buf: Vec<i16>), simple_fm.rs:430-438: raw native-endian s16 to stdout, flushed.
inline void output(const std::vector<int16_t>& buf, FILE* f = stdout)
{
    if (!buf.empty()) fwrite(buf.data(), sizeof(int16_t), buf.size(), f);
    fflush(f);
}

}  // namespace fm
