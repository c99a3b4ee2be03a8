//! Thin `extern "C"` binding + safe wrapper over `include/fmd.h` (libfmd_hip.so).
//!
//! NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no rustc/cargo.  It is the
//! binding a maintainer of ccostes/rtl-sdr-rs would add so that `examples/simple_fm.rs` can swap its CPU
//! `Demod` (examples/simple_fm.rs:232-427) for the GPU one without touching `receive()`/`process()`:
//! the buffers are exactly what `RtlSdr::read_sync` (src/lib.rs:153) fills.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

/// `struct DemodConfig`, examples/simple_fm.rs:179-185 (same field order as `fmd_demod_config`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct DemodConfig {
    pub rate_in: u32,
    pub rate_out: u32,
    pub rate_resample: u32,
    pub downsample: u32,
    pub output_scale: u32,
}

/// `struct RadioConfig`, examples/simple_fm.rs:173-176.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RadioConfig {
    pub capture_freq: u32,
    pub capture_rate: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct DeviceConfig {
    pub n_channels: u32,
    pub device_id: i32,
    pub flags: u32,
}

/// Mutable fields of `struct Demod`, examples/simple_fm.rs:232-239.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct DemodState {
    pub prev_index: u32,
    pub now_lpr: i32,
    pub prev_lpr_index: i32,
    pub lp_now_re: i32,
    pub lp_now_im: i32,
    pub demod_pre_re: i32,
    pub demod_pre_im: i32,
}

/// `fmd_synth_params`: the deterministic integer-only FM source that stands in for the absent capture.bin.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SynthParams {
    pub seed: u64,
    pub amplitude: u32,
    pub noise: u32,
    pub dev_q32: u32,
    pub mod_period: u32,
}

#[repr(C)]
pub struct fmd_demod {
    _private: [u8; 0],
}

#[repr(C)]
pub struct fmd_fir {
    _private: [u8; 0],
}

#[repr(C)]
pub struct fmd_firdemod {
    _private: [u8; 0],
}

#[repr(C)]
pub struct fmd_sink {
    _private: [u8; 0],
}

#[repr(C)]
pub struct fmd_rtltcp {
    _private: [u8; 0],
}

/// `fmd_sink_callback`: (user, seq, audio [n_channels][out_cap], out_len [n_channels], out_cap, status).
pub type fmd_sink_callback = Option<unsafe extern "C" fn(*mut c_void, u64, *const i16, *const usize, usize, c_int)>;

extern "C" {
    pub fn fmd_optimal_settings(freq: u32, rate: u32, rate_resample: u32, radio: *mut RadioConfig, demod: *mut DemodConfig) -> c_int;
    pub fn fmd_demod_new(config: *const DemodConfig, dev: *const DeviceConfig, out: *mut *mut fmd_demod) -> c_int;
    pub fn fmd_demod_free(d: *mut fmd_demod);
    pub fn fmd_demod_reset(d: *mut fmd_demod) -> c_int;
    pub fn fmd_demod_demodulate(d: *mut fmd_demod, iq: *const u8, nbytes: usize, out: *mut i16, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn fmd_demod_demodulate_batch(d: *mut fmd_demod, iq: *const u8, nbytes: usize, out: *mut i16, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn fmd_demod_demodulate_device(d: *mut fmd_demod, d_iq: *const c_void, nbytes: usize, d_out: *mut c_void, out_cap: usize, d_out_len: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn fmd_demod_set_block_len(d: *mut fmd_demod, block_bytes: usize) -> c_int;
    pub fn fmd_demod_check(d: *mut fmd_demod) -> c_int;
    pub fn fmd_demod_check_prev(d: *mut fmd_demod) -> c_int;
    pub fn fmd_demod_check_behind(d: *mut fmd_demod, back: u32) -> c_int;
    pub fn fmd_demod_set_event_ordering(d: *mut fmd_demod, on: c_int) -> c_int;
    pub fn fmd_demod_f64_stats(d: *const fmd_demod, guarded: *mut u64, patched: *mut u64) -> c_int;
    pub fn fmd_demod_last_out_len(d: *const fmd_demod, out_len: *mut usize) -> c_int;
    pub fn fmd_host_alloc(nbytes: usize, ptr: *mut *mut c_void) -> c_int;
    pub fn fmd_host_free(ptr: *mut c_void) -> c_int;
    pub fn fmd_out_cap(config: *const DemodConfig, nbytes: usize) -> usize;
    pub fn fmd_demod_get_state(d: *mut fmd_demod, channel: u32, state: *mut DemodState) -> c_int;
    pub fn fmd_demod_set_state(d: *mut fmd_demod, channel: u32, state: *const DemodState) -> c_int;
    pub fn fmd_strerror(status: c_int) -> *const c_char;
    pub fn fmd_last_error() -> *const c_char;
    pub fn fmd_device_count(count: *mut c_int) -> c_int;
    pub fn fmd_version() -> c_int;
    pub fn fmd_demod_tiling(d: *const fmd_demod, audio_per_tile: *mut u32, lds_bytes: *mut u32, block_threads: *mut u32) -> c_int;
    pub fn fmd_demod_tiling_plan(d: *const fmd_demod) -> c_int;
    pub fn fmd_demod_last_kernel(d: *const fmd_demod, name: *mut c_char, cap: usize) -> c_int;
    pub fn fmd_demod_set_tiling(d: *mut fmd_demod, audio_per_tile: u32) -> c_int;
    pub fn fmd_synth_fill_device(device_id: c_int, d_iq: *mut c_void, n_channels: u32, nbytes: usize, sample_offset: u64, p: *const SynthParams, stream: *mut c_void) -> c_int;
    pub fn fmd_fir_new(taps: *const i16, n_taps: u32, decim: u32, dev: *const DeviceConfig, out: *mut *mut fmd_fir) -> c_int;
    pub fn fmd_fir_free(f: *mut fmd_fir);
    pub fn fmd_fir_reset(f: *mut fmd_fir) -> c_int;
    pub fn fmd_fir_tap_digits(f: *const fmd_fir) -> c_int;
    pub fn fmd_fir_out_cap(n_taps: u32, decim: u32, nbytes: usize) -> usize;
    pub fn fmd_fir_filter_batch(f: *mut fmd_fir, iq: *const u8, nbytes: usize, out: *mut i32, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn fmd_fir_filter_device(f: *mut fmd_fir, d_iq: *const c_void, nbytes: usize, d_out: *mut c_void, out_cap: usize, out_len_each: *mut usize, stream: *mut c_void) -> c_int;
    pub fn fmd_firdemod_new(taps: *const i16, n_taps: u32, decim: u32, shift: u32, rate_out: u32, rate_resample: u32, dev: *const DeviceConfig, out: *mut *mut fmd_firdemod) -> c_int;
    pub fn fmd_firdemod_free(f: *mut fmd_firdemod);
    pub fn fmd_firdemod_reset(f: *mut fmd_firdemod) -> c_int;
    pub fn fmd_firdemod_out_cap(decim: u32, rate_out: u32, rate_resample: u32, nbytes: usize) -> usize;
    pub fn fmd_firdemod_demodulate_batch(f: *mut fmd_firdemod, iq: *const u8, nbytes: usize, out: *mut i16, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn fmd_firdemod_demodulate_device(f: *mut fmd_firdemod, d_iq: *const c_void, nbytes: usize, d_out: *mut c_void, out_cap: usize, out_len_each: *mut usize, stream: *mut c_void) -> c_int;
    pub fn fmd_firdemod_check(f: *mut fmd_firdemod) -> c_int;
    pub fn fmd_firdemod_get_state(f: *mut fmd_firdemod, channel: u32, state: *mut DemodState) -> c_int;
    pub fn fmd_firdemod_checkpoint_size(f: *const fmd_firdemod) -> usize;
    pub fn fmd_firdemod_checkpoint(f: *mut fmd_firdemod, blob: *mut c_void, cap: usize) -> c_int;
    pub fn fmd_firdemod_resume(f: *mut fmd_firdemod, blob: *const c_void, size: usize) -> c_int;
    pub fn fmd_firdemod_f64_stats(f: *const fmd_firdemod, guarded: *mut u64, patched: *mut u64) -> c_int;
    pub fn fmd_firdemod_tiling(f: *const fmd_firdemod, audio_per_tile: *mut u32, lds_bytes: *mut u32) -> c_int;
    pub fn fmd_firdemod_kernel_name(f: *const fmd_firdemod, name: *mut c_char, cap: usize) -> c_int;
    pub fn fmd_fir_kernel_name(f: *const fmd_fir, name: *mut c_char, cap: usize) -> c_int;
    pub fn fmd_sink_new(config: *const DemodConfig, n_channels: u32, device_ids: *const i32, n_devices: u32, nbytes: usize, depth: u32, callback: fmd_sink_callback, user: *mut c_void, out: *mut *mut fmd_sink) -> c_int;
    pub fn fmd_sink_free(s: *mut fmd_sink);
    pub fn fmd_sink_acquire(s: *mut fmd_sink, iq: *mut *mut u8) -> c_int;
    pub fn fmd_sink_submit(s: *mut fmd_sink) -> c_int;
    pub fn fmd_sink_release(s: *mut fmd_sink) -> c_int;
    pub fn fmd_sink_poll(s: *mut fmd_sink) -> c_int;
    pub fn fmd_sink_drain(s: *mut fmd_sink) -> c_int;
    pub fn fmd_sink_info(s: *const fmd_sink, out_cap: *mut usize, n_devices: *mut u32, in_flight: *mut u32) -> c_int;
    pub fn fmd_sink_f64_stats(s: *const fmd_sink, guarded: *mut u64, patched: *mut u64) -> c_int;
    pub fn fmd_rtltcp_open(host: *const c_char, port: u16, timeout_ms: u32, out: *mut *mut fmd_rtltcp) -> c_int;
    pub fn fmd_rtltcp_close(s: *mut fmd_rtltcp);
    pub fn fmd_rtltcp_info(s: *const fmd_rtltcp, tuner_type: *mut u32, gain_count: *mut u32) -> c_int;
    pub fn fmd_rtltcp_read_sync(s: *mut fmd_rtltcp, buf: *mut u8, nbytes: usize, n_read: *mut usize) -> c_int;
    pub fn fmd_rtltcp_command(s: *mut fmd_rtltcp, opcode: u8, param: u32) -> c_int;
    pub fn fmd_rtltcp_read_many(sources: *const *mut fmd_rtltcp, n: u32, base: *mut u8, row_stride: usize, nbytes: usize, n_read: *mut usize) -> c_int;
    pub fn fmd_sink_fill_from_rtltcp(s: *mut fmd_sink, sources: *const *mut fmd_rtltcp, n_sources: u32, n_short: *mut u32) -> c_int;
    pub fn fmd_sink_pump_rtltcp(s: *mut fmd_sink, sources: *const *mut fmd_rtltcp, n_sources: u32, max_buffers: u64, n_submitted: *mut u64) -> c_int;
}

/// Error in the crate's convention (`src/error.rs:8,40-44`: a Result, never a panic).
#[derive(Debug)]
pub struct FmdError {
    pub status: i32,
    pub message: String,
}

impl std::fmt::Display for FmdError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "fmd error {}: {}", self.status, self.message)
    }
}
impl std::error::Error for FmdError {}

pub type Result<T> = std::result::Result<T, FmdError>;

fn check(status: c_int) -> Result<()> {
    if status == 0 {
        return Ok(());
    }
    let msg = unsafe {
        format!("{}: {}", CStr::from_ptr(fmd_strerror(status)).to_string_lossy(), CStr::from_ptr(fmd_last_error()).to_string_lossy())
    };
    Err(FmdError { status, message: msg })
}

/// `optimal_settings(freq, rate)`, examples/simple_fm.rs:189-214.
pub fn optimal_settings(freq: u32, rate: u32, rate_resample: u32) -> Result<(RadioConfig, DemodConfig)> {
    let (mut r, mut d) = (RadioConfig::default(), DemodConfig::default());
    check(unsafe { fmd_optimal_settings(freq, rate, rate_resample, &mut r, &mut d) })?;
    Ok((r, d))
}

/// Drop-in for the example's `Demod`: `Demod::new(config)` / `demod.demodulate(buf)`.
pub struct Demod {
    handle: *mut fmd_demod,
    pub config: DemodConfig,
}

// One caller at a time, like `&mut self` in the reference; the handle may move between threads.
unsafe impl Send for Demod {}

impl Demod {
    /// examples/simple_fm.rs:243-252
    pub fn new(config: DemodConfig) -> Result<Self> {
        let dev = DeviceConfig { n_channels: 1, device_id: -1, flags: 0 };
        let mut handle: *mut fmd_demod = std::ptr::null_mut();
        check(unsafe { fmd_demod_new(&config, &dev, &mut handle) })?;
        Ok(Demod { handle, config })
    }

    /// examples/simple_fm.rs:256-269.  Where the reference panics (len % 8 != 0, < 2 decimated samples) this
    /// returns Err.
    pub fn demodulate(&mut self, buf: Vec<u8>) -> Result<Vec<i16>> {
        let cap = unsafe { fmd_out_cap(&self.config, buf.len()) } + 1;
        let mut out = vec![0i16; cap];
        let mut n: usize = 0;
        check(unsafe { fmd_demod_demodulate(self.handle, buf.as_ptr(), buf.len(), out.as_mut_ptr(), cap, &mut n) })?;
        out.truncate(n);
        Ok(out)
    }

    /// STREAM LIFETIME for callers of the raw `fmd_demod_demodulate_device` / `fmd_fir_filter_device` / `fmd_firdemod_demodulate_device`
    /// bindings above: the `stream` of a handle's most recent device launch must stay alive until the handle's next device launch or
    /// completion point (`check`, `check_prev`, `state`, `set_state`, a host entry) has returned (include/fmd.h) -- the safe wrappers
    /// of this crate only use the library's own stream and are not affected.
    ///
    /// Completion / verification point after device-side launches (`fmd_demod_check`): waits, surfaces device-side
    /// assertions and settles the guarded f64 samples against the host libm.  The host entry points used by
    /// `demodulate` do this themselves; it is here for callers that drive `fmd_demod_demodulate_device` directly.
    pub fn check(&mut self) -> Result<()> {
        check(unsafe { fmd_demod_check(self.handle) })
    }

    /// The same one launch back (`fmd_demod_check_prev`): with launches 1 ..= n enqueued on the device entry point, waits for launch
    /// n - 1 only and settles its f64 samples while launch n runs -- the `demodulate` -> `output` cadence of `simple_fm.rs:150-156`
    /// without serialising host and device.  Until a launch has been settled its output buffer must stay allocated and unread, its
    /// input buffer unmodified, and the stream of the most recent launch alive (include/fmd.h).
    pub fn check_prev(&mut self) -> Result<()> {
        check(unsafe { fmd_demod_check_prev(self.handle) })
    }

    /// `fmd_demod_check_behind`: the completion point `back` launches back (0 = `check`, 1 = `check_prev`, 2 keeps a whole launch queued
    /// behind the running one and absorbs a late host).  Launches are settled in order; the last THREE launches' output buffers must
    /// stay allocated and unread, their input buffers unmodified, until they are settled.
    pub fn check_behind(&mut self, back: u32) -> Result<()> {
        check(unsafe { fmd_demod_check_behind(self.handle, back) })
    }

    pub fn state(&mut self) -> Result<DemodState> {
        let mut s = DemodState::default();
        check(unsafe { fmd_demod_get_state(self.handle, 0, &mut s) })?;
        Ok(s)
    }
}

impl Drop for Demod {
    fn drop(&mut self) {
        unsafe { fmd_demod_free(self.handle) }
    }
}

/// Many independent streams on one GPU: `n_channels` reference `Demod`s behind one handle (the case the GPU is
/// for).  `iq` is channel-major `[n_channels][nbytes]`, exactly `n_channels` `read_sync` buffers back to back.
pub struct DemodBank {
    handle: *mut fmd_demod,
    pub config: DemodConfig,
    pub n_channels: usize,
}

unsafe impl Send for DemodBank {}

impl DemodBank {
    pub fn new(config: DemodConfig, n_channels: usize, device_id: i32) -> Result<Self> {
        let dev = DeviceConfig { n_channels: n_channels as u32, device_id, flags: 0 };
        let mut handle: *mut fmd_demod = std::ptr::null_mut();
        check(unsafe { fmd_demod_new(&config, &dev, &mut handle) })?;
        Ok(DemodBank { handle, config, n_channels })
    }

    /// One `Demod::demodulate` per channel; returns one `Vec<i16>` per channel.
    pub fn demodulate(&mut self, iq: &[u8]) -> Result<Vec<Vec<i16>>> {
        assert!(iq.len() % self.n_channels == 0, "iq must hold n_channels equal-sized buffers");
        let nbytes = iq.len() / self.n_channels;
        let cap = unsafe { fmd_out_cap(&self.config, nbytes) } + 1;
        let mut out = vec![0i16; cap * self.n_channels];
        let mut lens = vec![0usize; self.n_channels];
        check(unsafe { fmd_demod_demodulate_batch(self.handle, iq.as_ptr(), nbytes, out.as_mut_ptr(), cap, lens.as_mut_ptr()) })?;
        Ok((0..self.n_channels).map(|c| out[c * cap..c * cap + lens[c]].to_vec()).collect())
    }
}

impl Drop for DemodBank {
    fn drop(&mut self) {
        unsafe { fmd_demod_free(self.handle) }
    }
}

/// The producer side of the boundary as a trait: `RtlSdr::read_sync(&self, buf: &mut [u8]) -> Result<usize>`
/// (src/lib.rs:153).  The reference has no `read_async`; this does not add one.
pub trait IqSource {
    /// Fill `buf` with interleaved u8 I/Q; the number of bytes written, `< buf.len()` meaning samples were lost
    /// or the source ended (callers of the reference treat that as fatal, examples/simple_fm.rs:122).
    fn read_sync(&mut self, buf: &mut [u8]) -> std::result::Result<usize, Box<dyn std::error::Error>>;
}

/// File / stdin mode of the example (examples/simple_fm.rs:65-84): any `Read` is a source.
impl<R: std::io::Read> IqSource for R {
    fn read_sync(&mut self, buf: &mut [u8]) -> std::result::Result<usize, Box<dyn std::error::Error>> {
        let mut got = 0;
        while got < buf.len() {
            let n = self.read(&mut buf[got..])?;
            if n == 0 {
                break;
            }
            got += n;
        }
        Ok(got)
    }
}

/// An rtl_tcp server (the reference's own examples/rtl_tcp.rs, or osmocom's) as an `IqSource`: a dongle on another host
/// feeding the GPU `Demod`.  Handshake "RTL0" + tuner type + gain count (examples/rtl_tcp.rs:691-697), raw u8 IQ
/// (:609-631), 5-byte big-endian commands (:639-678) -- all behind `fmd_rtltcp_*` of the C ABI.
pub struct RtlTcpSource {
    handle: *mut fmd_rtltcp,
    pub tuner_type: u32,
    pub gain_count: u32,
}

unsafe impl Send for RtlTcpSource {}

impl RtlTcpSource {
    pub fn connect(host: &str, port: u16, timeout_ms: u32) -> Result<Self> {
        let chost = std::ffi::CString::new(host).map_err(|_| FmdError { status: -1, message: "host contains NUL".into() })?;
        let mut handle: *mut fmd_rtltcp = std::ptr::null_mut();
        check(unsafe { fmd_rtltcp_open(chost.as_ptr(), port, timeout_ms, &mut handle) })?;
        let (mut t, mut g) = (0u32, 0u32);
        check(unsafe { fmd_rtltcp_info(handle, &mut t, &mut g) })?;
        Ok(RtlTcpSource { handle, tuner_type: t, gain_count: g })
    }

    /// One 5-byte command (opcode + big-endian parameter); `config_sdr` of the example (examples/simple_fm.rs:217-229)
    /// maps to 0x03 (gain mode), 0x0e (bias tee), 0x01 (frequency), 0x02 (sample rate).
    pub fn command(&mut self, opcode: u8, param: u32) -> Result<()> {
        check(unsafe { fmd_rtltcp_command(self.handle, opcode, param) })
    }
}

impl IqSource for RtlTcpSource {
    fn read_sync(&mut self, buf: &mut [u8]) -> std::result::Result<usize, Box<dyn std::error::Error>> {
        let mut n: usize = 0;
        check(unsafe { fmd_rtltcp_read_sync(self.handle, buf.as_mut_ptr(), buf.len(), &mut n) })?;
        Ok(n)
    }
}

impl Drop for RtlTcpSource {
    fn drop(&mut self) {
        unsafe { fmd_rtltcp_close(self.handle) }
    }
}

/// `receive()` + `process()` of the example (examples/simple_fm.rs:100-170) in one loop: blocks of
/// `block_len` bytes (the reference uses DEFAULT_BUF_LENGTH = 262144, src/lib.rs:25) from `source` through the GPU
/// `Demod` into `sink` as raw native-endian s16 (`output()`, examples/simple_fm.rs:430-438).  A short final block
/// ends the stream as in the reference (:122-125).
pub fn run<S: IqSource, W: std::io::Write>(source: &mut S, demod: &mut Demod, sink: &mut W, block_len: usize)
    -> std::result::Result<u64, Box<dyn std::error::Error>> {
    let mut total = 0u64;
    let mut buf = vec![0u8; block_len];
    loop {
        let n = source.read_sync(&mut buf)?;
        if n < block_len {
            break;
        }
        let audio = demod.demodulate(buf.clone())?;
        let bytes = unsafe { std::slice::from_raw_parts(audio.as_ptr() as *const u8, audio.len() * 2) };
        sink.write_all(bytes)?;
        sink.flush()?;
        total += audio.len() as u64;
    }
    Ok(total)
}
