"""ctypes binding of libfmd_hip.so -- exactly the declarations of include/fmd.h."""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
SO_PATH = os.environ.get("FMD_LIB") or os.path.join(PKG_DIR, "libfmd_hip.so")   # FMD_LIB: tuning builds only
CSRC = os.path.join(PKG_DIR, "csrc")

FMD_OK = 0
FMD_ERR_INVALID_ARG = -1
FMD_ERR_BAD_LENGTH = -2
FMD_ERR_TOO_SHORT = -3
FMD_ERR_BAD_RATES = -4
FMD_ERR_CAPACITY = -5
FMD_ERR_UNSUPPORTED = -6
FMD_ERR_BAD_STATE = -7
FMD_ERR_NO_DEVICE = -8
FMD_ERR_HIP = -9
FMD_ERR_NOMEM = -10
FMD_ERR_IO = -11

DEFAULT_BUF_LENGTH = 16 * 16384      # src/lib.rs:25


class RadioConfig(C.Structure):
    """struct RadioConfig, simple_fm.rs:173-176"""
    _fields_ = [("capture_freq", C.c_uint32), ("capture_rate", C.c_uint32)]


class DemodConfig(C.Structure):
    """struct DemodConfig, simple_fm.rs:179-185"""
    _fields_ = [("rate_in", C.c_uint32), ("rate_out", C.c_uint32), ("rate_resample", C.c_uint32),
                ("downsample", C.c_uint32), ("output_scale", C.c_uint32)]

    def __repr__(self):
        return "DemodConfig(rate_in=%d, rate_out=%d, rate_resample=%d, downsample=%d, output_scale=%d)" % (
            self.rate_in, self.rate_out, self.rate_resample, self.downsample, self.output_scale)


class DemodState(C.Structure):
    """mutable fields of struct Demod, simple_fm.rs:232-239"""
    _fields_ = [("prev_index", C.c_uint32), ("now_lpr", C.c_int32), ("prev_lpr_index", C.c_int32),
                ("lp_now_re", C.c_int32), ("lp_now_im", C.c_int32),
                ("demod_pre_re", C.c_int32), ("demod_pre_im", C.c_int32)]

    def as_dict(self):
        return {"prev_index": self.prev_index, "now_lpr": self.now_lpr, "prev_lpr_index": self.prev_lpr_index,
                "lp_now": [self.lp_now_re, self.lp_now_im], "demod_pre": [self.demod_pre_re, self.demod_pre_im]}


class DeviceConfig(C.Structure):
    _fields_ = [("n_channels", C.c_uint32), ("device_id", C.c_int32), ("flags", C.c_uint32)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("amplitude", C.c_uint32), ("noise", C.c_uint32),
                ("dev_q32", C.c_uint32), ("mod_period", C.c_uint32)]


# name -> (restype, argtypes); must list every function include/fmd.h declares
_vp, _sz = C.c_void_p, C.c_size_t
_u8p, _i16p, _szp = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_size_t)
PROTOTYPES = {
    "fmd_optimal_settings": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RadioConfig), C.POINTER(DemodConfig)]),
    "fmd_demod_new": (C.c_int, [C.POINTER(DemodConfig), C.POINTER(DeviceConfig), C.POINTER(_vp)]),
    "fmd_demod_free": (None, [_vp]),
    "fmd_demod_reset": (C.c_int, [_vp]),
    "fmd_demod_demodulate": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "fmd_demod_demodulate_batch": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "fmd_demod_demodulate_device": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "fmd_demod_set_block_len": (C.c_int, [_vp, _sz]),
    "fmd_demod_check": (C.c_int, [_vp]),
    "fmd_demod_check_prev": (C.c_int, [_vp]),
    "fmd_demod_check_behind": (C.c_int, [_vp, C.c_uint32]),
    "fmd_demod_set_event_ordering": (C.c_int, [_vp, C.c_int]),
    "fmd_demod_f64_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "fmd_demod_last_out_len": (C.c_int, [_vp, _szp]),
    "fmd_host_alloc": (C.c_int, [_sz, C.POINTER(_vp)]),
    "fmd_host_free": (C.c_int, [_vp]),
    "fmd_out_cap": (_sz, [C.POINTER(DemodConfig), _sz]),
    "fmd_demod_get_state": (C.c_int, [_vp, C.c_uint32, C.POINTER(DemodState)]),
    "fmd_demod_set_state": (C.c_int, [_vp, C.c_uint32, C.POINTER(DemodState)]),
    "fmd_synth_fill_device": (C.c_int, [C.c_int, _vp, C.c_uint32, _sz, C.c_uint64, C.POINTER(SynthParams), _vp]),
    "fmd_strerror": (C.c_char_p, [C.c_int]),
    "fmd_last_error": (C.c_char_p, []),
    "fmd_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "fmd_version": (C.c_int, []),
    "fmd_demod_tiling": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fmd_demod_tiling_plan": (C.c_int, [_vp]),
    "fmd_demod_last_kernel": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "fmd_demod_set_tiling": (C.c_int, [_vp, C.c_uint32]),
    "fmd_fir_new": (C.c_int, [_i16p, C.c_uint32, C.c_uint32, C.POINTER(DeviceConfig), C.POINTER(_vp)]),
    "fmd_fir_free": (None, [_vp]),
    "fmd_fir_reset": (C.c_int, [_vp]),
    "fmd_fir_tap_digits": (C.c_int, [_vp]),
    "fmd_fir_out_cap": (_sz, [C.c_uint32, C.c_uint32, _sz]),
    "fmd_fir_filter_batch": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "fmd_fir_filter_device": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _szp, _vp]),
    "fmd_firdemod_new": (C.c_int, [_i16p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(DeviceConfig), C.POINTER(_vp)]),
    "fmd_firdemod_free": (None, [_vp]),
    "fmd_firdemod_reset": (C.c_int, [_vp]),
    "fmd_firdemod_out_cap": (_sz, [C.c_uint32, C.c_uint32, C.c_uint32, _sz]),
    "fmd_firdemod_demodulate_batch": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "fmd_firdemod_demodulate_device": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _szp, _vp]),
    "fmd_firdemod_check": (C.c_int, [_vp]),
    "fmd_firdemod_get_state": (C.c_int, [_vp, C.c_uint32, C.POINTER(DemodState)]),
    "fmd_firdemod_checkpoint_size": (C.c_size_t, [_vp]),
    "fmd_firdemod_checkpoint": (C.c_int, [_vp, _vp, _sz]),
    "fmd_firdemod_resume": (C.c_int, [_vp, _vp, _sz]),
    "fmd_firdemod_f64_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "fmd_firdemod_tiling": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fmd_firdemod_kernel_name": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "fmd_fir_kernel_name": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "fmd_sink_new": (C.c_int, [C.POINTER(DemodConfig), C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, _sz, C.c_uint32, _vp, _vp, C.POINTER(_vp)]),
    "fmd_sink_free": (None, [_vp]),
    "fmd_sink_acquire": (C.c_int, [_vp, C.POINTER(_vp)]),
    "fmd_sink_submit": (C.c_int, [_vp]),
    "fmd_sink_release": (C.c_int, [_vp]),
    "fmd_sink_poll": (C.c_int, [_vp]),
    "fmd_sink_drain": (C.c_int, [_vp]),
    "fmd_sink_info": (C.c_int, [_vp, _szp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fmd_sink_f64_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "fmd_rtltcp_open": (C.c_int, [C.c_char_p, C.c_uint16, C.c_uint32, C.POINTER(_vp)]),
    "fmd_rtltcp_close": (None, [_vp]),
    "fmd_rtltcp_info": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fmd_rtltcp_read_sync": (C.c_int, [_vp, _vp, _sz, _szp]),
    "fmd_rtltcp_command": (C.c_int, [_vp, C.c_uint8, C.c_uint32]),
    "fmd_rtltcp_read_many": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_size_t, C.c_size_t, _szp]),
    "fmd_sink_fill_from_rtltcp": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]),
    "fmd_sink_pump_rtltcp": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]),
}

SINK_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.POINTER(C.c_int16), C.POINTER(C.c_size_t), C.c_size_t, C.c_int)


def build(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", ".hpp"))]
    srcs.append(os.path.join(ROOT, "include", "fmd.h"))
    fresh = os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(s) for s in srcs)
    if force or not fresh:
        subprocess.check_call(["make", "-s", "-C", CSRC, "all"])
    return SO_PATH


_lib = None


def lib():
    """Load libfmd_hip.so.  Fails loudly if it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same soname as
        # /opt/rocm's).  Whichever loads first serves both, and device pointers / streams are
        # only shareable with torch when it is torch's -- so let torch load it first if present.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(SO_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if os.environ.get("FMD_LIB"):             # an older build loaded for an A/B (tools/ab.py): newer entry points absent
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class FmdError(RuntimeError):
    def __init__(self, status):
        l = lib()
        self.status = status
        super().__init__("%s (%d): %s" % (l.fmd_strerror(status).decode(), status, l.fmd_last_error().decode()))


def check(status):
    if status != FMD_OK:
        raise FmdError(status)
