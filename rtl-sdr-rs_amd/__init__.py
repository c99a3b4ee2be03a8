"""rtl-sdr-rs_amd -- MI355X-native FM demodulation path of ccostes/rtl-sdr-rs' simple_fm example.

Python host-side mirror of the reference interface for this path (Demod, DemodConfig,
optimal_settings; examples/simple_fm.rs:172-269) over the C ABI of include/fmd.h
(libfmd_hip.so: hand-written gfx950 HIP kernels).  No CPU implementation of the path lives
here; without the built library or a GPU every entry point raises.
"""
from ._ffi import (DEFAULT_BUF_LENGTH, DemodConfig, DemodState, DeviceConfig, FmdError, RadioConfig,
                   SynthParams, build, check, lib)
from .demod import Demod, DemodBank, PinnedBuffer, device_count, optimal_settings, out_cap
from .fir import FirBank, FirDemodBank, auto_shift
from .sink import Sink, pump
from . import shard, synth

__all__ = ["DEFAULT_BUF_LENGTH", "Demod", "DemodBank", "PinnedBuffer", "FirBank", "Sink", "pump", "FirDemodBank", "auto_shift", "DemodConfig", "DemodState", "DeviceConfig", "FmdError",
           "RadioConfig", "SynthParams", "build", "check", "lib", "device_count", "optimal_settings", "out_cap",
           "shard", "synth"]
