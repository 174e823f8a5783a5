"""mvslam_amd -- MI355X-native two-view geometry hot path of mvSLAM's front end.

csrc/    hand-written HIP kernels + the C ABI (include/mvslam_hip.h) -> lib/libmvslam_hip.so
compat/  C++ header shim that keeps the reference's call surface (namespace mvSLAM) on the C ABI
capi.py  ctypes plumbing used by tests/ and bench.py
synth.py synthetic pair generator
"""
from . import capi, synth  # noqa: F401
