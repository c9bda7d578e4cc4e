"""cvids_amd -- MI355X-native dense TSDF fusion backend for the OpenChisel path of CVIDS.

The product is the C-ABI shared library `libchisel_hip.so` (include/chisel_hip.h, sources in
cvids_amd/csrc) plus the C++ facade headers in cvids_amd/open_chisel.  This Python package is
only the host-side mirror used by the tests and bench.py: ctypes bindings (`cvids_amd.capi`),
classes with the reference's names (`cvids_amd.chisel`), synthetic depth streams
(`cvids_amd.synth`) and the multi-GPU sharding driver (`cvids_amd.sharded`).

There is no CPU implementation in this package: without the built library every call raises.
"""
from .capi import ChiselHipError, build_library, library_path, load_library  # noqa: F401
