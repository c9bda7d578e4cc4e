"""The C-ABI library loads here (no GPU) and exports every symbol include/chisel_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="chisel_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(chisel_hip_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(hip_lib):
    from cvids_amd import capi
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(hip_lib, s), "libchisel_hip.so does not export %s" % s
    assert sorted(capi.EXPORTS) == syms, "cvids_amd/capi.py EXPORTS out of sync with include/chisel_hip.h"
    txt = open(os.path.join(ROOT, "include", "chisel_hip.h")).read()
    assert hip_lib.chisel_hip_abi_version() == capi.ABI_VERSION == int(re.search(r"#define\s+CHISEL_HIP_ABI_VERSION\s+(\d+)", txt).group(1))


def test_every_exported_symbol_is_declared(hip_lib):
    """the library exports exactly what include/chisel_hip.h (the boundary) and include/chisel_hip_selftest.h
    (device self-tests, debug read-outs) declare -- checked against the dynamic symbol table."""
    import subprocess
    from cvids_amd import capi
    selftest = declared_symbols("chisel_hip_selftest.h")
    assert sorted(capi.SELFTEST_EXPORTS) == selftest
    out = subprocess.run(["nm", "-D", "--defined-only", capi.library_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("chisel_hip_")})
    assert exported == sorted(declared_symbols() + selftest), set(exported) ^ set(declared_symbols() + selftest)


def test_struct_layouts_match_header():
    from cvids_amd import capi
    # sizes implied by the field lists of include/chisel_hip.h on LP64
    assert ctypes.sizeof(capi.Config) == 48
    assert ctypes.sizeof(capi.Integrator) == 20
    assert ctypes.sizeof(capi.DepthFrame) == 96
    assert ctypes.sizeof(capi.ColorFrame) == 88


def test_array_lengths_of_the_python_mirror_match_the_header():
    """the arrays chisel_hip_get_launch_stats / get_counters / get_profile fill are sized by macros of the header: the Python mirror's
    name lists must have those lengths (a list one short would let the library write past the ctypes buffer)"""
    from cvids_amd import capi, chisel
    txt = open(os.path.join(ROOT, "include", "chisel_hip.h")).read()
    macro = lambda name: int(re.search(r"(?:#define\s+%s\s+|%s\s*=\s*)(\d+)" % (name, name), txt).group(1))  # (a macro or an enumerator)
    assert len(chisel.Chisel.LAUNCH_STATS) == macro("CHISEL_HIP_NUM_LAUNCH_STATS")
    assert capi.NUM_KERNELS == macro("CHISEL_HIP_NUM_KERNELS") == len(capi.KERNEL_NAMES)


def test_create_fails_loudly_without_gpu(hip_lib):
    """No CPU fallback: on a box without a gfx950 device create() must fail, not emulate."""
    from cvids_amd import capi
    if hip_lib.chisel_hip_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    from cvids_amd.chisel import Chisel
    with pytest.raises(capi.ChiselHipError):
        Chisel((16, 16, 16), 0.02, False)


def test_invalid_config_rejected(hip_lib):
    from cvids_amd import capi
    from cvids_amd.chisel import Chisel
    with pytest.raises(capi.ChiselHipError) as e:
        Chisel((16, 16, 8), 0.02, False)
    assert e.value.code == 5


def test_chunk_owner_is_a_partition(hip_lib):
    from cvids_amd.chisel import chunk_owner
    for n in (1, 2, 4, 8):
        seen = set()
        for x in range(-5, 6):
            for y in range(-5, 6):
                for z in range(-5, 6):
                    o = chunk_owner((x, y, z), n, 2)
                    assert 0 <= o < n
                    seen.add(o)
        assert seen == set(range(n))
    # 2x2x2 super-blocks share an owner
    assert len({chunk_owner((2 + dx, 4 + dy, -6 + dz), 8, 2) for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)}) == 1
