"""not-gpu: the C-ABI library loads, exports every symbol include/srz.h declares, and fails LOUDLY without a GPU."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, "include", "srz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(srz_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("srz_create", "srz_destroy", "srz_draw", "srz_texture_upload", "srz_frameset_create",
                 "srz_frameset_render", "srz_set_shard", "srz_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import srz
    lib = ctypes.CDLL(srz.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"libsrz.so does not export {name}"
    for name in srz.EXPORTS:
        assert name in declared_functions()
    from srz import abi
    header = int(re.search(r"#define SRZ_ABI_VERSION (\d+)", open(os.path.join(REPO, "include", "srz.h")).read()).group(1))
    assert lib.srz_abi_version() == header == abi.SRZ_ABI_VERSION == 7


def test_binding_refuses_a_library_of_another_abi_version(tmp_path):
    """a stale or foreign libsrz.so (SRZ_LIB_PATH) must be refused at load, not crash later: round 2 lost a test run to a
    binding that sent the v3 stream sentinel to a v2 library"""
    import subprocess
    import sys
    src = tmp_path / "fake.c"
    src.write_text("int srz_abi_version(void) { return 2; }\n")
    so = tmp_path / "libsrz_fake.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    code = ("import sys; sys.path.insert(0, %r); import srz\n"
            "try:\n    srz.lib()\nexcept ImportError as e:\n    print('REFUSED', e)\n" % os.path.join(REPO, "software-rasterizer_amd"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRZ_LIB_PATH=str(so)), capture_output=True, text=True)
    assert "REFUSED" in out.stdout and "SRZ_ABI_VERSION 2" in out.stdout, out.stdout + out.stderr


def test_struct_sizes_match_the_header():
    from srz import abi
    assert abi.TRI_DTYPE.itemsize == 96 and abi.LIGHT_DTYPE.itemsize == 24
    assert ctypes.sizeof(abi.SrzBatch) == 24
    assert ctypes.sizeof(abi.SrzFrame) == 88
    assert ctypes.sizeof(abi.SrzStats) == 56


def test_no_cpu_fallback():
    """Without a usable gfx950 device srz_create must fail with SRZ_E_NODEVICE and say why."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import srz
    with pytest.raises(srz.SrzError) as e:
        srz.Context(0)
    assert e.value.code == srz.abi.SRZ_E_NODEVICE
    assert "no CPU fallback" in str(e.value)


def test_host_library_loads():
    from srz import host
    host.lib()
