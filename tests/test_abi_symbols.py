"""The C-ABI shared library loads without a GPU and exports every entry point the public headers declare."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments mention Rust paths and types, not declarations
    names = set(re.findall(r"\b(xhe_[a-z0-9_]+)\s*\(", text))
    assert names, header
    return names


def test_library_exports_every_declared_entry_point():
    import xelis_he_amd
    lib = xelis_he_amd.load_library()
    declared = _declared("xhe.h") | _declared("xhe_builder.h")
    missing = sorted(n for n in declared if not hasattr(lib, n))
    assert not missing, "declared in include/*.h but not exported by libxhe.so: %s" % missing
    for must in ("xhe_create", "xhe_verify_batch", "xhe_batch_upload", "xhe_batch_verify", "xhe_batch_apply", "xhe_msm_vartime",
                 "xhe_ristretto_decompress_batch", "xhe_ristretto_compress_batch", "xhe_set_stage_timing"):
        assert must in declared


def test_create_fails_loudly_without_a_gpu():
    """No CPU fallback: without a gfx950 device the context is refused with a message (skipped where a GPU exists)."""
    import pytest
    import xelis_he_amd
    lib = xelis_he_amd.load_library()
    lib.xhe_create.restype = ctypes.c_void_p
    lib.xhe_last_error.restype = ctypes.c_char_p
    lib.xhe_last_error.argtypes = [ctypes.c_void_p]
    ctx = lib.xhe_create(0)
    if ctx:
        lib.xhe_destroy.argtypes = [ctypes.c_void_p]
        lib.xhe_destroy(ctx)
        pytest.skip("a GPU is present")
    assert lib.xhe_last_error(None)
