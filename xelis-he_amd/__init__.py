"""xelis_he_amd -- Python binding of libxhe.so, the MI355X batch-verification engine for XELIS-HE.

The binding is a thin ctypes layer over the C ABI of include/xhe.h; it contains no arithmetic.  If the
HIP library is missing or no gfx950 device is present, constructing an Engine raises: there is no CPU
fallback on the product path."""
from ._capi import EcdlpTable, Engine, XheError, lib_path, load_library  # noqa: F401
from . import codes  # noqa: F401
