"""modelmesh_amd — MI355X-native placement / eviction solver for ModelMesh's
ensureLoaded / invokeModel instance-selection hot path.

The compute lives in libmmplace (HIP, gfx950) behind the C ABI of
include/mmplace.h; this package is the thin ctypes veneer used by the tests and
bench.py.  There is no CPU implementation here.
"""
from . import _lib  # noqa: F401
