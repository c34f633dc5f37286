"""CPU-side checks of the drop-in boundary: libmmplace loads, exports every symbol that
include/mmplace.h declares, and refuses to run without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from modelmesh_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_header_symbols_are_all_exported_and_typed():
    _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "mmplace.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mmp_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mmplace.h but not exported"
    assert _lib.load().mmp_abi_version() == 3  # 3: latency-based rebalancers, single-caller requests, bounded device calls (round 5)


def test_struct_layouts_match_the_header():
    """The numpy mirrors must have the sizes the header documents (64 B pod row / request, 24 B model
    row, 16 B result)."""
    assert _lib.POD_ROW.itemsize == 64 and _lib.PLACE_REQ.itemsize == 64
    assert _lib.MODEL_ROW.itemsize == 24 and _lib.PLACE_OUT.itemsize == 16
    assert _lib.POD_ROW.fields["flags"][1] == 56 and _lib.PLACE_REQ.fields["fresh_lru"][1] == 32


def test_no_cpu_fallback_without_a_device():
    _ensure_built()
    lib = _lib.load()
    cfg = _lib.MmpConfig(0, 0, 100, 1000)
    h = C.c_void_p()
    rc = lib.mmp_create(C.byref(cfg), C.byref(h))
    if os.path.exists("/dev/kfd"):
        assert rc == 0
        lib.mmp_destroy(h)
    else:
        assert rc == _lib.MMP_ENODEVICE and not h.value
        assert b"no CPU path" in lib.mmp_last_error(None)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under modelmesh_amd/ may reference it."""
    pkg = os.path.join(ROOT, "modelmesh_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"


def test_bitmap_packing_roundtrip():
    from modelmesh_amd.solver import bitmap_from_bool
    from oracle.bind import unpack_bitmap
    rng = np.random.default_rng(0)
    for p in (1, 63, 64, 65, 1000):
        m = rng.random((3, p)) < 0.4
        assert np.array_equal(unpack_bitmap(bitmap_from_bool(m), p).astype(bool), m)


def test_library_never_uses_the_null_stream():
    """Source lint.  Synchronous hipMemcpy / hipMemset / hipDeviceSynchronize run on (or wait for) the legacy
    null stream; a single hipMemset there at context creation was measured to serialise, for the rest of the
    process, kernels issued on separate streams.  The library owns non-blocking streams and uses only the
    *Async forms on them."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for f in glob.glob(os.path.join(root, "modelmesh_amd", "csrc", "*")):
        for i, line in enumerate(open(f), 1):
            code = line.split("//")[0]
            if re.search(r"\bhip(Memcpy|Memset|Memcpy2D|MemcpyToSymbol|MemcpyFromSymbol)\s*\(", code) or "hipDeviceSynchronize" in code:
                bad.append(f"{os.path.basename(f)}:{i}: {line.strip()}")
            if re.search(r"<<<[^>]*>>>", code) or re.search(r"hipLaunchKernelGGL\([^;]*,\s*0\s*,\s*(0|nullptr)\s*,", code):
                bad.append(f"{os.path.basename(f)}:{i}: launch on the null stream: {line.strip()}")
    assert not bad, "\n".join(bad)


def _build_c_example(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "modelmesh_amd", "lib")
    exe = str(tmp_path / "place_one")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "place_one.c"), "-L" + libdir, "-lmmplace", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True, capture_output=True, text=True)
    return exe


def test_header_is_plain_c99_and_a_c_host_links(tmp_path):
    """include/mmplace.h is the boundary for JNI / cgo / FFI hosts: it must compile as strict C99, and a C program
    using it must link against libmmplace.so.  Without a GPU the program reports MMP_ENODEVICE (exit 77)."""
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 77 and "no CPU path" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_host_places_one_model(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "best=1" in r.stdout


def test_a_host_without_rccl_gets_an_error_code_and_the_reason():
    """the dlopen binding used to read dlerror() twice (the second call returns NULL: std::string from
    nullptr, a crash) and reported the reason of a fresh, empty object.  MMP_RCCL_PATH names the one library to bind;
    a path that does not exist is a host without RCCL."""
    import subprocess
    import sys
    _ensure_built()
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from modelmesh_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = C.create_string_buffer(128)\n"
        "rc = lib.mmp_shard_unique_id(buf)\n"
        "print(rc, lib.mmp_last_error(None).decode())\n" % ROOT)
    env = dict(os.environ, MMP_RCCL_PATH="/nonexistent/librccl-not-here.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rc, msg = out.stdout.strip().split(" ", 1)
    assert int(rc) == _lib.MMP_ENODEVICE
    assert "RCCL is not available" in msg and "librccl-not-here" in msg
