"""Python side of the mock JVM (tests/jni_mock/jni.h): builds integration/mmplace_jni.cc against it and calls the
veneer's Java_com_ibm_watson_modelmesh_MmPlace_* functions as the JVM would (JNIEnv*, jclass, then the Java arguments:
int -> c_int32, long -> c_int64, boolean -> c_uint8, ByteBuffer -> pointer to a (address, capacity) object)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI_CC = os.path.join(ROOT, "integration", "mmplace_jni.cc")
LIBDIR = os.path.join(ROOT, "modelmesh_amd", "lib")
PREFIX = "Java_com_ibm_watson_modelmesh_MmPlace_"


class JObject(C.Structure):
    _fields_ = [("addr", C.c_void_p), ("cap", C.c_int64), ("name", C.c_char_p)]


class JEnv(C.Structure):
    _fields_ = [("throws", C.c_int32), ("finds", C.c_int32), ("pending_class", C.c_char * 128), ("pending_msg", C.c_char * 512),
                ("cls", JObject), ("cls_name", C.c_char * 128)]

    def pending(self):
        return (self.pending_class.decode(), self.pending_msg.decode()) if self.throws else None

    def clear(self):
        self.throws = 0
        self.pending_class = b""
        self.pending_msg = b""


def build(out_dir) -> str:
    so = os.path.join(str(out_dir), "libmmplace_jni_mock.so")
    cmd = ["g++", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "jni_mock"),
           "-I" + os.path.join(ROOT, "include"), JNI_CC, "-L" + LIBDIR, "-lmmplace", "-Wl,--no-undefined",
           "-Wl,-rpath," + LIBDIR, "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return so


class ByteBuffer:
    """A direct ByteBuffer over a numpy array (kept alive here); `short` lies about the capacity for the bounds tests."""

    def __init__(self, arr, cap=None, direct=True):
        self.arr = np.ascontiguousarray(arr)
        self.obj = JObject(self.arr.ctypes.data if direct else None, (self.arr.nbytes if cap is None else cap) if direct else -1, None)

    @property
    def ref(self):
        return C.byref(self.obj)


class Veneer:
    """call("placeBatch", h, reqs_bb, n, ...) with Python ints / ByteBuffer / None, typed by the JAVA declaration of the
    native (integration/GpuPlacementLB.java), so that a drift between the two files shows up as a wrong call here."""
    JT = {"int": C.c_int32, "long": C.c_int64, "boolean": C.c_uint8, "double": C.c_double, "ByteBuffer": C.c_void_p}
    RT = {"int": C.c_int32, "long": C.c_int64, "double": C.c_double, "void": None, "boolean": C.c_uint8}

    def __init__(self, so_path, java_natives):
        from modelmesh_amd import _lib
        _lib.load()  # torch's HIP runtime first (see _lib.load), then libmmplace; the veneer binds to the loaded copy
        self.lib = C.CDLL(so_path)
        self.env = JEnv()
        self.natives = java_natives
        self.cls = JObject(None, -1, b"com/ibm/watson/modelmesh/MmPlace")

    def call(self, name, *args):
        ret, params = self.natives[name]
        assert len(params) == len(args), (name, params, args)
        fn = getattr(self.lib, PREFIX + name)
        fn.restype = self.RT[ret]
        fn.argtypes = [C.POINTER(JEnv), C.POINTER(JObject)] + [self.JT[p] for p in params]
        conv = []
        for p, a in zip(params, args):
            if p == "ByteBuffer":
                conv.append(None if a is None else C.cast(a.ref, C.c_void_p))
            else:
                conv.append(a)
        return fn(C.byref(self.env), C.byref(self.cls), *conv)
