"""integration/mmplace_jni.cc EXECUTED ("the veneer exists on paper only"): no JDK exists here or on the GPU
box, so a mock JVM (tests/jni_mock/jni.h: the five JNIEnv calls the veneer makes, behaving as the JNI specification says)
stands where the JVM would, and the tests call the Java_..._MmPlace_* functions with the argument types of the JAVA
declarations.  CPU part: the veneer's own logic (buffer-capacity checks, error mapping to exceptions).  GPU part
(tests/test_jni_exec_gpu.py): the decision calls against the direct C-ABI path."""
import os

import numpy as np
import pytest

from modelmesh_amd import _lib
from tests import jni_mock as jm
from tests.test_jni_veneer import _java_natives

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(jm.LIBDIR, "libmmplace.so")), reason="libmmplace.so not built")


@pytest.fixture(scope="module")
def veneer(tmp_path_factory):
    return jm.Veneer(jm.build(tmp_path_factory.mktemp("jni")), _java_natives())


def test_abi_version_and_min_space_units_through_the_veneer(veneer):
    assert veneer.call("abiVersion") == 3
    # MM.java:765-771: max(defaultModelSizeUnits * loadingThreads * 2, capacity / 20) unless there is no unload manager
    assert veneer.call("minSpaceUnits", 128, 8, 1_000_000, 1) == _lib.load().mmp_min_space_units(128, 8, 1_000_000, 1)


def test_a_short_direct_buffer_becomes_illegal_argument_exception_before_the_library_is_called(veneer):
    """The C ABI trusts its sizes; the veneer must not (placeBatch / serveBatch / shardUniqueId check capacities)."""
    env = veneer.env
    reqs = jm.ByteBuffer(np.zeros(4, dtype=_lib.PLACE_REQ))
    outs = jm.ByteBuffer(np.zeros(4, dtype=_lib.PLACE_OUT))
    short = jm.ByteBuffer(np.zeros(4, dtype=_lib.PLACE_OUT), cap=3 * 16)
    for args, msg in (((0, reqs, 5, None, 0, 0, outs), "reqs shorter than n"),
                      ((0, reqs, 4, None, 0, 0, short), "outs shorter than n"),
                      ((0, reqs, 4, None, 3, 0, outs), "extraPool shorter"),
                      ((0, reqs, 4, jm.ByteBuffer(np.zeros(2, np.int32), direct=False), 2, 0, outs), "extraPool shorter")):
        env.clear()
        rc = veneer.call("placeBatch", *args)
        assert rc == -1  # MMP_EINVAL
        cls, text = env.pending()
        assert cls == "java/lang/IllegalArgumentException" and msg in text, (cls, text)
    env.clear()
    assert veneer.call("shardUniqueId", jm.ByteBuffer(np.zeros(64, np.uint8))) == -1
    assert env.pending()[0] == "java/lang/IllegalArgumentException"
    sreqs = jm.ByteBuffer(np.zeros(2, dtype=_lib.SERVE_REQ))
    env.clear()
    rc = veneer.call("serveBatch", 0, sreqs, 2, jm.ByteBuffer(np.zeros(3, dtype=_lib.SERVE_COUNTER)), 4, None, None, 0, 0,
                     jm.ByteBuffer(np.zeros(2, dtype=_lib.SERVE_OUT)))
    assert rc == -1 and "counters shorter" in env.pending()[1]


def test_a_library_error_becomes_illegal_state_exception_with_its_message(veneer):
    """No GPU here: create fails inside the library (MMP_ENODEVICE) -> IllegalStateException carrying mmp_last_error; with a
    GPU: a decision before any commit (MMP_ESTATE)."""
    import torch
    env = veneer.env
    env.clear()
    h = veneer.call("create", 0, 6553, 60_000)
    if not torch.cuda.is_available():
        assert h == 0
        cls, text = env.pending()
        assert cls == "java/lang/IllegalStateException" and "no CPU path" in text
        return
    assert h != 0 and env.pending() is None
    try:
        reqs = jm.ByteBuffer(np.zeros(1, dtype=_lib.PLACE_REQ))
        outs = jm.ByteBuffer(np.zeros(1, dtype=_lib.PLACE_OUT))
        rc = veneer.call("placeBatch", h, reqs, 1, None, 0, 0, outs)
        assert rc != 0
        cls, text = env.pending()
        assert cls == "java/lang/IllegalStateException" and text
    finally:
        veneer.call("destroy", h)
