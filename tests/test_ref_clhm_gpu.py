"""GPU: cache_replay_kernel (through the C ABI: mmp_caches_load_keyed / mmp_cache_replay / mmp_cache_read) held to the
REFERENCE'S OWN local-cache text — tests/golden/ref_clhm.npz, see tests/test_ref_clhm.py.  Every operation's result, evicted
keys in listener order, unload-buffer weight, weightedSize() and oldestTime(), and every cache's final deque and manager
fields, bit for bit; each stream once in a single replay call and once cut into several (the state persists on the device)."""
import os

import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd.solver import Solver
from tests import ref_clhm_cases as rc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_clhm.npz")


def _check(vec, name, pieces):
    caps, reserved, ops, outs, ev = (vec[f"{name}/{k}"] for k in ("caps", "reserved", "ops", "outs", "evicted"))
    hdr = vec[f"{name}/final_hdr"]
    seg, lus, wts, keys = rc.initial_state(caps, reserved)
    ubm = np.zeros(len(caps), dtype=_lib.UBM_STATE)
    ubm["reserved"] = reserved
    s = Solver(100, 1000)
    try:
        s.load_caches_keyed(seg, lus, wts, keys, caps, ubm)
        bounds = np.linspace(0, len(ops), pieces + 1).astype(int)
        for a, b in zip(bounds[:-1], bounds[1:]):
            got, gev = s.cache_replay(ops[a:b].astype(_lib.CACHE_OP), rc.NOW)
            for i in range(a, b):
                o, g = outs[i], got[i - a]
                want_ev = list(ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]])
                have_ev = list(gev[g["evicted_off"]: g["evicted_off"] + g["n_evicted"]])
                assert (g["result"], have_ev, g["buffer_weight"], g["weighted_size"], g["oldest_time"]) == \
                    (o["result"], want_ev, o["buffer_weight"], o["weighted_size"], o["oldest_time"]), (name, i, ops[i])
        off = 0
        for c in range(len(caps)):
            st = s.cache_read(c)
            n = int(hdr[c, 0])
            assert np.array_equal(st["key"], vec[f"{name}/final_key"][off:off + n]), (name, c)
            assert np.array_equal(st["weight"], vec[f"{name}/final_weight"][off:off + n]), (name, c)
            assert np.array_equal(st["last_used"], vec[f"{name}/final_last_used"][off:off + n]), (name, c)
            assert (st["capacity"], st["weighted_size"]) == (hdr[c, 1], hdr[c, 2]), (name, c)
            if reserved[c] >= 0:
                u = st["ubm"]
                assert (u["total_unloading"], u["total_occupancy"], u["cache_deficit"]) == (hdr[c, 3], hdr[c, 4], hdr[c, 5]), (name, c)
            off += n
    finally:
        s.close()
    return len(ops)


def test_device_cache_replay_equals_the_reference_text():
    vec = np.load(GOLDEN)
    n = sum(_check(vec, str(name), 1) for name in vec["names"])
    assert n > 15_000


def test_device_cache_state_persists_like_the_reference_text():
    """The same streams in 5 replay calls each: weightedSize, the oldestTime field (stale after a failed unload) and the
    manager's fields carry over between calls."""
    vec = np.load(GOLDEN)
    for name in vec["names"]:
        if len(vec[f"{name}/ops"]) >= 60:
            _check(vec, str(name), 5)
