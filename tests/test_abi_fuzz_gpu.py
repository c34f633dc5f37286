"""GPU: the C ABI under hostile arguments: whatever a host passes — negative counts, null pointers, models and
instances that do not exist, exclusion ranges that leave their pool or overflow an int, garbage inside the pools — every call
returns an MMP_E* code or a well-formed result row, never a fault.  Where the arguments are legal the rows are also checked
against the oracle (a request naming a model or a caller that does not exist is legal: the reference's getNext returns null for
an unknown model, and an instance that is not in the table is simply not `self`).  hypothesis draws the requests; the fixed
cases name the argument classes one by one.  Runs under AddressSanitizer too (tools/asan_lib.sh)."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet

pytestmark = pytest.mark.gpu
INT_MAX, INT_MIN = 2**31 - 1, -(2**31)
EINVAL, ESTATE = -1, -5
BAD_REQUEST = -3


@pytest.fixture(scope="module")
def ctx():
    fleet = wl.make_fleet("C2", models=3000, pods=400)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s.load_fleet(fleet)
    yield s, fleet, OracleFleet(fleet)
    s.close()


def _well_formed(out, P):
    ch = out["chosen"]
    assert (((ch >= 0) & (ch < P)) | (ch == -1) | (ch == -2)).all(), ch[~(((ch >= 0) & (ch < P)) | (ch == -1) | (ch == -2))][:5]
    b = out["best"]
    assert (((b >= -1) & (b < P)) | (b == BAD_REQUEST)).all()
    assert (out["n_candidates"] >= 0).all() and (out["n_candidates"] <= P).all()


weird_i32 = st.sampled_from([0, 1, -1, -2, 5, 399, 400, 401, 2999, 3000, 3001, 65535, 2**20, INT_MAX, INT_MAX - 1, INT_MIN, INT_MIN + 1])
weird_i64 = st.sampled_from([0, 1, -1, 2**62, -(2**62), 2**63 - 1, -(2**63), 1_760_000_000_000, 1_760_000_020_000, 42])


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(data=st.data())
def test_requests_with_hostile_fields_are_decided_or_refused(ctx, data):
    s, fleet, orc = ctx
    P, M = fleet.n_pods, fleet.n_models
    n = data.draw(st.integers(1, 40))
    reqs, extra = wl.make_requests(fleet, data.draw(st.integers(0, 10_000)), n=n)
    for i in range(n):
        if data.draw(st.booleans()):
            reqs["model"][i] = data.draw(weird_i32)
        if data.draw(st.booleans()):
            reqs["self_pod"][i] = data.draw(weird_i32)
        if data.draw(st.integers(0, 3)) == 0:
            for f in ("last_used", "fresh_lru", "fresh_capacity", "fresh_used"):
                reqs[f][i] = data.draw(weird_i64)
            reqs["fresh_count"][i] = data.draw(weird_i32)
            reqs["fresh_rpm"][i] = data.draw(weird_i32)
            reqs["flags"][i] = data.draw(st.integers(0, 2**32 - 1))
            reqs["pick"][i] = data.draw(st.integers(0, 2**32 - 1))
    if len(extra):  # garbage inside the pool: instances that do not exist are nobody's exclusion
        k = data.draw(st.integers(0, len(extra)))
        extra[:k] = [data.draw(weird_i32) for _ in range(k)]
    got = s.place(reqs, extra, fleet.now)
    _well_formed(got, P)
    # legal as far as the reference goes (unknown model -> null; a caller that is not in the table is not `self`; an exclusion that
    # names no instance excludes nothing): the oracle decides the same — on the requests whose arithmetic inputs were left alone
    sane = (reqs["fresh_capacity"] >= 0) & (reqs["fresh_capacity"] < 2**40) & (reqs["fresh_used"] >= 0) & (reqs["fresh_used"] < 2**40) & \
           (np.abs(reqs["fresh_count"].astype(np.int64)) < 2**20) & (reqs["flags"] <= 1) & (np.abs(reqs["fresh_rpm"].astype(np.int64)) < 2**24)
    r2 = reqs.copy()
    r2["model"] = np.where((r2["model"] < 0) | (r2["model"] >= M), -1, r2["model"])
    known = r2["model"] >= 0
    r2["self_pod"] = np.where((r2["self_pod"] < 0) | (r2["self_pod"] >= P), -1, r2["self_pod"])
    e2 = np.where((extra < 0) | (extra >= P), -1, extra).astype(np.int32)
    chk = sane & known
    if chk.any():
        want = orc.place(r2[chk], e2, fleet.now)
        for f in ("chosen", "best", "n_candidates", "hash"):
            assert np.array_equal(got[f][chk], want[f]), (f, int(np.flatnonzero(got[f][chk] != want[f])[0]))
    assert (got["chosen"][~known] == -1).all() and (got["n_candidates"][~known] == 0).all()


@pytest.mark.parametrize("off,cnt", [(-1, 1), (0, -1), (INT_MAX, 1), (INT_MAX, INT_MAX), (5, 100), (0, 2**20), (INT_MIN, 3), (7, INT_MIN)])
def test_exclusion_ranges_that_leave_the_pool_are_refused(ctx, off, cnt):
    """Host-pointer calls validate every request's range against the pool's length (int64 arithmetic: off + cnt must not wrap);
    bounded device-pointer calls answer such a request {MMP_NONE, MMP_BAD_REQUEST, 0, 0} and decide the others."""
    import torch
    s, fleet, orc = ctx
    reqs, extra = wl.make_requests(fleet, 5, n=600)
    extra = np.concatenate([extra, np.zeros(8, np.int32)])
    bad = reqs.copy()
    bad["extra_off"][17], bad["n_extra"][17] = off, cnt
    out = np.zeros(len(bad), _lib.PLACE_OUT)
    rc = s.lib.mmp_place_batch(s.h, _lib.ptr(bad), len(bad), _lib.ptr(extra), len(extra), fleet.now, _lib.ptr(out))
    assert rc == EINVAL, rc
    caller, rc_rows = _lib.split_caller(_one_caller(fleet, bad))
    cp = np.ascontiguousarray(caller, dtype=_lib.PLACE_CALLER).reshape(1)
    assert s.lib.mmp_place_batch_c(s.h, _lib.ptr(cp), _lib.ptr(rc_rows), len(rc_rows), _lib.ptr(extra), len(extra), fleet.now, _lib.ptr(out)) == EINVAL
    # the bounded device-pointer call: the request is refused by the kernel, its neighbours are decided
    dev = torch.device("cuda", 0)
    d_r = torch.from_numpy(bad.view(np.uint8).reshape(-1)).to(dev)
    d_e = torch.from_numpy(extra).to(dev)
    d_o = torch.zeros(len(bad) * 16, dtype=torch.uint8, device=dev)
    st_ = torch.cuda.Stream(dev)
    s.place_dev2(d_r.data_ptr(), len(bad), d_e.data_ptr(), len(extra), fleet.now, d_o.data_ptr(), st_.cuda_stream)
    torch.cuda.synchronize()
    got = np.frombuffer(d_o.cpu().numpy().tobytes(), dtype=_lib.PLACE_OUT)
    assert (got["chosen"][17], got["best"][17], got["n_candidates"][17], got["hash"][17]) == (-1, BAD_REQUEST, 0, 0)
    want = orc.place(reqs, extra, fleet.now)
    keep = np.arange(len(bad)) != 17
    for f in ("chosen", "best", "n_candidates", "hash"):
        assert np.array_equal(got[f][keep], want[f][keep]), f
    s.lib.mmp_stream_retire(s.h, C.c_void_p(st_.cuda_stream))


def _one_caller(fleet, reqs):
    out = reqs.copy()
    row = fleet.pods[3]
    out["self_pod"], out["flags"], out["fresh_rpm"] = 3, 0, 0
    out["fresh_lru"], out["fresh_capacity"], out["fresh_used"], out["fresh_count"] = row["lru_time"], row["capacity"], row["used"], row["count"]
    return out


def test_null_pointers_and_negative_counts_are_refused(ctx):
    s, fleet, _ = ctx
    L = s.lib
    reqs, extra = wl.make_requests(fleet, 1, n=8)
    out = np.zeros(8, _lib.PLACE_OUT)
    null = None
    assert L.mmp_place_batch(None, _lib.ptr(reqs), 8, null, 0, fleet.now, _lib.ptr(out)) == EINVAL           # no context
    assert L.mmp_place_batch(s.h, null, 8, null, 0, fleet.now, _lib.ptr(out)) == EINVAL                      # no requests
    assert L.mmp_place_batch(s.h, _lib.ptr(reqs), 8, null, 0, fleet.now, null) == EINVAL                     # no result rows
    assert L.mmp_place_batch(s.h, _lib.ptr(reqs), -1, null, 0, fleet.now, _lib.ptr(out)) == EINVAL           # negative n
    assert L.mmp_place_batch(s.h, _lib.ptr(reqs), 8, null, 5, fleet.now, _lib.ptr(out)) == EINVAL            # a pool length without a pool
    assert L.mmp_place_batch(s.h, _lib.ptr(reqs), 8, null, -5, fleet.now, _lib.ptr(out)) == EINVAL
    assert L.mmp_place_batch(s.h, null, 0, null, 0, fleet.now, null) == 0                                    # nothing to do is not an error
    assert L.mmp_place_batch_dev(s.h, null, 4, null, fleet.now, null, null) == EINVAL
    assert L.mmp_place_batch_dev(s.h, null, -4, null, fleet.now, null, null) == EINVAL
    assert L.mmp_place_batch_dev2(s.h, null, 4, null, -1, fleet.now, null, null) == EINVAL
    assert L.mmp_place_batch_c(s.h, null, _lib.ptr(reqs), 8, null, 0, fleet.now, _lib.ptr(out)) == EINVAL    # no caller
    assert L.mmp_pods_upsert(s.h, null, null, 3) == EINVAL
    assert L.mmp_pods_upsert(s.h, null, null, -3) == EINVAL
    assert L.mmp_pods_remove(s.h, null, 2) == EINVAL
    assert L.mmp_models_upsert(s.h, null, null, 2, null, null, 0) == EINVAL
    assert L.mmp_models_load(s.h, null, 5, null, null, 0) == EINVAL
    assert L.mmp_models_load(s.h, null, -5, null, null, 0) == EINVAL
    n_out = C.c_int32(0)
    assert L.mmp_get_order(s.h, null, C.byref(n_out)) == EINVAL
    assert L.mmp_shortlists(s.h, null, 4, C.byref(n_out)) == EINVAL
    assert L.mmp_shortlists(s.h, null, -4, C.byref(n_out)) == EINVAL
    assert L.mmp_long_shortlists(s.h, null, 4, C.byref(n_out)) == EINVAL
    assert L.mmp_long_shortlists(s.h, null, -4, C.byref(n_out)) == EINVAL
    assert L.mmp_long_shortlists(s.h, null, 0, None) == EINVAL
    assert L.mmp_long_shortlists(None, null, 0, C.byref(n_out)) == EINVAL
    assert L.mmp_split_batches(None, None, None) == EINVAL
    assert L.mmp_cluster_stats(s.h, null) == EINVAL
    assert L.mmp_snapshot_commit(None) == EINVAL
    assert L.mmp_issue_threads(None, 2) == EINVAL
    idx = np.array([0, 5, fleet.n_pods + 1, -1], np.int32)  # rows that do not exist (== n_pods would append)
    rows = fleet.pods[:4].copy()
    assert L.mmp_pods_upsert(s.h, _lib.ptr(idx), _lib.ptr(rows), 4) == EINVAL
    assert L.mmp_pods_remove(s.h, _lib.ptr(idx), 4) == EINVAL
    midx = np.array([0, fleet.n_models + 1], np.int32)  # (== n_models appends; beyond that is refused)
    mrows = fleet.models[:2].copy()
    mrows["n_loaded"], mrows["n_failed"], mrows["ent_off"] = 0, 0, 0
    assert L.mmp_models_upsert(s.h, _lib.ptr(midx), _lib.ptr(mrows), 2, null, null, 0) == EINVAL
    mrows["n_loaded"] = 3  # entries the call does not bring
    midx[1] = 1
    assert L.mmp_models_upsert(s.h, _lib.ptr(midx), _lib.ptr(mrows), 2, null, null, 0) == EINVAL
    # the context still answers
    got = s.place(reqs, extra, fleet.now)
    _well_formed(got, fleet.n_pods)


def test_decisions_before_a_commit_are_a_state_error():
    fleet = wl.make_fleet("C1")
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        reqs, extra = wl.make_requests(fleet, 1, n=4)
        out = np.zeros(4, _lib.PLACE_OUT)
        assert s.lib.mmp_place_batch(s.h, _lib.ptr(reqs), 4, None, 0, fleet.now, _lib.ptr(out)) in (ESTATE, EINVAL)
        n_out = C.c_int32(0)
        order = np.zeros(8, np.int32)
        assert s.lib.mmp_get_order(s.h, _lib.ptr(order), C.byref(n_out)) == ESTATE
    finally:
        s.close()
