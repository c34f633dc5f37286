"""GPU: the single-caller request form (mmp_place_batch_c / _c_dev: the caller's side once per call, 24 bytes per decision —
include/mmplace.h) decides exactly what the same decisions decide as 64-byte mmp_place_req rows, on every kernel that carries
the form (window path, prefix-table path in both instantiations, wave path), through the host-pointer call (small batches ride
the latency path as expanded rows, large ones the 24-byte kernels) and the device-pointer call; and the BOUNDED device-pointer
calls (mmp_place_batch_dev2, mmp_place_batch_c_dev) answer a request whose exclusion range leaves the declared pool with
MMP_BAD_REQUEST instead of following it."""
import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl
from modelmesh_amd._lib import PLACE_OUT, PLACE_REQ_C
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet

pytestmark = pytest.mark.gpu
FIELDS = ("chosen", "best", "n_candidates", "hash")


def caller_batches(fleet, seed, n, k=4):
    """k callers of one fleet (in the table and not, favouring themselves and not, fresh rows equal to / drifted from their
    snapshot rows), n requests each: (caller, reqs_c, extra)."""
    rng = np.random.default_rng(seed)
    out = []
    for j in range(k):
        reqs, extra = wl.fuzz_requests(fleet, seed * 31 + j, n)
        sp = int(rng.integers(0, fleet.n_pods)) if j != 2 else -1
        row = fleet.pods[max(sp, 0)]
        reqs["self_pod"] = sp
        reqs["flags"] = j & 1
        reqs["fresh_lru"] = row["lru_time"] if j % 3 else fleet.now - 50_000
        reqs["fresh_capacity"] = row["capacity"]
        reqs["fresh_used"] = row["used"] if j % 2 else int(row["capacity"] * 0.2)
        reqs["fresh_count"] = int(row["count"]) + j
        reqs["fresh_rpm"] = [0, 120, 0, 400][j % 4]
        caller, rc = _lib.split_caller(reqs)
        out.append((caller, rc, extra, reqs))
    return out


@pytest.mark.parametrize("seed", [2, 9, 17, 40])
@pytest.mark.parametrize("profile", [None, "full", "prefer"])
def test_single_caller_form_equals_the_request_rows_and_the_oracle(seed, profile):
    fleet = wl.fuzz_fleet(seed, pods=300, models=400, profile=profile)
    orc = OracleFleet(fleet)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        for n in (700, 5000):  # the latency path (expanded rows) / the 24-byte kernels
            for caller, rc, extra, reqs in caller_batches(fleet, seed, n):
                want = orc.place(reqs, extra, fleet.now, threads=8)
                got = s.place_c(caller, rc, extra, fleet.now)
                rows = s.place(reqs, extra, fleet.now)
                for f in FIELDS:
                    assert np.array_equal(got[f], want[f]), (n, f)
                    assert np.array_equal(got[f], rows[f]), (n, f)
    finally:
        s.close()


def _dev(arr):
    import torch
    a = np.ascontiguousarray(arr)
    if a.size == 0:
        a = np.zeros(4, np.uint8)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(torch.device("cuda", 0))


@pytest.mark.parametrize("full", [False, True])
def test_single_caller_form_on_c2_through_the_device_pointer_call(full):
    """C2, 21 request sets (210k decisions: past the size from which a full cluster takes the four-wavefront instantiation) of one
    caller, device pointers; the full-cluster fleet takes the prefix-table kernels."""
    import torch
    fleet = wl.make_fleet("C2")
    if full:
        wl.make_full_cluster(fleet)
    parts = [wl.make_requests(fleet, 300 + i) for i in range(21)]
    reqs = np.concatenate([p[0] for p in parts])
    extra = np.concatenate([p[1] for p in parts])
    base = 0
    for i, p in enumerate(parts):
        reqs["extra_off"][i * fleet.n_models:(i + 1) * fleet.n_models] += base
        base += len(p[1])
    sp = 17
    row = fleet.pods[sp]
    reqs["self_pod"], reqs["flags"] = sp, 0
    reqs["fresh_lru"], reqs["fresh_capacity"], reqs["fresh_used"] = row["lru_time"], row["capacity"], row["used"] + 1000
    reqs["fresh_count"], reqs["fresh_rpm"] = row["count"], 0
    caller, rc = _lib.split_caller(reqs)
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=16)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        d_r, d_x = _dev(rc), _dev(extra)
        for n in (len(rc), 100_000):
            d_o = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
            st = torch.cuda.Stream()
            s.place_c_dev(caller, d_r.data_ptr(), n, d_x.data_ptr(), len(extra), fleet.now, d_o.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            got = np.frombuffer(d_o.cpu().numpy().tobytes(), dtype=PLACE_OUT)
            for f in FIELDS:
                assert np.array_equal(got[f], want[f][:n]), (n, f)
    finally:
        s.close()


@pytest.mark.parametrize("form", ["rows", "caller"])
def test_bounded_device_calls_refuse_requests_that_leave_the_pool(form):
    import torch
    fleet = wl.fuzz_fleet(6, pods=200, models=300)
    reqs, extra = wl.fuzz_requests(fleet, 3, 6000)
    sp = 5
    row = fleet.pods[sp]
    if form == "caller":
        reqs["self_pod"], reqs["flags"] = sp, 1
        reqs["fresh_lru"], reqs["fresh_capacity"], reqs["fresh_used"] = row["lru_time"], row["capacity"], row["used"]
        reqs["fresh_count"], reqs["fresh_rpm"] = row["count"], 0
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
    rng = np.random.default_rng(5)
    bad = np.sort(rng.choice(len(reqs), 40, replace=False))
    poisoned = reqs.copy()
    kinds = rng.integers(0, 4, len(bad))
    poisoned["n_extra"][bad] = np.where(kinds == 0, -1, np.where(kinds == 1, 3, np.where(kinds == 2, 2**30, 1)))
    poisoned["extra_off"][bad] = np.where(kinds == 1, len(extra) - 2, np.where(kinds == 3, -7, poisoned["extra_off"][bad]))
    poisoned["extra_off"][bad[kinds == 2]] = 2**30  # offset + count overflows an int
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        d_x = _dev(extra)
        d_o = torch.zeros(len(reqs) * 16, dtype=torch.uint8, device="cuda:0")
        if form == "rows":
            d_r = _dev(poisoned)
            s.place_dev2(d_r.data_ptr(), len(reqs), d_x.data_ptr(), len(extra), fleet.now, d_o.data_ptr())
        else:
            caller, rc = _lib.split_caller(poisoned)
            d_r = _dev(rc)
            s.place_c_dev(caller, d_r.data_ptr(), len(reqs), d_x.data_ptr(), len(extra), fleet.now, d_o.data_ptr())
        torch.cuda.synchronize()
        got = np.frombuffer(d_o.cpu().numpy().tobytes(), dtype=PLACE_OUT)
        good = np.ones(len(reqs), bool)
        good[bad] = False
        for f in FIELDS:
            assert np.array_equal(got[f][good], want[f][good]), f
        assert np.all(got["chosen"][bad] == _lib.MMP_NONE) and np.all(got["best"][bad] == _lib.MMP_BAD_REQUEST)
        assert np.all(got["n_candidates"][bad] == 0) and np.all(got["hash"][bad] == 0)
    finally:
        s.close()


def test_pool_of_zero_entries_and_host_side_validation():
    """A declared pool of zero entries refuses every request that names exclusions; the host-pointer call validates the ranges itself."""
    import torch
    fleet = wl.fuzz_fleet(8, pods=100, models=100)
    reqs, extra = wl.fuzz_requests(fleet, 1, 2000)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        d_r = _dev(reqs)
        d_o = torch.zeros(len(reqs) * 16, dtype=torch.uint8, device="cuda:0")
        s.place_dev2(d_r.data_ptr(), len(reqs), 0, 0, fleet.now, d_o.data_ptr())
        torch.cuda.synchronize()
        got = np.frombuffer(d_o.cpu().numpy().tobytes(), dtype=PLACE_OUT)
        has = reqs["n_extra"] > 0
        assert has.any() and np.all(got["best"][has] == _lib.MMP_BAD_REQUEST)
        clean = reqs[~has]
        want = OracleFleet(fleet).place(clean, None, fleet.now)
        for f in FIELDS:
            assert np.array_equal(got[f][~has], want[f]), f
        rc = np.zeros(3, dtype=PLACE_REQ_C)
        rc["n_extra"], rc["extra_off"] = 2, len(extra) - 1
        caller = np.zeros(1, dtype=_lib.PLACE_CALLER)
        with pytest.raises(Exception) as ei:
            s.place_c(caller, rc, extra, fleet.now)
        assert getattr(ei.value, "code", None) == _lib.MMP_EINVAL
    finally:
        s.close()
