"""GPU: mmp_place_multi_dev — several request arrays decided by ONE launch (multi_kernel.hpp) — equals the separate calls and the
oracle: ragged array sizes (not multiples of a workgroup), an empty array in the middle, arrays with and without exclusion pools of
their own, more arrays than one launch takes, and the full-cluster (long) kernels."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd._lib import PLACE_OUT
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet

pytestmark = pytest.mark.gpu


def _dev(arr):
    import torch
    a = np.ascontiguousarray(arr)
    if a.size == 0:
        a = np.zeros(1, a.dtype if a.dtype != np.dtype(object) else np.int32)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(torch.device("cuda", 0))


@pytest.mark.parametrize("memo_from", [None, "0"], ids=["default", "MMP_MEMO_FROM=0"])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("sizes", [[1, 255, 256, 257, 0, 3000], [5000] * 19, [100_000, 17, 100_000]])
def test_multi_equals_the_separate_calls_and_the_oracle(full, sizes, memo_from, monkeypatch):
    """(MMP_MEMO_FROM=0: launches of every size through place_multi_m_kernel, the per-type shortlists checked first)"""
    import torch
    if memo_from is not None:
        monkeypatch.setenv("MMP_MEMO_FROM", memo_from)
    fleet = wl.make_fleet("C2")
    if full:
        wl.make_full_cluster(fleet)
    orc = OracleFleet(fleet)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        batches = []
        for i, n in enumerate(sizes):
            reqs, extra = wl.make_requests(fleet, 100 + i, n=max(n, 1), extra_frac=0.0 if i % 3 == 2 else 0.05)
            batches.append((reqs[:n], extra))
        d_reqs = [_dev(r) for r, _ in batches]
        d_extra = [_dev(x) for _, x in batches]
        d_outs = [torch.zeros(max(len(r), 1) * 16, dtype=torch.uint8, device="cuda:0") for r, _ in batches]
        st = torch.cuda.Stream()
        s.place_multi_dev([t.data_ptr() for t in d_reqs], [len(r) for r, _ in batches],
                          [t.data_ptr() if len(x) else 0 for t, (_, x) in zip(d_extra, batches)], fleet.now,
                          [t.data_ptr() for t in d_outs], st.cuda_stream)
        torch.cuda.synchronize()
        for i, (reqs, extra) in enumerate(batches):
            if len(reqs) == 0:
                continue
            got = np.frombuffer(d_outs[i].cpu().numpy().tobytes(), dtype=PLACE_OUT)[:len(reqs)]
            sep = s.place(reqs, extra, fleet.now)
            want = orc.place(reqs, extra, fleet.now, threads=8)
            for f in ("chosen", "best", "n_candidates", "hash"):
                assert np.array_equal(got[f], sep[f]), (i, f)
                assert np.array_equal(got[f], want[f]), (i, f)
    finally:
        s.close()
