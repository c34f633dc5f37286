"""The bench line is a contract with the driver and the judge (task statement ④): guard its schema against
regressions on the committed sample of the last GPU run (profiles/r*/bench_C3_n1.json.log), and the pure
helpers of bench.py on CPU."""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_line():
    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_C3_n1.json.log")))
    assert logs, "no committed bench line"
    lines = [ln for ln in open(logs[-1]) if ln.startswith("{")]
    return json.loads(lines[-1])


def test_committed_bench_line_has_the_contract_fields():
    d = _last_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].replace("x", "×") == base["metric"].replace("x", "×")
    assert d["unit"] == "decisions/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "i64"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == 1 and abs(d["value"] - d["config"]["decisions_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    if "bytes_per_launch" in r:  # round 2 on: frac is a fraction of the HBM peak on the bytes the kernel moves
        assert abs(r["achieved"] - r["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        assert 0.0 < r["frac"] <= 1.0
        assert r["bytes_per_launch"] == (r["traffic"] if r["traffic"] else r["kernel_bytes_per_launch"])
        assert r["scan_equivalent"]["algorithmic_bytes_per_launch"] > r["bytes_per_launch"]
        assert d["config"]["resident_input_bytes"] > 256 * 2**20  # the rotating batches do not fit the Infinity Cache
    else:  # the round-1 line: SURVEY 8(d) scan-equivalent bytes / kernel time
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "decisions/s" and c["cores"] >= 1 and c["sample"]
    assert d["parity_vs_oracle"] is True


def test_algorithmic_bytes_follow_survey_8d():
    """bytes(d) = 32*P + (24 + 12*(k+f) + 4*e) + 16."""
    import bench
    from modelmesh_amd import workload as wl
    fleet = wl.make_fleet("C1")
    reqs, extra = wl.make_requests(fleet, 3)
    m = fleet.models[reqs["model"]]
    want = sum(32 * fleet.n_pods + 24 + 12 * (int(a) + int(b)) + 4 * int(e) + 16
               for a, b, e in zip(m["n_loaded"], m["n_failed"], reqs["n_extra"]))
    assert bench.algorithmic_bytes(fleet, reqs) == want
    assert bench.kernel_bytes(fleet, reqs) < want
    assert bench.HBM_PEAK_GBS == 8000.0


def test_a_step_batch_of_several_request_sets_decides_like_its_sets():
    """bench.make_step_batch concatenates request sets and moves their exclusion-pool offsets: the oracle's decisions on
    the batch are the decisions on the sets, in order (what the GPU parity gate of bench.py then compares against)."""
    import bench
    from modelmesh_amd import workload as wl
    from oracle.bind import OracleFleet
    fleet = wl.make_fleet("C1")
    seeds = [11, 12, 13]
    rq, ex, pool0 = bench.make_step_batch(fleet, seeds)
    assert len(rq) == 3 * fleet.n_models
    orc = OracleFleet(fleet)
    whole = orc.place(rq, ex, fleet.now)
    for j, seed in enumerate(seeds):
        r1, e1 = wl.make_requests(fleet, seed=seed)
        if j == 0:
            assert pool0 == len(e1) and np.array_equal(rq[:len(r1)], r1)
        one = orc.place(r1, e1, fleet.now)
        sl = whole[j * fleet.n_models:(j + 1) * fleet.n_models]
        for f in ("chosen", "best", "n_candidates", "hash"):
            assert np.array_equal(sl[f], one[f]), (j, f)
    assert (rq["n_extra"] > 0).any(), "the sets carry late-bound exclusions, or the offsets were never exercised"
