"""Two more nets under the parity-unpinned half of the path (comparator, filter, breaks, rpm filter):

* gcov branch coverage of the C restatement of getNext under the golden / scenario / fuzz fleets must be complete
  (tools/oracle_coverage.py; tests/golden/make_golden.py refuses to write place_fuzz.npz otherwise);
* a hypothesis-driven differential test of the C restatement against the independently written Python restatement
  on tiny adversarial tables — values drawn from the thresholds the Java compares against — with shrinking, so that a
  disagreement arrives as a minimal table instead of a 200-instance fuzz fleet."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Fleet, bitmap_from_bool
from tests.test_oracle_cross import _compare

NOW = wl.NOW_MS
LONG_MAX = np.iinfo(np.int64).max


def test_getnext_restatement_has_full_branch_coverage():
    import shutil
    if not (shutil.which("gcc") and shutil.which("gcov")):
        pytest.skip("gcc / gcov not available")
    from tools.oracle_coverage import uncovered
    miss = uncovered()
    assert not miss, "\n".join(f"oracle/mm_oracle.c:{ln}: {note}: {text}" for ln, text, note in miss)


AGES = [1_000, 44_000, 46_000, 119_000, 121_000, 599_000, 1_201_000, 3_600_000, 86_400_000]
pod = st.tuples(
    st.sampled_from([0, 1, 8, 9, 10, 11, 12, 13, 14, 40]),             # count (the count break sits at 10 and f + f/4)
    st.sampled_from([0, 999, 1000, 1001, 3999, 4000, 4001, 250_000]),  # remaining around minSpace = 1000 and rem >> 2
    st.sampled_from(AGES),                                              # lruTime age
    st.sampled_from([0, 99, 100, 101, 150, 151, 300, 301, 400, 401]),   # rpm around the 1.1x / 1.5x / 3x / 4x rules
    st.sampled_from([1, 1, 1, 2]),                                      # instanceVersion
    st.integers(0, 3),                                                  # loadingInProgress
    st.booleans(), st.booleans(), st.booleans(),                        # allowed, preferred, live
    st.sampled_from([0, 0, 0, 1, 2]))                                   # flags: present / shutting down / tombstone
req = st.tuples(st.integers(-1, 7), st.booleans(), st.integers(0, 2**32 - 1),
                st.sampled_from([0, 500, 4_000, 6_000, 700_000, 800_000, 80_000_000, 90_000_000, 433_000_000, -20_000]),
                st.lists(st.integers(0, 7), max_size=3),                # tried / explicit exclusions
                st.sampled_from([0, 1, 2]),                             # how the caller's fresh row drifted
                st.lists(st.integers(0, 7), max_size=3, unique=True),   # instanceIds of the model
                st.lists(st.integers(0, 7), max_size=2, unique=True))   # loadFailedInstanceIds


@settings(max_examples=1000, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(st.lists(pod, min_size=1, max_size=8), st.lists(req, min_size=1, max_size=6), st.booleans(), st.booleans(),
       st.sampled_from([(), (0,), (0, 1)]))
def test_c_and_python_restatements_agree_on_tiny_adversarial_tables(pods, reqs_in, has_allowed, has_prefer, replaced):
    P = len(pods)
    rows = np.zeros(P, dtype=wl.POD_ROW)
    al, pf = np.ones((2, P), bool), np.zeros((2, P), bool)
    for i, (cnt, rem, age, rpm, ver, lip, allowed, preferred, live, fl) in enumerate(pods):
        rows[i]["capacity"], rows[i]["used"] = 1_000_000, 1_000_000 - rem
        rows[i]["count"], rows[i]["rpm"], rows[i]["version"] = cnt, rpm, ver
        rows[i]["lru_time"] = LONG_MAX if cnt == 0 else NOW - age
        rows[i]["loading_threads"], rows[i]["loading_in_progress"] = 8, lip
        rows[i]["id_order"], rows[i]["replica_set"] = i, i % 3
        rows[i]["flags"] = (wl.POD_LIVE if live else 0) | (wl.POD_SHUTTING_DOWN if fl == 1 else 0) | (wl.POD_TOMBSTONE if fl == 2 else 0)
        al[1, i], pf[1, i] = allowed, preferred
    # a full instance with lruTime <= 2 * minChurnAgeMs next to differing versions makes PLACEMENT_ORDER cyclic
    # (SURVEY.md §7 "Transitivity"): outside the contract of both restatements
    if len({int(v) for v in rows["version"]}) > 1:
        rows["version"] = 1
    n = len(reqs_in)
    models = np.zeros(n, dtype=wl.MODEL_ROW)
    reqs = np.zeros(n, dtype=wl.PLACE_REQ)
    ent, extra = [], []
    for i, (self_pod, favour, pick, ago, tried, drift, loaded, failed) in enumerate(reqs_in):
        loaded = [p for p in loaded if p < P]
        failed = [p for p in failed if p < P and p not in loaded]
        tried = [p for p in tried if p < P]
        models[i]["type"], models[i]["ent_off"] = 1, len(ent)
        models[i]["n_loaded"], models[i]["n_failed"] = len(loaded), len(failed)
        ent += sorted(loaded) + sorted(failed)
        sp = min(self_pod, P - 1)
        reqs[i]["model"], reqs[i]["self_pod"], reqs[i]["flags"], reqs[i]["pick"] = i, sp, int(favour), pick
        reqs[i]["last_used"] = 0 if ago == 0 else NOW - ago
        reqs[i]["extra_off"], reqs[i]["n_extra"] = len(extra), len(tried)
        extra += tried
        row = rows[max(sp, 0)]
        reqs[i]["fresh_lru"], reqs[i]["fresh_capacity"] = row["lru_time"], row["capacity"]
        reqs[i]["fresh_used"] = row["used"] + (0, 500, 300_000)[drift]
        reqs[i]["fresh_count"] = row["count"] + drift
    fleet = Fleet(pods=rows, models=models, ent_pod=np.array(ent, np.int32), ent_time=np.full(len(ent), NOW - 5000, np.int64),
                  min_space_units=1000, min_churn_age_ms=600_000, now=NOW, n_types=2,
                  allowed=bitmap_from_bool(al), prefer=bitmap_from_bool(pf),
                  has_allowed=np.array([0, int(has_allowed)], np.uint8), has_prefer=np.array([0, int(has_prefer)], np.uint8),
                  replaced_rs=np.array(replaced, np.int32))
    _compare(fleet, reqs, np.array(extra, np.int32))
