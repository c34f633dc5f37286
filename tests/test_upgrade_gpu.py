"""Row a19: the library's UpgradeTracker (host-side state of the context) against the Python
restatement of UpgradeTracker.java:85-200 on random rolling-update event streams, and its effect on
load-target selection (pods of a likely-replaced replica set are avoided unless nothing else is
eligible, MM.java:4769-4770,4792-4805)."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet
from oracle.py_upgrade import UpgradeTracker
from tests.util import assert_same_decisions

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(6))
def test_tracker_matches_restatement(seed):
    rng = np.random.default_rng(300 + seed)
    s = Solver(100, 1000)
    ref = UpgradeTracker()
    try:
        now = 1_760_000_000_000
        starts = {}  # replica set -> base start time (distinct per set, see the header on ties)
        members = []
        for step in range(500):
            now += int(rng.choice([50, 5_000, 60_000, 400_000]))
            r = rng.random()
            if r < 0.55 or not members:
                lk = int(rng.choice([0, 0, 0, 7]))
                rs = int(rng.choice([-1, 0, 1, 2, 3, 4]))
                if rs not in starts:
                    starts[rs] = now - int(rng.integers(0, 3_000_000)) * 7 - rs
                st = starts[rs] + int(rng.integers(0, 100_000)) * 11
                s.upgrade_instance_added(lk, rs, st, now)
                ref.instanceAdded(lk, rs, st, now)
                members.append((lk, rs))
            elif r < 0.9:
                lk, rs = members.pop(int(rng.integers(0, len(members))))
                if rng.random() < 0.1:
                    lk += 100  # a re-deserialised record: different labels array identity
                s.upgrade_instance_removed(lk, rs, now)
                ref.instanceRemoved(lk, rs, now)
            else:
                s.upgrade_housekeeping(now)
                ref.doHousekeeping(now)
            assert s.upgrade_replaced() == ref.likelyReplacedReplicaSets, (seed, step)
    finally:
        s.close()


def test_replaced_replica_sets_steer_placement():
    fleet = wl.fuzz_fleet(77, pods=400)
    fleet.replaced_rs = np.zeros(0, np.int32)
    reqs, extra = wl.fuzz_requests(fleet, 77, 1500)
    now = fleet.now
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        # a rolling update: replica set 0 started long ago, replica set 1 is starting now
        for _ in range(3):
            s.upgrade_instance_added(0, 0, now - 86_400_000, now - 86_400_000)
        s.upgrade_instance_added(0, 1, now - 60_000, now - 60_000)
        s.upgrade_instance_removed(0, 0, now - 30_000)
        s.upgrade_instance_added(0, 1, now - 20_000, now - 20_000)
        assert set(s.upgrade_replaced()) == {0}
        s.commit()
        fleet.replaced_rs = np.array([0], np.int32)
        want = OracleFleet(fleet).place(reqs, extra, now, threads=4)
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, now), want)
    finally:
        s.close()
