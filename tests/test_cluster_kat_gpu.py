"""GPU twin of test_cluster_kat.py: the reference's cluster tests (ModelMeshEvictionsTest.java :292-310,
:323-358, :371-409) with the DEVICE (libmmplace through the C ABI: mmp_place_batch, mmp_gate_batch,
mmp_cache_replay, mmp_pods_upsert / mmp_models_upsert / mmp_snapshot_commit) and the CPU oracle replaying
the same mesh in lock step — tests/minimesh.py asserts after every decision, guard and cache operation that
both took the same step; the reference's own assertions are then made on the shared outcome."""
import pytest

from tests.minimesh import DeviceBackend, MiniMesh, OracleBackend
from tests.test_cluster_kat import CLUSTER, SCENARIOS, skewed_ingress

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scenario", SCENARIOS, ids=lambda f: f.__name__)
@pytest.mark.parametrize("seed", [0, 1])
def test_cluster_kat_device(scenario, seed):
    trace = scenario([DeviceBackend, OracleBackend], seed)
    assert trace == scenario([OracleBackend], seed)


@pytest.mark.parametrize("ingress", ["random", "single"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_skewed_ingress_device(seed, ingress):
    """Requests arriving unevenly: decisions are forwarded (chosen >= 0), the shortlists differ per step."""
    trace = skewed_ingress([DeviceBackend, OracleBackend], seed, ingress)
    assert any(chosen >= 0 for _, _, chosen, _, _ in trace)


def test_device_alone_meets_the_reference_assertions():
    """No oracle in the loop: the device's own decisions and caches satisfy :323-358."""
    mesh = MiniMesh(CLUSTER, [DeviceBackend], 3)
    try:
        for _ in range(30):
            mesh.add_model()
        assert set(range(9, 30)) <= set(mesh.loaded())
    finally:
        mesh.close()
