"""ctypes binding of libmmplace (include/mmplace.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no Python or CPU implementation behind this module: if the shared
object is missing, import of the solver fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMP_LIB_PATH") or os.path.join(HERE, "lib", "libmmplace.so")

MMP_OK = 0
MMP_EINVAL, MMP_ENODEVICE, MMP_EHIP, MMP_EORDER, MMP_ESTATE, MMP_ENOMEM = -1, -2, -3, -4, -5, -6
MMP_NONE, MMP_SELF = -1, -2
POD_SHUTTING_DOWN, POD_LIVE, POD_TOMBSTONE = 1, 2, 4
REQ_FAVOUR_SELF = 1
SERVE_EXCLUDE_SELF, SERVE_PREFER_SELF = 1, 2
ANY_TIME = -(2**63)
JAVA_LONG_MAX = 2**63 - 1

# numpy mirrors of the C structs (all naturally aligned, no padding surprises)
POD_ROW = np.dtype(
    [("lru_time", "<i8"), ("capacity", "<i8"), ("used", "<i8"), ("version", "<i8"),
     ("count", "<i4"), ("loading_threads", "<i4"), ("loading_in_progress", "<i4"), ("rpm", "<i4"),
     ("id_order", "<u4"), ("replica_set", "<i4"), ("flags", "<u4"), ("reserved", "<u4")])
MODEL_ROW = np.dtype(
    [("type", "<i4"), ("ent_off", "<i4"), ("n_loaded", "<i4"), ("n_failed", "<i4"), ("last_used", "<i8")])
PLACE_REQ = np.dtype(
    [("model", "<i4"), ("self_pod", "<i4"), ("flags", "<u4"), ("pick", "<u4"), ("last_used", "<i8"),
     ("extra_off", "<i4"), ("n_extra", "<i4"), ("fresh_lru", "<i8"), ("fresh_capacity", "<i8"),
     ("fresh_used", "<i8"), ("fresh_count", "<i4"), ("fresh_rpm", "<i4")])
PLACE_OUT = np.dtype([("chosen", "<i4"), ("best", "<i4"), ("n_candidates", "<i4"), ("hash", "<u4")])
# the single-caller form: the caller's side once per call, 24 bytes per decision
PLACE_CALLER = np.dtype([("self_pod", "<i4"), ("flags", "<u4"), ("fresh_lru", "<i8"), ("fresh_capacity", "<i8"), ("fresh_used", "<i8"),
                         ("fresh_count", "<i4"), ("fresh_rpm", "<i4")])
PLACE_REQ_C = np.dtype([("model", "<i4"), ("pick", "<u4"), ("last_used", "<i8"), ("extra_off", "<i4"), ("n_extra", "<i4")])
assert PLACE_CALLER.itemsize == 40 and PLACE_REQ_C.itemsize == 24
MMP_BAD_REQUEST = -3


def split_caller(reqs):
    """mmp_place_req rows that share one caller -> (PLACE_CALLER[1], PLACE_REQ_C[n]); asserts that they do."""
    reqs = np.ascontiguousarray(reqs, dtype=PLACE_REQ)
    caller = np.zeros(1, dtype=PLACE_CALLER)
    for f in PLACE_CALLER.names:
        assert len(reqs) == 0 or np.all(reqs[f] == reqs[f][0]), f
        if len(reqs):
            caller[f] = reqs[f][0]
    rc = np.zeros(len(reqs), dtype=PLACE_REQ_C)
    for f in PLACE_REQ_C.names:
        rc[f] = reqs[f]
    return caller, rc


def join_caller(caller, reqs_c):
    """The same decisions as mmp_place_req rows."""
    caller = np.asarray(caller).reshape(-1)[0]
    out = np.zeros(len(reqs_c), dtype=PLACE_REQ)
    for f in PLACE_REQ_C.names:
        out[f] = reqs_c[f]
    for f in PLACE_CALLER.names:
        out[f] = caller[f]
    return out
SERVE_REQ = np.dtype(
    [("model", "<i4"), ("self_pod", "<i4"), ("flags", "<u4"), ("local_in_flight", "<i4"),
     ("last_invoke_time", "<i8"), ("assume_completed_ms", "<i8"), ("excl_off", "<i4"), ("n_excl", "<i4"),
     ("cnt_off", "<i4"), ("n_cnt", "<i4")])
SERVE_COUNTER = np.dtype([("pod", "<i4"), ("in_use", "<i4"), ("last_used", "<i8")])
SERVE_OUT = np.dtype([("chosen", "<i4"), ("pad", "<i4"), ("chosen_load_start", "<i8")])
STATS = np.dtype(
    [("total_capacity", "<i8"), ("total_free", "<i8"), ("global_lru", "<i8"),
     ("instance_count", "<i4"), ("model_copy_count", "<i4")])
EVICT_REQ = np.dtype([("cache", "<i4"), ("weight", "<i4"), ("last_used", "<i8")])
EVICT_OUT = np.dtype(
    [("insert_pos", "<i4"), ("n_victims", "<i4"), ("self_evicted", "<i4"), ("pad", "<i4"),
     ("weighted_size", "<i8"), ("oldest_time", "<i8")])

GATE_REQ = np.dtype(
    [("model", "<i4"), ("self_pod", "<i4"), ("flags", "<u4"), ("excl_off", "<i4"), ("n_excl", "<i4"),
     ("explicit_off", "<i4"), ("n_explicit", "<i4"), ("size_hint", "<i4"), ("last_used_time", "<i8"),
     ("cache_capacity", "<i8"), ("cache_weighted_size", "<i8"), ("cache_oldest_time", "<i8"),
     ("loader_predicted", "<i4"), ("loading_count", "<i4"), ("weight_predict_cutoff", "<i4"), ("reserved", "<i4"),
     ("loaded_time", "<i8"), ("load_timeout_ms", "<i8"), ("fresh_lru", "<i8"), ("fresh_capacity", "<i8"),
     ("fresh_used", "<i8"), ("fresh_count", "<i4"), ("fresh_loading_threads", "<i4"), ("fresh_in_progress", "<i4"),
     ("fresh_rpm", "<i4"), ("last_published", "<i8")])
GATE_OUT = np.dtype([("bits", "<u4"), ("initial_size", "<i4")])
assert GATE_REQ.itemsize == 144 and GATE_OUT.itemsize == 8
GATE_GO_LOCAL, GATE_FAILURES_BREACHED, GATE_LOCATIONS_BREACHED, GATE_LOCAL_NOT_ALLOWED = 1, 2, 4, 8
GATE_CHURN_REJECT, GATE_EARLY_REJECT, GATE_RELOAD_ELSEWHERE, GATE_SHOULD_PUBLISH = 16, 32, 64, 128

PROACTIVE_INFO = np.dtype(
    [("size_estimate", "<i4"), ("free_count", "<i4"), ("total_count", "<i4"), ("n_candidates", "<i4"),
     ("n_selected", "<i4"), ("error", "<i4"), ("space_to_fill", "<i8"), ("cutoff", "<i8")])
assert PROACTIVE_INFO.itemsize == 40

CACHE_ENTRY = np.dtype(
    [("model", "<i4"), ("weight", "<i4"), ("last_used", "<i8"), ("interval_count", "<i8"), ("last_heavy_time", "<i8"),
     ("last_unload_time", "<i8"), ("earlier_use_iteration", "<i4"), ("last_used_iteration", "<i4"), ("flags", "<u4"),
     ("reserved", "<i4")])
SCALEUP_PARAMS = np.dtype(
    [("self_pod", "<i4"), ("iteration_counter", "<i4"), ("second_copy_max_age_iters", "<i4"),
     ("second_copy_min_age_iters", "<i4"), ("scale_up_rpm_threshold", "<i4"), ("our_rpm", "<i4"), ("now", "<i8"),
     ("last_check_time", "<i8"), ("rate_check_interval_ms", "<i8"), ("second_copy_lru_threshold_ms", "<i8"),
     ("assume_completed_ms", "<i8")])
CONC_ENTRY = np.dtype([("count_and_time_sum", "<i8"), ("prior_sum", "<i8"), ("prior_count", "<i4"), ("max_conc", "<i4"),
                       ("queued_requests", "<i4"), ("reserved", "<i4")])
CONC_OUT = np.dtype([("threshold", "<i4"), ("reset", "<i4"), ("new_prior_sum", "<i8"), ("new_prior_count", "<i4"), ("reserved", "<i4")])
CONC_PARAMS = np.dtype([("dynamic_rpm_scale_constant", "<i8"), ("average_model_parallelism", "<f8")])
CONC_RESULT = np.dtype([("average_model_parallelism", "<f8"), ("exclude_set_rpms", "<i4"), ("model_parallelism_sum", "<i4")])
assert CONC_ENTRY.itemsize == 32 and CONC_OUT.itemsize == 24 and CONC_PARAMS.itemsize == 16 and CONC_RESULT.itemsize == 16
CONC_COUNT_BITS = 21
SCALEUP_OUT = np.dtype([("action", "<i4"), ("copies", "<i4"), ("timestamp", "<i8"), ("new_i1", "<i4"),
                        ("new_i2", "<i4"), ("heavy", "<i4"), ("rpm", "<i4")])
SCALEDOWN_PARAMS = np.dtype(
    [("self_pod", "<i4"), ("shutting_down", "<i4"), ("now", "<i8"), ("last_check_time", "<i8"),
     ("rate_check_interval_ms", "<i8"), ("adjusted_cache_capacity", "<i8"), ("scale_up_rpm_threshold", "<i4"),
     ("reserved", "<i4")])
assert CACHE_ENTRY.itemsize == 56 and SCALEUP_PARAMS.itemsize == 64 and SCALEUP_OUT.itemsize == 32
assert SCALEDOWN_PARAMS.itemsize == 48

CACHE_OP = np.dtype([("cache", "<i4"), ("op", "<i4"), ("key", "<i4"), ("arg", "<i4"), ("time", "<i8"), ("flag", "<i4"),
                     ("reserved", "<i4")])
CACHE_OP_OUT = np.dtype([("result", "<i4"), ("n_evicted", "<i4"), ("evicted_off", "<i4"), ("buffer_weight", "<i4"),
                         ("weighted_size", "<i8"), ("oldest_time", "<i8")])
UBM_STATE = np.dtype([("reserved", "<i4"), ("total_unloading", "<i4"), ("total_occupancy", "<i8"),
                      ("cache_deficit", "<i4"), ("pad", "<i4")])
assert CACHE_OP.itemsize == 32 and CACHE_OP_OUT.itemsize == 32 and UBM_STATE.itemsize == 24
UNLOADBUF_KEY = -1000000
(COP_PUT_IF_ABSENT, COP_GET, COP_UPDATE_WEIGHT, COP_REMOVE, COP_UBM_INSERT_NEW_ENTRY, COP_UBM_ADJUST_SPACE_REQUEST,
 COP_UBM_SPACE_IS_READY, COP_UBM_CLAIM_SPACE, COP_UBM_ADJUST_AFTER_LOAD, COP_UBM_UNLOAD_COMPLETE, COP_UBM_REMOVE_ENTRY,
 COP_UBM_DISCARD_FAILED, COP_UBM_INSERT_FAILED_PLACEHOLDER) = range(13)

assert POD_ROW.itemsize == 64 and MODEL_ROW.itemsize == 24 and PLACE_REQ.itemsize == 64
assert PLACE_OUT.itemsize == 16 and SERVE_REQ.itemsize == 48 and SERVE_COUNTER.itemsize == 16 and SERVE_OUT.itemsize == 16
assert STATS.itemsize == 32 and EVICT_REQ.itemsize == 16 and EVICT_OUT.itemsize == 32


class MmpConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("reserved0", C.c_int32),
                ("min_space_units", C.c_int64), ("min_churn_age_ms", C.c_int64)]


# every symbol include/mmplace.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("mmp_abi_version", C.c_int, []),
    ("mmp_create", C.c_int, [C.POINTER(MmpConfig), C.POINTER(_P)]),
    ("mmp_destroy", None, [_P]),
    ("mmp_last_error", C.c_char_p, [_P]),
    ("mmp_backend", C.c_int, [_P]),
    ("mmp_min_space_units", C.c_int64, [C.c_int32, C.c_int32, C.c_int64, C.c_int]),
    ("mmp_pods_load", C.c_int, [_P, _P, C.c_int32]),
    ("mmp_pods_upsert", C.c_int, [_P, _P, _P, C.c_int32]),
    ("mmp_pods_remove", C.c_int, [_P, _P, C.c_int32]),
    ("mmp_types_load", C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    ("mmp_types_from_labels", C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    ("mmp_replaced_rs_load", C.c_int, [_P, _P, C.c_int32]),
    ("mmp_upgrade_instance_added", C.c_int, [_P, C.c_int64, C.c_int32, C.c_int64, C.c_int64]),
    ("mmp_upgrade_instance_removed", C.c_int, [_P, C.c_int64, C.c_int32, C.c_int64]),
    ("mmp_upgrade_housekeeping", C.c_int, [_P, C.c_int64]),
    ("mmp_upgrade_replaced", C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    ("mmp_models_load", C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32]),
    ("mmp_models_upsert", C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_int32]),
    ("mmp_snapshot_commit", C.c_int, [_P]),
    ("mmp_get_order", C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    ("mmp_delta_commits", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("mmp_shortlists", C.c_int, [_P, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    ("mmp_long_shortlists", C.c_int, [_P, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    ("mmp_split_batches", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("mmp_place_batch_dev2", C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int64, _P, _P]),
    ("mmp_place_batch_c", C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int32, C.c_int64, _P]),
    ("mmp_place_batch_c_dev", C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int32, C.c_int64, _P, _P]),
    ("mmp_cluster_stats", C.c_int, [_P, _P]),
    ("mmp_type_stats", C.c_int, [_P, C.c_int32, _P]),
    ("mmp_partition_count", C.c_int, [_P, _P]),
    ("mmp_partition_stats", C.c_int, [_P, C.c_int32, _P, _P, C.c_int32]),
    ("mmp_pod_partitions", C.c_int, [_P, _P, C.c_int32, _P]),
    ("mmp_place_batch", C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int64, _P]),
    ("mmp_place_batch_dev", C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, _P, _P]),
    ("mmp_place_multi_dev", C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int64, _P, _P]),
    ("mmp_stream_retire", C.c_int, [_P, _P]),
    ("mmp_resident", C.c_int, [_P, C.c_int]),
    ("mmp_resident_stats", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("mmp_issue_threads", C.c_int, [_P, C.c_int32]),
    ("mmp_issue_flush", C.c_int, [_P]),
    ("mmp_serve_batch", C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_int64, _P]),
    ("mmp_caches_load", C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    ("mmp_evict_batch", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    ("mmp_caches_load_keyed", C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    ("mmp_cache_replay", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    ("mmp_cache_read", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64), _P]),
    ("mmp_gate_batch", C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int32, C.c_int64, C.c_int64, _P]),
    ("mmp_miss_batch", C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int64, C.c_int64, _P, _P]),
    ("mmp_route_batch", C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int32, C.c_int64, C.c_int64, _P, _P]),
    ("mmp_proactive_plan", C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P]),
    ("mmp_proactive_plan_subset", C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P]),
    ("mmp_scaleup_plan", C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32)]),
    ("mmp_scaledown_plan", C.c_int, [_P, _P, C.c_int32, _P, _P]),
    ("mmp_scaleup_plan_conc", C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, _P, _P, C.POINTER(C.c_int32), _P]),
    ("mmp_scaledown_plan_conc", C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int64, _P]),
    ("mmp_migration_plan", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P, _P]),
    ("mmp_pod_ids_load", C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    ("mmp_pods_ingest_json", C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    ("mmp_type_names_load", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32]),
    ("mmp_models_ingest_json", C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    ("mmp_pods_get", C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    ("mmp_models_get", C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("mmp_shard_configure", C.c_int, [_P, C.c_int32, C.c_int32]),
    ("mmp_shard_xchg_slots", C.c_int32, [C.c_int32, C.c_int32]),
    ("mmp_shard_xchg_is_sum", C.c_int32, [C.c_int32]),
    ("mmp_shard_rank_dev", C.c_int, [_P, _P]),
    ("mmp_shard_commit_dev", C.c_int, [_P, _P]),
    ("mmp_shard_place_phase_dev", C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int64, _P, _P, _P]),
    ("mmp_shard_fast_slots", C.c_int32, []),
    ("mmp_shard_place_fast_dev", C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, _P, _P]),
    ("mmp_shard_place_fast_finish_dev", C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                                  C.POINTER(C.c_void_p)]),
    ("mmp_shard_place_fast_scatter_dev", C.c_int, [_P, C.c_int32, _P, _P]),
    ("mmp_shard_unique_id", C.c_int, [_P]),
    ("mmp_shard_group_set_exchange", C.c_int, [_P, _P, _P]),
    ("mmp_shard_group_init", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("mmp_shard_group_destroy", C.c_int, [_P]),
    ("mmp_shard_commit", C.c_int, [_P]),
    ("mmp_shard_place_batch", C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int64, _P, C.POINTER(C.c_int32)]),
    ("mmp_shard_place_batch_dev", C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, _P, C.POINTER(C.c_int32)]),
    ("mmp_shard_place_batch_async_dev", C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, _P]),
    ("mmp_shard_wait", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("mmp_sync", C.c_int, [_P]),
    ("mmp_profile", C.c_int, [_P, C.c_int]),
    ("mmp_last_kernel_ms", C.c_double, [_P]),
]

_lib = None


def load() -> C.CDLL:
    """dlopen libmmplace.so and type every exported entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64; if
    # /opt/rocm's copy (libmmplace's DT_NEEDED) initialises first, torch later reports "No HIP GPUs are
    # available".  Python hosts of this library use torch for device buffers and collectives, so let
    # torch's runtime load first; libmmplace then binds to the same, already loaded, runtime.
    if os.environ.get("MMP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError here == a header symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    """void* of a numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
