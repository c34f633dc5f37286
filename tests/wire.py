"""Test helper: the synthetic KV wire-format generators live in modelmesh_amd/wire.py (bench.py uses them
too); here, additionally, what a JSON library says the rows are (the oracle of the device parser)."""
import json

import numpy as np

from modelmesh_amd.wire import adopt_ids, make_ids, model_values, pod_values  # noqa: F401


def parsed_pod_rows(values, proto_rows):
    """What a JSON library makes of the values (the parser's oracle)."""
    rows = proto_rows.copy()
    st = np.zeros(len(values), np.int64)
    for i, v in enumerate(values):
        d = json.loads(v)
        rows["lru_time"][i] = d.get("lruTime", 0)
        rows["count"][i] = d.get("count", 0)
        rows["capacity"][i] = d.get("cap", 0)
        rows["used"][i] = d.get("used", 0)
        rows["loading_threads"][i] = d.get("lThreads", 0)
        rows["loading_in_progress"][i] = d.get("lInProg", 0)
        rows["rpm"][i] = d.get("rpm", 0)
        rows["version"][i] = d.get("vers", 0)
        rows["flags"][i] = (rows["flags"][i] & ~np.uint32(1)) | (1 if d.get("shutdown", False) else 0)
        st[i] = d.get("startTime", 0)
    return rows, st

