"""(De)serialisation of a Fleet into the flat key space of an .npz (tests/golden/)."""
import numpy as np

from modelmesh_amd.solver import Fleet

_ARRAYS = ("pods", "models", "ent_pod", "ent_time", "allowed", "prefer", "has_allowed", "has_prefer", "replaced_rs")
_SCALARS = ("min_space_units", "min_churn_age_ms", "now", "n_types")


def pack_fleet(out: dict, prefix: str, f: Fleet) -> None:
    for k in _ARRAYS:
        v = getattr(f, k)
        if v is not None:
            out[prefix + k] = np.ascontiguousarray(v)
    for k in _SCALARS:
        out[prefix + k] = np.int64(getattr(f, k))


def unpack_fleet(z, prefix: str) -> Fleet:
    kw = {k: (z[prefix + k] if prefix + k in z.files else None) for k in _ARRAYS}
    if kw["replaced_rs"] is None:
        kw["replaced_rs"] = np.zeros(0, np.int32)
    for k in _SCALARS:
        kw[k] = int(z[prefix + k])
    return Fleet(**kw)
