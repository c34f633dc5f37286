"""py_types.py — CPU restatement of TypeConstraintManager's per-type instance sets
(TypeConstraintManager.java:416-447 fromInstanceSet, :478-486 instanceMatches, :680-725
refreshPerTypeInstanceSets, :727-747 inferPreferredInstances).  TEST INFRASTRUCTURE ONLY.
Pinning: held to the reference's own text since round 3 (oracle/ref_harness -> tests/golden/ref_getnext.npz,
tests/test_ref_vectors.py: 8 configurations); the reference's own test C.4 (ModelMeshErrorPropagationTest.java:52-95:
a type that requires a label only one instance has is placed on exactly that instance)."""
from __future__ import annotations

INT_MAX = (1 << 31) - 1


def instance_matches(instance_labels: set, type_labels: list, match_all: bool) -> bool:
    if len(instance_labels) == 0 or len(type_labels) == 0:  # :480-482
        return False
    has = [l in instance_labels for l in type_labels]
    return all(has) if match_all else any(has)


def infer_preferred(scores: dict, include):  # :727-747
    ids = set()
    mn, mx = INT_MAX, 0
    for iid, score in scores.items():
        if include is not None and iid not in include:
            continue
        if score < mn:
            mn = score
        if score >= mx:
            if score > mx:
                ids.clear()
                mx = score
            ids.add(iid)
    return frozenset(ids) if mn < mx else None


def type_tables(instances: dict, type_config: dict):
    """instances: iid -> set(labels) for every record in clusterState.
    type_config: type -> (requiredLabels list, preferredLabels list).
    Returns (per_type: type -> (allowedInstances|None, preferredInstances|None), defaultPreferred|None)."""
    mtc = {}
    for t, (req, pref) in type_config.items():  # fromInstanceSet :416-447
        required = None if len(req) == 0 else set()
        preferred = None
        for iid, labels in instances.items():
            if required is not None and instance_matches(labels, req, True):
                required.add(iid)
            elif instance_matches(labels, pref, False):
                if preferred is None:
                    preferred = set()
                preferred.add(iid)
        mtc[t] = dict(allowed=required, configured=preferred, preferred=preferred)
    # refreshPerTypeInstanceSets :680-725
    scores = {}
    for iid, labels in instances.items():
        prohibited = sum(1 for t, (req, _) in type_config.items()
                         if len(req) != 0 and not instance_matches(labels, req, True))
        scores[iid] = prohibited * 4
    for t, m in mtc.items():
        if m["configured"] is not None:
            for p in m["configured"]:
                if p in scores:
                    scores[p] -= 1
    default_preferred = infer_preferred(scores, None)
    out = {}
    for t, m in mtc.items():
        if m["allowed"] is None:
            pref = default_preferred  # updateInstanceSetStats(null, defaultPreferred), :714-715
        elif m["configured"] is not None or len(m["allowed"]) == 0:
            pref = m["preferred"]
        else:
            pref = infer_preferred(scores, m["allowed"])
        out[t] = (m["allowed"], pref)
    return out, default_preferred
