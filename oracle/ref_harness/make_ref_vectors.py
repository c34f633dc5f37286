#!/usr/bin/env python3
"""Generates tests/golden/ref_getnext.npz: what the REFERENCE'S OWN TEXT decides (oracle/_ref/ref_harness, built by
build.sh from /root/reference) on the fleets of tests/ref_fleets.py.  Run in the build container (the GPU box has no
reference tree and only reads the committed file).

Per load-target case: the clusterState iteration order, and per request (chosen, candidates.size(), survivors of the rpm
filter, audit hash of the shortlist).  Per serve-target case: (chosen, chosenTimeStamp).  Per guard case: (MMP_GATE_* bits,
loadLocal's initial size).  Plus a digest of each case's inputs.
usage: python oracle/ref_harness/make_ref_vectors.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import ref_fleets as rf  # noqa: E402

# MMP_REF_HARNESS / MMP_REF_OUT: run another build of the harness (tools/ref_coverage.sh: the --coverage build) without
# touching the committed vectors
HARNESS = os.environ.get("MMP_REF_HARNESS") or os.path.join(ROOT, "oracle", "_ref", "ref_harness")
OUT = os.environ.get("MMP_REF_OUT") or os.path.join(ROOT, "tests", "golden", "ref_getnext.npz")


def proactive_inputs(fleet, units, partitioned):
    """What the harness is GIVEN for a reaper run (rows a5 / a18 produce them in a mesh): the cluster's stats and, with type
    constraints, per ProhibitedTypeSet partition its stats, its prohibited type rows and its instances."""
    from oracle import bind as ob
    g = np.zeros(1, dtype=ob.ORC_STATS)
    g[0] = ob.OracleFleet(fleet).stats()
    if not partitioned:
        return units, g, np.zeros(0, dtype=ob.ORC_STATS), [], None
    pts, sets, pst = ob.partition_stats(fleet)
    return units, g, pst, [sorted(s) for s in sets], pts


def run(blob: bytes, n_place: int, n_serve: int, n_gate: int = 0, n_scale: int = -1, n_pods: int = 0, n_sd: int = -1, proactive: bool = False,
        events: bool = False, upgrade: int = -1, types=None, migration: int = -1, conc: bool = False, env=None):
    """env: one of the harness' CONTEXT switches (harness.cc: MMP_REF_EXC_CONTEXT, MMP_REF_SEND_DEST, MMP_REF_MIGRATION_FAULTS) for
    a second pass over a case: state around the decision that the reference's text consults for side effects only."""
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        open(fin, "wb").write(blob)
        subprocess.run([HARNESS, fin, fout], check=True, env=dict(os.environ, **env) if env else None)
        raw = open(fout, "rb").read()
    n_present = int(np.frombuffer(raw, "<i8", 1)[0])
    off = 8
    order = np.frombuffer(raw, "<i4", n_present, off).copy()
    off += 4 * n_present
    place = np.frombuffer(raw, "<i4", 4 * n_place, off).reshape(n_place, 4).copy()
    off += 16 * n_place
    serve = np.frombuffer(raw, "<i8", 2 * n_serve, off).reshape(n_serve, 2).copy()
    off += 16 * n_serve
    gate = np.frombuffer(raw, "<i4", 2 * n_gate, off).reshape(n_gate, 2).copy()
    off += 8 * n_gate
    if migration >= 0:
        bits = np.frombuffer(raw, "u1", migration, off).copy()
        assert off + migration == len(raw)
        return bits
    if types is not None:  # (T, P): see harness.cc "a18"
        T, P = types
        W = (P + 63) // 64
        w = np.frombuffer(raw, "<i8", (len(raw) - off) // 8, off)
        i = 0
        rows = np.zeros((T + 1, 2 + 2 * W), np.int64)
        for t in range(T + 1):
            rows[t] = w[i: i + 2 + 2 * W]
            i += 2 + 2 * W
        pod_part = w[i: i + P].copy()
        i += P
        n_part = int(w[i])
        i += 1
        part_types, part_stats = [], []
        for _ in range(n_part):
            k = int(w[i])
            part_types.append(w[i + 1: i + 1 + k].copy())
            part_stats.append(w[i + 1 + k: i + 6 + k].copy())
            i += 6 + k
        order = w[i: i + n_part].copy()
        i += n_part
        tstats = w[i: i + 6 * T].reshape(T, 6).copy()
        assert i + 6 * T == len(w)
        return rows, pod_part, part_types, np.array(part_stats).reshape(n_part, 5), order, tstats
    if upgrade >= 0:  # after every call: n, then n x (replica set, expiry)
        w = np.frombuffer(raw, "<i8", (len(raw) - off) // 8, off)
        maps, i = [], 0
        for _ in range(upgrade):
            n = int(w[i])
            maps.append(w[i + 1: i + 1 + 2 * n].reshape(n, 2).copy())
            i += 1 + 2 * n
        assert i == len(w)
        return maps
    if events:  # per checkpoint: 5 stats words, n, n instance indices (clusterState order); then 4 counters
        w = np.frombuffer(raw, "<i8", (len(raw) - off) // 8, off)
        n_ck, i = int(w[0]), 1
        stats, orders = [], []
        for _ in range(n_ck):
            stats.append(w[i: i + 5].copy())
            n = int(w[i + 5])
            orders.append(w[i + 6: i + 6 + n].astype(np.int32))
            i += 6 + n
        assert i + 4 == len(w)
        return np.array(stats).reshape(n_ck, 5), orders, w[i: i + 4].copy()
    if proactive:
        n_calls = int(np.frombuffer(raw, "<i8", 1, off)[0])
        calls = np.frombuffer(raw, "<i8", 3 * n_calls, off + 8).reshape(n_calls, 3).copy()
        assert off + 8 + 24 * n_calls == len(raw)
        return calls
    if n_sd >= 0:
        removed = np.frombuffer(raw, "u1", n_sd, off).copy()
        assert off + n_sd == len(raw)
        return removed
    if n_scale < 0:
        assert off == len(raw)
        return order, place, serve, gate
    scale = np.frombuffer(raw, "<i8", 6 * n_scale, off).reshape(n_scale, 6).copy()
    off += 48 * n_scale
    called = int(np.frombuffer(raw, "<i8", 1, off)[0])
    off += 8
    ov = np.frombuffer(raw, "u1", n_pods, off).copy()
    off += n_pods
    if conc:  # per entry (threshold, resets, priorSum, priorCount), then the bits of averageModelParallelism
        co = np.frombuffer(raw, "<i8", 4 * n_scale + 1, off).copy()
        assert off + 8 * (4 * n_scale + 1) == len(raw)
        return order, place, serve, gate, scale, called, ov, co[:-1].reshape(n_scale, 4), co[-1:].view(np.float64)
    assert off == len(raw)
    return order, place, serve, gate, scale, called, ov


def main():
    if not os.environ.get("MMP_REF_HARNESS"):
        subprocess.run(["bash", os.path.join(ROOT, "oracle", "ref_harness", "build.sh")], check=True)
    out = {}
    names = []
    for name, fleet, ids, reqs, extra in rf.place_cases():
        blob = rf.input_blob(fleet, ids, reqs, extra)
        order, place, _, _ = run(blob, len(reqs), 0)
        if name in ("fuzz_None_3", "fuzz_prefer_5"):  # the same decisions with the destination id sent along (:4997-4999)
            o2, p2, _, _ = run(blob, len(reqs), 0, env={"MMP_REF_SEND_DEST": "1"})
            assert np.array_equal(o2, order) and np.array_equal(p2, place), name
        out[f"{name}/order"], out[f"{name}/place"] = order, place
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(order)} instances in clusterState, {len(reqs)} decisions: "
              f"{int((place[:, 0] >= 0).sum())} remote, {int((place[:, 0] == -2).sum())} self, {int((place[:, 0] == -1).sum())} none; "
              f"mean shortlist {place[:, 1].mean():.1f}")
    for name, fleet, ids, reqs, extra in rf.caller_place_cases():
        blob = rf.input_blob(fleet, ids, reqs, extra)
        order, place, _, _ = run(blob, len(reqs), 0)
        out[f"{name}/order"], out[f"{name}/place"] = order, place
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: one caller (instance {int(reqs['self_pod'][0])}), {len(reqs)} decisions: "
              f"{int((place[:, 0] >= 0).sum())} remote, {int((place[:, 0] == -2).sum())} self, {int((place[:, 0] == -1).sum())} none; "
              f"mean shortlist {place[:, 1].mean():.1f}")
    for name, fleet, ids, reqs, in_use, last_used, xp, xt in rf.serve_cases():
        blob = rf.input_blob(fleet, ids, serve=(reqs, in_use, last_used, xp, xt))
        _, _, serve, _ = run(blob, 0, len(reqs))
        if name.endswith("_0"):  # ... and the serve target's (:4386-4388)
            assert np.array_equal(run(blob, 0, len(reqs), env={"MMP_REF_SEND_DEST": "1"})[2], serve), name
        out[f"{name}/serve"] = serve
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(reqs)} serve decisions: {int((serve[:, 0] >= 0).sum())} remote, {int((serve[:, 0] == -2).sum())} self")
    from oracle import bind as ob
    for name, fleet, ids, reqs, xp, xt, expl, expiry in rf.gate_cases():
        tstats = np.ascontiguousarray(ob.type_set_stats(fleet))  # an INPUT of the guards (rows a5 / a18 are pinned separately)
        blob = rf.input_blob(fleet, ids, gates=(reqs, xp, xt, expl, expiry, tstats))
        _, _, _, gate = run(blob, 0, 0, len(reqs))
        # which exception flies when a guard refuses (one seen earlier in the request, or a new one: :4019-4031, :4598-4601,
        # :4618-4621) is not part of the decision: the same bits with failures "seen before" and for internal requests
        assert np.array_equal(run(blob, 0, 0, len(reqs), env={"MMP_REF_EXC_CONTEXT": "1"})[3], gate), name
        out[f"{name}/gate"] = gate
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        bits = gate[:, 0].astype(np.uint32)
        print(f"{name}: {len(reqs)} guard evaluations; fired: " + " ".join(f"{b}:{int(((bits >> k) & 1).sum())}" for k, b in enumerate(
            ["goLocal", "failures", "locations", "notAllowed", "churn", "earlyReject", "reload", "publish"])))
    for name, fleet, ids, entries, sp in rf.scaleup_cases():
        cstats = np.zeros(1, dtype=ob.ORC_STATS)
        cstats[0] = ob.OracleFleet(fleet).stats()  # clusterStats / typeSetStats: INPUTS of the planner (rows a5 / a18)
        blob = rf.input_blob(fleet, ids, scaleup=(entries, sp, cstats, np.ascontiguousarray(ob.type_set_stats(fleet))))
        *_, scale, called, ov = run(blob, 0, 0, 0, len(entries), fleet.n_pods)
        out[f"{name}/scale"], out[f"{name}/overloaded"] = scale, ov
        out[f"{name}/exclude_set_built"] = np.array([called])
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(entries)} cache entries: {int((scale[:, 0] == 1).sum())} second copies, {int((scale[:, 0] == 2).sum())} scale-ups, "
              f"{int(scale[:, 5].sum())} heavy, {int(ov.sum())} overloaded instances")
    for name, fleet, ids, entries, sp in rf.scaleup_edge_cases():
        cstats = np.zeros(1, dtype=ob.ORC_STATS)
        cstats[0] = ob.OracleFleet(fleet).stats()
        blob = rf.input_blob(fleet, ids, scaleup=(entries, sp, cstats, np.ascontiguousarray(ob.type_set_stats(fleet))))
        *_, scale, called, ov = run(blob, 0, 0, 0, len(entries), fleet.n_pods)
        out[f"{name}/scale"], out[f"{name}/overloaded"] = scale, ov
        out[f"{name}/exclude_set_built"] = np.array([called])
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(entries)} cache entries: {int((scale[:, 0] == 1).sum())} second copies, {int((scale[:, 0] == 2).sum())} scale-ups, "
              f"{int(ov.sum())} overloaded instances, exclude set built: {called}")
    for name, fleet, ids, entries, conc, sp, cp in rf.scaleup_conc_cases():
        cstats = np.zeros(1, dtype=ob.ORC_STATS)
        cstats[0] = ob.OracleFleet(fleet).stats()
        blob = rf.input_blob(fleet, ids, scaleup=(entries, sp, cstats, np.ascontiguousarray(ob.type_set_stats(fleet))), conc=(conc, cp))
        *_, scale, called, ov, cout, avg = run(blob, 0, 0, 0, len(entries), fleet.n_pods, conc=True)
        out[f"{name}/scale"], out[f"{name}/overloaded"] = scale, ov
        out[f"{name}/exclude_set_built"] = np.array([called])
        out[f"{name}/conc"], out[f"{name}/average_model_parallelism"] = cout, avg
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(entries)} MaxConcCacheEntry entries: {int((scale[:, 0] == 1).sum())} second copies, {int((scale[:, 0] == 2).sum())} scale-ups, "
              f"{int(scale[:, 5].sum())} heavy, {int(cout[:, 1].sum())} counter resets, {len(np.unique(cout[:, 0]))} distinct thresholds, "
              f"{int(ov.sum())} overloaded instances; averageModelParallelism {float(cp['average_model_parallelism'][0])} -> {float(avg[0])!r}")
    for name, fleet, ids, entries, conc, dp, dyn in rf.scaledown_conc_cases():
        istats = ob.instance_set_stats(fleet, int(dp["self_pod"][0]))
        cp = np.zeros(1, dtype=rf._lib.CONC_PARAMS)
        cp["dynamic_rpm_scale_constant"], cp["average_model_parallelism"] = dyn, 1.0
        blob = rf.input_blob(fleet, ids, scaledown=(entries, dp, istats), conc=(conc, cp))
        removed = run(blob, 0, 0, 0, -1, 0, len(entries))
        out[f"{name}/removed"] = removed
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(entries)} MaxConcCacheEntry candidates: {int(removed.sum())} local copies removed")
    for name, fleet, ids, entries, dp in rf.scaledown_edge_cases():
        istats = ob.instance_set_stats(fleet, int(dp["self_pod"][0]))
        blob = rf.input_blob(fleet, ids, scaledown=(entries, dp, istats))
        removed = run(blob, 0, 0, 0, -1, 0, len(entries))
        out[f"{name}/removed"] = removed
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(entries)} candidates: {int(removed.sum())} local copies removed")
    for name, fleet, ids, entries, dp in rf.scaledown_cases():
        istats = ob.instance_set_stats(fleet, int(dp["self_pod"][0]))  # instanceSetStats(): an INPUT (rows a5 / a18)
        blob = rf.input_blob(fleet, ids, scaledown=(entries, dp, istats))
        removed = run(blob, 0, 0, 0, -1, 0, len(entries))
        out[f"{name}/removed"] = removed
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(entries)} candidates: {int(removed.sum())} local copies removed")
    for name, fleet, ids, units, partitioned in rf.proactive_cases():
        blob = rf.input_blob(fleet, ids, proactive=proactive_inputs(fleet, units, partitioned))
        calls = run(blob, 0, 0, proactive=True)
        out[f"{name}/proactive"] = calls
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {fleet.n_models} registry rows: {len(calls)} proactive loads over {len(np.unique(calls[:, 2])) if len(calls) else 0} instance subset(s)")
    for name, fleet, ids, ev, ck, tables in rf.table_event_cases():
        blob = rf.input_blob(fleet, ids, events=(ev, ck))
        stats, orders, counters = run(blob, 0, 0, events=True)
        assert len(orders) == len(tables)
        out[f"{name}/stats"] = stats
        out[f"{name}/order_len"] = np.array([len(o) for o in orders], np.int32)
        out[f"{name}/orders"] = np.concatenate(orders) if orders else np.zeros(0, np.int32)
        out[f"{name}/counters"] = counters
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(ev)} listener events, {len(orders)} checkpoints; upgradeTracker added/removed {counters[0]}/{counters[1]}, "
              f"housekeepings {counters[2]}")
    for name, fleet, ids, pod_bits, req_bits, pref_bits in rf.type_constraint_cases():
        blob = rf.input_blob(fleet, ids, types=(pod_bits, req_bits, pref_bits))
        rows, pod_part, part_types, part_stats, order, tstats = run(blob, 0, 0, types=(len(req_bits), fleet.n_pods))
        out[f"{name}/rows"], out[f"{name}/pod_part"], out[f"{name}/part_stats"] = rows, pod_part, part_stats
        out[f"{name}/part_types_len"] = np.array([len(x) for x in part_types], np.int32)
        out[f"{name}/part_types"] = np.concatenate(part_types) if part_types else np.zeros(0, np.int64)
        out[f"{name}/order"], out[f"{name}/tstats"] = order, tstats
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(req_bits)} types over {fleet.n_pods} instances: {len(part_types)} ProhibitedTypeSet partitions, "
              f"{int(rows[:-1, 0].sum())} types with requirements, {int(rows[:, 1].sum())} rows with preferred instances")
    for name, fleet, ids, entries, self_pod, now in rf.migration_cases():
        blob = rf.input_blob(fleet, ids, migration=(entries, self_pod, now))
        bits = run(blob, 0, 0, migration=len(entries))
        # aborted loads (deregistered at once, :7027-7030) and copies that fail to load elsewhere (:7033-7036): the same copies
        # are triggered; the shutdown does not wait for a copy that failed
        b2 = run(blob, 0, 0, migration=len(entries), env={"MMP_REF_MIGRATION_FAULTS": "1"})
        assert np.array_equal(b2 & 1, bits & 1) and np.all((b2 >> 1) <= (bits >> 1)) and (b2 >> 1).sum() < (bits >> 1).sum(), name
        out[f"{name}/migration"] = bits
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(entries)} cache entries: {int((bits & 1).sum())} copies triggered elsewhere, the shutdown waits for {int((bits >> 1).sum())}")
    small = rf.wl.fuzz_fleet(1, pods=4, models=4)  # the harness wants a fleet in every input; the tracker never looks at it
    small_ids = rf.string_ids(small, 1)
    for name, ev in rf.upgrade_event_cases():
        blob = rf.input_blob(small, small_ids, upgrade=ev)
        maps = run(blob, 0, 0, upgrade=len(ev))
        out[f"{name}/map_len"] = np.array([len(m) for m in maps], np.int32)
        out[f"{name}/maps"] = np.concatenate(maps) if maps else np.zeros((0, 2), np.int64)
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(ev)} tracker calls; the likely-replaced map is non-empty after {int((out[f'{name}/map_len'] > 0).sum())} of them, "
              f"up to {int(out[f'{name}/map_len'].max())} replica sets")
    out["names"] = np.array(names)
    out["manifest"] = np.array(open(os.path.join(ROOT, "oracle", "_ref", "gen", "MANIFEST.txt")).read())
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB, {len(names)} cases")


if __name__ == "__main__":
    main()
