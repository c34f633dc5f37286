"""Second, independent restatement of the batch rebalancers (test infrastructure, never imported by the product):
rateTrackingTask's scale-up decisions (MM.java:5636-5856), the janitor's scale-down (MM.java:6110-6145 with
removeModelCopies :6197-6310 and removeSecondModelCopy :6314-6335), preShutdown's migration order (:6985-7040) and the
reaper's proactive loads (:6455-6488, :6574-6577, :6616-6747).
Written from the Java text in plain Python ints with Java's wrap / truncation semantics, and compared with the C
restatement (oracle/mm_rebalance_oracle.c) on random fleets by tests/test_oracle_cross.py.

Inputs are plain Python structures:
  pods[i]    dict(rpm, shutting_down, in_table)            (in_table: instanceInfo.get(iid) != null)
  models[m]  dict(type, last_used, loaded=[(pod, loadStart)...] in instance-id order, failed=[(pod, time)...])
  entries[e] dict(model (-1: no ModelRecord), weight, last_used, interval_count, last_heavy_time, last_unload_time,
                  earlier_use_iteration, last_used_iteration, failed)
  stats      dict(total_capacity, total_free, global_lru, instance_count, model_copy_count)
"""
from oracle.py_oracle import _i, _jdiv, _l

SECOND_COPY_REMOVE_MAX_AGE_MS = 10 * 3_600_000  # MM.java:257


def _loaded_since(mr, cutoff, ignore):  # MM.java:5860-5871
    for pod, start in mr["loaded"]:
        if ignore is not None and pod == ignore:
            continue
        if start is not None and start > cutoff:
            return True
    return False


CONC_COUNT_BITS = 21  # MM.java:2653
INT_MAX = 2**31 - 1


def rpm_scale_threshold(m, and_reset, scale_up_rpm_threshold, dyn_const):
    """MaxConcCacheEntry.getRpmScaleThreshold(andReset), MM.java:2766-2796, on
    m = dict(count_and_time_sum, prior_sum, prior_count, max_conc) -> (threshold, reset, new_prior_sum, new_prior_count)."""
    cur_val = m["count_and_time_sum"] & (2**64 - 1)
    count = _i(cur_val & ((1 << CONC_COUNT_BITS) - 1))
    prior_sum, prior_count, reset = m["prior_sum"], m["prior_count"], 0
    if count >= 64:
        time_sum = _l(cur_val >> CONC_COUNT_BITS)              # `>>>`
        if and_reset:                                          # sumThenReset(), :2771-2773
            prior_sum, prior_count, reset = time_sum, count, 1
    else:
        pc = m["prior_count"]
        if pc <= 0 and count < 8:
            return scale_up_rpm_threshold, reset, prior_sum, prior_count
        time_sum = _l(m["prior_sum"] + ((cur_val >> CONC_COUNT_BITS) if count > 0 else 0))
        count = _i(count + pc)
    if time_sum == 0:
        return INT_MAX, reset, prior_sum, prior_count
    num = _l(m["max_conc"] * _l(count * dyn_const))
    return _i(_jdiv(num, time_sum)), reset, prior_sum, prior_count


def _j2i_double(d):
    """(int) of a double, JLS 5.1.3"""
    if d != d:
        return 0
    if d >= 2147483647.0:
        return INT_MAX
    if d <= -2147483648.0:
        return -2**31
    return int(d)


def scaleup(pods, order, stats, type_stats, has_type_constraints, models, entries, p, conc=None, conc_params=None):
    """-> (list of dict(action, copies, timestamp, new_i1, new_i2, heavy, rpm), overloaded set, returned_early).
    action 0 none, 1 second copy (ensureLoadedInternalAsync(id, lastTime, weight, excludeThisInstance, 0)),
    2 scale up by `copies` (timestamp = now + 20 s).
    conc / conc_params (limitModelConcurrency == true, latencyBased :5677): per entry dict(count_and_time_sum, prior_sum,
    prior_count, max_conc), dict(dynamic_rpm_scale_constant, average_model_parallelism); the outs then carry threshold / reset /
    new_prior_sum / new_prior_count and a fourth value is returned: dict(average_model_parallelism, model_parallelism_sum)."""
    latency_based = conc is not None
    outs = [dict(action=0, copies=0, timestamp=0, new_i1=e["earlier_use_iteration"], new_i2=e["last_used_iteration"],
                 heavy=0, rpm=0) for e in entries]
    if latency_based:
        for o, m in zip(outs, conc):
            o.update(threshold=0, reset=0, new_prior_sum=m["prior_sum"], new_prior_count=m["prior_count"])
        res = dict(average_model_parallelism=conc_params["average_model_parallelism"], model_parallelism_sum=0)
        early = scaleup_body(pods, order, stats, type_stats, has_type_constraints, models, entries, p, conc, conc_params, outs, res)
        return outs, early[0], early[1], res
    ov, early = scaleup_body(pods, order, stats, type_stats, has_type_constraints, models, entries, p, None, None, outs, None)
    return outs, ov, early


def scaleup_body(pods, order, stats, type_stats, has_type_constraints, models, entries, p, conc, conc_params, outs, res):
    latency_based = conc is not None
    now, last_time = p["now"], p["last_check_time"]
    time_delta = _l(now - last_time)
    if _l(time_delta * 5) < _l(p["rate_check_interval_ms"] * 3):
        return set(), True
    lower = _i(p["iteration_counter"] - p["second_copy_max_age_iters"])
    upper = _i(p["iteration_counter"] - p["second_copy_min_age_iters"])
    inst_count = stats["instance_count"]
    if inst_count < 2:
        return set(), True
    if not entries:                                    # usedSinceLastRun.isEmpty()
        return set(), True
    new_copies_timestamp = _l(now + 20_000)
    scale_up_rpms = heavy_rpms = 0
    if not latency_based:                              # :5679-5682
        scale_up_rpms = p["scale_up_rpm_threshold"]
        heavy_rpms = _jdiv(_i(scale_up_rpms * 3), 4)
    exclude_set = None
    model_parallelism_sum = 0
    for e, ce in enumerate(entries):
        o = outs[e]
        try:
            count = ce["interval_count"]
            mr = models[ce["model"]] if ce["model"] >= 0 else None
            cluster_stats = stats
            if has_type_constraints and mr is not None:       # typeSetStats(ce.modelInfo.serviceType)
                t = mr["type"]
                cluster_stats = type_stats[t if 0 <= t < len(type_stats) else 0]
            suitable = inst_count
            if has_type_constraints:
                suitable = cluster_stats["instance_count"]
                if suitable < 2:
                    continue
            if latency_based:                           # :5702-5707
                scale_up_rpms, o["reset"], o["new_prior_sum"], o["new_prior_count"] = rpm_scale_threshold(
                    conc[e], True, p["scale_up_rpm_threshold"], conc_params["dynamic_rpm_scale_constant"])
                o["threshold"] = scale_up_rpms
                heavy_rpms = _jdiv(_i(scale_up_rpms * 3), 4)
                model_parallelism_sum = _i(model_parallelism_sum + conc[e]["max_conc"])
            rpm = _i(_jdiv(_l(count * 60_000), time_delta))
            o["rpm"] = rpm
            if rpm > heavy_rpms:
                o["heavy"] = 1
            if mr is None:
                continue
            loaded_count = len(mr["loaded"])
            if loaded_count == 0:
                continue
            failed_count = len(mr["failed"])
            candidates = suitable - (loaded_count + failed_count)
            if candidates <= 0:
                continue
            if loaded_count == 1:
                i1, i2 = ce["earlier_use_iteration"], ce["last_used_iteration"]
                i1_in = i2_in = False
                if i2 >= lower and i1 <= upper:
                    i1_in = i1 >= lower
                    i2_in = i2 <= upper
                if i2_in or not i1_in:
                    o["new_i1"] = i2
                o["new_i2"] = p["iteration_counter"]
                if i1_in or i2_in:
                    if cluster_stats["total_capacity"] == 0:
                        raise ZeroDivisionError        # caught like the RuntimeException at :5810
                    if (_jdiv(_l(10 * cluster_stats["total_free"]), cluster_stats["total_capacity"]) >= 1
                            or _l(now - cluster_stats["global_lru"]) > p["second_copy_lru_threshold_ms"]):
                        o["action"], o["copies"], o["timestamp"] = 1, 1, last_time
                        continue
            if rpm < scale_up_rpms:
                continue
            recent_cutoff = _l(now - _l(time_delta + p["rate_check_interval_ms"] + 2 * p["assume_completed_ms"]))
            if _loaded_since(mr, recent_cutoff, p["self_pod"]):
                continue
            if exclude_set is None:                     # getExcludeSet(), :5835-5856
                # getExcludeSet's OWN scaleUpRpms, :5836
                x_rpms = _j2i_double(900.0 * conc_params["average_model_parallelism"]) if latency_based else p["scale_up_rpm_threshold"]
                max_rpm = max(_i(x_rpms * 4), _i(p["our_rpm"] - _i(2 * x_rpms)))
                exclude_set = set()
                for iid in order:
                    if iid == p["self_pod"]:
                        continue
                    if pods[iid]["rpm"] > max_rpm:
                        exclude_set.add(iid)
            excluded_count = len(exclude_set)
            if excluded_count != 0:
                held = {pod for pod, _ in mr["loaded"]} | {pod for pod, _ in mr["failed"]}
                for iid in exclude_set:
                    if iid not in held:
                        candidates -= 1
                candidates -= excluded_count
                if candidates <= 0:
                    continue
            if scale_up_rpms == 0:
                raise ZeroDivisionError
            copies = min(_jdiv(rpm, scale_up_rpms), candidates)
            if copies > 2:
                copies = min(copies, _jdiv(suitable, 3))
            o["action"], o["copies"], o["timestamp"] = 2, copies, new_copies_timestamp
        except ZeroDivisionError:
            continue
    if latency_based:                                   # :5815-5818
        res["average_model_parallelism"] = max(1.0, float(model_parallelism_sum) / len(entries))
        res["model_parallelism_sum"] = model_parallelism_sum
    return (exclude_set or set()), False


def scaledown(pods, pos_of, stats, models, entries, p, conc=None, dyn_const=0):
    """-> list of bool (removeModelCopies returned true).  entries = scaleCopiesCandidates, oldest first.
    conc: the entries are MaxConcCacheEntry objects (per entry dict(count_and_time_sum, prior_sum, prior_count, max_conc,
    queued_requests)), MM.java:6294-6305."""
    removed_out = [False] * len(entries)
    if p["shutting_down"]:
        return removed_out
    now = p["now"]
    max_weight = _jdiv(p["adjusted_cache_capacity"], 20)
    removed_count = 0
    for e, ce in enumerate(entries):
        weight = ce["weight"]
        can_remove = removed_count == 0 or weight <= max_weight
        removed = _remove_model_copies(pods, pos_of, stats, models, ce, ce["last_used"], now, can_remove, p,
                                       conc[e] if conc is not None else None, dyn_const)
        if removed:
            removed_out[e] = True
            removed_count += 1
            max_weight -= weight
    return removed_out


def _remove_model_copies(pods, pos_of, stats, models, ce, last_used, now, can_remove, p, mcce=None, dyn_const=0):
    if last_used == 0:
        return False
    if ce["model"] < 0:
        return False                                    # never became a candidate: no ModelRecord
    mr = models[ce["model"]]
    num = len(mr["loaded"])
    if not can_remove or num < 2:
        return False
    if stats["total_capacity"] == 0 or _jdiv(_l(stats["total_free"] * 100), stats["total_capacity"]) > 5:
        return False
    other = None
    for iid, _ in mr["loaded"]:
        if iid != p["self_pod"]:
            ir = pods[iid]
            if ir["in_table"] and not ir["shutting_down"]:
                other = iid
                break
    if other is None:
        return False
    if num == 2:
        last_heavy = ce["last_heavy_time"]
        cache_age = _l(now - stats["global_lru"])
        scale_down_age = _jdiv(cache_age, 10)
        if last_heavy == 0 or _l(now - last_heavy) < _jdiv(cache_age, 5):
            scale_down_age = min(SECOND_COPY_REMOVE_MAX_AGE_MS, scale_down_age)
        if _l(now - last_used) > scale_down_age:
            # removeSecondModelCopy
            sp = p["self_pod"]
            if sp < 0 or not pods[sp]["in_table"] or pods[sp]["shutting_down"]:
                return False
            if pos_of[other] > pos_of[sp]:              # PLACEMENT_ORDER.compare(other, this) > 0
                return False
            return True
        return False
    last_unload = ce["last_unload_time"]
    if last_unload > 0 and _l(now - last_unload) < 8 * p["rate_check_interval_ms"]:
        return False
    if _loaded_since(mr, _l(now - 1_800_000), None):
        return False
    min_age = _jdiv(_l(_l(3 * stats["global_lru"]) + 10_400_000), 100)
    if min_age < 600_000:
        min_age = 600_000
    elif min_age > 18_000_000:
        min_age = 18_000_000
    if _l(now - ce["last_heavy_time"]) < min_age:
        return False
    since = _l(now - p["last_check_time"])
    if since < _jdiv(p["rate_check_interval_ms"], 10):
        return False
    cnt = ce["interval_count"]
    rpm = 0 if cnt == 0 else _jdiv(_l(60_000 * cnt), since)
    threshold = p["scale_up_rpm_threshold"] if mcce is None else rpm_scale_threshold(mcce, False, p["scale_up_rpm_threshold"], dyn_const)[0]
    if rpm > _jdiv(_l(threshold * 2), 3):
        return False
    if mcce is not None and mcce["queued_requests"] > 1:   # :6303
        return False
    return True


def migration(models, entries, self_pod, now, cutoff_age_ms):
    """-> (trigger[], wait[]): triggerNewModelCopyElsewhere is called / the shutdown waits for that copy."""
    cutoff = _l(now - cutoff_age_ms)
    act, wait = [], []
    for ce in entries:
        a = w = False
        mr = models[ce["model"]] if ce["model"] >= 0 else None
        if mr is not None and any(pod == self_pod for pod, _ in mr["loaded"]):
            if not ce["failed"]:                        # ce == null || ce.isFailed() -> return null
                lru_time = ce["last_used"]
                if lru_time > 0:
                    a = True
                    w = lru_time >= cutoff              # (when the status comes back LOADING)
        act.append(a)
        wait.append(w)
    return act, wait


def proactive(pods, global_stats, stats, in_subset, prohibited_types, skip, models, default_units, now):
    """The reaper's proactive loads for one instance subset (MM.java:6455-6488 candidate collection with the rule of
    :6574-6577, triggerProactiveLoadsForInstanceSubset :6616-6747).  pods[i] = dict(capacity, used, loading_threads,
    loading_in_progress, shutting_down) (clusterState holds only present rows); in_subset[i]: the row's
    prohibitedTypes equal excludeTypes (None: no type constraints); prohibited_types: set of type rows (or None);
    skip: set of candidates already nulled by earlier subsets.
    -> (selected [(model, lastUsed)] in call order, info dict) or (None, info) when sizeEstimate == 0 throws."""
    info = dict(size_estimate=0, free_count=0, total_count=0, n_candidates=0, n_selected=0, error=0, space_to_fill=0, cutoff=0)
    free_count = total_count = 0
    if stats["total_capacity"] > 0 and stats["total_free"] > 0:
        if stats["model_copy_count"] < 3:
            size_estimate = default_units
        else:
            average = _jdiv(_i(_l(stats["total_capacity"] - stats["total_free"])), stats["model_copy_count"])
            size_estimate = average if stats["model_copy_count"] > 10 else _jdiv(_i(average + default_units), 2)
        info["size_estimate"] = size_estimate
        if size_estimate == 0:
            info["error"] = 1                            # spaceToFill / sizeEstimate throws
            return None, info
        space = 0
        for i, ir in enumerate(pods):
            if ir["shutting_down"]:
                continue
            if in_subset is not None and not in_subset[i]:
                continue
            max_loads = _i(_i(ir["loading_threads"] * 50) - ir["loading_in_progress"])
            if max_loads <= 0:
                continue
            reserve = _jdiv(ir["capacity"], 8)
            avail = _l(max(0, _l(ir["capacity"] - ir["used"])) - reserve)
            if avail > 0:
                space = _l(space + min(avail, _i(max_loads * size_estimate)))
        space = _jdiv(space, 2)
        info["space_to_fill"] = space
        free_count = _i(_jdiv(space, size_estimate))
        total_count = max(free_count, _i(_jdiv(stats["total_capacity"], _l(20 * size_estimate))))
    info["free_count"], info["total_count"] = free_count, total_count
    lru = stats["global_lru"]
    age = 0 if lru == 0 else _l(now - lru)
    cutoff = 0 if lru == (1 << 63) - 1 else _l(lru + max(_jdiv(age, 3), 1_200_000))
    info["cutoff"] = cutoff
    # candidate collection (pruneModelRegistry): only when the cluster has capacity; globalLru == 0 <=> free space
    cand_enabled = global_stats["total_capacity"] > 0
    g_lru = 0 if global_stats["total_free"] > 0 else global_stats["global_lru"]
    to_load = []                                         # TreeSet<ModelToLoad>: descending lastUsed, equal = duplicate
    for i, mr in enumerate(models):
        if not (cand_enabled and not mr["loaded"] and len(mr["failed"]) < 2 and (g_lru == 0 or mr["last_used"] > g_lru)):
            continue
        info["n_candidates"] += 1
        if skip and i in skip:
            continue
        if prohibited_types is not None and mr["type"] in prohibited_types:
            continue
        last_used = mr["last_used"]
        if total_count > 0 and (free_count > 0 or last_used > cutoff):
            if len(to_load) < total_count or to_load[-1][1] < last_used:
                if all(lu != last_used for _, lu in to_load):
                    to_load.append((i, last_used))
                    to_load.sort(key=lambda t: -t[1])
                if len(to_load) > total_count:
                    to_load.pop()
    selected = []
    fs = free_count
    for i, lu in to_load:
        if fs > 0:
            fs -= 1
        elif lu < cutoff:
            break
        selected.append((i, lu))
    info["n_selected"] = len(selected)
    return selected, info
