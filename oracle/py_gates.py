"""Second, independent restatement of the request-level guards (test infrastructure, never imported by the
product).  Written from the Java text alone — not from oracle/mm_gates_oracle.c — so that the two readings can be
compared on random inputs (tests/test_oracle_cross.py): none of these guards is named by a reference test
(SURVEY.md §8c); since round 3 the C restatement is ALSO held to the reference's own text (oracle/ref_harness,
tests/test_ref_vectors.py).

Java arithmetic is kept literally: `long` / `int` wrap, `/` truncates toward zero, Math.abs(MIN_VALUE) stays negative.
"""
from oracle.py_oracle import _i, _jdiv, _l

LONG_MAX = (1 << 63) - 1
MAX_LOAD_FAILURES = 3          # MM.java:222
MAX_LOAD_LOCATIONS = 5         # MM.java:224
PUBLISH_FREQ_MS = 40_000       # INSTANCE_REC_PUBLISH_FREQ_MS, MM.java:231
PUBLISH_MIN_PERIOD_MS = 2_000  # INSTANCE_REC_PUBLISH_MIN_PERIOD_MS, MM.java:232


def _age(t, now):  # MM.java:4162-4164: 0 means "now"
    return 0 if t == 0 else _l(now - t)


def _labs(x):  # Math.abs(long)
    x = _l(x)
    return x if x >= 0 or x == -(1 << 63) else -x


def _iabs(x):  # Math.abs(int)
    x = _i(x)
    return x if x >= 0 or x == -(1 << 31) else -x


def go_local(copies, self_pod, favour_self_for_hits, have_cache_entry, entry_done, now):
    """MM.java:3598-3626.  copies = [(instance, loadStartTime)] = filteredInstances."""
    filtered_count = len(copies)
    if filtered_count <= 0:
        return False
    local_loaded = None
    for pod, t in copies:
        if pod == self_pod:
            local_loaded = t
    go = False
    if local_loaded is not None:
        go = filtered_count == 1
        if not go and favour_self_for_hits:
            if have_cache_entry:            # getFromCache(...) != null
                if entry_done:              # cacheEntry.isDone()
                    go = True
                else:
                    oldest = min(t for _, t in copies)   # oldest(filteredInstances), :4166
                    if oldest == local_loaded or _age(oldest, now) < 1500:
                        go = True
    return go


def load_failures_breached(fail_times, now, in_use_expiry_ms):
    """checkLoadFailureCount, MM.java:4607-4627 (true = it would throw)."""
    if not fail_times:
        return False
    count = 0
    cutoff = _l(now - in_use_expiry_ms)
    for t in fail_times:
        if t > cutoff:
            count += 1
        if count >= MAX_LOAD_FAILURES:
            return True
    return False


def load_locations_breached(loaded_pods, explicit_excludes, in_table):
    """checkLoadLocationCount, MM.java:4590-4604.  in_table[p] = instanceInfo.contains(p)."""
    count = 0
    for p in loaded_pods:
        if (explicit_excludes is None or p not in explicit_excludes) and in_table[p]:
            count += 1
            if count >= MAX_LOAD_LOCATIONS:
                return True
    return False


def churn_reject(min_churn_age_ms, min_space_units, capacity, weighted_size, oldest_time, now):
    """MM.java:3870-3884."""
    if min_churn_age_ms > 0:
        remaining = _l(capacity - weighted_size)
        if remaining < min_space_units:      # isFull(remaining), :4640
            lru = oldest_time
            if lru >= 0 and lru != LONG_MAX and _age(lru, now) < min_churn_age_ms:
                return True
    return False


def load_local_initial_size(have_size_hint, size_hint, loading_count, weight_predict_cutoff, loader_predicted,
                            stats, we_created_entry, last_used_time, capacity, weighted_size, oldest_time):
    """loadLocal's size prediction and early reject, MM.java:5158-5197.  stats = dict(total_capacity, total_free,
    model_copy_count).  -> (initialSize, rejected)"""
    initial = 0
    if have_size_hint:
        initial = _i(size_hint)
    elif loading_count > weight_predict_cutoff:
        copy_count = stats["model_copy_count"]
        if copy_count >= 10:
            # -(1 + (int) (stats.totalCapacity - stats.totalFree) / copyCount): the cast binds to the difference
            diff = _i(_l(stats["total_capacity"] - stats["total_free"]))
            initial = _i(-_i(1 + _jdiv(diff, copy_count)))
    if initial == 0:
        initial = _i(loader_predicted)
    abs_size = _iabs(initial)
    rejected = False
    if we_created_entry:
        if abs_size > capacity or (last_used_time > 0 and abs_size > _l(capacity - weighted_size)
                                   and last_used_time < oldest_time):
            rejected = True
    return initial, rejected


def reload_elsewhere(entry_failed, loaded_time, load_timeout_ms, now, stats):
    """onEviction's rebalancing rule, MM.java:2886-2920.  loaded_time < 0: the instance is in neither map."""
    attempt = False
    if not entry_failed:
        in_registry = loaded_time >= 0
        attempt = in_registry and _l(now - loaded_time) > _l(2 * load_timeout_ms)
    if not attempt:
        return False
    return (stats["total_capacity"] > 0 and stats["instance_count"] > 1
            and _jdiv(_l(20 * stats["total_free"]), stats["total_capacity"]) >= 1)


def _loading_change(cur, load_in_prog):  # MM.java:5536-5543
    cur_in_prog = cur["loading_in_progress"]
    if load_in_prog == cur_in_prog:
        return False
    if (load_in_prog == 0) != (cur_in_prog == 0):
        return True
    threads = cur["loading_threads"]
    if (load_in_prog <= threads) != (cur_in_prog <= threads):
        return True
    return _iabs(load_in_prog - cur_in_prog) >= 3


def _load_change(cur_rpm, rpm):  # MM.java:5546-5550
    diff = _iabs(cur_rpm - rpm)
    if diff >= 100:
        return True
    if cur_rpm == 0:
        return rpm != 0
    return _jdiv(_i(100 * diff), cur_rpm) > 10


def should_publish(cur, fresh, now, last_published, force, pre_shutdown, min_space_units):
    """publishInstanceRecord, MM.java:5388-5468: does it write an update?  cur = the record in the table (None if
    absent), fresh = the values it would publish (dicts: lru_time, capacity, used, count, loading_threads,
    loading_in_progress, rpm, shutting_down)."""
    last_done = _l(now - last_published)
    if not pre_shutdown and (last_done < PUBLISH_MIN_PERIOD_MS or (not force and last_done < PUBLISH_FREQ_MS - 1000)):
        return False
    old = last_done > PUBLISH_FREQ_MS * 4
    shutting = bool(fresh["shutting_down"])
    if cur is None:
        return not shutting
    oldest, cap, used, count = fresh["lru_time"], fresh["capacity"], fresh["used"], fresh["count"]
    if oldest == -1:                                      # :5423-5425: runtimeCache.oldestTime() of an empty cache
        oldest = LONG_MAX
    threads, in_prog, rpms = fresh["loading_threads"], fresh["loading_in_progress"], fresh["rpm"]

    def is_full(avail):
        return avail < min_space_units

    cur_rem = max(0, _l(cur["capacity"] - cur["used"]))   # InstanceRecord.getRemaining, :203-205
    if not old:
        ok = bool(cur["shutting_down"]) == shutting
        ok = ok and _labs(cur["capacity"] - cap) < _jdiv(cap, 50)
        diff = _labs(cur["lru_time"] - oldest)
        ok = ok and diff < 20_000
        ok = ok and (cur["lru_time"] == LONG_MAX or diff < _jdiv(_l(now - cur["lru_time"]), 16))
        diff = _iabs(cur["count"] - count)
        ok = ok and diff < 10
        ok = ok and ((count == 0) if cur["count"] == 0 else (_jdiv(diff * 100, cur["count"]) < 15))
        ok = ok and ((used == 0) if cur["used"] == 0 else
                     (_jdiv(_l(_labs(cur["used"] - used) * 100), cur["used"]) < 20))
        ok = ok and is_full(cur_rem) == is_full(max(0, _l(cap - used)))
        ok = ok and cur["loading_threads"] == threads
        ok = ok and not _loading_change(cur, in_prog)
        ok = ok and not _load_change(cur["rpm"], rpms)
        if ok:
            return False
    elif (bool(cur["shutting_down"]) == shutting and cur["capacity"] == cap and cur["count"] == count
          and cur["lru_time"] == oldest and cur["used"] == used and cur["loading_threads"] == threads
          and cur["loading_in_progress"] == in_prog and cur["rpm"] == rpms):
        return False
    return True
