"""py_oracle.py — a SECOND, independently written restatement of the reference's
selection logic, in pure Python integers, used only to cross-check the C oracle
(oracle/mm_oracle.c) on random snapshots.  TEST INFRASTRUCTURE ONLY.

Written from the Java text (MM.java = src/main/java/com/ibm/watson/modelmesh/
ModelMesh.java), not from the C: real iterators, Python lists for `candidates`
and `clusterStateReplay`, dicts for records.  Slow on purpose.
"""
from __future__ import annotations

from functools import cmp_to_key

I64 = (1 << 64) - 1
LONG_MAX = (1 << 63) - 1
INT_MAX = (1 << 31) - 1
NONE, SELF = -1, -2  # null / LoadBalancer.ABORT_REQUEST
BRANCHES = set()  # branch tags visited by get_next (tests assert the fuzz fleets reach all of them)


def _hit(tag):
    BRANCHES.add(tag)


def _l(x):  # wrap to Java long
    x &= I64
    return x - (1 << 64) if x >> 63 else x


def _i(x):  # wrap to Java int
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x


def _jdiv(a, b):  # Java '/' truncates toward zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _d2i(d):  # (int) of a double
    if d != d:
        return 0
    if d >= 2147483647.0:
        return INT_MAX
    if d <= -2147483648.0:
        return -INT_MAX - 1
    return int(d)


def remaining(ir):  # InstanceRecord.java:203-205
    return max(0, _l(ir["capacity"] - ir["used"]))


class Mesh:
    """The state CacheMissForwardingLB reads: parameters + typeConstraints + upgradeTracker."""

    def __init__(self, min_space_units, min_churn_age_ms, now):
        self.minSpaceUnits = min_space_units
        self.minChurnAgeMs = min_churn_age_ms
        self.now = now

    def isFull(self, avail):  # MM.java:4640-4642
        return avail < self.minSpaceUnits

    def age(self, t):  # MM.java:4162-4164
        return 0 if t == 0 else _l(self.now - t)

    # MM.java:4646-4703; entries are (id_order, record)
    def placement_order(self, e1, e2):
        ir1, ir2 = e1[1], e2[1]
        if ir1 is ir2:
            return (e1[0] > e2[0]) - (e1[0] < e2[0])
        sd1 = ir1["shutting_down"]
        if sd1 != ir2["shutting_down"]:
            return 1 if sd1 else -1
        vers1, vers2 = ir1["version"], ir2["version"]
        rem1, rem2 = remaining(ir1), remaining(ir2)
        full1, full2 = self.isFull(rem1), self.isFull(rem2)
        if vers1 != vers2:
            if vers1 > vers2:
                if not full1 or ir1["lru_time"] > _l(self.minChurnAgeMs * 2):
                    return -1
            elif not full2 or ir2["lru_time"] > _l(self.minChurnAgeMs * 2):
                return 1
        if full1 != full2:
            return 1 if full1 else -1
        if full1:
            if ir1["lru_time"] != ir2["lru_time"]:
                return -1 if ir1["lru_time"] < ir2["lru_time"] else 1
        countDiff = _i(ir1["count"] - ir2["count"])
        if countDiff != 0:
            return countDiff
        if rem1 != rem2:
            return -1 if rem2 < rem1 else 1
        if not full1:
            if ir1["lru_time"] != ir2["lru_time"]:
                return -1 if ir1["lru_time"] < ir2["lru_time"] else 1
        lip1, lip2 = ir1["loading_in_progress"], ir2["loading_in_progress"]
        chain = [(_i(ir2["loading_threads"] - lip2), _i(ir1["loading_threads"] - lip1)), (lip1, lip2),
                 (ir2["capacity"], ir1["capacity"]), (ir1["rpm"], ir2["rpm"]), (e1[0], e2[0])]
        for a, b in chain:
            if a != b:
                return -1 if a < b else 1
        return 0

    def sorted_cluster_state(self, pods):
        """pods: list of record dicts (index = pod id). Returns pod ids in clusterState order;
        shutting-down records are never in the set (MM.java:1462-1464)."""
        ents = [(p["id_order"], p, i) for i, p in enumerate(pods) if not p["shutting_down"]]
        ents.sort(key=cmp_to_key(lambda a, b: self.placement_order(a, b)))
        return [e[2] for e in ents]


def get_next(mesh, pods, cluster_state, live, replaced_rs, constrain_to, prefer, exclude_sets, self_id,
             favour_self, fresh, last_used_time, pick):
    """CacheMissForwardingLB.getNext, MM.java:4776-5005.

    constrain_to / prefer: set of pod ids or None; exclude_sets: iterable of sets
    (the HashSet itself, loaded, failed, explicit).  Returns (chosen, bestIid, candidates, remaining)."""

    def isExcluded(iid):  # MM.java:4740-4743
        return any(iid in s for s in exclude_sets)

    def filt(exclude_replica_sets):  # MM.java:4760-4772
        for iid in cluster_state:
            if (constrain_to is not None and iid not in constrain_to) or isExcluded(iid) or iid not in live:
                continue
            rs = pods[iid]["replica_set"]
            if not exclude_replica_sets or rs < 0 or rs not in exclude_replica_sets:
                yield iid

    class It:  # java.util.Iterator with hasNext
        def __init__(self, gen):
            self.g = iter(gen)
            self.buf = []

        def hasNext(self):
            if not self.buf:
                try:
                    self.buf.append(next(self.g))
                except StopIteration:
                    return False
            return True

        def next(self):
            assert self.hasNext()
            return self.buf.pop()

    excludeSelf = isExcluded(self_id)
    it = It(filt(replaced_rs))
    if not it.hasNext():
        if not replaced_rs:
            _hit("none_no_candidates")
            return NONE, -1, [], 0
        it = It(filt(set()))
        _hit("retry_without_replica_sets")
        if not it.hasNext():
            _hit("none_after_retry")
            return NONE, -1, [], 0
    bestEntry = it.next()
    bestIid = bestEntry
    us = (not excludeSelf) and self_id == bestIid
    bestInst = fresh if us else pods[bestEntry]
    bestIsFull = mesh.isFull(remaining(bestInst))
    candidates, instReqLoad = [], []

    simpleCase = prefer is None or bestIid in prefer
    if not simpleCase:
        replay = []
        if not bestIsFull:
            found = False
            while it.hasNext():
                ent = it.next()
                if ent in prefer:
                    _hit("case_a_found")
                    found = True
                    bestIid = ent
                    bestInst = pods[ent]
                    us = (not us) and (not excludeSelf) and self_id == bestIid
                    break
                if mesh.isFull(remaining(pods[ent])):
                    _hit("case_a_stop_at_full")
                    break
                replay.append(ent)
            if not found:
                _hit("case_a_not_found")
                it = It(replay)
                prefer = None
            simpleCase = True
        else:
            oldest = bestInst["lru_time"]
            while it.hasNext():
                iid = it.next()
                curInst = pods[iid]
                diff = _l(curInst["lru_time"] - oldest)
                if diff > 120_000 and diff > _jdiv(mesh.age(oldest), 4):
                    _hit("case_b_age_break")
                    break
                if iid in prefer:
                    us = (not us) and (not excludeSelf) and self_id == iid
                    if us and favour_self:
                        _hit("case_b_self_null")
                        return NONE, bestIid, [], 0
                    _hit("case_b_preferred")
                    replay = None
                    candidates.append(iid)
                    instReqLoad.append(curInst["rpm"])
                elif replay is not None:
                    replay.append(iid)
            if replay is not None:
                _hit("case_b_no_preferred")
                it = It(replay)
                prefer = None
                simpleCase = True

    if simpleCase:
        if us and favour_self:
            _hit("self_is_best")
            return SELF, bestIid, [], 0
        candidates.append(bestIid)
        instReqLoad.append(bestInst["rpm"])
        oldest = bestInst["lru_time"]
        while it.hasNext():
            iid = it.next()
            if prefer is not None and iid not in prefer:
                _hit("skip_non_preferred")
                continue
            us = (not us) and (not excludeSelf) and self_id == iid
            curInst = pods[bestEntry] if us else fresh
            if bestIsFull:
                diff = _l(curInst["lru_time"] - oldest)
                _hit("full_mode_self" if us else "full_mode")
                if diff > 45_000 and diff > _jdiv(mesh.age(oldest), 10):
                    _hit("break_lru_self" if us else "break_lru")
                    break
            else:
                rem = remaining(curInst)
                if mesh.isFull(rem) or rem < (remaining(bestInst) >> 2):
                    _hit("break_rem_self" if us else "break_rem")
                    break
                count, firstCount = pods[iid]["count"], bestInst["count"]
                if count >= 10 and count > _i(firstCount + (firstCount >> 2)):
                    _hit("break_count")
                    break
            if us and favour_self:
                _hit("self_in_shortlist")
                return SELF, bestIid, [], 0
            candidates.append(iid)
            instReqLoad.append(curInst["rpm"])

    ccount = len(candidates)
    if ccount == 0:
        _hit("empty_candidates")
        return NONE, bestIid, [], 0
    shortlist = list(candidates)
    lastUsedAgo = mesh.age(last_used_time)
    if ccount == 1:
        chosen, remainingCount = candidates[0], 1
    else:
        remainingCount = ccount
        if not lastUsedAgo < 5 * 86_400_000:
            _hit("older_than_five_days")
        if lastUsedAgo < 5 * 86_400_000:
            minLoad = max(100, min(instReqLoad))
            m11, m15 = _d2i(1.1 * minLoad), _d2i(1.5 * minLoad)
            for i in range(ccount):
                rpm = instReqLoad[i]
                if rpm >= 100 and ((lastUsedAgo < -1000 and rpm > m11) or (lastUsedAgo < 5000 and rpm > m15)
                                   or (lastUsedAgo < 720_000 and rpm > _i(minLoad * 3))
                                   or (lastUsedAgo < 86_400_000 and rpm > _i(minLoad * 4))):
                    _hit("rpm_nulled_first" if i == 0 else "rpm_nulled_other")
                    candidates[i] = None
                    remainingCount -= 1
                    if remainingCount == 1:
                        _hit("rpm_break_at_one")
                        break
        index = 0 if remainingCount == 1 else (pick * remainingCount) >> 32
        chosen, j = None, 0
        for i in range(ccount):
            chosen = candidates[i]
            if chosen is not None:
                if index == j:
                    break
                j += 1
    if (not favour_self) and self_id == chosen:
        _hit("chosen_is_self")
        return SELF, bestIid, shortlist, remainingCount
    return (NONE if chosen is None else chosen), bestIid, shortlist, remainingCount


def serve_get_next(filtered, self_id, exclude_self, prefer_self, live, now, assume_completed_ms, local_in_flight,
                   last_invoke_time, in_use, last_used):
    """ForwardingLB.getNext, MM.java:4315-4392.  filtered: list of (iid, loadStart) in TreeMap order."""
    if not filtered:
        return NONE, 0
    seenSelf = False
    chosen, chosenTs = None, 0
    mn, lru, firstStarted = INT_MAX, LONG_MAX, LONG_MAX
    cutoff = -1
    for iid, loadStarted in filtered:
        us = False
        if not seenSelf and iid == self_id:
            seenSelf = True
            if exclude_self:
                continue
            us = True
        if iid not in live:
            continue
        if cutoff == -1:
            cutoff = _l(now - assume_completed_ms)
        if loadStarted < cutoff:
            inuse = local_in_flight if us else in_use[iid]
            if inuse > mn:
                continue
            nlu = (0 if prefer_self else last_invoke_time) if us else last_used[iid]
            if inuse < mn:
                mn = inuse
            elif nlu >= lru:
                continue
            chosen, chosenTs, lru = iid, loadStarted, nlu
        elif mn == INT_MAX and loadStarted < firstStarted:
            chosen, chosenTs, firstStarted = iid, loadStarted, loadStarted
    if chosen is None:
        return NONE, chosenTs
    if not exclude_self and chosen == self_id:
        return SELF, chosenTs
    return chosen, chosenTs


class Clhm:
    """clhm/ConcurrentLinkedHashMap.java in its drained, single-threaded order."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.deque = []  # [key, lastUsed, weight], head first
        self.weightedSize = 0

    def _insert(self, node):  # LinkedDeque.java:259-288
        l = len(self.deque) - 1
        while l >= 0:
            if self.deque[l][1] <= node[1]:
                break
            l -= 1
        self.deque.insert(l + 1, node)
        return l + 1

    def _reposition(self, node):  # LinkedDeque.java:243-256
        i = self.deque.index(node)
        lu = node[1]
        if i == 0 or self.deque[i - 1][1] <= lu:
            if i == len(self.deque) - 1 or self.deque[i + 1][1] >= lu:
                return
        self.deque.pop(i)
        self._insert(node)

    def _evict(self):  # clhm :329-352
        out = []
        self.lastEvictedWeights = []  # the weights the listener sees (CacheEntry.getWeight())
        while self.weightedSize > self.capacity:
            if not self.deque:
                break
            node = self.deque.pop(0)
            self.weightedSize -= abs(node[2])
            out.append(node[0])
            self.lastEvictedWeights.append(node[2])
        return out

    def _find(self, key):
        for n in self.deque:
            if n[0] == key:
                return n
        return None

    @staticmethod
    def _touch(node, t, now):  # clhm :1357-1360
        node[1] = now if t == 0 else max(node[1], t)

    def putIfAbsent(self, key, weight, lastUsed, now):
        prior = self._find(key)
        if prior is not None:
            self._touch(prior, lastUsed, now)
            self._reposition(prior)
            return None
        node = [key, 0, weight]
        self._touch(node, lastUsed, now)
        self.weightedSize += weight
        self._insert(node)
        return self._evict()

    def get(self, key, lastUsed, now):
        n = self._find(key)
        if n is None:
            return False
        self._touch(n, lastUsed, now)
        self._reposition(n)
        return True

    def updateWeight(self, key, newWeight, newTime, now):
        """replace/replaceQuietly + UpdateTask (clhm :629-652); newTime -1 = quiet."""
        n = self._find(key)
        if n is None:
            return None
        diff = newWeight - n[2]
        n[2] = newWeight
        if diff == 0:
            if newTime >= 0:
                self._touch(n, newTime, now)
                self._reposition(n)
            return []
        self.weightedSize += diff
        if newTime >= 0 and newTime != n[1]:
            self._touch(n, newTime, now)
            self._reposition(n)
        return self._evict()

    def remove(self, key):
        n = self._find(key)
        if n is None:
            return False
        self.deque.remove(n)
        self.weightedSize -= abs(n[2])
        return True

    def oldestTime(self):
        return self.deque[0][1] if self.deque else -1

    def keys(self):
        return [n[0] for n in self.deque]


class UnloadBufManager:
    """ModelCacheUnloadBufManager.java: a pinned pseudo-entry whose weight is
    max(reserved, totalUnloadingWeight), so models that are evicted but still unloading keep
    occupying capacity.  `cache` is a Clhm; evictions it reports are fed back through
    entryRemoved exactly as ModelMesh.onEviction does (MM.java:2876-2878)."""
    KEY = "___UNLOADBUF"

    def __init__(self, cache: Clhm, reserved: int, now: int):
        self.cache = cache
        self.reserved = reserved
        self.totalUnloadingWeight = 0
        self.totalModelCacheOccupancy = 0
        self.cacheDeficit = 0
        self.evicted = []  # (key, weight) in eviction order
        self.weights = {}  # key -> current weight (CacheEntry.getWeight())
        cache.putIfAbsent(self.KEY, reserved, LONG_MAX, now)  # MM.java:1617-1622

    def buffer_weight(self):
        n = self.cache._find(self.KEY)
        return n[2] if n is not None else 0  # evicted itself: only with a capacity below the reserve

    def _on_evicted(self, keys):  # ModelMesh.onEviction → entryRemoved, still under the lock
        ws = list(self.cache.lastEvictedWeights) if keys else []
        for k, w in zip(keys or [], ws):
            self.weights.pop(k, None)  # the pinned buffer entry itself can be evicted (pathological capacity)
            self.evicted.append((k, w))
            self.entryRemoved(w)

    def _set_weight(self, key, w, now):  # CacheEntry.updateWeightLocked → replaceQuietly
        if self.cache._find(key) is None:
            return  # replaceQuietly fails and the weight is restored, MM.java:1784-1786
        if key != self.KEY:
            self.weights[key] = w
        self._on_evicted(self.cache.updateWeight(key, w, -1, now))

    def _adjustAggregateUnloadingWeight(self, delta, now):  # :375-392
        if delta == 0:
            return
        self.totalUnloadingWeight += delta
        nw = self.totalUnloadingWeight
        if nw <= self.reserved:
            nw = self.reserved
        else:
            cap = min(self.cache.capacity, INT_MAX)
            if cap < nw:
                nw = cap
        if nw != self.buffer_weight():
            self._set_weight(self.KEY, nw, now)

    def insertNewEntry(self, key, weight, lastUsed, now):  # :130-145
        self._adjustAggregateUnloadingWeight(-weight, now)
        ev = self.cache.putIfAbsent(key, weight, lastUsed, now)
        if ev is None:
            self._adjustAggregateUnloadingWeight(weight, now)
            return False
        self.weights[key] = weight
        self.totalModelCacheOccupancy += weight
        self._on_evicted(ev)
        return True

    def adjustNewEntrySpaceRequest(self, increase, key, now):  # :152-166
        nw = self.weights[key] + increase
        self.totalModelCacheOccupancy += increase
        self._adjustAggregateUnloadingWeight(-increase, now)
        if key in self.weights:  # still in the cache
            self._set_weight(key, nw, now)

    def cacheSpaceIsReady(self, required):  # :395-402
        newTuw = self.totalUnloadingWeight + required
        if newTuw <= self.reserved:
            return True
        return newTuw + self.totalModelCacheOccupancy <= self.cache.capacity

    def claimRequestedSpaceIfReady(self, required, now):  # :190-202
        if self.cacheSpaceIsReady(required):
            self._adjustAggregateUnloadingWeight(required, now)
            return True
        return False

    def _cacheRemaining(self):
        return min(self.cache.capacity - self.cache.weightedSize, INT_MAX)

    def adjustWeightAfterLoad(self, delta, key, now):  # :224-246
        if delta == 0:
            return
        if delta > 0:
            deficit = delta - self._cacheRemaining()
            if deficit > 0:
                self._adjustAggregateUnloadingWeight(-deficit, now)
                self.cacheDeficit += deficit
        self.totalModelCacheOccupancy += delta
        self._set_weight(key, self.weights[key] + delta, now)
        if delta < 0:
            self._payDown(-delta, False, now)

    def entryRemoved(self, weight, now=0):  # :311-316
        self.totalModelCacheOccupancy -= weight
        self._adjustAggregateUnloadingWeight(weight, now)

    def _payDown(self, weight, release, now):  # :351-366
        reduction = min(weight, self.cacheDeficit)
        if reduction:
            self.cacheDeficit -= reduction
            weight -= reduction
        self._adjustAggregateUnloadingWeight(-weight if release else reduction, now)

    def unloadComplete(self, weight, success, now):  # :318-338
        if success:
            self._payDown(weight, True, now)
            return
        cap = self.cache.capacity
        self._adjustAggregateUnloadingWeight(-weight, now)
        self.cache.capacity = max(1, cap - weight)
        self._on_evicted(self.cache._evict())  # setCapacity evicts + notifies under the lock, clhm :305-316

    def removeEntry(self, key, now):  # :281-298
        n = self.cache._find(key)
        if n is None:
            return -1
        w = n[2]
        self.cache.remove(key)
        self.weights.pop(key, None)
        self.entryRemoved(w, now)
        return w

    def discardFailedEntry(self, weight, now):  # :343-349
        self.totalModelCacheOccupancy -= weight
        self._payDown(weight, False, now)

    def insertFailedPlaceholderEntry(self, key, weight, lastUsed, now):  # :250-274
        deficit = weight - self._cacheRemaining()
        if deficit > 0:
            self._adjustAggregateUnloadingWeight(-deficit, now)
        ev = self.cache.putIfAbsent(key, weight, lastUsed, now)
        if ev is None:
            if deficit > 0:
                self._adjustAggregateUnloadingWeight(deficit, now)
            return False
        self.weights[key] = weight
        self._on_evicted(ev)
        self.totalModelCacheOccupancy += weight
        if deficit > 0:
            self.cacheDeficit += deficit
        return True

    def adjusted_capacity(self):  # getAdjustedCacheCapacity :90-92
        return self.cache.capacity - self.buffer_weight()
