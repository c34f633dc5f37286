"""Python restatement of UpgradeTracker.java:47-201 (TEST INFRASTRUCTURE ONLY).

`labels_key` plays the role of the identity of the record's labels array: the reference's
`HashMap<String[], PerTypeLabelStats>` (:67) hashes arrays by identity (NO_LABELS is shared,
InstanceRecord.java:35,89).  No reference test names this class; held to the
reference's own text since round 3 (oracle/ref_harness, tests/test_ref_vectors.py: twelve rolling-update streams)."""
TEN_MINS, FIFTEEN_MINS, TWENTY_MINS = 600_000, 900_000, 1_200_000
LONG_MAX = 2**63 - 1


class ReplicaSetStats:
    def __init__(self):
        self.size, self.earliestStartTime, self.latestStartTime, self.lastChangeTime = 0, LONG_MAX, 0, 0


class UpgradeTracker:
    def __init__(self):
        self.upgradeTracker = {}
        self.likelyReplacedReplicaSets = {}

    def instanceRemoved(self, labels_key, rs, now):  # :85-116
        if rs < 0:
            return
        ptls = self.upgradeTracker.get(labels_key)
        if ptls is None:
            return
        rss = ptls.get(rs)
        if rss is not None:
            rss.size -= 1
            if rss.size > 0:
                rss.lastChangeTime = now
            else:
                del ptls[rs]
            if rs in self.likelyReplacedReplicaSets:
                repl = dict(self.likelyReplacedReplicaSets)
                if rss.size <= 0:
                    del repl[rs]
                else:
                    repl[rs] = rss.lastChangeTime + FIFTEEN_MINS
                self.likelyReplacedReplicaSets = repl

    def instanceAdded(self, labels_key, rs, startTime, now):  # :121-186
        if rs < 0:
            return
        ptls = self.upgradeTracker.setdefault(labels_key, {})
        rss = ptls.get(rs)
        if rss is None:
            rss = ptls[rs] = ReplicaSetStats()
        rss.lastChangeTime = now
        rss.size += 1
        rss.earliestStartTime = min(rss.earliestStartTime, startTime)
        rss.latestStartTime = max(rss.latestStartTime, startTime)
        old = set()
        if len(ptls) > 1:
            newest = max(ptls.values(), key=lambda r: r.earliestStartTime)
            if newest.latestStartTime > now - TWENTY_MINS:
                old = {k for k, v in ptls.items()
                       if v.latestStartTime < newest.earliestStartTime and
                       (newest.latestStartTime > now - TEN_MINS or v.lastChangeTime > now - FIFTEEN_MINS)}
        if not self.likelyReplacedReplicaSets and not old:
            return
        repl = None
        for k, v in ptls.items():
            if k in old:
                if k not in self.likelyReplacedReplicaSets:
                    repl = repl if repl is not None else dict(self.likelyReplacedReplicaSets)
                    repl[k] = v.lastChangeTime + FIFTEEN_MINS
            elif k in self.likelyReplacedReplicaSets:
                repl = repl if repl is not None else dict(self.likelyReplacedReplicaSets)
                del repl[k]
        if repl is not None:
            self.likelyReplacedReplicaSets = repl

    def doHousekeeping(self, now):  # :191-200
        self.likelyReplacedReplicaSets = {k: e for k, e in self.likelyReplacedReplicaSets.items() if not now >= e}
