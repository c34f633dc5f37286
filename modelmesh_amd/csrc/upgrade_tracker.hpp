// upgrade_tracker.hpp — host-side state of UpgradeTracker (UpgradeTracker.java:47-201): which replica
// sets are "very likely being replaced by a rolling update" and must be avoided by the load-target
// filter unless nothing else is eligible (MM.java:4769-4770, :4792-4805).  It is sequential
// bookkeeping over a handful of entries fed by instance-table events, so it lives on the host side of
// the library; its output (the replica-set list) is what the commit turns into the `elig` bitmap.
//
// `labels_key` stands for the identity of the InstanceRecord's labels array: the reference keys its map
// with a String[] (HashMap<String[], PerTypeLabelStats>, :67), i.e. by object identity — all unlabelled
// records share the static NO_LABELS array (InstanceRecord.java:35,89), every labelled record has its
// own.  The caller passes whatever reproduces that identity (0 for NO_LABELS).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <map>
#include <set>
#include <vector>

namespace mmp {

struct UpgradeTracker {
    static constexpr int64_t kTenMins = 600000, kFifteenMins = 900000, kTwentyMins = 1200000;
    struct RsStats {
        int32_t size = 0;
        int64_t earliest_start = INT64_MAX, latest_start = 0, last_change = 0;
    };
    // insertion-ordered like the small HashMaps they mirror is not needed: results are sets
    std::map<int64_t, std::map<int32_t, RsStats>> tracker;  // labels_key -> replica set -> stats
    std::map<int32_t, int64_t> replaced;                   // replica set -> expiry time

    // instanceRemoved, :85-116 (replica_set < 0: instance id shorter than 7 chars)
    void instance_removed(int64_t labels_key, int32_t rs, int64_t now)
    {
        if (rs < 0) return;
        auto pt = tracker.find(labels_key);
        if (pt == tracker.end()) return;
        auto it = pt->second.find(rs);
        if (it == pt->second.end()) return;
        RsStats st = it->second;
        st.size--;
        if (st.size > 0) {
            st.last_change = now;
            it->second = st;
        } else
            pt->second.erase(it);
        if (replaced.count(rs)) {
            if (st.size <= 0)
                replaced.erase(rs);
            else
                replaced[rs] = st.last_change + kFifteenMins;
        }
    }

    // instanceAdded, :121-186
    void instance_added(int64_t labels_key, int32_t rs, int64_t start_time, int64_t now)
    {
        if (rs < 0) return;
        auto &ptls = tracker[labels_key];
        RsStats &st = ptls[rs];
        st.last_change = now;
        st.size++;
        if (start_time < st.earliest_start) st.earliest_start = start_time;
        if (start_time > st.latest_start) st.latest_start = start_time;
        std::set<int32_t> old;
        if (ptls.size() > 1) {
            // replica set with the same labels that was started most recently (max earliestStartTime).
            // Equal maxima: Stream.max keeps the first in the HashMap's iteration order, which is not
            // reproducible; replica sets of one Deployment never start in the same millisecond.
            const RsStats *newest = nullptr;
            for (auto &e : ptls)
                if (!newest || e.second.earliest_start > newest->earliest_start) newest = &e.second;
            if (newest->latest_start > now - kTwentyMins) {
                for (auto &e : ptls)
                    if (e.second.latest_start < newest->earliest_start &&
                        (newest->latest_start > now - kTenMins || e.second.last_change > now - kFifteenMins))
                        old.insert(e.first);
            }
        }
        if (replaced.empty() && old.empty()) return;
        for (auto &e : ptls) {
            if (old.count(e.first)) {
                if (!replaced.count(e.first)) replaced[e.first] = e.second.last_change + kFifteenMins;
            } else
                replaced.erase(e.first);
        }
    }

    // doHousekeeping, :191-200
    void housekeeping(int64_t now)
    {
        for (auto it = replaced.begin(); it != replaced.end();)
            it = now >= it->second ? replaced.erase(it) : std::next(it);
    }
};

}  // namespace mmp
