/*
 * GpuPlacementLB.java — the reference-side binding a ModelMesh maintainer adds.
 * Lives in package com.ibm.watson.modelmesh next to ModelMesh.java.  The two litelinks clients are built at
 * ModelMesh.java:1103-1110; the patch replaces the two LoadBalancer factories there:
 *
 *     runtimeClient   = ... .withLoadBalancer(gpu ? () -> new GpuForwardingLB(binding)  : ForwardingLB::new) ...
 *     cacheMissClient = ... .withLoadBalancer(gpu ? () -> new GpuCacheMissLB(binding)   : CacheMissForwardingLB::new) ...
 *
 * or, to PIN the solver against the reference on live traffic before switching (INTEGRATION.md §4):
 *
 *     ... .withLoadBalancer(() -> new ShadowCacheMissLB(new CacheMissForwardingLB(), new GpuCacheMissLB(binding), binding))
 *     ... .withLoadBalancer(() -> new ShadowForwardingLB(new ForwardingLB(), new GpuForwardingLB(binding), binding))
 *
 * This repository's image has no JDK, so nothing here is compiled in CI; tests/test_jni_veneer.py checks the
 * `native` declarations against integration/mmplace_jni.cc (names, arity, types) and against include/mmplace.h.
 * Everything the classes need from the enclosing ModelMesh instance goes through GpuMeshBinding, which the
 * patch implements as an inner class of ModelMesh (it reads instanceId, clusterState, registry, the two
 * ThreadLocals and the litelinks ServiceInstance counters).
 */
package com.ibm.watson.modelmesh;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.Collection;
import java.util.HashSet;
import java.util.Map;
import java.util.Set;
import java.util.concurrent.ThreadLocalRandom;
import java.util.concurrent.atomic.AtomicLong;

import com.ibm.watson.litelinks.ThreadContext;
import com.ibm.watson.litelinks.client.LoadBalancer;
import com.ibm.watson.litelinks.client.ServiceInstance;
import com.ibm.watson.litelinks.client.ServiceInstanceInfo;

/** static natives implemented by integration/mmplace_jni.cc */
final class MmPlace {
    static { System.loadLibrary("mmplace_jni"); }
    static native long create(int device, long minSpaceUnits, long minChurnAgeMs);
    static native void destroy(long h);
    static native int podsLoad(long h, ByteBuffer rows, int n);
    static native int podsUpsert(long h, ByteBuffer idx, ByteBuffer rows, int n);
    static native int podsRemove(long h, ByteBuffer idx, int n);
    static native int typesLoad(long h, int nTypes, ByteBuffer allowed, ByteBuffer prefer,
                                ByteBuffer hasAllowed, ByteBuffer hasPrefer);
    static native int replacedReplicaSetsLoad(long h, ByteBuffer rs, int n);
    static native int modelsLoad(long h, ByteBuffer rows, int nModels, ByteBuffer entPod, ByteBuffer entTime, int nEntries);
    static native int modelsUpsert(long h, ByteBuffer idx, ByteBuffer rows, int n, ByteBuffer entPod, ByteBuffer entTime, int nEntries);
    static native int commit(long h);
    static native int placeBatch(long h, ByteBuffer reqs, int n, ByteBuffer extraPool, int nExtra, long nowMs, ByteBuffer outs);
    /** the single-caller form: caller = one mmp_place_caller (this instance's index, favourSelf, getFreshInstanceRecord()),
     *  reqs = n mmp_place_req_c rows of 24 bytes {model, pick, lastUsedTime, extraOff, nExtra} — what the rate task, the janitor,
     *  the reaper and preShutdown issue: every decision of such a batch has the same self */
    static native int placeBatchCaller(long h, ByteBuffer caller, ByteBuffer reqs, int n, ByteBuffer extraPool, int nExtra,
                                       long nowMs, ByteBuffer outs);
    static native int serveBatch(long h, ByteBuffer reqs, int n, ByteBuffer counters, int nCounters,
                                 ByteBuffer exclPod, ByteBuffer exclTime, int nExcl, long nowMs, ByteBuffer outs);
    /** keep one wavefront resident that serves placeBatch(n = 1) from 64 pinned request slots: no launch per request */
    static native int resident(long h, boolean enable);
    // the pod-axis group: several GPUs of one node, RCCL runs inside libmmplace (include/mmplace.h)
    static native int shardUniqueId(ByteBuffer idOut);
    static native int shardGroupInit(long h, ByteBuffer id, int rank, int world);
    static native int shardGroupDestroy(long h);
    static native int shardCommit(long h);
    static native int shardPlaceBatch(long h, ByteBuffer reqs, int n, ByteBuffer extraPool, int nExtra, long nowMs,
                                      ByteBuffer outs, ByteBuffer nRestOut);
    static native int clusterStats(long h, ByteBuffer out);
    static native int typeStats(long h, int type, ByteBuffer out);
    static native int partitionCount(long h, ByteBuffer nOut);
    static native int partitionStats(long h, int partition, ByteBuffer out, ByteBuffer prohibitedOut, int maxWords);
    static native int podPartitions(long h, ByteBuffer partitionOut, int maxPods, ByteBuffer nOut);
    // eviction (clhm) and the unload-buffer manager
    static native int cachesLoad(long h, int nCaches, ByteBuffer segOff, ByteBuffer lastUsed, ByteBuffer weight,
                                 ByteBuffer capacity);
    static native int evictBatch(long h, ByteBuffer reqs, int n, long nowMs, ByteBuffer outs);
    static native int cachesLoadKeyed(long h, int nCaches, ByteBuffer segOff, ByteBuffer lastUsed, ByteBuffer weight,
                                      ByteBuffer key, ByteBuffer capacity, ByteBuffer ubm);
    static native int cacheReplay(long h, ByteBuffer ops, int nOps, long nowMs, ByteBuffer outs, ByteBuffer evictedKeys,
                                  int maxEvicted, ByteBuffer nEvictedSlots);
    static native int cacheRead(long h, int cache, int maxEntries, ByteBuffer lastUsed, ByteBuffer weight, ByteBuffer key,
                                ByteBuffer scalars, ByteBuffer ubm);
    // request guards and rebalancer plans
    static native int gateBatch(long h, ByteBuffer reqs, int n, ByteBuffer exclPod, ByteBuffer exclTime, int nExcl,
                                ByteBuffer explicitPool, int nExplicit, long nowMs, long inUseFailureExpiryMs,
                                ByteBuffer outs);
    static native int missBatch(long h, ByteBuffer gateReqs, ByteBuffer placeReqs, int n, ByteBuffer exclPod, ByteBuffer exclTime, int nExcl,
                                ByteBuffer explicitPool, int nExplicit, ByteBuffer extraPool, int nExtra, long nowMs,
                                long inUseFailureExpiryMs, ByteBuffer gateOuts, ByteBuffer placeOuts);
    static native int routeBatch(long h, ByteBuffer gateReqs, ByteBuffer serveReqs, int n, ByteBuffer counters, int nCounters,
                                 ByteBuffer exclPod, ByteBuffer exclTime, int nExcl, ByteBuffer explicitPool, int nExplicit,
                                 long nowMs, long inUseFailureExpiryMs, ByteBuffer gateOuts, ByteBuffer serveOuts);
    static native int proactivePlan(long h, int defaultModelSizeUnits, long nowMs, int maxOut, ByteBuffer outModel,
                                    ByteBuffer outLastUsed, ByteBuffer info);
    static native int proactivePlanSubset(long h, int partition, ByteBuffer skipModels, int nSkip, int defaultModelSizeUnits,
                                          long nowMs, int maxOut, ByteBuffer outModel, ByteBuffer outLastUsed, ByteBuffer info);
    static native int scaleupPlan(long h, ByteBuffer entries, int n, ByteBuffer params, ByteBuffer outs,
                                  ByteBuffer overloadedOut, ByteBuffer skipped);
    static native int scaledownPlan(long h, ByteBuffer entries, int n, ByteBuffer params, ByteBuffer removedOut);
    // limitModelConcurrency == true: mmp_conc_entry rows (MaxConcCacheEntry: countAndTimeSum.sum(), priorSum, priorCount, maxConc,
    // queuedRequestCount()) beside the entries; concOuts tell the caller which entries to sumThenReset() and what to store in
    // priorSum / priorCount, result carries the task's averageModelParallelism after the run (a double at offset 0)
    static native int scaleupPlanConc(long h, ByteBuffer entries, ByteBuffer conc, int n, ByteBuffer params, ByteBuffer concParams,
                                      ByteBuffer outs, ByteBuffer concOuts, ByteBuffer overloadedOut, ByteBuffer skipped,
                                      ByteBuffer result);
    static native int scaledownPlanConc(long h, ByteBuffer entries, ByteBuffer conc, int n, ByteBuffer params,
                                        long dynamicRpmScaleConstant, ByteBuffer removedOut);
    static native int migrationPlan(long h, ByteBuffer entries, int n, int selfPod, long nowMs, long cutoffAgeMs,
                                    ByteBuffer actionOut, ByteBuffer waitOut);
    // type constraints, upgrade tracker
    static native int typesFromLabels(long h, int nTypes, ByteBuffer required, ByteBuffer preferred, ByteBuffer podLabels,
                                      ByteBuffer allowedOut, ByteBuffer preferOut, ByteBuffer hasAllowedOut,
                                      ByteBuffer hasPreferOut);
    static native int upgradeInstanceAdded(long h, long labelsKey, int replicaSet, long startTime, long nowMs);
    static native int upgradeInstanceRemoved(long h, long labelsKey, int replicaSet, long nowMs);
    static native int upgradeHousekeeping(long h, long nowMs);
    static native int upgradeReplaced(long h, ByteBuffer rsOut, ByteBuffer expiryOut, int max, ByteBuffer nOut);
    // KV wire format (the raw values of the KV events)
    static native int podIdsLoad(long h, ByteBuffer ids, ByteBuffer idOff, int nPods, ByteBuffer idOrderOut,
                                 ByteBuffer replicaSetOut);
    static native int podsIngestJson(long h, ByteBuffer json, ByteBuffer off, int n, ByteBuffer podIdx, ByteBuffer live,
                                     ByteBuffer startTimeOut, ByteBuffer statusOut);
    static native int typeNamesLoad(long h, ByteBuffer names, ByteBuffer nameOff, int nTypes, int unknownType);
    static native int modelsIngestJson(long h, ByteBuffer json, ByteBuffer off, int nModels, ByteBuffer lastUnloadOut,
                                       ByteBuffer statusOut);
    // misc
    static native long minSpaceUnits(int defaultModelSizeUnits, int loadingThreads, long capacityUnits,
                                     boolean haveUnloadManager);
    static native int getOrder(long h, ByteBuffer orderOut, ByteBuffer nOut);
    static native int deltaCommits(long h, ByteBuffer nOut);
    static native int profile(long h, boolean enable);
    static native double lastKernelMs(long h);
    static native int abiVersion();
    static final int NONE = -1, SELF = -2;

    static ByteBuffer direct(int bytes) { return ByteBuffer.allocateDirect(bytes).order(ByteOrder.LITTLE_ENDIAN); }
}

/**
 * What the GPU load balancers need from the enclosing ModelMesh instance.  The patch implements it as an inner
 * class of ModelMesh: the interner maps instance id <-> dense pod index (id_order == rank under String.compareTo,
 * replica_set == interned id.substring(0,6)) and model id -> dense model index; the snapshot handle is refreshed by
 * the instance-table listener (handleInstanceTableChange, ModelMesh.java:1455: podsUpsert / podsRemove + commit) and
 * by the registry listener (modelsLoad once, modelsUpsert per ModelRecord event).
 */
interface GpuMeshBinding {
    long handle();
    int podCount();
    int podIndexOf(String instanceId);      // -1 if unknown
    String instanceIdOf(int podIndex);
    int modelIndexOf(String modelId);       // -1 if unknown
    String selfInstanceId();
    boolean sendDestinationId();            // ModelMesh.sendDestinationId
    InstanceRecord freshSelf();             // getFreshInstanceRecord(), ModelMesh.java:5369
    String currentModelId();                // the model id of the request in flight on this thread
    ModelMesh.CacheMissExcludeSet cacheMissExcludes();            // cacheMissExcludeTl.get(), ModelMesh.java:4755
    ModelMesh.MapFilteringSet<String, Long> cacheHitExcludes();   // cacheHitExcludeTl.get(), ModelMesh.java:4307
    Collection<String> cacheHitKeyExcludes();                     // its private keyExcludes field (:4269), may be null
    int localInvokesInFlight();             // ModelMesh.java:4303
    long lastInvokeTime();                  // ModelMesh.java:4304
    long assumeCompletedAfterMillis(String modelType);            // loadingTimeStats(type), ModelMesh.java:4350
}

/**
 * Replaces the BODY of CacheMissForwardingLB.getNext (ModelMesh.java:4776-5005): one mmp_place_batch(n = 1) per
 * request; everything around it (thread-locals, ThreadContext side effects, the returned ServiceInstanceInfo) is
 * as in the reference.
 */
class GpuCacheMissLB extends ModelMesh.IdBasedLoadBalancer {
    static final int MAX_LIVE_RETRIES = 8;
    final GpuMeshBinding mesh;
    /**
     * The result row (chosen, best, n_candidates, hash) of the last decision OF THE CALLING THREAD, for the shadow harness.
     * litelinks shares one LB instance across all request threads (which is why the reference keeps its request state in
     * ThreadLocals, ModelMesh.java:4755), so nothing a decision produces may live in a field of the LB.
     */
    private static final ThreadLocal<int[]> LAST_OUT = ThreadLocal.withInitial(() -> new int[4]);
    int[] lastOut() { return LAST_OUT.get(); }

    GpuCacheMissLB(GpuMeshBinding mesh) { this.mesh = mesh; }

    private static final ThreadLocal<ByteBuffer> REQ = ThreadLocal.withInitial(() -> MmPlace.direct(64));
    private static final ThreadLocal<ByteBuffer> OUT = ThreadLocal.withInitial(() -> MmPlace.direct(16));
    private static final ThreadLocal<ByteBuffer[]> EXTRA = ThreadLocal.withInitial(() -> new ByteBuffer[] { MmPlace.direct(4 * 64) });

    /** the per-thread exclusion pool, grown (never truncated) to hold n pod indices */
    static ByteBuffer extraPool(int n) {
        ByteBuffer[] box = EXTRA.get();
        if (box[0].capacity() < 4 * n) box[0] = MmPlace.direct(4 * Math.max(n, 2 * (box[0].capacity() / 4)));
        box[0].clear();
        return box[0];
    }

    /** One decision for the request state on this thread; pick replaces ThreadLocalRandom.nextInt(remaining) (:4981). */
    int decide(Map<String, ServiceInstanceInfo> siMap, ModelMesh.CacheMissExcludeSet exclude, Set<String> notLive, int pick,
               long nowMs) {
        final InstanceRecord fresh = mesh.freshSelf();
        // the HashSet itself ∪ explicit (:4740-4743); loaded / failed come from the library's registry view.
        // notLive: instances the snapshot holds as live but siMap no longer contains (!siMap.containsKey(iid), :4766)
        final int bound = exclude.size() + (exclude.explicit != null ? exclude.explicit.size() : 0) + notLive.size();
        ByteBuffer x = extraPool(Math.max(bound, 1));
        int nExtra = 0;
        for (String iid : exclude) { int p = mesh.podIndexOf(iid); if (p >= 0) { x.putInt(p); nExtra++; } }
        if (exclude.explicit != null)
            for (String iid : exclude.explicit) { int p = mesh.podIndexOf(iid); if (p >= 0) { x.putInt(p); nExtra++; } }
        for (String iid : notLive) { int p = mesh.podIndexOf(iid); if (p >= 0) { x.putInt(p); nExtra++; } }

        ByteBuffer q = REQ.get(); q.clear();               // mmp_place_req, 64 bytes (include/mmplace.h)
        q.putInt(mesh.modelIndexOf(mesh.currentModelId())); // model
        q.putInt(mesh.podIndexOf(mesh.selfInstanceId()));   // self_pod
        q.putInt(exclude.favourSelf ? 1 : 0);               // flags (MMP_REQ_FAVOUR_SELF)
        q.putInt(pick);                                     // pick
        q.putLong(exclude.lastUsedTime);                    // last_used (:4951)
        q.putInt(0); q.putInt(nExtra);                      // extra_off, n_extra
        q.putLong(fresh.getLruTime()); q.putLong(fresh.getCapacity()); q.putLong(fresh.getUsed());
        q.putInt(fresh.getCount()); q.putInt(fresh.getReqsPerMinute()); // 0, InstanceRecord.java:97-109

        ByteBuffer o = OUT.get();
        MmPlace.placeBatch(mesh.handle(), q, 1, x, nExtra, nowMs, o); // throws IllegalStateException on error
        final int[] last = LAST_OUT.get();
        for (int i = 0; i < 4; i++) last[i] = o.getInt(4 * i);        // chosen, best, n_candidates, hash
        return o.getInt(0);  // read from this thread's own OUT buffer, never from shared state
    }

    @SuppressWarnings("unchecked")
    @Override
    public <T> T getNext(Object[] sis, String method, Object[] args) {
        final ModelMesh.CacheMissExcludeSet exclude = mesh.cacheMissExcludes();
        final Map<String, ServiceInstanceInfo> siMap = getMap(sis);
        final long now = System.currentTimeMillis();
        // The reference filters on siMap membership per call (:4765-4766); the snapshot freezes liveness at commit
        // (MMP_POD_LIVE).  An instance chosen from the snapshot that siMap does not contain is excluded and the
        // decision repeated — the same walk the reference's filter would have taken.
        Set<String> notLive = java.util.Collections.emptySet();
        String chosenInstId = null;
        for (int attempt = 0;; attempt++) {
            final int chosen = decide(siMap, exclude, notLive, ThreadLocalRandom.current().nextInt(), now);
            if (chosen == MmPlace.NONE) return null;                           // :4796, :4803, :4872, :4942
            if (chosen == MmPlace.SELF) return (T) LoadBalancer.ABORT_REQUEST;  // :4894, :4932, :4990
            chosenInstId = mesh.instanceIdOf(chosen);
            if (siMap.containsKey(chosenInstId)) break;
            if (attempt == MAX_LIVE_RETRIES) return null;  // the table is far behind litelinks: let the caller retry
            if (notLive.isEmpty()) notLive = new HashSet<>();
            notLive.add(chosenInstId);
        }
        // side effects exactly as at ModelMesh.java:4992-5003
        final boolean exclusions = !exclude.isEmpty();
        if (exclusions || mesh.sendDestinationId()) {
            Map<String, String> contextMap = ModelMesh.ensureContextMapIsMutable(ThreadContext.getCurrentContext());
            if (exclusions) contextMap.put(ModelMesh.CACHE_MISS_EXCLUDES_KEY, ModelMesh.COMMA_JOIN.join(exclude));
            if (mesh.sendDestinationId()) contextMap.put(ModelMesh.DEST_INST_ID_KEY, chosenInstId);
        }
        exclude.add(chosenInstId);
        return (T) siMap.get(chosenInstId);
    }
}

/**
 * Replaces the BODY of ForwardingLB.getNext (ModelMesh.java:4315-4392), the cache-hit routing: one
 * mmp_serve_batch(n = 1) per request.  The copies of the model come from the library's registry view; the litelinks
 * counters (getInUseCount / getLastUsedTime, :4356, :4360) of THOSE copies ride in the request — O(copies) per call.
 */
class GpuForwardingLB extends ModelMesh.IdBasedLoadBalancer {
    final GpuMeshBinding mesh;
    /** chosen, chosen_load_start of the calling thread's last decision (one LB instance serves every request thread) */
    private static final ThreadLocal<long[]> LAST_OUT = ThreadLocal.withInitial(() -> new long[2]);
    long[] lastOut() { return LAST_OUT.get(); }

    GpuForwardingLB(GpuMeshBinding mesh) { this.mesh = mesh; }

    private static final ThreadLocal<ByteBuffer> REQ = ThreadLocal.withInitial(() -> MmPlace.direct(48));
    private static final ThreadLocal<ByteBuffer> OUT = ThreadLocal.withInitial(() -> MmPlace.direct(16));
    private static final ThreadLocal<ByteBuffer[]> POOLS = ThreadLocal.withInitial(() -> new ByteBuffer[4]);

    private static ByteBuffer pool(int slot, int bytes) {
        ByteBuffer[] p = POOLS.get();
        if (p[slot] == null || p[slot].capacity() < bytes) p[slot] = MmPlace.direct(Math.max(bytes, 256));
        p[slot].clear();
        return p[slot];
    }

    int decide(Map<String, ServiceInstanceInfo> siMap, ModelMesh.MapFilteringSet<String, Long> filtered, long nowMs) {
        // The reference touches litelinks' counters only for the model's copies (si.getInUseCount() / getLastUsedTime(),
        // :4356, :4360, inside the loop over filteredInstances): one mmp_serve_counter per copy that siMap lists.  A copy
        // without an entry is one litelinks does not list (sii == null -> continue, :4343-4347).  O(copies), nothing P-sized.
        final Map<String, Long> copies = filtered.map();
        ByteBuffer cnt = pool(0, 16 * Math.max(copies.size(), 1));
        int nCnt = 0;
        for (String iid : copies.keySet()) {
            final ServiceInstanceInfo sii = siMap.get(iid);
            if (sii == null) continue;
            final int p = mesh.podIndexOf(iid);
            if (p < 0) continue;
            final ServiceInstance<?> si = (ServiceInstance<?>) sii;
            cnt.putInt(16 * nCnt, p);
            cnt.putInt(16 * nCnt + 4, si.getInUseCount());
            cnt.putLong(16 * nCnt + 8, si.getLastUsedTime());
            nCnt++;
        }
        // already-tried (instance, loadStart) pairs — the MapFilteringSet's own keys — plus keyExcludes (:4278-4280);
        // an exclusion with load start Long.MIN_VALUE excludes the instance whatever its time stamp
        final Collection<String> keyExcludes = mesh.cacheHitKeyExcludes();
        final int bound = filtered.size() + (keyExcludes != null ? keyExcludes.size() : 0) + 1;
        ByteBuffer ep = pool(2, 4 * bound), et = pool(3, 8 * bound);
        int nExcl = 0;
        if (keyExcludes != null) for (String iid : keyExcludes) {
            int p = mesh.podIndexOf(iid);
            if (p >= 0) { ep.putInt(4 * nExcl, p); et.putLong(8 * nExcl, Long.MIN_VALUE); nExcl++; }
        }
        for (Map.Entry<String, Long> tried : filtered.keySet()) {
            int p = mesh.podIndexOf(tried.getKey());
            if (p >= 0) { ep.putInt(4 * nExcl, p); et.putLong(8 * nExcl, tried.getValue()); nExcl++; }
        }
        ByteBuffer q = REQ.get(); q.clear();            // mmp_serve_req, 48 bytes
        q.putInt(mesh.modelIndexOf(mesh.currentModelId()));
        q.putInt(mesh.podIndexOf(mesh.selfInstanceId()));
        q.putInt((filtered.excludeSelf ? 1 : 0) | (filtered.preferSelf ? 2 : 0)); // MMP_SERVE_EXCLUDE_SELF | _PREFER_SELF
        q.putInt(mesh.localInvokesInFlight());
        q.putLong(mesh.lastInvokeTime());
        q.putLong(mesh.assumeCompletedAfterMillis(filtered.modelType));
        q.putInt(0); q.putInt(nExcl);                   // excl_off, n_excl
        q.putInt(0); q.putInt(nCnt);                    // cnt_off, n_cnt
        ByteBuffer o = OUT.get();
        MmPlace.serveBatch(mesh.handle(), q, 1, cnt, nCnt, ep, et, nExcl, nowMs, o);
        final long[] last = LAST_OUT.get();
        last[0] = o.getInt(0);
        last[1] = o.getLong(8);
        return o.getInt(0);
    }

    @SuppressWarnings("unchecked")
    @Override
    public <T> T getNext(Object[] sis, String method, Object[] args) {
        final ModelMesh.MapFilteringSet<String, Long> filtered = mesh.cacheHitExcludes();
        final Map<String, Long> filteredInstances = filtered.map();
        if (filteredInstances == null || filteredInstances.isEmpty()) return null;   // :4318-4321
        final Map<String, ServiceInstanceInfo> siMap = getMap(sis);
        final int chosen = decide(siMap, filtered, System.currentTimeMillis());
        if (chosen == MmPlace.NONE) return null;
        if (chosen == MmPlace.SELF) return (T) LoadBalancer.ABORT_REQUEST;            // :4381-4385
        final String chosenId = mesh.instanceIdOf(chosen);
        if (mesh.sendDestinationId()) {                                               // :4386-4388
            ModelMesh.ensureContextMapIsMutable(ThreadContext.getCurrentContext()).put(ModelMesh.DEST_INST_ID_KEY, chosenId);
        }
        filtered.add(chosenId, LAST_OUT.get()[1]);                                    // :4389 (this thread's own decision)
        return (T) siMap.get(chosenId);
    }
}

/** Counters of a shadow run; exported by the patch through the mesh's metrics. */
final class ShadowStats {
    final AtomicLong calls = new AtomicLong(), compared = new AtomicLong(), agree = new AtomicLong(),
            disagree = new AtomicLong(), gpuErrors = new AtomicLong();
    volatile String lastDisagreement;

    @Override
    public String toString() {
        return "shadow[calls=" + calls + " compared=" + compared + " agree=" + agree + " disagree=" + disagree
                + " gpuErrors=" + gpuErrors + (lastDisagreement != null ? " last=" + lastDisagreement : "") + "]";
    }
}

/**
 * Shadow mode for the load-target decision: the reference LB answers the request; on a sample of the calls the GPU
 * solver decides the SAME request state and the two are compared.  The reference draws its candidate with
 * ThreadLocalRandom.nextInt(remaining) (:4981), which cannot be replayed — so the solver is asked for EVERY index
 * (4 x n_candidates evenly spaced picks cover every index of any remaining <= n_candidates) and the reference's
 * choice must be one of the solver's possible choices; null / ABORT_REQUEST must match exactly, and a shortlist of
 * one must name the same instance.  This pins PLACEMENT_ORDER, the filter, the breaks and the rpm filter — the rows
 * SURVEY.md §8(c) lists as parity-unpinned — on live traffic, one JVM-equipped box, one system property.
 */
class ShadowCacheMissLB extends ModelMesh.IdBasedLoadBalancer {
    static final ShadowStats STATS = new ShadowStats();
    static final int SAMPLE_ONE_IN = Integer.getInteger("mmesh.gpu.shadow.sample", 64);
    final LoadBalancer reference;
    final GpuCacheMissLB gpu;
    final GpuMeshBinding mesh;

    ShadowCacheMissLB(LoadBalancer reference, GpuCacheMissLB gpu, GpuMeshBinding mesh) {
        this.reference = reference; this.gpu = gpu; this.mesh = mesh;
    }

    @Override
    public <T> T getNext(Object[] sis, String method, Object[] args) {
        STATS.calls.incrementAndGet();
        final boolean sample = ThreadLocalRandom.current().nextInt(SAMPLE_ONE_IN) == 0;
        final ModelMesh.CacheMissExcludeSet exclude = mesh.cacheMissExcludes();
        Set<Integer> possible = null;
        boolean sawNone = false, sawSelf = false;
        if (sample) {
            try {  // BEFORE the reference call: it adds its choice to the exclude set (:5002)
                final Map<String, ServiceInstanceInfo> siMap = getMap(sis);
                final long now = System.currentTimeMillis();
                final Set<String> none = java.util.Collections.emptySet();
                int first = gpu.decide(siMap, exclude, none, 0, now);
                final int nCand = Math.max(gpu.lastOut()[2], 1);
                possible = new HashSet<>();
                for (int j = 0, picks = 4 * nCand; j < picks; j++) {
                    int c = j == 0 ? first : gpu.decide(siMap, exclude, none, (int) ((((long) j) << 32) / picks), now);
                    if (c == MmPlace.NONE) sawNone = true; else if (c == MmPlace.SELF) sawSelf = true; else possible.add(c);
                }
            } catch (RuntimeException e) {
                STATS.gpuErrors.incrementAndGet();
                possible = null;
            }
        }
        final T ref = reference.getNext(sis, method, args);
        if (possible != null) {
            STATS.compared.incrementAndGet();
            final boolean ok;
            if (ref == null) ok = sawNone && possible.isEmpty() && !sawSelf;
            else if (ref == LoadBalancer.ABORT_REQUEST) ok = sawSelf;
            else ok = possible.contains(mesh.podIndexOf(((ServiceInstanceInfo) ref).getInstanceId()));
            if (ok) STATS.agree.incrementAndGet();
            else {
                STATS.disagree.incrementAndGet();
                STATS.lastDisagreement = "model=" + mesh.currentModelId() + " ref=" + ref + " gpu=" + possible
                        + (sawNone ? "+null" : "") + (sawSelf ? "+self" : "") + " excl=" + exclude;
            }
        }
        return ref;
    }
}

/** Shadow mode for the serve-target decision (deterministic: the two answers must be the same instance). */
class ShadowForwardingLB extends ModelMesh.IdBasedLoadBalancer {
    static final ShadowStats STATS = new ShadowStats();
    final LoadBalancer reference;
    final GpuForwardingLB gpu;
    final GpuMeshBinding mesh;

    ShadowForwardingLB(LoadBalancer reference, GpuForwardingLB gpu, GpuMeshBinding mesh) {
        this.reference = reference; this.gpu = gpu; this.mesh = mesh;
    }

    @Override
    public <T> T getNext(Object[] sis, String method, Object[] args) {
        STATS.calls.incrementAndGet();
        final ModelMesh.MapFilteringSet<String, Long> filtered = mesh.cacheHitExcludes();
        int chosen = Integer.MIN_VALUE;
        if (ThreadLocalRandom.current().nextInt(ShadowCacheMissLB.SAMPLE_ONE_IN) == 0
                && filtered.map() != null && !filtered.map().isEmpty()) {
            try {  // before the reference adds its choice to the filtering set (:4389)
                chosen = gpu.decide(getMap(sis), filtered, System.currentTimeMillis());
            } catch (RuntimeException e) {
                STATS.gpuErrors.incrementAndGet();
            }
        }
        final T ref = reference.getNext(sis, method, args);
        if (chosen != Integer.MIN_VALUE) {
            STATS.compared.incrementAndGet();
            final int refIdx = ref == null ? MmPlace.NONE : ref == LoadBalancer.ABORT_REQUEST ? MmPlace.SELF
                    : mesh.podIndexOf(((ServiceInstanceInfo) ref).getInstanceId());
            if (refIdx == chosen) STATS.agree.incrementAndGet();
            else {
                STATS.disagree.incrementAndGet();
                STATS.lastDisagreement = "model=" + mesh.currentModelId() + " ref=" + refIdx + " gpu=" + chosen;
            }
        }
        return ref;
    }
}
