/*
 * GpuPlacementLB.java — the reference-side binding a ModelMesh maintainer adds.
 * Lives in package com.ibm.watson.modelmesh next to ModelMesh.java and is installed where the two
 * litelinks clients are built (ModelMesh.java:1103-1110):
 *
 *     cacheMissClient = ThriftClientBuilder.newBuilder(iface).withServiceName(serviceName)
 *             .withLoadBalancer(useGpu ? () -> new GpuCacheMissLB() : CacheMissForwardingLB::new) ...
 *
 * It is NOT compiled in this repository (no JDK in the image); it documents the exact mapping
 * between the Java objects of the hot path and the C ABI of include/mmplace.h.
 */
package com.ibm.watson.modelmesh;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.Map;
import java.util.concurrent.ThreadLocalRandom;

import com.ibm.watson.litelinks.client.LoadBalancer;
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
    static native int serveBatch(long h, ByteBuffer reqs, int n, ByteBuffer inUse, ByteBuffer lastUsed,
                                 ByteBuffer exclPod, ByteBuffer exclTime, int nExcl, long nowMs, ByteBuffer outs);
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
    static native int proactivePlan(long h, int defaultModelSizeUnits, long nowMs, int maxOut, ByteBuffer outModel,
                                    ByteBuffer outLastUsed, ByteBuffer info);
    static native int proactivePlanSubset(long h, int partition, ByteBuffer skipModels, int nSkip, int defaultModelSizeUnits,
                                          long nowMs, int maxOut, ByteBuffer outModel, ByteBuffer outLastUsed, ByteBuffer info);
    static native int scaleupPlan(long h, ByteBuffer entries, int n, ByteBuffer params, ByteBuffer outs,
                                  ByteBuffer overloadedOut, ByteBuffer skipped);
    static native int scaledownPlan(long h, ByteBuffer entries, int n, ByteBuffer params, ByteBuffer removedOut);
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
    static native int profile(long h, boolean enable);
    static native double lastKernelMs(long h);
    static native int abiVersion();
    static final int NONE = -1, SELF = -2;
}

/**
 * Inner class of ModelMesh in the real patch (it needs instanceId, clusterState listeners,
 * getFreshInstanceRecord(), cacheMissExcludeTl). Replaces the BODY of
 * CacheMissForwardingLB.getNext (ModelMesh.java:4776-5005); everything around it is unchanged.
 */
abstract class GpuCacheMissLB extends ModelMesh.IdBasedLoadBalancer {
    // One snapshot handle per ModelMesh instance, refreshed by the instance-table listener
    // (handleInstanceTableChange, ModelMesh.java:1455): podsUpsert/podsRemove + commit, and by the
    // registry listener: modelsLoad once, modelsUpsert per ModelRecord event. The interner maps instance id -> dense pod index and keeps
    // id_order == rank under String.compareTo, replica_set == interned id.substring(0,6).
    abstract long handle();
    abstract int podIndexOf(String instanceId);     // -1 if unknown
    abstract String instanceIdOf(int podIndex);
    abstract int modelIndexOf(String modelId);
    abstract String selfInstanceId();
    abstract InstanceRecord freshSelf();            // getFreshInstanceRecord(), ModelMesh.java:5369
    abstract ModelMesh.CacheMissExcludeSet excludeSet(); // cacheMissExcludeTl.get()
    abstract String currentModelId();

    private static final ThreadLocal<ByteBuffer> REQ = ThreadLocal.withInitial(
            () -> ByteBuffer.allocateDirect(64).order(ByteOrder.LITTLE_ENDIAN));
    private static final ThreadLocal<ByteBuffer> OUT = ThreadLocal.withInitial(
            () -> ByteBuffer.allocateDirect(16).order(ByteOrder.LITTLE_ENDIAN));
    private static final ThreadLocal<ByteBuffer> EXTRA = ThreadLocal.withInitial(
            () -> ByteBuffer.allocateDirect(4 * 64).order(ByteOrder.LITTLE_ENDIAN));

    @SuppressWarnings("unchecked")
    @Override
    public <T> T getNext(Object[] sis, String method, Object[] args) {
        final ModelMesh.CacheMissExcludeSet exclude = excludeSet();
        final Map<String, ServiceInstanceInfo> siMap = getMap(sis);
        final InstanceRecord fresh = freshSelf();

        // mmp_place_req, 64 bytes (include/mmplace.h)
        ByteBuffer q = REQ.get(); q.clear();
        q.putInt(modelIndexOf(currentModelId()));          // model
        q.putInt(podIndexOf(selfInstanceId()));            // self_pod
        q.putInt(exclude.favourSelf ? 1 : 0);              // flags (MMP_REQ_FAVOUR_SELF)
        q.putInt(ThreadLocalRandom.current().nextInt());   // pick (replaces nextInt(remaining), :4981)
        q.putLong(exclude.lastUsedTime);                   // last_used (:4951)
        ByteBuffer x = EXTRA.get(); x.clear();
        int nExtra = 0;                                    // the HashSet itself ∪ explicit (:4740-4743);
        for (String iid : exclude) {                       // loaded/failed come from the model table
            int p = podIndexOf(iid); if (p >= 0 && nExtra < 64) { x.putInt(p); nExtra++; }
        }
        if (exclude.explicit != null) for (String iid : exclude.explicit) {
            int p = podIndexOf(iid); if (p >= 0 && nExtra < 64) { x.putInt(p); nExtra++; }
        }
        q.putInt(0); q.putInt(nExtra);                     // extra_off, n_extra
        q.putLong(fresh.getLruTime()); q.putLong(fresh.getCapacity()); q.putLong(fresh.getUsed());
        q.putInt(fresh.getCount()); q.putInt(fresh.getReqsPerMinute()); // 0, InstanceRecord.java:97-109

        ByteBuffer o = OUT.get();
        MmPlace.placeBatch(handle(), q, 1, x, nExtra, System.currentTimeMillis(), o); // throws on error
        final int chosen = o.getInt(0);
        if (chosen == MmPlace.NONE) return null;                       // :4796, :4803, :4872, :4942
        if (chosen == MmPlace.SELF) return (T) LoadBalancer.ABORT_REQUEST; // :4894, :4932, :4990
        final String chosenInstId = instanceIdOf(chosen);
        // side effects stay in Java exactly as at ModelMesh.java:4992-5003
        exclude.add(chosenInstId);
        return (T) siMap.get(chosenInstId);
    }
}
