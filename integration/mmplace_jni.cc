// mmplace_jni.cc — thin JNI veneer over include/mmplace.h for the Java mesh.
//
// This repository's image has no JDK / jni.h, so the shipped build does not include it; it is the
// file a ModelMesh maintainer compiles next to libmmplace.so (tests/test_jni_veneer.py compiles and
// links it against a stub jni.h and checks the exported names against the Java `native` declarations):
//   g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include mmplace_jni.cc
//       -L../modelmesh_amd/lib -lmmplace -o libmmplace_jni.so
//
// Java side: integration/GpuPlacementLB.java (class com.ibm.watson.modelmesh.MmPlace).
// All buffers are direct ByteBuffers laid out exactly as the C structs (little
// endian), so nothing is copied or translated here.
#include <jni.h>

#include "mmplace.h"

namespace {
inline mmp_ctx *ctx_of(jlong h) { return reinterpret_cast<mmp_ctx *>(static_cast<intptr_t>(h)); }

// Never let a solver error take the JVM down: surface it as IllegalStateException.
jint check(JNIEnv *env, mmp_ctx *c, int rc)
{
    if (rc != MMP_OK) {
        jclass ex = env->FindClass("java/lang/IllegalStateException");
        if (ex) env->ThrowNew(ex, mmp_last_error(c));
    }
    return rc;
}
template <class T>
T *buf(JNIEnv *env, jobject bb) { return bb ? static_cast<T *>(env->GetDirectBufferAddress(bb)) : nullptr; }

// A direct buffer that must hold `count` elements of T: the C ABI trusts its sizes, the JVM must not (a short buffer
// would be read or written past its end by the library).  Throws IllegalArgumentException and returns false.
template <class T>
bool holds(JNIEnv *env, jobject bb, jlong count, const char *what)
{
    if (count <= 0) return true;
    const jlong cap = bb ? env->GetDirectBufferCapacity(bb) : -1;
    if (cap >= 0 && cap >= count * static_cast<jlong>(sizeof(T))) return true;
    jclass ex = env->FindClass("java/lang/IllegalArgumentException");
    if (ex) env->ThrowNew(ex, what);
    return false;
}
}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_com_ibm_watson_modelmesh_MmPlace_create(JNIEnv *env, jclass, jint device,
                                                                     jlong minSpaceUnits, jlong minChurnAgeMs)
{
    mmp_config cfg{};
    cfg.device = device;
    cfg.min_space_units = minSpaceUnits;  // MM.java:765-771
    cfg.min_churn_age_ms = minChurnAgeMs; // MM.java:697
    mmp_ctx *c = nullptr;
    if (check(env, nullptr, mmp_create(&cfg, &c)) != MMP_OK) return 0;
    return static_cast<jlong>(reinterpret_cast<intptr_t>(c));
}

JNIEXPORT void JNICALL Java_com_ibm_watson_modelmesh_MmPlace_destroy(JNIEnv *, jclass, jlong h) { mmp_destroy(ctx_of(h)); }

// handleInstanceTableChange (MM.java:1455): whole-table load or single-row upserts, then commit.
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_podsLoad(JNIEnv *env, jclass, jlong h, jobject rows, jint n)
{
    return check(env, ctx_of(h), mmp_pods_load(ctx_of(h), buf<mmp_pod_row>(env, rows), n));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_podsUpsert(JNIEnv *env, jclass, jlong h, jobject idx,
                                                                        jobject rows, jint n)
{
    return check(env, ctx_of(h), mmp_pods_upsert(ctx_of(h), buf<int32_t>(env, idx), buf<mmp_pod_row>(env, rows), n));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_podsRemove(JNIEnv *env, jclass, jlong h, jobject idx, jint n)
{
    return check(env, ctx_of(h), mmp_pods_remove(ctx_of(h), buf<int32_t>(env, idx), n));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_typesLoad(JNIEnv *env, jclass, jlong h, jint nTypes,
                                                                       jobject allowed, jobject prefer,
                                                                       jobject hasAllowed, jobject hasPrefer)
{
    return check(env, ctx_of(h),
                 mmp_types_load(ctx_of(h), nTypes, buf<uint64_t>(env, allowed), buf<uint64_t>(env, prefer),
                                buf<uint8_t>(env, hasAllowed), buf<uint8_t>(env, hasPrefer)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_replacedReplicaSetsLoad(JNIEnv *env, jclass, jlong h,
                                                                                     jobject rs, jint n)
{
    return check(env, ctx_of(h), mmp_replaced_rs_load(ctx_of(h), buf<int32_t>(env, rs), n));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_modelsLoad(JNIEnv *env, jclass, jlong h, jobject rows,
                                                                        jint nModels, jobject entPod, jobject entTime,
                                                                        jint nEntries)
{
    return check(env, ctx_of(h),
                 mmp_models_load(ctx_of(h), buf<mmp_model_row>(env, rows), nModels, buf<int32_t>(env, entPod),
                                 buf<int64_t>(env, entTime), nEntries));
}
// registry listener events: whole ModelRecords replaced by index (see mmp_models_upsert)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_modelsUpsert(JNIEnv *env, jclass, jlong h, jobject idx,
                                                                          jobject rows, jint n, jobject entPod,
                                                                          jobject entTime, jint nEntries)
{
    return check(env, ctx_of(h),
                 mmp_models_upsert(ctx_of(h), buf<int32_t>(env, idx), buf<mmp_model_row>(env, rows), n,
                                   buf<int32_t>(env, entPod), buf<int64_t>(env, entTime), nEntries));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_commit(JNIEnv *env, jclass, jlong h)
{
    return check(env, ctx_of(h), mmp_snapshot_commit(ctx_of(h)));
}

// CacheMissForwardingLB.getNext (MM.java:4776): n requests in, n 16-byte results out.
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_placeBatch(JNIEnv *env, jclass, jlong h, jobject reqs,
                                                                        jint n, jobject extraPool, jint nExtra,
                                                                        jlong nowMs, jobject outs)
{
    if (!holds<mmp_place_req>(env, reqs, n, "placeBatch: reqs shorter than n requests") ||
        !holds<int32_t>(env, extraPool, nExtra, "placeBatch: extraPool shorter than nExtra entries") ||
        !holds<mmp_place_out>(env, outs, n, "placeBatch: outs shorter than n results"))
        return MMP_EINVAL;
    return check(env, ctx_of(h),
                 mmp_place_batch(ctx_of(h), buf<mmp_place_req>(env, reqs), n, buf<int32_t>(env, extraPool), nExtra,
                                 nowMs, buf<mmp_place_out>(env, outs)));
}
// the single-caller form: the caller's side (mmp_place_caller, 40 bytes) once per call, 24-byte requests — the batches of the rate
// task, the janitor, the reaper and preShutdown, which all run on ONE instance
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_placeBatchCaller(JNIEnv *env, jclass, jlong h, jobject caller,
                                                                              jobject reqs, jint n, jobject extraPool,
                                                                              jint nExtra, jlong nowMs, jobject outs)
{
    if (!holds<mmp_place_caller>(env, caller, 1, "placeBatchCaller: caller shorter than one mmp_place_caller") ||
        !holds<mmp_place_req_c>(env, reqs, n, "placeBatchCaller: reqs shorter than n requests") ||
        !holds<int32_t>(env, extraPool, nExtra, "placeBatchCaller: extraPool shorter than nExtra entries") ||
        !holds<mmp_place_out>(env, outs, n, "placeBatchCaller: outs shorter than n results"))
        return MMP_EINVAL;
    return check(env, ctx_of(h),
                 mmp_place_batch_c(ctx_of(h), buf<mmp_place_caller>(env, caller), buf<mmp_place_req_c>(env, reqs), n,
                                   buf<int32_t>(env, extraPool), nExtra, nowMs, buf<mmp_place_out>(env, outs)));
}

// The resident decision kernel: placeBatch(n = 1) without a kernel launch (include/mmplace.h: mmp_resident).
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_resident(JNIEnv *env, jclass, jlong h, jboolean enable)
{
    return check(env, ctx_of(h), mmp_resident(ctx_of(h), enable ? 1 : 0));
}

// ---- the pod-axis group (several GPUs of one node; RCCL runs inside libmmplace) ----
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_shardUniqueId(JNIEnv *env, jclass, jobject idOut)
{
    if (!holds<char>(env, idOut, MMP_SHARD_UNIQUE_ID_BYTES, "shardUniqueId: idOut shorter than 128 bytes")) return MMP_EINVAL;
    return check(env, nullptr, mmp_shard_unique_id(buf<char>(env, idOut)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_shardGroupInit(JNIEnv *env, jclass, jlong h, jobject id,
                                                                            jint rank, jint world)
{
    if (id && !holds<char>(env, id, MMP_SHARD_UNIQUE_ID_BYTES, "shardGroupInit: id shorter than 128 bytes")) return MMP_EINVAL;
    return check(env, ctx_of(h), mmp_shard_group_init(ctx_of(h), buf<char>(env, id), rank, world));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_shardGroupDestroy(JNIEnv *env, jclass, jlong h)
{
    return check(env, ctx_of(h), mmp_shard_group_destroy(ctx_of(h)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_shardCommit(JNIEnv *env, jclass, jlong h)
{
    return check(env, ctx_of(h), mmp_shard_commit(ctx_of(h)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_shardPlaceBatch(JNIEnv *env, jclass, jlong h, jobject reqs,
                                                                             jint n, jobject extraPool, jint nExtra,
                                                                             jlong nowMs, jobject outs, jobject nRestOut)
{
    if (!holds<mmp_place_req>(env, reqs, n, "shardPlaceBatch: reqs shorter than n requests") ||
        !holds<int32_t>(env, extraPool, nExtra, "shardPlaceBatch: extraPool shorter than nExtra entries") ||
        !holds<mmp_place_out>(env, outs, n, "shardPlaceBatch: outs shorter than n results") ||
        (nRestOut && !holds<int32_t>(env, nRestOut, 1, "shardPlaceBatch: nRestOut shorter than one int")))
        return MMP_EINVAL;
    return check(env, ctx_of(h),
                 mmp_shard_place_batch(ctx_of(h), buf<mmp_place_req>(env, reqs), n, buf<int32_t>(env, extraPool), nExtra,
                                       nowMs, buf<mmp_place_out>(env, outs), buf<int32_t>(env, nRestOut)));
}

// ForwardingLB.getNext (MM.java:4315): the request brings the counters of its own copies (O(copies), no table-sized buffer)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_serveBatch(JNIEnv *env, jclass, jlong h, jobject reqs,
                                                                        jint n, jobject counters, jint nCounters,
                                                                        jobject exclPod, jobject exclTime, jint nExcl,
                                                                        jlong nowMs, jobject outs)
{
    if (!holds<mmp_serve_req>(env, reqs, n, "serveBatch: reqs shorter than n requests") ||
        !holds<mmp_serve_counter>(env, counters, nCounters, "serveBatch: counters shorter than nCounters entries") ||
        !holds<int32_t>(env, exclPod, nExcl, "serveBatch: exclPod shorter than nExcl entries") ||
        !holds<int64_t>(env, exclTime, nExcl, "serveBatch: exclTime shorter than nExcl entries") ||
        !holds<mmp_serve_out>(env, outs, n, "serveBatch: outs shorter than n results"))
        return MMP_EINVAL;
    return check(env, ctx_of(h),
                 mmp_serve_batch(ctx_of(h), buf<mmp_serve_req>(env, reqs), n, buf<mmp_serve_counter>(env, counters), nCounters,
                                 buf<int32_t>(env, exclPod), buf<int64_t>(env, exclTime), nExcl, nowMs, buf<mmp_serve_out>(env, outs)));
}

JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_clusterStats(JNIEnv *env, jclass, jlong h, jobject out)
{
    return check(env, ctx_of(h), mmp_cluster_stats(ctx_of(h), buf<mmp_stats>(env, out)));
}

// typeSetStats / instanceSetStats and the ProhibitedTypeSet partitions (MM.java:1432-1448, TypeConstraintManager)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_typeStats(JNIEnv *env, jclass, jlong h, jint type, jobject out)
{
    return check(env, ctx_of(h), mmp_type_stats(ctx_of(h), type, buf<mmp_stats>(env, out)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_partitionCount(JNIEnv *env, jclass, jlong h, jobject nOut)
{
    return check(env, ctx_of(h), mmp_partition_count(ctx_of(h), buf<int32_t>(env, nOut)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_partitionStats(JNIEnv *env, jclass, jlong h, jint partition,
                                                                            jobject out, jobject prohibitedOut, jint maxWords)
{
    return check(env, ctx_of(h),
                 mmp_partition_stats(ctx_of(h), partition, buf<mmp_stats>(env, out), buf<uint64_t>(env, prohibitedOut), maxWords));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_podPartitions(JNIEnv *env, jclass, jlong h, jobject partitionOut,
                                                                           jint maxPods, jobject nOut)
{
    return check(env, ctx_of(h), mmp_pod_partitions(ctx_of(h), buf<int32_t>(env, partitionOut), maxPods, buf<int32_t>(env, nOut)));
}

// ---- eviction: clhm put/evict (ConcurrentLinkedHashMap.java:590-611,329-352) -----------------------
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_cachesLoad(JNIEnv *env, jclass, jlong h, jint nCaches,
                                                                        jobject segOff, jobject lastUsed,
                                                                        jobject weight, jobject capacity)
{
    return check(env, ctx_of(h),
                 mmp_caches_load(ctx_of(h), nCaches, buf<int32_t>(env, segOff), buf<int64_t>(env, lastUsed),
                                 buf<int32_t>(env, weight), buf<int64_t>(env, capacity)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_evictBatch(JNIEnv *env, jclass, jlong h, jobject reqs,
                                                                        jint n, jlong nowMs, jobject outs)
{
    return check(env, ctx_of(h),
                 mmp_evict_batch(ctx_of(h), buf<mmp_evict_req>(env, reqs), n, nowMs, buf<mmp_evict_out>(env, outs)));
}

// ---- stateful caches + ModelCacheUnloadBufManager (ModelCacheUnloadBufManager.java:130-402) --------
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_cachesLoadKeyed(JNIEnv *env, jclass, jlong h,
                                                                             jint nCaches, jobject segOff,
                                                                             jobject lastUsed, jobject weight,
                                                                             jobject key, jobject capacity,
                                                                             jobject ubm)
{
    return check(env, ctx_of(h),
                 mmp_caches_load_keyed(ctx_of(h), nCaches, buf<int32_t>(env, segOff), buf<int64_t>(env, lastUsed),
                                       buf<int32_t>(env, weight), buf<int32_t>(env, key),
                                       buf<int64_t>(env, capacity), buf<mmp_ubm_state>(env, ubm)));
}
// nEvictedSlots: direct IntBuffer of one element
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_cacheReplay(JNIEnv *env, jclass, jlong h, jobject ops,
                                                                         jint nOps, jlong nowMs, jobject outs,
                                                                         jobject evictedKeys, jint maxEvicted,
                                                                         jobject nEvictedSlots)
{
    return check(env, ctx_of(h),
                 mmp_cache_replay(ctx_of(h), buf<mmp_cache_op>(env, ops), nOps, nowMs,
                                  buf<mmp_cache_op_out>(env, outs), buf<int32_t>(env, evictedKeys), maxEvicted,
                                  buf<int32_t>(env, nEvictedSlots)));
}
// scalars: direct buffer {int32 n; int32 pad; int64 capacity; int64 weightedSize}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_cacheRead(JNIEnv *env, jclass, jlong h, jint cache,
                                                                       jint maxEntries, jobject lastUsed,
                                                                       jobject weight, jobject key, jobject scalars,
                                                                       jobject ubm)
{
    struct Scalars {
        int32_t n, pad;
        int64_t capacity, weighted_size;
    } *sc = buf<Scalars>(env, scalars);
    if (!sc) return check(env, ctx_of(h), MMP_EINVAL);
    return check(env, ctx_of(h),
                 mmp_cache_read(ctx_of(h), cache, maxEntries, buf<int64_t>(env, lastUsed), buf<int32_t>(env, weight),
                                buf<int32_t>(env, key), &sc->n, &sc->capacity, &sc->weighted_size,
                                buf<mmp_ubm_state>(env, ubm)));
}

// ---- request guards (MM.java:3603-3626, 3870-3884, 4003-4042, 4590-4627, 5158-5197, 2867-2933, 5440-5468)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_gateBatch(JNIEnv *env, jclass, jlong h, jobject reqs,
                                                                       jint n, jobject exclPod, jobject exclTime,
                                                                       jint nExcl, jobject explicitPool,
                                                                       jint nExplicit, jlong nowMs,
                                                                       jlong inUseFailureExpiryMs, jobject outs)
{
    return check(env, ctx_of(h),
                 mmp_gate_batch(ctx_of(h), buf<mmp_gate_req>(env, reqs), n, buf<int32_t>(env, exclPod),
                                buf<int64_t>(env, exclTime), nExcl, buf<int32_t>(env, explicitPool), nExplicit, nowMs,
                                inUseFailureExpiryMs, buf<mmp_gate_out>(env, outs)));
}

// ---- the cache-MISS route in one launch: the guards + the load target of every request (include/mmplace.h: mmp_miss_batch)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_missBatch(JNIEnv *env, jclass, jlong h, jobject gateReqs,
                                                                       jobject placeReqs, jint n, jobject exclPod, jobject exclTime,
                                                                       jint nExcl, jobject explicitPool, jint nExplicit,
                                                                       jobject extraPool, jint nExtra, jlong nowMs,
                                                                       jlong inUseFailureExpiryMs, jobject gateOuts, jobject placeOuts)
{
    return check(env, ctx_of(h),
                 mmp_miss_batch(ctx_of(h), buf<mmp_gate_req>(env, gateReqs), buf<mmp_place_req>(env, placeReqs), n,
                                buf<int32_t>(env, exclPod), buf<int64_t>(env, exclTime), nExcl, buf<int32_t>(env, explicitPool), nExplicit,
                                buf<int32_t>(env, extraPool), nExtra, nowMs, inUseFailureExpiryMs, buf<mmp_gate_out>(env, gateOuts),
                                buf<mmp_place_out>(env, placeOuts)));
}
// ---- the cache-HIT route in one launch: the guards + the serve target of every request (include/mmplace.h: mmp_route_batch)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_routeBatch(JNIEnv *env, jclass, jlong h, jobject gateReqs,
                                                                        jobject serveReqs, jint n, jobject counters,
                                                                        jint nCounters, jobject exclPod, jobject exclTime,
                                                                        jint nExcl, jobject explicitPool, jint nExplicit,
                                                                        jlong nowMs, jlong inUseFailureExpiryMs,
                                                                        jobject gateOuts, jobject serveOuts)
{
    return check(env, ctx_of(h),
                 mmp_route_batch(ctx_of(h), buf<mmp_gate_req>(env, gateReqs), buf<mmp_serve_req>(env, serveReqs), n,
                                 buf<mmp_serve_counter>(env, counters), nCounters, buf<int32_t>(env, exclPod),
                                 buf<int64_t>(env, exclTime), nExcl, buf<int32_t>(env, explicitPool), nExplicit, nowMs,
                                 inUseFailureExpiryMs, buf<mmp_gate_out>(env, gateOuts), buf<mmp_serve_out>(env, serveOuts)));
}

// ---- rebalancers (MM.java:5636-5871, 6110-6335, 6616-6747, 6959-7147) ---------------------------------
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_proactivePlan(JNIEnv *env, jclass, jlong h,
                                                                           jint defaultModelSizeUnits, jlong nowMs,
                                                                           jint maxOut, jobject outModel,
                                                                           jobject outLastUsed, jobject info)
{
    return check(env, ctx_of(h),
                 mmp_proactive_plan(ctx_of(h), defaultModelSizeUnits, nowMs, maxOut, buf<int32_t>(env, outModel),
                                    buf<int64_t>(env, outLastUsed), buf<mmp_proactive_info>(env, info)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_proactivePlanSubset(JNIEnv *env, jclass, jlong h, jint partition,
                                                                                 jobject skipModels, jint nSkip,
                                                                                 jint defaultModelSizeUnits, jlong nowMs,
                                                                                 jint maxOut, jobject outModel,
                                                                                 jobject outLastUsed, jobject info)
{
    return check(env, ctx_of(h),
                 mmp_proactive_plan_subset(ctx_of(h), partition, buf<int32_t>(env, skipModels), nSkip, defaultModelSizeUnits,
                                           nowMs, maxOut, buf<int32_t>(env, outModel), buf<int64_t>(env, outLastUsed),
                                           buf<mmp_proactive_info>(env, info)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_scaleupPlan(JNIEnv *env, jclass, jlong h, jobject entries,
                                                                         jint n, jobject params, jobject outs,
                                                                         jobject overloadedOut, jobject skipped)
{
    return check(env, ctx_of(h),
                 mmp_scaleup_plan(ctx_of(h), buf<mmp_cache_entry>(env, entries), n,
                                  buf<mmp_scaleup_params>(env, params), buf<mmp_scaleup_out>(env, outs),
                                  buf<uint8_t>(env, overloadedOut), buf<int32_t>(env, skipped)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_scaledownPlan(JNIEnv *env, jclass, jlong h,
                                                                           jobject entries, jint n, jobject params,
                                                                           jobject removedOut)
{
    return check(env, ctx_of(h),
                 mmp_scaledown_plan(ctx_of(h), buf<mmp_cache_entry>(env, entries), n,
                                    buf<mmp_scaledown_params>(env, params), buf<uint8_t>(env, removedOut)));
}
// the rate task / the janitor of a mesh that runs with limitModelConcurrency == true (MaxConcCacheEntry rows beside the entries)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_scaleupPlanConc(JNIEnv *env, jclass, jlong h, jobject entries,
                                                                             jobject conc, jint n, jobject params,
                                                                             jobject concParams, jobject outs, jobject concOuts,
                                                                             jobject overloadedOut, jobject skipped, jobject result)
{
    if (!holds<mmp_cache_entry>(env, entries, n, "scaleupPlanConc: entries shorter than n rows") ||
        !holds<mmp_conc_entry>(env, conc, n, "scaleupPlanConc: conc shorter than n rows") ||
        !holds<mmp_scaleup_params>(env, params, 1, "scaleupPlanConc: params shorter than one mmp_scaleup_params") ||
        !holds<mmp_conc_params>(env, concParams, 1, "scaleupPlanConc: concParams shorter than one mmp_conc_params") ||
        !holds<mmp_scaleup_out>(env, outs, n, "scaleupPlanConc: outs shorter than n rows") ||
        !holds<mmp_conc_out>(env, concOuts, n, "scaleupPlanConc: concOuts shorter than n rows") ||
        !holds<uint8_t>(env, overloadedOut, 1, "scaleupPlanConc: overloadedOut shorter than one byte") ||
        !holds<int32_t>(env, skipped, 1, "scaleupPlanConc: skipped shorter than one int") ||
        !holds<mmp_conc_result>(env, result, 1, "scaleupPlanConc: result shorter than one mmp_conc_result"))
        return MMP_EINVAL;
    return check(env, ctx_of(h),
                 mmp_scaleup_plan_conc(ctx_of(h), buf<mmp_cache_entry>(env, entries), buf<mmp_conc_entry>(env, conc), n,
                                       buf<mmp_scaleup_params>(env, params), buf<mmp_conc_params>(env, concParams),
                                       buf<mmp_scaleup_out>(env, outs), buf<mmp_conc_out>(env, concOuts),
                                       buf<uint8_t>(env, overloadedOut), buf<int32_t>(env, skipped),
                                       buf<mmp_conc_result>(env, result)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_scaledownPlanConc(JNIEnv *env, jclass, jlong h, jobject entries,
                                                                               jobject conc, jint n, jobject params,
                                                                               jlong dynamicRpmScaleConstant, jobject removedOut)
{
    if (!holds<mmp_cache_entry>(env, entries, n, "scaledownPlanConc: entries shorter than n rows") ||
        !holds<mmp_conc_entry>(env, conc, n, "scaledownPlanConc: conc shorter than n rows") ||
        !holds<mmp_scaledown_params>(env, params, 1, "scaledownPlanConc: params shorter than one mmp_scaledown_params") ||
        !holds<uint8_t>(env, removedOut, n, "scaledownPlanConc: removedOut shorter than n bytes"))
        return MMP_EINVAL;
    return check(env, ctx_of(h),
                 mmp_scaledown_plan_conc(ctx_of(h), buf<mmp_cache_entry>(env, entries), buf<mmp_conc_entry>(env, conc), n,
                                         buf<mmp_scaledown_params>(env, params), dynamicRpmScaleConstant,
                                         buf<uint8_t>(env, removedOut)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_migrationPlan(JNIEnv *env, jclass, jlong h,
                                                                           jobject entries, jint n, jint selfPod,
                                                                           jlong nowMs, jlong cutoffAgeMs,
                                                                           jobject actionOut, jobject waitOut)
{
    return check(env, ctx_of(h),
                 mmp_migration_plan(ctx_of(h), buf<mmp_cache_entry>(env, entries), n, selfPod, nowMs, cutoffAgeMs,
                                    buf<uint8_t>(env, actionOut), buf<uint8_t>(env, waitOut)));
}

// ---- type constraints from labels (TypeConstraintManager.java:337-506, 680-747) ----------------------
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_typesFromLabels(JNIEnv *env, jclass, jlong h,
                                                                             jint nTypes, jobject required,
                                                                             jobject preferred, jobject podLabels,
                                                                             jobject allowedOut, jobject preferOut,
                                                                             jobject hasAllowedOut,
                                                                             jobject hasPreferOut)
{
    return check(env, ctx_of(h),
                 mmp_types_from_labels(ctx_of(h), nTypes, buf<uint64_t>(env, required), buf<uint64_t>(env, preferred),
                                       buf<uint64_t>(env, podLabels), buf<uint64_t>(env, allowedOut),
                                       buf<uint64_t>(env, preferOut), buf<uint8_t>(env, hasAllowedOut),
                                       buf<uint8_t>(env, hasPreferOut)));
}

// ---- UpgradeTracker (UpgradeTracker.java:85-200; called at MM.java:1532,1553,1563) -------------------
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_upgradeInstanceAdded(JNIEnv *env, jclass, jlong h,
                                                                                  jlong labelsKey, jint replicaSet,
                                                                                  jlong startTime, jlong nowMs)
{
    return check(env, ctx_of(h), mmp_upgrade_instance_added(ctx_of(h), labelsKey, replicaSet, startTime, nowMs));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_upgradeInstanceRemoved(JNIEnv *env, jclass, jlong h,
                                                                                    jlong labelsKey, jint replicaSet,
                                                                                    jlong nowMs)
{
    return check(env, ctx_of(h), mmp_upgrade_instance_removed(ctx_of(h), labelsKey, replicaSet, nowMs));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_upgradeHousekeeping(JNIEnv *env, jclass, jlong h,
                                                                                 jlong nowMs)
{
    return check(env, ctx_of(h), mmp_upgrade_housekeeping(ctx_of(h), nowMs));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_upgradeReplaced(JNIEnv *env, jclass, jlong h,
                                                                             jobject rsOut, jobject expiryOut,
                                                                             jint max, jobject nOut)
{
    return check(env, ctx_of(h),
                 mmp_upgrade_replaced(ctx_of(h), buf<int32_t>(env, rsOut), buf<int64_t>(env, expiryOut), max,
                                      buf<int32_t>(env, nOut)));
}

// ---- KV wire format: the raw byte[] of the KV events, packed back to back in a direct buffer ----------
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_podIdsLoad(JNIEnv *env, jclass, jlong h, jobject ids,
                                                                        jobject idOff, jint nPods,
                                                                        jobject idOrderOut, jobject replicaSetOut)
{
    return check(env, ctx_of(h),
                 mmp_pod_ids_load(ctx_of(h), buf<char>(env, ids), buf<int32_t>(env, idOff), nPods,
                                  buf<uint32_t>(env, idOrderOut), buf<int32_t>(env, replicaSetOut)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_podsIngestJson(JNIEnv *env, jclass, jlong h, jobject json,
                                                                            jobject off, jint n, jobject podIdx,
                                                                            jobject live, jobject startTimeOut,
                                                                            jobject statusOut)
{
    return check(env, ctx_of(h),
                 mmp_pods_ingest_json(ctx_of(h), buf<char>(env, json), buf<int64_t>(env, off), n,
                                      buf<int32_t>(env, podIdx), buf<uint8_t>(env, live),
                                      buf<int64_t>(env, startTimeOut), buf<int32_t>(env, statusOut)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_typeNamesLoad(JNIEnv *env, jclass, jlong h, jobject names,
                                                                           jobject nameOff, jint nTypes,
                                                                           jint unknownType)
{
    return check(env, ctx_of(h),
                 mmp_type_names_load(ctx_of(h), buf<char>(env, names), buf<int32_t>(env, nameOff), nTypes, unknownType));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_modelsIngestJson(JNIEnv *env, jclass, jlong h,
                                                                              jobject json, jobject off,
                                                                              jint nModels, jobject lastUnloadOut,
                                                                              jobject statusOut)
{
    return check(env, ctx_of(h),
                 mmp_models_ingest_json(ctx_of(h), buf<char>(env, json), buf<int64_t>(env, off), nModels,
                                        buf<int64_t>(env, lastUnloadOut), buf<int32_t>(env, statusOut)));
}

// ---- misc -----------------------------------------------------------------------------------------
JNIEXPORT jlong JNICALL Java_com_ibm_watson_modelmesh_MmPlace_minSpaceUnits(JNIEnv *, jclass,
                                                                            jint defaultModelSizeUnits,
                                                                            jint loadingThreads, jlong capacityUnits,
                                                                            jboolean haveUnloadManager)
{
    return mmp_min_space_units(defaultModelSizeUnits, loadingThreads, capacityUnits, haveUnloadManager ? 1 : 0);
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_getOrder(JNIEnv *env, jclass, jlong h, jobject orderOut,
                                                                      jobject nOut)
{
    return check(env, ctx_of(h), mmp_get_order(ctx_of(h), buf<int32_t>(env, orderOut), buf<int32_t>(env, nOut)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_deltaCommits(JNIEnv *env, jclass, jlong h, jobject nOut)
{
    return check(env, ctx_of(h), mmp_delta_commits(ctx_of(h), buf<int64_t>(env, nOut)));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_profile(JNIEnv *env, jclass, jlong h, jboolean enable)
{
    return check(env, ctx_of(h), mmp_profile(ctx_of(h), enable ? 1 : 0));
}
JNIEXPORT jdouble JNICALL Java_com_ibm_watson_modelmesh_MmPlace_lastKernelMs(JNIEnv *, jclass, jlong h)
{
    return mmp_last_kernel_ms(ctx_of(h));
}
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_abiVersion(JNIEnv *, jclass) { return mmp_abi_version(); }

}  // extern "C"
