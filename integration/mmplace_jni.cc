// mmplace_jni.cc — thin JNI veneer over include/mmplace.h for the Java mesh.
//
// NOT built in this repository's image (there is no JDK / jni.h here); it is the
// file a ModelMesh maintainer compiles next to libmmplace.so:
//   g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include \
//       mmplace_jni.cc -L../modelmesh_amd/lib -lmmplace -o libmmplace_jni.so
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
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_commit(JNIEnv *env, jclass, jlong h)
{
    return check(env, ctx_of(h), mmp_snapshot_commit(ctx_of(h)));
}

// CacheMissForwardingLB.getNext (MM.java:4776): n requests in, n 16-byte results out.
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_placeBatch(JNIEnv *env, jclass, jlong h, jobject reqs,
                                                                        jint n, jobject extraPool, jint nExtra,
                                                                        jlong nowMs, jobject outs)
{
    return check(env, ctx_of(h),
                 mmp_place_batch(ctx_of(h), buf<mmp_place_req>(env, reqs), n, buf<int32_t>(env, extraPool), nExtra,
                                 nowMs, buf<mmp_place_out>(env, outs)));
}

// ForwardingLB.getNext (MM.java:4315)
JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_serveBatch(JNIEnv *env, jclass, jlong h, jobject reqs,
                                                                        jint n, jobject inUse, jobject lastUsed,
                                                                        jobject exclPod, jobject exclTime, jint nExcl,
                                                                        jlong nowMs, jobject outs)
{
    return check(env, ctx_of(h),
                 mmp_serve_batch(ctx_of(h), buf<mmp_serve_req>(env, reqs), n, buf<int32_t>(env, inUse),
                                 buf<int64_t>(env, lastUsed), buf<int32_t>(env, exclPod), buf<int64_t>(env, exclTime),
                                 nExcl, nowMs, buf<mmp_serve_out>(env, outs)));
}

JNIEXPORT jint JNICALL Java_com_ibm_watson_modelmesh_MmPlace_clusterStats(JNIEnv *env, jclass, jlong h, jobject out)
{
    return check(env, ctx_of(h), mmp_cluster_stats(ctx_of(h), buf<mmp_stats>(env, out)));
}

}  // extern "C"
