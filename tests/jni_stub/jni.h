/* jni.h — STUB for tests only.  This image has no JDK, so integration/mmplace_jni.cc could not even be
 * parsed here; this header declares just the handful of JNI names that file uses (types and the three
 * JNIEnv calls, per the public JNI specification) so that tests/test_jni_veneer.py can compile and link
 * the veneer and check its exported symbols against the `native` declarations in
 * integration/GpuPlacementLB.java.  It is NOT a JNI implementation and is never shipped; a real build
 * uses $JAVA_HOME/include/jni.h. */
#ifndef MMP_TEST_JNI_STUB_H
#define MMP_TEST_JNI_STUB_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef int32_t jsize;
typedef signed char jbyte;

class _jobject {};
typedef _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jbyteArray;

struct JNIEnv {
    jclass FindClass(const char *) { return nullptr; }
    jint ThrowNew(jclass, const char *) { return 0; }
    void *GetDirectBufferAddress(jobject) { return nullptr; }
    jlong GetDirectBufferCapacity(jobject) { return 0; }
    jstring NewStringUTF(const char *) { return nullptr; }
};
#endif
