/* jni.h — a MOCK JVM for tests only (no JDK exists in this image or on the GPU box: profiles/r3/jdk_probe_*.txt).
 * Unlike tests/jni_stub/jni.h (types only: "does it compile and export"), the five JNIEnv calls
 * integration/mmplace_jni.cc makes BEHAVE here as the JNI specification says, on plain structs a test can build
 * through ctypes: a direct ByteBuffer is (address, capacity), FindClass returns a class object that remembers its
 * name, ThrowNew records the pending exception in the environment.  tests/test_jni_exec*.py load the veneer built
 * against this header and call its Java_..._MmPlace_* functions the way the JVM would — the veneer's own code (buffer
 * checks, error mapping, argument order) then really runs, on the GPU for the decision calls.
 * It is NOT a JNI implementation and is never shipped; a real build uses $JAVA_HOME/include/jni.h. */
#ifndef MMP_TEST_JNI_MOCK_H
#define MMP_TEST_JNI_MOCK_H
#include <stdint.h>
#include <string.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef int32_t jsize;
typedef signed char jbyte;

struct _jobject {       /* ctypes: class JObject(Structure) in tests/jni_mock.py */
    void *addr;         /* direct ByteBuffer: its address (NULL for a non-direct buffer, as the spec says) */
    int64_t cap;        /* ... and capacity in bytes (-1 for a non-direct buffer) */
    const char *name;   /* a class object: its name */
};
typedef _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jbyteArray;

struct JNIEnv {         /* ctypes: class JEnv(Structure) */
    int32_t throws;             /* ThrowNew calls so far */
    int32_t finds;              /* FindClass calls so far */
    char pending_class[128];    /* the pending exception (empty: none) */
    char pending_msg[512];
    _jobject cls;               /* the class object FindClass hands out (one at a time is all the veneer needs) */
    char cls_name[128];

    jclass FindClass(const char *n)
    {
        finds++;
        strncpy(cls_name, n ? n : "", sizeof cls_name - 1);
        cls_name[sizeof cls_name - 1] = 0;
        cls.addr = nullptr;
        cls.cap = -1;
        cls.name = cls_name;
        return &cls;
    }
    jint ThrowNew(jclass c, const char *msg)
    {
        throws++;
        strncpy(pending_class, c && c->name ? c->name : "", sizeof pending_class - 1);
        pending_class[sizeof pending_class - 1] = 0;
        strncpy(pending_msg, msg ? msg : "", sizeof pending_msg - 1);
        pending_msg[sizeof pending_msg - 1] = 0;
        return 0;
    }
    void *GetDirectBufferAddress(jobject b) { return b ? b->addr : nullptr; }
    jlong GetDirectBufferCapacity(jobject b) { return b ? b->cap : -1; }
    jstring NewStringUTF(const char *) { return nullptr; }
};
#endif
