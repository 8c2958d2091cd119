/* tests/jni_typecheck/jni.h -- NOT the JDK's header: the handful of declarations of the public JNI specification (types, JNI_ABORT, the eight
 * JNIEnv functions jni/flashfry_jni.c calls) that let gcc TYPE-CHECK the shim on a box without a JDK (tests/test_jni_binding.py, -fsyntax-only).
 * Nothing is linked or run against it; the function table below has only the members the shim uses, so its layout is NOT the real one.
 * A build against a real JDK (jni/Makefile, JAVA_HOME) never sees this file. */
#ifndef FFH_TEST_JNI_TYPECHECK_H
#define FFH_TEST_JNI_TYPECHECK_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef jint jsize;
typedef unsigned char jboolean;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jlongArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    jlongArray (*NewLongArray)(JNIEnv *, jsize);
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
    jstring (*NewStringUTF)(JNIEnv *, const char *);
};
#endif
