// Clean-room subset of the Java Native Interface, for hosts without a JDK.
//
// This build container (and the GPU boxes) have no JDK, hence no <jni.h>.  The JNI
// function table is a stable binary interface fixed by the JNI specification
// ("JNI Functions", Interface Function Table): JNIEnv is a pointer to a pointer to
// an array of function pointers whose INDICES never change.  Only those indices and
// the primitive type widths are needed to call into a JVM, so this header declares
// the table as an indexed array plus typed accessors for the handful of functions
// the PairHMM shim uses.  When a real <jni.h> is available, define
// GKL_USE_SYSTEM_JNI and it is used instead (include/gkl_pairhmm_jni.h).
//
// Slots used (spec index): FindClass 6, ThrowNew 14, ExceptionClear 17,
// DeleteLocalRef 23, GetFieldID 94, GetObjectField 95, GetArrayLength 171,
// GetObjectArrayElement 173, GetByteArrayRegion 200, SetDoubleArrayRegion 214,
// ExceptionCheck 228; the PDHMM shim adds NewDoubleArray 182 and GetLongArrayRegion 204, the
// Smith-Waterman shim SetByteArrayRegion 208 and SetIntArrayRegion 211.  The PairHMM shim's block-wise marshalling
// uses PushLocalFrame 19 / PopLocalFrame 20, and its helper threads NewGlobalRef 21 / DeleteGlobalRef 22, GetJavaVM
// 219 and, of the invocation interface (JavaVM: "The Invocation API", JNIInvokeInterface), DetachCurrentThread 5,
// GetEnv 6 and AttachCurrentThreadAsDaemon 7.
#pragma once
#include <stdint.h>

extern "C" {

typedef uint8_t jboolean;
typedef int8_t jbyte;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jbyteArray;
typedef jarray jdoubleArray;
typedef jarray jlongArray;
typedef jarray jintArray;
struct _jfieldID;
typedef struct _jfieldID* jfieldID;

#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define JNI_VERSION_1_8 0x00010008

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

enum {
  kJniSlotFindClass = 6,
  kJniSlotThrowNew = 14,
  kJniSlotExceptionClear = 17,
  kJniSlotPushLocalFrame = 19,
  kJniSlotPopLocalFrame = 20,
  kJniSlotNewGlobalRef = 21,
  kJniSlotDeleteGlobalRef = 22,
  kJniSlotDeleteLocalRef = 23,
  kJniSlotGetFieldID = 94,
  kJniSlotGetObjectField = 95,
  kJniSlotGetArrayLength = 171,
  kJniSlotGetObjectArrayElement = 173,
  kJniSlotNewDoubleArray = 182,
  kJniSlotGetByteArrayRegion = 200,
  kJniSlotGetLongArrayRegion = 204,
  kJniSlotSetByteArrayRegion = 208,
  kJniSlotSetIntArrayRegion = 211,
  kJniSlotSetDoubleArrayRegion = 214,
  kJniSlotGetJavaVM = 219,
  kJniSlotExceptionCheck = 228,
  kJniSlotCount = 235  // JNI 9+: GetModule is 233, IsVirtualThread (21) is 234
};

struct JNINativeInterface_ {
  void* slot[kJniSlotCount];
};

// In C++ the real JNIEnv is `struct JNIEnv_ { const JNINativeInterface_* functions; ... }`
struct JNIEnv_ {
  const struct JNINativeInterface_* functions;
};
typedef struct JNIEnv_ JNIEnv;

// The invocation interface: JavaVM is a pointer to a pointer to a table of eight slots (three reserved, DestroyJavaVM,
// AttachCurrentThread, DetachCurrentThread, GetEnv, AttachCurrentThreadAsDaemon).
enum {
  kJvmSlotDetachCurrentThread = 5,
  kJvmSlotGetEnv = 6,
  kJvmSlotAttachCurrentThreadAsDaemon = 7,
  kJvmSlotCount = 8
};
struct JNIInvokeInterface_ {
  void* slot[kJvmSlotCount];
};
struct JavaVM_ {
  const struct JNIInvokeInterface_* functions;
};
typedef struct JavaVM_ JavaVM;
#define JNI_ERR (-1)
#define JNI_EDETACHED (-2)

}  // extern "C"

namespace gkljni {
template <typename F>
inline F fn(JNIEnv* env, int slot) { return reinterpret_cast<F>(env->functions->slot[slot]); }

inline jclass FindClass(JNIEnv* e, const char* name) {
  return fn<jclass (*)(JNIEnv*, const char*)>(e, kJniSlotFindClass)(e, name);
}
inline jint ThrowNew(JNIEnv* e, jclass c, const char* msg) {
  return fn<jint (*)(JNIEnv*, jclass, const char*)>(e, kJniSlotThrowNew)(e, c, msg);
}
inline void ExceptionClear(JNIEnv* e) { fn<void (*)(JNIEnv*)>(e, kJniSlotExceptionClear)(e); }
inline jint PushLocalFrame(JNIEnv* e, jint capacity) { return fn<jint (*)(JNIEnv*, jint)>(e, kJniSlotPushLocalFrame)(e, capacity); }
inline jobject PopLocalFrame(JNIEnv* e, jobject result) { return fn<jobject (*)(JNIEnv*, jobject)>(e, kJniSlotPopLocalFrame)(e, result); }
inline jobject NewGlobalRef(JNIEnv* e, jobject o) { return fn<jobject (*)(JNIEnv*, jobject)>(e, kJniSlotNewGlobalRef)(e, o); }
inline void DeleteGlobalRef(JNIEnv* e, jobject o) { fn<void (*)(JNIEnv*, jobject)>(e, kJniSlotDeleteGlobalRef)(e, o); }
inline jint GetJavaVM(JNIEnv* e, JavaVM** vm) { return fn<jint (*)(JNIEnv*, JavaVM**)>(e, kJniSlotGetJavaVM)(e, vm); }
inline jint AttachCurrentThreadAsDaemon(JavaVM* vm, JNIEnv** penv) {
  return reinterpret_cast<jint (*)(JavaVM*, void**, void*)>(vm->functions->slot[kJvmSlotAttachCurrentThreadAsDaemon])(vm, reinterpret_cast<void**>(penv), nullptr);
}
inline jint DetachCurrentThread(JavaVM* vm) { return reinterpret_cast<jint (*)(JavaVM*)>(vm->functions->slot[kJvmSlotDetachCurrentThread])(vm); }
inline jboolean ExceptionCheck(JNIEnv* e) { return fn<jboolean (*)(JNIEnv*)>(e, kJniSlotExceptionCheck)(e); }
inline void DeleteLocalRef(JNIEnv* e, jobject o) { fn<void (*)(JNIEnv*, jobject)>(e, kJniSlotDeleteLocalRef)(e, o); }
inline jfieldID GetFieldID(JNIEnv* e, jclass c, const char* name, const char* sig) {
  return fn<jfieldID (*)(JNIEnv*, jclass, const char*, const char*)>(e, kJniSlotGetFieldID)(e, c, name, sig);
}
inline jobject GetObjectField(JNIEnv* e, jobject o, jfieldID f) {
  return fn<jobject (*)(JNIEnv*, jobject, jfieldID)>(e, kJniSlotGetObjectField)(e, o, f);
}
inline jsize GetArrayLength(JNIEnv* e, jarray a) {
  return fn<jsize (*)(JNIEnv*, jarray)>(e, kJniSlotGetArrayLength)(e, a);
}
inline jobject GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) {
  return fn<jobject (*)(JNIEnv*, jobjectArray, jsize)>(e, kJniSlotGetObjectArrayElement)(e, a, i);
}
inline void GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize start, jsize len, jbyte* buf) {
  fn<void (*)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*)>(e, kJniSlotGetByteArrayRegion)(e, a, start, len, buf);
}
inline jdoubleArray NewDoubleArray(JNIEnv* e, jsize len) {
  return fn<jdoubleArray (*)(JNIEnv*, jsize)>(e, kJniSlotNewDoubleArray)(e, len);
}
inline void GetLongArrayRegion(JNIEnv* e, jlongArray a, jsize start, jsize len, jlong* buf) {
  fn<void (*)(JNIEnv*, jlongArray, jsize, jsize, jlong*)>(e, kJniSlotGetLongArrayRegion)(e, a, start, len, buf);
}
inline void SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize start, jsize len, const jbyte* buf) {
  fn<void (*)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*)>(e, kJniSlotSetByteArrayRegion)(e, a, start, len, buf);
}
inline void SetIntArrayRegion(JNIEnv* e, jintArray a, jsize start, jsize len, const jint* buf) {
  fn<void (*)(JNIEnv*, jintArray, jsize, jsize, const jint*)>(e, kJniSlotSetIntArrayRegion)(e, a, start, len, buf);
}
inline void SetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize start, jsize len, const jdouble* buf) {
  fn<void (*)(JNIEnv*, jdoubleArray, jsize, jsize, const jdouble*)>(e, kJniSlotSetDoubleArrayRegion)(e, a, start, len, buf);
}
}  // namespace gkljni
