/*
 * gkl_utils_jni.h -- the JNI symbols of libgkl_utils.so, the companion library GKL's
 * IntelPairHmm.load() / IntelPDHMM.load() call first (reference
 * src/main/java/com/intel/gkl/pairhmm/IntelPairHmm.java:65-75; natives
 * src/main/native/utils/utils.h:38-79, bodies utils.cc:36-115).  SURVEY.md section 8 f3: a
 * from-scratch equivalent so the drop-in is self-contained and the `isAvxSupported()` gate of
 * IntelPairHmm.load() describes THIS backend (a usable gfx950 device) instead of the host CPU.
 */
#ifndef GKL_UTILS_JNI_H
#define GKL_UTILS_JNI_H

#ifdef GKL_USE_SYSTEM_JNI
#include <jni.h>
#else
#include "../gkl_amd/csrc/jni_min.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* FTZ state of the calling thread's MXCSR (utils.cc:36-55); kept because GATK logs it. */
JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_getFlushToZeroNative(JNIEnv*, jobject);
JNIEXPORT void JNICALL Java_com_intel_gkl_IntelGKLUtils_setFlushToZeroNative(JNIEnv*, jobject, jboolean);
/* "is the accelerated PairHMM usable here": true iff a gfx950 HIP device is visible. IntelPairHmm.load()
 * returns false when this is false, which is GATK's cue to fall back to its Java PairHMM. */
JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_isAvxSupportedNative(JNIEnv*, jobject);
JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_isAvx2SupportedNative(JNIEnv*, jobject);
/* only used for an info log line ("Using CPU-supported AVX-512 instructions"): false. */
JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_isAvx512SupportedNative(JNIEnv*, jobject);
/* host threads available to the reference-exact log10 finalisation. */
JNIEXPORT jint JNICALL Java_com_intel_gkl_IntelGKLUtils_getAvailableOmpThreadsNative(JNIEnv*, jobject);

#ifdef __cplusplus
}
#endif
#endif /* GKL_UTILS_JNI_H */
