/*
 * gkl_pairhmm_jni.h -- the JNI symbols of libgkl_pairhmm.so, i.e. exactly what GKL's
 * com.intel.gkl.pairhmm.IntelPairHmm binds (reference
 * src/main/java/com/intel/gkl/pairhmm/IntelPairHmm.java:157-166; native prototypes
 * src/main/native/pairhmm/IntelPairHmm.h:38-55; bodies IntelPairHmm.cc:55-57,125-127,189-190).
 * IntelPairHmmOMP binds the same names (it only changes the library name,
 * IntelPairHmmOMP.java:29-35), so the same file also serves as libgkl_pairhmm_omp.so.
 *
 * Each symbol is a thin shim (gkl_amd/csrc/jni_shim.cpp) over the C ABI of
 * include/gkl_hip_pairhmm.h.  Define GKL_USE_SYSTEM_JNI to compile against a JDK's
 * <jni.h>; otherwise the clean-room subset in gkl_amd/csrc/jni_min.h is used (this
 * image has no JDK).
 */
#ifndef GKL_PAIRHMM_JNI_H
#define GKL_PAIRHMM_JNI_H

#ifdef GKL_USE_SYSTEM_JNI
#include <jni.h>
#else
#include "../gkl_amd/csrc/jni_min.h"
#endif

#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* static native void initNative(Class<?> readDataHolderClass, Class<?> haplotypeDataHolderClass,
 *                               boolean doublePrecision, int maxThreads)
 * Replaces IntelPairHmm.cc:55-118. Caches the six byte[] field IDs (JavaData.h:55-62; failure ->
 * IllegalArgumentException "Unable to get field ID"), then gklhip_init(). No usable gfx950
 * device -> java/lang/RuntimeException (there is no CPU path to fall back to). */
JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative(
    JNIEnv* env, jclass cls, jclass readDataHolder, jclass haplotypeDataHolder,
    jboolean use_double, jint max_threads);

/* native void computeLikelihoodsNative(Object[] readDataArray, Object[] haplotypeDataArray,
 *                                      double[] likelihoodArray)
 * Replaces IntelPairHmm.cc:125-181 + JavaData::getData (JavaData.h:65-111): copies the byte[]
 * fields into a flat batch, gklhip_compute(), writes likelihoodArray[r*numHaps + h].  A big call is pipelined:
 * read ranges are marshalled while the ranges before them compute on the slot's two engines. */
JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative(
    JNIEnv* env, jobject obj, jobjectArray readDataArray, jobjectArray haplotypeDataArray,
    jdoubleArray likelihoodArray);

/* native void doneNative()  -- replaces IntelPairHmm.cc:189-192; releases the device context. */
JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative(JNIEnv* env, jobject obj);

/* Device probe run by the JVM at System.load (not in the reference): JNI_VERSION_1_8 when a gfx950 device is usable,
 * JNI_ERR otherwise -- load() then returns false and the caller falls back, as it does on a CPU without AVX.
 * GKL_HIP_LOAD_WITHOUT_DEVICE=1 disables the probe.  (gkl_amd/csrc/jni_onload.h) */
#ifndef GKL_USE_SYSTEM_JNI
struct JavaVM_;
#endif
JNIEXPORT jint JNICALL JNI_OnLoad(struct JavaVM_* vm, void* reserved);

/* Diagnostics (not a JNI native, nothing in the reference): where the time of computeLikelihoodsNative goes, summed
 * over the calls of the process since the last reset, in nanoseconds -- [0] marshalling the holders on the calling
 * thread, [1] waiting for compute that marshalling did not cover (a small call: all of gklhip_compute), [2] writing the
 * likelihoods back, [3] whole calls, [4] number of calls, [5] calls that were pipelined (read ranges marshalled while
 * earlier ranges compute; GKL_HIP_JNI_PIPELINE_PAIRS, default 160000 pairs, sets the size from which that happens). */
void gkl_pairhmm_jni_timing(int64_t out[6], int reset);
/* ... and: [0] ns of marshalling done on the slots' helper threads (maxNumberOfThreads > 1: threads attached through the
 * JavaVM marshal read ranges beside the calling thread), [1] read ranges marshalled by helpers, [2] by calling threads,
 * [3] calls retried once on fresh contexts after a HIP failure, [4] streams / engines given back by idle slots
 * (GKL_HIP_IDLE_RELEASE_MS, default 1000: a slot unused for that long keeps one engine with its own stream). */
void gkl_pairhmm_jni_helpers(int64_t out[5], int reset);

#ifdef __cplusplus
}
#endif
#endif /* GKL_PAIRHMM_JNI_H */
