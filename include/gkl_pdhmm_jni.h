/*
 * gkl_pdhmm_jni.h -- the JNI symbols of libgkl_pdhmm.so: what GKL's com.intel.gkl.pdhmm.IntelPDHMM
 * binds (reference src/main/java/com/intel/gkl/pdhmm/IntelPDHMM.java:188-204; native prototypes
 * src/main/native/pdhmm/IntelPDHMM.h:32-48; bodies IntelPDHMM.cc:43-249).  Thin shims
 * (gkl_amd/csrc/jni_shim_pdhmm.cpp) over the C ABI of include/gkl_hip_pdhmm.h.
 */
#ifndef GKL_PDHMM_JNI_H
#define GKL_PDHMM_JNI_H

#ifdef GKL_USE_SYSTEM_JNI
#include <jni.h>
#else
#include "../gkl_amd/csrc/jni_min.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* static native void initNative(Class<?> readDataHolderClass, Class<?> haplotypeDataHolderClass,
 *                               int openMPSetting, int maxThreads, int avxLevel, int maxMemoryInMB)
 * openMPSetting / maxThreads / avxLevel select CPU engines in the reference (IntelPDHMM.cc:43-60) and
 * are accepted and ignored here; maxMemoryInMB bounds the staging batch of computeLikelihoodsNative. */
JNIEXPORT void JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_initNative(
    JNIEnv* env, jclass cls, jclass readDataHolder, jclass haplotypeDataHolder, jint openMPSetting,
    jint max_threads, jint avxLevel, jint maxMemoryInMB);

/* native void computeLikelihoodsNative(Object[] readDataArray, Object[] haplotypeDataArray, double[] likelihoodArray)
 * reads x haplotypes cross product, likelihoodArray[r*numHaps + h] (JavaData.h:177-242, IntelPDHMM.cc:62-133);
 * HaplotypeDataHolder carries haplotypeBases and haplotypePDBases (JavaData.h:172-173). */
JNIEXPORT void JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_computeLikelihoodsNative(
    JNIEnv* env, jobject obj, jobjectArray readDataArray, jobjectArray haplotypeDataArray,
    jdoubleArray likelihoodArray);

/* native double[] computePDHMMNative(byte[] hap_bases, byte[] hap_pdbases, byte[] read_bases, byte[] read_qual,
 *     byte[] read_ins_qual, byte[] read_del_qual, byte[] gcp, long[] hap_lengths, long[] read_lengths,
 *     int testcase, int maxHapLength, int maxReadLength)      (IntelPDHMM.cc:140-243) */
JNIEXPORT jdoubleArray JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_computePDHMMNative(
    JNIEnv* env, jobject obj, jbyteArray hap_bases, jbyteArray hap_pdbases, jbyteArray read_bases,
    jbyteArray read_qual, jbyteArray read_ins_qual, jbyteArray read_del_qual, jbyteArray gcp,
    jlongArray hap_lengths, jlongArray read_lengths, jint testcase, jint maxHapLength, jint maxReadLength);

/* static native void doneNative()   (IntelPDHMM.cc:245-248) */
JNIEXPORT void JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_doneNative(JNIEnv* env, jclass cls);

/* Device probe run by the JVM at System.load (not in the reference): JNI_VERSION_1_8 when a gfx950 device is usable,
 * JNI_ERR otherwise -- load() then returns false and the caller falls back, as it does on a CPU without AVX.
 * GKL_HIP_LOAD_WITHOUT_DEVICE=1 disables the probe.  (gkl_amd/csrc/jni_onload.h) */
#ifndef GKL_USE_SYSTEM_JNI
struct JavaVM_;
#endif
JNIEXPORT jint JNICALL JNI_OnLoad(struct JavaVM_* vm, void* reserved);

#ifdef __cplusplus
}
#endif
#endif /* GKL_PDHMM_JNI_H */
