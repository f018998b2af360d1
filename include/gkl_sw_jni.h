/*
 * gkl_sw_jni.h -- the JNI symbols of libgkl_smithwaterman_hip.so (= libgkl_smithwaterman.so after `make dropin-sw`): what GKL's
 * com.intel.gkl.smithwaterman.IntelSmithWaterman binds (reference
 * src/main/java/com/intel/gkl/smithwaterman/IntelSmithWaterman.java:188-190; native prototypes
 * src/main/native/smithwaterman/IntelSmithWaterman.h:32-52; bodies IntelSmithWaterman.cc:47-132).
 * Thin shims (gkl_amd/csrc/jni_shim_sw.cpp) over the C ABI of include/gkl_hip_sw.h.
 */
#ifndef GKL_SW_JNI_H
#define GKL_SW_JNI_H

#ifdef GKL_USE_SYSTEM_JNI
#include <jni.h>
#else
#include "../gkl_amd/csrc/jni_min.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* private native static void initNative()   (IntelSmithWaterman.cc:47-66: picks the AVX2 or AVX-512 engine;
 * here: creates the device context; no usable gfx950 device -> java/lang/RuntimeException) */
JNIEXPORT void JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_initNative(JNIEnv* env, jclass cls);

/* private native static int alignNative(byte[] refArray, byte[] altArray, byte[] cigar, int match, int mismatch,
 *                                        int open, int extend, byte strategy)          (IntelSmithWaterman.cc:71-124)
 * Writes the CIGAR text into `cigar` (the Java side sized it 2*max(ref, alt) and trims the trailing zeros) and
 * returns the alignment offset; on error throws IllegalArgumentException / OutOfMemoryError and returns -1. */
JNIEXPORT jint JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_alignNative(
    JNIEnv* env, jclass cls, jbyteArray ref, jbyteArray alt, jbyteArray cigar, jint match, jint mismatch, jint open,
    jint extend, jbyte strategy);

/* NOT in the reference: the batch entry point a caller has to adopt for the GPU to pay off.
 *   private native static int alignBatchNative(byte[] refs, long[] refOffsets, byte[] alts, long[] altOffsets,
 *                                              byte[] cigars, int cigarStride, int[] offsets,
 *                                              int match, int mismatch, int open, int extend, byte strategy)
 * n = refOffsets.length - 1 pairs; pair k aligns refs[refOffsets[k] .. refOffsets[k+1]) against
 * alts[altOffsets[k] .. altOffsets[k+1]); its CIGAR text goes to cigars[k * cigarStride ..] (zero padded, like
 * alignNative's array), its alignment offset to offsets[k].  Returns n, or -1 after throwing. */
JNIEXPORT jint JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_alignBatchNative(
    JNIEnv* env, jclass cls, jbyteArray refs, jlongArray refOffsets, jbyteArray alts, jlongArray altOffsets,
    jbyteArray cigars, jint cigarStride, jintArray offsets, jint match, jint mismatch, jint open, jint extend,
    jbyte strategy);

/* private native static void doneNative()   (IntelSmithWaterman.cc:130-132) */
JNIEXPORT void JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_doneNative(JNIEnv* env, jclass cls);

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
#endif /* GKL_SW_JNI_H */
