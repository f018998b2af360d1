// JNI layer of libgkl_smithwaterman_hip.so (built under the reference's name libgkl_smithwaterman.so only by `make dropin-sw`): the three natives of
// com.intel.gkl.smithwaterman.IntelSmithWaterman (include/gkl_sw_jni.h) over the C ABI of
// include/gkl_hip_sw.h.  Replaces the reference's IntelSmithWaterman.cc; the arrays are copied with
// Get/SetByteArrayRegion instead of GetPrimitiveArrayCritical (no JVM critical section is held while the
// GPU works), and the context is created once in initNative instead of three _mm_malloc per call
// (PairWiseSW.h:471-473).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/gkl_hip_pairhmm.h"
#include "../../include/gkl_hip_sw.h"
#include "../../include/gkl_sw_jni.h"
#include "jni_onload.h"

#ifdef GKL_USE_SYSTEM_JNI
namespace gkljni {
inline jclass FindClass(JNIEnv* e, const char* n) { return e->FindClass(n); }
inline jint ThrowNew(JNIEnv* e, jclass c, const char* m) { return e->ThrowNew(c, m); }
inline void ExceptionClear(JNIEnv* e) { e->ExceptionClear(); }
inline jboolean ExceptionCheck(JNIEnv* e) { return e->ExceptionCheck(); }
inline jsize GetArrayLength(JNIEnv* e, jarray a) { return e->GetArrayLength(a); }
inline void GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, jbyte* b) { e->GetByteArrayRegion(a, s, l, b); }
inline void SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, const jbyte* b) { e->SetByteArrayRegion(a, s, l, b); }
inline void GetLongArrayRegion(JNIEnv* e, jlongArray a, jsize s, jsize l, jlong* b) { e->GetLongArrayRegion(a, s, l, b); }
inline void SetIntArrayRegion(JNIEnv* e, jintArray a, jsize s, jsize l, const jint* b) { e->SetIntArrayRegion(a, s, l, b); }
}  // namespace gkljni
#endif

namespace {
constexpr const char* kIAE = "java/lang/IllegalArgumentException";
constexpr const char* kOOM = "java/lang/OutOfMemoryError";
constexpr const char* kRTE = "java/lang/RuntimeException";

struct State {
  std::mutex mu;  // one context, one caller at a time (alignNative is a static native; GATK calls it per thread)
  gklhip_sw_ctx* ctx = nullptr;
  std::vector<int8_t> ref, alt, cigar;
} g;

void throw_java(JNIEnv* env, const char* cls, const char* msg) {
  gkljni::ExceptionClear(env);  // IntelSmithWaterman.cc:85-87,116
  jclass c = gkljni::FindClass(env, cls);
  if (c) gkljni::ThrowNew(env, c, msg);
}

void throw_status(JNIEnv* env, int st) {
  const char* d = gklhip_sw_last_error();
  char msg[600];
  snprintf(msg, sizeof msg, "%s", (d && *d) ? d : "GKL-HIP Smith-Waterman failure");
  // the reference's two messages where it has them (IntelSmithWaterman.cc:88,117)
  if (st == GKLHIP_ERR_OOM) throw_java(env, kOOM, "Memory allocation issue");
  else throw_java(env, st == GKLHIP_ERR_INVALID_ARG ? kIAE : kRTE, msg);
}
}  // namespace

extern "C" {

JNIEXPORT void JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_initNative(JNIEnv* env, jclass) {
  std::lock_guard<std::mutex> lock(g.mu);
  if (g.ctx) return;  // load() may run more than once per JVM (IntelSmithWaterman.java:77-112)
  const char* dev = getenv("GKL_HIP_DEVICE");
  const int st = gklhip_sw_init((dev && *dev) ? atoi(dev) : -1, &g.ctx);
  if (st != GKLHIP_OK) { g.ctx = nullptr; throw_status(env, st); }
}

JNIEXPORT jint JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_alignNative(
    JNIEnv* env, jclass, jbyteArray ref, jbyteArray alt, jbyteArray cigar, jint match, jint mismatch, jint open,
    jint extend, jbyte strategy) {
  if (!ref || !alt || !cigar) { throw_java(env, kIAE, "Arrays aren't valid."); return -1; }  // IntelSmithWaterman.cc:80-104
  std::lock_guard<std::mutex> lock(g.mu);
  if (!g.ctx) { throw_java(env, kRTE, "GKL-HIP Smith-Waterman: alignNative before initNative"); return -1; }
  try {
    const jsize ref_len = gkljni::GetArrayLength(env, ref), alt_len = gkljni::GetArrayLength(env, alt),
                cigar_len = gkljni::GetArrayLength(env, cigar);
    g.ref.resize((size_t)ref_len);
    g.alt.resize((size_t)alt_len);
    g.cigar.assign((size_t)cigar_len, 0);
    if (ref_len > 0) gkljni::GetByteArrayRegion(env, ref, 0, ref_len, g.ref.data());
    if (alt_len > 0) gkljni::GetByteArrayRegion(env, alt, 0, alt_len, g.alt.data());
    if (gkljni::ExceptionCheck(env)) return -1;
    gklhip_sw_params prm{match, mismatch, open, extend};
    uint32_t count = 0;
    int32_t offset = 0;
    const int st = gklhip_sw_align(g.ctx, &prm, (int32_t)strategy, reinterpret_cast<const uint8_t*>(g.ref.data()),
                                   ref_len, reinterpret_cast<const uint8_t*>(g.alt.data()), alt_len,
                                   reinterpret_cast<char*>(g.cigar.data()), cigar_len, &count, &offset);
    if (st != GKLHIP_OK) { throw_status(env, st); return -1; }
    if (cigar_len > 0) gkljni::SetByteArrayRegion(env, cigar, 0, cigar_len, g.cigar.data());
    return offset;
  } catch (const std::bad_alloc&) {
    throw_java(env, kOOM, "Memory allocation issue");
    return -1;
  }
}

JNIEXPORT jint JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_alignBatchNative(
    JNIEnv* env, jclass, jbyteArray refs, jlongArray refOffsets, jbyteArray alts, jlongArray altOffsets,
    jbyteArray cigars, jint cigarStride, jintArray offsets, jint match, jint mismatch, jint open, jint extend,
    jbyte strategy) {
  if (!refs || !refOffsets || !alts || !altOffsets || !cigars || !offsets) { throw_java(env, kIAE, "Arrays aren't valid."); return -1; }
  std::lock_guard<std::mutex> lock(g.mu);
  if (!g.ctx) { throw_java(env, kRTE, "GKL-HIP Smith-Waterman: alignBatchNative before initNative"); return -1; }
  try {
    const jsize n = gkljni::GetArrayLength(env, refOffsets) - 1;
    if (n < 0 || gkljni::GetArrayLength(env, altOffsets) != n + 1 || gkljni::GetArrayLength(env, offsets) < n ||
        cigarStride <= 0 || (int64_t)gkljni::GetArrayLength(env, cigars) < (int64_t)n * cigarStride) {
      throw_java(env, kIAE, "Arrays aren't valid.");
      return -1;
    }
    if (n == 0) return 0;
    std::vector<int64_t> ro((size_t)n + 1), ao((size_t)n + 1);
    gkljni::GetLongArrayRegion(env, refOffsets, 0, n + 1, reinterpret_cast<jlong*>(ro.data()));
    gkljni::GetLongArrayRegion(env, altOffsets, 0, n + 1, reinterpret_cast<jlong*>(ao.data()));
    if (gkljni::ExceptionCheck(env)) return -1;
    const jsize ref_bytes = gkljni::GetArrayLength(env, refs), alt_bytes = gkljni::GetArrayLength(env, alts);
    if (ro[0] != 0 || ao[0] != 0 || ro[(size_t)n] > ref_bytes || ao[(size_t)n] > alt_bytes) {
      throw_java(env, kIAE, "Arrays aren't valid.");
      return -1;
    }
    g.ref.resize((size_t)ro[(size_t)n]);
    g.alt.resize((size_t)ao[(size_t)n]);
    g.cigar.assign((size_t)n * (size_t)cigarStride, 0);
    if (!g.ref.empty()) gkljni::GetByteArrayRegion(env, refs, 0, (jsize)g.ref.size(), g.ref.data());
    if (!g.alt.empty()) gkljni::GetByteArrayRegion(env, alts, 0, (jsize)g.alt.size(), g.alt.data());
    if (gkljni::ExceptionCheck(env)) return -1;
    gklhip_sw_params prm{match, mismatch, open, extend};
    std::vector<uint32_t> counts((size_t)n);
    std::vector<int32_t> offs((size_t)n);
    const int st = gklhip_sw_align_batch(g.ctx, &prm, (int32_t)strategy, n, reinterpret_cast<const uint8_t*>(g.ref.data()),
                                         ro.data(), reinterpret_cast<const uint8_t*>(g.alt.data()), ao.data(),
                                         reinterpret_cast<char*>(g.cigar.data()), cigarStride, counts.data(), offs.data());
    if (st != GKLHIP_OK) { throw_status(env, st); return -1; }
    gkljni::SetByteArrayRegion(env, cigars, 0, (jsize)g.cigar.size(), g.cigar.data());
    gkljni::SetIntArrayRegion(env, offsets, 0, n, reinterpret_cast<const jint*>(offs.data()));
    return n;
  } catch (const std::bad_alloc&) {
    throw_java(env, kOOM, "Memory allocation issue");
    return -1;
  }
}

JNIEXPORT void JNICALL Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_doneNative(JNIEnv*, jclass) {
  std::lock_guard<std::mutex> lock(g.mu);
  if (g.ctx) { gklhip_sw_done(g.ctx); g.ctx = nullptr; }
}

}  // extern "C"
