// JNI drop-in layer of libgkl_pdhmm.so: the four natives of com.intel.gkl.pdhmm.IntelPDHMM
// (include/gkl_pdhmm_jni.h) over the C ABI of include/gkl_hip_pdhmm.h.  Replaces the reference's
// IntelPDHMM.cc + JavaData.h; arrays are copied with Get<T>ArrayRegion (no critical sections held
// across GPU work), every pair goes through the same (vector-arithmetic) kernel.  The holders of
// computeLikelihoodsNative are marshalled like the PairHMM shim's (jni_shim.cpp: marshal_reads): 13 JNI calls per
// read and 7 per haplotype inside a local frame per 32 holders (r05: 40 and 16, plus seven heap vectors per holder --
// for the fixture's 276 x 48 region more JNI time in a real JVM than the 0.3 ms the call itself takes).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/gkl_hip_pairhmm.h"
#include "../../include/gkl_hip_pdhmm.h"
#include "../../include/gkl_pdhmm_jni.h"
#include "jni_onload.h"

#ifdef GKL_USE_SYSTEM_JNI
namespace gkljni {
inline jclass FindClass(JNIEnv* e, const char* n) { return e->FindClass(n); }
inline jint ThrowNew(JNIEnv* e, jclass c, const char* m) { return e->ThrowNew(c, m); }
inline void ExceptionClear(JNIEnv* e) { e->ExceptionClear(); }
inline jboolean ExceptionCheck(JNIEnv* e) { return e->ExceptionCheck(); }
inline void DeleteLocalRef(JNIEnv* e, jobject o) { e->DeleteLocalRef(o); }
inline jint PushLocalFrame(JNIEnv* e, jint capacity) { return e->PushLocalFrame(capacity); }
inline jobject PopLocalFrame(JNIEnv* e, jobject result) { return e->PopLocalFrame(result); }
inline jfieldID GetFieldID(JNIEnv* e, jclass c, const char* n, const char* s) { return e->GetFieldID(c, n, s); }
inline jobject GetObjectField(JNIEnv* e, jobject o, jfieldID f) { return e->GetObjectField(o, f); }
inline jsize GetArrayLength(JNIEnv* e, jarray a) { return e->GetArrayLength(a); }
inline jobject GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) { return e->GetObjectArrayElement(a, i); }
inline void GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, jbyte* b) { e->GetByteArrayRegion(a, s, l, b); }
inline void GetLongArrayRegion(JNIEnv* e, jlongArray a, jsize s, jsize l, jlong* b) { e->GetLongArrayRegion(a, s, l, b); }
inline jdoubleArray NewDoubleArray(JNIEnv* e, jsize l) { return e->NewDoubleArray(l); }
inline void SetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize s, jsize l, const jdouble* b) { e->SetDoubleArrayRegion(a, s, l, b); }
}  // namespace gkljni
#endif

namespace {
constexpr const char* kIAE = "java/lang/IllegalArgumentException";
constexpr const char* kOOM = "java/lang/OutOfMemoryError";
constexpr const char* kRTE = "java/lang/RuntimeException";

// The context is shared by reference count: a call holds its own reference for as long as it runs, so a second
// IntelPDHMM.initialize() or a close() on another thread can never free a context under a running call (the
// reference has no such state at all: it re-allocates its DP table on every call, IntelPDHMM.cc:101-120).
struct State {
  std::mutex mu;
  std::shared_ptr<gklhip_pdhmm_ctx> ctx;
  bool initialised = false;  // initNative has cached the field IDs: a context can be (re)created on demand
  int max_memory_mb = 512;
  jfieldID readBases = nullptr, readQuals = nullptr, insertionGOP = nullptr, deletionGOP = nullptr,
           overallGCP = nullptr, haplotypeBases = nullptr, haplotypePDBases = nullptr;
} g;

void throw_java(JNIEnv* env, const char* cls, const char* msg) {
  gkljni::ExceptionClear(env);
  jclass c = gkljni::FindClass(env, cls);
  if (c) gkljni::ThrowNew(env, c, msg);
}

void throw_status(JNIEnv* env, int st) {
  const char* d = gklhip_pdhmm_last_error();
  char msg[600];
  snprintf(msg, sizeof msg, "%s", (d && *d) ? d : "GKL-HIP PDHMM failure");
  throw_java(env, st == GKLHIP_ERR_INVALID_ARG ? kIAE : st == GKLHIP_ERR_OOM ? kOOM : kRTE, msg);
}

// New context on the configured device; NULL after throwing.  Called with g.mu held.
std::shared_ptr<gklhip_pdhmm_ctx> make_context(JNIEnv* env) {
  const char* dev = getenv("GKL_HIP_DEVICE");
  gklhip_pdhmm_ctx* raw = nullptr;
  const int st = gklhip_pdhmm_init((dev && *dev) ? atoi(dev) : -1, &raw);
  if (st != GKLHIP_OK) { throw_status(env, st); return nullptr; }
  std::shared_ptr<gklhip_pdhmm_ctx> c(raw, [](gklhip_pdhmm_ctx* p) { gklhip_pdhmm_done(p); });
  const char* fm = getenv("GKL_HIP_FMA_MODE");  // 1 (default): GKL's AVX-512 arithmetic, 0: its AVX2 arithmetic
  if (fm && *fm) {
    const int st2 = gklhip_pdhmm_set_fma_mode(raw, atoi(fm));
    if (st2 != GKLHIP_OK) { throw_status(env, st2); return nullptr; }
  }
  return c;
}

// The caller's own reference to the context.  After doneNative the reference keeps working (it holds no state),
// so a compute call then gets a fresh context here; before any initNative it throws.  NULL after throwing.
std::shared_ptr<gklhip_pdhmm_ctx> context(JNIEnv* env, const char* who) {
  std::lock_guard<std::mutex> lock(g.mu);
  if (!g.ctx) {
    if (!g.initialised) {
      char msg[128];
      snprintf(msg, sizeof msg, "GKL-HIP PDHMM: %s before initNative", who);
      throw_java(env, kRTE, msg);
      return nullptr;
    }
    g.ctx = make_context(env);
  }
  return g.ctx;
}

// PushLocalFrame / PopLocalFrame around a block of holders (also when a C++ exception passes through)
struct LocalFrame {
  JNIEnv* env;
  bool pushed;
  LocalFrame(JNIEnv* e, jint capacity) : env(e), pushed(gkljni::PushLocalFrame(e, capacity) == 0) {
    if (!pushed) gkljni::ExceptionClear(env);   // (PushLocalFrame raised OutOfMemoryError)
  }
  ~LocalFrame() { if (pushed) gkljni::PopLocalFrame(env, nullptr); }
  LocalFrame(const LocalFrame&) = delete;
};
constexpr jsize kFrameHolders = 32;

// The byte[] fields `fids[0..n_fields)` of holders[0..n): field f of holder i -> flat[f][off[i] .. off[i+1]) (unpadded, one
// growing array per field), the length of a holder = the length of its FIRST field (JavaData.h:190-196: readBases /
// haplotypeBases); an array shorter than that surfaces as the region copy's ArrayIndexOutOfBoundsException and is
// reported as IllegalArgumentException(`short_msg`), an empty first field as `empty_msg`.  Per holder: the element,
// n_fields fields, one length, n_fields region copies, one exception check.  False after throwing.
bool marshal_holders(JNIEnv* env, jobjectArray holders, jsize n, const jfieldID* fids, int n_fields, std::vector<int8_t>* flat,
                     std::vector<int64_t>& off, const char* empty_msg, const char* short_msg) {
  off.assign((size_t)n + 1, 0);
  for (int f = 0; f < n_fields; f++) flat[f].clear();
  for (jsize b0 = 0; b0 < n; b0 += kFrameHolders) {
    const jsize b1 = std::min<jsize>(n, b0 + kFrameHolders);
    const char* err = nullptr;
    {
      LocalFrame frame(env, 8 * kFrameHolders);
      if (!frame.pushed) { throw_java(env, kOOM, "Memory allocation issue."); return false; }
      for (jsize i = b0; i < b1 && !err; i++) {
        jobject holder = gkljni::GetObjectArrayElement(env, holders, i);
        if (!holder) { err = "null element in data holder array"; break; }
        jbyteArray arr[8];
        for (int f = 0; f < n_fields; f++) {
          arr[f] = (jbyteArray)gkljni::GetObjectField(env, holder, fids[f]);
          if (!arr[f]) err = "null byte[] field in data holder";
        }
        if (err) break;
        const jsize len = gkljni::GetArrayLength(env, arr[0]);
        if (len == 0) { err = empty_msg; break; }
        const size_t at = (size_t)off[(size_t)i];
        for (int f = 0; f < n_fields; f++) flat[f].resize(at + (size_t)len);   // (space for all first: see jni_shim.cpp)
        for (int f = 0; f < n_fields; f++) gkljni::GetByteArrayRegion(env, arr[f], 0, len, reinterpret_cast<jbyte*>(flat[f].data() + at));
        if (gkljni::ExceptionCheck(env)) { gkljni::ExceptionClear(env); err = short_msg; break; }
        off[(size_t)i + 1] = off[(size_t)i] + len;
      }
    }
    if (err) { throw_java(env, kIAE, err); return false; }
  }
  return true;
}
}  // namespace

extern "C" {

JNIEXPORT void JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_initNative(JNIEnv* env, jclass, jclass readDataHolder,
                                                                     jclass haplotypeDataHolder, jint, jint, jint,
                                                                     jint maxMemoryInMB) {
  std::lock_guard<std::mutex> lock(g.mu);
  struct { jfieldID* dst; jclass cls; const char* name; } fields[] = {
      {&g.readBases, readDataHolder, "readBases"},       {&g.readQuals, readDataHolder, "readQuals"},
      {&g.insertionGOP, readDataHolder, "insertionGOP"}, {&g.deletionGOP, readDataHolder, "deletionGOP"},
      {&g.overallGCP, readDataHolder, "overallGCP"},     {&g.haplotypeBases, haplotypeDataHolder, "haplotypeBases"},
      {&g.haplotypePDBases, haplotypeDataHolder, "haplotypePDBases"}};
  for (auto& f : fields) {
    jfieldID id = f.cls ? gkljni::GetFieldID(env, f.cls, f.name, "[B") : nullptr;
    if (!id) { throw_java(env, kIAE, "Unable to get field ID"); return; }  // JavaData.h:282-290
    *f.dst = id;
  }
  // (the free-RAM cap is applied here, once -- pdhmm-implementation.h:204-235 -- not per call)
  g.max_memory_mb = gklhip_pdhmm_available_memory_mb(maxMemoryInMB > 0 ? maxMemoryInMB : 512);
  g.initialised = true;
  // a second initialize() keeps the context (calls of other threads may be running on it); the first one creates it
  // here so that "no GPU" surfaces from initNative
  if (!g.ctx) g.ctx = make_context(env);
}

JNIEXPORT void JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_computeLikelihoodsNative(
    JNIEnv* env, jobject, jobjectArray readDataArray, jobjectArray haplotypeDataArray, jdoubleArray likelihoodArray) {
  if (!readDataArray || !haplotypeDataArray || !likelihoodArray) { throw_java(env, kIAE, "null argument"); return; }
  const std::shared_ptr<gklhip_pdhmm_ctx> ctx_ref = context(env, "computeLikelihoodsNative");
  gklhip_pdhmm_ctx* ctx = ctx_ref.get();
  if (!ctx) return;
  int max_memory_mb;
  {
    std::lock_guard<std::mutex> lock(g.mu);
    max_memory_mb = g.max_memory_mb;
  }
  try {
    const jsize n_reads = gkljni::GetArrayLength(env, readDataArray);
    const jsize n_haps = gkljni::GetArrayLength(env, haplotypeDataArray);
    const int64_t total = (int64_t)n_reads * n_haps;
    if (total == 0) {  // JavaData.h:93-96
      throw_java(env, kIAE, "Batch size is too small because there are no pairs to process. Ensure that the input arrays are not empty.");
      return;
    }
    if (total > 0x7fffffffLL || (int64_t)gkljni::GetArrayLength(env, likelihoodArray) < total) {
      throw_java(env, kIAE, "likelihoodArray length must be equal to readDataArray length * haplotypeDataArray length");
      return;
    }
    std::vector<int8_t> rflat[5], hflat[2];
    std::vector<int64_t> roff, hoff;
    const jfieldID rf[5] = {g.readBases, g.readQuals, g.insertionGOP, g.deletionGOP, g.overallGCP};
    const jfieldID hf[2] = {g.haplotypeBases, g.haplotypePDBases};
    if (!marshal_holders(env, readDataArray, n_reads, rf, 5, rflat, roff, "empty read or read quality array shorter than readBases",
                         "empty read or read quality array shorter than readBases") ||
        !marshal_holders(env, haplotypeDataArray, n_haps, hf, 2, hflat, hoff, "empty haplotype or haplotypePDBases shorter than haplotypeBases",
                         "empty haplotype or haplotypePDBases shorter than haplotypeBases"))
      return;
    int max_r = 0, max_h = 0;
    for (jsize r = 0; r < n_reads; r++) max_r = (int)std::max<int64_t>(max_r, roff[(size_t)r + 1] - roff[(size_t)r]);
    for (jsize h = 0; h < n_haps; h++) max_h = (int)std::max<int64_t>(max_h, hoff[(size_t)h + 1] - hoff[(size_t)h]);
    // The reference expands the cross product into padded PAIRS, in batches of min(total, maxMemoryInMB /
    // memoryPerPair) pairs (JavaData.h:83-101,177-242), because computePDHMM takes pairs; every batch ends in its own
    // scalar tail.  Here every read and every haplotype is staged once and the device walks the cross product itself;
    // the batch size is computed the reference's way only to put the tails where GKL puts them (and to raise its error).
    const int64_t ref_batch = gklhip_pdhmm_reference_batch_pairs(max_memory_mb, max_r, max_h, total);
    if (ref_batch <= 0) {
      throw_java(env, kIAE, "Batch size is too small. Please increase the memory limit for PDHMM by using the maxMemoryInMB argument.");
      return;
    }
    std::vector<int8_t> b_hb((size_t)n_haps * max_h, 0), b_hp((size_t)n_haps * max_h, 0), b_rb((size_t)n_reads * max_r, 0),
        b_rq((size_t)n_reads * max_r, 0), b_ri((size_t)n_reads * max_r, 0), b_rd((size_t)n_reads * max_r, 0),
        b_rc((size_t)n_reads * max_r, 0);
    std::vector<int64_t> hl((size_t)n_haps), rl((size_t)n_reads);
    for (jsize r = 0; r < n_reads; r++) {
      const size_t at = (size_t)roff[(size_t)r], R = (size_t)(roff[(size_t)r + 1] - roff[(size_t)r]);
      memcpy(&b_rb[(size_t)r * max_r], rflat[0].data() + at, R); memcpy(&b_rq[(size_t)r * max_r], rflat[1].data() + at, R);
      memcpy(&b_ri[(size_t)r * max_r], rflat[2].data() + at, R); memcpy(&b_rd[(size_t)r * max_r], rflat[3].data() + at, R);
      memcpy(&b_rc[(size_t)r * max_r], rflat[4].data() + at, R);
      rl[(size_t)r] = (int64_t)R;
    }
    for (jsize h = 0; h < n_haps; h++) {
      const size_t at = (size_t)hoff[(size_t)h], H = (size_t)(hoff[(size_t)h + 1] - hoff[(size_t)h]);
      memcpy(&b_hb[(size_t)h * max_h], hflat[0].data() + at, H); memcpy(&b_hp[(size_t)h * max_h], hflat[1].data() + at, H);
      hl[(size_t)h] = (int64_t)H;
    }
    std::vector<double> out((size_t)total);
    gklhip_pdhmm_cross x = {n_reads, n_haps, max_h, max_r, b_hb.data(), b_hp.data(), b_rb.data(), b_rq.data(),
                            b_ri.data(), b_rd.data(), b_rc.data(), hl.data(), rl.data()};
    const int st = gklhip_pdhmm_compute_cross_batched(ctx, &x, ref_batch, out.data());
    if (st != GKLHIP_OK) { throw_status(env, st); return; }
    gkljni::SetDoubleArrayRegion(env, likelihoodArray, 0, (jsize)total, out.data());
  } catch (const std::bad_alloc&) {
    throw_java(env, kOOM, "Memory allocation issue.");
  }
}

JNIEXPORT jdoubleArray JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_computePDHMMNative(
    JNIEnv* env, jobject, jbyteArray jhap_bases, jbyteArray jhap_pdbases, jbyteArray jread_bases, jbyteArray jread_qual,
    jbyteArray jread_ins_qual, jbyteArray jread_del_qual, jbyteArray jgcp, jlongArray jhap_lengths,
    jlongArray jread_lengths, jint testcase, jint maxHapLength, jint maxReadLength) {
  const std::shared_ptr<gklhip_pdhmm_ctx> ctx_ref = context(env, "computePDHMMNative");
  gklhip_pdhmm_ctx* ctx = ctx_ref.get();
  if (!ctx) return nullptr;
  if (!jhap_bases || !jhap_pdbases || !jread_bases || !jread_qual || !jread_ins_qual || !jread_del_qual || !jgcp ||
      !jhap_lengths || !jread_lengths || testcase <= 0 || maxHapLength <= 0 || maxReadLength <= 0) {
    throw_java(env, kIAE, "Input arrays aren't valid.");  // IntelPDHMM.cc:165-189
    return nullptr;
  }
  try {
    const size_t hb = (size_t)testcase * (size_t)maxHapLength, rb = (size_t)testcase * (size_t)maxReadLength;
    jbyteArray arrs[7] = {jhap_bases, jhap_pdbases, jread_bases, jread_qual, jread_ins_qual, jread_del_qual, jgcp};
    const size_t want[7] = {hb, hb, rb, rb, rb, rb, rb};
    std::vector<int8_t> bufs[7];
    for (int i = 0; i < 7; i++) {
      if ((size_t)gkljni::GetArrayLength(env, arrs[i]) < want[i]) { throw_java(env, kIAE, "Input arrays aren't valid."); return nullptr; }
      bufs[i].resize(want[i]);
      gkljni::GetByteArrayRegion(env, arrs[i], 0, (jsize)want[i], reinterpret_cast<jbyte*>(bufs[i].data()));
    }
    if (gkljni::GetArrayLength(env, jhap_lengths) < testcase || gkljni::GetArrayLength(env, jread_lengths) < testcase) {
      throw_java(env, kIAE, "Input arrays aren't valid.");
      return nullptr;
    }
    std::vector<int64_t> hl((size_t)testcase), rl((size_t)testcase);
    gkljni::GetLongArrayRegion(env, jhap_lengths, 0, testcase, reinterpret_cast<jlong*>(hl.data()));
    gkljni::GetLongArrayRegion(env, jread_lengths, 0, testcase, reinterpret_cast<jlong*>(rl.data()));
    if (gkljni::ExceptionCheck(env)) return nullptr;
    jdoubleArray jresult = gkljni::NewDoubleArray(env, testcase);
    if (!jresult) { throw_java(env, kOOM, "Memory allocation issue."); return nullptr; }
    std::vector<double> out((size_t)testcase);
    gklhip_pdhmm_batch b = {testcase, maxHapLength, maxReadLength, bufs[0].data(), bufs[1].data(), bufs[2].data(),
                            bufs[3].data(), bufs[4].data(), bufs[5].data(), bufs[6].data(), hl.data(), rl.data()};
    const int st = gklhip_pdhmm_compute(ctx, &b, out.data());
    if (st != GKLHIP_OK) { throw_status(env, st); return jresult; }  // the reference also returns the array (IntelPDHMM.cc:241)
    gkljni::SetDoubleArrayRegion(env, jresult, 0, testcase, out.data());
    return jresult;
  } catch (const std::bad_alloc&) {
    throw_java(env, kOOM, "Memory allocation issue.");
    return nullptr;
  }
}

JNIEXPORT void JNICALL Java_com_intel_gkl_pdhmm_IntelPDHMM_doneNative(JNIEnv*, jclass) {
  // drops the library's reference: device memory goes when the last running call has returned; a later compute call
  // gets a fresh context (the reference's doneNative frees its DP table and the next call re-allocates it)
  std::lock_guard<std::mutex> lock(g.mu);
  g.ctx.reset();
}

}  // extern "C"
