// JNI drop-in layer of libgkl_pdhmm.so: the four natives of com.intel.gkl.pdhmm.IntelPDHMM
// (include/gkl_pdhmm_jni.h) over the C ABI of include/gkl_hip_pdhmm.h.  Replaces the reference's
// IntelPDHMM.cc + JavaData.h; arrays are copied with Get<T>ArrayRegion (no critical sections held
// across GPU work), every pair goes through the same (vector-arithmetic) kernel.
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

// holder[i].<field> -> bytes; false after throwing
bool read_field(JNIEnv* env, jobjectArray arr, jsize i, jfieldID fid, std::vector<int8_t>& dst) {
  jobject holder = gkljni::GetObjectArrayElement(env, arr, i);
  if (gkljni::ExceptionCheck(env)) return false;
  if (!holder) { throw_java(env, kIAE, "null element in data holder array"); return false; }
  jbyteArray bytes = (jbyteArray)gkljni::GetObjectField(env, holder, fid);
  if (!bytes) { gkljni::DeleteLocalRef(env, holder); throw_java(env, kIAE, "null byte[] field in data holder"); return false; }
  const jsize len = gkljni::GetArrayLength(env, bytes);
  dst.resize((size_t)len);
  if (len > 0) gkljni::GetByteArrayRegion(env, bytes, 0, len, reinterpret_cast<jbyte*>(dst.data()));
  gkljni::DeleteLocalRef(env, bytes);
  gkljni::DeleteLocalRef(env, holder);
  return !gkljni::ExceptionCheck(env);
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
    std::vector<std::vector<int8_t>> rb(n_reads), rq(n_reads), ri(n_reads), rd(n_reads), rc(n_reads), hb(n_haps), hp(n_haps);
    int max_r = 0, max_h = 0;
    for (jsize r = 0; r < n_reads; r++) {
      if (!read_field(env, readDataArray, r, g.readBases, rb[r]) || !read_field(env, readDataArray, r, g.readQuals, rq[r]) ||
          !read_field(env, readDataArray, r, g.insertionGOP, ri[r]) || !read_field(env, readDataArray, r, g.deletionGOP, rd[r]) ||
          !read_field(env, readDataArray, r, g.overallGCP, rc[r]))
        return;
      const size_t len = rb[r].size();
      if (len == 0 || rq[r].size() < len || ri[r].size() < len || rd[r].size() < len || rc[r].size() < len) {
        throw_java(env, kIAE, "empty read or read quality array shorter than readBases");
        return;
      }
      max_r = (int)std::max<size_t>(max_r, len);
    }
    for (jsize h = 0; h < n_haps; h++) {
      if (!read_field(env, haplotypeDataArray, h, g.haplotypeBases, hb[h]) ||
          !read_field(env, haplotypeDataArray, h, g.haplotypePDBases, hp[h]))
        return;
      if (hb[h].empty() || hp[h].size() < hb[h].size()) { throw_java(env, kIAE, "empty haplotype or haplotypePDBases shorter than haplotypeBases"); return; }
      max_h = (int)std::max<size_t>(max_h, hb[h].size());
    }
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
      const size_t R = rb[r].size();
      memcpy(&b_rb[(size_t)r * max_r], rb[r].data(), R); memcpy(&b_rq[(size_t)r * max_r], rq[r].data(), R);
      memcpy(&b_ri[(size_t)r * max_r], ri[r].data(), R); memcpy(&b_rd[(size_t)r * max_r], rd[r].data(), R);
      memcpy(&b_rc[(size_t)r * max_r], rc[r].data(), R);
      rl[(size_t)r] = (int64_t)R;
    }
    for (jsize h = 0; h < n_haps; h++) {
      const size_t H = hb[h].size();
      memcpy(&b_hb[(size_t)h * max_h], hb[h].data(), H); memcpy(&b_hp[(size_t)h * max_h], hp[h].data(), H);
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
