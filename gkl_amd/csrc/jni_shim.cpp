// JNI drop-in layer of libgkl_pairhmm.so: the three natives of
// com.intel.gkl.pairhmm.IntelPairHmm (include/gkl_pairhmm_jni.h) as thin shims over the
// C ABI.  Replaces the reference's L2 (src/main/native/pairhmm/IntelPairHmm.cc +
// JavaData.h); differences, all deliberate:
//   * byte[] fields are COPIED with GetByteArrayRegion into one flat batch instead of
//     being pinned one by one (50k pins at 10k reads) and described by a 56-byte
//     testcase per PAIR (JavaData.h:94-110); local refs are deleted as we go
//     (the reference leaks them until return, JavaData.h:135-145);
//   * null holders / null byte[] fields / a too-short likelihood array raise
//     IllegalArgumentException instead of crashing the JVM;
//   * HIP failures raise java/lang/RuntimeException, allocation failures
//     java/lang/OutOfMemoryError (the reference's two classes, JavaData.h:130,140,150).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/gkl_hip_pairhmm.h"
#include "../../include/gkl_pairhmm_jni.h"

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
inline void SetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize s, jsize l, const jdouble* b) { e->SetDoubleArrayRegion(a, s, l, b); }
}  // namespace gkljni
#endif

namespace {

constexpr const char* kIAE = "java/lang/IllegalArgumentException";
constexpr const char* kOOM = "java/lang/OutOfMemoryError";
constexpr const char* kRTE = "java/lang/RuntimeException";

// Process-wide state, like the reference's globals (IntelPairHmm.cc:41-48) and the
// static field IDs of JavaData (JavaData.h:160-176).
struct State {
  std::mutex mu;
  gklhip_ctx* ctx = nullptr;
  jfieldID readBases = nullptr, readQuals = nullptr, insertionGOP = nullptr, deletionGOP = nullptr,
           overallGCP = nullptr, haplotypeBases = nullptr;
} g;

void throw_java(JNIEnv* env, const char* class_path, const char* msg) {
  gkljni::ExceptionClear(env);  // IntelPairHmm.cc:65-66
  jclass c = gkljni::FindClass(env, class_path);
  if (c) gkljni::ThrowNew(env, c, msg);
}

void throw_status(JNIEnv* env, int status) {
  const char* detail = gklhip_last_error();
  char msg[600];
  snprintf(msg, sizeof msg, "GKL-HIP PairHMM: %s%s%s", gklhip_strerror(status),
           (detail && *detail) ? ": " : "", (detail && *detail) ? detail : "");
  const char* cls = status == GKLHIP_ERR_INVALID_ARG ? kIAE : status == GKLHIP_ERR_OOM ? kOOM : kRTE;
  throw_java(env, cls, msg);
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// Append holder[i].<field> (a byte[]) to dst; returns its length, or -1 after throwing.
long append_field(JNIEnv* env, jobjectArray arr, jsize i, jfieldID fid, std::vector<uint8_t>& dst,
                  long expect_at_least) {
  jobject holder = gkljni::GetObjectArrayElement(env, arr, i);
  if (gkljni::ExceptionCheck(env)) return -1;
  if (!holder) { throw_java(env, kIAE, "null element in data holder array"); return -1; }
  jbyteArray bytes = (jbyteArray)gkljni::GetObjectField(env, holder, fid);
  if (!bytes) {
    gkljni::DeleteLocalRef(env, holder);
    throw_java(env, kIAE, "null byte[] field in data holder");
    return -1;
  }
  const jsize len = gkljni::GetArrayLength(env, bytes);
  long take = len;
  if (expect_at_least >= 0) {
    // JavaData.h:86-91: the read length is readBases.length; the other four arrays are read
    // for that many bytes. A shorter array is an error here (the reference reads past it).
    if (len < expect_at_least) {
      gkljni::DeleteLocalRef(env, bytes);
      gkljni::DeleteLocalRef(env, holder);
      throw_java(env, kIAE, "read quality array shorter than readBases");
      return -1;
    }
    take = expect_at_least;
  }
  const size_t at = dst.size();
  dst.resize(at + (size_t)take);
  if (take > 0) gkljni::GetByteArrayRegion(env, bytes, 0, (jsize)take, reinterpret_cast<jbyte*>(dst.data() + at));
  gkljni::DeleteLocalRef(env, bytes);
  gkljni::DeleteLocalRef(env, holder);
  if (gkljni::ExceptionCheck(env)) return -1;
  return take;
}

}  // namespace

extern "C" {

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative(
    JNIEnv* env, jclass, jclass readDataHolder, jclass haplotypeDataHolder, jboolean use_double,
    jint max_threads) {
  std::lock_guard<std::mutex> lock(g.mu);
  struct { jfieldID* dst; jclass cls; const char* name; } fields[] = {
      {&g.readBases, readDataHolder, "readBases"},       {&g.readQuals, readDataHolder, "readQuals"},
      {&g.insertionGOP, readDataHolder, "insertionGOP"}, {&g.deletionGOP, readDataHolder, "deletionGOP"},
      {&g.overallGCP, readDataHolder, "overallGCP"},     {&g.haplotypeBases, haplotypeDataHolder, "haplotypeBases"}};
  for (auto& f : fields) {
    jfieldID id = f.cls ? gkljni::GetFieldID(env, f.cls, f.name, "[B") : nullptr;
    if (!id) {  // JavaData.h:127-133
      throw_java(env, kIAE, "Unable to get field ID");
      return;
    }
    *f.dst = id;
  }
  if (g.ctx) { gklhip_done(g.ctx); g.ctx = nullptr; }
  gklhip_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = GKLHIP_ABI_VERSION;
  cfg.device = env_int("GKL_HIP_DEVICE", -1);
  cfg.use_double = use_double ? 1 : 0;
  cfg.max_threads = max_threads;
  cfg.fma_mode = env_int("GKL_HIP_FMA_MODE", 1);
  cfg.finalize = env_int("GKL_HIP_FINALIZE", GKLHIP_FINALIZE_REFERENCE_HOST);
  cfg.record_events = 0;
  cfg.rows_per_lane = 0;
  const int st = gklhip_init(&cfg, &g.ctx);
  if (st != GKLHIP_OK) { g.ctx = nullptr; throw_status(env, st); }
}

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative(
    JNIEnv* env, jobject, jobjectArray readDataArray, jobjectArray haplotypeDataArray,
    jdoubleArray likelihoodArray) {
  if (!readDataArray || !haplotypeDataArray || !likelihoodArray) {
    throw_java(env, kIAE, "null argument");  // the Java wrapper already raised NPE (IntelPairHmm.java:134-136)
    return;
  }
  gklhip_ctx* ctx;
  {
    std::lock_guard<std::mutex> lock(g.mu);
    ctx = g.ctx;
  }
  if (!ctx) { throw_java(env, kRTE, "GKL-HIP PairHMM: computeLikelihoodsNative before initNative"); return; }
  try {
    const jsize n_reads = gkljni::GetArrayLength(env, readDataArray);
    const jsize n_haps = gkljni::GetArrayLength(env, haplotypeDataArray);
    std::vector<uint8_t> hap_bases, read_bases, read_quals, ins, del, gcp;
    std::vector<int64_t> hap_off((size_t)n_haps + 1, 0), read_off((size_t)n_reads + 1, 0);
    for (jsize h = 0; h < n_haps; h++) {
      const long len = append_field(env, haplotypeDataArray, h, g.haplotypeBases, hap_bases, -1);
      if (len < 0) return;
      hap_off[h + 1] = hap_off[h] + len;
    }
    for (jsize r = 0; r < n_reads; r++) {
      const long len = append_field(env, readDataArray, r, g.readBases, read_bases, -1);
      if (len < 0) return;
      if (append_field(env, readDataArray, r, g.insertionGOP, ins, len) < 0) return;
      if (append_field(env, readDataArray, r, g.deletionGOP, del, len) < 0) return;
      if (append_field(env, readDataArray, r, g.overallGCP, gcp, len) < 0) return;
      if (append_field(env, readDataArray, r, g.readQuals, read_quals, len) < 0) return;
      read_off[r + 1] = read_off[r] + len;
    }
    const int64_t n_pairs = (int64_t)n_reads * n_haps;
    if (n_pairs > 0x7fffffffLL) { throw_java(env, kIAE, "more than 2^31 read x haplotype pairs"); return; }
    if ((int64_t)gkljni::GetArrayLength(env, likelihoodArray) < n_pairs) {
      throw_java(env, kIAE, "likelihood array shorter than reads x haplotypes");
      return;
    }
    if (n_pairs == 0) return;
    gklhip_batch b;
    b.n_reads = n_reads; b.n_haps = n_haps;
    b.read_off = read_off.data(); b.hap_off = hap_off.data();
    b.read_bases = read_bases.data(); b.read_quals = read_quals.data(); b.ins_gop = ins.data();
    b.del_gop = del.data(); b.gcp = gcp.data(); b.hap_bases = hap_bases.data();
    std::vector<double> out((size_t)n_pairs);
    const int st = gklhip_compute(ctx, &b, out.data());
    if (st != GKLHIP_OK) { throw_status(env, st); return; }
    gkljni::SetDoubleArrayRegion(env, likelihoodArray, 0, (jsize)n_pairs, out.data());
  } catch (const std::bad_alloc&) {
    throw_java(env, kOOM, "Unable to allocate the PairHMM batch");
  }
}

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative(JNIEnv*, jobject) {
  std::lock_guard<std::mutex> lock(g.mu);
  if (g.ctx) { gklhip_done(g.ctx); g.ctx = nullptr; }
}

}  // extern "C"
