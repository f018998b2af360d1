// Test harness: a mock JVM side for libgkl_pairhmm.so (no JDK in this image).
//
// Builds a JNINativeInterface_ function table (spec slot indices, gkl_amd/csrc/jni_min.h)
// over a tiny fake object model, dlopen()s the drop-in library the way
// NativeLibraryLoader/System.load would (reference
// src/main/java/com/intel/gkl/NativeLibraryLoader.java:114-128), resolves the three
// Java_com_intel_gkl_pairhmm_IntelPairHmm_* symbols by name like the JVM does, and drives
// initNative -> computeLikelihoodsNative -> doneNative with ReadDataHolder /
// HaplotypeDataHolder look-alikes.  Every JNI slot the shim does not declare aborts, so the
// test also pins the exact set of JNI functions the shim may call.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../gkl_amd/csrc/jni_min.h"

namespace {

struct Obj {
  enum Kind { CLASS, BYTES, DOUBLES, LONGS, INTS, OBJARRAY, HOLDER } kind;
  std::string name;                      // CLASS
  std::set<std::string> class_fields;    // CLASS
  std::vector<int8_t> bytes;             // BYTES
  std::vector<double> doubles;           // DOUBLES
  std::vector<int64_t> longs;            // LONGS
  std::vector<int32_t> ints;             // INTS
  std::vector<Obj*> elems;               // OBJARRAY
  std::map<std::string, Obj*> fields;    // HOLDER (value may be nullptr)
};

struct Mock {
  JNIEnv_ env;                 // must be first: JNIEnv* == Mock*
  JNINativeInterface_ table;
  std::vector<std::unique_ptr<Obj>> heap;
  std::set<std::string> field_names;     // interned jfieldIDs
  bool pending = false;
  std::string exc_class, exc_msg;
  long refs_created = 0, refs_deleted = 0, unimplemented_calls = 0;
  // What -Xcheck:jni enforces (the reference's test JVMs run with it, build.gradle:101-104):
  //  * no JNI call with an exception pending, except the handful the specification allows;
  //  * at most 16 live local references without EnsureLocalCapacity;
  //  * DeleteLocalRef only of references that are live.
  long violations = 0, max_live_refs = 0;
  std::string first_violation;
  std::multiset<const void*> live;
  void violation(const std::string& what) { if (!violations++) first_violation = what; }
  void check_no_pending(const char* fn) { if (pending) violation(std::string(fn) + " called with an exception pending (" + exc_class + ")"); }
  void hand_out(const void* ref) {
    refs_created++;
    live.insert(ref);
    max_live_refs = std::max<long>(max_live_refs, (long)live.size());
  }
  Obj* make(Obj::Kind k) { heap.emplace_back(new Obj()); heap.back()->kind = k; return heap.back().get(); }
  void raise(const char* cls, const std::string& msg) { pending = true; exc_class = cls; exc_msg = msg; }
};

Mock* M(JNIEnv* e) { return reinterpret_cast<Mock*>(e); }
Obj* O(jobject o) { return reinterpret_cast<Obj*>(o); }

void unimplemented() {
  fprintf(stderr, "mock_jni: the shim called a JNI function it does not declare\n");
  abort();
}

jclass m_FindClass(JNIEnv* e, const char* name) {
  M(e)->check_no_pending("FindClass");
  Obj* c = M(e)->make(Obj::CLASS);
  c->name = name;
  M(e)->hand_out(c);
  return reinterpret_cast<jclass>(c);
}
jint m_ThrowNew(JNIEnv* e, jclass c, const char* msg) {
  M(e)->check_no_pending("ThrowNew");
  M(e)->raise(O(c)->name.c_str(), msg ? msg : "");
  return 0;
}
void m_ExceptionClear(JNIEnv* e) { M(e)->pending = false; }
jboolean m_ExceptionCheck(JNIEnv* e) { return M(e)->pending ? JNI_TRUE : JNI_FALSE; }
void m_DeleteLocalRef(JNIEnv* e, jobject o) {  // allowed with an exception pending
  M(e)->refs_deleted++;
  auto it = M(e)->live.find(o);
  if (it == M(e)->live.end()) M(e)->violation("DeleteLocalRef of a reference that is not live");
  else M(e)->live.erase(it);
}
jfieldID m_GetFieldID(JNIEnv* e, jclass c, const char* name, const char* sig) {
  M(e)->check_no_pending("GetFieldID");
  if (strcmp(sig, "[B") != 0 || !O(c)->class_fields.count(name)) {
    M(e)->raise("java/lang/NoSuchFieldError", name);
    return nullptr;
  }
  auto it = M(e)->field_names.insert(name).first;
  return reinterpret_cast<jfieldID>(const_cast<std::string*>(&*it));
}
jobject m_GetObjectField(JNIEnv* e, jobject o, jfieldID f) {
  M(e)->check_no_pending("GetObjectField");
  const std::string& name = *reinterpret_cast<std::string*>(f);
  auto it = O(o)->fields.find(name);
  if (it == O(o)->fields.end() || !it->second) return nullptr;
  M(e)->hand_out(it->second);
  return reinterpret_cast<jobject>(it->second);
}
jsize m_GetArrayLength(JNIEnv* e, jarray a) {
  M(e)->check_no_pending("GetArrayLength");
  Obj* o = O(a);
  return (jsize)(o->kind == Obj::BYTES ? o->bytes.size() : o->kind == Obj::DOUBLES ? o->doubles.size()
                 : o->kind == Obj::LONGS ? o->longs.size() : o->kind == Obj::INTS ? o->ints.size() : o->elems.size());
}
jobject m_GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) {
  M(e)->check_no_pending("GetObjectArrayElement");
  Obj* o = O(a);
  if (i < 0 || (size_t)i >= o->elems.size()) { M(e)->raise("java/lang/ArrayIndexOutOfBoundsException", "element"); return nullptr; }
  if (o->elems[i]) M(e)->hand_out(o->elems[i]);
  return reinterpret_cast<jobject>(o->elems[i]);
}
void m_GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize start, jsize len, jbyte* buf) {
  M(e)->check_no_pending("GetByteArrayRegion");
  Obj* o = O(a);
  if (start < 0 || len < 0 || (size_t)start + (size_t)len > o->bytes.size()) {
    M(e)->raise("java/lang/ArrayIndexOutOfBoundsException", "byte region");
    return;
  }
  memcpy(buf, o->bytes.data() + start, (size_t)len);
}
void m_SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize start, jsize len, const jbyte* buf) {
  M(e)->check_no_pending("SetByteArrayRegion");
  Obj* o = O(a);
  if (start < 0 || len < 0 || (size_t)start + (size_t)len > o->bytes.size()) {
    M(e)->raise("java/lang/ArrayIndexOutOfBoundsException", "byte region");
    return;
  }
  memcpy(o->bytes.data() + start, buf, (size_t)len);
}
void m_SetIntArrayRegion(JNIEnv* e, jintArray a, jsize start, jsize len, const jint* buf) {
  M(e)->check_no_pending("SetIntArrayRegion");
  Obj* o = O(a);
  if (start < 0 || len < 0 || (size_t)start + (size_t)len > o->ints.size()) {
    M(e)->raise("java/lang/ArrayIndexOutOfBoundsException", "int region");
    return;
  }
  memcpy(o->ints.data() + start, buf, sizeof(int32_t) * (size_t)len);
}
void m_SetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize start, jsize len, const jdouble* buf) {
  M(e)->check_no_pending("SetDoubleArrayRegion");
  Obj* o = O(a);
  if (start < 0 || len < 0 || (size_t)start + (size_t)len > o->doubles.size()) {
    M(e)->raise("java/lang/ArrayIndexOutOfBoundsException", "double region");
    return;
  }
  memcpy(o->doubles.data() + start, buf, sizeof(double) * (size_t)len);
}

jdoubleArray m_NewDoubleArray(JNIEnv* e, jsize len) {
  M(e)->check_no_pending("NewDoubleArray");
  Obj* o = M(e)->make(Obj::DOUBLES);
  o->doubles.assign((size_t)len, 0.0);
  M(e)->hand_out(o);
  return reinterpret_cast<jdoubleArray>(o);
}
void m_GetLongArrayRegion(JNIEnv* e, jlongArray a, jsize start, jsize len, jlong* buf) {
  M(e)->check_no_pending("GetLongArrayRegion");
  Obj* o = O(a);
  if (start < 0 || len < 0 || (size_t)start + (size_t)len > o->longs.size()) {
    M(e)->raise("java/lang/ArrayIndexOutOfBoundsException", "long region");
    return;
  }
  memcpy(buf, o->longs.data() + start, sizeof(int64_t) * (size_t)len);
}

void install_table(Mock& m);

Obj* bytes_obj(Mock& m, const uint8_t* p, int64_t n) {
  Obj* o = m.make(Obj::BYTES);
  o->bytes.assign(reinterpret_cast<const int8_t*>(p), reinterpret_cast<const int8_t*>(p) + n);
  return o;
}

void install_table(Mock& m) {
  for (auto& s : m.table.slot) s = reinterpret_cast<void*>(&unimplemented);
  m.table.slot[kJniSlotFindClass] = (void*)&m_FindClass;
  m.table.slot[kJniSlotThrowNew] = (void*)&m_ThrowNew;
  m.table.slot[kJniSlotExceptionClear] = (void*)&m_ExceptionClear;
  m.table.slot[kJniSlotExceptionCheck] = (void*)&m_ExceptionCheck;
  m.table.slot[kJniSlotDeleteLocalRef] = (void*)&m_DeleteLocalRef;
  m.table.slot[kJniSlotGetFieldID] = (void*)&m_GetFieldID;
  m.table.slot[kJniSlotGetObjectField] = (void*)&m_GetObjectField;
  m.table.slot[kJniSlotGetArrayLength] = (void*)&m_GetArrayLength;
  m.table.slot[kJniSlotGetObjectArrayElement] = (void*)&m_GetObjectArrayElement;
  m.table.slot[kJniSlotGetByteArrayRegion] = (void*)&m_GetByteArrayRegion;
  m.table.slot[kJniSlotSetByteArrayRegion] = (void*)&m_SetByteArrayRegion;
  m.table.slot[kJniSlotSetIntArrayRegion] = (void*)&m_SetIntArrayRegion;
  m.table.slot[kJniSlotSetDoubleArrayRegion] = (void*)&m_SetDoubleArrayRegion;
  m.table.slot[kJniSlotNewDoubleArray] = (void*)&m_NewDoubleArray;
  m.table.slot[kJniSlotGetLongArrayRegion] = (void*)&m_GetLongArrayRegion;
  m.env.functions = &m.table;
}

typedef void (*init_fn)(JNIEnv*, jclass, jclass, jclass, jboolean, jint);
typedef void (*compute_fn)(JNIEnv*, jobject, jobjectArray, jobjectArray, jdoubleArray);
typedef void (*done_fn)(JNIEnv*, jobject);

}  // namespace

extern "C" {

enum {
  MOCK_DROP_GCP_FIELD = 1,    // ReadDataHolder class lacks overallGCP -> initNative must throw IAE
  MOCK_NULL_READQUALS = 2,    // read 0 has readQuals == null
  MOCK_SKIP_INIT = 4,         // call compute without initNative
  MOCK_SHORT_QUALS = 8,       // read 0's insertionGOP is one byte short
  MOCK_NULL_READ_ELEMENT = 16, // readDataArray[0] == null
  MOCK_COMPUTE_AFTER_DONE = 32, // initNative, doneNative, THEN computeLikelihoodsNative (the reference keeps working)
  MOCK_REINIT_TWICE = 64        // initNative again (same arguments, then the other precision and back) before computing
};

// Returns 0 = ran without a Java exception, 1 = exception pending after initNative,
// 2 = after computeLikelihoodsNative, -1 = could not load / resolve the library.
// counters: [0] local refs handed out, [1] DeleteLocalRef calls, [2] -Xcheck:jni-style violations, [3] most local
// references live at once.  The first violation's text replaces the exception message when there is no exception.
int mockjni_run(const char* lib_path, int use_double, int max_threads, int n_reads, int n_haps,
                const int64_t* read_off, const int64_t* hap_off, const uint8_t* rb, const uint8_t* rq,
                const uint8_t* ri, const uint8_t* rd, const uint8_t* rc, const uint8_t* hb, double* out,
                int out_len, int flags, char* exc_class, char* exc_msg, long* counters) {
  exc_class[0] = exc_msg[0] = 0;
  void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { snprintf(exc_msg, 512, "dlopen: %s", dlerror()); return -1; }
  init_fn f_init = (init_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative");
  compute_fn f_compute = (compute_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative");
  done_fn f_done = (done_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative");
  if (!f_init || !f_compute || !f_done) { snprintf(exc_msg, 512, "missing JNI symbol"); return -1; }

  Mock m;
  install_table(m);
  JNIEnv* env = &m.env;

  Obj* read_cls = m.make(Obj::CLASS);
  read_cls->name = "org/broadinstitute/gatk/nativebindings/pairhmm/ReadDataHolder";
  read_cls->class_fields = {"readBases", "readQuals", "insertionGOP", "deletionGOP", "overallGCP"};
  if (flags & MOCK_DROP_GCP_FIELD) read_cls->class_fields.erase("overallGCP");
  Obj* hap_cls = m.make(Obj::CLASS);
  hap_cls->name = "org/broadinstitute/gatk/nativebindings/pairhmm/HaplotypeDataHolder";
  hap_cls->class_fields = {"haplotypeBases"};

  Obj* reads = m.make(Obj::OBJARRAY);
  for (int r = 0; r < n_reads; r++) {
    const int64_t a = read_off[r], n = read_off[r + 1] - a;
    Obj* holder = m.make(Obj::HOLDER);
    holder->fields["readBases"] = bytes_obj(m, rb + a, n);
    holder->fields["readQuals"] = (r == 0 && (flags & MOCK_NULL_READQUALS)) ? nullptr : bytes_obj(m, rq + a, n);
    holder->fields["insertionGOP"] = bytes_obj(m, ri + a, (r == 0 && (flags & MOCK_SHORT_QUALS)) ? n - 1 : n);
    holder->fields["deletionGOP"] = bytes_obj(m, rd + a, n);
    holder->fields["overallGCP"] = bytes_obj(m, rc + a, n);
    reads->elems.push_back((r == 0 && (flags & MOCK_NULL_READ_ELEMENT)) ? nullptr : holder);
  }
  Obj* haps = m.make(Obj::OBJARRAY);
  for (int k = 0; k < n_haps; k++) {
    Obj* holder = m.make(Obj::HOLDER);
    holder->fields["haplotypeBases"] = bytes_obj(m, hb + hap_off[k], hap_off[k + 1] - hap_off[k]);
    haps->elems.push_back(holder);
  }
  Obj* likelihoods = m.make(Obj::DOUBLES);
  likelihoods->doubles.assign((size_t)out_len, -12345.0);

  int rc_ = 0;
  if (!(flags & MOCK_SKIP_INIT)) {
    f_init(env, nullptr, reinterpret_cast<jclass>(read_cls), reinterpret_cast<jclass>(hap_cls),
           use_double ? JNI_TRUE : JNI_FALSE, max_threads);
    if (m.pending) rc_ = 1;
    if (rc_ == 0 && (flags & MOCK_REINIT_TWICE)) {
      f_init(env, nullptr, reinterpret_cast<jclass>(read_cls), reinterpret_cast<jclass>(hap_cls), use_double ? JNI_TRUE : JNI_FALSE, max_threads);
      f_init(env, nullptr, reinterpret_cast<jclass>(read_cls), reinterpret_cast<jclass>(hap_cls), use_double ? JNI_FALSE : JNI_TRUE, max_threads);
      f_init(env, nullptr, reinterpret_cast<jclass>(read_cls), reinterpret_cast<jclass>(hap_cls), use_double ? JNI_TRUE : JNI_FALSE, max_threads);
      if (m.pending) rc_ = 1;
    }
  }
  if (rc_ == 0 && (flags & MOCK_COMPUTE_AFTER_DONE)) f_done(env, nullptr);
  if (rc_ == 0) {
    f_compute(env, nullptr, reinterpret_cast<jobjectArray>(reads), reinterpret_cast<jobjectArray>(haps),
              reinterpret_cast<jdoubleArray>(likelihoods));
    if (m.pending) rc_ = 2;
  }
  // (a JVM would have the exception pending on return to Java; doneNative takes no JNI calls)
  f_done(env, nullptr);
  memcpy(out, likelihoods->doubles.data(), sizeof(double) * (size_t)out_len);
  if (m.pending) {
    snprintf(exc_class, 256, "%s", m.exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m.exc_msg.c_str());
  } else if (m.violations) {
    snprintf(exc_msg, 512, "%s", m.first_violation.c_str());
  }
  if (counters) { counters[0] = m.refs_created; counters[1] = m.refs_deleted; counters[2] = m.violations; counters[3] = m.max_live_refs; }
  return rc_;
}


int g_warm_iters = 0;             // mockjni_set_warm_iters: untimed calls per thread in front of mockjni_run_concurrent's timed part
int64_t g_last_timing[6] = {0};   // the shim's call-time split over the timed part of the last mockjni_run_concurrent
void mockjni_set_warm_iters(int n) { g_warm_iters = n < 0 ? 0 : n; }
void mockjni_last_timing(int64_t* out) { for (int i = 0; i < 6; i++) out[i] = g_last_timing[i]; }

// Concurrent callers (GATK Spark): one initNative, then `n_threads` threads, each with its own JNIEnv,
// call computeLikelihoodsNative `iters` times on its own contiguous slice of the reads (all haplotypes),
// then one doneNative.  out = the whole batch's likelihoods, read-major.  Returns 0, or 1/2 like
// mockjni_run (first failing thread's exception), -1 on load errors.  wall_ms = time of the threaded part.
int mockjni_run_concurrent(const char* lib_path, int use_double, int max_threads, int n_threads, int iters,
                           int n_reads, int n_haps, const int64_t* read_off, const int64_t* hap_off,
                           const uint8_t* rb, const uint8_t* rq, const uint8_t* ri, const uint8_t* rd,
                           const uint8_t* rc, const uint8_t* hb, double* out, char* exc_class, char* exc_msg,
                           double* wall_ms) {
  exc_class[0] = exc_msg[0] = 0;
  void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { snprintf(exc_msg, 512, "dlopen: %s", dlerror()); return -1; }
  init_fn f_init = (init_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative");
  compute_fn f_compute = (compute_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative");
  done_fn f_done = (done_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative");
  if (!f_init || !f_compute || !f_done) { snprintf(exc_msg, 512, "missing JNI symbol"); return -1; }

  Mock m;
  install_table(m);
  Obj* read_cls = m.make(Obj::CLASS);
  read_cls->class_fields = {"readBases", "readQuals", "insertionGOP", "deletionGOP", "overallGCP"};
  Obj* hap_cls = m.make(Obj::CLASS);
  hap_cls->class_fields = {"haplotypeBases"};
  std::vector<Obj*> read_holders;
  for (int r = 0; r < n_reads; r++) {
    const int64_t a = read_off[r], n = read_off[r + 1] - a;
    Obj* holder = m.make(Obj::HOLDER);
    holder->fields["readBases"] = bytes_obj(m, rb + a, n);
    holder->fields["readQuals"] = bytes_obj(m, rq + a, n);
    holder->fields["insertionGOP"] = bytes_obj(m, ri + a, n);
    holder->fields["deletionGOP"] = bytes_obj(m, rd + a, n);
    holder->fields["overallGCP"] = bytes_obj(m, rc + a, n);
    read_holders.push_back(holder);
  }
  Obj* haps = m.make(Obj::OBJARRAY);
  for (int k = 0; k < n_haps; k++) {
    Obj* holder = m.make(Obj::HOLDER);
    holder->fields["haplotypeBases"] = bytes_obj(m, hb + hap_off[k], hap_off[k + 1] - hap_off[k]);
    haps->elems.push_back(holder);
  }
  f_init(&m.env, nullptr, reinterpret_cast<jclass>(read_cls), reinterpret_cast<jclass>(hap_cls),
         use_double ? JNI_TRUE : JNI_FALSE, max_threads);
  if (m.pending) {
    snprintf(exc_class, 256, "%s", m.exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m.exc_msg.c_str());
    return 1;
  }
  std::vector<std::unique_ptr<Mock>> envs;
  std::vector<Obj*> slices, results;
  std::vector<int> first(n_threads + 1, 0);
  for (int t = 0; t < n_threads; t++) {
    envs.emplace_back(new Mock());
    install_table(*envs.back());
    first[t + 1] = (int)((int64_t)n_reads * (t + 1) / n_threads);
    Obj* arr = envs.back()->make(Obj::OBJARRAY);
    for (int r = first[t]; r < first[t + 1]; r++) arr->elems.push_back(read_holders[r]);
    Obj* res = envs.back()->make(Obj::DOUBLES);
    res->doubles.assign((size_t)(first[t + 1] - first[t]) * n_haps, -12345.0);
    slices.push_back(arr);
    results.push_back(res);
  }
  // g_warm_iters untimed calls per thread first (slots, engines and arenas exist afterwards), then all threads start
  // the timed part together; the shim's call-time split (gkl_pairhmm_jni_timing) is reset at that point
  typedef void (*timing_fn)(int64_t*, int);
  timing_fn f_timing = (timing_fn)dlsym(h, "gkl_pairhmm_jni_timing");
  std::atomic<int> warmed{0};
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  std::mutex t0_mu;
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; t++)
    pool.emplace_back([&, t] {
      for (int k = 0; k < g_warm_iters && !envs[t]->pending; k++)
        f_compute(&envs[t]->env, nullptr, reinterpret_cast<jobjectArray>(slices[t]),
                  reinterpret_cast<jobjectArray>(haps), reinterpret_cast<jdoubleArray>(results[t]));
      if (warmed.fetch_add(1) + 1 == n_threads) {
        std::lock_guard<std::mutex> l(t0_mu);
        int64_t scratch[6];
        if (f_timing) f_timing(scratch, 1);
        t0 = std::chrono::steady_clock::now();
        warmed.fetch_add(n_threads);  // release
      }
      while (warmed.load() < 2 * n_threads) std::this_thread::yield();
      for (int k = 0; k < iters && !envs[t]->pending; k++)
        f_compute(&envs[t]->env, nullptr, reinterpret_cast<jobjectArray>(slices[t]),
                  reinterpret_cast<jobjectArray>(haps), reinterpret_cast<jdoubleArray>(results[t]));
    });
  // MOCKJNI_CHURN=1: meanwhile another IntelPairHmm instance of the same JVM comes and goes -- initNative with the same
  // arguments and doneNative, over and over (the reference's initNative only re-sets globals, its doneNative is empty)
  std::atomic<bool> stop{false};
  std::thread churn;
  Mock churn_env;
  install_table(churn_env);
  const char* ch = getenv("MOCKJNI_CHURN");
  if (ch && *ch == '1')
    churn = std::thread([&] {
      while (!stop.load()) {
        f_done(&churn_env.env, nullptr);
        f_init(&churn_env.env, nullptr, reinterpret_cast<jclass>(read_cls), reinterpret_cast<jclass>(hap_cls),
               use_double ? JNI_TRUE : JNI_FALSE, max_threads);
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    });
  for (auto& th : pool) th.join();
  stop.store(true);
  if (churn.joinable()) churn.join();
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (f_timing) f_timing(g_last_timing, 0);
  if (churn_env.pending) { envs[0]->pending = true; envs[0]->exc_class = churn_env.exc_class; envs[0]->exc_msg = "churn thread: " + churn_env.exc_msg; }
  f_done(&m.env, nullptr);
  int rc_ = 0;
  for (int t = 0; t < n_threads; t++) {
    memcpy(out + (size_t)first[t] * n_haps, results[t]->doubles.data(), sizeof(double) * results[t]->doubles.size());
    if (envs[t]->pending && rc_ == 0) {
      rc_ = 2;
      snprintf(exc_class, 256, "%s", envs[t]->exc_class.c_str());
      snprintf(exc_msg, 512, "%s", envs[t]->exc_msg.c_str());
    }
  }
  return rc_;
}


// ---- PDHMM: IntelPDHMM natives (include/gkl_pdhmm_jni.h) ----
typedef void (*pd_init_fn)(JNIEnv*, jclass, jclass, jclass, jint, jint, jint, jint);
typedef jdoubleArray (*pd_flat_fn)(JNIEnv*, jobject, jbyteArray, jbyteArray, jbyteArray, jbyteArray, jbyteArray, jbyteArray,
                                   jbyteArray, jlongArray, jlongArray, jint, jint, jint);
typedef void (*pd_done_fn)(JNIEnv*, jclass);

enum { MOCKPD_DROP_PDBASES_FIELD = 1, MOCKPD_SKIP_INIT = 2, MOCKPD_HOLDERS = 4 };

// Padded 1:1 batch (IntelPDHMM.computePDHMM layout). With MOCKPD_HOLDERS the same data is instead
// presented as reads x haplotypes holder arrays (n_reads = batch of distinct reads, n_haps haplotypes,
// flat arrays then hold [n_reads][max_read] and [n_haps][max_hap]) and computeLikelihoodsNative is driven.
int mockjni_run_pdhmm(const char* lib_path, int n_a, int n_b, int max_hap, int max_read, const uint8_t* hb,
                      const uint8_t* hp, const uint8_t* rb, const uint8_t* rq, const uint8_t* ri, const uint8_t* rd,
                      const uint8_t* rc, const int64_t* hap_len, const int64_t* read_len, double* out, int out_len,
                      int flags, int max_memory_mb, char* exc_class, char* exc_msg) {
  exc_class[0] = exc_msg[0] = 0;
  void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { snprintf(exc_msg, 512, "dlopen: %s", dlerror()); return -1; }
  pd_init_fn f_init = (pd_init_fn)dlsym(h, "Java_com_intel_gkl_pdhmm_IntelPDHMM_initNative");
  pd_flat_fn f_flat = (pd_flat_fn)dlsym(h, "Java_com_intel_gkl_pdhmm_IntelPDHMM_computePDHMMNative");
  compute_fn f_cl = (compute_fn)dlsym(h, "Java_com_intel_gkl_pdhmm_IntelPDHMM_computeLikelihoodsNative");
  pd_done_fn f_done = (pd_done_fn)dlsym(h, "Java_com_intel_gkl_pdhmm_IntelPDHMM_doneNative");
  if (!f_init || !f_flat || !f_cl || !f_done) { snprintf(exc_msg, 512, "missing JNI symbol"); return -1; }
  Mock m;
  install_table(m);
  JNIEnv* env = &m.env;
  Obj* read_cls = m.make(Obj::CLASS);
  read_cls->class_fields = {"readBases", "readQuals", "insertionGOP", "deletionGOP", "overallGCP"};
  Obj* hap_cls = m.make(Obj::CLASS);
  hap_cls->class_fields = {"haplotypeBases", "haplotypePDBases"};
  if (flags & MOCKPD_DROP_PDBASES_FIELD) hap_cls->class_fields.erase("haplotypePDBases");
  int rc_ = 0;
  if (!(flags & MOCKPD_SKIP_INIT)) {
    f_init(env, nullptr, reinterpret_cast<jclass>(read_cls), reinterpret_cast<jclass>(hap_cls), 0, 1, 0, max_memory_mb);
    if (m.pending) rc_ = 1;
  }
  if (rc_ == 0 && (flags & MOCKPD_HOLDERS)) {
    const int n_reads = n_a, n_haps = n_b;
    Obj* reads = m.make(Obj::OBJARRAY);
    for (int r = 0; r < n_reads; r++) {
      Obj* holder = m.make(Obj::HOLDER);
      const int64_t n = read_len[r];
      holder->fields["readBases"] = bytes_obj(m, rb + (int64_t)r * max_read, n);
      holder->fields["readQuals"] = bytes_obj(m, rq + (int64_t)r * max_read, n);
      holder->fields["insertionGOP"] = bytes_obj(m, ri + (int64_t)r * max_read, n);
      holder->fields["deletionGOP"] = bytes_obj(m, rd + (int64_t)r * max_read, n);
      holder->fields["overallGCP"] = bytes_obj(m, rc + (int64_t)r * max_read, n);
      reads->elems.push_back(holder);
    }
    Obj* haps = m.make(Obj::OBJARRAY);
    for (int k = 0; k < n_haps; k++) {
      Obj* holder = m.make(Obj::HOLDER);
      holder->fields["haplotypeBases"] = bytes_obj(m, hb + (int64_t)k * max_hap, hap_len[k]);
      holder->fields["haplotypePDBases"] = bytes_obj(m, hp + (int64_t)k * max_hap, hap_len[k]);
      haps->elems.push_back(holder);
    }
    Obj* lik = m.make(Obj::DOUBLES);
    lik->doubles.assign((size_t)out_len, -12345.0);
    f_cl(env, nullptr, reinterpret_cast<jobjectArray>(reads), reinterpret_cast<jobjectArray>(haps),
         reinterpret_cast<jdoubleArray>(lik));
    if (m.pending) rc_ = 2;
    memcpy(out, lik->doubles.data(), sizeof(double) * (size_t)out_len);
  } else if (rc_ == 0) {
    const int batch = n_a;
    Obj* arrs[7];
    const uint8_t* src[7] = {hb, hp, rb, rq, ri, rd, rc};
    for (int i = 0; i < 7; i++) arrs[i] = bytes_obj(m, src[i], (int64_t)batch * (i < 2 ? max_hap : max_read));
    Obj* hl = m.make(Obj::LONGS); hl->longs.assign(hap_len, hap_len + batch);
    Obj* rl = m.make(Obj::LONGS); rl->longs.assign(read_len, read_len + batch);
    jdoubleArray res = f_flat(env, nullptr, (jbyteArray)arrs[0], (jbyteArray)arrs[1], (jbyteArray)arrs[2], (jbyteArray)arrs[3],
                              (jbyteArray)arrs[4], (jbyteArray)arrs[5], (jbyteArray)arrs[6], (jlongArray)hl, (jlongArray)rl,
                              batch, max_hap, max_read);
    if (m.pending) rc_ = 2;
    else if (!res) rc_ = 3;
    else memcpy(out, O(res)->doubles.data(), sizeof(double) * (size_t)std::min<size_t>(out_len, O(res)->doubles.size()));
  }
  f_done(env, nullptr);
  if (m.pending) { snprintf(exc_class, 256, "%s", m.exc_class.c_str()); snprintf(exc_msg, 512, "%s", m.exc_msg.c_str()); }
  return rc_;
}

// ---- Smith-Waterman: IntelSmithWaterman natives (include/gkl_sw_jni.h) ----
typedef void (*sw_init_fn)(JNIEnv*, jclass);
typedef jint (*sw_align_fn)(JNIEnv*, jclass, jbyteArray, jbyteArray, jbyteArray, jint, jint, jint, jint, jbyte);
typedef void (*sw_done_fn)(JNIEnv*, jclass);
enum { SW_SKIP_INIT = 1, SW_NULL_REF = 2 };

// initNative -> `iters` x alignNative -> doneNative.  cigar_out[cigar_len] = the Java byte[] after the call.
// Returns 0, 1 (exception after initNative), 2 (after alignNative), -1 (load error); *offset = alignNative's value.
int mockjni_run_sw(const char* lib_path, const uint8_t* ref, int ref_len, const uint8_t* alt, int alt_len, int cigar_len,
                   int match, int mismatch, int open, int extend, int strategy, int flags, int iters, uint8_t* cigar_out,
                   int* offset, char* exc_class, char* exc_msg, double* wall_ms) {
  exc_class[0] = exc_msg[0] = 0;
  void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { snprintf(exc_msg, 512, "dlopen: %s", dlerror()); return -1; }
  sw_init_fn f_init = (sw_init_fn)dlsym(h, "Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_initNative");
  sw_align_fn f_align = (sw_align_fn)dlsym(h, "Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_alignNative");
  sw_done_fn f_done = (sw_done_fn)dlsym(h, "Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_doneNative");
  if (!f_init || !f_align || !f_done) { snprintf(exc_msg, 512, "missing JNI symbol"); return -1; }
  Mock m;
  install_table(m);
  JNIEnv* env = &m.env;
  Obj* jref = bytes_obj(m, ref, ref_len);
  Obj* jalt = bytes_obj(m, alt, alt_len);
  Obj* jcig = m.make(Obj::BYTES);
  jcig->bytes.assign((size_t)cigar_len, 0);
  int rc_ = 0;
  if (!(flags & SW_SKIP_INIT)) {
    f_init(env, nullptr);
    if (m.pending) rc_ = 1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < iters && rc_ == 0; k++) {
    std::fill(jcig->bytes.begin(), jcig->bytes.end(), 0);  // the Java wrapper allocates a fresh array per call
    *offset = f_align(env, nullptr, (flags & SW_NULL_REF) ? nullptr : reinterpret_cast<jbyteArray>(jref),
                      reinterpret_cast<jbyteArray>(jalt), reinterpret_cast<jbyteArray>(jcig), match, mismatch, open,
                      extend, (jbyte)strategy);
    if (m.pending) rc_ = 2;
  }
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  f_done(env, nullptr);
  memcpy(cigar_out, jcig->bytes.data(), (size_t)cigar_len);
  if (m.pending) {
    snprintf(exc_class, 256, "%s", m.exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m.exc_msg.c_str());
  }
  return rc_;
}


typedef jint (*sw_batch_fn)(JNIEnv*, jclass, jbyteArray, jlongArray, jbyteArray, jlongArray, jbyteArray, jint, jintArray, jint,
                            jint, jint, jint, jbyte);
// initNative -> alignBatchNative -> doneNative.  cigars_out[n * stride], offsets_out[n].  Return codes as mockjni_run_sw.
int mockjni_run_sw_batch(const char* lib_path, int n, const uint8_t* refs, const int64_t* ref_off, const uint8_t* alts,
                         const int64_t* alt_off, int stride, int match, int mismatch, int open, int extend, int strategy,
                         uint8_t* cigars_out, int32_t* offsets_out, int* returned, char* exc_class, char* exc_msg) {
  exc_class[0] = exc_msg[0] = 0;
  void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { snprintf(exc_msg, 512, "dlopen: %s", dlerror()); return -1; }
  sw_init_fn f_init = (sw_init_fn)dlsym(h, "Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_initNative");
  sw_batch_fn f_batch = (sw_batch_fn)dlsym(h, "Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_alignBatchNative");
  sw_done_fn f_done = (sw_done_fn)dlsym(h, "Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_doneNative");
  if (!f_init || !f_batch || !f_done) { snprintf(exc_msg, 512, "missing JNI symbol"); return -1; }
  Mock m;
  install_table(m);
  JNIEnv* env = &m.env;
  Obj* jrefs = bytes_obj(m, refs, ref_off[n]);
  Obj* jalts = bytes_obj(m, alts, alt_off[n]);
  Obj* jro = m.make(Obj::LONGS); jro->longs.assign(ref_off, ref_off + n + 1);
  Obj* jao = m.make(Obj::LONGS); jao->longs.assign(alt_off, alt_off + n + 1);
  Obj* jcig = m.make(Obj::BYTES); jcig->bytes.assign((size_t)n * stride, 0);
  Obj* joff = m.make(Obj::INTS); joff->ints.assign((size_t)n, -777);
  int rc_ = 0;
  f_init(env, nullptr);
  if (m.pending) rc_ = 1;
  if (rc_ == 0) {
    *returned = f_batch(env, nullptr, reinterpret_cast<jbyteArray>(jrefs), reinterpret_cast<jlongArray>(jro),
                        reinterpret_cast<jbyteArray>(jalts), reinterpret_cast<jlongArray>(jao),
                        reinterpret_cast<jbyteArray>(jcig), stride, reinterpret_cast<jintArray>(joff), match, mismatch,
                        open, extend, (jbyte)strategy);
    if (m.pending) rc_ = 2;
  }
  f_done(env, nullptr);
  memcpy(cigars_out, jcig->bytes.data(), (size_t)n * stride);
  memcpy(offsets_out, joff->ints.data(), sizeof(int32_t) * (size_t)n);
  if (m.pending) {
    snprintf(exc_class, 256, "%s", m.exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m.exc_msg.c_str());
  }
  return rc_;
}


// ---- libgkl_utils.so (include/gkl_utils_jni.h): six natives, no JNIEnv use at all ----
// out[0..5] = getFlushToZero(before), isAvx, isAvx2, isAvx512, ompThreads, getFlushToZero(after set true)
int mockjni_run_utils(const char* lib_path, int* out) {
  void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return -1;
  typedef jboolean (*bfn)(JNIEnv*, jobject);
  typedef void (*sfn)(JNIEnv*, jobject, jboolean);
  typedef jint (*ifn)(JNIEnv*, jobject);
  bfn get = (bfn)dlsym(h, "Java_com_intel_gkl_IntelGKLUtils_getFlushToZeroNative");
  sfn set = (sfn)dlsym(h, "Java_com_intel_gkl_IntelGKLUtils_setFlushToZeroNative");
  bfn avx = (bfn)dlsym(h, "Java_com_intel_gkl_IntelGKLUtils_isAvxSupportedNative");
  bfn avx2 = (bfn)dlsym(h, "Java_com_intel_gkl_IntelGKLUtils_isAvx2SupportedNative");
  bfn avx512 = (bfn)dlsym(h, "Java_com_intel_gkl_IntelGKLUtils_isAvx512SupportedNative");
  ifn omp = (ifn)dlsym(h, "Java_com_intel_gkl_IntelGKLUtils_getAvailableOmpThreadsNative");
  if (!get || !set || !avx || !avx2 || !avx512 || !omp) return -2;
  Mock m;
  install_table(m);
  JNIEnv* env = &m.env;
  out[0] = get(env, nullptr);
  out[1] = avx(env, nullptr); out[2] = avx2(env, nullptr); out[3] = avx512(env, nullptr);
  out[4] = omp(env, nullptr);
  set(env, nullptr, JNI_TRUE);
  out[5] = get(env, nullptr);
  set(env, nullptr, out[0] ? JNI_TRUE : JNI_FALSE);
  return 0;
}

}  // extern "C"
