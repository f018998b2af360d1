// JNI drop-in layer of libgkl_pairhmm.so: the three natives of
// com.intel.gkl.pairhmm.IntelPairHmm (include/gkl_pairhmm_jni.h) as thin shims over the
// C ABI.  Replaces the reference's L2 (src/main/native/pairhmm/IntelPairHmm.cc +
// JavaData.h); differences, all deliberate:
//   * byte[] fields are COPIED with GetByteArrayRegion into one flat batch instead of
//     being pinned one by one (50k pins at 10k reads) and described by a 56-byte
//     testcase per PAIR (JavaData.h:94-110); local refs are deleted as we go
//     (the reference leaks them until return, JavaData.h:135-145);
//   * the flat batch lives in page-locked arenas that are kept across calls (gklhip_host_alloc), so
//     the host-to-device copies are plain DMA, and every concurrent caller gets its own slot
//     (context + stream + arenas, up to GKL_HIP_SLOTS, default 4): one Java thread marshals or
//     finalises while another one's kernels run;
//   * a BIG call is pipelined: the reads are marshalled range by range on the calling thread (the only one that may
//     use its JNIEnv) while two engines of the slot compute the ranges already marshalled and the finished ranges
//     are written back to the Java array -- the 250 000 JNI calls of a 10k-read batch then hide behind the kernels
//     (the reference pins everything, computes, releases: JavaData.h:65-111, IntelPairHmm.cc:150-186);
//   * null holders / null byte[] fields / a too-short likelihood array raise
//     IllegalArgumentException instead of crashing the JVM;
//   * HIP failures raise java/lang/RuntimeException, allocation failures
//     java/lang/OutOfMemoryError (the reference's two classes, JavaData.h:130,140,150).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gkl_hip_pairhmm.h"
#include "../../include/gkl_pairhmm_jni.h"
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
inline void SetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize s, jsize l, const jdouble* b) { e->SetDoubleArrayRegion(a, s, l, b); }
}  // namespace gkljni
#endif

namespace {

constexpr const char* kIAE = "java/lang/IllegalArgumentException";
constexpr const char* kOOM = "java/lang/OutOfMemoryError";
constexpr const char* kRTE = "java/lang/RuntimeException";

// Page-locked byte arena (one per marshalled field and slot): grows with the biggest call, and an arena above 32 MB that the
// last 16 calls each filled to less than a quarter is given back (like the context's own buffers, pairhmm_api.hip: trim_due)
// -- one 1.28 M-pair call must not keep ~100 MB pinned per slot for the life of the JVM.  clear() runs at the start of a
// call, when nothing of the slot's previous call is in flight any more.
struct PinnedBytes {
  uint8_t* p = nullptr;
  size_t cap = 0, len = 0;
  int small_uses = 0;
  PinnedBytes() = default;
  PinnedBytes(const PinnedBytes&) = delete;
  PinnedBytes& operator=(const PinnedBytes&) = delete;
  ~PinnedBytes() { gklhip_host_free(p); }
  void clear() {
    if (cap > ((size_t)32 << 20) && len < cap / 4) {
      if (++small_uses >= 16) { gklhip_host_free(p); p = nullptr; cap = 0; small_uses = 0; }
    } else {
      small_uses = 0;
    }
    len = 0;
  }
  uint8_t* grow(size_t add) {  // returns where the new bytes go
    if (len + add > cap) {
      const size_t want = std::max<size_t>(2 * cap, std::max<size_t>(len + add, 1 << 16));
      uint8_t* q = static_cast<uint8_t*>(gklhip_host_alloc(want));
      if (!q) throw std::bad_alloc();
      if (len) memcpy(q, p, len);
      gklhip_host_free(p);
      p = q;
      cap = want;
    }
    uint8_t* at = p + len;
    len += add;
    return at;
  }
};

// The five read arrays of one read range (the whole batch in a small call).
struct ReadArena {
  PinnedBytes read_bases, read_quals, ins, del, gcp;
  std::vector<int64_t> read_off;
  void clear() { for (PinnedBytes* a : {&read_bases, &read_quals, &ins, &del, &gcp}) a->clear(); }
};

// Pipelined big calls: a compute thread bound to one engine of the slot takes read ranges from `todo`, runs
// gklhip_compute on them and reports on `done`.  The JNI side (marshalling, write-back, exceptions) stays on the
// calling thread.
struct RangeTask {
  int k = 0;
  gklhip_batch batch;
  double* out = nullptr;
  int status = GKLHIP_OK;
  std::string error;
};
struct Pipeline {
  std::mutex mu;
  std::condition_variable has_todo, has_done;
  std::deque<RangeTask*> todo, done;
  bool quit = false;
  std::vector<std::thread> threads;
  void start(gklhip_ctx* ctx) {
    threads.emplace_back([this, ctx] {
      std::unique_lock<std::mutex> l(mu);
      for (;;) {
        has_todo.wait(l, [&] { return quit || !todo.empty(); });
        if (todo.empty()) return;  // quit
        RangeTask* t = todo.front();
        todo.pop_front();
        l.unlock();
        // (gklhip_compute itself lets no exception out; the string below can still fail to allocate)
        try {
          t->status = gklhip_compute(ctx, &t->batch, t->out);
          if (t->status != GKLHIP_OK) { const char* d = gklhip_last_error(); t->error = d ? d : ""; }  // (thread-local detail)
        } catch (...) {
          t->status = GKLHIP_ERR_OOM;
        }
        l.lock();
        done.push_back(t);
        has_done.notify_all();
      }
    });
  }
  void submit(RangeTask* t) {
    { std::lock_guard<std::mutex> l(mu); todo.push_back(t); }
    has_todo.notify_one();
  }
  RangeTask* take_done(bool wait) {
    std::unique_lock<std::mutex> l(mu);
    if (wait) has_done.wait(l, [&] { return !done.empty(); });
    if (done.empty()) return nullptr;
    RangeTask* t = done.front();
    done.pop_front();
    return t;
  }
  ~Pipeline() {
    { std::lock_guard<std::mutex> l(mu); quit = true; }
    has_todo.notify_all();
    for (auto& th : threads) th.join();
  }
};

// Everything one call needs; a slot serves one caller at a time.
struct Slot {
  gklhip_ctx* ctx = nullptr;
  gklhip_ctx* ctx2 = nullptr;          // second engine of pipelined big calls (created by the first of them)
  std::unique_ptr<Pipeline> pipe;      // declared after the contexts: its threads are joined before they go
  PinnedBytes hap_bases;
  std::vector<int64_t> hap_off;
  ReadArena whole;                     // a small call's reads
  std::vector<std::unique_ptr<ReadArena>> ranges;  // a pipelined call's read ranges
  std::vector<RangeTask> tasks;
  std::vector<double> out;
  int calls_since_pipelined = 0;       // one-shot calls since the slot's last pipelined one: after 16 its range arenas (and a result vector above 32 MB) go
  bool busy = false;
  int gen = 0;  // configuration generation (initNative with other arguments starts a new one)
  gklhip_config cfg;  // what `ctx` was created with: the second engine of a pipelined call gets the same
  ~Slot() {
    pipe.reset();
    if (ctx2) gklhip_done(ctx2);
    if (ctx) gklhip_done(ctx);
  }
};

// Where a call's time goes, summed over the calls of the process (nanoseconds; gkl_pairhmm_jni_timing reads them):
// [0] marshalling on the calling thread, [1] waiting for compute that marshalling did not cover (a small call: the
// whole gklhip_compute), [2] write-back into the Java array, [3] whole calls, [4] number of calls, [5] pipelined calls.
std::atomic<int64_t> g_timing[6];
int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Process-wide state, like the reference's globals (IntelPairHmm.cc:41-48) and the
// static field IDs of JavaData (JavaData.h:160-176).
struct State {
  std::mutex mu;
  std::condition_variable slot_free;
  std::vector<std::unique_ptr<Slot>> slots;
  int max_slots = 4;
  int creating = 0;  // slots being initialised outside the lock
  bool ready = false;  // initNative has run: configuration and field IDs are valid (stays true after doneNative)
  int gen = 0;         // current configuration generation; slots of older generations die when they come back
  gklhip_config cfg;
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

// Append holder.<field> (a byte[]) to dst; returns its length, or -1 after throwing.
long append_field(JNIEnv* env, jobject holder, jfieldID fid, PinnedBytes& dst, long expect_at_least) {
  jbyteArray bytes = (jbyteArray)gkljni::GetObjectField(env, holder, fid);
  if (!bytes) {
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
      throw_java(env, kIAE, "read quality array shorter than readBases");
      return -1;
    }
    take = expect_at_least;
  }
  if (take > 0) gkljni::GetByteArrayRegion(env, bytes, 0, (jsize)take, reinterpret_cast<jbyte*>(dst.grow((size_t)take)));
  gkljni::DeleteLocalRef(env, bytes);
  if (gkljni::ExceptionCheck(env)) return -1;
  return take;
}

// holder = arr[i], or NULL after throwing.
jobject holder_at(JNIEnv* env, jobjectArray arr, jsize i) {
  jobject holder = gkljni::GetObjectArrayElement(env, arr, i);
  if (gkljni::ExceptionCheck(env)) return nullptr;
  if (!holder) throw_java(env, kIAE, "null element in data holder array");
  return holder;
}

// A free slot of the current configuration, creating one (context + stream) while fewer than max_slots exist;
// blocks otherwise.  Returns NULL after throwing.
Slot* acquire_slot(JNIEnv* env) {
  std::unique_lock<std::mutex> lock(g.mu);
  for (;;) {
    if (!g.ready) {
      lock.unlock();
      throw_java(env, kRTE, "GKL-HIP PairHMM: computeLikelihoodsNative before initNative");
      return nullptr;
    }
    for (auto& s : g.slots)
      if (!s->busy && s->gen == g.gen) { s->busy = true; return s.get(); }
    int live = g.creating;
    for (auto& s : g.slots) live += s->gen == g.gen;
    if (live < g.max_slots) {
      g.creating++;
      const gklhip_config cfg = g.cfg;
      const int gen = g.gen;
      lock.unlock();
      std::unique_ptr<Slot> s(new (std::nothrow) Slot());
      const int st = s ? gklhip_init(&cfg, &s->ctx) : GKLHIP_ERR_OOM;
      lock.lock();
      g.creating--;
      if (st != GKLHIP_OK) {
        // could not add a slot (e.g. out of device memory): share the existing ones instead
        int have = 0;
        for (auto& o : g.slots) have += o->gen == g.gen;
        g.max_slots = std::max<int>(1, have);
        if (have == 0) { lock.unlock(); g.slot_free.notify_all(); throw_status(env, st); return nullptr; }
        continue;
      }
      s->gen = gen;
      s->cfg = cfg;
      if (gen != g.gen) {  // re-configured meanwhile: the new slot (old arguments) is dropped -- outside the lock
        lock.unlock();
        s.reset();
        lock.lock();
        continue;
      }
      s->busy = true;
      g.slots.push_back(std::move(s));
      return g.slots.back().get();
    }
    g.slot_free.wait(lock);
  }
}

struct SlotLease {
  Slot* s;
  ~SlotLease() {
    if (!s) return;
    std::unique_ptr<Slot> dead;
    {
      std::lock_guard<std::mutex> lock(g.mu);
      s->busy = false;
      if (s->gen != g.gen)  // initNative changed the configuration while this call ran: the slot is not reused
        for (auto it = g.slots.begin(); it != g.slots.end(); ++it)
          if (it->get() == s) { dead = std::move(*it); g.slots.erase(it); break; }
    }
    g.slot_free.notify_all();  // waiters differ (a caller that wants a slot, any number of them): wake all
  }
};

}  // namespace

extern "C" {

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative(
    JNIEnv* env, jclass, jclass readDataHolder, jclass haplotypeDataHolder, jboolean use_double,
    jint max_threads) {
  std::unique_lock<std::mutex> lock(g.mu);
  struct { jfieldID* dst; jclass cls; const char* name; } fields[] = {
      {&g.readBases, readDataHolder, "readBases"},       {&g.readQuals, readDataHolder, "readQuals"},
      {&g.insertionGOP, readDataHolder, "insertionGOP"}, {&g.deletionGOP, readDataHolder, "deletionGOP"},
      {&g.overallGCP, readDataHolder, "overallGCP"},     {&g.haplotypeBases, haplotypeDataHolder, "haplotypeBases"}};
  for (auto& f : fields) {
    jfieldID id = f.cls ? gkljni::GetFieldID(env, f.cls, f.name, "[B") : nullptr;
    if (!id) {  // JavaData.h:127-133
      lock.unlock();
      throw_java(env, kIAE, "Unable to get field ID");
      return;
    }
    *f.dst = id;
  }
  // The reference's initNative only re-sets globals (IntelPairHmm.cc:70-116) and other threads may be inside
  // computeLikelihoodsNative right now: nothing is torn down here.  Same arguments: nothing to do.  Other
  // arguments: a new generation of slots; calls in flight finish on theirs, which then retire.
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
  const int max_slots = std::max(1, env_int("GKL_HIP_SLOTS", 4));
  bool have_slot = false;
  for (auto& sl : g.slots) have_slot |= sl->gen == g.gen;
  if (g.ready && memcmp(&cfg, &g.cfg, sizeof cfg) == 0 && have_slot) { g.max_slots = max_slots; return; }
  // The first slot is created here so that "no GPU" surfaces from initNative, like a failed dlopen would -- but NOT
  // under the lock: device and stream creation take milliseconds, and concurrent computeLikelihoodsNative callers
  // pass through g.mu when they take and return their slots.
  lock.unlock();
  std::unique_ptr<Slot> first(new (std::nothrow) Slot());
  const int st = first ? gklhip_init(&cfg, &first->ctx) : GKLHIP_ERR_OOM;
  if (st != GKLHIP_OK) { throw_status(env, st); return; }
  std::vector<std::unique_ptr<Slot>> dead;   // retired slots are destroyed after the lock is released
  lock.lock();
  g.cfg = cfg;
  g.max_slots = max_slots;
  first->cfg = cfg;
  first->gen = ++g.gen;   // (two racing initNative calls: the later one's generation wins, the other's slot retires like any old one)
  g.slots.push_back(std::move(first));
  g.ready = true;
  for (auto it = g.slots.begin(); it != g.slots.end();) {
    if (!(*it)->busy && (*it)->gen != g.gen) { dead.push_back(std::move(*it)); it = g.slots.erase(it); }
    else ++it;
  }
  lock.unlock();
  g.slot_free.notify_all();
}

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative(
    JNIEnv* env, jobject, jobjectArray readDataArray, jobjectArray haplotypeDataArray,
    jdoubleArray likelihoodArray) {
  if (!readDataArray || !haplotypeDataArray || !likelihoodArray) {
    throw_java(env, kIAE, "null argument");  // the Java wrapper already raised NPE (IntelPairHmm.java:134-136)
    return;
  }
  SlotLease lease{acquire_slot(env)};
  Slot* sl = lease.s;
  if (!sl) return;
  const int64_t t_call = now_ns();
  try {
    const jsize n_reads = gkljni::GetArrayLength(env, readDataArray);
    const jsize n_haps = gkljni::GetArrayLength(env, haplotypeDataArray);
    const int64_t n_pairs = (int64_t)n_reads * n_haps;
    if (n_pairs > 0x7fffffffLL) { throw_java(env, kIAE, "more than 2^31 read x haplotype pairs"); return; }
    if ((int64_t)gkljni::GetArrayLength(env, likelihoodArray) < n_pairs) {
      throw_java(env, kIAE, "likelihood array shorter than reads x haplotypes");
      return;
    }
    sl->hap_bases.clear();
    sl->hap_off.assign((size_t)n_haps + 1, 0);
    for (jsize h = 0; h < n_haps; h++) {
      jobject holder = holder_at(env, haplotypeDataArray, h);
      if (!holder) return;
      const long len = append_field(env, holder, g.haplotypeBases, sl->hap_bases, -1);
      gkljni::DeleteLocalRef(env, holder);
      if (len < 0) return;
      sl->hap_off[h + 1] = sl->hap_off[h] + len;
    }
    // reads [r0, r1) into `a` (offsets rebased to 0); false after throwing
    auto marshal_reads = [&](ReadArena& a, jsize r0, jsize r1) -> bool {
      a.clear();
      a.read_off.assign((size_t)(r1 - r0) + 1, 0);
      for (jsize r = r0; r < r1; r++) {
        jobject holder = holder_at(env, readDataArray, r);
        if (!holder) return false;
        const long len = append_field(env, holder, g.readBases, a.read_bases, -1);
        const bool ok = len >= 0 && append_field(env, holder, g.insertionGOP, a.ins, len) >= 0 &&
                        append_field(env, holder, g.deletionGOP, a.del, len) >= 0 &&
                        append_field(env, holder, g.overallGCP, a.gcp, len) >= 0 &&
                        append_field(env, holder, g.readQuals, a.read_quals, len) >= 0;
        gkljni::DeleteLocalRef(env, holder);
        if (!ok) return false;
        a.read_off[(size_t)(r - r0) + 1] = a.read_off[(size_t)(r - r0)] + len;
      }
      return true;
    };
    auto batch_of = [&](const ReadArena& a, int32_t n) {
      gklhip_batch b;
      b.n_reads = n; b.n_haps = n_haps;
      b.read_off = a.read_off.data(); b.hap_off = sl->hap_off.data();
      b.read_bases = a.read_bases.p; b.read_quals = a.read_quals.p; b.ins_gop = a.ins.p;
      b.del_gop = a.del.p; b.gcp = a.gcp.p; b.hap_bases = sl->hap_bases.p;
      return b;
    };
    const char* pv = getenv("GKL_HIP_JNI_PIPELINE_PAIRS");   // (read per call: tests switch it)
    const int64_t pipeline_from = pv && *pv ? atoll(pv) : 160000LL;   // (a 4000 x 50 call: 3.5 -> 3.2 ms pipelined; below ~150k pairs one range is all there is)
    bool pipelined = !(n_pairs < pipeline_from || n_reads < 64 || pipeline_from <= 0);
    if (pipelined && !sl->pipe) {
      // first big call of this slot: second engine (same configuration as the slot's first, whatever initNative has
      // been told since) and the compute threads.  If they cannot be had -- device memory, thread limit -- the call
      // runs in one shot like a small one.
      try {
        if (!sl->ctx2 && gklhip_init(&sl->cfg, &sl->ctx2) != GKLHIP_OK) sl->ctx2 = nullptr;   // one engine
        std::unique_ptr<Pipeline> p(new Pipeline());
        p->start(sl->ctx);
        if (sl->ctx2) { try { p->start(sl->ctx2); } catch (const std::exception&) {} }        // one compute thread
        sl->pipe = std::move(p);
      } catch (const std::exception&) {
        pipelined = false;
      }
    }
    if (!pipelined) {
      // ---- one shot (a GATK active region): marshal, compute, write back ----
      if (!sl->ranges.empty() && ++sl->calls_since_pipelined >= 16) {   // (the slot's engines are idle here: nothing reads the arenas)
        sl->ranges.clear();
        sl->ranges.shrink_to_fit();
        if (sl->out.capacity() * sizeof(double) > ((size_t)32 << 20)) std::vector<double>().swap(sl->out);
      }
      if (!marshal_reads(sl->whole, 0, n_reads)) return;
      const int64_t t_m = now_ns();
      if (n_pairs == 0) return;
      const gklhip_batch b = batch_of(sl->whole, n_reads);
      sl->out.resize((size_t)n_pairs);
      const int st = gklhip_compute(sl->ctx, &b, sl->out.data());
      const int64_t t_c = now_ns();
      if (st != GKLHIP_OK) { throw_status(env, st); return; }
      gkljni::SetDoubleArrayRegion(env, likelihoodArray, 0, (jsize)n_pairs, sl->out.data());
      const int64_t t_w = now_ns();
      g_timing[0] += t_m - t_call; g_timing[1] += t_c - t_m; g_timing[2] += t_w - t_c; g_timing[3] += t_w - t_call; g_timing[4]++;
      return;
    }
    // ---- pipelined: read ranges of ~150k pairs (100k..320k measure the same); range k+1 is marshalled while the ranges before it compute on the
    // slot's two engines, finished ranges go back to the Java array in between ----
    const char* rv = getenv("GKL_HIP_JNI_RANGE_PAIRS");
    const int64_t range_pairs = rv && atoll(rv) > 0 ? atoll(rv) : 150000LL;
    // Range boundaries: the first range is small (its marshalling is the only part nothing overlaps with), the later ones
    // grow geometrically (marshalling outruns compute, and bigger ranges use the chip better): sizes range_pairs * g^k.
    const char* gv = getenv("GKL_HIP_JNI_RANGE_GROWTH");
    const double growth = gv && atof(gv) >= 1.0 ? atof(gv) : 1.0;
    std::vector<jsize> cut{0};
    {
      double want = (double)range_pairs / (double)n_haps;   // reads in the next range
      double at = 0;
      while ((jsize)at < n_reads && cut.size() < 33) {
        at += std::max(1.0, want);
        cut.push_back((jsize)std::min<double>(at, n_reads));
        want *= growth;
      }
      cut.back() = n_reads;
      if (cut.size() >= 3 && cut.back() - cut[cut.size() - 2] < (cut[cut.size() - 2] - cut[cut.size() - 3]) / 3) {  // no sliver at the end
        cut.erase(cut.end() - 2);
      }
    }
    const int n_ranges = (int)cut.size() - 1;
    sl->calls_since_pipelined = 0;
    while ((int)sl->ranges.size() < n_ranges) sl->ranges.emplace_back(new ReadArena());
    sl->tasks.assign((size_t)n_ranges, RangeTask());
    sl->out.resize((size_t)n_pairs);
    int64_t ns_marshal = now_ns() - t_call, ns_wait = 0, ns_write = 0;
    int submitted = 0, finished = 0, failed_status = GKLHIP_OK;
    std::string failed_detail;
    bool java_exception = false;
    auto retire = [&](RangeTask* t) {   // a finished range: write it back (calling thread), or remember its error
      finished++;
      if (t->status != GKLHIP_OK) { if (failed_status == GKLHIP_OK) { failed_status = t->status; failed_detail = t->error; } return; }
      if (java_exception || failed_status != GKLHIP_OK) return;
      const int64_t t0 = now_ns();
      const int64_t at = t->out - sl->out.data();
      gkljni::SetDoubleArrayRegion(env, likelihoodArray, (jsize)at, (jsize)((int64_t)t->batch.n_reads * n_haps), t->out);
      if (gkljni::ExceptionCheck(env)) java_exception = true;
      ns_write += now_ns() - t0;
    };
    for (int k = 0; k < n_ranges && !java_exception && failed_status == GKLHIP_OK; k++) {
      const jsize r0 = cut[(size_t)k], r1 = cut[(size_t)k + 1];
      const int64_t t0 = now_ns();
      if (!marshal_reads(*sl->ranges[(size_t)k], r0, r1)) { java_exception = true; break; }
      ns_marshal += now_ns() - t0;
      RangeTask& t = sl->tasks[(size_t)k];
      t.k = k;
      t.batch = batch_of(*sl->ranges[(size_t)k], r1 - r0);
      t.out = sl->out.data() + (int64_t)r0 * n_haps;
      sl->pipe->submit(&t);
      submitted++;
      while (RangeTask* d = sl->pipe->take_done(false)) retire(d);
    }
    // whatever happened, the ranges in flight read this call's arenas: wait for all of them
    while (finished < submitted) {
      const int64_t t0 = now_ns();
      RangeTask* d = sl->pipe->take_done(true);
      ns_wait += now_ns() - t0;
      retire(d);
    }
    if (java_exception) return;
    if (failed_status != GKLHIP_OK) {
      char msg[600];
      snprintf(msg, sizeof msg, "GKL-HIP PairHMM: %s%s%s", gklhip_strerror(failed_status), failed_detail.empty() ? "" : ": ", failed_detail.c_str());
      throw_java(env, failed_status == GKLHIP_ERR_INVALID_ARG ? kIAE : failed_status == GKLHIP_ERR_OOM ? kOOM : kRTE, msg);
      return;
    }
    g_timing[0] += ns_marshal; g_timing[1] += ns_wait; g_timing[2] += ns_write; g_timing[3] += now_ns() - t_call; g_timing[4]++; g_timing[5]++;
  } catch (const std::bad_alloc&) {
    throw_java(env, kOOM, "Unable to allocate the PairHMM batch");
  } catch (const std::exception& e) {   // nothing may unwind into the JVM
    char msg[300];
    snprintf(msg, sizeof msg, "GKL-HIP PairHMM: %s", e.what());
    throw_java(env, kRTE, msg);
  }
}

// Diagnostics (not a JNI native): the call-time split summed since the last reset, nanoseconds -- see g_timing.
__attribute__((visibility("default"))) void gkl_pairhmm_jni_timing(int64_t out[6], int reset) {
  for (int i = 0; i < 6; i++) { out[i] = g_timing[i].load(); if (reset) g_timing[i].store(0); }
}

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative(JNIEnv*, jobject) {
  // The reference's doneNative is empty (IntelPairHmm.cc:189-192): other IntelPairHmm instances of the JVM keep
  // working after one of them closes.  Here it releases what no call is using (device memory, pinned arenas) and
  // keeps the configuration, so a later call simply gets a fresh slot.
  std::vector<std::unique_ptr<Slot>> dead;   // destroyed (streams synchronised, buffers freed) after the lock is released
  {
    std::lock_guard<std::mutex> lock(g.mu);
    for (auto it = g.slots.begin(); it != g.slots.end();) {
      if (!(*it)->busy) { dead.push_back(std::move(*it)); it = g.slots.erase(it); }
      else ++it;
    }
  }
}

}  // extern "C"
