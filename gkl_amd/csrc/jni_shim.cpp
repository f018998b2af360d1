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
//   * a read costs 13 JNI calls (r05: 28): the holder, its five byte[] fields, ONE GetArrayLength (readBases), five
//     GetByteArrayRegion and one ExceptionCheck -- a quality array shorter than readBases surfaces as the region copy's
//     ArrayIndexOutOfBoundsException and becomes IllegalArgumentException -- inside PushLocalFrame / PopLocalFrame
//     per block of 32 reads instead of six DeleteLocalRef per read;
//   * a BIG call is pipelined: the reads are marshalled range by range while two engines of the slot compute the
//     ranges already marshalled and the finished ranges are written back to the Java array -- the JNI calls of a
//     10k-read batch then hide behind the kernels (the reference pins everything, computes, releases:
//     JavaData.h:65-111, IntelPairHmm.cc:150-186).  With maxNumberOfThreads > 1 the ranges are marshalled by up to
//     that many threads: helpers of the slot that attach to the JVM (JavaVM from GetJavaVM, AttachCurrentThreadAsDaemon)
//     and read the holders through a global reference to readDataArray; Java exceptions are only ever raised on the
//     calling thread;
//   * a HIP failure inside a call is retried ONCE on fresh contexts (the reference never fails mid-run: it always has a
//     CPU kernel, IntelPairHmm.cc:99-113) before it becomes a RuntimeException;
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
#include <functional>
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
inline jint PushLocalFrame(JNIEnv* e, jint capacity) { return e->PushLocalFrame(capacity); }
inline jobject PopLocalFrame(JNIEnv* e, jobject result) { return e->PopLocalFrame(result); }
inline jobject NewGlobalRef(JNIEnv* e, jobject o) { return e->NewGlobalRef(o); }
inline void DeleteGlobalRef(JNIEnv* e, jobject o) { e->DeleteGlobalRef(o); }
inline jint GetJavaVM(JNIEnv* e, JavaVM** vm) { return e->GetJavaVM(vm); }
inline jint AttachCurrentThreadAsDaemon(JavaVM* vm, JNIEnv** penv) { return vm->AttachCurrentThreadAsDaemon(reinterpret_cast<void**>(penv), nullptr); }
inline jint DetachCurrentThread(JavaVM* vm) { return vm->DetachCurrentThread(); }
}  // namespace gkljni
#endif

namespace {

constexpr const char* kIAE = "java/lang/IllegalArgumentException";
constexpr const char* kOOM = "java/lang/OutOfMemoryError";
constexpr const char* kRTE = "java/lang/RuntimeException";

// Page-locked byte arena (one per marshalled field and slot): grows with the biggest call, and an arena above 32 MB that the
// last 16 calls each filled to less than a quarter -- and that has not grown for 64 calls: hipHostFree synchronises the whole
// device -- is given back (like the context's own buffers, pairhmm_api.hip: trim_due) -- one 1.28 M-pair call must not keep
// ~100 MB pinned per slot for the life of the JVM.  clear() runs at the start of a call, when nothing of the slot's previous
// call is in flight any more.
struct PinnedBytes {
  uint8_t* p = nullptr;
  size_t cap = 0, len = 0;
  int small_uses = 0, since_grow = 64;
  PinnedBytes() = default;
  PinnedBytes(const PinnedBytes&) = delete;
  PinnedBytes& operator=(const PinnedBytes&) = delete;
  ~PinnedBytes() { gklhip_host_free(p); }
  void clear() {
    if (since_grow < 64) since_grow++;
    if (cap > ((size_t)32 << 20) && len < cap / 4) {
      if (small_uses < 16) small_uses++;
      if (small_uses >= 16 && since_grow >= 64) { gklhip_host_free(p); p = nullptr; cap = 0; small_uses = 0; }
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
      since_grow = 0;
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
// gklhip_compute on them and reports on `done`.  Write-back and exceptions stay on the calling thread; marshalling is
// the calling thread's and, with maxNumberOfThreads > 1, the slot's helper threads' (MarshalHelpers below).
struct RangeTask {
  int k = 0;
  gklhip_batch batch;
  double* out = nullptr;
  int status = GKLHIP_OK;
  bool computed = false;   // gklhip_compute returned GKLHIP_OK for it
  std::string error;
};
struct Pipeline {
  std::mutex mu;
  std::condition_variable has_todo, has_done;
  std::deque<RangeTask*> todo, done;
  int marshalling = 0;   // helper threads still inside the current call's marshalling job (take_done's other wake-up reason)
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
          t->computed = t->status == GKLHIP_OK;
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
  // a finished range; with wait: blocks until there is one, or -- NULL -- until no helper is marshalling any more
  RangeTask* take_done(bool wait) {
    std::unique_lock<std::mutex> l(mu);
    if (wait) has_done.wait(l, [&] { return !done.empty() || marshalling == 0; });
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

// Helper threads of a slot that marshal read ranges beside the calling thread (maxNumberOfThreads > 1).  A thread the
// library starts has no JNIEnv: it attaches to the JVM once (as a daemon: it never keeps the JVM alive), keeps its env
// for the life of the slot and detaches when the slot goes (doneNative, a re-configuration, process exit).  What it may
// touch: GLOBAL references only (the calling thread's jobjectArray is a local reference of THAT thread).
std::atomic<bool> g_process_exiting{false};   // set by the process-wide state's destructor: the JVM may be gone -- no JNI call from then on
struct MarshalHelpers {
  JavaVM* vm = nullptr;
  std::mutex mu;
  std::condition_variable wake, reported;
  std::function<void(JNIEnv*)> job;   // what a helper that takes a ticket runs once
  int tickets = 0;                    // helpers the current job still wants
  int started = 0, live = 0;          // threads that have reported in / that hold a JNIEnv
  bool quit = false;
  std::vector<std::thread> threads;
  explicit MarshalHelpers(JavaVM* v) : vm(v) {}
  // at least n helper threads, if the JVM lets them attach; returns how many are usable
  int ensure(int n) {
    while ((int)threads.size() < n) {
      threads.emplace_back([this] {
        JNIEnv* env = nullptr;
        const bool ok = gkljni::AttachCurrentThreadAsDaemon(vm, &env) == JNI_OK && env;
        std::unique_lock<std::mutex> l(mu);
        started++;
        if (ok) live++;
        reported.notify_all();
        if (!ok) return;   // (jobs simply run with fewer threads)
        for (;;) {
          wake.wait(l, [&] { return quit || tickets > 0; });
          if (quit) break;
          tickets--;
          auto fn = job;
          l.unlock();
          fn(env);
          l.lock();
        }
        live--;
        l.unlock();
        // (not at process exit: static destructors run after the JVM has shut down, and the invocation interface of a
        //  destroyed VM must not be called -- the thread simply ends)
        if (!g_process_exiting.load()) gkljni::DetachCurrentThread(vm);
      });
    }
    std::unique_lock<std::mutex> l(mu);
    reported.wait(l, [&] { return started == (int)threads.size(); });
    return live;
  }
  void run(int n, std::function<void(JNIEnv*)>&& fn) noexcept {   // n <= ensure()'s answer helpers run fn once each; returns at once
    { std::lock_guard<std::mutex> l(mu); job = std::move(fn); tickets = n; }
    wake.notify_all();
  }
  ~MarshalHelpers() {
    { std::lock_guard<std::mutex> l(mu); quit = true; }
    wake.notify_all();
    for (auto& th : threads) th.join();
  }
};

// Everything one call needs; a slot serves one caller at a time.
struct Slot {
  gklhip_ctx* ctx = nullptr;
  gklhip_ctx* ctx2 = nullptr;          // second engine of pipelined big calls (created by the first of them)
  std::unique_ptr<Pipeline> pipe;      // declared after the contexts: its threads are joined before they go
  std::unique_ptr<MarshalHelpers> helpers;
  PinnedBytes hap_bases;
  std::vector<int64_t> hap_off;
  ReadArena whole;                     // a small call's reads
  std::vector<std::unique_ptr<ReadArena>> ranges;  // a pipelined call's read ranges
  std::vector<RangeTask> tasks;
  std::vector<double> out;
  int calls_since_pipelined = 0;       // one-shot calls since the slot's last pipelined one: after 16 its range arenas (and a result vector above 32 MB) go
  bool busy = false;
  int64_t idle_since = 0;              // when the last call returned it (steady clock, ns)
  bool idle_released = false;          // the janitor has been through it since then
  bool with_janitor = false;           // leased by the janitor right now (doneNative waits for it: milliseconds)
  int gen = 0;  // configuration generation (initNative with other arguments starts a new one)
  gklhip_config cfg;  // what `ctx` was created with: the second engine of a pipelined call gets the same
  ~Slot() {
    helpers.reset();
    pipe.reset();
    if (ctx2) gklhip_done(ctx2);
    if (ctx) gklhip_done(ctx);
  }
};

// Where a call's time goes, summed over the calls of the process (nanoseconds; gkl_pairhmm_jni_timing reads them):
// [0] marshalling on the calling thread, [1] waiting for compute that marshalling did not cover (a small call: the
// whole gklhip_compute), [2] write-back into the Java array, [3] whole calls, [4] number of calls, [5] pipelined calls.
std::atomic<int64_t> g_timing[6];
// gkl_pairhmm_jni_helpers: [0] ns of marshalling on helper threads, [1] read ranges marshalled by helpers, [2] by
// calling threads, [3] calls retried after a HIP failure ([4]: streams / engines the janitor gave back, g_idle_released)
std::atomic<int64_t> g_helpers[4];
int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Process-wide state, like the reference's globals (IntelPairHmm.cc:41-48) and the
// static field IDs of JavaData (JavaData.h:160-176).
struct FieldIds {
  jfieldID readBases = nullptr, readQuals = nullptr, insertionGOP = nullptr, deletionGOP = nullptr,
           overallGCP = nullptr, haplotypeBases = nullptr;
};
struct State {
  std::mutex mu;
  std::condition_variable slot_free;
  std::vector<std::unique_ptr<Slot>> slots;
  int max_slots = 4;
  int creating = 0;  // slots being initialised outside the lock
  bool ready = false;  // initNative has run: configuration and field IDs are valid (stays true after doneNative)
  int gen = 0;         // current configuration generation; slots of older generations die when they come back
  gklhip_config cfg;
  FieldIds f;
  JavaVM* vm = nullptr;   // from initNative's GetJavaVM: what helper threads attach to
  // The janitor: a slot nobody has used for a second (GKL_HIP_IDLE_RELEASE_MS; 0 = never) gives back what it holds only
  // for speed -- the second engine and compute threads of pipelined calls, the first engine's extra streams and twin
  // engines (gklhip_release_idle) -- so that an idle JVM that once sent a big batch holds one or two hardware queues on
  // the device, not a dozen (docs/NOTES.md 49, 55: the device's scheduler rotates every process's queues).
  std::thread janitor;
  std::condition_variable janitor_wake;
  bool janitor_quit = false;
  int64_t idle_release_ns = 1000000000LL;
  ~State() {
    g_process_exiting = true;
    { std::lock_guard<std::mutex> l(mu); janitor_quit = true; }
    janitor_wake.notify_all();
    if (janitor.joinable()) janitor.join();
  }
} g;
std::atomic<int64_t> g_idle_released{0};   // streams given back by the janitor so far (gkl_pairhmm_jni_helpers [4])

void throw_java(JNIEnv* env, const char* class_path, const char* msg) {
  gkljni::ExceptionClear(env);  // IntelPairHmm.cc:65-66
  jclass c = gkljni::FindClass(env, class_path);
  if (c) gkljni::ThrowNew(env, c, msg);
}

void throw_status_text(JNIEnv* env, int status, const char* detail) {
  char msg[600];
  snprintf(msg, sizeof msg, "GKL-HIP PairHMM: %s%s%s", gklhip_strerror(status),
           (detail && *detail) ? ": " : "", (detail && *detail) ? detail : "");
  const char* cls = status == GKLHIP_ERR_INVALID_ARG ? kIAE : status == GKLHIP_ERR_OOM ? kOOM : kRTE;
  throw_java(env, cls, msg);
}
void throw_status(JNIEnv* env, int status) { throw_status_text(env, status, gklhip_last_error()); }

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// Marshalling errors travel as (class, message): a helper thread must not raise them on ITS JNIEnv -- nobody would see
// them -- so every marshalling function reports, and the calling thread throws.
struct MarshalError {
  const char* cls = nullptr;   // nullptr = fine
  const char* msg = nullptr;
  explicit operator bool() const { return cls != nullptr; }
};

constexpr jsize kFrameReads = 32;   // reads per local frame: 6 references each

// PushLocalFrame / PopLocalFrame around a block (also when a C++ exception -- an arena that cannot grow -- passes through)
struct LocalFrame {
  JNIEnv* env;
  bool pushed;
  LocalFrame(JNIEnv* e, jint capacity) : env(e), pushed(gkljni::PushLocalFrame(e, capacity) == 0) {
    if (!pushed) gkljni::ExceptionClear(env);   // (PushLocalFrame raised OutOfMemoryError)
  }
  ~LocalFrame() { if (pushed) gkljni::PopLocalFrame(env, nullptr); }
  LocalFrame(const LocalFrame&) = delete;
};

// Reads [r0, r1) of `reads` (a reference valid on THIS thread) into `a`, offsets rebased to 0.  Per read: the holder,
// its five fields, readBases' length, five region copies, one ExceptionCheck = 13 JNI calls (JavaData.h:84-105 pins
// five arrays per read and keeps them until the call returns).  JavaData.h:86-91: the read length is readBases.length
// and the other four arrays are read for that many bytes; a shorter one is an error here (the reference reads past it)
// -- it shows as the region copy's ArrayIndexOutOfBoundsException, which is cleared and reported as IllegalArgumentException.
MarshalError marshal_reads(JNIEnv* env, jobjectArray reads, const FieldIds& f, ReadArena& a, jsize r0, jsize r1) {
  a.clear();
  a.read_off.assign((size_t)(r1 - r0) + 1, 0);
  for (jsize b0 = r0; b0 < r1; b0 += kFrameReads) {
    const jsize b1 = std::min<jsize>(r1, b0 + kFrameReads);
    LocalFrame frame(env, 6 * kFrameReads);
    if (!frame.pushed) return {kOOM, "Unable to allocate a local reference frame"};
    MarshalError err;
    for (jsize r = b0; r < b1; r++) {
      jobject holder = gkljni::GetObjectArrayElement(env, reads, r);
      if (!holder) { err = {kIAE, "null element in data holder array"}; break; }
      jbyteArray bases = (jbyteArray)gkljni::GetObjectField(env, holder, f.readBases);
      jbyteArray ins = (jbyteArray)gkljni::GetObjectField(env, holder, f.insertionGOP);
      jbyteArray del = (jbyteArray)gkljni::GetObjectField(env, holder, f.deletionGOP);
      jbyteArray gcp = (jbyteArray)gkljni::GetObjectField(env, holder, f.overallGCP);
      jbyteArray quals = (jbyteArray)gkljni::GetObjectField(env, holder, f.readQuals);
      if (!bases || !ins || !del || !gcp || !quals) { err = {kIAE, "null byte[] field in data holder"}; break; }
      const jsize len = gkljni::GetArrayLength(env, bases);
      const size_t n = (size_t)len;
      // (space for all five first: a failed allocation must not leave a region copy half-way)
      jbyte* dst[5] = {reinterpret_cast<jbyte*>(a.read_bases.grow(n)), reinterpret_cast<jbyte*>(a.ins.grow(n)),
                       reinterpret_cast<jbyte*>(a.del.grow(n)), reinterpret_cast<jbyte*>(a.gcp.grow(n)),
                       reinterpret_cast<jbyte*>(a.read_quals.grow(n))};
      gkljni::GetByteArrayRegion(env, bases, 0, len, dst[0]);
      gkljni::GetByteArrayRegion(env, ins, 0, len, dst[1]);
      gkljni::GetByteArrayRegion(env, del, 0, len, dst[2]);
      gkljni::GetByteArrayRegion(env, gcp, 0, len, dst[3]);
      gkljni::GetByteArrayRegion(env, quals, 0, len, dst[4]);
      if (gkljni::ExceptionCheck(env)) {
        gkljni::ExceptionClear(env);
        err = {kIAE, "read quality array shorter than readBases"};
        break;
      }
      a.read_off[(size_t)(r - r0) + 1] = a.read_off[(size_t)(r - r0)] + len;
    }
    if (err) return err;
  }
  return {};
}

// The haplotypes: four JNI calls each, one frame per block.
MarshalError marshal_haps(JNIEnv* env, jobjectArray haps, const FieldIds& f, PinnedBytes& bytes, std::vector<int64_t>& off, jsize n_haps) {
  bytes.clear();
  off.assign((size_t)n_haps + 1, 0);
  for (jsize b0 = 0; b0 < n_haps; b0 += 3 * kFrameReads) {
    const jsize b1 = std::min<jsize>(n_haps, b0 + 3 * kFrameReads);
    LocalFrame frame(env, 6 * kFrameReads);
    if (!frame.pushed) return {kOOM, "Unable to allocate a local reference frame"};
    MarshalError err;
    for (jsize h = b0; h < b1; h++) {
      jobject holder = gkljni::GetObjectArrayElement(env, haps, h);
      if (!holder) { err = {kIAE, "null element in data holder array"}; break; }
      jbyteArray bases = (jbyteArray)gkljni::GetObjectField(env, holder, f.haplotypeBases);
      if (!bases) { err = {kIAE, "null byte[] field in data holder"}; break; }
      const jsize len = gkljni::GetArrayLength(env, bases);
      if (len > 0) gkljni::GetByteArrayRegion(env, bases, 0, len, reinterpret_cast<jbyte*>(bytes.grow((size_t)len)));
      off[(size_t)h + 1] = off[(size_t)h] + len;
    }
    if (err) return err;
  }
  return {};
}

// Read ranges of a pipelined call: boundaries cut[0] = 0 < ... < cut.back() = n_reads (at most 32 ranges).
// What the schedule trades (measured on the 10 000 x 128 batch, tools/jni_marshal_probe.py, docs/NOTES.md 54): the first
// range's marshalling is the only part nothing overlaps with -- small first range; a range is one gklhip_compute, and
// small ones use the chip badly (nine ranges of 150k pairs: 13.7 ms of GPU time for 11.7 ms of work) -- few, big
// ranges; the GPU must not run dry while the big ranges are still being marshalled -- sizes that GROW (a thread marshals
// whole ranges, so helpers shorten the queue of ranges, not one range); what follows the last kernel (the fp64 pairs'
// log10 on the host, the write-back) is proportional to the last range -- a descending tail:
//   4 / 12 / 28 / 36 / 14 / 6 per cent of the reads, whatever the number of marshalling threads.
// With the mock JVM's 5 ns JNI functions every schedule of 4-8 ranges measures the same (13.3-13.4 ms at four threads,
// 13.7-14.0 at one); with 25 ns more per function -- a real JVM's order -- this one holds (13.3-13.9 / 13.9-14.2) where
// 4 / 32 / 32 / 32 leaves the GPU waiting for its second range (14.0-14.3 / 15.2-16.6).  r05's equal 150k ranges: 14.9-15.3.
// A range below 40 000 pairs (the first) / 60 000 pairs (the others) is merged with its neighbour: a call of 200k pairs
// becomes two ranges.  GKL_HIP_JNI_RANGE_SHARES="a,b,c" sets the per cents, GKL_HIP_JNI_RANGE_PAIRS=n asks for equal
// ranges of n pairs (tests force many small ranges with it); both are read per call.
std::vector<jsize> plan_ranges(jsize n_reads, jsize n_haps, int marshal_threads) {
  std::vector<jsize> cut{0};
  const char* rv = getenv("GKL_HIP_JNI_RANGE_PAIRS");
  if (rv && atoll(rv) > 0) {
    const double per = std::max(1.0, (double)atoll(rv) / (double)std::max<jsize>(1, n_haps));
    for (double at = per; cut.size() < 32 && (jsize)at < n_reads; at += per) if ((jsize)at > cut.back()) cut.push_back((jsize)at);
    cut.push_back(n_reads);
    if (cut.size() >= 4 && cut.back() - cut[cut.size() - 2] < (cut[cut.size() - 2] - cut[cut.size() - 3]) / 3) cut.erase(cut.end() - 2);  // no sliver at the end
    return cut;
  }
  std::vector<double> shares;
  const char* sv = getenv("GKL_HIP_JNI_RANGE_SHARES");
  bool floors = true;
  if (sv && *sv) {
    floors = false;   // (an explicit schedule is taken as given)
    for (const char* q = sv; *q && shares.size() < 32;) {
      char* end = nullptr;
      const double v = strtod(q, &end);
      if (end == q) break;
      if (v > 0) shares.push_back(v);
      q = *end ? end + 1 : end;
    }
  }
  if (shares.empty()) {
    floors = true;
    shares = {4, 12, 28, 36, 14, 6};
  }
  (void)marshal_threads;
  double total = 0;
  for (double v : shares) total += v;
  const double floor_first = 40000.0 / std::max<jsize>(1, n_haps), floor_rest = 60000.0 / std::max<jsize>(1, n_haps);   // in reads
  double acc = 0;
  for (size_t k = 0; k + 1 < shares.size(); k++) {
    acc += shares[k];
    const jsize at = (jsize)std::min<double>(n_reads, acc / total * n_reads);
    const double need = floors ? (cut.size() == 1 ? floor_first : floor_rest) : 1.0;
    if (at - cut.back() >= need && at < n_reads) cut.push_back(at);
  }
  if (floors && cut.size() > 1 && n_reads - cut.back() < floor_rest) cut.pop_back();   // (the last range takes a too-small remainder in)
  cut.push_back(n_reads);
  return cut;
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
  bool janitor = false;
  ~SlotLease() {
    if (!s) return;
    std::unique_ptr<Slot> dead;
    {
      std::lock_guard<std::mutex> lock(g.mu);
      s->busy = false;
      if (janitor) { s->idle_released = true; s->with_janitor = false; }
      else { s->idle_since = now_ns(); s->idle_released = false; }
      if (s->gen != g.gen)  // initNative changed the configuration while this call ran: the slot is not reused
        for (auto it = g.slots.begin(); it != g.slots.end(); ++it)
          if (it->get() == s) { dead = std::move(*it); g.slots.erase(it); break; }
    }
    g.slot_free.notify_all();  // waiters differ (a caller that wants a slot, any number of them): wake all
  }
};

// A HIP failure inside a call (GKLHIP_ERR_HIP: a launch or copy that failed, a stream in an error state): the slot
// drops its engines -- both contexts and the compute threads bound to them -- and makes a fresh first one.  The
// reference cannot fail mid-run; a GATK job of hours should not die of one transient device error either.  False when
// no new context can be had (the caller then throws the original error).
bool renew_engines(Slot* sl, int status, const char* detail) {
  fprintf(stderr, "GKL-HIP PairHMM: %s%s%s -- retrying the call once on a fresh device context\n", gklhip_strerror(status),
          (detail && *detail) ? ": " : "", (detail && *detail) ? detail : "");
  sl->pipe.reset();
  if (sl->ctx2) { gklhip_done(sl->ctx2); sl->ctx2 = nullptr; }
  if (sl->ctx) { gklhip_done(sl->ctx); sl->ctx = nullptr; }
  g_helpers[3]++;
  if (gklhip_init(&sl->cfg, &sl->ctx) != GKLHIP_OK) {
    sl->ctx = nullptr;
    // the slot has no engine any more: it must not be leased again
    std::lock_guard<std::mutex> lock(g.mu);
    sl->gen = -1;
    return false;
  }
  return true;
}

void janitor_loop() {
  std::unique_lock<std::mutex> lock(g.mu);
  while (!g.janitor_quit) {
    g.janitor_wake.wait_for(lock, std::chrono::milliseconds(250));
    if (g.janitor_quit) break;
    if (g.idle_release_ns <= 0) continue;
    const int64_t now = now_ns();
    Slot* pick = nullptr;
    for (auto& s : g.slots)
      if (!s->busy && !s->idle_released && now - s->idle_since >= g.idle_release_ns) { pick = s.get(); break; }
    if (!pick) continue;
    pick->busy = true;   // leased like a caller would: nobody else touches it, initNative skips it, doneNative waits for it
    pick->with_janitor = true;
    lock.unlock();
    {
      SlotLease lease{pick, true};
      pick->pipe.reset();
      if (pick->ctx2) { gklhip_done(pick->ctx2); pick->ctx2 = nullptr; g_idle_released += 1; }
      pick->ranges.clear();
      int32_t n = 0;
      if (pick->ctx && gklhip_release_idle(pick->ctx, &n) == GKLHIP_OK) g_idle_released += n;
    }
    lock.lock();
  }
}

void start_janitor_locked() {   // g.mu held
  if (g.janitor.joinable()) return;
  g.idle_release_ns = (int64_t)std::max(0, env_int("GKL_HIP_IDLE_RELEASE_MS", 1000)) * 1000000LL;
  try { g.janitor = std::thread(janitor_loop); } catch (const std::exception&) {}   // (no thread: nothing is given back early, that is all)
}

}  // namespace

extern "C" {

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative(
    JNIEnv* env, jclass, jclass readDataHolder, jclass haplotypeDataHolder, jboolean use_double,
    jint max_threads) {
  std::unique_lock<std::mutex> lock(g.mu);
  struct { jfieldID* dst; jclass cls; const char* name; } fields[] = {
      {&g.f.readBases, readDataHolder, "readBases"},       {&g.f.readQuals, readDataHolder, "readQuals"},
      {&g.f.insertionGOP, readDataHolder, "insertionGOP"}, {&g.f.deletionGOP, readDataHolder, "deletionGOP"},
      {&g.f.overallGCP, readDataHolder, "overallGCP"},     {&g.f.haplotypeBases, haplotypeDataHolder, "haplotypeBases"}};
  for (auto& f : fields) {
    jfieldID id = f.cls ? gkljni::GetFieldID(env, f.cls, f.name, "[B") : nullptr;
    if (!id) {  // JavaData.h:127-133
      lock.unlock();
      throw_java(env, kIAE, "Unable to get field ID");
      return;
    }
    *f.dst = id;
  }
  {
    JavaVM* vm = nullptr;   // for the marshalling helpers of big calls; without it they are simply not used
    if (gkljni::GetJavaVM(env, &vm) == JNI_OK && vm) g.vm = vm;
    else gkljni::ExceptionClear(env);
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
  first->idle_since = now_ns();
  g.slots.push_back(std::move(first));
  g.ready = true;
  start_janitor_locked();
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
  jobject reads_global = nullptr;   // for the helper threads of this call; deleted on every way out
  struct GlobalGuard { JNIEnv* env; jobject* ref; ~GlobalGuard() { if (*ref) gkljni::DeleteGlobalRef(env, *ref); } } global_guard{env, &reads_global};
  try {
    FieldIds fid;
    JavaVM* vm;
    { std::lock_guard<std::mutex> lock(g.mu); fid = g.f; vm = g.vm; }
    const jsize n_reads = gkljni::GetArrayLength(env, readDataArray);
    const jsize n_haps = gkljni::GetArrayLength(env, haplotypeDataArray);
    const int64_t n_pairs = (int64_t)n_reads * n_haps;
    if (n_pairs > 0x7fffffffLL) { throw_java(env, kIAE, "more than 2^31 read x haplotype pairs"); return; }
    if ((int64_t)gkljni::GetArrayLength(env, likelihoodArray) < n_pairs) {
      throw_java(env, kIAE, "likelihood array shorter than reads x haplotypes");
      return;
    }
    if (MarshalError e = marshal_haps(env, haplotypeDataArray, fid, sl->hap_bases, sl->hap_off, n_haps)) { throw_java(env, e.cls, e.msg); return; }
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
      if (MarshalError e = marshal_reads(env, readDataArray, fid, sl->whole, 0, n_reads)) { throw_java(env, e.cls, e.msg); return; }
      const int64_t t_m = now_ns();
      if (n_pairs == 0) return;
      const gklhip_batch b = batch_of(sl->whole, n_reads);
      sl->out.resize((size_t)n_pairs);
      int st = gklhip_compute(sl->ctx, &b, sl->out.data());
      if (st == GKLHIP_ERR_HIP) {
        const char* d = gklhip_last_error();
        const std::string detail = d ? d : "";
        if (!renew_engines(sl, st, detail.c_str())) { throw_status_text(env, st, detail.c_str()); return; }
        st = gklhip_compute(sl->ctx, &b, sl->out.data());
      }
      const int64_t t_c = now_ns();
      if (st != GKLHIP_OK) { throw_status(env, st); return; }
      gkljni::SetDoubleArrayRegion(env, likelihoodArray, 0, (jsize)n_pairs, sl->out.data());
      const int64_t t_w = now_ns();
      g_timing[0] += t_m - t_call; g_timing[1] += t_c - t_m; g_timing[2] += t_w - t_c; g_timing[3] += t_w - t_call; g_timing[4]++;
      return;
    }
    // ---- pipelined: read ranges; range k+1 is marshalled while the ranges before it compute on the slot's two engines,
    // finished ranges go back to the Java array in between ----
    int want_threads;   // marshalling threads this call may use: the calling thread + helpers
    {
      const char* hv = getenv("GKL_HIP_JNI_MARSHAL_THREADS");
      want_threads = std::max(1, std::min(hv && *hv ? atoi(hv) : sl->cfg.max_threads, 8));
      if (!vm) want_threads = 1;
    }
    const std::vector<jsize> cut = plan_ranges(n_reads, n_haps, want_threads);
    const int n_ranges = (int)cut.size() - 1;
    sl->calls_since_pipelined = 0;
    while ((int)sl->ranges.size() < n_ranges) sl->ranges.emplace_back(new ReadArena());
    sl->tasks.assign((size_t)n_ranges, RangeTask());
    sl->out.resize((size_t)n_pairs);
    for (int k = 0; k < n_ranges; k++) {
      sl->tasks[(size_t)k].k = k;
      sl->tasks[(size_t)k].out = sl->out.data() + (int64_t)cut[(size_t)k] * n_haps;
    }
    // helper threads (maxNumberOfThreads - 1 of them, at most 7 and one per range beyond the first): attached to the JVM,
    // reading the holders through a global reference
    int n_helpers = std::min(want_threads - 1, n_ranges - 1);
    {
      if (n_helpers > 0 && vm) {
        try {
          if (!sl->helpers || sl->helpers->vm != vm) sl->helpers.reset(new MarshalHelpers(vm));
          n_helpers = std::min(n_helpers, sl->helpers->ensure(n_helpers));
          if (n_helpers > 0) reads_global = gkljni::NewGlobalRef(env, readDataArray);
        } catch (const std::exception&) { n_helpers = 0; }   // (no thread to be had: the calling thread marshals alone)
      }
      if (!reads_global || !sl->helpers) n_helpers = 0;
    }
    // shared by the marshalling threads of this call
    struct Job {
      std::atomic<int> next{0};
      std::atomic<bool> stop{false};
      std::mutex mu;
      MarshalError error;             // first marshalling error
      std::string exception_what;     // a C++ exception on a helper (bad_alloc of an arena)
    } job;
    auto marshal_range = [&](JNIEnv* e, jobjectArray arr, int k) -> bool {
      const jsize r0 = cut[(size_t)k], r1 = cut[(size_t)k + 1];
      MarshalError err = marshal_reads(e, arr, fid, *sl->ranges[(size_t)k], r0, r1);
      if (err) {
        std::lock_guard<std::mutex> l(job.mu);
        if (!job.error) job.error = err;
        job.stop = true;
        return false;
      }
      RangeTask& t = sl->tasks[(size_t)k];
      t.batch = batch_of(*sl->ranges[(size_t)k], r1 - r0);
      sl->pipe->submit(&t);
      return true;
    };
    std::atomic<int> submitted{0};
    if (n_helpers > 0) {
      std::function<void(JNIEnv*)> helper_job = [&, arr = (jobjectArray)reads_global](JNIEnv* e) {
        const int64_t t0 = now_ns();
        try {
          for (;;) {
            if (job.stop) break;
            const int k = job.next.fetch_add(1);
            if (k >= n_ranges) break;
            if (!marshal_range(e, arr, k)) break;
            submitted++;
            g_helpers[1]++;
          }
        } catch (const std::exception& ex) {
          std::lock_guard<std::mutex> l(job.mu);
          if (job.exception_what.empty()) job.exception_what = ex.what();
          job.stop = true;
        }
        g_helpers[0] += now_ns() - t0;
        { std::lock_guard<std::mutex> l(sl->pipe->mu); sl->pipe->marshalling--; }
        sl->pipe->has_done.notify_all();
      };
      { std::lock_guard<std::mutex> l(sl->pipe->mu); sl->pipe->marshalling = n_helpers; }
      sl->helpers->run(n_helpers, std::move(helper_job));   // (from here on the helpers use this frame's variables: see the wait below)
    }
    int64_t ns_marshal = now_ns() - t_call, ns_wait = 0, ns_write = 0;
    int finished = 0, failed_status = GKLHIP_OK;
    std::string failed_detail;
    bool java_exception = false;
    auto retire = [&](RangeTask* t) {   // a finished range: write it back (calling thread), or remember its error
      finished++;
      if (t->status != GKLHIP_OK) {
        if (failed_status == GKLHIP_OK) { failed_status = t->status; failed_detail = t->error; }
        if (t->status != GKLHIP_ERR_HIP) job.stop = true;   // (a HIP failure: the ranges are still wanted -- the call is retried from them)
        return;
      }
      if (java_exception || failed_status != GKLHIP_OK) return;
      const int64_t t0 = now_ns();
      const int64_t at = t->out - sl->out.data();
      gkljni::SetDoubleArrayRegion(env, likelihoodArray, (jsize)at, (jsize)((int64_t)t->batch.n_reads * n_haps), t->out);
      if (gkljni::ExceptionCheck(env)) { java_exception = true; job.stop = true; }
      ns_write += now_ns() - t0;
    };
    // Whatever happens below, helpers may be marshalling into this call's arenas and ranges in flight read them: nothing
    // leaves this block before the helpers are out of the job and every submitted range has come back.
    std::exception_ptr pending_cxx;
    try {
      for (;;) {
        if (job.stop) break;
        const int k = job.next.fetch_add(1);
        if (k >= n_ranges) break;
        const int64_t t0 = now_ns();
        const bool ok = marshal_range(env, readDataArray, k);
        ns_marshal += now_ns() - t0;
        if (!ok) break;
        submitted++;
        g_helpers[2]++;
        while (RangeTask* d = sl->pipe->take_done(false)) retire(d);   // (an exception in here: the range is counted, see retire)
      }
    } catch (...) {
      pending_cxx = std::current_exception();
      job.stop = true;
    }
    for (;;) {
      const int64_t t0 = now_ns();
      RangeTask* d = sl->pipe->take_done(true);   // NULL: no helper is marshalling any more and nothing is finished right now
      ns_wait += now_ns() - t0;
      if (d) { try { retire(d); } catch (...) { if (!pending_cxx) pending_cxx = std::current_exception(); } continue; }   // (retire counts the range first thing)
      if (finished >= submitted.load()) break;
      // (helpers are done, ranges are still computing: wait for one)
      const int64_t t1 = now_ns();
      {
        std::unique_lock<std::mutex> l(sl->pipe->mu);
        sl->pipe->has_done.wait(l, [&] { return !sl->pipe->done.empty(); });
      }
      ns_wait += now_ns() - t1;
    }
    if (pending_cxx) std::rethrow_exception(pending_cxx);
    if (java_exception) return;
    if (job.error) { throw_java(env, job.error.cls, job.error.msg); return; }
    if (!job.exception_what.empty()) { throw_java(env, kOOM, "Unable to allocate the PairHMM batch"); return; }
    if (failed_status == GKLHIP_ERR_HIP) {
      // every range is marshalled (a HIP failure does not stop the marshalling): fresh engines, and the ranges that did
      // not come back good are computed again, one after the other on the calling thread
      if (job.next.load() < n_ranges || !renew_engines(sl, failed_status, failed_detail.c_str())) { throw_status_text(env, failed_status, failed_detail.c_str()); return; }
      for (int k = 0; k < n_ranges; k++) {
        RangeTask& t = sl->tasks[(size_t)k];
        const int st = t.computed ? GKLHIP_OK : gklhip_compute(sl->ctx, &t.batch, t.out);
        if (st != GKLHIP_OK) { throw_status(env, st); return; }
        gkljni::SetDoubleArrayRegion(env, likelihoodArray, (jsize)(t.out - sl->out.data()), (jsize)((int64_t)t.batch.n_reads * n_haps), t.out);
        if (gkljni::ExceptionCheck(env)) return;
      }
      failed_status = GKLHIP_OK;
    }
    if (failed_status != GKLHIP_OK) { throw_status_text(env, failed_status, failed_detail.c_str()); return; }
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
// ... and what the marshalling helpers did -- see g_helpers.
__attribute__((visibility("default"))) void gkl_pairhmm_jni_helpers(int64_t out[5], int reset) {
  for (int i = 0; i < 4; i++) { out[i] = g_helpers[i].load(); if (reset) g_helpers[i].store(0); }
  out[4] = g_idle_released.load();
  if (reset) g_idle_released.store(0);
}

JNIEXPORT void JNICALL Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative(JNIEnv*, jobject) {
  // The reference's doneNative is empty (IntelPairHmm.cc:189-192): other IntelPairHmm instances of the JVM keep
  // working after one of them closes.  Here it releases what no call is using (device memory, pinned arenas, the
  // helper threads -- they detach from the JVM) and keeps the configuration, so a later call simply gets a fresh slot.
  std::vector<std::unique_ptr<Slot>> dead;   // destroyed (streams synchronised, buffers freed) after the lock is released
  {
    std::unique_lock<std::mutex> lock(g.mu);
    // (a slot the janitor is tidying right now is idle, not in use: it goes too, once the janitor has let go of it)
    g.slot_free.wait(lock, [] { for (auto& s : g.slots) if (s->with_janitor) return false; return true; });
    for (auto it = g.slots.begin(); it != g.slots.end();) {
      if (!(*it)->busy) { dead.push_back(std::move(*it)); it = g.slots.erase(it); }
      else ++it;
    }
  }
}

}  // extern "C"
