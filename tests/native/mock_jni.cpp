// Test harness: a mock JVM side for libgkl_pairhmm.so (no JDK in this image, none on the GPU boxes: docs/NOTES.md 52).
//
// Builds a JNINativeInterface_ function table (spec slot indices, gkl_amd/csrc/jni_min.h) and a JavaVM invocation
// table over a small object model, dlopen()s the drop-in library the way NativeLibraryLoader/System.load would (reference
// src/main/java/com/intel/gkl/NativeLibraryLoader.java:114-128), resolves the three
// Java_com_intel_gkl_pairhmm_IntelPairHmm_* symbols by name like the JVM does, and drives
// initNative -> computeLikelihoodsNative -> doneNative with ReadDataHolder / HaplotypeDataHolder look-alikes.
// Every JNI slot the shims do not declare aborts, so the tests also pin the exact set of JNI functions they may call.
//
// The reference model follows HotSpot's: a jobject is a POINTER TO A SLOT that holds the object (local references:
// slots of the thread's own handle arena, bumped by every function that returns a reference, rewound by PopLocalFrame
// and on return to Java; global references: slots of a VM-wide arena), so creating, deleting and resolving a
// reference cost a few nanoseconds like in a JVM -- r05's mock kept a std::multiset of live references and looked holder
// fields up in a std::map<std::string>, and the "marshalling" time it reported was mostly its own (NOTES 53) -- while
// everything -Xcheck:jni enforces (the reference's test JVMs run with it, build.gradle:101-104) is still checked on
// every call:
//  * no JNI call with an exception pending, except the handful the specification allows;
//  * at most 16 live local references in a frame without PushLocalFrame / EnsureLocalCapacity (a pushed frame: its capacity);
//  * DeleteLocalRef only of references that are live; no use of a deleted / popped reference;
//  * a JNIEnv only on its own thread; a local reference only on the thread that owns it (helper threads must attach
//    through the JavaVM and work on global references).
#include <dlfcn.h>
#include <sched.h>

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

constexpr int kMaxFields = 8;

// A Java-heap object: an 8-byte header with the payload right behind it, bump-allocated from the VM's heap chunks in the
// order the harness creates them -- a read's holder and its five byte[] lie next to each other like objects a Java thread
// allocates in its TLAB (the first slot-based mock of this round still kept every object as a ~280-byte struct of
// std::vectors, a 150 MB graph for the C2 batch whose cache and TLB misses were most of the "marshalling" time; this
// heap is 10 MB: docs/NOTES.md 54).
struct ClassInfo { std::string name; std::set<std::string> fields; };
struct Obj {
  enum Kind : uint32_t { CLASS, BYTES, DOUBLES, LONGS, INTS, OBJARRAY, HOLDER } kind;
  uint32_t n;   // array length; HOLDER: number of field slots (kMaxFields); CLASS: 1 (a ClassInfo*)
  template <typename T> T* data() { return reinterpret_cast<T*>(this + 1); }
  int8_t* bytes() { return data<int8_t>(); }
  double* doubles() { return data<double>(); }
  int64_t* longs() { return data<int64_t>(); }
  int32_t* ints() { return data<int32_t>(); }
  Obj** elems() { return data<Obj*>(); }     // OBJARRAY
  Obj** fields() { return data<Obj*>(); }    // HOLDER: by field index (jfieldID = index + 1); nullptr = Java null
  ClassInfo* cls() { return *data<ClassInfo*>(); }
};
static_assert(sizeof(Obj) == 8, "payload stays 8-byte aligned");

// interned field names: jfieldID k + 1 <-> name k (VM-independent, like a symbol table)
std::mutex g_names_mu;
std::vector<std::string> g_names;
int field_index(const char* name) {
  std::lock_guard<std::mutex> l(g_names_mu);
  for (size_t i = 0; i < g_names.size(); i++) if (g_names[i] == name) return (int)i;
  if ((int)g_names.size() >= kMaxFields) { fprintf(stderr, "mock_jni: more than %d field names\n", kMaxFields); abort(); }
  g_names.push_back(name);
  return (int)g_names.size() - 1;
}

// What a JNI function costs on top of the mock's own few nanoseconds: 0 by default; bench.py also measures with 25 ns -- the
// order of a HotSpot JNI function's native -> VM -> native transition -- to show what a real JVM's slower calls do to the call.
uint64_t g_call_cost_ticks = 0;

struct Mock;
struct MockVM {
  JavaVM_ vm;                  // must be first: JavaVM* == MockVM*
  JNIInvokeInterface_ itable;
  std::mutex mu;
  static constexpr size_t kChunk = (size_t)32 << 20;
  std::vector<std::unique_ptr<char[]>> chunks;   // the Java heap: bump-allocated
  size_t chunk_left = 0;
  char* chunk_at = nullptr;
  std::vector<std::unique_ptr<ClassInfo>> classes;
  std::vector<std::unique_ptr<Mock>> owned;   // every JNIEnv of this VM
  std::vector<Mock*> envs;
  static constexpr size_t kGlobalCap = 1 << 12;
  Obj** gslots;
  size_t gtop = 0;
  std::vector<size_t> gfree;
  std::atomic<long> violations{0}, attach_calls{0}, detach_calls{0}, globals_created{0}, globals_deleted{0};
  std::string first_violation;
  MockVM();
  ~MockVM();
  void violation(const std::string& what) {
    if (!violations.fetch_add(1)) { std::lock_guard<std::mutex> l(mu); first_violation = what; }
  }
  Obj* make(Obj::Kind k, size_t n, size_t payload_bytes) {   // zeroed
    const size_t want = (sizeof(Obj) + payload_bytes + 7) & ~(size_t)7;
    std::lock_guard<std::mutex> l(mu);
    if (want > chunk_left) {
      const size_t sz = std::max(kChunk, want);
      chunks.emplace_back(new char[sz]());
      chunk_at = chunks.back().get();
      chunk_left = sz;
    }
    Obj* o = reinterpret_cast<Obj*>(chunk_at);
    chunk_at += want; chunk_left -= want;
    o->kind = k; o->n = (uint32_t)n;
    return o;
  }
  Obj* make_bytes(const void* p, size_t n) { Obj* o = make(Obj::BYTES, n, n); if (p && n) memcpy(o->bytes(), p, n); return o; }
  Obj* make_doubles(size_t n, double fill) { Obj* o = make(Obj::DOUBLES, n, 8 * n); std::fill(o->doubles(), o->doubles() + n, fill); return o; }
  Obj* make_longs(const int64_t* p, size_t n) { Obj* o = make(Obj::LONGS, n, 8 * n); if (n) memcpy(o->longs(), p, 8 * n); return o; }
  Obj* make_ints(size_t n, int32_t fill) { Obj* o = make(Obj::INTS, n, 4 * n); std::fill(o->ints(), o->ints() + n, fill); return o; }
  Obj* make_array(size_t n) { return make(Obj::OBJARRAY, n, 8 * n); }
  Obj* make_holder() { return make(Obj::HOLDER, kMaxFields, 8 * kMaxFields); }
  Obj* make_class(const std::string& name, std::set<std::string> fields) {
    Obj* o = make(Obj::CLASS, 1, 8);
    std::lock_guard<std::mutex> l(mu);
    classes.emplace_back(new ClassInfo{name, std::move(fields)});
    *o->data<ClassInfo*>() = classes.back().get();
    return o;
  }
  Mock* new_env();
};

struct Mock {
  JNIEnv_ env;                 // must be first: JNIEnv* == Mock*
  JNINativeInterface_ table;
  MockVM* vm = nullptr;
  static constexpr size_t kCap = 1 << 20;   // slots of the local handle arena (virtual memory until touched)
  Obj** slots = nullptr;
  size_t top = 0;
  struct Frame { size_t base; long cap; long live; };
  std::vector<Frame> frames;
  std::thread::id owner;
  bool attached_by_library = false, detached = false;
  bool pending = false;
  std::string exc_class, exc_msg;
  long refs_created = 0, refs_deleted = 0, max_live_refs = 0, live_total = 0, jni_calls = 0, frames_pushed = 0;
  Mock() {
    slots = static_cast<Obj**>(calloc(kCap, sizeof(Obj*)));
    frames.push_back({0, 16, 0});
    owner = std::this_thread::get_id();
  }
  ~Mock() { free(slots); }
  Mock(const Mock&) = delete;
  void adopt() { owner = std::this_thread::get_id(); }   // (the harness makes an env on one thread and runs it on another)
  void violation(const std::string& what) { vm->violation(what); }
  void enter(const char* fn, bool pending_ok = false) {
    jni_calls++;
    if (g_call_cost_ticks) {   // mockjni_set_call_cost_ns: every JNI function costs at least this (a JVM's thread-state transition)
      const uint64_t until = __builtin_ia32_rdtsc() + g_call_cost_ticks;
      while (__builtin_ia32_rdtsc() < until) {}
    }
    if (std::this_thread::get_id() != owner) violation(std::string(fn) + ": JNIEnv used on a thread it does not belong to");
    if (detached) violation(std::string(fn) + ": JNIEnv of a detached thread");
    if (pending && !pending_ok) violation(std::string(fn) + " called with an exception pending (" + exc_class + ")");
  }
  jobject hand_out(Obj* o) {   // a new local reference in the current frame
    if (!o) return nullptr;
    if (top >= kCap) { fprintf(stderr, "mock_jni: local handle arena exhausted (references leaking?)\n"); abort(); }
    slots[top] = o;
    Frame& f = frames.back();
    if (++f.live > f.cap) violation("more live local references than the frame's capacity (" + std::to_string(f.cap) + ")");
    refs_created++;
    max_live_refs = std::max(max_live_refs, ++live_total);
    return reinterpret_cast<jobject>(&slots[top++]);
  }
  jobject arg(Obj* o) {   // an argument of the native method: a reference the caller's frame owns
    if (!o) return nullptr;
    slots[top] = o;
    return reinterpret_cast<jobject>(&slots[top++]);
  }
  void native_return() {   // back in Java: every local reference of the call is gone
    if (frames.size() > 1) violation("native method returned with a pushed local frame");
    for (const Frame& f : frames) { refs_deleted += f.live; live_total -= f.live; }
    frames.assign(1, Frame{0, 16, 0});
    top = 0;
  }
  void raise(const char* cls, const std::string& msg) { pending = true; exc_class = cls; exc_msg = msg; }
  Obj* deref(jobject h, const char* fn) {
    Obj** p = reinterpret_cast<Obj**>(h);
    if (p >= slots && p < slots + top) {
      if (!*p) { violation(std::string(fn) + ": deleted local reference used"); return nullptr; }
      return *p;
    }
    if (p >= vm->gslots && p < vm->gslots + MockVM::kGlobalCap) {
      if (!*p) { violation(std::string(fn) + ": deleted global reference used"); return nullptr; }
      return *p;
    }
    if (p >= slots + top && p < slots + kCap) { violation(std::string(fn) + ": local reference used after its frame was popped"); return *p; }
    Obj* foreign = nullptr;
    bool found = false;
    {
      std::lock_guard<std::mutex> l(vm->mu);
      for (Mock* o : vm->envs)
        if (p >= o->slots && p < o->slots + kCap) { found = true; foreign = *p; break; }
    }
    if (found) { violation(std::string(fn) + ": local reference used on a thread that does not own it"); return foreign; }
    fprintf(stderr, "mock_jni: %s got something that is not a JNI reference\n", fn);
    abort();
  }
};

MockVM::MockVM() { gslots = static_cast<Obj**>(calloc(kGlobalCap, sizeof(Obj*))); }
MockVM::~MockVM() { owned.clear(); free(gslots); }

Mock* M(JNIEnv* e) { return reinterpret_cast<Mock*>(e); }

void unimplemented() {
  fprintf(stderr, "mock_jni: the shim called a JNI function it does not declare\n");
  abort();
}

jclass m_FindClass(JNIEnv* e, const char* name) {
  M(e)->enter("FindClass");
  Obj* c = M(e)->vm->make_class(name, {});
  return reinterpret_cast<jclass>(M(e)->hand_out(c));
}
jint m_ThrowNew(JNIEnv* e, jclass c, const char* msg) {
  M(e)->enter("ThrowNew");
  Obj* o = M(e)->deref(c, "ThrowNew");
  M(e)->raise(o ? o->cls()->name.c_str() : "?", msg ? msg : "");
  return 0;
}
void m_ExceptionClear(JNIEnv* e) { M(e)->enter("ExceptionClear", true); M(e)->pending = false; }
jboolean m_ExceptionCheck(JNIEnv* e) { M(e)->enter("ExceptionCheck", true); return M(e)->pending ? JNI_TRUE : JNI_FALSE; }
void m_DeleteLocalRef(JNIEnv* e, jobject o) {  // allowed with an exception pending
  Mock* m = M(e);
  m->enter("DeleteLocalRef", true);
  if (!o) return;
  Obj** p = reinterpret_cast<Obj**>(o);
  if (p < m->slots || p >= m->slots + m->top || !*p) { m->violation("DeleteLocalRef of a reference that is not live"); return; }
  *p = nullptr;
  const size_t at = (size_t)(p - m->slots);
  for (size_t k = m->frames.size(); k-- > 0;)
    if (at >= m->frames[k].base) {
      if (m->frames[k].live > 0) { m->frames[k].live--; m->live_total--; m->refs_deleted++; }   // (an argument slot: not counted)
      break;
    }
}
jint m_PushLocalFrame(JNIEnv* e, jint capacity) {  // allowed with an exception pending
  Mock* m = M(e);
  m->enter("PushLocalFrame", true);
  m->frames.push_back({m->top, capacity, 0});
  m->frames_pushed++;
  return 0;
}
jobject m_PopLocalFrame(JNIEnv* e, jobject result) {  // allowed with an exception pending
  Mock* m = M(e);
  m->enter("PopLocalFrame", true);
  if (m->frames.size() < 2) { m->violation("PopLocalFrame without a pushed frame"); return nullptr; }
  Obj* keep = result ? m->deref(result, "PopLocalFrame") : nullptr;
  const Mock::Frame f = m->frames.back();
  m->frames.pop_back();
  m->refs_deleted += f.live;
  m->live_total -= f.live;
  for (size_t i = f.base; i < m->top; i++) m->slots[i] = nullptr;
  m->top = f.base;
  return keep ? m->hand_out(keep) : nullptr;
}
jobject m_NewGlobalRef(JNIEnv* e, jobject o) {
  Mock* m = M(e);
  m->enter("NewGlobalRef");
  Obj* t = o ? m->deref(o, "NewGlobalRef") : nullptr;
  if (!t) return nullptr;
  MockVM* vm = m->vm;
  std::lock_guard<std::mutex> l(vm->mu);
  size_t at;
  if (!vm->gfree.empty()) { at = vm->gfree.back(); vm->gfree.pop_back(); }
  else if (vm->gtop < MockVM::kGlobalCap) at = vm->gtop++;
  else { fprintf(stderr, "mock_jni: global references leaking\n"); abort(); }
  vm->gslots[at] = t;
  vm->globals_created++;
  return reinterpret_cast<jobject>(&vm->gslots[at]);
}
void m_DeleteGlobalRef(JNIEnv* e, jobject o) {  // allowed with an exception pending
  Mock* m = M(e);
  m->enter("DeleteGlobalRef", true);
  if (!o) return;
  MockVM* vm = m->vm;
  Obj** p = reinterpret_cast<Obj**>(o);
  if (p < vm->gslots || p >= vm->gslots + MockVM::kGlobalCap || !*p) { m->violation("DeleteGlobalRef of a reference that is not a live global one"); return; }
  std::lock_guard<std::mutex> l(vm->mu);
  *p = nullptr;
  vm->gfree.push_back((size_t)(p - vm->gslots));
  vm->globals_deleted++;
}
jint m_GetJavaVM(JNIEnv* e, JavaVM** out) {
  M(e)->enter("GetJavaVM");
  *out = &M(e)->vm->vm;
  return JNI_OK;
}
jfieldID m_GetFieldID(JNIEnv* e, jclass c, const char* name, const char* sig) {
  M(e)->enter("GetFieldID");
  Obj* cls = M(e)->deref(c, "GetFieldID");
  if (!cls || strcmp(sig, "[B") != 0 || !cls->cls()->fields.count(name)) {
    M(e)->raise("java/lang/NoSuchFieldError", name);
    return nullptr;
  }
  return reinterpret_cast<jfieldID>((intptr_t)field_index(name) + 1);
}
jobject m_GetObjectField(JNIEnv* e, jobject o, jfieldID f) {
  Mock* m = M(e);
  m->enter("GetObjectField");
  Obj* h = m->deref(o, "GetObjectField");
  const intptr_t k = reinterpret_cast<intptr_t>(f) - 1;
  if (!h || k < 0 || k >= kMaxFields) return nullptr;
  return m->hand_out(h->fields()[k]);
}
jsize m_GetArrayLength(JNIEnv* e, jarray a) {
  M(e)->enter("GetArrayLength");
  Obj* o = M(e)->deref(a, "GetArrayLength");
  if (!o) return 0;
  return (jsize)o->n;
}
jobject m_GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) {
  Mock* m = M(e);
  m->enter("GetObjectArrayElement");
  Obj* o = m->deref(a, "GetObjectArrayElement");
  if (!o || i < 0 || (uint32_t)i >= o->n) { m->raise("java/lang/ArrayIndexOutOfBoundsException", "element"); return nullptr; }
  return m->hand_out(o->elems()[i]);
}
bool region_ok(Mock* m, Obj* o, Obj::Kind kind, jsize start, jsize len, const char* what) {
  if (!o) return false;
  if (o->kind != kind) { m->violation(std::string(what) + " of an array of another type"); return false; }
  if (start < 0 || len < 0 || (size_t)start + (size_t)len > o->n) {
    m->raise("java/lang/ArrayIndexOutOfBoundsException", what);
    return false;
  }
  return true;
}
void m_GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize start, jsize len, jbyte* buf) {
  M(e)->enter("GetByteArrayRegion");
  Obj* o = M(e)->deref(a, "GetByteArrayRegion");
  if (region_ok(M(e), o, Obj::BYTES, start, len, "byte region")) memcpy(buf, o->bytes() + start, (size_t)len);
}
void m_SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize start, jsize len, const jbyte* buf) {
  M(e)->enter("SetByteArrayRegion");
  Obj* o = M(e)->deref(a, "SetByteArrayRegion");
  if (region_ok(M(e), o, Obj::BYTES, start, len, "byte region")) memcpy(o->bytes() + start, buf, (size_t)len);
}
void m_SetIntArrayRegion(JNIEnv* e, jintArray a, jsize start, jsize len, const jint* buf) {
  M(e)->enter("SetIntArrayRegion");
  Obj* o = M(e)->deref(a, "SetIntArrayRegion");
  if (region_ok(M(e), o, Obj::INTS, start, len, "int region")) memcpy(o->ints() + start, buf, sizeof(int32_t) * (size_t)len);
}
void m_SetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize start, jsize len, const jdouble* buf) {
  M(e)->enter("SetDoubleArrayRegion");
  Obj* o = M(e)->deref(a, "SetDoubleArrayRegion");
  if (region_ok(M(e), o, Obj::DOUBLES, start, len, "double region")) memcpy(o->doubles() + start, buf, sizeof(double) * (size_t)len);
}
jdoubleArray m_NewDoubleArray(JNIEnv* e, jsize len) {
  M(e)->enter("NewDoubleArray");
  return reinterpret_cast<jdoubleArray>(M(e)->hand_out(M(e)->vm->make_doubles((size_t)len, 0.0)));
}
void m_GetLongArrayRegion(JNIEnv* e, jlongArray a, jsize start, jsize len, jlong* buf) {
  M(e)->enter("GetLongArrayRegion");
  Obj* o = M(e)->deref(a, "GetLongArrayRegion");
  if (region_ok(M(e), o, Obj::LONGS, start, len, "long region")) memcpy(buf, o->longs() + start, sizeof(int64_t) * (size_t)len);
}

void install_table(Mock& m) {
  for (auto& s : m.table.slot) s = reinterpret_cast<void*>(&unimplemented);
  m.table.slot[kJniSlotFindClass] = (void*)&m_FindClass;
  m.table.slot[kJniSlotThrowNew] = (void*)&m_ThrowNew;
  m.table.slot[kJniSlotExceptionClear] = (void*)&m_ExceptionClear;
  m.table.slot[kJniSlotExceptionCheck] = (void*)&m_ExceptionCheck;
  m.table.slot[kJniSlotDeleteLocalRef] = (void*)&m_DeleteLocalRef;
  m.table.slot[kJniSlotPushLocalFrame] = (void*)&m_PushLocalFrame;
  m.table.slot[kJniSlotPopLocalFrame] = (void*)&m_PopLocalFrame;
  m.table.slot[kJniSlotNewGlobalRef] = (void*)&m_NewGlobalRef;
  m.table.slot[kJniSlotDeleteGlobalRef] = (void*)&m_DeleteGlobalRef;
  m.table.slot[kJniSlotGetJavaVM] = (void*)&m_GetJavaVM;
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

Mock* MockVM::new_env() {
  std::unique_ptr<Mock> m(new Mock());
  m->vm = this;
  install_table(*m);
  std::lock_guard<std::mutex> l(mu);
  envs.push_back(m.get());
  owned.push_back(std::move(m));
  return owned.back().get();
}

// ---- the invocation interface: threads the library starts itself attach here ----
thread_local Mock* t_attached = nullptr;   // the env this thread got from AttachCurrentThread*
jint vm_AttachCurrentThreadAsDaemon(JavaVM* v, void** penv, void*) {
  MockVM* vm = reinterpret_cast<MockVM*>(v);
  vm->attach_calls++;
  if (!t_attached || t_attached->vm != vm || t_attached->detached) {
    t_attached = vm->new_env();
    t_attached->attached_by_library = true;
  }
  *penv = &t_attached->env;
  return JNI_OK;
}
jint vm_DetachCurrentThread(JavaVM* v) {
  MockVM* vm = reinterpret_cast<MockVM*>(v);
  if (!t_attached || t_attached->vm != vm || t_attached->detached) { vm->violation("DetachCurrentThread on a thread that is not attached"); return JNI_EDETACHED; }
  vm->detach_calls++;
  if (t_attached->live_total != 0 || t_attached->frames.size() > 1) t_attached->native_return();   // (a detaching thread's references are released)
  t_attached->detached = true;
  t_attached = nullptr;
  return JNI_OK;
}
jint vm_GetEnv(JavaVM* v, void** penv, jint) {
  MockVM* vm = reinterpret_cast<MockVM*>(v);
  if (t_attached && t_attached->vm == vm && !t_attached->detached) { *penv = &t_attached->env; return JNI_OK; }
  *penv = nullptr;
  return JNI_EDETACHED;
}
void vm_unimplemented() {
  fprintf(stderr, "mock_jni: the shim called a JavaVM function it does not declare\n");
  abort();
}
void install_vm(MockVM& vm) {
  for (auto& s : vm.itable.slot) s = reinterpret_cast<void*>(&vm_unimplemented);
  vm.itable.slot[kJvmSlotDetachCurrentThread] = (void*)&vm_DetachCurrentThread;
  vm.itable.slot[kJvmSlotGetEnv] = (void*)&vm_GetEnv;
  vm.itable.slot[kJvmSlotAttachCurrentThreadAsDaemon] = (void*)&vm_AttachCurrentThreadAsDaemon;
  vm.vm.functions = &vm.itable;
}

Obj* bytes_obj(MockVM& vm, const uint8_t* p, int64_t n) { return vm.make_bytes(p, (size_t)n); }
void set_field(Obj* holder, const char* name, Obj* value) { holder->fields()[field_index(name)] = value; }

// totals over every env of the VM: [0] local refs handed out, [1] released (DeleteLocalRef, PopLocalFrame, return to
// Java), [2] -Xcheck:jni-style violations, [3] most local references live at once in one thread, [4] JNI function
// calls, [5] AttachCurrentThread* calls, [6] DetachCurrentThread calls, [7] global references made, [8] deleted,
// [9] JNI function calls made on threads the library attached itself, [10] local frames pushed
void collect_counters(MockVM& vm, long* c) {
  if (!c) return;
  for (int i = 0; i < 11; i++) c[i] = 0;
  for (auto& m : vm.owned) {
    c[0] += m->refs_created; c[1] += m->refs_deleted; c[3] = std::max(c[3], m->max_live_refs); c[4] += m->jni_calls;
    if (m->attached_by_library) c[9] += m->jni_calls;
    c[10] += m->frames_pushed;
  }
  c[2] = vm.violations; c[5] = vm.attach_calls; c[6] = vm.detach_calls; c[7] = vm.globals_created; c[8] = vm.globals_deleted;
}

typedef void (*init_fn)(JNIEnv*, jclass, jclass, jclass, jboolean, jint);
typedef void (*compute_fn)(JNIEnv*, jobject, jobjectArray, jobjectArray, jdoubleArray);
typedef void (*done_fn)(JNIEnv*, jobject);

// One native call = one trip out of Java: the arguments become references of the caller's frame, and every local
// reference is gone when the call returns.
struct PairHmmLib {
  void* h = nullptr;
  init_fn f_init = nullptr;
  compute_fn f_compute = nullptr;
  done_fn f_done = nullptr;
  bool load(const char* path, char* msg) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { snprintf(msg, 512, "dlopen: %s", dlerror()); return false; }
    f_init = (init_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative");
    f_compute = (compute_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative");
    f_done = (done_fn)dlsym(h, "Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative");
    if (!f_init || !f_compute || !f_done) { snprintf(msg, 512, "missing JNI symbol"); return false; }
    return true;
  }
  void init(Mock* m, Obj* read_cls, Obj* hap_cls, int use_double, int max_threads) const {
    f_init(&m->env, nullptr, (jclass)m->arg(read_cls), (jclass)m->arg(hap_cls), use_double ? JNI_TRUE : JNI_FALSE, max_threads);
    m->native_return();
  }
  void compute(Mock* m, Obj* reads, Obj* haps, Obj* lik) const {
    f_compute(&m->env, nullptr, (jobjectArray)m->arg(reads), (jobjectArray)m->arg(haps), (jdoubleArray)m->arg(lik));
    m->native_return();
  }
  void done(Mock* m) const { f_done(&m->env, nullptr); m->native_return(); }
};

Obj* read_class(MockVM& vm) {
  return vm.make_class("org/broadinstitute/gatk/nativebindings/pairhmm/ReadDataHolder", {"readBases", "readQuals", "insertionGOP", "deletionGOP", "overallGCP"});
}
Obj* hap_class(MockVM& vm) {
  return vm.make_class("org/broadinstitute/gatk/nativebindings/pairhmm/HaplotypeDataHolder", {"haplotypeBases"});
}

}  // namespace

extern "C" {

enum {
  MOCK_DROP_GCP_FIELD = 1,    // ReadDataHolder class lacks overallGCP -> initNative must throw IAE
  MOCK_NULL_READQUALS = 2,    // read 0 has readQuals == null
  MOCK_SKIP_INIT = 4,         // call compute without initNative
  MOCK_SHORT_QUALS = 8,       // read 0's insertionGOP is one byte short
  MOCK_NULL_READ_ELEMENT = 16, // readDataArray[0] == null
  MOCK_COMPUTE_AFTER_DONE = 32, // initNative, doneNative, THEN computeLikelihoodsNative (the reference keeps working)
  MOCK_REINIT_TWICE = 64,       // initNative again (same arguments, then the other precision and back) before computing
  MOCK_LAST_READ_BAD = 128,     // NULL_READQUALS / SHORT_QUALS / NULL_READ_ELEMENT hit the LAST read instead of read 0
  MOCK_PAUSE_AND_AGAIN = 256    // after the call: MOCKJNI_PAUSE_MS (default 700) of nothing -- an idle JVM -- then the same call again
};

// Returns 0 = ran without a Java exception, 1 = exception pending after initNative,
// 2 = after computeLikelihoodsNative, -1 = could not load / resolve the library.
// counters[11]: see collect_counters.  The first violation's text replaces the exception message when there is no exception.
int mockjni_run(const char* lib_path, int use_double, int max_threads, int n_reads, int n_haps,
                const int64_t* read_off, const int64_t* hap_off, const uint8_t* rb, const uint8_t* rq,
                const uint8_t* ri, const uint8_t* rd, const uint8_t* rc, const uint8_t* hb, double* out,
                int out_len, int flags, char* exc_class, char* exc_msg, long* counters) {
  exc_class[0] = exc_msg[0] = 0;
  PairHmmLib lib;
  if (!lib.load(lib_path, exc_msg)) return -1;

  MockVM vm;
  install_vm(vm);
  Mock* m = vm.new_env();
  t_attached = m;   // (a Java thread is attached: GetEnv on it answers)

  Obj* read_cls = read_class(vm);
  if (flags & MOCK_DROP_GCP_FIELD) read_cls->cls()->fields.erase("overallGCP");
  Obj* hap_cls = hap_class(vm);

  const int bad = (flags & MOCK_LAST_READ_BAD) ? n_reads - 1 : 0;
  Obj* reads = vm.make_array((size_t)n_reads);
  for (int r = 0; r < n_reads; r++) {
    const int64_t a = read_off[r], n = read_off[r + 1] - a;
    Obj* holder = vm.make_holder();
    set_field(holder, "readBases", bytes_obj(vm, rb + a, n));
    set_field(holder, "readQuals", (r == bad && (flags & MOCK_NULL_READQUALS)) ? nullptr : bytes_obj(vm, rq + a, n));
    set_field(holder, "insertionGOP", bytes_obj(vm, ri + a, (r == bad && (flags & MOCK_SHORT_QUALS)) ? n - 1 : n));
    set_field(holder, "deletionGOP", bytes_obj(vm, rd + a, n));
    set_field(holder, "overallGCP", bytes_obj(vm, rc + a, n));
    reads->elems()[r] = (r == bad && (flags & MOCK_NULL_READ_ELEMENT)) ? nullptr : holder;
  }
  Obj* haps = vm.make_array((size_t)n_haps);
  for (int k = 0; k < n_haps; k++) {
    Obj* holder = vm.make_holder();
    set_field(holder, "haplotypeBases", bytes_obj(vm, hb + hap_off[k], hap_off[k + 1] - hap_off[k]));
    haps->elems()[k] = holder;
  }
  Obj* likelihoods = vm.make_doubles((size_t)out_len, -12345.0);

  int rc_ = 0;
  if (!(flags & MOCK_SKIP_INIT)) {
    lib.init(m, read_cls, hap_cls, use_double, max_threads);
    if (m->pending) rc_ = 1;
    if (rc_ == 0 && (flags & MOCK_REINIT_TWICE)) {
      lib.init(m, read_cls, hap_cls, use_double, max_threads);
      lib.init(m, read_cls, hap_cls, !use_double, max_threads);
      lib.init(m, read_cls, hap_cls, use_double, max_threads);
      if (m->pending) rc_ = 1;
    }
  }
  if (rc_ == 0 && (flags & MOCK_COMPUTE_AFTER_DONE)) lib.done(m);
  if (rc_ == 0) {
    lib.compute(m, reads, haps, likelihoods);
    if (m->pending) rc_ = 2;
  }
  if (rc_ == 0 && (flags & MOCK_PAUSE_AND_AGAIN)) {
    const char* pv = getenv("MOCKJNI_PAUSE_MS");
    std::this_thread::sleep_for(std::chrono::milliseconds(pv ? atoi(pv) : 700));
    std::fill(likelihoods->doubles(), likelihoods->doubles() + likelihoods->n, -12345.0);
    lib.compute(m, reads, haps, likelihoods);
    if (m->pending) rc_ = 2;
  }
  // (a JVM would have the exception pending on return to Java; doneNative takes no JNI calls)
  lib.done(m);
  memcpy(out, likelihoods->doubles(), sizeof(double) * (size_t)out_len);
  if (m->pending) {
    snprintf(exc_class, 256, "%s", m->exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m->exc_msg.c_str());
  } else if (vm.violations) {
    snprintf(exc_msg, 512, "%s", vm.first_violation.c_str());
  }
  collect_counters(vm, counters);
  t_attached = nullptr;
  return rc_;
}


int g_warm_iters = 0;             // mockjni_set_warm_iters: untimed calls per thread in front of mockjni_run_concurrent's timed part
int64_t g_last_timing[6] = {0};   // the shim's call-time split over the timed part of the last mockjni_run_concurrent
long g_last_counters[11] = {0};   // collect_counters of the last mockjni_run_concurrent
struct CallRecord { double ms; int32_t thread, cpu_begin, cpu_end; double ghz_begin, ghz_end; };
std::vector<CallRecord> g_last_calls;   // every timed call of the last mockjni_run_concurrent
bool g_measure_clock = false;           // mockjni_measure_clock: sample the calling core's clock around every timed call
std::vector<int> g_affinity;            // mockjni_set_affinity: CPUs the caller threads are bound to (empty: wherever the scheduler puts them)
void mockjni_set_warm_iters(int n) { g_warm_iters = n < 0 ? 0 : n; }
// every JNI function of the mock takes at least `ns` nanoseconds from now on (0: off); returns the TSC ticks per nanosecond it measured
double mockjni_set_call_cost_ns(double ns) {
  static double ticks_per_ns = 0.0;
  if (ticks_per_ns == 0.0) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t c0 = __builtin_ia32_rdtsc();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 2000.0) {}
    const uint64_t c1 = __builtin_ia32_rdtsc();
    ticks_per_ns = (double)(c1 - c0) / std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  }
  g_call_cost_ticks = ns > 0 ? (uint64_t)(ns * ticks_per_ns) : 0;
  return ticks_per_ns;
}
void mockjni_measure_clock(int on) { g_measure_clock = on != 0; }
void mockjni_last_timing(int64_t* out) { for (int i = 0; i < 6; i++) out[i] = g_last_timing[i]; }
void mockjni_last_counters(long* out) { for (int i = 0; i < 11; i++) out[i] = g_last_counters[i]; }
void mockjni_set_affinity(const int* cpus, int n) { g_affinity.assign(cpus, cpus + (n > 0 ? n : 0)); }
// How fast the calling thread's core runs right now: a chain of 30 000 dependent adds (one per clock on any x86 core),
// ~10 us; returns adds per nanosecond = the core's effective clock in GHz at this moment.
double core_ghz_now() {
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t x = 1;
  for (int i = 0; i < 30000; i++) asm volatile("add %1, %0" : "+r"(x) : "r"(x | 1));
  const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  return x ? 30000.0 / ns : 0.0;
}
// ... of the last mockjni_run_concurrent's timed calls: the calling core's clock just before and just after each call
int mockjni_last_call_clocks(double* ghz_begin, double* ghz_end, int cap) {
  const int n = std::min<int>(cap, (int)g_last_calls.size());
  for (int i = 0; i < n; i++) { ghz_begin[i] = g_last_calls[i].ghz_begin; ghz_end[i] = g_last_calls[i].ghz_end; }
  return (int)g_last_calls.size();
}
// the timed calls of the last mockjni_run_concurrent: wall ms, caller thread, CPU at entry and at return; returns their number
int mockjni_last_calls(double* ms, int32_t* thread, int32_t* cpu_begin, int32_t* cpu_end, int cap) {
  const int n = std::min<int>(cap, (int)g_last_calls.size());
  for (int i = 0; i < n; i++) { ms[i] = g_last_calls[i].ms; thread[i] = g_last_calls[i].thread; cpu_begin[i] = g_last_calls[i].cpu_begin; cpu_end[i] = g_last_calls[i].cpu_end; }
  return (int)g_last_calls.size();
}

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
  PairHmmLib lib;
  if (!lib.load(lib_path, exc_msg)) return -1;

  MockVM vm;
  install_vm(vm);
  Mock* m = vm.new_env();
  t_attached = m;
  Obj* read_cls = read_class(vm);
  Obj* hap_cls = hap_class(vm);
  std::vector<Obj*> read_holders;
  for (int r = 0; r < n_reads; r++) {
    const int64_t a = read_off[r], n = read_off[r + 1] - a;
    Obj* holder = vm.make_holder();
    set_field(holder, "readBases", bytes_obj(vm, rb + a, n));
    set_field(holder, "readQuals", bytes_obj(vm, rq + a, n));
    set_field(holder, "insertionGOP", bytes_obj(vm, ri + a, n));
    set_field(holder, "deletionGOP", bytes_obj(vm, rd + a, n));
    set_field(holder, "overallGCP", bytes_obj(vm, rc + a, n));
    read_holders.push_back(holder);
  }
  Obj* haps = vm.make_array((size_t)n_haps);
  for (int k = 0; k < n_haps; k++) {
    Obj* holder = vm.make_holder();
    set_field(holder, "haplotypeBases", bytes_obj(vm, hb + hap_off[k], hap_off[k + 1] - hap_off[k]));
    haps->elems()[k] = holder;
  }
  lib.init(m, read_cls, hap_cls, use_double, max_threads);
  if (m->pending) {
    snprintf(exc_class, 256, "%s", m->exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m->exc_msg.c_str());
    t_attached = nullptr;
    return 1;
  }
  std::vector<Mock*> envs;
  std::vector<Obj*> slices, results;
  std::vector<int> first(n_threads + 1, 0);
  for (int t = 0; t < n_threads; t++) {
    envs.push_back(vm.new_env());
    first[t + 1] = (int)((int64_t)n_reads * (t + 1) / n_threads);
    Obj* arr = vm.make_array((size_t)(first[t + 1] - first[t]));
    for (int r = first[t]; r < first[t + 1]; r++) arr->elems()[r - first[t]] = read_holders[r];
    Obj* res = vm.make_doubles((size_t)(first[t + 1] - first[t]) * n_haps, -12345.0);
    slices.push_back(arr);
    results.push_back(res);
  }
  // g_warm_iters untimed calls per thread first (slots, engines and arenas exist afterwards), then all threads start
  // the timed part together; the shim's call-time split (gkl_pairhmm_jni_timing) is reset at that point
  typedef void (*timing_fn)(int64_t*, int);
  timing_fn f_timing = (timing_fn)dlsym(lib.h, "gkl_pairhmm_jni_timing");
  std::atomic<int> warmed{0};
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  std::mutex t0_mu;
  std::vector<std::vector<CallRecord>> records((size_t)n_threads);
  const char* spin_env = getenv("MOCKJNI_SPIN_BETWEEN_CALLS_US");
  const long spin_us = spin_env ? atol(spin_env) : 0;
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; t++)
    pool.emplace_back([&, t] {
      if (!g_affinity.empty()) {
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : g_affinity) CPU_SET(c, &set);
        sched_setaffinity(0, sizeof set, &set);
      }
      envs[t]->adopt();
      t_attached = envs[t];
      for (int k = 0; k < g_warm_iters && !envs[t]->pending; k++) lib.compute(envs[t], slices[t], haps, results[t]);
      if (warmed.fetch_add(1) + 1 == n_threads) {
        std::lock_guard<std::mutex> l(t0_mu);
        int64_t scratch[6];
        if (f_timing) f_timing(scratch, 1);
        t0 = std::chrono::steady_clock::now();
        warmed.fetch_add(n_threads);  // release
      }
      while (warmed.load() < 2 * n_threads) std::this_thread::yield();
      records[(size_t)t].reserve((size_t)iters);
      for (int k = 0; k < iters && !envs[t]->pending; k++) {
        const int c0 = sched_getcpu();
        const double g0 = g_measure_clock ? core_ghz_now() : 0.0;
        const auto a = std::chrono::steady_clock::now();
        lib.compute(envs[t], slices[t], haps, results[t]);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        records[(size_t)t].push_back({ms, t, c0, sched_getcpu(), g0, g_measure_clock ? core_ghz_now() : 0.0});
        if (spin_us > 0) {   // MOCKJNI_SPIN_BETWEEN_CALLS_US: a Java thread that keeps computing between its native calls
          const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
          while (std::chrono::steady_clock::now() < until) {}
        }
      }
      t_attached = nullptr;
    });
  // MOCKJNI_CHURN=1: meanwhile another IntelPairHmm instance of the same JVM comes and goes -- initNative with the same
  // arguments and doneNative, over and over (the reference's initNative only re-sets globals, its doneNative is empty)
  std::atomic<bool> stop{false};
  std::thread churn;
  Mock* churn_env = vm.new_env();
  const char* ch = getenv("MOCKJNI_CHURN");
  if (ch && *ch == '1')
    churn = std::thread([&] {
      churn_env->adopt();
      t_attached = churn_env;
      while (!stop.load()) {
        lib.done(churn_env);
        lib.init(churn_env, read_cls, hap_cls, use_double, max_threads);
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
      t_attached = nullptr;
    });
  for (auto& th : pool) th.join();
  stop.store(true);
  if (churn.joinable()) churn.join();
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (f_timing) f_timing(g_last_timing, 0);
  g_last_calls.clear();
  for (auto& v : records) g_last_calls.insert(g_last_calls.end(), v.begin(), v.end());
  if (churn_env->pending) { envs[0]->pending = true; envs[0]->exc_class = churn_env->exc_class; envs[0]->exc_msg = "churn thread: " + churn_env->exc_msg; }
  lib.done(m);
  int rc_ = 0;
  for (int t = 0; t < n_threads; t++) {
    memcpy(out + (size_t)first[t] * n_haps, results[t]->doubles(), sizeof(double) * results[t]->n);
    if (envs[t]->pending && rc_ == 0) {
      rc_ = 2;
      snprintf(exc_class, 256, "%s", envs[t]->exc_class.c_str());
      snprintf(exc_msg, 512, "%s", envs[t]->exc_msg.c_str());
    }
  }
  if (rc_ == 0 && vm.violations) snprintf(exc_msg, 512, "%s", vm.first_violation.c_str());
  collect_counters(vm, g_last_counters);
  t_attached = nullptr;
  return rc_;
}


// ---- PDHMM: IntelPDHMM natives (include/gkl_pdhmm_jni.h) ----
double g_last_pd_ms = 0.0, g_last_pd_jni_calls = 0.0;
void mockjni_last_pd_call(double* ms, double* jni_calls) { *ms = g_last_pd_ms; *jni_calls = g_last_pd_jni_calls; }
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
  MockVM vm;
  install_vm(vm);
  Mock* m = vm.new_env();
  t_attached = m;
  JNIEnv* env = &m->env;
  Obj* read_cls = read_class(vm);
  Obj* hap_cls = vm.make_class("HaplotypeDataHolder (PDHMM)", {"haplotypeBases", "haplotypePDBases"});
  if (flags & MOCKPD_DROP_PDBASES_FIELD) hap_cls->cls()->fields.erase("haplotypePDBases");
  int rc_ = 0;
  if (!(flags & MOCKPD_SKIP_INIT)) {
    f_init(env, nullptr, (jclass)m->arg(read_cls), (jclass)m->arg(hap_cls), 0, 1, 0, max_memory_mb);
    m->native_return();
    if (m->pending) rc_ = 1;
  }
  if (rc_ == 0 && (flags & MOCKPD_HOLDERS)) {
    const int n_reads = n_a, n_haps = n_b;
    Obj* reads = vm.make_array((size_t)n_reads);
    for (int r = 0; r < n_reads; r++) {
      Obj* holder = vm.make_holder();
      const int64_t n = read_len[r];
      set_field(holder, "readBases", bytes_obj(vm, rb + (int64_t)r * max_read, n));
      set_field(holder, "readQuals", bytes_obj(vm, rq + (int64_t)r * max_read, n));
      set_field(holder, "insertionGOP", bytes_obj(vm, ri + (int64_t)r * max_read, n));
      set_field(holder, "deletionGOP", bytes_obj(vm, rd + (int64_t)r * max_read, n));
      set_field(holder, "overallGCP", bytes_obj(vm, rc + (int64_t)r * max_read, n));
      reads->elems()[r] = holder;
    }
    Obj* haps = vm.make_array((size_t)n_haps);
    for (int k = 0; k < n_haps; k++) {
      Obj* holder = vm.make_holder();
      set_field(holder, "haplotypeBases", bytes_obj(vm, hb + (int64_t)k * max_hap, hap_len[k]));
      set_field(holder, "haplotypePDBases", bytes_obj(vm, hp + (int64_t)k * max_hap, hap_len[k]));
      haps->elems()[k] = holder;
    }
    Obj* lik = vm.make_doubles((size_t)out_len, -12345.0);
    f_cl(env, nullptr, (jobjectArray)m->arg(reads), (jobjectArray)m->arg(haps), (jdoubleArray)m->arg(lik));
    m->native_return();
    if (m->pending) rc_ = 2;
    memcpy(out, lik->doubles(), sizeof(double) * (size_t)out_len);
    // MOCKJNI_PD_ITERS=n: the same call n more times, timed one by one (mockjni_last_pd_call: median ms, JNI calls per call)
    const char* iv = getenv("MOCKJNI_PD_ITERS");
    if (rc_ == 0 && iv && atoi(iv) > 0) {
      std::vector<double> ms;
      const long calls0 = m->jni_calls;
      for (int k = 0; k < atoi(iv) && !m->pending; k++) {
        const auto a = std::chrono::steady_clock::now();
        f_cl(env, nullptr, (jobjectArray)m->arg(reads), (jobjectArray)m->arg(haps), (jdoubleArray)m->arg(lik));
        m->native_return();
        ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count());
      }
      if (!ms.empty()) {
        std::sort(ms.begin(), ms.end());
        g_last_pd_ms = ms[ms.size() / 2];
        g_last_pd_jni_calls = (double)(m->jni_calls - calls0) / (double)ms.size();
      }
      if (m->pending) rc_ = 2;
    }
  } else if (rc_ == 0) {
    const int batch = n_a;
    Obj* arrs[7];
    const uint8_t* src[7] = {hb, hp, rb, rq, ri, rd, rc};
    for (int i = 0; i < 7; i++) arrs[i] = bytes_obj(vm, src[i], (int64_t)batch * (i < 2 ? max_hap : max_read));
    Obj* hl = vm.make_longs(hap_len, (size_t)batch);
    Obj* rl = vm.make_longs(read_len, (size_t)batch);
    jdoubleArray res = f_flat(env, nullptr, (jbyteArray)m->arg(arrs[0]), (jbyteArray)m->arg(arrs[1]), (jbyteArray)m->arg(arrs[2]),
                              (jbyteArray)m->arg(arrs[3]), (jbyteArray)m->arg(arrs[4]), (jbyteArray)m->arg(arrs[5]), (jbyteArray)m->arg(arrs[6]),
                              (jlongArray)m->arg(hl), (jlongArray)m->arg(rl), batch, max_hap, max_read);
    Obj* ro = res ? m->deref(res, "return value") : nullptr;   // (the returned local reference, read before the frame goes)
    m->native_return();
    if (m->pending) rc_ = 2;
    else if (!ro) rc_ = 3;
    else memcpy(out, ro->doubles(), sizeof(double) * (size_t)std::min<size_t>(out_len, ro->n));
  }
  f_done(env, nullptr);
  m->native_return();
  if (m->pending) { snprintf(exc_class, 256, "%s", m->exc_class.c_str()); snprintf(exc_msg, 512, "%s", m->exc_msg.c_str()); }
  else if (vm.violations) snprintf(exc_msg, 512, "%s", vm.first_violation.c_str());
  t_attached = nullptr;
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
  MockVM vm;
  install_vm(vm);
  Mock* m = vm.new_env();
  t_attached = m;
  JNIEnv* env = &m->env;
  Obj* jref = bytes_obj(vm, ref, ref_len);
  Obj* jalt = bytes_obj(vm, alt, alt_len);
  Obj* jcig = vm.make_bytes(nullptr, (size_t)cigar_len);
  int rc_ = 0;
  if (!(flags & SW_SKIP_INIT)) {
    f_init(env, nullptr);
    m->native_return();
    if (m->pending) rc_ = 1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < iters && rc_ == 0; k++) {
    memset(jcig->bytes(), 0, jcig->n);  // the Java wrapper allocates a fresh array per call
    *offset = f_align(env, nullptr, (flags & SW_NULL_REF) ? nullptr : (jbyteArray)m->arg(jref), (jbyteArray)m->arg(jalt),
                      (jbyteArray)m->arg(jcig), match, mismatch, open, extend, (jbyte)strategy);
    m->native_return();
    if (m->pending) rc_ = 2;
  }
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  f_done(env, nullptr);
  m->native_return();
  memcpy(cigar_out, jcig->bytes(), (size_t)cigar_len);
  if (m->pending) {
    snprintf(exc_class, 256, "%s", m->exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m->exc_msg.c_str());
  } else if (vm.violations) snprintf(exc_msg, 512, "%s", vm.first_violation.c_str());
  t_attached = nullptr;
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
  MockVM vm;
  install_vm(vm);
  Mock* m = vm.new_env();
  t_attached = m;
  JNIEnv* env = &m->env;
  Obj* jrefs = bytes_obj(vm, refs, ref_off[n]);
  Obj* jalts = bytes_obj(vm, alts, alt_off[n]);
  Obj* jro = vm.make_longs(ref_off, (size_t)n + 1);
  Obj* jao = vm.make_longs(alt_off, (size_t)n + 1);
  Obj* jcig = vm.make_bytes(nullptr, (size_t)n * stride);
  Obj* joff = vm.make_ints((size_t)n, -777);
  int rc_ = 0;
  f_init(env, nullptr);
  m->native_return();
  if (m->pending) rc_ = 1;
  if (rc_ == 0) {
    *returned = f_batch(env, nullptr, (jbyteArray)m->arg(jrefs), (jlongArray)m->arg(jro), (jbyteArray)m->arg(jalts), (jlongArray)m->arg(jao),
                        (jbyteArray)m->arg(jcig), stride, (jintArray)m->arg(joff), match, mismatch, open, extend, (jbyte)strategy);
    m->native_return();
    if (m->pending) rc_ = 2;
  }
  f_done(env, nullptr);
  m->native_return();
  memcpy(cigars_out, jcig->bytes(), (size_t)n * stride);
  memcpy(offsets_out, joff->ints(), sizeof(int32_t) * (size_t)n);
  if (m->pending) {
    snprintf(exc_class, 256, "%s", m->exc_class.c_str());
    snprintf(exc_msg, 512, "%s", m->exc_msg.c_str());
  } else if (vm.violations) snprintf(exc_msg, 512, "%s", vm.first_violation.c_str());
  t_attached = nullptr;
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
  MockVM vm;
  install_vm(vm);
  JNIEnv* env = &vm.new_env()->env;
  out[0] = get(env, nullptr);
  out[1] = avx(env, nullptr); out[2] = avx2(env, nullptr); out[3] = avx512(env, nullptr);
  out[4] = omp(env, nullptr);
  set(env, nullptr, JNI_TRUE);
  out[5] = get(env, nullptr);
  set(env, nullptr, out[0] ? JNI_TRUE : JNI_FALSE);
  return 0;
}

// The checker checked: bit 0 = a local reference used on a thread that does not own it is flagged, bit 1 = a JNIEnv used
// on a foreign thread, bit 2 = a reference used after its frame was popped, bit 3 = a 17th live reference in the base frame.
int mockjni_selfcheck() {
  int got = 0;
  auto fresh = [](MockVM& vm) { install_vm(vm); return vm.new_env(); };
  {
    MockVM vm; Mock* a = fresh(vm); Mock* b = vm.new_env();
    Obj* arr = vm.make_array(0);
    jobject la = a->hand_out(arr);
    std::thread([&] { b->adopt(); gkljni::GetArrayLength(&b->env, (jarray)la); }).join();
    if (vm.violations == 1 && vm.first_violation.find("does not own it") != std::string::npos) got |= 1;
  }
  {
    MockVM vm; Mock* a = fresh(vm);
    Obj* arr = vm.make_array(0);
    jobject la = a->hand_out(arr);
    std::thread([&] { gkljni::GetArrayLength(&a->env, (jarray)la); }).join();
    if (vm.violations == 1 && vm.first_violation.find("does not belong") != std::string::npos) got |= 2;
  }
  {
    MockVM vm; Mock* a = fresh(vm);
    Obj* arr = vm.make_array(0);
    gkljni::PushLocalFrame(&a->env, 4);
    jobject la = a->hand_out(arr);
    gkljni::PopLocalFrame(&a->env, nullptr);
    gkljni::GetArrayLength(&a->env, (jarray)la);
    if (vm.violations == 1 && vm.first_violation.find("popped") != std::string::npos) got |= 4;
  }
  {
    MockVM vm; Mock* a = fresh(vm);
    Obj* arr = vm.make_array(0);
    for (int i = 0; i < 16; i++) a->hand_out(arr);
    const bool clean = vm.violations == 0;
    a->hand_out(arr);
    if (clean && vm.violations == 1) got |= 8;
  }
  return got;
}

// What the mock's own JNI functions cost: the per-read call sequence of the PairHMM shim (one holder, five fields of
// `len` bytes, block frames of 32 reads) over `n_reads` synthetic holders, nanoseconds per JNI call.  For scale only:
// HotSpot's functions do the same work behind a thread-state transition each.
double mockjni_selfbench(int n_reads, int len) {
  MockVM vm;
  install_vm(vm);
  Mock* m = vm.new_env();
  JNIEnv* env = &m->env;
  static const char* names[5] = {"readBases", "readQuals", "insertionGOP", "deletionGOP", "overallGCP"};
  std::vector<uint8_t> bytes((size_t)len, 7);
  Obj* reads = vm.make_array((size_t)n_reads);
  for (int r = 0; r < n_reads; r++) {
    Obj* holder = vm.make_holder();
    for (const char* n : names) set_field(holder, n, bytes_obj(vm, bytes.data(), len));
    reads->elems()[r] = holder;
  }
  jfieldID fid[5];
  for (int i = 0; i < 5; i++) fid[i] = reinterpret_cast<jfieldID>((intptr_t)field_index(names[i]) + 1);
  std::vector<jbyte> buf((size_t)len * 5);
  jobjectArray arr = (jobjectArray)m->arg(reads);
  const long calls0 = m->jni_calls;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r0 = 0; r0 < n_reads; r0 += 32) {
    gkljni::PushLocalFrame(env, 32 * 6);
    for (int r = r0; r < std::min(n_reads, r0 + 32); r++) {
      jobject holder = gkljni::GetObjectArrayElement(env, arr, r);
      jbyteArray f[5];
      for (int i = 0; i < 5; i++) f[i] = (jbyteArray)gkljni::GetObjectField(env, holder, fid[i]);
      const jsize n = gkljni::GetArrayLength(env, f[0]);
      for (int i = 0; i < 5; i++) gkljni::GetByteArrayRegion(env, f[i], 0, n, buf.data() + (size_t)i * len);
      if (gkljni::ExceptionCheck(env)) return -1.0;
    }
    gkljni::PopLocalFrame(env, nullptr);
  }
  const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  return ns / (double)(m->jni_calls - calls0);
}

}  // extern "C"
