// libgkl_utils.so replacement (include/gkl_utils_jni.h; reference src/main/native/utils/utils.cc).
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <thread>
#if defined(__x86_64__)
#include <xmmintrin.h>
#endif

#include "../../include/gkl_utils_jni.h"

namespace {
bool gfx950_present() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return false; }
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) return true;
  }
  return false;
}
}  // namespace

extern "C" {

JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_getFlushToZeroNative(JNIEnv*, jobject) {
#if defined(__x86_64__)
  return _MM_GET_FLUSH_ZERO_MODE() == _MM_FLUSH_ZERO_ON ? JNI_TRUE : JNI_FALSE;
#else
  return JNI_FALSE;
#endif
}

JNIEXPORT void JNICALL Java_com_intel_gkl_IntelGKLUtils_setFlushToZeroNative(JNIEnv*, jobject, jboolean value) {
#if defined(__x86_64__)
  _MM_SET_FLUSH_ZERO_MODE(value ? _MM_FLUSH_ZERO_ON : _MM_FLUSH_ZERO_OFF);
#else
  (void)value;
#endif
}

JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_isAvxSupportedNative(JNIEnv*, jobject) {
  return gfx950_present() ? JNI_TRUE : JNI_FALSE;
}
JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_isAvx2SupportedNative(JNIEnv*, jobject) {
  return gfx950_present() ? JNI_TRUE : JNI_FALSE;
}
JNIEXPORT jboolean JNICALL Java_com_intel_gkl_IntelGKLUtils_isAvx512SupportedNative(JNIEnv*, jobject) { return JNI_FALSE; }

JNIEXPORT jint JNICALL Java_com_intel_gkl_IntelGKLUtils_getAvailableOmpThreadsNative(JNIEnv*, jobject) {
  const unsigned n = std::thread::hardware_concurrency();
  return (jint)(n ? n : 1);
}

}  // extern "C"
