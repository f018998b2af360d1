// JNI_OnLoad device probe shared by the JNI drop-in libraries.
//
// The reference has no JNI_OnLoad: System.load succeeding is all that IntelPairHmm.load() checks
// (NativeLibraryLoader.java:99-140), and a machine without the required ISA is filtered earlier by the
// libgkl_utils.so AVX gate.  A GPU library needs the equivalent for "no usable gfx950 device": returning JNI_ERR
// makes System.load throw UnsatisfiedLinkError, NativeLibraryLoader.load() returns false, and GATK falls back to its
// Java implementation instead of failing later in initNative.  GKL_HIP_LOAD_WITHOUT_DEVICE=1 keeps the load
// succeeding (initNative then raises RuntimeException).  Include once per shared library.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

extern "C" JNIEXPORT jint JNICALL JNI_OnLoad(JavaVM*, void*) {
  const char* force = getenv("GKL_HIP_LOAD_WITHOUT_DEVICE");
  if (force && *force == '1') return JNI_VERSION_1_8;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return JNI_ERR; }
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) return JNI_VERSION_1_8;
  }
  return JNI_ERR;
}
