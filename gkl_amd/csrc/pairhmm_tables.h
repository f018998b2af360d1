// Host-side lookup tables for the PairHMM kernels.
//
// The values must be bit-identical to the reference's Context<float> /
// Context<double> tables (reference src/main/native/pairhmm/Context.h:65-89,
// 133-148, 174-189), because the forward kernels are checked bit-for-bit against
// GKL's objects.  They are therefore built on the host with the host libm (pow,
// powf, log10, log1p), in the reference's operation order, and uploaded once per
// context; nothing here is evaluated with device math.
#pragma once
#include <cstdint>
#include <vector>

namespace gklhip {

constexpr int kQuals = 128;                         // quals are masked with &127
constexpr int kMmEntries = kQuals * (kQuals + 1) / 2;  // triangle (max,min), max<=127

template <typename T>
struct HostTables {
  std::vector<T> ph2pr;  // [128] 10^(-q/10)
  std::vector<T> div3;   // [128] ph2pr[q] / 3   (mismatch prior, template.h:183)
  std::vector<T> mm;     // [8256] matchToMatchProb[(max*(max+1))/2 + min]
  T initial_constant;    // 2^120 (float) / 2^1020 (double)
  T log10_initial;       // log10f / log10 of it
};

const HostTables<float>& host_tables_f32();
const HostTables<double>& host_tables_f64();

inline int mm_index(int ins, int del) {  // Context.h:156-167 with quals <= 127
  const int mx = ins > del ? ins : del, mn = ins > del ? del : ins;
  return ((mx * (mx + 1)) >> 1) + mn;
}

// base -> code, pairhmm_common.h:53-62: A0 C1 T2 G3 N4, every other byte 0.
inline uint8_t base_code(uint8_t b) {
  switch (b) { case 'C': return 1; case 'T': return 2; case 'G': return 3; case 'N': return 4; default: return 0; }
}

}  // namespace gklhip
