#include "pairhmm_tables.h"

#include <algorithm>
#include <cmath>
#include <mutex>

namespace gklhip {
namespace {

// Jacobian-logarithm correction table: log10(1 + 10^(-k*1e-4)), k = 0..80000,
// computed in double and narrowed to T (Context.h:65-72).
constexpr int kJacSize = 80001;
constexpr double kJacStep = 0.0001;
constexpr double kJacTolerance = 8.0;

template <typename T>
struct Builder {
  std::vector<T> jac;

  Builder() : jac(kJacSize) {
    for (int k = 0; k < kJacSize; k++)
      jac[k] = static_cast<T>(std::log10(1.0 + std::pow(10.0, -static_cast<double>(k) * kJacStep)));
  }

  static int round_half_away(T d) {  // Context.h:91-94, in T
    return d > T(0) ? static_cast<int>(d + T(0.5)) : static_cast<int>(d - T(0.5));
  }

  // log10(10^a + 10^b) by table, all arithmetic in T (Context.h:96-122).
  T log10_sum(T a, T b) const {
    T lo = std::min(a, b), hi = std::max(a, b);
    if (std::isinf(lo) || std::isinf(hi)) return hi;
    const T diff = hi - lo;
    if (diff >= static_cast<T>(kJacTolerance)) return hi;
    const int idx = round_half_away(static_cast<T>(diff * static_cast<T>(1.0 / kJacStep)));
    return hi + jac[idx];
  }

  // P(match->match) for gap-open quals (i,j): 1 - 10^log10_sum, via log1p and the
  // reference's truncated 1/ln10 (Context.h:75-89).
  T match_to_match(int i, int j) const {
    const double inv_ln10 = 0.434294;
    const double s = log10_sum(static_cast<T>(-0.1) * static_cast<T>(i),
                               static_cast<T>(-0.1) * static_cast<T>(j));
    const double m = std::log1p(-std::min(1.0, std::pow(10, s))) * inv_ln10;
    return static_cast<T>(std::pow(10, m));
  }
};

template <typename T> T phred_to_prob(int q);
template <> float phred_to_prob<float>(int q) { return powf(10.f, -static_cast<float>(q) / 10.f); }      // Context.h:180
template <> double phred_to_prob<double>(int q) { return pow(10.0, -static_cast<double>(q) / 10.0); }    // Context.h:139

template <typename T>
HostTables<T> build(int scale_exp) {
  HostTables<T> t;
  Builder<T> b;
  t.ph2pr.resize(kQuals);
  t.div3.resize(kQuals);
  for (int q = 0; q < kQuals; q++) {
    t.ph2pr[q] = phred_to_prob<T>(q);
    t.div3[q] = t.ph2pr[q] / T(3.0);
  }
  t.mm.resize(kMmEntries);
  for (int i = 0; i < kQuals; i++)
    for (int j = 0; j <= i; j++) t.mm[((i * (i + 1)) >> 1) + j] = b.match_to_match(i, j);
  t.initial_constant = std::ldexp(T(1), scale_exp);
  t.log10_initial = std::log10(t.initial_constant);  // log10f for float (Context.h:184)
  return t;
}

}  // namespace

const HostTables<float>& host_tables_f32() {
  static const HostTables<float> t = build<float>(120);
  return t;
}
const HostTables<double>& host_tables_f64() {
  static const HostTables<double> t = build<double>(1020);
  return t;
}

}  // namespace gklhip
