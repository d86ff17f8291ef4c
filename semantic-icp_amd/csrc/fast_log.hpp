// fast_log.hpp -- log(x) for finite x >= 1, the only arguments the robust losses produce (1 + s / a^2 and
// 1 + sqrt(s) / a^2; sqloss.h:13-18 and Ceres' CauchyLoss, em_icp.hpp:109-117).  Host + device, one source.
//
//   x = 2^k m, m in [1, 2);  j = the top 8 mantissa bits;  r = m inv_c_j - 1 (|r| <= 2^-8, exact in one fma:
//   inv_c_j has at most 10 significant bits);  log x = (k ln2 + logc_j) + log1p(r), log1p by its Taylor
//   polynomial of degree 7 (truncation 2^-59 relative).  Every term is >= 0 for x >= 1: no cancellation, and
//   entry 0 is {1, 0}, so x -> 1 keeps full relative accuracy.  About 1 ulp; 17 instructions on gfx950 against
//   ~35 for an fdlibm-style logarithm with its division (the accumulate kernel is bound by FP64 issue).
//   The table (log_table.inc, tools/make_log_table.py) is 4 KB; the accumulate kernel keeps it in LDS.
#ifndef SICP_FAST_LOG_HPP_
#define SICP_FAST_LOG_HPP_

#ifndef SICP_HD
#define SICP_HD
#endif

namespace sicp {

constexpr int kLogTableEntries = 256;

// byte offset of x's table entry (16 bytes per entry)
SICP_HD inline unsigned log_entry_offset(double x) {
  unsigned long long u;
  __builtin_memcpy(&u, &x, 8);
  return ((unsigned)(u >> 44) & 255u) << 4;
}

SICP_HD inline double log_from_entry(double x, double inv_c, double logc) {
  unsigned long long u;
  __builtin_memcpy(&u, &x, 8);
  const int k = (int)(unsigned)(u >> 52) - 1023;  // x >= 1: sign clear, exponent >= 1023
  u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double m;
  __builtin_memcpy(&m, &u, 8);
  const double r = __builtin_fma(m, inv_c, -1.0);
  double q = __builtin_fma(r, 1.0 / 7.0, -1.0 / 6.0);
  q = __builtin_fma(q, r, 1.0 / 5.0);
  q = __builtin_fma(q, r, -1.0 / 4.0);
  q = __builtin_fma(q, r, 1.0 / 3.0);
  q = __builtin_fma(q, r, -0.5);
  const double lp = __builtin_fma(r * r, q, r);
  return __builtin_fma((double)k, 6.93147180559945286227e-01, logc) + lp;
}

}  // namespace sicp
#endif
