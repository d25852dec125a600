// common.hip.h -- error plumbing and wavefront primitives shared by all kernels.
// gfx950 only: wave = 64 lanes, DPP wave shifts available (GFX9 family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "../../include/pymoc_hip.h"

namespace pm {

// ---------------------------------------------------------------- host errors
extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);

#define PM_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess)                                                         \
      return pm::fail(PM_EHIP, "%s failed: %s (%s:%d)", #call,                    \
                      hipGetErrorString(e_), __FILE__, __LINE__);                 \
  } while (0)

#define PM_REQUIRE(cond, ...)                                                     \
  do {                                                                            \
    if (!(cond)) return pm::fail(PM_EINVAL, __VA_ARGS__);                         \
  } while (0)

hipStream_t resolve_stream(pm_stream_t s);

// ------------------------------------------------------------- lane primitives
constexpr int WAVE = 64;

// DPP controls (GFX9): wave_shl:1 = 0x130 (lane i <- lane i+1),
//                      wave_shr:1 = 0x138 (lane i <- lane i-1).
// Lanes without a source keep their own value (bound_ctrl = 0, old = src).
__device__ __forceinline__ double from_next_lane(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double from_prev_lane(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Zero-filling variants: the lane without a source receives 0.0 (bound_ctrl), which lets the
// DPP move read its source register directly -- no preparatory copy.  For stencils whose
// edge lanes only need a FINITE neighbour value (K1: boundary levels advance with dt = 0).
__device__ __forceinline__ double from_next_lane_z(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double from_prev_lane_z(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------- exact division
// a / d with a precomputed y = RN(1/d) (itself from a true IEEE division): two
// Markstein correction steps.  q0 = RN(a*y) is within 1.5 ulp of a/d; after the first
// step q1 is faithful, and for a faithful q1 and correctly rounded y the second step
// returns the correctly rounded quotient (Markstein 1990; Muller et al., Handbook of
// Floating-Point Arithmetic, sec. 4.7).  Valid for finite operands whose quotient and
// residuals stay in the normal range -- every use in this library (buoyancy gradients,
// fluxes over areas) is ~100 binades away from the limits; non-finite inputs give NaN
// where IEEE division gives +-inf, which only matters for members already blown up.
// 5 FMA-class instructions instead of the ~12-instruction v_div_scale / v_rcp / v_fma /
// v_div_fmas / v_div_fixup sequence.  pm_selftest_fastdiv() checks it against `/`.
__device__ __forceinline__ double div_by_recip(double a, double d, double y) {
  double q = a * y;
  double r = __builtin_fma(-d, q, a);
  q = __builtin_fma(r, y, q);
  r = __builtin_fma(-d, q, a);
  q = __builtin_fma(r, y, q);
  return q;
}

// The same quotient in 4 instructions from the double-double reciprocal yh + yl of d:
// yh = RN(1/d) (true division), yl = recip_lo(d, yh) ~ 1/d - yh, so that yh + yl = (1/d)(1 + eta)
// with |eta| < 2^-104.  u = RN(a*yl) and q0 = RN(a*yh + u) give q0 = RN(x(1 + eta')) with
// x = a/d and |eta'| < 2^-103: q0 is within half an ulp + 2^-103 |x| of x, i.e. a FAITHFUL
// rounding of x (in fact RN(x) except when x lies within 2^-103 |x| of a midpoint).  The
// residual r = a - d*q0 of a faithful quotient is exact in one fma, and one Markstein
// correction with the correctly rounded yh then returns RN(a/d) (same theorem as above; the
// initial multiply-add is the constant-multiplication scheme of Brisebarre & Muller, IEEE TC
// 2008, used here only to obtain faithfulness).  Same validity range as div_by_recip.
__device__ __forceinline__ double recip_lo(double d, double yh) {
  const double e = __builtin_fma(-d, yh, 1.0);  // exact: |1 - d*yh| <= 2^-53, 53 bits suffice
  return e * yh;                                // (1/d - yh)(1 + O(2^-52))
}
__device__ __forceinline__ double div_by_recip2(double a, double d, double yh, double yl) {
  const double u = a * yl;
  double q = __builtin_fma(a, yh, u);
  const double r = __builtin_fma(-d, q, a);
  q = __builtin_fma(r, yh, q);
  return q;
}

// The same quotient in 3 instructions from RN(1/d) alone.  q0 may be 1.5 ulp off, so this is NOT
// covered by the theorem above for an arbitrary denominator -- but for a GIVEN d only the few
// numerators whose quotient lies within 3 * 2^-53 ulp of a rounding boundary could fail, and the
// host enumerates and tests them (pm_div3_proven, pymoc_hip.hip: div3_proof).  Used only for
// STATIC denominators that have passed that proof (PM_COLS_DIV3_PROVEN / PM_JN_DIV3_PROVEN) and
// whose device-side reciprocal the host has checked (pm_recip_check).
__device__ __forceinline__ double div_by_recip3(double a, double d, double yh) {
  double q = a * yh;
  const double r = __builtin_fma(-d, q, a);
  q = __builtin_fma(r, yh, q);
  return q;
}

// Operand window of the forms above.  With every input of a column step (state, forcing,
// coefficients, grid spacings, dt) either zero or of magnitude in [2^-200, 2^200], every
// quotient, product and residual of the step stays at least 2^-600 and at most 2^1000 in
// magnitude or is exactly zero, so no fma of the division sequences rounds into the subnormal
// range or overflows and the quotients are the IEEE ones.  Outside the window (and for
// non-finite operands: inf * 0 in the sequences gives NaN where IEEE division gives inf) the
// kernels take their IEEE-division path or flag the member; see col_inputs_in_fast_range.
constexpr int FAST_DIV_EXP = 200;
__device__ __forceinline__ bool in_fast_div_range(double x) {
  const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;  // biased exponent
  return (int)(e - (1023u - FAST_DIV_EXP) <= 2u * FAST_DIV_EXP) | (int)(x == 0.0);
}

// ... for kernels whose IEEE form is a follow-up launch (the fused JN2018 loop): a NON-FINITE operand
// does not send the member there.  Such a member is lost -- the reference raises at its next
// overturning update -- and is reported as non-finite; what its NaNs do until then is not defined.
__device__ __forceinline__ bool in_fast_div_range_or_lost(double x) {
  const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
  return (e >= 1023u - FAST_DIV_EXP && e <= 1023u + FAST_DIV_EXP) || x == 0.0 || e == 0x7ffu;
}

// Phase clocks for profiling builds (make -B lib EXTRA=-DPM_PHASE_PROFILE; read and cleared
// through pm_debug_prof).  PM_TICK(k) adds the cycles since the previous tick of this wave to
// slot k of a per-wave accumulator; a kernel that uses it declares `PM_TICK_INIT` once and ends
// with `PM_TICK_FLUSH` (a sample of the waves reports; slot 15 counts the reports).
#ifdef PM_PHASE_PROFILE
__device__ unsigned long long pm_prof[16];
// start / end of every wave on the constant 100 MHz clock (PM_WAVE_BEGIN / PM_WAVE_END(id)):
// the timeline shows whether a grid ran as one batch of resident waves or several
__device__ unsigned long long pm_wave_times[2 * 16384];
#define PM_WAVE_BEGIN const unsigned long long pm_t0 = wall_clock64();
#define PM_WAVE_END(id)                                             \
  if ((threadIdx.x & 63) == 0 && (id) < 16384) {                    \
    pm_wave_times[2 * (id)] = pm_t0;                                \
    pm_wave_times[2 * (id) + 1] = wall_clock64();                   \
  }
#define PM_TICK_INIT                                             \
  unsigned long long pm_acc[16] = {0};                           \
  unsigned long long pm_tprev = __builtin_readcyclecounter();
#define PM_TICK(k)                                                \
  {                                                               \
    const unsigned long long t_ = __builtin_readcyclecounter();   \
    pm_acc[k] += t_ - pm_tprev;                                   \
    pm_tprev = t_;                                                \
  }
#define PM_COUNT(k) pm_acc[k] += 1ull;
#define PM_TICK_FLUSH /* one wave in 64 reports (slot 15 counts them) */ \
  if (threadIdx.x == 0 && (blockIdx.x & 15) == 0) {               \
    for (int k_ = 0; k_ < 15; ++k_)                               \
      if (pm_acc[k_]) atomicAdd(&pm_prof[k_], pm_acc[k_]);        \
    atomicAdd(&pm_prof[15], 1ull);                                \
  }
#define PM_TICK_PARAM , unsigned long long &pm_tprev, unsigned long long (&pm_acc)[16]
#define PM_TICK_ARG , pm_tprev, pm_acc
#else
#define PM_WAVE_BEGIN
#define PM_WAVE_END(id)
#define PM_TICK_INIT
#define PM_TICK(k)
#define PM_COUNT(k)
#define PM_TICK_FLUSH
#define PM_TICK_PARAM
#define PM_TICK_ARG
#endif

// 64-bit mask of the G lanes of this lane's group inside the wave.
template <int G>
__device__ __forceinline__ unsigned long long group_mask(int lane) {
  if constexpr (G == 64) {
    return ~0ull;
  } else {
    return ((1ull << G) - 1ull) << ((lane / G) * G);
  }
}

// min / max over the G lanes of a group (butterfly; every lane gets the result)
template <int G>
__device__ __forceinline__ double group_min(double x) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    double y = __shfl_xor(x, o, WAVE);
    x = (y < x) ? y : x;
  }
  return x;
}
template <int G>
__device__ __forceinline__ double group_max(double x) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    double y = __shfl_xor(x, o, WAVE);
    x = (y > x) ? y : x;
  }
  return x;
}

// value of np.linspace(start, stop, num)[i]  (numpy/_core/function_base.py)
struct Linspace {
  double start, stop, delta, step;
  int num;
  __device__ __forceinline__ void init(double a, double b, int n) {
    start = a;
    stop = b;
    num = n;
    delta = b - a;
    step = (n > 1) ? delta / (double)(n - 1) : 0.;
  }
  __device__ __forceinline__ double at(int i) const {
    if (i == num - 1 && num > 1) return stop;
    if (num == 1) return start;
    if (step == 0.) return ((double)i / (double)(num - 1)) * delta + start;
    return (double)i * step + start;
  }
};

// np.interp(x, xp, fp) for sorted xp (LDS), single query
__device__ __forceinline__ double interp_sorted(double x, const double *xp, const double *fp,
                                                int n) {
  if (x != x) return x;
  if (n == 1) return (x < xp[0]) ? fp[0] : ((x > xp[0]) ? fp[n - 1] : fp[0]);
  if (x > xp[n - 1]) return fp[n - 1];
  if (x < xp[0]) return fp[0];
  int lo = 0, hi = n;  // upper bound: first index with xp > x
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (x >= xp[mid])
      lo = mid + 1;
    else
      hi = mid;
  }
  const int j = lo - 1;
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double r = slope * (x - xp[j]) + fp[j];
  if (r != r) {
    r = slope * (x - xp[j + 1]) + fp[j + 1];
    if (r != r && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

}  // namespace pm
