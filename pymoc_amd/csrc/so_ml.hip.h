// so_ml.hip.h -- K5: Southern-Ocean mixed-layer buoyancy step; D1: the Jansen & Nadeau
// driver's per-step bottom-BC switch; and the fused per-member JN2018 time loop.
//
// Arithmetic restated from the reference (nothing copied):
//   SO_ML.set_boundary_conditions   src/pymoc/modules/SO_ML.py:77-98
//   SO_ML.calc_advective_tendency   src/pymoc/modules/SO_ML.py:100-134
//   SO_ML.calc_implicit_diffusion   src/pymoc/modules/SO_ML.py:136-196
//   SO_ML.advdiff / timestep        src/pymoc/modules/SO_ML.py:198-303
//   bottom-BC / kappa switching     examples/run_JansenNadeau_2018.py:233-254
//   time loop                       examples/run_JansenNadeau_2018.py:201-261
//
// One wavefront per member; the meridional profile (ny points) and the basin profiles
// (nz levels) are staged in LDS.  The Crank-Nicolson solve U x = V bs is the Thomas
// algorithm swept through LDS (the reference forms inv(U) densely with np.linalg.inv):
// same forward / backward recurrences, in index order, as the oracle.  U has constant
// coefficients, so the elimination factors cp[i] = c/den[i] and RN(1/den[i]) are tabulated
// once per launch; the per-row quotient uses the correctly rounded div_by_recip, so the
// sweep stays bit-identical to the oracle's plain divisions.
#pragma once
#include "column.hip.h"
#include "common.hip.h"

namespace pm {

constexpr int ML_WAVES_PER_BLOCK = 4;

// smallest index i in [0, n) with pred(s[i]) true, or n (wave-cooperative, LDS array)
template <class Pred>
__device__ __forceinline__ int wave_first_index(const double *s, int n, int lane, Pred pred) {
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const unsigned long long m = __ballot(i < n && pred(s[i]));
    if (m != 0ull) return i0 + (int)__ffsll((long long)m) - 1;
  }
  return n;
}

// LDS workspace of one member's mixed layer
struct MlLds {
  double *bb;    // [nz] b_basin
  double *pm;    // [nz] Psi_b -> Psi_mod
  double *bs;    // [ny]
  double *ps;    // [ny] Psi_s
  double *rhs;   // [ny]
  double *dp;    // [ny]
  double *cp;    // [ny] Thomas c'   (constant coefficients: tabulated once per launch)
  double *den;   // [ny] Thomas denominators
  double *rden;  // [ny] RN(1/den)
  double *f1;    // [ny] surflux/h                 (loop-invariant part of the flux, :250)
  double *f2;    // [ny] rest_mask*v_pist/h
  double *br;    // [ny] b_rest
  __device__ static int doubles(int nz, int ny) { return 2 * nz + 10 * ny; }
  __device__ void carve(double *base, int nz, int ny) {
    bb = base;
    pm = bb + nz;
    bs = pm + nz;
    ps = bs + ny;
    rhs = ps + ny;
    dp = rhs + ny;
    cp = dp + ny;
    den = cp + ny;
    rden = den + ny;
    f1 = rden + ny;
    f2 = f1 + ny;
    br = f2 + ny;
  }
};

// Psi_mod (SO_ML.py:228-230) in w.pm; returns the reference's IndexError condition.
// ind = first non-zero of Psi_b, first_pos = first Psi_b > 0 (:95).
__device__ __forceinline__ bool ml_prepare(const MlLds &w, int nz, int lane, int &first_pos) {
  const int ind = wave_first_index(w.pm, nz, lane, [](double v) { return v != 0.; });
  first_pos = wave_first_index(w.pm, nz, lane, [](double v) { return v > 0.; });
  if (ind >= nz) return false;
  const double fillv = w.pm[ind];
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < ind; i += 64) w.pm[i] = fillv;
  __builtin_amdgcn_wave_barrier();
  return true;
}

struct MlStatic {
  const double *surflux, *rest_mask, *b_rest;  // this member's rows in global memory
  double h, L, v_pist, dy, s;
  double rh, rL, rdy;  // RN(1/h), RN(1/L), RN(1/dy) for the correctly rounded divisions
  double rh_l, rL_l, rdy_l;  // their low parts (div_by_recip2)
};

// loop-invariant parts of the surface-flux tendency (:250-252), same operations as the
// reference: surflux/h and rest_mask*v_pist/h
__device__ __forceinline__ void ml_flux_tables(const MlLds &w, MlStatic &c, int ny, int lane) {
  for (int j = lane; j < ny; j += 64) {
    w.f1[j] = c.surflux[j] / c.h;
    w.f2[j] = c.rest_mask[j] * c.v_pist / c.h;
    w.br[j] = c.b_rest[j];
  }
  c.rh = 1.0 / c.h;
  c.rL = 1.0 / c.L;
  c.rdy = 1.0 / c.dy;
  c.rh_l = recip_lo(c.h, c.rh);
  c.rL_l = recip_lo(c.L, c.rL);
  c.rdy_l = recip_lo(c.dy, c.rdy);
  __builtin_amdgcn_wave_barrier();
}

// Thomas factors of U = tridiag(-s/2, 1+s, -s/2) with identity boundary rows (:155-165)
__device__ __forceinline__ void ml_tables(const MlLds &w, int ny, double s) {
  const double ta = -s / 2., tb = 1 + s, tc = -s / 2.;
  double cp = 0.;
  w.cp[0] = 0.;
  for (int i = 1; i < ny - 1; ++i) {
    const double den = tb - ta * cp;
    cp = tc / den;
    w.cp[i] = cp;
    w.den[i] = den;
    w.rden[i] = 1.0 / den;
  }
  __builtin_amdgcn_wave_barrier();
}

// The Crank-Nicolson system U x = V bs (:155-196; U = tridiag(-s/2, 1+s, -s/2) with identity
// boundary rows) depends only on s = Ks dt / dy^2: it is the same for every member and every
// step of a launch.  For ny <= 64 it is solved by PARALLEL CYCLIC REDUCTION with one row per
// lane: level l (stride k = 2^l) eliminates x[i-k] and x[i+k] from row i,
//     alpha = -a[i] / b[i-k],  gamma = -c[i] / b[i+k],
//     a'[i] = alpha a[i-k],  c'[i] = gamma c[i+k],  b'[i] = fma(gamma, a[i+k], fma(alpha, c[i-k], b[i])),
//     r'[i] = fma(gamma, r[i+k], fma(alpha, r[i-k], r[i])),
// and after 6 levels every row is decoupled: x[i] = r[i] / b[i].  The multipliers alpha, gamma
// and the final diagonal are tabulated once per block (PCR_ROWS x 64 doubles in LDS); a step is
// then 6 x (two lane shuffles of r + two fma) and one exact division -- no 51 x 51 propagator
// to stream through the CU's one LDS pipe (the dense form read 52 KB of LDS per wave and step
// and was bound by that pipe), no serial sweep.  Rows at and beyond ny are identity rows with
// r = 0; a missing neighbour has a zero multiplier.  The oracle runs the same recurrences in
// the same order (orc: so_ml_pcr_*), so the two stay bit-identical; against the reference's
// LAPACK inverse the step agrees to ~1e-15 like before.
constexpr int PCR_LEVELS = 6;
constexpr int PCR_ROWS = 2 * PCR_LEVELS + 2;  // alpha[6], gamma[6], b_final, RN(1/b_final)

__device__ __forceinline__ void ml_build_pcr(double *T, int ny, double s, int lane) {
  const bool interior = lane >= 1 && lane <= ny - 2;
  double a = interior ? -s / 2. : 0., b = interior ? 1 + s : 1., c = a;
#pragma unroll
  for (int l = 0; l < PCR_LEVELS; ++l) {
    const int k = 1 << l;
    const double a_lo = __shfl_up(a, k, 64), b_lo = __shfl_up(b, k, 64), c_lo = __shfl_up(c, k, 64);
    const double a_hi = __shfl_down(a, k, 64), b_hi = __shfl_down(b, k, 64),
                 c_hi = __shfl_down(c, k, 64);
    // no neighbour (lane - k < 0 / lane + k > 63): the shuffle returns the lane's own row, whose
    // b is non-zero, and the coupling a / c towards it is 0 by construction
    const double alpha = (lane >= k) ? -a / b_lo : 0.;
    const double gamma = (lane + k <= 63) ? -c / b_hi : 0.;
    T[l * 64 + lane] = alpha;
    T[(PCR_LEVELS + l) * 64 + lane] = gamma;
    const double bn = __builtin_fma(gamma, a_hi, __builtin_fma(alpha, c_lo, b));
    a = alpha * a_lo;
    c = gamma * c_hi;
    b = bn;
  }
  T[2 * PCR_LEVELS * 64 + lane] = b;
  T[(2 * PCR_LEVELS + 1) * 64 + lane] = 1.0 / b;
}

// x = U^-1 r for the row held by this lane (r = 0 on lanes >= ny)
__device__ __forceinline__ double ml_pcr_solve(double r, const double *T, int lane) {
#pragma unroll
  for (int l = 0; l < PCR_LEVELS; ++l) {
    const int k = 1 << l;
    const double r_lo = __shfl_up(r, k, 64), r_hi = __shfl_down(r, k, 64);
    r = __builtin_fma(T[(PCR_LEVELS + l) * 64 + lane], r_hi,
                      __builtin_fma(T[l * 64 + lane], r_lo, r));
  }
  return div_by_recip(r, T[2 * PCR_LEVELS * 64 + lane], T[(2 * PCR_LEVELS + 1) * 64 + lane]);
}

// One SO_ML.advdiff step on the member staged in `w` (bs, bb, pm valid; tables valid).
// Any ny (used for ny > 64; shorter profiles take ml_step_reg): ordered Thomas sweep.
// Returns false where the reference raises IndexError (state untouched).
__device__ __forceinline__ bool ml_step(const MlLds &w, const MlStatic &c, int nz, int ny,
                                        int lane, int first_pos, double dt) {
  // Psi_s = np.interp(bs, b_basin, Psi_mod) (:232)
  for (int j = lane; j < ny; j += 64) w.ps[j] = interp_sorted(w.bs[j], w.bb, w.pm, nz);
  // argmin(bs): first minimum, a NaN wins (np.argmin)
  double mn = __builtin_inf();
  int mi = 0x7fffffff;
  for (int j = lane; j < ny; j += 64) {
    const double v = w.bs[j];
    if (v < mn) {
      mn = v;
      mi = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(mn, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (ov < mn || (ov == mn && oi < mi)) {
      mn = ov;
      mi = oi;
    }
  }
  const int first_nan = wave_first_index(w.bs, ny, lane, [](double v) { return v != v; });
  const int amin = first_nan < ny ? first_nan : (mi < ny ? mi : 0);
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < ny; j += 64)
    if (j < amin || j == 0) w.ps[j] = 0.;  // :240-243
  __builtin_amdgcn_wave_barrier();

  const bool upwell = w.ps[1] > 0;  // set_boundary_conditions, :93-98
  if (upwell && first_pos >= nz) return false;
  const double bsouth = upwell ? w.bb[first_pos] : 0.;
  {
    const double v = upwell ? bsouth : w.bs[1];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) w.bs[0] = v;
    __builtin_amdgcn_wave_barrier();
  }
  // tendencies from surface flux / restoring and upwind advection (:250-259)
  for (int j = lane; j < ny; j += 64) {
    const double bsj = w.bs[j];
    const double flux = w.f1[j] + w.f2[j] * (w.br[j] - bsj);
    double adv = 0.;
    if (j >= 1 && j <= ny - 2) {
      const double ps = w.ps[j];
      // -Psi_s*1e6*(db)/h/L/dy (:128-133): the three divisions are correctly rounded
      // through the precomputed reciprocals, so the value equals the plain quotient chain
      double num = 0.;
      if (ps < 0.)
        num = -ps * 1e6 * (w.bs[j + 1] - bsj);
      else if (ps > 0.)
        num = -ps * 1e6 * (bsj - w.bs[j - 1]);
      if (ps != 0. && ps == ps)
        adv = div_by_recip(div_by_recip(div_by_recip(num, c.h, c.rh), c.L, c.rL), c.dy, c.rdy);
    }
    w.rhs[j] = bsj + dt * (flux + adv);  // staged: every tendency uses the old bs
  }
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < ny; j += 64) w.bs[j] = w.rhs[j];
  __builtin_amdgcn_wave_barrier();
  if (!upwell) {  // no-flux BC re-set (:264-266)
    const double v = w.bs[1];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) w.bs[0] = v;
    __builtin_amdgcn_wave_barrier();
  }
  // Crank-Nicolson diffusion (:191-196)
  const double s = c.s;
  // U x = V bs by the Thomas algorithm
  for (int j = lane; j < ny; j += 64) {
    double r;
    if (j == 0 || j == ny - 1)
      r = w.bs[j];
    else
      r = (s / 2.) * w.bs[j - 1] + (1 - s) * w.bs[j] + (s / 2.) * w.bs[j + 1];
    w.rhs[j] = r;
  }
  __builtin_amdgcn_wave_barrier();
  if (ny <= 64) {
    // Thomas sweep, every lane redundantly; lane i keeps row i's result in a register so the
    // loops contain no LDS stores and their (broadcast) loads pipeline ahead of the chain
    const double ta = -s / 2.;
    double dp = w.rhs[0];
    double mine = dp;  // lane 0: dp_0
#pragma unroll 4
    for (int i = 1; i < ny - 1; ++i) {
      dp = div_by_recip(w.rhs[i] - ta * dp, w.den[i], w.rden[i]);
      mine = (lane == i) ? dp : mine;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < ny - 1) w.dp[lane] = mine;
    __builtin_amdgcn_wave_barrier();
    double x = w.rhs[ny - 1];
    mine = x;  // lane ny-1
#pragma unroll 4
    for (int i = ny - 2; i >= 0; --i) {
      x = w.dp[i] - w.cp[i] * x;
      mine = (lane == i) ? x : mine;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < ny) w.bs[lane] = mine;
  } else {
    const double ta = -s / 2.;
    double dp = w.rhs[0];
    w.dp[0] = dp;
    for (int i = 1; i < ny - 1; ++i) {
      dp = div_by_recip(w.rhs[i] - ta * dp, w.den[i], w.rden[i]);
      w.dp[i] = dp;
    }
    double x = w.rhs[ny - 1];
    w.bs[ny - 1] = x;
    for (int i = ny - 2; i >= 0; --i) {
      x = w.dp[i] - w.cp[i] * x;
      w.bs[i] = x;
    }
  }
  __builtin_amdgcn_wave_barrier();
  {
    const double v = upwell ? bsouth : w.bs[1];  // final BC re-set (:274)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) w.bs[0] = v;
    __builtin_amdgcn_wave_barrier();
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// Register-resident mixed-layer step for ny <= 64 (the fused JN2018 loop): lane j holds point j
// of the meridional profile for the whole launch.  Same arithmetic, operation by operation, as
// ml_step above (which keeps the profile in LDS and serves any ny) -- the two are bit-identical
// (tests/test_so_ml_gpu.py: fused == stepwise) -- but a step makes 3-4 dependent LDS round
// trips instead of ~30: neighbours come by DPP wave shifts, the minimum by a DPP butterfly,
// scalars by v_readlane, and np.interp starts from the interval the point was in one step
// earlier (verified with the two table reads the interpolation needs anyway; a lane whose
// point left its interval searches again).  The wave spent half its life waiting on such round
// trips (profiles/r02/jn2018_steps_sq_counters.txt).
struct MlReg {
  double bs;          // bs[lane]
  double ps;          // Psi_s[lane] of the last step
  double f1, f2, br;  // surflux/h, rest_mask*v_pist/h, b_rest  (this lane's point)
  int jh;             // interval of the last interpolation: bb[jh] <= bs < bb[jh+1]
};

// minimum over the 64 lanes by DPP (quad swaps, half-row and row mirrors, then the four row
// results by v_readlane); NaNs are ignored (v_min_f64), every lane gets the result
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min_f64(double x) {
  x = __builtin_fmin(x, dpp_move<0xB1>(x));   // quad_perm [1,0,3,2]
  x = __builtin_fmin(x, dpp_move<0x4E>(x));   // quad_perm [2,3,0,1]
  x = __builtin_fmin(x, dpp_move<0x141>(x));  // row_half_mirror
  x = __builtin_fmin(x, dpp_move<0x140>(x));  // row_mirror: every lane holds its row's minimum
  const double r0 = lane_value(x, 0), r1 = lane_value(x, 16), r2 = lane_value(x, 32),
               r3 = lane_value(x, 48);
  return __builtin_fmin(__builtin_fmin(r0, r1), __builtin_fmin(r2, r3));
}

// tail of np.interp once the interval j is known (same expressions as interp_sorted)
__device__ __forceinline__ double interp_finish(double x, int j, double x0, double x1, double f0,
                                                double f1) {
  if (x0 == x) return f0;
  const double slope = (f1 - f0) / (x1 - x0);
  double r = slope * (x - x0) + f0;
  if (r != r) {
    r = slope * (x - x1) + f1;
    if (r != r && f0 == f1) r = f0;
  }
  return r;
}

// np.interp(x, xp, fp) with a hint: interval jh is tried first and updated.  The common cases --
// x still in its interval, x beyond either end of the table, x NaN -- are settled with selects
// from one batch of independent LDS reads; only a lane whose point moved to another interval
// searches (rare, divergent).
__device__ __forceinline__ double interp_hinted(double x, const double *xp, const double *fp,
                                                int n, int &jh, bool active) {
  const int j0 = jh < n - 1 ? jh : n - 2;
  const double x0 = xp[j0], x1 = xp[j0 + 1], f0 = fp[j0], f1 = fp[j0 + 1];
  const double xlo = xp[0], xhi = xp[n - 1], flo = fp[0], fhi = fp[n - 1];
  const bool hit = (x0 <= x) && (x < x1);  // => x is no NaN, inside the table, j0 = upper_bound-1
  double r = interp_finish(x, j0, x0, x1, f0, f1);
  // np.interp's order of tests: NaN, above the table, below the table
  const bool isnan_x = x != x, above = x > xhi, below = x < xlo;
  if (below) r = flo;
  if (above) r = fhi;
  if (isnan_x) r = x;
  const bool search = active && !hit && !isnan_x && !above && !below;
  if (__builtin_expect(__ballot(search) != 0ull, 0)) {
    if (search) {  // this lane's point left its interval: upper-bound search from scratch
      int lo = 0, hi = n;  // first index with xp > x
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (x >= xp[mid])
          lo = mid + 1;
        else
          hi = mid;
      }
      const int j = lo - 1;
      if (j == n - 1) {
        r = fp[j];
        jh = n - 2;
      } else {
        r = interp_finish(x, j, xp[j], xp[j + 1], fp[j], fp[j + 1]);
        jh = j;
      }
    }
  }
  return r;
}

// One SO_ML.advdiff step, ny <= 64, profile in registers.  w.bb / w.pm hold b_basin / Psi_mod,
// T the block's PCR tables.  Returns false where the reference raises IndexError (state
// untouched).
__device__ __forceinline__ bool ml_step_reg(MlReg &q, const MlLds &w, const MlStatic &c, int nz,
                                            int ny, int lane, int first_pos, double dt,
                                            const double *T) {
  const bool act = lane < ny;
  // Psi_s = np.interp(bs, b_basin, Psi_mod) (:232)
  double ps = interp_hinted(q.bs, w.bb, w.pm, nz, q.jh, act);
  // argmin(bs): first minimum, a NaN wins (np.argmin)
  const double v = act ? q.bs : __builtin_inf();
  const double mn = wave_min_f64(v);
  const unsigned long long at_min = __ballot(act && v == mn);
  const unsigned long long nanm = __ballot(act && v != v);
  const int mi = at_min ? (int)__ffsll((long long)at_min) - 1 : 0;
  const int amin = nanm ? (int)__ffsll((long long)nanm) - 1 : mi;
  if (lane < amin || lane == 0) ps = 0.;  // :240-243
  const bool upwell = lane_value(ps, 1) > 0;  // set_boundary_conditions, :93-98
  if (upwell && first_pos >= nz) return false;
  const double bsouth = upwell ? w.bb[first_pos] : 0.;
  double bs = q.bs;
  {
    const double v0 = upwell ? bsouth : lane_value(bs, 1);
    if (lane == 0) bs = v0;
  }
  // tendencies from surface flux / restoring and upwind advection (:250-259)
  const double bs_up = from_next_lane(bs), bs_dn = from_prev_lane(bs);
  const double flux = q.f1 + q.f2 * (q.br - bs);
  double adv = 0.;
  if (lane >= 1 && lane <= ny - 2) {
    double num = 0.;
    if (ps < 0.)
      num = -ps * 1e6 * (bs_up - bs);
    else if (ps > 0.)
      num = -ps * 1e6 * (bs - bs_dn);
    if (ps != 0. && ps == ps)
      adv = div_by_recip2(div_by_recip2(div_by_recip2(num, c.h, c.rh, c.rh_l), c.L, c.rL, c.rL_l),
                          c.dy, c.rdy, c.rdy_l);
  }
  bs = bs + dt * (flux + adv);  // every tendency uses the old bs
  if (!upwell) {  // no-flux BC re-set (:264-266)
    const double v1 = lane_value(bs, 1);
    if (lane == 0) bs = v1;
  }
  // Crank-Nicolson diffusion (:191-196): U x = V bs by parallel cyclic reduction
  double xi;
  {
    const double bl = from_prev_lane(bs), bu = from_next_lane(bs);
    const double sh = c.s / 2.;
    double r = bs;  // rows 0 and ny-1 of V are identity rows
    if (lane >= 1 && lane <= ny - 2) r = sh * bl + (1 - c.s) * bs + sh * bu;
    if (!act) r = 0.;
    xi = ml_pcr_solve(r, T, lane);
  }
  bs = xi;
  {
    const double v2 = upwell ? bsouth : lane_value(bs, 1);  // final BC re-set (:274)
    if (lane == 0) bs = v2;
  }
  q.bs = bs;
  q.ps = ps;
  return true;
}

#ifndef PM_SO_ML_DEVICE_FUNCTIONS_ONLY  // (jn2018_fast.hip shares the device functions only)
__global__ __launch_bounds__(64 * ML_WAVES_PER_BLOCK) void k_so_ml_step(pm_so_ml a, double dt) {
  extern __shared__ double lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m_raw = blockIdx.x * (blockDim.x >> 6) + wave;
  const bool m_ok = m_raw < a.n;
  const int m = m_ok ? m_raw : a.n - 1;
  const int nz = a.nz, ny = a.ny;
  MlLds w;
  w.carve(lds_all + (size_t)wave * MlLds::doubles(nz, ny), nz, ny);
  const size_t bz = (size_t)m * nz, by = (size_t)m * ny;

  for (int i = lane; i < nz; i += 64) {
    w.bb[i] = a.b_basin[bz + i];
    w.pm[i] = a.Psi_b[bz + i];
  }
  for (int j = lane; j < ny; j += 64) w.bs[j] = a.bs[by + j];
  __builtin_amdgcn_wave_barrier();

  int first_pos;
  MlStatic c;
  c.surflux = a.surflux + by;
  c.rest_mask = a.rest_mask + by;
  c.b_rest = a.b_rest + by;
  c.h = a.h;
  c.L = a.L;
  c.v_pist = a.v_pist;
  c.dy = a.y[1] - a.y[0];
  c.s = a.Ks * dt / (c.dy * c.dy);  // :191
  bool ok = ml_prepare(w, nz, lane, first_pos);
  const bool small = ny <= 64;  // block-uniform: register-resident step with PCR diffusion
  if (!small) ml_tables(w, ny, c.s);  // Thomas factors: only the ordered sweep reads them
  const double *T = nullptr;
  if (small) {
    double *Tw = lds_all + (size_t)(blockDim.x >> 6) * MlLds::doubles(nz, ny);
    if (wave == 0) ml_build_pcr(Tw, ny, c.s, lane);
    __syncthreads();
    T = Tw;
  }
  MlReg q;
  q.bs = q.ps = q.f1 = q.f2 = q.br = 0.;
  q.jh = 0;
  if (ok) {
    ml_flux_tables(w, c, ny, lane);
    if (small) {
      if (lane < ny) {
        q.bs = w.bs[lane];
        q.f1 = w.f1[lane];
        q.f2 = w.f2[lane];
        q.br = w.br[lane];
      }
      ok = ml_step_reg(q, w, c, nz, ny, lane, first_pos, dt, T);
    } else {
      ok = ml_step(w, c, nz, ny, lane, first_pos, dt);
    }
  }
  if (!ok) {  // IndexError in the reference: leave the state untouched
    if (a.status && lane == 0 && m_ok) a.status[m] = 1;
    return;
  }
  bool bad = false;
  if (small) {
    if (lane < ny) {
      bad |= !isfinite(q.bs);
      if (m_ok) {
        a.bs[by + lane] = q.bs;
        if (a.Psi_s) a.Psi_s[by + lane] = q.ps;
      }
    }
  } else {
    for (int j = lane; j < ny; j += 64) {
      const double v = w.bs[j];
      bad |= !isfinite(v);
      if (m_ok) {
        a.bs[by + j] = v;
        if (a.Psi_s) a.Psi_s[by + j] = w.ps[j];
      }
    }
  }
  if (a.status) {
    const int status = (__ballot(bad) != 0ull) ? 2 : 0;
    if (lane == 0 && m_ok) a.status[m] = status;
  }
}

#endif

// Bottom boundary condition / BBL diffusivity switching of run_JansenNadeau_2018.py:233-254.
// Columns are stored basin rows [0, n), north rows [n, 2n); coefficient set 0 = kappa,
// set 1 = kappaeff.  Scalar form shared by the stand-alone kernel and the fused loop.
struct BcState {
  double bbot_b, bbot_n;
  int ksel_b, ksel_n;
};
__device__ __forceinline__ void jn2018_bc(BcState &st, double PsiSO1, double Pb1, double Pn1,
                                          double bb0, double bb1, double bn0, double bn1,
                                          double bs0) {
  if (PsiSO1 < 0) {  // bottom water coming in from the south
    st.bbot_b = bs0;
    st.ksel_b = 1;
  }
  if (Pb1 > 0 && bn0 < bb1 && bn0 < bs0) {  // bottom water coming in from the north
    st.bbot_b = bn0;
    st.ksel_b = 1;
  } else if (PsiSO1 >= 0) {  // no bottom water coming in: no-flux BBC, full kappa
    st.bbot_b = bb1;
    st.ksel_b = 0;
  }
  if (Pn1 < 0 && bb0 < bn1) {  // bottom water coming in from the basin
    st.bbot_n = bb0;
    st.ksel_n = 1;
  } else {
    st.bbot_n = bn1;
    st.ksel_n = 0;
  }
}

#ifndef PM_SO_ML_DEVICE_FUNCTIONS_ONLY
__global__ void k_jn2018_bc_switch(pm_jn2018_bc a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= a.n) return;
  const size_t bz = (size_t)m * a.nz;
  BcState st;
  st.bbot_b = a.bbot[m];
  st.bbot_n = a.bbot[a.n + m];
  st.ksel_b = a.ksel[m];
  st.ksel_n = a.ksel[a.n + m];
  jn2018_bc(st, a.Psi_SO[bz + 1], a.Psi_res_b[bz + 1], a.Psi_res_n[bz + 1], a.b_basin[bz],
            a.b_basin[bz + 1], a.b_north[bz], a.b_north[bz + 1], a.bs_SO[(size_t)m * a.ny]);
  a.bbot[m] = st.bbot_b;
  a.bbot[a.n + m] = st.bbot_n;
  a.ksel[m] = st.ksel_b;
  a.ksel[a.n + m] = st.ksel_n;
}

#endif

// ---------------------------------------------------------------------------------------
// Fused JN2018 time loop: nsteps x [BC switch -> basin.timestep -> north.timestep ->
// channel.timestep] for one member per wavefront, wA / Psi_SO / Psibz held fixed (they only
// change at MOC updates).  Both columns live in registers (lane l owns levels [l*P, l*P+P)),
// the mixed layer in LDS; the scalars the BC switch needs are wave broadcasts -- no launch,
// no host round trip inside a MOC block.  Same device functions as the stand-alone kernels,
// so the result is bit-identical to stepping with pm_jn2018_bc_switch + pm_column_steps +
// pm_so_ml_step.
template <int P>
__device__ __forceinline__ void col_load_coef(ColRegs<P> &r, const pm_columns &c, int col,
                                              int sel, int lg) {
  const int nz = c.nz;
  const size_t sbase = ((size_t)sel * c.ncols + col) * nz;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lg * P + p;
    const int ic = i < nz ? i : nz - 1;
    r.kap[p] = c.kappa[sbase + ic];
    r.dAk[p] = c.dAkappa[sbase + ic];
  }
}

// SMALLNY (ny <= 64): the mixed layer lives in registers (ml_step_reg); otherwise in LDS.
// UA (pm_jn2018.hints & PM_JN_UNIFORM_AREA): Area constant in z -- scalar registers instead of
// 4 P vector registers per column, which pays for the low parts of the 4-instruction division.
template <int P, bool SMALLNY, bool UA>
__global__ __launch_bounds__(64 * ML_WAVES_PER_BLOCK, 2) void k_jn2018_steps(pm_jn2018 a,
                                                                          double dt,
                                                                          int nsteps) {
  extern __shared__ double lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m_raw = blockIdx.x * (blockDim.x >> 6) + wave;
  const bool m_ok = m_raw < a.n;
  const int m = m_ok ? m_raw : a.n - 1;
  const int n = a.n, nz = a.cols.nz, ny = a.ml.ny;
  PM_WAVE_BEGIN
  const pm_columns &c = a.cols;
  MlLds w;
  w.carve(lds_all + (size_t)wave * MlLds::doubles(nz, ny), nz, ny);
  const size_t bz = (size_t)m * nz, by = (size_t)m * ny;
  const int colb = m, coln = n + m;

  BcState st;
  st.bbot_b = c.bbot[colb];
  st.bbot_n = c.bbot[coln];
  st.ksel_b = c.ksel[colb];
  st.ksel_n = c.ksel[coln];
  ColGrid<P> g;
  ColRegs<P> rb, rn;
  constexpr int DV = UA ? 2 : 1;  // division form of the column steps
  col_load_grid<P, DV>(g, c, lane);
  col_load_static<P, DV>(rb, c, colb, st.ksel_b, lane);
  col_load_static<P, DV>(rn, c, coln, st.ksel_n, lane);
  bool hint_ok = true;
  if constexpr (UA) {
    rb.area_u = lane_value(rb.area[0], 0);
    rb.rarea_u = lane_value(rb.rarea[0], 0);
    rb.rarea_lu = lane_value(rb.rarea_l[0], 0);
    rn.area_u = lane_value(rn.area[0], 0);
    rn.rarea_u = lane_value(rn.rarea[0], 0);
    rn.rarea_lu = lane_value(rn.rarea_l[0], 0);
    bool same = true;
#pragma unroll
    for (int p = 0; p < P; ++p) same = same && rb.area[p] == rb.area_u && rn.area[p] == rn.area_u;
    hint_ok = __ballot(!same) == 0ull;  // padding lanes hold copies of the top level
  }
  // weff = wA - d(A kappa)/dz (column.py:241) is static between coefficient-set switches:
  // kept instead of wA and dAkappa (16 registers less for the two columns, 2P subtractions
  // less per step) and rebuilt when the BC switch changes a column's set
  double weff_b[P], weff_n[P];
  // kappa and weff of coefficient set `sel` (d(A kappa)/dz itself is not kept)
  auto load_coef = [&](ColRegs<P> &r, double (&weff)[P], int col, int sel) {
    const size_t sbase = ((size_t)sel * c.ncols + col) * nz;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      const int ic = i < nz ? i : nz - 1;
      r.kap[p] = c.kappa[sbase + ic];
      weff[p] = a.wA[(size_t)col * nz + ic] - c.dAkappa[sbase + ic];
    }
  };
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    const int ic = i < nz ? i : nz - 1;
    rb.b[p] = c.b[(size_t)colb * nz + ic];
    rn.b[p] = c.b[(size_t)coln * nz + ic];
  }
  load_coef(rb, weff_b, colb, st.ksel_b);
  load_coef(rn, weff_n, coln, st.ksel_n);
  const double bs_b = c.bs[colb], bs_n = c.bs[coln];
  const double N2_b = c.N2min[colb], N2_n = c.N2min[coln];
  const double PsiSO1 = a.Psi_SO[bz + 1], Pb1 = a.Psi_res_b[bz + 1], Pn1 = a.Psi_res_n[bz + 1];

  for (int i = lane; i < nz; i += 64) w.pm[i] = a.Psi_SO[bz + i];
  for (int j = lane; j < ny; j += 64) w.bs[j] = a.ml.bs[by + j];
  __builtin_amdgcn_wave_barrier();
  MlStatic mc;
  mc.surflux = a.ml.surflux + by;
  mc.rest_mask = a.ml.rest_mask + by;
  mc.b_rest = a.ml.b_rest + by;
  mc.h = a.ml.h;
  mc.L = a.ml.L;
  mc.v_pist = a.ml.v_pist;
  mc.dy = a.ml.y[1] - a.ml.y[0];
  mc.s = a.ml.Ks * dt / (mc.dy * mc.dy);
  int first_pos;
  bool ml_ok = ml_prepare(w, nz, lane, first_pos);
  if constexpr (!SMALLNY) ml_tables(w, ny, mc.s);  // Thomas factors: only the ordered sweep
  if constexpr (SMALLNY) {  // the block's PCR tables of the Crank-Nicolson system
    double *Tw = lds_all + (size_t)(blockDim.x >> 6) * MlLds::doubles(nz, ny);
    if (wave == 0) ml_build_pcr(Tw, ny, mc.s, lane);
    __syncthreads();
  }
  if (ml_ok) ml_flux_tables(w, mc, ny, lane);
  int status = ml_ok ? 0 : 1;
  MlReg q;
  q.bs = q.ps = q.f1 = q.f2 = q.br = 0.;
  q.jh = 0;
  bool ps_valid = false;
  if constexpr (SMALLNY) {
    if (lane < ny) {
      q.bs = w.bs[lane];
      if (ml_ok) {
        q.f1 = w.f1[lane];
        q.f2 = w.f2[lane];
        q.br = w.br[lane];
      }
    }
  }

  ConvCache<P> ccb, ccn;
#pragma unroll
  for (int p = 0; p < P; ++p) ccb.mask[p] = ccn.mask[p] = 0ull;
  ccb.zconv = ccn.zconv = 0.;
  // lanes / slots holding levels 0 and 1 of a column
  constexpr int L1 = 1 / P, S1 = 1 % P;
  for (int s = 0; s < (hint_ok ? nsteps : 0); ++s) {
    // ---- bottom-BC switch (run_JansenNadeau_2018.py:233-254)
    // levels 0 and 1 of both columns by v_readlane (scalar registers, no LDS round trip)
    const double bb0 = lane_value(rb.b[0], 0), bb1 = lane_value(rb.b[S1], L1);
    const double bn0 = lane_value(rn.b[0], 0), bn1 = lane_value(rn.b[S1], L1);
    const int kb = st.ksel_b, kn = st.ksel_n;
    const double bs0 = SMALLNY ? lane_value(q.bs, 0) : w.bs[0];
    jn2018_bc(st, PsiSO1, Pb1, Pn1, bb0, bb1, bn0, bn1, bs0);
    if (st.ksel_b != kb) load_coef(rb, weff_b, colb, st.ksel_b);
    if (st.ksel_n != kn) load_coef(rn, weff_n, coln, st.ksel_n);
    // ---- basin.timestep / north.timestep, do_conv=True (:257-258)
    col_convect_cached<P>(rb.b, g.z, bs_b, N2_b, lane, nz, ccb);
    col_vertadvdiff<64, P, DV, true, true, UA, false>(g, rb, weff_b, dt, true, bs_b, st.bbot_b,
                                                      false, 0., lane, nz);
    col_convect_cached<P>(rn.b, g.z, bs_n, N2_n, lane, nz, ccn);
    col_vertadvdiff<64, P, DV, true, true, UA, false>(g, rn, weff_n, dt, true, bs_n, st.bbot_n,
                                                      false, 0., lane, nz);
    // ---- channel.timestep(b_basin=basin.b, Psi_b=PsiSO.Psi) (:261)
    if (ml_ok) {
      // The workspace pointers are re-derived from an offset the optimiser cannot see through:
      // otherwise it hoists every LDS address of the mixed-layer step (one per array and
      // access pattern, ~60 vector registers) out of the time loop and keeps them live across
      // the column steps, which is what pushed this kernel into scratch spills.
      int woff = __builtin_amdgcn_readfirstlane(wave) * MlLds::doubles(nz, ny);
      int moff = (blockDim.x >> 6) * MlLds::doubles(nz, ny);
      asm volatile("" : "+s"(woff), "+s"(moff));
      MlLds w;
      w.carve(lds_all + woff, nz, ny);
      const double *T = lds_all + moff;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = lane * P + p;
        if (i < nz) w.bb[i] = rb.b[p];
      }
      __builtin_amdgcn_wave_barrier();
      bool stepped;
      if constexpr (SMALLNY)
        stepped = ml_step_reg(q, w, mc, nz, ny, lane, first_pos, dt, T);
      else
        stepped = ml_step(w, mc, nz, ny, lane, first_pos, dt);
      if (!stepped) {
        ml_ok = false;  // IndexError in the reference; the mixed layer stops evolving
        status = 1;
      } else {
        ps_valid = true;
      }
    }
  }

  bool bad = false;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    if (i < nz) {
      if (m_ok) {
        c.b[(size_t)colb * nz + i] = rb.b[p];
        c.b[(size_t)coln * nz + i] = rn.b[p];
      }
      bad |= !isfinite(rb.b[p]) || !isfinite(rn.b[p]);
    }
  }
  if constexpr (SMALLNY) {
    if (lane < ny) {
      bad |= !isfinite(q.bs);
      if (m_ok) {
        a.ml.bs[by + lane] = q.bs;
        // Psi_s of the last step that ran (like the stand-alone kernel, which leaves the array
        // untouched where the reference raises)
        if (a.ml.Psi_s && ps_valid) a.ml.Psi_s[by + lane] = q.ps;
      }
    }
  } else {
    for (int j = lane; j < ny; j += 64) {
      const double v = w.bs[j];
      bad |= !isfinite(v);
      if (m_ok) {
        a.ml.bs[by + j] = v;
        if (a.ml.Psi_s) a.ml.Psi_s[by + j] = w.ps[j];
      }
    }
  }
  const bool anybad = __ballot(bad) != 0ull;
  if (lane == 0 && m_ok) {
    const_cast<double *>(c.bbot)[colb] = st.bbot_b;
    const_cast<double *>(c.bbot)[coln] = st.bbot_n;
    const_cast<int32_t *>(c.ksel)[colb] = st.ksel_b;
    const_cast<int32_t *>(c.ksel)[coln] = st.ksel_n;
    if (c.nonfinite) {
      c.nonfinite[colb] = anybad ? 1 : 0;
      c.nonfinite[coln] = anybad ? 1 : 0;
    }
    if (a.ml.status) a.ml.status[m] = status | (anybad ? 2 : 0) | (hint_ok ? 0 : 16);
  }
  PM_WAVE_END(m_raw)
}

inline size_t ml_lds_bytes(int nz, int ny) {
  return (size_t)(2 * nz + 10 * ny) * sizeof(double);
}
// the block-shared PCR tables of the Crank-Nicolson step (ny <= 64)
inline size_t ml_prop_bytes(int ny) {
  return ny <= 64 ? (size_t)PCR_ROWS * 64 * sizeof(double) : 0;
}

#ifndef PM_SO_ML_DEVICE_FUNCTIONS_ONLY
inline int launch_so_ml(const pm_so_ml &a, double dt, hipStream_t st) {
  const size_t per_wave = ml_lds_bytes(a.nz, a.ny), prop = ml_prop_bytes(a.ny);
  int wpb = ML_WAVES_PER_BLOCK;
  while (wpb > 1 && per_wave * wpb + prop > 160 * 1024) wpb >>= 1;
  const size_t lds = per_wave * wpb + prop;
  if (lds > 160 * 1024) return fail(PM_EINVAL, "so_ml needs %zu B of LDS per member", lds);
  if (lds > 64 * 1024)
    PM_HIP(hipFuncSetAttribute((const void *)k_so_ml_step,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((a.n + wpb - 1) / wpb);
  hipLaunchKernelGGL(k_so_ml_step, dim3(grid), dim3(64 * wpb), lds, st, a, dt);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

#endif

template <int P>
int launch_jn2018_steps(const pm_jn2018 &a, double dt, int nsteps, hipStream_t st) {
  const size_t per_wave = ml_lds_bytes(a.cols.nz, a.ml.ny), prop = ml_prop_bytes(a.ml.ny);
  int wpb = ML_WAVES_PER_BLOCK;
  while (wpb > 1 && per_wave * wpb + prop > 160 * 1024) wpb >>= 1;
  const size_t lds = per_wave * wpb + prop;
  if (lds > 160 * 1024) return fail(PM_EINVAL, "jn2018 needs %zu B of LDS per member", lds);
  const unsigned grid = (unsigned)((a.n + wpb - 1) / wpb);
  const bool ua = (a.hints & PM_JN_UNIFORM_AREA) != 0;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      PM_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * wpb), lds, st, a, dt, nsteps);
    return PM_OK;
  };
  int rc;
  if (a.ml.ny <= 64)
    rc = ua ? go(k_jn2018_steps<P, true, true>) : go(k_jn2018_steps<P, true, false>);
  else
    rc = ua ? go(k_jn2018_steps<P, false, true>) : go(k_jn2018_steps<P, false, false>);
  if (rc != PM_OK) return rc;
  PM_HIP(hipGetLastError());
  return PM_OK;
}

// the residency-first rebuild of the fused loop (jn2018_fast.hip): uniform Area, ny <= 64
bool jn2018_fast_applies(const pm_jn2018 &a);
int launch_jn2018_fast(const pm_jn2018 &a, double dt, int nsteps, hipStream_t st);
// whole coupled runs in one launch (coupled_run.hip.h, compiled with jn2018_fast.hip)
int launch_twocol_run(const pm_twocol_loop &r, hipStream_t st);
int launch_jn2018_run(const pm_jn2018_loop &r, hipStream_t st);
size_t run_lds_bytes(int kind, int nz, int nb, int ny);
int launch_so_tw_update(const pm_psi_so &so, const pm_thermwind &tw, int tw_ops, hipStream_t st);

}  // namespace pm
