// so_ml.hip.h -- K5: Southern-Ocean mixed-layer buoyancy step, and the per-step bottom-BC
// switch of the Jansen & Nadeau driver.
//
// Arithmetic restated from the reference (nothing copied):
//   SO_ML.set_boundary_conditions   src/pymoc/modules/SO_ML.py:77-98
//   SO_ML.calc_advective_tendency   src/pymoc/modules/SO_ML.py:100-134
//   SO_ML.calc_implicit_diffusion   src/pymoc/modules/SO_ML.py:136-196
//   SO_ML.advdiff / timestep        src/pymoc/modules/SO_ML.py:198-303
//   bottom-BC / kappa switching     examples/run_JansenNadeau_2018.py:233-254
//
// One wavefront per member; the meridional profile (ny points) and the basin profiles
// (nz levels) are staged in LDS.  The Crank-Nicolson solve U x = V bs is the Thomas
// algorithm swept through LDS (the reference forms inv(U) densely with np.linalg.inv):
// same forward / backward recurrences, in index order, as the oracle.
#pragma once
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

__global__ __launch_bounds__(64 * ML_WAVES_PER_BLOCK) void k_so_ml_step(pm_so_ml a, double dt) {
  extern __shared__ double lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m_raw = blockIdx.x * (blockDim.x >> 6) + wave;
  const bool m_ok = m_raw < a.n;
  const int m = m_ok ? m_raw : a.n - 1;
  const int nz = a.nz, ny = a.ny;
  const int per_wave = 2 * nz + 5 * ny;
  double *s_bb = lds_all + (size_t)wave * per_wave;  // [nz] b_basin
  double *s_pm = s_bb + nz;                          // [nz] Psi_b -> Psi_mod
  double *s_bs = s_pm + nz;                          // [ny]
  double *s_ps = s_bs + ny;                          // [ny] Psi_s
  double *s_rhs = s_ps + ny;                         // [ny]
  double *s_cp = s_rhs + ny;                         // [ny]
  double *s_dp = s_cp + ny;                          // [ny]
  const size_t bz = (size_t)m * nz, by = (size_t)m * ny;

  for (int i = lane; i < nz; i += 64) {
    s_bb[i] = a.b_basin[bz + i];
    s_pm[i] = a.Psi_b[bz + i];
  }
  for (int j = lane; j < ny; j += 64) s_bs[j] = a.bs[by + j];
  __builtin_amdgcn_wave_barrier();

  // Psi_mod: fill below the first non-zero entry (SO_ML.py:228-230); first Psi_b > 0 (:95)
  const int ind = wave_first_index(s_pm, nz, lane, [](double v) { return v != 0.; });
  const int first_pos = wave_first_index(s_pm, nz, lane, [](double v) { return v > 0.; });
  int status = 0;
  if (ind >= nz) {  // IndexError in the reference: leave the state untouched
    if (a.status && lane == 0 && m_ok) a.status[m] = 1;
    return;
  }
  const double fillv = s_pm[ind];
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < ind; i += 64) s_pm[i] = fillv;
  __builtin_amdgcn_wave_barrier();

  // Psi_s = np.interp(bs, b_basin, Psi_mod) (:232)
  for (int j = lane; j < ny; j += 64) s_ps[j] = interp_sorted(s_bs[j], s_bb, s_pm, nz);
  // argmin(bs): first minimum, a NaN wins (np.argmin)
  double mn = __builtin_inf();
  int mi = 0x7fffffff;
  for (int j = lane; j < ny; j += 64) {
    const double v = s_bs[j];
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
  const int first_nan = wave_first_index(s_bs, ny, lane, [](double v) { return v != v; });
  const int amin = first_nan < ny ? first_nan : (mi < ny ? mi : 0);
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < ny; j += 64)
    if (j < amin || j == 0) s_ps[j] = 0.;  // :240-243
  __builtin_amdgcn_wave_barrier();

  const bool upwell = s_ps[1] > 0;  // set_boundary_conditions, :93-98
  if (upwell && first_pos >= nz) {
    if (a.status && lane == 0 && m_ok) a.status[m] = 1;  // IndexError in the reference
    return;
  }
  const double bsouth = upwell ? s_bb[first_pos] : 0.;
  {
    const double v = upwell ? bsouth : s_bs[1];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) s_bs[0] = v;
    __builtin_amdgcn_wave_barrier();
  }

  // tendencies from surface flux / restoring and upwind advection (:250-259)
  const double dy = a.y[1] - a.y[0];
  for (int j = lane; j < ny; j += 64) {
    const double bsj = s_bs[j];
    const double flux = a.surflux[by + j] / a.h +
                        a.rest_mask[by + j] * a.v_pist / a.h * (a.b_rest[by + j] - bsj);
    double adv = 0.;
    if (j >= 1 && j <= ny - 2) {
      const double ps = s_ps[j];
      if (ps < 0.)
        adv = -ps * 1e6 * (s_bs[j + 1] - bsj) / a.h / a.L / dy;
      else if (ps > 0.)
        adv = -ps * 1e6 * (bsj - s_bs[j - 1]) / a.h / a.L / dy;
    }
    s_rhs[j] = bsj + dt * (flux + adv);  // staged: every tendency uses the old bs
  }
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < ny; j += 64) s_bs[j] = s_rhs[j];
  __builtin_amdgcn_wave_barrier();
  if (!upwell) {  // no-flux BC re-set (:264-266)
    const double v = s_bs[1];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) s_bs[0] = v;
    __builtin_amdgcn_wave_barrier();
  }

  // Crank-Nicolson diffusion (:191-196): U x = V bs, U = tridiag(-s/2, 1+s, -s/2) with
  // identity boundary rows
  const double s = a.Ks * dt / (dy * dy);
  for (int j = lane; j < ny; j += 64) {
    double r;
    if (j == 0 || j == ny - 1)
      r = s_bs[j];
    else
      r = (s / 2.) * s_bs[j - 1] + (1 - s) * s_bs[j] + (s / 2.) * s_bs[j + 1];
    s_rhs[j] = r;
  }
  __builtin_amdgcn_wave_barrier();
  {
    const double ta = -s / 2., tb = 1 + s, tc = -s / 2.;
    double cp = 0., dp = s_rhs[0];
    s_cp[0] = cp;
    s_dp[0] = dp;
    for (int i = 1; i < ny - 1; ++i) {
      const double den = tb - ta * cp;
      cp = tc / den;
      dp = (s_rhs[i] - ta * dp) / den;
      s_cp[i] = cp;
      s_dp[i] = dp;
    }
    double x = s_rhs[ny - 1];
    s_bs[ny - 1] = x;
    for (int i = ny - 2; i >= 0; --i) {
      x = s_dp[i] - s_cp[i] * x;
      s_bs[i] = x;
    }
  }
  __builtin_amdgcn_wave_barrier();
  {
    const double v = upwell ? bsouth : s_bs[1];  // final BC re-set (:274)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) s_bs[0] = v;
    __builtin_amdgcn_wave_barrier();
  }
  bool bad = false;
  for (int j = lane; j < ny; j += 64) {
    const double v = s_bs[j];
    bad |= !isfinite(v);
    if (m_ok) {
      a.bs[by + j] = v;
      if (a.Psi_s) a.Psi_s[by + j] = s_ps[j];
    }
  }
  if (a.status) {
    status = (__ballot(bad) != 0ull) ? 2 : 0;
    if (lane == 0 && m_ok) a.status[m] = status;
  }
}

// Bottom boundary condition / BBL diffusivity switching of run_JansenNadeau_2018.py:233-254,
// one thread per member.  Columns are stored basin rows [0, n), north rows [n, 2n);
// coefficient set 0 = kappa, set 1 = kappaeff.
__global__ void k_jn2018_bc_switch(pm_jn2018_bc a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= a.n) return;
  const size_t bz = (size_t)m * a.nz;
  const double PsiSO1 = a.Psi_SO[bz + 1], Pb1 = a.Psi_res_b[bz + 1], Pn1 = a.Psi_res_n[bz + 1];
  const double bb0 = a.b_basin[bz], bb1 = a.b_basin[bz + 1];
  const double bn0 = a.b_north[bz], bn1 = a.b_north[bz + 1];
  const double bs0 = a.bs_SO[(size_t)m * a.ny];
  if (PsiSO1 < 0) {  // bottom water coming in from the south
    a.bbot[m] = bs0;
    a.ksel[m] = 1;
  }
  if (Pb1 > 0 && bn0 < bb1 && bn0 < bs0) {  // bottom water coming in from the north
    a.bbot[m] = bn0;
    a.ksel[m] = 1;
  } else if (PsiSO1 >= 0) {  // no bottom water coming in: no-flux BBC, full kappa
    a.bbot[m] = bb1;
    a.ksel[m] = 0;
  }
  if (Pn1 < 0 && bb0 < bn1) {  // bottom water coming in from the basin
    a.bbot[a.n + m] = bb0;
    a.ksel[a.n + m] = 1;
  } else {
    a.bbot[a.n + m] = bn1;
    a.ksel[a.n + m] = 0;
  }
}

inline int launch_so_ml(const pm_so_ml &a, double dt, hipStream_t st) {
  const size_t per_wave = (size_t)(2 * a.nz + 5 * a.ny) * sizeof(double);
  int wpb = ML_WAVES_PER_BLOCK;
  while (wpb > 1 && per_wave * wpb > 160 * 1024) wpb >>= 1;
  const size_t lds = per_wave * wpb;
  if (lds > 160 * 1024) return fail(PM_EINVAL, "so_ml needs %zu B of LDS per member", lds);
  if (lds > 64 * 1024)
    PM_HIP(hipFuncSetAttribute((const void *)k_so_ml_step,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((a.n + wpb - 1) / wpb);
  hipLaunchKernelGGL(k_so_ml_step, dim3(grid), dim3(64 * wpb), lds, st, a, dt);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

}  // namespace pm
