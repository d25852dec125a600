// jn2018_fast.hip -- the fused Jansen & Nadeau time loop, rebuilt for residency.
//
// Same work and the same arithmetic, operation by operation, as k_jn2018_steps (so_ml.hip.h):
//   nsteps x [bottom-BC switch -> basin.timestep(do_conv) -> north.timestep(do_conv) ->
//             channel.timestep]                   examples/run_JansenNadeau_2018.py:229-261
//   Column.convect / vertadvdiff                  src/pymoc/modules/column.py:210-271
//   SO_ML.advdiff                                 src/pymoc/modules/SO_ML.py:198-274
// for one member per wavefront, bit-identical to the stepwise launches.  What changed is where
// things live, so that FOUR waves fit a SIMD (<= 128 VGPRs) instead of two and a step issues
// about half the vector instructions:
//   * the grid metrics and their double-double reciprocals (7 tables) are block-shared LDS,
//     laid out so that a lane's slot pair is one conflict-free 16-byte read (k_jn2018_steps
//     kept 7P doubles of them in every lane);
//   * every wave-uniform double (bs, N2min, Area and its reciprocal parts, zconv, the mixed
//     layer's h, L, dy and their reciprocal parts, ...) sits in LDS and is read by broadcast
//     where it is used: as scalar registers they overflowed the scalar file, and the spills into
//     vector lanes came back as v_readlane instructions in the time loop;
//   * boundary and padding levels are held fixed by the TABLES (1/dzc = 0 and weff = 0 there
//     make their tendency an exact zero) instead of a per-slot dt select; padding levels hold a
//     large negative constant, so `b > bs` needs no validity test;
//   * the bottom-BC switch runs as vector code in lane 0, which owns levels 0 and 1 of both
//     columns and point 0 of the mixed layer: no readlanes, no scalar double juggling;
//   * the convective adjustment takes its branch-free cached form only when the cached pattern
//     holds convecting levels at all (the basin column almost never does);
//   * argmin(bs) of the mixed layer is 0 whenever no point lies below the first one (checked
//     with one compare); Psi_mod at a lane's interpolation interval stays in registers;
//   * rare events (a column switching its coefficient set) leave the step loop instead of
//     carrying their register merges through it; kernel arguments the loop does not need are
//     re-read from the argument segment where the rare paths use them.
// Requires: Area constant in z (PM_JN_UNIFORM_AREA, verified on the device), ny <= 64,
// 4 <= nz <= 256.  Everything else takes k_jn2018_steps.
// (profiling builds, -DPM_PHASE_PROFILE: the per-wave clocks live in pymoc_hip.hip's translation
// unit, which then includes this file; the stand-alone object is left empty)
#if !defined(PM_PHASE_PROFILE) || defined(PM_JF_IN_MAIN_TU)
#ifndef PM_JF_IN_MAIN_TU
#define PM_SO_ML_DEVICE_FUNCTIONS_ONLY
#define PM_DIAG_DEVICE_FUNCTIONS_ONLY
#endif
#include <type_traits>
#include "so_ml.hip.h"

namespace pm {

// One block of 16 waves per CU (4096 members = 256 blocks): the block's tables are shared by 16
// members, and the four waves of a SIMD are waves w, w+4, w+8, w+12 of ONE block, which is what
// the priority rotation in the step loop relies on.
#ifndef JF_WAVES_N
#define JF_WAVES_N 16
#endif
constexpr int JF_WAVES = JF_WAVES_N;
// profiling builds count how often the rare paths of the step loop are taken (pm_debug_prof)
#ifdef PM_PHASE_PROFILE
__device__ unsigned long long jf_rare[8 + 64];  // [8 + lane]: lanes whose point left its interval
#endif
#if defined(PM_PHASE_PROFILE) && defined(JF_COUNT_RARE)
#define JF_RARE(k) \
  if ((threadIdx.x & 63) == 0) atomicAdd(&jf_rare[k], 1ull);
#else
#define JF_RARE(k)
#endif
#ifndef JF_OCC_ATTR
#define JF_OCC_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif

// wave-uniform mixed-layer constants (block-shared), laid out in the pairs the three chained
// exact divisions num/h/L/dy and the Crank-Nicolson row read them in (one 16-byte access each)
enum { K_RHL = 0, K_RH, K_H, K_RLL, K_RL, K_L, K_RDYL, K_RDY, K_DY, K_SH, K_1MS, K_N = 12 };
// ... and per-member ones (per wave); a column's block is {bs, area | N2min, zc | rarea, rarea_l}
enum { S_BS = 0, S_AREA, S_N2, S_ZC, S_RAREA, S_RAREAL, S_COL = 6 };
enum { S_B = 0, S_NN = S_COL, S_BBOT0 = 2 * S_COL, S_N = 16 };

// Parallel-cyclic-reduction tables of the Crank-Nicolson system (ml_build_pcr's recurrences,
// so_ml.hip.h), laid out for 16-byte reads: level l: {alpha, gamma}[lane]; then {b_final,
// RN(1/b_final)}[lane]; then the low part of 1/b_final [lane]
constexpr int JF_PCR_DOUBLES = (PCR_LEVELS + 1) * 128 + 64;

template <int P>
struct JfLds {
  static constexpr int NZP = 64 * P;  // levels incl. padding
  // block-shared tables, entry of (lane, slot p): ((p / 2) * 64 + lane) * 2 + (p & 1)
  static constexpr int T_Z = 0, T_DZ = NZP, T_RDZ = 2 * NZP, T_RDZL = 3 * NZP, T_DZC = 4 * NZP,
                       T_RDZC = 5 * NZP, T_RDZCL = 6 * NZP;
  static constexpr int PCR = 7 * NZP;
  static constexpr int KML = PCR + JF_PCR_DOUBLES;
  static constexpr int PROG = KML + K_N;  // int[16]: the waves' step counters, [wave & 3][wave >> 2]
  static constexpr int WAVE0 = PROG + 8;
  // per wave: b_basin[level] and Psi_mod[level] for np.interp (level order: a lane's four levels
  // are two 16-byte stores; interleaving the two tables put the stores 8-way onto two banks),
  // kappa of the northern column in the tables' slot-pair layout (the basin's stays in
  // registers), {surflux/h, rest_mask*v_pist/h}[64], b_rest[64], Psi_s[64], the member's scalars
  static constexpr int W_BB = 0, W_PM = NZP, W_KN = 2 * NZP, W_F = 3 * NZP, W_BR = 3 * NZP + 128,
                       W_PS = 3 * NZP + 192, W_S = 3 * NZP + 256;
  static constexpr int PER_WAVE = W_S + S_N;
  static constexpr int TOTAL = WAVE0 + JF_WAVES * PER_WAVE;
};

template <int P>
__device__ __forceinline__ int jf_entry(int lane, int p) {
  return ((p >> 1) * 64 + lane) * 2 + (p & 1);
}

template <int P>
struct JfCol {
  double b[P];  // state; padding levels hold JF_PAD
  // weff = wA - d(A kappa)/dz (column.py:241), split for the select-free upwind flux of K1
  // (col_vertadvdiff FLUXFMA, column.hip.h): wn = -weff where weff < 0 (else 0), wp = -weff
  // where weff >= 0 (else 0); both 0 on boundary and padding levels
  double wn[P], wp[P];
  // kappa(z_i) of the coefficient set in use: registers for the basin column, the wave's LDS
  // rows for the northern one (KAPREG of jf_vertadvdiff); there is room for one of them
  double kap[P];
};
constexpr double JF_PAD = -1e300;  // finite, never above bs: padding never "convects"

// what stays in scalar registers of a column's convective state
template <int P>
struct JfConv {
  unsigned long long cm[P];  // convecting pattern the cached zc belongs to
  bool any;                  // the cached pattern has convecting levels
  bool valid;                // a pattern has been established in this launch
};

// A lane's P consecutive levels of a row in HBM.  VEC (P = 2 or 4, rows of a multiple of P
// levels, 16-byte aligned: decided by the launcher): 16-byte loads per lane instead of 8-byte
// loads with stride 8 P (a launch's memory-bound prologue: 170 -> 162 us per 36-step launch);
// lanes past the last level re-read the last P levels (callers mask them).
template <int P, bool VEC>
__device__ __forceinline__ void jf_load_row(double (&out)[P], const double *__restrict__ row,
                                            int lane, int nz) {
  if constexpr (VEC && (P == 4 || P == 2)) {
    const int i0 = lane * P < nz - P ? lane * P : nz - P;
#pragma unroll
    for (int h = 0; h < P / 2; ++h) {
      const double2 v = *reinterpret_cast<const double2 *>(row + i0 + 2 * h);
      out[2 * h] = v.x;
      out[2 * h + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      out[p] = row[i < nz ? i : nz - 1];
    }
  }
}

template <int P, bool VEC>
__device__ __forceinline__ void jf_store_row(double *__restrict__ row, const double (&v)[P],
                                             int lane, int nz) {
  if constexpr (VEC && (P == 4 || P == 2)) {
    if (lane * P < nz) {
#pragma unroll
      for (int h = 0; h < P / 2; ++h)
        *reinterpret_cast<double2 *>(row + lane * P + 2 * h) = make_double2(v[2 * h], v[2 * h + 1]);
    }
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      if (i < nz) row[i] = v[p];
    }
  }
}

// two slots of a table: one 16-byte LDS read
__device__ __forceinline__ double2 jf_pair(const double *T, int lane, int h) {
  return *reinterpret_cast<const double2 *>(T + (h * 64 + lane) * 2);
}

// the kernel's argument block, re-read where a rare path needs it (so that the pointers do not
// occupy scalar registers through the time loop)
typedef const __attribute__((address_space(4))) pm_jn2018 *jf_kargs;
__device__ __forceinline__ jf_kargs jf_args() {
  jf_kargs p = (jf_kargs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// Column.convect (column.py:251-271), do_conv = True, for a wave-owned column; ws = the
// column's scalar block in LDS.
template <int P>
__device__ __forceinline__ void jf_convect(double (&b)[P], JfConv<P> &s, const double *lds,
                                           double *ws, int lane, int nz) {
  using L = JfLds<P>;
  const double bs = ws[S_BS];
  unsigned long long m[P], acc = 0ull;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    m[p] = __builtin_amdgcn_ballot_w64(b[p] > bs);  // column.py:264 (padding: never)
    acc |= m[p] ^ s.cm[p];
  }
  if (__builtin_expect(acc != 0ull || !s.valid, 0)) {
    JF_RARE(1)
    // new pattern: zc = max(z[~ind]) -- z ascends, so z at the highest non-convecting level,
    // the bottom of the ocean if every level convects (column.py:265-267)
    unsigned long long anym = 0ull;
    int jmax = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      anym |= m[p];
      const int nv = (nz - p + P - 1) / P;  // lanes whose slot p is a real level
      const unsigned long long vmask = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
      const unsigned long long nm = ~m[p] & vmask;
      if (nm != 0ull) {
        const int j = (63 - __clzll((long long)nm)) * P + p;
        jmax = j > jmax ? j : jmax;
      }
      s.cm[p] = m[p];
    }
    s.any = anym != 0ull;
    s.valid = true;
    if (s.any) {
      const double zv = lds[L::T_Z + jf_entry<P>(jmax / P, jmax % P)];
      if (lane == 0) ws[S_ZC] = zv;
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (lane * P + p == nz - 1) b[p] = bs;  // column.py:271
    }
  }
  if (s.any) {
    JF_RARE(6)
    const double2 nc = *reinterpret_cast<const double2 *>(ws + S_N2);  // {N2min, zc}
#pragma unroll
    for (int h = 0; h < P / 2; ++h) {
      const double2 z2 = jf_pair(lds + L::T_Z, lane, h);
      const double adj0 = bs + nc.x * (z2.x - nc.y);  // column.py:268
      const double adj1 = bs + nc.x * (z2.y - nc.y);
      b[2 * h] = (b[2 * h] > bs) ? adj0 : b[2 * h];
      b[2 * h + 1] = (b[2 * h + 1] > bs) ? adj1 : b[2 * h + 1];
    }
  }
}

// Column.vertadvdiff (column.py:210-249) of one column, the bottom value already imposed.
// Same operations in the same order as col_vertadvdiff<64, P, 2, ..., UA> (column.hip.h).
// D3 (PM_JN_DIV3_PROVEN): the three quotients in 3 instructions each (div_by_recip3) -- the low
// parts of the reciprocals are then not even read.
template <int P, bool KAPREG, bool D3 = false>
__device__ __forceinline__ void jf_vertadvdiff(JfCol<P> &c, const double *lds, const double *ws,
                                               const double *kap, int lane, double dt) {
  using L = JfLds<P>;
  double bz[P];  // (b[i+1]-b[i])/dz[i] (column.py:235)
  const double nb0 = from_next_lane_z(c.b[0]);
#pragma unroll
  for (int h = 0; h < P / 2; ++h) {
    const double2 dz = jf_pair(lds + L::T_DZ, lane, h), y = jf_pair(lds + L::T_RDZ, lane, h);
    double2 yl = make_double2(0., 0.);
    if constexpr (!D3) yl = jf_pair(lds + L::T_RDZL, lane, h);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = 2 * h + q;
      const double up = (p < P - 1) ? c.b[p + 1 < P ? p + 1 : p] : nb0;
      // 1/dz = 0 at and above the top level
      if constexpr (D3)
        bz[p] = div_by_recip3(up - c.b[p], q ? dz.y : dz.x, q ? y.y : y.x);
      else
        bz[p] = div_by_recip2(up - c.b[p], q ? dz.y : dz.x, q ? y.y : y.x, q ? yl.y : yl.x);
    }
    if (P > 2) __builtin_amdgcn_sched_barrier(0);  // one slot pair's tables live at a time
  }
  // (scheduling fence: the second phase's table reads otherwise start before the first phase's
  // are consumed, and twelve more live registers tip the kernel into scratch)
  __builtin_amdgcn_sched_barrier(0);
  const double pbz = from_prev_lane_z(bz[P - 1]);
  const double area = ws[S_AREA];
  const double2 ra = *reinterpret_cast<const double2 *>(ws + S_RAREA);  // {1/area, its low part}
#pragma unroll
  for (int h = 0; h < P / 2; ++h) {
    const double2 dzc = jf_pair(lds + L::T_DZC, lane, h), y = jf_pair(lds + L::T_RDZC, lane, h);
    double2 yl = make_double2(0., 0.);
    if constexpr (!D3) yl = jf_pair(lds + L::T_RDZCL, lane, h);
    double2 kp;
    if constexpr (KAPREG)
      kp = make_double2(c.kap[2 * h], c.kap[2 * h + 1]);
    else
      kp = jf_pair(kap, lane, h);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = 2 * h + q;
      const double dn = (p > 0) ? bz[p > 0 ? p - 1 : 0] : pbz;
      const double bzz =
          D3 ? div_by_recip3(bz[p] - dn, q ? dzc.y : dzc.x, q ? y.y : y.x)
             : div_by_recip2(bz[p] - dn, q ? dzc.y : dzc.x, q ? y.y : y.x, q ? yl.y : yl.x);  // :238
      // upwind flux (-weff) bz* (column.py:242-246): exactly one of wn, wp is -weff, the other
      // product an exact zero
      const double flx = __builtin_fma(c.wn[p], bz[p], c.wp[p] * dn);
      const double adv = D3 ? div_by_recip3(flx, area, ra.x) : div_by_recip2(flx, area, ra.x, ra.y);
      c.b[p] = c.b[p] + dt * (adv + (q ? kp.y : kp.x) * bzz);  // column.py:245-249
    }
    if (P > 2) __builtin_amdgcn_sched_barrier(0);
  }
}

// The same step in the reference-faithful form for ANY operand, finite or not (col_vertadvdiff's
// DIV == 0, column.hip.h): IEEE divisions, the compare-select upwind flux of column.py:242-246,
// boundary / padding levels left untouched by a select.  Taken by a wave whose operands leave
// the window of the exact-division shortcuts (in_fast_div_range); never on the hot path.
template <int P, bool KAPREG>
__device__ __forceinline__ void jf_vertadvdiff_ieee(JfCol<P> &c, const double *lds,
                                                    const double *ws, const double *kap, int lane,
                                                    double dt, int nz) {
  using L = JfLds<P>;
  double bz[P];
  const double nb0 = from_next_lane_z(c.b[0]);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    const double up = (p < P - 1) ? c.b[p + 1 < P ? p + 1 : p] : nb0;
    const double qv = (up - c.b[p]) / lds[L::T_DZ + jf_entry<P>(lane, p)];  // column.py:235
    bz[p] = (i < nz - 1) ? qv : 0.0;
  }
  const double pbz = from_prev_lane_z(bz[P - 1]);
  const double area = ws[S_AREA];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    const double dn = (p > 0) ? bz[p > 0 ? p - 1 : 0] : pbz;
    const double bzz = (bz[p] - dn) / lds[L::T_DZC + jf_entry<P>(lane, p)];  // :238
    // weff < 0 exactly where wn holds -weff != 0 (load_coef); -weff is wn there, wp elsewhere
    const bool neg = c.wn[p] != 0.0;
    const double flx = (neg ? c.wn[p] : c.wp[p]) * (neg ? bz[p] : dn);  // :242-246
    const double adv = flx / area;
    const double kp = KAPREG ? c.kap[p] : kap[jf_entry<P>(lane, p)];
    const double nb = c.b[p] + dt * (adv + kp * bzz);  // :245-249
    c.b[p] = (i >= 1 && i <= nz - 2) ? nb : c.b[p];
  }
}

// mixed-layer state of a lane (point j = lane); everything else about the point lives in the
// wave's LDS rows
struct JfMl {
  double bs;  // bs[j]
  int jh;     // interval of the last interpolation: bb[jh] <= bs < bb[jh+1]
};

// Tables of the parallel cyclic reduction: the recurrences of ml_build_pcr (so_ml.hip.h),
// stored in the layout above.
__device__ __forceinline__ void jf_build_pcr(double *T, int ny, double s, int lane) {
  const bool interior = lane >= 1 && lane <= ny - 2;
  double a = interior ? -s / 2. : 0., b = interior ? 1 + s : 1., c = a;
#pragma unroll
  for (int l = 0; l < PCR_LEVELS; ++l) {
    const int k = 1 << l;
    const double a_lo = __shfl_up(a, k, 64), b_lo = __shfl_up(b, k, 64), c_lo = __shfl_up(c, k, 64);
    const double a_hi = __shfl_down(a, k, 64), b_hi = __shfl_down(b, k, 64),
                 c_hi = __shfl_down(c, k, 64);
    const double alpha = (lane >= k) ? -a / b_lo : 0.;
    const double gamma = (lane + k <= 63) ? -c / b_hi : 0.;
    T[l * 128 + lane * 2] = alpha;
    T[l * 128 + lane * 2 + 1] = gamma;
    const double bn = __builtin_fma(gamma, a_hi, __builtin_fma(alpha, c_lo, b));
    a = alpha * a_lo;
    c = gamma * c_hi;
    b = bn;
  }
  const double rb = 1.0 / b;
  T[PCR_LEVELS * 128 + lane * 2] = b;
  T[PCR_LEVELS * 128 + lane * 2 + 1] = rb;
  T[(PCR_LEVELS + 1) * 128 + lane] = recip_lo(b, rb);
}

// r of the lane 32 away (lanes < 32: lane + 32, lanes >= 32: lane - 32) without an LDS round
// trip: v_permlane32_swap (gfx950) exchanges the upper half of one register with the lower half
// of another; swapping a copy of r with itself leaves {r.lo, r.lo} and {r.hi, r.hi}.
__device__ __forceinline__ double jf_swap_halves(double r, bool low_half) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(r), __double2loint(r), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(r), __double2hiint(r), false, false);
  // element 0 = {r.lo, r.lo} (per half), element 1 = {r.hi, r.hi}
  return __hiloint2double(low_half ? hi[1] : hi[0], low_half ? lo[1] : lo[0]);
}

// x = U^-1 r by parallel cyclic reduction with the block's tables (ml_pcr_solve's recurrences).
// A level needs r of the lanes k below and k above; a missing neighbour has a zero multiplier,
// so any finite value may stand in for it.  k = 1 comes by DPP wave shifts and k = 32 by
// v_permlane32_swap (no LDS round trip on the dependent chain); k = 2 ... 16 by ds_bpermute,
// whose address is taken modulo 64 lanes (the constants fold into the offset field).
__device__ __forceinline__ double jf_pcr_solve(double r, const double *T, int lane, int a4) {
#pragma unroll
  for (int l = 0; l < PCR_LEVELS; ++l) {
    const int k = 1 << l;
    const double2 ag = *reinterpret_cast<const double2 *>(T + l * 128 + lane * 2);
    double r_lo, r_hi;
    if (k == 1) {
      r_lo = from_prev_lane_z(r);
      r_hi = from_next_lane_z(r);
    } else if (k == 32) {
      r_lo = r_hi = jf_swap_halves(r, lane < 32);
    } else {
      const int lo_i = a4 + (256 - 4 * k), hi_i = a4 + 4 * k;
      r_lo = __hiloint2double(__builtin_amdgcn_ds_bpermute(lo_i, __double2hiint(r)),
                              __builtin_amdgcn_ds_bpermute(lo_i, __double2loint(r)));
      r_hi = __hiloint2double(__builtin_amdgcn_ds_bpermute(hi_i, __double2hiint(r)),
                              __builtin_amdgcn_ds_bpermute(hi_i, __double2loint(r)));
    }
    r = __builtin_fma(ag.y, r_hi, __builtin_fma(ag.x, r_lo, r));
    // (fence: a level's multipliers are read with its shuffles, not all 28 registers of them
    // before the first level)
    __builtin_amdgcn_sched_barrier(0);
  }
  const double2 br = *reinterpret_cast<const double2 *>(T + PCR_LEVELS * 128 + lane * 2);
  return div_by_recip2(r, br.x, br.y, T[(PCR_LEVELS + 1) * 128 + lane]);
}

// Contracted column step (opt-in tolerance mode, PM_JN_CONTRACTED): with weff, kappa, Area and
// the grid static between coefficient-set switches, column.py:235-249 is
//   b_i += cu_i (b_{i+1} - b_i) + cl_i (b_i - b_{i-1})
// with cu in c.wn and cl in c.wp (load_coef; zero on boundary and padding levels): one
// subtraction and two fma per level, no table reads.  K1's PM_OP_CONTRACTED in the fused loop.
template <int P>
__device__ __forceinline__ void jf_vertadvdiff_contracted(JfCol<P> &c) {
  const double nb0 = from_next_lane_z(c.b[0]);
  double d_up[P];
#pragma unroll
  for (int p = 0; p < P; ++p) d_up[p] = ((p < P - 1) ? c.b[p + 1 < P ? p + 1 : p] : nb0) - c.b[p];
  const double pd = from_prev_lane_z(d_up[P - 1]);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double d_dn = (p > 0) ? d_up[p > 0 ? p - 1 : 0] : pd;
    c.b[p] = __builtin_fma(c.wn[p], d_up[p], __builtin_fma(c.wp[p], d_dn, c.b[p]));
  }
}

// The block's tables: grid metrics and their double-double reciprocals (col_load_grid's
// operations), the mixed layer's PCR multipliers and metric constants.  Every thread of the block
// calls it once; the caller's __syncthreads() publishes the tables.
template <int P>
__device__ __forceinline__ void jf_block_tables(const pm_jn2018 &a, double dt, double *lds,
                                                int wave, int lane) {
  using L = JfLds<P>;
  const int nz = a.cols.nz, ny = a.ml.ny;
  for (int e = threadIdx.x; e < L::NZP; e += blockDim.x) {
    const double *z = a.cols.z;
    const int h = e >> 7, ln = (e >> 1) & 63, q = e & 1;
    const int i = ln * P + 2 * h + q;
    const int ic = i < nz ? i : nz - 1;
    const int iu = ic + 1 < nz ? ic + 1 : nz - 1;
    const int id = ic > 0 ? ic - 1 : 0;
    const double zc = z[ic];
    const double dz = z[iu] - zc;
    const double dzc = 0.5 * (dz + (zc - z[id]));
    const bool has_up = i < nz - 1, interior = i >= 1 && i <= nz - 2;
    const double rdz = has_up ? 1.0 / dz : 0.0;
    const double rdzc = interior ? 1.0 / dzc : 0.0;  // 0: boundary / padding levels stay put
    lds[L::T_Z + e] = zc;
    lds[L::T_DZ + e] = dz;
    lds[L::T_RDZ + e] = rdz;
    lds[L::T_RDZL + e] = has_up ? recip_lo(dz, rdz) : 0.0;
    lds[L::T_DZC + e] = dzc;
    lds[L::T_RDZC + e] = rdzc;
    lds[L::T_RDZCL + e] = interior ? recip_lo(dzc, rdzc) : 0.0;
  }
  if (wave == 0) {
    if (lane < 16) reinterpret_cast<int *>(lds + L::PROG)[lane] = 0;
    if (ny >= 3 && a.ml.y != nullptr) {  // (the two-column run kernel has no mixed layer)
      const double ml_h = a.ml.h, ml_L = a.ml.L, dy = a.ml.y[1] - a.ml.y[0];
      const double ms_s = a.ml.Ks * dt / (dy * dy);  // SO_ML.py:191
      jf_build_pcr(lds + L::PCR, ny, ms_s, lane);
      if (lane == 0) {
        double *K = lds + L::KML;
        const double rh = 1.0 / ml_h, rL = 1.0 / ml_L, rdy = 1.0 / dy;
        K[K_H] = ml_h;
        K[K_RH] = rh;
        K[K_RHL] = recip_lo(ml_h, rh);
        K[K_L] = ml_L;
        K[K_RL] = rL;
        K[K_RLL] = recip_lo(ml_L, rL);
        K[K_DY] = dy;
        K[K_RDY] = rdy;
        K[K_RDYL] = recip_lo(dy, rdy);
        K[K_SH] = ms_s / 2.;
        K[K_1MS] = 1 - ms_s;
      }
    }
  }
}

// One member's launch: `nsteps` x [BC switch -> both columns -> mixed layer] from the state in
// HBM and back.  The body of k_jn2018_fast, and the stepping phase of the persistent run kernel
// (coupled_run.hip: `wstride` = the wave's share of LDS there, `s0` = the steps the wave has
// behind it, for the lag-based issue priority).  The kernel's argument block must START with the
// pm_jn2018 (jf_args re-reads it there).  SYNC: the block's tables are being built by this very
// block (k_jn2018_fast): a __syncthreads() after the member's own loads publishes them.
// IEEE = false: the exact-division shortcuts (div_by_recip2, the select-free upwind flux), for
// members whose column operands all lie in the window of those shortcuts (in_fast_div_range; K1's
// operand guard, checked where K1 checks it: at the launch's start -- state, forcing, the
// coefficients, grid, dt -- and again when a BC switch brings in the other coefficient set).  A
// member outside the window STOPS there: its state after the s steps done so far (0 at the
// launch's start) goes back to HBM, status gets bit 5 and s in bits 8..19, and s is returned; the
// caller's follow-up launch takes the remaining steps with IEEE = true (true divisions,
// compare-select flux, boundary levels held by a select: col_vertadvdiff's DIV == 0 form) -- a
// separate kernel, so that the hot kernel, which sits exactly at 128 registers, carries none of
// that code (inlined as a second leg it cost 6.5 % of config 5).  Returns the steps done.
// CT: the columns step in the contracted form (tolerance mode); otherwise every operation is
// the reference's, in its order.
template <int P, bool CT, bool VEC, bool SYNC, bool IEEE, bool D3 = false>
__device__ __forceinline__ int jf_member_run(const pm_jn2018 &a, double dt, int nsteps, int s0,
                                              int m_raw, double *lds, int wstride, int wave,
                                              int lane) {
  using L = JfLds<P>;
  const bool m_ok = m_raw < a.n;
  const int m = m_ok ? m_raw : a.n - 1;
  const int n = a.n, nz = a.cols.nz, ny = a.ml.ny;
  PM_WAVE_BEGIN
  PM_TICK_INIT
  double *wl = lds + L::WAVE0 + wave * wstride;  // this wave's rows; wl[level] = basin b
  double *ws = wl + L::W_S;                      // this member's scalars

  // ---- this member's columns: state into registers, scalars into the wave's LDS block
  JfCol<P> cb, cn;
  JfConv<P> vb, vn;
  int ksel_b, ksel_n;
  bool hint_ok = true, range_ok = in_fast_div_range_or_lost(dt);
  const bool shared = (a.hints & PM_JN_SHARED_COEF) != 0;
  {
    const pm_columns &c = a.cols;
    auto load_col = [&](JfCol<P> &r, JfConv<P> &v, int col, double *wsc) {
      bool same = true;
      // (PM_JN_SHARED_COEF: the column kind's one coefficient row, resident in L2)
      const size_t arow = (size_t)(shared ? (col < n ? 0 : n) : col) * nz;
      const double a0 = c.area[arow];
      double rb[P], ra_[P];
      jf_load_row<P, VEC>(rb, c.b + (size_t)col * nz, lane, nz);
      jf_load_row<P, VEC>(ra_, c.area + arow, lane, nz);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = lane * P + p;
        r.b[p] = i < nz ? rb[p] : JF_PAD;
        same = same && ra_[p] == a0;  // (lanes past the last level hold copies of real levels)
      }
      hint_ok = hint_ok && __ballot(!same) == 0ull;
      {  // operands inside the exact-division window? (common.hip.h; flagged, not branched on)
        bool ok = in_fast_div_range_or_lost(a0) && a0 != 0.0 && in_fast_div_range_or_lost(c.bs[col]) &&
                  in_fast_div_range_or_lost(c.N2min[col]) && in_fast_div_range_or_lost(c.bbot[col]);
#pragma unroll
        for (int p = 0; p < P; ++p) ok = ok && (lane * P + p >= nz || in_fast_div_range_or_lost(r.b[p]));
        range_ok = range_ok && __ballot(!ok) == 0ull;
      }
      if (lane == 0) {
        const double ra = 1.0 / a0;
        wsc[S_BS] = c.bs[col];
        wsc[S_N2] = c.N2min[col];
        wsc[S_ZC] = 0.;
        wsc[S_AREA] = a0;
        wsc[S_RAREA] = ra;
        wsc[S_RAREAL] = recip_lo(a0, ra);
      }
#pragma unroll
      for (int p = 0; p < P; ++p) v.cm[p] = 0ull;
      v.any = false;
      v.valid = false;
    };
    load_col(cb, vb, m, ws + S_B);
    load_col(cn, vn, n + m, ws + S_NN);
    ksel_b = c.ksel[m];
    ksel_n = c.ksel[n + m];
  }
  // static conditions of the BC switch (Psi only changes at MOC updates): bit 0 Psi_SO[1] < 0,
  // bit 1 Psi_SO[1] >= 0, bit 2 Psi_res_b[1] > 0, bit 3 Psi_res_n[1] < 0
  int cbits;
  int ml_ind = nz, first_pos = nz;  // SO_ML.py:228-229, :95
  bool ml_ok;
  JfMl q;
  q.bs = 0.;
  q.jh = 0;
  {
    const size_t bz = (size_t)m * nz, by = (size_t)m * ny;
    const double PsiSO1 = a.Psi_SO[bz + 1], Pb1 = a.Psi_res_b[bz + 1], Pn1 = a.Psi_res_n[bz + 1];
    cbits = (PsiSO1 < 0 ? 1 : 0) | (PsiSO1 >= 0 ? 2 : 0) | (Pb1 > 0 ? 4 : 0) | (Pn1 < 0 ? 8 : 0);
    // ---- mixed layer: Psi_mod (SO_ML.py:228-230) is Psi_b with its leading zeros filled
    const double *Psi_b = a.Psi_SO + bz;
    double rpso[P];
    jf_load_row<P, VEC>(rpso, Psi_b, lane, nz);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      const double v = i < nz ? rpso[p] : 0.;
      const unsigned long long nzm = __ballot(i < nz && v != 0.), pm_ = __ballot(i < nz && v > 0.);
      if (nzm) {
        const int j = ((int)__ffsll((long long)nzm) - 1) * P + p;
        ml_ind = j < ml_ind ? j : ml_ind;
      }
      if (pm_) {
        const int j = ((int)__ffsll((long long)pm_) - 1) * P + p;
        first_pos = j < first_pos ? j : first_pos;
      }
    }
    ml_ok = ml_ind < nz;  // all-zero Psi_b: IndexError in the reference
    const double fillv = ml_ok ? Psi_b[ml_ind] : 0.;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      wl[L::W_PM + i] = (i < ml_ind || i >= nz) ? fillv : rpso[p];
    }
    double f1 = 0., f2 = 0., br = 0.;
    if (lane < ny) {
      q.bs = a.ml.bs[by + lane];
      if (ml_ok) {  // loop-invariant parts of the surface-flux tendency (SO_ML.py:250-252)
        const double ml_h = a.ml.h;
        f1 = a.ml.surflux[by + lane] / ml_h;
        f2 = a.ml.rest_mask[by + lane] * a.ml.v_pist / ml_h;
        br = a.ml.b_rest[by + lane];
      }
    }
    wl[L::W_F + 2 * lane] = f1;
    wl[L::W_F + 2 * lane + 1] = f2;
    wl[L::W_BR + lane] = br;
    if (lane == 0) {
      // Column.bbot of the basin: only an undefined Psi_SO (neither < 0 nor >= 0) carries it
      // over a step, and then it is what the last step left in level 0 (the launch's first
      // step: the caller's value).  The north's is rewritten every step.
      ws[S_BBOT0] = a.cols.bbot[m];
    }
  }
  if constexpr (SYNC)
    __syncthreads();
  else
    __builtin_amdgcn_wave_barrier();
  {  // the grid's part of the operand window (z, dz, dzc: col_inputs_in_fast_range, column.hip.h)
    bool ok = true;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p, e = jf_entry<P>(lane, p);
      const double zv = lds[L::T_Z + e], dzv = lds[L::T_DZ + e], dzcv = lds[L::T_DZC + e];
      ok = ok && (i >= nz || (in_fast_div_range_or_lost(zv) &&
                              (i >= nz - 1 || (in_fast_div_range_or_lost(dzv) && dzv != 0.0)) &&
                              in_fast_div_range_or_lost(dzcv) && dzcv != 0.0));
    }
    range_ok = range_ok && __ballot(!ok) == 0ull;
  }
  int status = ml_ok ? 0 : 1;
  bool ps_valid = false;
  const bool lane0 = lane == 0;
  const bool ml_act = lane < ny, ml_int = lane >= 1 && lane <= ny - 2;

  // A change of a column's coefficient set (rare) leaves the step loop, reloads and re-enters at
  // the same step: inside the loop the reload's merge with the resident coefficients cost the
  // register allocator more than the reload itself.
  int s = hint_ok ? 0 : nsteps;
  bool skipped = false;
  while (s < nsteps) {
    {
      jf_kargs ka = jf_args();
      bool coef_ok = true;
      auto load_coef = [&](JfCol<P> &r, double *kap, int col, int sel) {
        const double *kappa = ka->cols.kappa, *dAk = ka->cols.dAkappa, *wA = ka->wA;
        const bool shared_rows = (ka->hints & PM_JN_SHARED_COEF) != 0;
        const size_t sbase = ((size_t)sel * (2 * n) + (shared_rows ? (col < n ? 0 : n) : col)) * nz;
        double rk[P], rd[P], rw[P];
        jf_load_row<P, VEC>(rk, kappa + sbase, lane, nz);
        jf_load_row<P, VEC>(rd, dAk + sbase, lane, nz);
        jf_load_row<P, VEC>(rw, wA + (size_t)col * nz, lane, nz);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = lane * P + p;
          r.kap[p] = rk[p];
          if (kap) kap[jf_entry<P>(lane, p)] = r.kap[p];
          const double w = rw[p] - rd[p];
          const bool interior = i >= 1 && i <= nz - 2;
          const double we = interior ? w : 0.0;
          r.wn[p] = (we < 0.0) ? -we : 0.0;
          r.wp[p] = (we < 0.0) ? 0.0 : -we;
          if constexpr (!IEEE && !CT)
            coef_ok = coef_ok && in_fast_div_range_or_lost(we) && in_fast_div_range_or_lost(r.kap[p]);
          if constexpr (CT) {
            // cu = dt (kappa / (dzc dz) + wn / (A dz)), cl = dt (-kappa / (dzc dz') + wp / (A dz')),
            // dz' = the spacing below the level (col_make_contracted, column.hip.h)
            const int iq = interior ? i : 1;
            const double dz_up = lds[L::T_DZ + jf_entry<P>(iq / P, iq % P)],
                         dz_dn = lds[L::T_DZ + jf_entry<P>((iq - 1) / P, (iq - 1) % P)],
                         dzc = lds[L::T_DZC + jf_entry<P>(iq / P, iq % P)];
            const double area = ka->cols.area[(size_t)(shared_rows ? (col < n ? 0 : n) : col) * nz];
            const double cu = dt * (r.kap[p] / (dzc * dz_up) + r.wn[p] / (area * dz_up));
            const double cl = dt * (-r.kap[p] / (dzc * dz_dn) + r.wp[p] / (area * dz_dn));
            r.wn[p] = interior ? cu : 0.0;
            r.wp[p] = interior ? cl : 0.0;
          }
        }
      };
      load_coef(cb, nullptr, m, ksel_b);
      load_coef(cn, wl + L::W_KN, n + m, ksel_n);
      if constexpr (!IEEE && !CT) {
        range_ok = range_ok && __ballot(!coef_ok) == 0ull;
        if (__builtin_expect(!range_ok, 0)) {
          // not this kernel's member from step s on: the state of step s goes back to HBM, the
          // IEEE follow-up launch takes the remaining steps (status bits 8.. carry s)
          skipped = true;
          break;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // BC switch + both columns of step s; false: a coefficient set changed (nothing done yet)
    auto columns_step = [&](int lane_o, double *wl, double *ws) -> bool {
#ifndef JF_NO_PRIO_ROTATE
      // The SIMD's arbiter favours its oldest wave: left alone its four waves (waves w, w+4,
      // w+8, w+12 of the block) finish ~20 us apart and the last runs alone; with priorities
      // rotated by step number most SIMDs even out, but a wave that falls behind stays behind
      // (one wave in ~15 blocks ended 25 us after its mates).  So each wave publishes its step
      // counter and takes its priority from its rank: the one furthest behind issues first.
#ifndef JF_PRIO_EVERY
#define JF_PRIO_EVERY 1
#endif
      if (JF_PRIO_EVERY == 1 || (s & (JF_PRIO_EVERY - 1)) == 0) {
        int *prog = reinterpret_cast<int *>(lds + L::PROG) + (wave & 3) * 4;
        const int sg = s0 + s;  // steps behind this wave (over all launches of a persistent run)
        prog[wave >> 2] = sg;
        const int4 pv = *reinterpret_cast<const int4 *>(prog);
        const int p0 = __builtin_amdgcn_readfirstlane(pv.x), p1 = __builtin_amdgcn_readfirstlane(pv.y),
                  p2 = __builtin_amdgcn_readfirstlane(pv.z), p3 = __builtin_amdgcn_readfirstlane(pv.w);
        const int lo = min(min(p0, p1), min(p2, p3)), hi = max(max(p0, p1), max(p2, p3));
        if (sg <= lo) __builtin_amdgcn_s_setprio(3);
        else if (sg >= hi) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(1);
      }
#endif
      // ---- bottom-BC switch (run_JansenNadeau_2018.py:233-254), vector code: lane 0 owns
      // levels 0 and 1 of both columns and point 0 of the channel; the other lanes compute along
      double bbot_b, bbot_n;
      {
        const bool c_south = (cbits & 1) != 0, c_nosouth = (cbits & 2) != 0, c_pb = (cbits & 4) != 0,
                   c_pn = (cbits & 8) != 0;
        const double bb0 = cb.b[0], bb1 = cb.b[1], bn0 = cn.b[0], bn1 = cn.b[1], bs0 = q.bs;
        const bool from_north = c_pb && bn0 < bb1 && bn0 < bs0;
        double nb = (s == 0) ? ws[S_BBOT0] : bb0;
        nb = c_south ? bs0 : nb;         // bottom water coming in from the south
        nb = c_nosouth ? bb1 : nb;       // no bottom water coming in: no-flux BBC
        bbot_b = from_north ? bn0 : nb;  // bottom water coming in from the north
        const bool from_basin = c_pn && bb0 < bn1;
        bbot_n = from_basin ? bb0 : bn1;
        const int kb_l = from_north ? 1 : (c_nosouth ? 0 : (c_south ? 1 : ksel_b));
        const int kn_l = from_basin ? 1 : 0;
        const int kb = __builtin_amdgcn_readfirstlane(kb_l),
                  kn = __builtin_amdgcn_readfirstlane(kn_l);
        if (__builtin_expect(kb != ksel_b || kn != ksel_n, 0)) {
          JF_RARE(2)
          ksel_b = kb;
          ksel_n = kn;
          return false;
        }
      }
      // ---- basin.timestep / north.timestep, do_conv=True (:257-258)
      // (both convective adjustments first: their rare branches then precede one straight block
      // of arithmetic, whose scheduling fences keep one slot pair's tables live at a time)
      jf_convect<P>(cb.b, vb, lds, ws + S_B, lane_o, nz);
      jf_convect<P>(cn.b, vn, lds, ws + S_NN, lane_o, nz);
      cb.b[0] = lane0 ? bbot_b : cb.b[0];  // column.py:232 (after convect: it may write level 0)
      cn.b[0] = lane0 ? bbot_n : cn.b[0];
      if constexpr (CT) {
        jf_vertadvdiff_contracted<P>(cb);
        jf_vertadvdiff_contracted<P>(cn);
      } else if constexpr (IEEE) {
        jf_vertadvdiff_ieee<P, true>(cb, lds, ws + S_B, nullptr, lane_o, dt, nz);
        jf_vertadvdiff_ieee<P, false>(cn, lds, ws + S_NN, wl + L::W_KN, lane_o, dt, nz);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        jf_vertadvdiff<P, true, D3>(cb, lds, ws + S_B, nullptr, lane_o, dt);
        __builtin_amdgcn_sched_barrier(0);
        jf_vertadvdiff<P, false, D3>(cn, lds, ws + S_NN, wl + L::W_KN, lane_o, dt);
        __builtin_amdgcn_sched_barrier(0);
      }
      return true;
    };
    if (ml_ok) {
    for (; s < nsteps; ++s) {
      PM_TICK(6)
      // address bases the optimiser must not see through: it would hoist one derived address per
      // access pattern out of the loop (dozens of registers, then spilled) instead of folding
      // the constants into the instructions' offset fields
      int lane_o = lane, woff = wave * wstride;
      asm volatile("" : "+v"(lane_o), "+s"(woff));
      const int a4 = lane_o << 2;
      double *wl = lds + L::WAVE0 + woff, *ws = wl + L::W_S;
      if (__builtin_expect(!columns_step(lane_o, wl, ws), 0)) break;
      PM_TICK(0)
      JF_RARE(0)
      // ---- channel.timestep(b_basin=basin.b, Psi_b=PsiSO.Psi) (:261), ml_step_reg's operations
      {
        const double *bb = wl + L::W_BB, *pm = wl + L::W_PM;
#pragma unroll
        for (int hh = 0; hh < P / 2; ++hh)
          *reinterpret_cast<double2 *>(wl + L::W_BB + lane_o * P + 2 * hh) =
              make_double2(cb.b[2 * hh], cb.b[2 * hh + 1]);
        __builtin_amdgcn_wave_barrier();
        // Psi_s = np.interp(bs, b_basin, Psi_mod) (:232).  A point is in the interval it was in
        // one step earlier or -- every other step for some point of a member -- in the one next
        // to it: that is settled with two table reads; the interpolation then reads its interval
        // (like np.interp this takes b_basin as sorted); anything else searches like np.interp.
        double ps;
        {
          const double x = q.bs;
          const int j0 = q.jh;  // <= nz - 2
          const double x0 = bb[j0], x1 = bb[j0 + 1];
          int j = j0 + (x >= x1 ? 1 : 0) - (x < x0 ? 1 : 0);
          j = j < 0 ? 0 : (j > nz - 2 ? nz - 2 : j);
          const double lx = bb[j], hx = bb[j + 1], lf = pm[j], hf = pm[j + 1];
          const double xlo = bb[0], xhi = bb[nz - 1];
          const bool hit = (lx <= x) && (x < hx);
          ps = interp_finish(x, j, lx, hx, lf, hf);
          // np.interp: NaN first, then beyond the table's ends (x equal to the last node also
          // returns the last value)
          const bool isnan_x = x != x, above = x >= xhi, below = x < xlo;
          if (below) ps = pm[0];
          if (above) ps = pm[nz - 1];
          if (isnan_x) ps = x;
          q.jh = j;
          const bool search = ml_act && !hit && !isnan_x && !above && !below;
          if (__builtin_expect(__ballot(search) != 0ull, 0)) {
            JF_RARE(3)
            if (search) {  // np.interp's upper-bound search from scratch
              int lo_i = 0, hi_i = nz;
              while (lo_i < hi_i) {
                const int mid = lo_i + ((hi_i - lo_i) >> 1);
                if (x >= bb[mid])
                  lo_i = mid + 1;
                else
                  hi_i = mid;
              }
              const int jb = lo_i - 1;
              // (jb = -1 only when the basin column holds NaNs: stay inside the row)
              const int jj = (jb == nz - 1) ? nz - 2 : (jb < 0 ? 0 : jb);
              q.jh = jj;
              ps = (jb == nz - 1) ? pm[nz - 1]
                                  : interp_finish(x, jb, bb[jj], bb[jj + 1], pm[jj], pm[jj + 1]);
            }
          }
        }
        PM_TICK(1)
        // argmin(bs): first minimum, a NaN wins (np.argmin); 0 when no point lies below point 0
        int amin = 0;
        {
          const double v = ml_act ? q.bs : __builtin_inf();
          const double v0 = lane_value(q.bs, 0);
          if (__builtin_expect(__ballot(!(v >= v0)) != 0ull, 0)) {
            JF_RARE(5)
            const double mn = wave_min_f64(v);
            const unsigned long long at_min = __ballot(ml_act && v == mn);
            const unsigned long long nanm = __ballot(ml_act && v != v);
            const int mi = at_min ? (int)__ffsll((long long)at_min) - 1 : 0;
            amin = nanm ? (int)__ffsll((long long)nanm) - 1 : mi;
          }
        }
        if (lane < amin || lane0) ps = 0.;  // :240-243
        const bool upwell = (__ballot(ps > 0) & 2ull) != 0ull;  // set_boundary_conditions, :93-98
        if (__builtin_expect(upwell && first_pos >= nz, 0)) {
          ml_ok = false;  // IndexError in the reference; the mixed layer stops evolving
          status = 1;
          ++s;
          break;  // the rest of the launch steps the columns only (loop below)
        } else {
          PM_TICK(2)
          __builtin_amdgcn_sched_barrier(0);  // (fence: the constants below are read here, not earlier)
          const double2 *K2 = reinterpret_cast<const double2 *>(lds + L::KML);
          const double2 k0 = K2[0], k1 = K2[1], k2 = K2[2], k3 = K2[3], k4 = K2[4];
          const double k_1ms = lds[L::KML + K_1MS];
          const double2 ff = *reinterpret_cast<const double2 *>(wl + L::W_F + 2 * lane_o);
          const double brest = wl[L::W_BR + lane_o];
          const double bsouth = upwell ? bb[first_pos < nz ? first_pos : 0] : 0.;
          double bs = q.bs;
          // lane 0 takes the boundary value; the shifted copies are taken where lane 0's new
          // value is (bs_dn, bl) or is not (bs_up, bu) part of them
          const double bs_up = from_next_lane_z(bs);
          bs = lane0 ? (upwell ? bsouth : bs_up) : bs;
          const double bs_dn = from_prev_lane_z(bs);
          // tendencies from surface flux / restoring and upwind advection (:250-259)
          const double flux = ff.x + ff.y * (brest - bs);
          double adv;
          {
            const double d = (ps < 0.) ? (bs_up - bs) : (bs - bs_dn);
            const double num = -ps * 1e6 * d;
            // num / h / L / dy, each quotient correctly rounded (k0..k4 = {1/h lo, 1/h | h,
            // 1/L lo | 1/L, L | 1/dy lo, 1/dy | dy, s/2})
            // (D3: h, L and dy have passed the host's proof as well)
            const double q1 = D3 ? div_by_recip3(num, k1.x, k0.y) : div_by_recip2(num, k1.x, k0.y, k0.x);
            const double q2 = D3 ? div_by_recip3(q1, k2.y, k2.x) : div_by_recip2(q1, k2.y, k2.x, k1.y);
            const double t = D3 ? div_by_recip3(q2, k4.x, k3.y) : div_by_recip2(q2, k4.x, k3.y, k3.x);
            adv = (ml_int && ps != 0. && ps == ps) ? t : 0.;
          }
          bs = bs + dt * (flux + adv);  // every tendency uses the old bs
          const double bu = from_next_lane_z(bs);
          if (!upwell) bs = lane0 ? bu : bs;  // no-flux BC re-set (:264-266)
          PM_TICK(3)
          // Crank-Nicolson diffusion (:191-196): U x = V bs by parallel cyclic reduction
          {
            const double bl = from_prev_lane_z(bs);
            const double sh = k4.y;
            double r = bs;  // rows 0 and ny-1 of V are identity rows
            if (ml_int) r = sh * bl + k_1ms * bs + sh * bu;
            if (!ml_act) r = 0.;
            bs = jf_pcr_solve(r, lds + L::PCR, lane_o, a4);
          }
          {
            const double up = from_next_lane_z(bs);
            const double v2 = upwell ? bsouth : up;  // final BC re-set (:274)
            bs = lane0 ? v2 : bs;
          }
          q.bs = bs;
          wl[L::W_PS + lane_o] = ps;
          ps_valid = true;
          PM_TICK(4)
        }
      }
    }
    } else {
      for (; s < nsteps; ++s) {
        int lane_o = lane, woff = wave * wstride;
        asm volatile("" : "+v"(lane_o), "+s"(woff));
        double *wl = lds + L::WAVE0 + woff, *ws = wl + L::W_S;
        if (__builtin_expect(!columns_step(lane_o, wl, ws), 0)) break;
      }
    }
  }

  PM_TICK_FLUSH
  // ---- results
  jf_kargs ka = jf_args();
  const size_t by = (size_t)m * ny;
  bool bad = false;
  {
    double *bout = ka->cols.b;
    if (m_ok && hint_ok) {
      jf_store_row<P, VEC>(bout + (size_t)m * nz, cb.b, lane, nz);
      jf_store_row<P, VEC>(bout + (size_t)(n + m) * nz, cn.b, lane, nz);
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (lane * P + p < nz) bad |= !isfinite(cb.b[p]) || !isfinite(cn.b[p]);
  }
  if (lane < ny) {
    bad |= !isfinite(q.bs);
    if (m_ok && hint_ok) {
      ka->ml.bs[by + lane] = q.bs;
      double *Psi_s = ka->ml.Psi_s;
      if (Psi_s && ps_valid) Psi_s[by + lane] = wl[L::W_PS + lane];
    }
  }
  const bool anybad = __ballot(bad) != 0ull;
  if (lane == 0 && m_ok) {
    if (hint_ok && s > 0) {  // (s: the steps done)
      // Column.bbot after the last step = what that step imposed on level 0
      double *bbot = const_cast<double *>(ka->cols.bbot);
      int32_t *ksel = const_cast<int32_t *>(ka->cols.ksel);
      bbot[m] = cb.b[0];
      bbot[n + m] = cn.b[0];
      ksel[m] = ksel_b;
      ksel[n + m] = ksel_n;
    }
    int32_t *nonfinite = ka->cols.nonfinite;
    if (nonfinite) {
      nonfinite[m] = anybad ? 1 : 0;
      nonfinite[n + m] = anybad ? 1 : 0;
    }
    int32_t *st = ka->ml.status;
    // (a persistent run ORs the intervals' flags together: its caller zeroes the array first)
    if (st)
      st[m] = (SYNC ? 0 : st[m]) | status | (anybad ? 2 : 0) | (hint_ok ? 0 : 16) |
              ((IEEE || skipped) ? 32 : 0) | (skipped ? (s << 8) : 0);
  }
  PM_WAVE_END(m_raw)
  return s;
}

template <int P, bool CT, bool VEC, bool D3 = false>
__global__ __launch_bounds__(64 * JF_WAVES) JF_OCC_ATTR
void k_jn2018_fast(pm_jn2018 a, double dt, int nsteps) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  jf_block_tables<P>(a, dt, lds, wave, lane);
  jf_member_run<P, CT, VEC, true, false, D3>(a, dt, nsteps, 0, blockIdx.x * JF_WAVES + wave, lds,
                                         JfLds<P>::PER_WAVE, wave, lane);
}

// The next flagged member (status bit `bit`) of a one-wave block's share of the ensemble, -1 when
// none is left: every lane looks at one member's status, a ballot collects 64 of them (the scan
// of k_psi_so's follow-up launch).
struct FlagScan {
  int next, base;
  unsigned long long pending;
  __device__ __forceinline__ void init() {
    next = blockIdx.x * 64;
    base = 0;
    pending = 0ull;
  }
  __device__ __forceinline__ int take(const int32_t *status, int n, int bit, int lane) {
    while (pending == 0ull) {
      if (next >= n) return -1;
      base = next;
      next += gridDim.x * 64;
      const int mm = base + lane;
      pending = __ballot(mm < n && (status[mm] & bit) != 0);
    }
    const int m = base + __builtin_ctzll(pending);
    pending &= pending - 1ull;
    return m;
  }
};

// Follow-up launch of k_jn2018_fast: the members it flagged (operands outside the window of the
// exact-division shortcuts) stepped in the IEEE form.  One-wave blocks; ~2 us when no member is
// flagged (no table is built then).
template <int P, bool VEC>
__global__ __launch_bounds__(64) void k_jn2018_ieee(pm_jn2018 a, double dt, int nsteps) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  FlagScan scan;
  scan.init();
  bool tables = false;
  for (;;) {
    const int m = scan.take(a.ml.status, a.n, 32, lane);
    if (m < 0) return;
    if (!tables) {
      jf_block_tables<P>(a, dt, lds, 0, lane);
      __builtin_amdgcn_wave_barrier();
      tables = true;
    }
    const int s0 = (a.ml.status[m] >> 8) & 0xfff;  // steps the main launch has done
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) a.ml.status[m] &= 0xff;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    jf_member_run<P, false, VEC, false, true>(a, dt, nsteps - s0, s0, m, lds, JfLds<P>::PER_WAVE, 0,
                                              lane);
    __builtin_amdgcn_wave_barrier();
  }
}

// =============================================================================================
// Round 5: the SPLIT lane layout.  One wavefront still owns one member, but the two columns of
// the member step TOGETHER: lanes 0..31 hold the basin column, lanes 32..63 the northern one,
// P = ceil(nz / 32) consecutive levels per lane.  One pass of the column arithmetic then advances
// both columns -- at nz = 200 that is 7 level slots (32 x 7 = 224, 89 % of the lanes' slots are
// real levels) instead of 2 x 4 slots at 64 x 4 = 256 (78 %): an eighth fewer vector instructions
// in the column phase, one joint convective-pattern test instead of two, and every coefficient in
// registers (the northern column's kappa no longer comes from LDS).  The arithmetic per level is
// the same as in jf_vertadvdiff, operation by operation: bit-identical.
//   * block-shared grid tables are indexed by (lane & 31, slot): both halves read the same entry;
//   * per-column scalars (bs, N2min, zconv, Area and its reciprocal pair) sit in the wave's LDS
//     block at `half * S_COL`, read with a per-lane address;
//   * the wave shifts of the 3-point stencil cross the half boundary only into padding /
//     boundary levels, whose table entries (1/dz = 0, 1/dzc = 0, weff = 0) null them exactly;
//   * the bottom-BC switch runs in lane 0 on the basin's levels 0, 1 and the northern ones
//     brought over by v_permlane32_swap; the northern bottom value goes back the same way;
//   * rows move between HBM and the slot layout THROUGH LDS: global loads and stores are
//     lane-contiguous (coalesced 512-byte runs), the transposition is an LDS write + read.
// Shapes: 4 <= nz <= 224, ny <= 64, uniform Area (as k_jn2018_fast); nz > 224 keeps that kernel.
// The IEEE follow-up launch (k_jn2018_ieee) is shared: a member outside the exact-division window
// stops here with its state in HBM, whatever the lane layout was.
template <int P>
struct JsLds {
  static constexpr int HP = (P + 1) / 2;   // slot pairs
  static constexpr int NT = 64 * HP;       // doubles per table: entry (hl, p) at ((p/2)*32 + hl)*2 + (p&1)
  static constexpr int NLEV = 32 * P + (32 * P) % 2;  // levels incl. padding
  static constexpr int T_Z = 0, T_DZ = NT, T_RDZ = 2 * NT, T_RDZL = 3 * NT, T_DZC = 4 * NT,
                       T_RDZC = 5 * NT, T_RDZCL = 6 * NT;
  static constexpr int PCR = 7 * NT;
  static constexpr int KML = PCR + JF_PCR_DOUBLES;
  static constexpr int PROG = KML + K_N;
  static constexpr int WAVE0 = PROG + 8;
  // per wave: b_basin[level] (np.interp's xp, and the basin half of every row transposition),
  // Psi_mod[level], a staging row (the northern half of a transposition), {surflux/h,
  // rest_mask*v_pist/h}[64], b_rest[64], Psi_s[64], the member's scalars
  // (W_ST doubles as the cache of the northern column's adjusted values bs + N2min (z - zc) inside
  // the step loop, in the tables' slot-pair layout: 32 lanes x 2 HP doubles)
  static constexpr int NST = NLEV > 64 * HP ? NLEV : 64 * HP;
  static constexpr int W_BB = 0, W_PM = NLEV, W_ST = 2 * NLEV, W_F = 2 * NLEV + NST,
                       W_BR = W_F + 128, W_PS = W_F + 192, W_S = W_F + 256;
  static constexpr int PER_WAVE = W_S + S_N;
  static constexpr int TOTAL = WAVE0 + JF_WAVES * PER_WAVE;
};

template <int P>
__device__ __forceinline__ int js_entry(int hl, int p) {
  return ((p >> 1) * 32 + hl) * 2 + (p & 1);
}
__device__ __forceinline__ double2 js_pair(const double *T, int hl, int h) {
  return *reinterpret_cast<const double2 *>(T + (h * 32 + hl) * 2);
}

template <int P>
struct JsCol {
  double b[P];          // both columns' state: lanes 0..31 basin, 32..63 north; padding JF_PAD
  double wn[P], wp[P];  // upwind split of -weff (contracted mode: cu, cl)
  double kap[P];
};

template <int P>
__device__ __forceinline__ void js_block_tables(const pm_jn2018 &a, double dt, double *lds,
                                                int wave, int lane) {
  using L = JsLds<P>;
  const int nz = a.cols.nz, ny = a.ml.ny;
  for (int e = threadIdx.x; e < L::NT; e += blockDim.x) {
    const double *z = a.cols.z;
    const int h = e >> 6, hl = (e >> 1) & 31, q = e & 1;
    const int p = 2 * h + q;
    const int i = (p < P) ? hl * P + p : nz;  // (the unused half of an odd P's last pair: padding)
    const int ic = i < nz ? i : nz - 1;
    const int iu = ic + 1 < nz ? ic + 1 : nz - 1;
    const int id = ic > 0 ? ic - 1 : 0;
    const double zc = z[ic];
    const double dz = z[iu] - zc;
    const double dzc = 0.5 * (dz + (zc - z[id]));
    const bool has_up = i < nz - 1, interior = i >= 1 && i <= nz - 2;
    const double rdz = has_up ? 1.0 / dz : 0.0;
    const double rdzc = interior ? 1.0 / dzc : 0.0;
    lds[L::T_Z + e] = zc;
    lds[L::T_DZ + e] = dz;
    lds[L::T_RDZ + e] = rdz;
    lds[L::T_RDZL + e] = has_up ? recip_lo(dz, rdz) : 0.0;
    lds[L::T_DZC + e] = dzc;
    lds[L::T_RDZC + e] = rdzc;
    lds[L::T_RDZCL + e] = interior ? recip_lo(dzc, rdzc) : 0.0;
  }
  if (wave == 0) {
    if (lane < 16) reinterpret_cast<int *>(lds + L::PROG)[lane] = 0;
    const double ml_h = a.ml.h, ml_L = a.ml.L, dy = a.ml.y[1] - a.ml.y[0];
    const double ms_s = a.ml.Ks * dt / (dy * dy);  // SO_ML.py:191
    jf_build_pcr(lds + L::PCR, ny, ms_s, lane);
    if (lane == 0) {
      double *K = lds + L::KML;
      const double rh = 1.0 / ml_h, rL = 1.0 / ml_L, rdy = 1.0 / dy;
      K[K_H] = ml_h;
      K[K_RH] = rh;
      K[K_RHL] = recip_lo(ml_h, rh);
      K[K_L] = ml_L;
      K[K_RL] = rL;
      K[K_RLL] = recip_lo(ml_L, rL);
      K[K_DY] = dy;
      K[K_RDY] = rdy;
      K[K_RDYL] = recip_lo(dy, rdy);
      K[K_SH] = ms_s / 2.;
      K[K_1MS] = 1 - ms_s;
    }
  }
}

// rows in the lane-contiguous layout.  VEC (rows 16-byte aligned, nz even: decided by the
// launcher): element k of a lane = level 128 (k / 2) + 2 lane + (k & 1), two 16-byte accesses per
// row; else level 64 k + lane, four 8-byte accesses.
constexpr int JS_CH = 4;  // covers nz <= 256
template <bool VEC>
__device__ __forceinline__ int js_ci(int k, int lane) {
  return VEC ? 128 * (k >> 1) + 2 * lane + (k & 1) : 64 * k + lane;
}
template <bool VEC>
__device__ __forceinline__ void js_load_chunks(double (&g)[JS_CH], const double *__restrict__ row,
                                               int lane, int nz) {
  if constexpr (VEC) {
#pragma unroll
    for (int kk = 0; kk < JS_CH / 2; ++kk) {
      const int i0 = 128 * kk + 2 * lane;
      const double2 v = *reinterpret_cast<const double2 *>(row + (i0 < nz - 2 ? i0 : nz - 2));
      g[2 * kk] = v.x;
      g[2 * kk + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < JS_CH; ++k) {
      const int i = 64 * k + lane;
      g[k] = row[i < nz ? i : nz - 1];
    }
  }
}
// into a level-ordered LDS row; levels >= nz go to `dump` (16 bytes nobody reads): no branches
template <bool VEC>
__device__ __forceinline__ void js_stage(double *stage, double *dump, const double (&g)[JS_CH],
                                         int lane, int nz) {
  if constexpr (VEC) {
#pragma unroll
    for (int kk = 0; kk < JS_CH / 2; ++kk) {
      const int i0 = 128 * kk + 2 * lane;
      double *dst = i0 < nz ? stage + i0 : dump;
      *reinterpret_cast<double2 *>(dst) = make_double2(g[2 * kk], g[2 * kk + 1]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < JS_CH; ++k) {
      const int i = 64 * k + lane;
      double *dst = i < nz ? stage + i : dump;
      *dst = g[k];
    }
  }
}

// Column.convect for both columns at once (column.py:251-271, do_conv = True).  wsh = the lane's
// column's scalar block (ws + half * S_COL); adjn = the wave's cache of the NORTHERN column's
// adjusted values bs + N2min (z - zc) under the cached pattern, slot-pair layout (the basin column
// almost never convects: while it does, the values are formed per step).
template <int P>
struct JsConv {
  unsigned long long cm[P];
  bool any;    // some level of either column convects under the cached pattern
  bool any_b;  // ... of the basin column
  bool valid;
};

template <int P>
__device__ __forceinline__ void js_convect(double (&b)[P], JsConv<P> &s, const double *lds,
                                           double *wsh, double *adjn, int lane, int hl, int nz) {
  using L = JsLds<P>;
  const double bs = wsh[S_BS];
  const bool north = lane >= 32;
  unsigned long long m[P], acc = 0ull;
  bool cvp[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    cvp[p] = b[p] > bs;  // column.py:264 (padding: never)
    m[p] = __builtin_amdgcn_ballot_w64(cvp[p]);
    acc |= m[p] ^ s.cm[p];
  }
  if (__builtin_expect(acc != 0ull || !s.valid, 0)) {
    JF_RARE(1)
    // new pattern: per column, zc = z at the highest non-convecting level (column.py:265-267)
    unsigned long long anym = 0ull;
    int jmax_b = 0, jmax_n = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      anym |= m[p];
      const int nv = (nz - p + P - 1) / P;  // lanes of a half whose slot p is a real level (<= 32)
      const unsigned int vmask = nv >= 32 ? ~0u : ((1u << nv) - 1u);
      const unsigned int nb_ = ~(unsigned int)(m[p] & 0xffffffffull) & vmask;
      const unsigned int nn_ = ~(unsigned int)(m[p] >> 32) & vmask;
      if (nb_ != 0u) {
        const int j = (31 - __clz((int)nb_)) * P + p;
        jmax_b = j > jmax_b ? j : jmax_b;
      }
      if (nn_ != 0u) {
        const int j = (31 - __clz((int)nn_)) * P + p;
        jmax_n = j > jmax_n ? j : jmax_n;
      }
      s.cm[p] = m[p];
    }
    const bool any_b = (anym & 0xffffffffull) != 0ull, any_n = (anym >> 32) != 0ull;
    s.any = anym != 0ull;
    s.any_b = any_b;
    s.valid = true;
    const bool half = north;
    const int jm = half ? jmax_n : jmax_b;
    const double zv = lds[L::T_Z + js_entry<P>(jm / P, jm % P)];
    if (hl == 0) wsh[S_ZC] = zv;  // (only read where the column's pattern has convecting levels)
    const double n2 = wsh[S_N2];
    const bool none_here = half ? !any_n : !any_b;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (none_here && hl * P + p == nz - 1) b[p] = bs;  // column.py:271
      // the northern column's adjusted values under this pattern (column.py:268)
      const double adj = bs + n2 * (lds[L::T_Z + js_entry<P>(hl, p)] - zv);
      if (half) adjn[js_entry<P>(hl, p)] = adj;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (s.any) {
    JF_RARE(6)
    // the northern half takes its cached values where it convects
#pragma unroll
    for (int h = 0; h < L::HP; ++h) {
      const double2 a2 = js_pair(adjn, hl, h);
      b[2 * h] = (cvp[2 * h] && north) ? a2.x : b[2 * h];
      if (2 * h + 1 < P) {
        const int p1 = 2 * h + 1 < P ? 2 * h + 1 : 0;
        b[p1] = (cvp[p1] && north) ? a2.y : b[p1];
      }
    }
    if (__builtin_expect(s.any_b, 0)) {
      // the basin column convects (rare): its lanes are patched in place with values formed here
      const double2 nc = *reinterpret_cast<const double2 *>(wsh + S_N2);  // {N2min, zc}
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const double adj = bs + nc.x * (lds[L::T_Z + js_entry<P>(hl, p)] - nc.y);  // column.py:268
        b[p] = (cvp[p] && !north) ? adj : b[p];
      }
    }
  }
}

// Column.vertadvdiff of both columns (jf_vertadvdiff's operations per level).
template <int P>
__device__ __forceinline__ void js_vertadvdiff(JsCol<P> &c, const double *lds, const double *wsh,
                                               int hl, double dt) {
  using L = JsLds<P>;
  constexpr int HP = L::HP;
  double bz[P];  // (b[i+1]-b[i])/dz[i] (column.py:235)
  const double nb0 = from_next_lane_z(c.b[0]);
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    const double2 dz = js_pair(lds + L::T_DZ, hl, h), y = js_pair(lds + L::T_RDZ, hl, h),
                  yl = js_pair(lds + L::T_RDZL, hl, h);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = 2 * h + q;
      if (p < P) {
        const double up = (p < P - 1) ? c.b[p + 1 < P ? p + 1 : p] : nb0;
        bz[p] = div_by_recip2(up - c.b[p], q ? dz.y : dz.x, q ? y.y : y.x, q ? yl.y : yl.x);
      }
    }
    if (P > 2) __builtin_amdgcn_sched_barrier(0);  // one slot pair's tables live at a time
  }
  __builtin_amdgcn_sched_barrier(0);
  const double pbz = from_prev_lane_z(bz[P - 1]);
  const double area = wsh[S_AREA];
  const double2 ra = *reinterpret_cast<const double2 *>(wsh + S_RAREA);  // {1/area, its low part}
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    const double2 dzc = js_pair(lds + L::T_DZC, hl, h), y = js_pair(lds + L::T_RDZC, hl, h),
                  yl = js_pair(lds + L::T_RDZCL, hl, h);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = 2 * h + q;
      if (p < P) {
        const double dn = (p > 0) ? bz[p > 0 ? p - 1 : 0] : pbz;
        const double bzz =
            div_by_recip2(bz[p] - dn, q ? dzc.y : dzc.x, q ? y.y : y.x, q ? yl.y : yl.x);  // :238
        const double flx = __builtin_fma(c.wn[p], bz[p], c.wp[p] * dn);  // column.py:242-246
        const double adv = div_by_recip2(flx, area, ra.x, ra.y);
        c.b[p] = c.b[p] + dt * (adv + c.kap[p] * bzz);  // column.py:245-249
      }
    }
    if (P > 2) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int P>
__device__ __forceinline__ void js_vertadvdiff_contracted(JsCol<P> &c) {
  const double nb0 = from_next_lane_z(c.b[0]);
  double d_up[P];
#pragma unroll
  for (int p = 0; p < P; ++p) d_up[p] = ((p < P - 1) ? c.b[p + 1 < P ? p + 1 : p] : nb0) - c.b[p];
  const double pd = from_prev_lane_z(d_up[P - 1]);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double d_dn = (p > 0) ? d_up[p > 0 ? p - 1 : 0] : pd;
    c.b[p] = __builtin_fma(c.wn[p], d_up[p], __builtin_fma(c.wp[p], d_dn, c.b[p]));
  }
}

template <int P, bool CT, bool VEC>
__device__ __forceinline__ int js_member_run(const pm_jn2018 &a, double dt, int nsteps, int m_raw,
                                              double *lds, int wave, int lane) {
  using L = JsLds<P>;
  const bool m_ok = m_raw < a.n;
  const int m = m_ok ? m_raw : a.n - 1;
  const int n = a.n, nz = a.cols.nz, ny = a.ml.ny;
  const int half = lane >> 5, hl = lane & 31;
  double *wl = lds + L::WAVE0 + wave * L::PER_WAVE;
  double *ws = wl + L::W_S;
  double *dump = ws + 14;  // two unused doubles of the scalar block
  const int col = m + half * n;  // this lane's column

  JsCol<P> c;
  JsConv<P> cv;
  int ksel_b, ksel_n;
  bool hint_ok = true, range_ok = in_fast_div_range_or_lost(dt);
  const bool shared = (a.hints & PM_JN_SHARED_COEF) != 0;
  int cbits;
  int ml_ind = nz, first_pos = nz;  // SO_ML.py:228-229, :95
  bool ml_ok;
  JfMl q;
  q.bs = 0.;
  q.jh = 0;
  {
    const pm_columns &cc = a.cols;
    // ---- all rows of the launch's start requested at once, lane-contiguous
    double gb[JS_CH], gn[JS_CH], ab[JS_CH], an[JS_CH], gp[JS_CH];
    const size_t arow_b = (size_t)(shared ? 0 : m) * nz, arow_n = (size_t)(shared ? n : n + m) * nz;
    const size_t bzr = (size_t)m * nz, by = (size_t)m * ny;
    js_load_chunks<VEC>(gb, cc.b + (size_t)m * nz, lane, nz);
    js_load_chunks<VEC>(gn, cc.b + (size_t)(n + m) * nz, lane, nz);
    js_load_chunks<VEC>(ab, cc.area + arow_b, lane, nz);
    js_load_chunks<VEC>(an, cc.area + arow_n, lane, nz);
    js_load_chunks<VEC>(gp, a.Psi_SO + bzr, lane, nz);
    const double a0b = cc.area[arow_b], a0n = cc.area[arow_n];
    const double a0 = half ? a0n : a0b;
    ksel_b = cc.ksel[m];
    ksel_n = cc.ksel[n + m];
    {  // Area constant in z?  operands inside the exact-division window?
      bool same = true, ok = in_fast_div_range_or_lost(a0) && a0 != 0.0 &&
                             in_fast_div_range_or_lost(cc.bs[col]) &&
                             in_fast_div_range_or_lost(cc.N2min[col]) &&
                             in_fast_div_range_or_lost(cc.bbot[col]);
#pragma unroll
      for (int k = 0; k < JS_CH; ++k) {
        const bool real = js_ci<VEC>(k, lane) < nz;
        same = same && (!real || (ab[k] == a0b && an[k] == a0n));
        ok = ok && (!real || (in_fast_div_range_or_lost(gb[k]) && in_fast_div_range_or_lost(gn[k])));
      }
      hint_ok = __ballot(!same) == 0ull;
      range_ok = range_ok && __ballot(!ok) == 0ull;
    }
    if (hl == 0) {
      double *wsc = ws + half * S_COL;
      const double ra = 1.0 / a0;
      wsc[S_BS] = cc.bs[col];
      wsc[S_N2] = cc.N2min[col];
      wsc[S_ZC] = 0.;
      wsc[S_AREA] = a0;
      wsc[S_RAREA] = ra;
      wsc[S_RAREAL] = recip_lo(a0, ra);
    }
    // ---- state into the slot layout: basin through W_BB, north through the staging row
    js_stage<VEC>(wl + L::W_BB, dump, gb, lane, nz);
    js_stage<VEC>(wl + L::W_ST, dump, gn, lane, nz);
    __builtin_amdgcn_wave_barrier();
    {
      const double *src = wl + (half ? L::W_ST : L::W_BB);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = hl * P + p;
        const double v = src[i < nz ? i : nz - 1];  // (unconditional read, then a select)
        c.b[p] = i < nz ? v : JF_PAD;
        cv.cm[p] = 0ull;
      }
      cv.any = false;
      cv.any_b = false;
      cv.valid = false;
    }
    // static conditions of the BC switch: bit 0 Psi_SO[1] < 0, bit 1 Psi_SO[1] >= 0, bit 2
    // Psi_res_b[1] > 0, bit 3 Psi_res_n[1] < 0
    const double PsiSO1 = a.Psi_SO[bzr + 1], Pb1 = a.Psi_res_b[bzr + 1], Pn1 = a.Psi_res_n[bzr + 1];
    cbits = (PsiSO1 < 0 ? 1 : 0) | (PsiSO1 >= 0 ? 2 : 0) | (Pb1 > 0 ? 4 : 0) | (Pn1 < 0 ? 8 : 0);
    // ---- mixed layer: Psi_mod (SO_ML.py:228-230) is Psi_b with its leading zeros filled
#pragma unroll
    for (int k = 0; k < JS_CH; ++k) {
      const int i = js_ci<VEC>(k, lane);
      const bool real = i < nz;
      const unsigned long long nzm = __ballot(real && gp[k] != 0.), pm_ = __ballot(real && gp[k] > 0.);
      // (lowest set lane = lowest level of this element in both layouts)
      if (nzm) {
        const int j = js_ci<VEC>(k, (int)__ffsll((long long)nzm) - 1);
        ml_ind = j < ml_ind ? j : ml_ind;
      }
      if (pm_) {
        const int j = js_ci<VEC>(k, (int)__ffsll((long long)pm_) - 1);
        first_pos = j < first_pos ? j : first_pos;
      }
    }
    ml_ok = ml_ind < nz;  // all-zero Psi_b: IndexError in the reference
    const double fillv = ml_ok ? a.Psi_SO[bzr + ml_ind] : 0.;
    {
      double pmv[JS_CH];
#pragma unroll
      for (int k = 0; k < JS_CH; ++k) pmv[k] = (js_ci<VEC>(k, lane) < ml_ind) ? fillv : gp[k];
      js_stage<VEC>(wl + L::W_PM, dump, pmv, lane, nz);
    }
    double f1 = 0., f2 = 0., br = 0.;
    if (lane < ny) {
      q.bs = a.ml.bs[by + lane];
      if (ml_ok) {  // loop-invariant parts of the surface-flux tendency (SO_ML.py:250-252)
        const double ml_h = a.ml.h;
        f1 = a.ml.surflux[by + lane] / ml_h;
        f2 = a.ml.rest_mask[by + lane] * a.ml.v_pist / ml_h;
        br = a.ml.b_rest[by + lane];
      }
    }
    wl[L::W_F + 2 * lane] = f1;
    wl[L::W_F + 2 * lane + 1] = f2;
    wl[L::W_BR + lane] = br;
    if (lane == 0) ws[S_BBOT0] = a.cols.bbot[m];
  }
  __syncthreads();  // the block's tables (js_block_tables) are complete
  {  // the grid's part of the operand window (z, dz, dzc)
    bool ok = true;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = hl * P + p, e = js_entry<P>(hl, p);
      const double zv = lds[L::T_Z + e], dzv = lds[L::T_DZ + e], dzcv = lds[L::T_DZC + e];
      ok = ok && (i >= nz || (in_fast_div_range_or_lost(zv) &&
                              (i >= nz - 1 || (in_fast_div_range_or_lost(dzv) && dzv != 0.0)) &&
                              in_fast_div_range_or_lost(dzcv) && dzcv != 0.0));
    }
    range_ok = range_ok && __ballot(!ok) == 0ull;
  }
  int status = ml_ok ? 0 : 1;
  bool ps_valid = false;
  const bool lane0 = lane == 0;
  const bool ml_act = lane < ny, ml_int = lane >= 1 && lane <= ny - 2;

  int s = hint_ok ? 0 : nsteps;
  bool skipped = false;
  while (s < nsteps) {
    {
      // ---- coefficients of the sets in use: kappa, weff = wA - d(A kappa)/dz of both columns;
      // six lane-contiguous rows, transposed one array at a time (basin half through W_BB --
      // rewritten by the first step anyway --, northern half through the staging row)
      jf_kargs ka = jf_args();
      bool coef_ok = true;
      cv.valid = false;  // (the transposes below overwrite the cache of adjusted values in W_ST)
      const double *kappa = ka->cols.kappa, *dAk = ka->cols.dAkappa, *wA = ka->wA;
      const bool shared_rows = (ka->hints & PM_JN_SHARED_COEF) != 0;
      const size_t sb_b = ((size_t)ksel_b * (2 * n) + (shared_rows ? 0 : m)) * nz;
      const size_t sb_n = ((size_t)ksel_n * (2 * n) + (shared_rows ? n : n + m)) * nz;
      __builtin_amdgcn_sched_barrier(0);  // (the launch's first rows are consumed before these are requested)
      double kb_[JS_CH], kn_[JS_CH], wb_[JS_CH], wn_[JS_CH];
      js_load_chunks<VEC>(kb_, kappa + sb_b, lane, nz);
      js_load_chunks<VEC>(kn_, kappa + sb_n, lane, nz);
      {  // weff = wA - d(A kappa)/dz (column.py:241) while the rows are still lane-contiguous
        double db_[JS_CH], dn_[JS_CH];
        js_load_chunks<VEC>(db_, dAk + sb_b, lane, nz);
        js_load_chunks<VEC>(dn_, dAk + sb_n, lane, nz);
        js_load_chunks<VEC>(wb_, wA + (size_t)m * nz, lane, nz);
        js_load_chunks<VEC>(wn_, wA + (size_t)(n + m) * nz, lane, nz);
#pragma unroll
        for (int k = 0; k < JS_CH; ++k) {
          wb_[k] = wb_[k] - db_[k];
          wn_[k] = wn_[k] - dn_[k];
        }
      }
      const double *src = wl + (half ? L::W_ST : L::W_BB);
      double rk[P], rw[P];
      auto transpose = [&](double (&out)[P], const double (&g0)[JS_CH], const double (&g1)[JS_CH]) {
        __builtin_amdgcn_wave_barrier();
        js_stage<VEC>(wl + L::W_BB, dump, g0, lane, nz);
        js_stage<VEC>(wl + L::W_ST, dump, g1, lane, nz);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = hl * P + p;
          out[p] = src[i < nz ? i : nz - 1];
        }
      };
      transpose(rk, kb_, kn_);
      transpose(rw, wb_, wn_);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = hl * P + p;
        c.kap[p] = rk[p];
        const double w = rw[p];
        const bool interior = i >= 1 && i <= nz - 2;
        const double we = interior ? w : 0.0;
        c.wn[p] = (we < 0.0) ? -we : 0.0;
        c.wp[p] = (we < 0.0) ? 0.0 : -we;
        if constexpr (!CT)
          coef_ok = coef_ok && in_fast_div_range_or_lost(we) && in_fast_div_range_or_lost(c.kap[p]);
        if constexpr (CT) {
          // cu = dt (kappa / (dzc dz) + wn / (A dz)), cl = dt (-kappa / (dzc dz') + wp / (A dz'))
          // (col_make_contracted, column.hip.h)
          const int iq = interior ? i : 1;
          const double dz_up = lds[L::T_DZ + js_entry<P>(iq / P, iq % P)],
                       dz_dn = lds[L::T_DZ + js_entry<P>((iq - 1) / P, (iq - 1) % P)],
                       dzc = lds[L::T_DZC + js_entry<P>(iq / P, iq % P)];
          const double area = ws[half * S_COL + S_AREA];
          const double cu = dt * (c.kap[p] / (dzc * dz_up) + c.wn[p] / (area * dz_up));
          const double cl = dt * (-c.kap[p] / (dzc * dz_dn) + c.wp[p] / (area * dz_dn));
          c.wn[p] = interior ? cu : 0.0;
          c.wp[p] = interior ? cl : 0.0;
        }
      }
      if constexpr (!CT) {
        range_ok = range_ok && __ballot(!coef_ok) == 0ull;
        if (__builtin_expect(!range_ok, 0)) {
          skipped = true;  // the IEEE follow-up launch takes the member from step s on
          break;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // BC switch + both columns of step s; false: a coefficient set changed (nothing done yet)
    // static part of the BC switch (Psi only changes at MOC updates)
    const bool c_south = (cbits & 1) != 0, c_nosouth = (cbits & 2) != 0, c_pb = (cbits & 4) != 0,
               c_pn = (cbits & 8) != 0;
    const int kb_static = c_nosouth ? 0 : (c_south ? 1 : ksel_b);
    auto columns_step = [&](int lane_o, int hl_o, double *wl, double *ws, double *wsh) -> bool {
#ifndef JF_NO_PRIO_ROTATE
      {  // issue priority by lag (see jf_member_run)
        int *prog = reinterpret_cast<int *>(lds + L::PROG) + (wave & 3) * 4;
        prog[wave >> 2] = s;
        const int4 pv = *reinterpret_cast<const int4 *>(prog);
        const int p0 = __builtin_amdgcn_readfirstlane(pv.x), p1 = __builtin_amdgcn_readfirstlane(pv.y),
                  p2 = __builtin_amdgcn_readfirstlane(pv.z), p3 = __builtin_amdgcn_readfirstlane(pv.w);
        const int lo = min(min(p0, p1), min(p2, p3)), hi = max(max(p0, p1), max(p2, p3));
        if (s <= lo) __builtin_amdgcn_s_setprio(3);
        else if (s >= hi) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(1);
      }
#endif
      // ---- bottom-BC switch (run_JansenNadeau_2018.py:233-254).  Lane 0 decides the basin's
      // bottom value, lane 32 the northern one; each needs the OTHER column's level 0 (one
      // v_permlane32_swap) and its own level 1:
      //   basin: bn0 if Psi_res_b[1] > 0 and bn0 < bb1 and bn0 < bs_SO[0]   (from the north)
      //          else bb1 (no bottom water from the south) / bs_SO[0] (from the south) / unchanged
      //   north: bb0 if Psi_res_n[1] < 0 and bb0 < bn1                      (from the basin)
      //          else bn1
      // The static conditions are lane masks (bc_*), the diffusivity sets follow from two bits.
      double bbot;
      {
        const bool north = lane_o >= 32;
        const double o0 = jf_swap_halves(c.b[0], !north);
        const bool p1 = o0 < c.b[1], p2 = o0 < q.bs;
        // (mask algebra on the scalar unit: bit 0 decides for the basin, bit 32 for the north)
        const bool cnd = p1 && ((north && c_pn) || (!north && c_pb && p2));
        const unsigned long long cm_ = __builtin_amdgcn_ballot_w64(cnd);
        const double carry = (s == 0) ? ws[S_BBOT0] : c.b[0];
        const double alt = (north || c_nosouth) ? c.b[1] : (c_south ? q.bs : carry);
        bbot = cnd ? o0 : alt;
        const int kb = (cm_ & 1ull) ? 1 : kb_static;
        const int kn = (int)((cm_ >> 32) & 1ull);
        if (__builtin_expect(kb != ksel_b || kn != ksel_n, 0)) {
          JF_RARE(2)
          ksel_b = kb;
          ksel_n = kn;
          return false;
        }
      }
      // ---- basin.timestep / north.timestep, do_conv=True (:257-258)
      js_convect<P>(c.b, cv, lds, wsh, wl + L::W_ST, lane_o, hl_o, nz);
      c.b[0] = (hl_o == 0) ? bbot : c.b[0];  // column.py:232 (after convect: it may write level 0)
      if constexpr (CT) {
        js_vertadvdiff_contracted<P>(c);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        js_vertadvdiff<P>(c, lds, wsh, hl_o, dt);
        __builtin_amdgcn_sched_barrier(0);
      }
      return true;
    };
    if (ml_ok) {
    for (; s < nsteps; ++s) {
      int lane_o = lane, woff = wave * L::PER_WAVE;
      asm volatile("" : "+v"(lane_o), "+s"(woff));
      const int a4 = lane_o << 2, hl_o = lane_o & 31;
      double *wl = lds + L::WAVE0 + woff, *ws = wl + L::W_S;
      double *wsh = ws + (lane_o >> 5) * S_COL;
      if (__builtin_expect(!columns_step(lane_o, hl_o, wl, ws, wsh), 0)) break;
      JF_RARE(0)
      // ---- channel.timestep(b_basin=basin.b, Psi_b=PsiSO.Psi) (:261), ml_step_reg's operations
      {
        const double *bb = wl + L::W_BB, *pm = wl + L::W_PM;
        if (lane_o < 32) {  // basin.b by level for np.interp
          double *dst = wl + L::W_BB + hl_o * P;
#pragma unroll
          for (int p = 0; p < P; ++p) dst[p] = c.b[p];
        }
        __builtin_amdgcn_wave_barrier();
        double ps;
        {
          const double x = q.bs;
          const int j0 = q.jh;  // <= nz - 2
          const double x0 = bb[j0], x1 = bb[j0 + 1];
          int j = j0 + (x >= x1 ? 1 : 0) - (x < x0 ? 1 : 0);
          j = j < 0 ? 0 : (j > nz - 2 ? nz - 2 : j);
          const double lx = bb[j], hx = bb[j + 1], lf = pm[j], hf = pm[j + 1];
          const double xlo = bb[0], xhi = bb[nz - 1];
          const bool hit = (lx <= x) && (x < hx);
          ps = interp_finish(x, j, lx, hx, lf, hf);
          const bool isnan_x = x != x, above = x >= xhi, below = x < xlo;
          if (below) ps = pm[0];
          if (above) ps = pm[nz - 1];
          if (isnan_x) ps = x;
          q.jh = j;
          const bool search = ml_act && !hit && !isnan_x && !above && !below;
          if (__builtin_expect(__ballot(search) != 0ull, 0)) {
            JF_RARE(3)
            if (search) {  // np.interp's upper-bound search from scratch
              int lo_i = 0, hi_i = nz;
              while (lo_i < hi_i) {
                const int mid = lo_i + ((hi_i - lo_i) >> 1);
                if (x >= bb[mid])
                  lo_i = mid + 1;
                else
                  hi_i = mid;
              }
              const int jb = lo_i - 1;
              const int jj = (jb == nz - 1) ? nz - 2 : (jb < 0 ? 0 : jb);
              q.jh = jj;
              ps = (jb == nz - 1) ? pm[nz - 1]
                                  : interp_finish(x, jb, bb[jj], bb[jj + 1], pm[jj], pm[jj + 1]);
            }
          }
        }
        // argmin(bs): first minimum, a NaN wins (np.argmin); 0 when no point lies below point 0
        int amin = 0;
        {
          const double v = ml_act ? q.bs : __builtin_inf();
          const double v0 = lane_value(q.bs, 0);
          if (__builtin_expect(__ballot(!(v >= v0)) != 0ull, 0)) {
            JF_RARE(5)
            const double mn = wave_min_f64(v);
            const unsigned long long at_min = __ballot(ml_act && v == mn);
            const unsigned long long nanm = __ballot(ml_act && v != v);
            const int mi = at_min ? (int)__ffsll((long long)at_min) - 1 : 0;
            amin = nanm ? (int)__ffsll((long long)nanm) - 1 : mi;
          }
        }
        if (lane < amin || lane0) ps = 0.;  // :240-243
        const bool upwell = (__ballot(ps > 0) & 2ull) != 0ull;  // set_boundary_conditions, :93-98
        if (__builtin_expect(upwell && first_pos >= nz, 0)) {
          ml_ok = false;  // IndexError in the reference; the mixed layer stops evolving
          status = 1;
          ++s;
          break;
        } else {
          __builtin_amdgcn_sched_barrier(0);
          const double2 *K2 = reinterpret_cast<const double2 *>(lds + L::KML);
          const double2 k0 = K2[0], k1 = K2[1], k2 = K2[2], k3 = K2[3], k4 = K2[4];
          const double k_1ms = lds[L::KML + K_1MS];
          const double2 ff = *reinterpret_cast<const double2 *>(wl + L::W_F + 2 * lane_o);
          const double brest = wl[L::W_BR + lane_o];
          const double bsouth = upwell ? bb[first_pos < nz ? first_pos : 0] : 0.;
          double bs = q.bs;
          const double bs_up = from_next_lane_z(bs);
          bs = lane0 ? (upwell ? bsouth : bs_up) : bs;
          const double bs_dn = from_prev_lane_z(bs);
          const double flux = ff.x + ff.y * (brest - bs);  // (:250-259)
          double adv;
          {
            const double d = (ps < 0.) ? (bs_up - bs) : (bs - bs_dn);
            const double num = -ps * 1e6 * d;
            const double q1 = div_by_recip2(num, k1.x, k0.y, k0.x);
            const double q2 = div_by_recip2(q1, k2.y, k2.x, k1.y);
            const double t = div_by_recip2(q2, k4.x, k3.y, k3.x);
            adv = (ml_int && ps != 0. && ps == ps) ? t : 0.;
          }
          bs = bs + dt * (flux + adv);  // every tendency uses the old bs
          const double bu = from_next_lane_z(bs);
          if (!upwell) bs = lane0 ? bu : bs;  // no-flux BC re-set (:264-266)
          {  // Crank-Nicolson diffusion (:191-196): U x = V bs by parallel cyclic reduction
            const double bl = from_prev_lane_z(bs);
            const double sh = k4.y;
            double r = bs;  // rows 0 and ny-1 of V are identity rows
            if (ml_int) r = sh * bl + k_1ms * bs + sh * bu;
            if (!ml_act) r = 0.;
            bs = jf_pcr_solve(r, lds + L::PCR, lane_o, a4);
          }
          {
            const double up = from_next_lane_z(bs);
            const double v2 = upwell ? bsouth : up;  // final BC re-set (:274)
            bs = lane0 ? v2 : bs;
          }
          q.bs = bs;
          wl[L::W_PS + lane_o] = ps;
          ps_valid = true;
        }
      }
    }
    } else {
      for (; s < nsteps; ++s) {
        int lane_o = lane, woff = wave * L::PER_WAVE;
        asm volatile("" : "+v"(lane_o), "+s"(woff));
        double *wl = lds + L::WAVE0 + woff, *ws = wl + L::W_S;
        if (__builtin_expect(!columns_step(lane_o, lane_o & 31, wl, ws, ws + (lane_o >> 5) * S_COL), 0))
          break;
      }
    }
  }

  // ---- results: slot layout -> level order through LDS -> lane-contiguous stores
  jf_kargs ka = jf_args();
  const size_t by = (size_t)m * ny;
  bool bad = false;
  {
    __builtin_amdgcn_wave_barrier();
    double *dst = wl + (half ? L::W_ST : L::W_BB);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = hl * P + p;
      if (i < nz) {
        dst[i] = c.b[p];
        bad |= !isfinite(c.b[p]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (m_ok && hint_ok) {
      double *bout = ka->cols.b;
      if constexpr (VEC) {
#pragma unroll
        for (int kk = 0; kk < JS_CH / 2; ++kk) {
          const int i0 = 128 * kk + 2 * lane;
          if (i0 < nz) {
            *reinterpret_cast<double2 *>(bout + (size_t)m * nz + i0) =
                *reinterpret_cast<const double2 *>(wl + L::W_BB + i0);
            *reinterpret_cast<double2 *>(bout + (size_t)(n + m) * nz + i0) =
                *reinterpret_cast<const double2 *>(wl + L::W_ST + i0);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < JS_CH; ++k) {
          const int i = 64 * k + lane;
          if (i < nz) {
            bout[(size_t)m * nz + i] = wl[L::W_BB + i];
            bout[(size_t)(n + m) * nz + i] = wl[L::W_ST + i];
          }
        }
      }
    }
  }
  if (lane < ny) {
    bad |= !isfinite(q.bs);
    if (m_ok && hint_ok) {
      ka->ml.bs[by + lane] = q.bs;
      double *Psi_s = ka->ml.Psi_s;
      if (Psi_s && ps_valid) Psi_s[by + lane] = wl[L::W_PS + lane];
    }
  }
  const bool anybad = __ballot(bad) != 0ull;
  if (hl == 0 && m_ok) {
    if (hint_ok && s > 0) {  // Column.bbot after the last step = what that step imposed on level 0
      double *bbot = const_cast<double *>(ka->cols.bbot);
      int32_t *ksel = const_cast<int32_t *>(ka->cols.ksel);
      bbot[col] = c.b[0];
      ksel[col] = half ? ksel_n : ksel_b;
    }
    int32_t *nonfinite = ka->cols.nonfinite;
    if (nonfinite) nonfinite[col] = anybad ? 1 : 0;
    int32_t *st = ka->ml.status;
    if (st && lane == 0)
      st[m] = status | (anybad ? 2 : 0) | (hint_ok ? 0 : 16) | (skipped ? 32 : 0) |
              (skipped ? (s << 8) : 0);
  }
  return s;
}

template <int P, bool CT, bool VEC>
__global__ __launch_bounds__(64 * JF_WAVES) JF_OCC_ATTR
void k_jn2018_split(pm_jn2018 a, double dt, int nsteps) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  js_block_tables<P>(a, dt, lds, wave, lane);
  js_member_run<P, CT, VEC>(a, dt, nsteps, blockIdx.x * JF_WAVES + wave, lds, wave, lane);
}

template <int P>
static void launch_split_kernel(const pm_jn2018 &a, double dt, int nsteps, bool vec, hipStream_t st) {
  const size_t lds = (size_t)JsLds<P>::TOTAL * sizeof(double);
  const unsigned grid = (unsigned)((a.n + JF_WAVES - 1) / JF_WAVES);
  const bool ct = (a.hints & PM_JN_CONTRACTED) != 0;
#define JS_LAUNCH(CT_, VEC_)                                                                     \
  hipLaunchKernelGGL((k_jn2018_split<P, CT_, VEC_>), dim3(grid), dim3(64 * JF_WAVES), lds, st, a, \
                     dt, nsteps)
  if (ct && vec)
    JS_LAUNCH(true, true);
  else if (ct)
    JS_LAUNCH(true, false);
  else if (vec)
    JS_LAUNCH(false, true);
  else
    JS_LAUNCH(false, false);
#undef JS_LAUNCH
}

template <int P>
static int launch_fast(const pm_jn2018 &a, double dt, int nsteps, hipStream_t st) {
  const size_t lds = (size_t)JfLds<P>::TOTAL * sizeof(double);
  const unsigned grid = (unsigned)((a.n + JF_WAVES - 1) / JF_WAVES);
  // 16-byte row accesses (jf_load_row): every row starts 16-byte aligned and holds whole lanes
  auto al = [](const void *q) { return (((unsigned long long)q) & 15ull) == 0ull; };
  const bool vec = a.cols.nz % P == 0 && al(a.cols.b) && al(a.cols.area) && al(a.cols.kappa) &&
                   al(a.cols.dAkappa) && al(a.wA) && al(a.Psi_SO);
  const bool ct = (a.hints & PM_JN_CONTRACTED) != 0;
  // round 5: both columns of a member side by side in the wave's two halves (k_jn2018_split).
  // Opt-in (PM_JN_SPLIT_LANES, or PYMOC_JN_SPLIT=1 for A/B runs): 7 % fewer vector instructions
  // per launch, measured a tie on config 5; shapes 65..128 and 193..224
  static const bool env_split = getenv("PYMOC_JN_SPLIT") != nullptr;
  const int PS = (a.cols.nz + 31) / 32;
  const bool split = ((a.hints & PM_JN_SPLIT_LANES) || env_split) && (PS == 3 || PS == 4 || PS == 7);
  if (split) {
    // 16-byte row accesses: every row starts 16-byte aligned and holds an even number of levels
    const bool v2 = a.cols.nz % 2 == 0 && al(a.cols.b) && al(a.cols.area) && al(a.cols.kappa) &&
                    al(a.cols.dAkappa) && al(a.wA) && al(a.Psi_SO);
    switch (PS) {
      case 3: launch_split_kernel<3>(a, dt, nsteps, v2, st); break;
      case 4: launch_split_kernel<4>(a, dt, nsteps, v2, st); break;
      default: launch_split_kernel<7>(a, dt, nsteps, v2, st); break;
    }
  }
#define JF_LAUNCH(CT_, VEC_)                                                                  \
  hipLaunchKernelGGL((k_jn2018_fast<P, CT_, VEC_>), dim3(grid), dim3(64 * JF_WAVES), lds, st, a, \
                     dt, nsteps)
  // PM_JN_DIV3_PROVEN: the caller's proof for every static denominator (pm_div3_proven)
  const bool d3 = (a.hints & PM_JN_DIV3_PROVEN) != 0 && !ct;
  if (split)
    ;
  else if (d3 && vec)
    hipLaunchKernelGGL((k_jn2018_fast<P, false, true, true>), dim3(grid), dim3(64 * JF_WAVES), lds, st,
                       a, dt, nsteps);
  else if (d3)
    hipLaunchKernelGGL((k_jn2018_fast<P, false, false, true>), dim3(grid), dim3(64 * JF_WAVES), lds, st,
                       a, dt, nsteps);
  else if (ct && vec)
    JF_LAUNCH(true, true);
  else if (ct)
    JF_LAUNCH(true, false);
  else if (vec)
    JF_LAUNCH(false, true);
  else
    JF_LAUNCH(false, false);
#undef JF_LAUNCH
  PM_HIP(hipGetLastError());
  if (!ct && a.ml.status) {  // the IEEE leg of the members this launch flagged
    const size_t lds1 = (size_t)(JfLds<P>::WAVE0 + JfLds<P>::PER_WAVE) * sizeof(double);
    const unsigned g1 = (unsigned)((a.n + 63) / 64 < 64 ? (a.n + 63) / 64 : 64);
    if (vec)
      hipLaunchKernelGGL((k_jn2018_ieee<P, true>), dim3(g1), dim3(64), lds1, st, a, dt, nsteps);
    else
      hipLaunchKernelGGL((k_jn2018_ieee<P, false>), dim3(g1), dim3(64), lds1, st, a, dt, nsteps);
    PM_HIP(hipGetLastError());
  }
  return PM_OK;
}

bool jn2018_fast_applies(const pm_jn2018 &a) {
  // (status: how the launch hands members outside the exact-division window to its IEEE
  // follow-up launch)
  return (a.hints & PM_JN_UNIFORM_AREA) != 0 && a.ml.ny <= 64 && a.cols.nz <= 256 &&
         a.cols.nz >= 4 && a.ml.status != nullptr;
}

int launch_jn2018_fast(const pm_jn2018 &a, double dt, int nsteps, hipStream_t st) {
  return a.cols.nz <= 128 ? launch_fast<2>(a, dt, nsteps, st) : launch_fast<4>(a, dt, nsteps, st);
}

}  // namespace pm

#include "coupled_run.hip.h"
#endif
