// column.hip.h -- K1: the advective-diffusive column update, fused over nsteps.
//
// Arithmetic restated from the reference (nothing copied):
//   Column.convect      src/pymoc/modules/column.py:251-271
//   Column.vertadvdiff  src/pymoc/modules/column.py:210-249
//   Column.horadv       src/pymoc/modules/column.py:288-313
//   Column.timestep     src/pymoc/modules/column.py:315-348
// Every operation is an elementwise IEEE fp64 +,-,*,/ in the reference's order
// (compiled with -ffp-contract=off), so results are bit-identical to NumPy.
//
// Work decomposition: G lanes (16, 32 or 64) of one wavefront own one column; lane
// lg holds the P contiguous levels [lg*P, lg*P+P) in registers for the whole launch.
// The 3-point stencil needs one value from each neighbour lane per step (b of the
// level above, bz of the interface below): two DPP wave shifts, no LDS, no barrier.
#pragma once
#include <type_traits>
#include "common.hip.h"

namespace pm {

template <int P>
struct ColGrid {     // shared by every column of a batch (and by the columns of one member)
  double z[P];     // level depth
  double dz[P];    // z[i+1]-z[i]           (interface above level i)
  double dzc[P];   // 0.5*(dz[i]+dz[i-1])   (column.py:238)
  double rdz[P], rdzc[P];  // RN(1/dz) (0 above the top level), RN(1/dzc)
  double rdz_l[P], rdzc_l[P];  // low parts: RN(1/d - RN(1/d)), for div_by_recip2 (DIV == 2)
};
template <int P>
struct ColRegs {
  double b[P];     // state
  double kap[P];   // kappa(z_i)
  double area[P];  // Area(z_i)
  double dAk[P];   // d(Area*kappa)/dz at z_i (np.gradient, host precomputed)
  double rarea[P]; // RN(1/area)
  double rarea_l[P];  // low part of 1/area (DIV == 2)
  // Area constant in z (UA instantiations): one wave-uniform value instead of the arrays
  double area_u, rarea_u, rarea_lu;
};

// P consecutive levels of one column row (`row` points at level 0 of the column) starting at
// level lvl0 + lg*P.  For P == 2, even nz and a 16-byte aligned row the lane issues ONE 16-byte
// load (a wave then reads 1 KiB contiguously) instead of two 8-byte loads with stride 16;
// padding lanes re-read the last pair.  Otherwise: per-level loads clamped to the last level.
template <int P>
__device__ __forceinline__ void load_levels(double (&out)[P], const double *__restrict__ row,
                                            int lg, int nz, int lvl0 = 0) {
  if constexpr (P == 2) {
    if (lvl0 == 0 && (nz & 1) == 0 && (((unsigned long long)row) & 15ull) == 0ull) {
      const int i0 = lg * 2 < nz - 2 ? lg * 2 : nz - 2;
      const double2 v = *reinterpret_cast<const double2 *>(row + i0);
      out[0] = v.x;
      out[1] = v.y;
      return;
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lvl0 + lg * P + p;
    out[p] = row[i < nz ? i : nz - 1];
  }
}

// Value held for level j (wave-uniform index) of a wave-owned column: v_readlane of every slot
// of the owning lane, then a scalar select.  (Selecting the slot per lane first makes the
// compiler index the register array dynamically, i.e. spill the whole grid to scratch.)
template <int P>
__device__ __forceinline__ double level_value(const double (&x)[P], int j) {
  const int jl = __builtin_amdgcn_readfirstlane(j / P);
  const int jp = __builtin_amdgcn_readfirstlane(j % P);
  double out = 0.;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x[p]), jl);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x[p]), jl);
    const double v = __hiloint2double(hi, lo);
    out = (p == 0 || jp == p) ? v : out;
  }
  return out;
}

// a lane's value as a wave-uniform scalar (two v_readlane)
__device__ __forceinline__ double lane_value(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// Convecting-level pattern of the previous step and its zconv: persistent convection keeps
// the same levels convecting for many steps, so the find-last-bit / readlane chain that
// locates zconv is skipped while the pattern is unchanged (G = 64 only).
template <int P>
struct ConvCache {
  unsigned long long mask[P];
  double zconv;
  bool valid = false;
};

// Column.convect (column.py:251-271).  `zg` is the shared grid in global memory.
// Returns true when some level convected (rare), false when only b[-1] = bs was imposed.
template <int G, int P>
__device__ __forceinline__ bool col_convect(double (&b)[P], const double (&z)[P],
                                            double bs, double N2min, int lg, int lane,
                                            int nz, const double *__restrict__ zg,
                                            ConvCache<P> *cache = nullptr) {
  const unsigned long long gm = group_mask<G>(lane);
  bool ind[P];
  unsigned long long im[P];  // per-slot ballot of the convecting levels (column.py:264)
  unsigned long long anym = 0ull;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lg * P + p;
    ind[p] = (i < nz) && (b[p] > bs);
    im[p] = __builtin_amdgcn_ballot_w64(ind[p]);
    anym |= im[p] & gm;
  }
  if (__builtin_expect(anym != 0ull, 0)) {
    double zconv;
    bool hit = false;
    if constexpr (G == 64) {
      if (cache != nullptr && cache->valid) {
        hit = true;
#pragma unroll
        for (int p = 0; p < P; ++p) hit = hit && (cache->mask[p] == im[p]);
      }
    }
    if (hit) {
      zconv = cache->zconv;
    } else {
      // zconv = max(z[~ind]) (column.py:267): z ascends, so it is z at the highest
      // non-convecting level; bottom of the ocean if every level convects.
      int jmax = 0;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = lg * P + p;
        const unsigned long long m = __builtin_amdgcn_ballot_w64((i < nz) && !ind[p]) & gm;
        if (m != 0ull) {
          const int hl = 63 - __clzll((long long)m);
          const int j = (hl % G) * P + p;
          jmax = j > jmax ? j : jmax;
        }
      }
      if constexpr (G == 64) {
        // the wave owns the whole column: fetch z[jmax] from the owning lane's registers
        // (two v_readlane) instead of a scalar memory load in the middle of the time loop
        zconv = level_value<P>(z, jmax);
        if (cache != nullptr) {
#pragma unroll
          for (int p = 0; p < P; ++p) cache->mask[p] = im[p];
          cache->zconv = zconv;
          cache->valid = true;
        }
      } else {
        zconv = zg[jmax];
      }
    }
    // (selects, not `if (ind[p])` blocks: with the blocks hipcc 7.2 lost slot 0's adjusted value
    // on the IEEE path of k_column_stream -- an inf level survived convect in the release
    // build and not in one with a printf next to it; test_operands_outside_..._streaming_kernel)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const double adj = bs + N2min * (z[p] - zconv);  // column.py:268
      b[p] = ind[p] ? adj : b[p];
    }
    return true;
  }
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (lg * P + p == nz - 1) b[p] = bs;  // column.py:271
  return false;
}

// Column.convect for a wave-owned column (G = 64) inside a time loop, branch-free in the
// steady state: the convecting-level pattern (one ballot per slot, scalar registers) is
// compared with the previous step's; only when it CHANGES does the wave take the branch that
// re-derives zconv.  The adjustment itself is applied with selects every step.
template <int P>
__device__ __forceinline__ void col_convect_cached(double (&b)[P], const double (&z)[P],
                                                   double bs, double N2min, int lane, int nz,
                                                   ConvCache<P> &cc) {
  bool ind[P];
  unsigned long long im[P], anym = 0ull;
  bool same = cc.valid;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    ind[p] = (lane * P + p < nz) && (b[p] > bs);  // column.py:264
    im[p] = __builtin_amdgcn_ballot_w64(ind[p]);
    anym |= im[p];
    same = same && (im[p] == cc.mask[p]);
  }
  if (__builtin_expect(!same, 0)) {
    // zconv = max(z[~ind]) (column.py:267): z at the highest non-convecting level, bottom of
    // the ocean if every level convects
    int jmax = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64((lane * P + p < nz) && !ind[p]);
      if (m != 0ull) {
        const int j = (63 - __clzll((long long)m)) * P + p;
        jmax = j > jmax ? j : jmax;
      }
    }
    cc.zconv = level_value<P>(z, jmax);
#pragma unroll
    for (int p = 0; p < P; ++p) cc.mask[p] = im[p];
    cc.valid = true;
  }
  const bool none = anym == 0ull;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double adj = bs + N2min * (z[p] - cc.zconv);  // column.py:268
    b[p] = ind[p] ? adj : b[p];
    if (lane * P + p == nz - 1) b[p] = none ? bs : b[p];  // column.py:271
  }
}

// Column.vertadvdiff (column.py:210-249), one explicit step.  DIV selects how the three
// divisions by static denominators are done -- 0: IEEE `/`; 1: div_by_recip (5 instructions,
// needs RN(1/d)); 2: div_by_recip2 (4 instructions, needs the double-double reciprocal); 3: the
// two divisions by grid metrics as in 1, the division by Area as in 0 (streaming kernel: the
// grid is shared by the columns a wave walks through, the areas are not).  All are correctly
// rounded, hence bit-identical to each other and to NumPy.
// BC = false: the caller has already imposed the (constant) boundary values, which no
// interior update ever touches -- valid when bzbot is None and the surface value is bs.
// WEFF: `wA` already holds weff = wA - d(A kappa)/dz (callers that keep it across steps).
// UA: Area is constant in z -- r.area_u / rarea_u / rarea_lu replace the per-level arrays.
// FLUXFMA: the select-free upwind flux (needs wn / wp, two more values per level, which the
// register-starved fused JN2018 loop cannot afford).
// DIV == 0 is also the REFERENCE-FAITHFUL form for any operand, finite or not: IEEE division,
// compare-select flux, and boundary / padding levels left untouched by a select (the other
// forms advance them with dt = 0, which turns a level next to an inf or NaN into NaN).
template <int G, int P, int DIV, bool BC = true, bool WEFF = false, bool UA = false,
          bool FLUXFMA = (DIV == 2 || DIV == 6)>
__device__ __forceinline__ void col_vertadvdiff(const ColGrid<P> &g, ColRegs<P> &r,
                                                const double (&wA)[P],
                                                double dt, bool do_conv, double bs,
                                                double bbot, bool use_bzbot,
                                                double bzbot, int lg, int nz,
                                                int lvl0 = 0) {
  // level of slot p of this lane: lvl0 + lg*P + p
  // surface boundary condition (column.py:230-231)
  if (BC && !do_conv) {
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (lvl0 + lg * P + p == nz - 1) r.b[p] = bs;
  }
  // b at the level above each owned level
  // edge lanes only need a finite value here (their levels are boundary / padding slots)
  const double nb0 = from_next_lane_z(r.b[0]);
  double bup[P];
#pragma unroll
  for (int p = 0; p < P; ++p) bup[p] = (p < P - 1) ? r.b[p + 1 < P ? p + 1 : p] : nb0;
  // bottom boundary condition (column.py:232-233); level 0 = lane 0, slot 0
  if (BC && lvl0 + lg * P == 0) r.b[0] = use_bzbot ? (bup[0] - bzbot * g.dz[0]) : bbot;

  if constexpr (DIV == 4) {
    // Contracted form (opt-in tolerance mode, PM_OP_CONTRACTED).  With weff, kappa, Area and
    // the grid static over a launch, column.py:235-249 is linear in the state:
    //   b_i += cu_i (b_{i+1} - b_i) + cl_i (b_i - b_{i-1}),
    // with the per-launch coefficients of col_make_contracted in r.kap (cu) and r.dAk (cl),
    // zero on boundary and padding levels: one subtraction and two fma per level instead of 21
    // instructions.  Not the reference's operation order: agrees to rounding (1e-12 over the
    // BASELINE runs, tests/test_column_gpu.py), not bit for bit.
    double d_up[P];
#pragma unroll
    for (int p = 0; p < P; ++p) d_up[p] = bup[p] - r.b[p];
    const double pd = from_prev_lane_z(d_up[P - 1]);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const double d_dn = (p > 0) ? d_up[p > 0 ? p - 1 : 0] : pd;
      r.b[p] = __builtin_fma(r.kap[p], d_up[p], __builtin_fma(r.dAk[p], d_dn, r.b[p]));
    }
    return;
  }
  // Every stage below is written across the P slots so that the in-order wave always has
  // P independent dependency chains in flight (one wave per SIMD is latency-bound).
  double bz[P];  // (b[i+1]-b[i])/dz[i]  (column.py:235)
  {
    double num[P], q[P], rr[P];
#pragma unroll
    for (int p = 0; p < P; ++p) num[p] = bup[p] - r.b[p];
    if constexpr (DIV == 2) {
      // u = a*yl; q = fma(a, yh, u) (faithful); r = fma(-d, q, a); q = fma(r, yh, q)
#pragma unroll
      for (int p = 0; p < P; ++p) rr[p] = num[p] * g.rdz_l[p];
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = __builtin_fma(num[p], g.rdz[p], rr[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) rr[p] = __builtin_fma(-g.dz[p], q[p], num[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = __builtin_fma(rr[p], g.rdz[p], q[p]);
    } else if constexpr (DIV == 6) {
      // q0 = a*y; r = fma(-d, q0, a); q = fma(r, y, q0): correctly rounded for EVERY numerator when
      // the denominator has passed pm_div3_proven (PM_COLS_DIV3_PROVEN; common.hip.h)
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = num[p] * g.rdz[p];
#pragma unroll
      for (int p = 0; p < P; ++p) rr[p] = __builtin_fma(-g.dz[p], q[p], num[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = __builtin_fma(rr[p], g.rdz[p], q[p]);
    } else if constexpr (DIV == 1 || DIV == 3 || DIV == 5) {
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = num[p] * g.rdz[p];
#pragma unroll
      for (int p = 0; p < P; ++p) rr[p] = __builtin_fma(-g.dz[p], q[p], num[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = __builtin_fma(rr[p], g.rdz[p], q[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) rr[p] = __builtin_fma(-g.dz[p], q[p], num[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = __builtin_fma(rr[p], g.rdz[p], q[p]);
    } else {
#pragma unroll
      for (int p = 0; p < P; ++p) q[p] = num[p] / g.dz[p];
    }
    if constexpr (DIV != 0) {
      // rdz (and its low part) is 0 above the top level, so q is already 0 there
#pragma unroll
      for (int p = 0; p < P; ++p) bz[p] = q[p];
    } else {
#pragma unroll
      for (int p = 0; p < P; ++p) bz[p] = (lvl0 + lg * P + p < nz - 1) ? q[p] : 0.0;
    }
  }
  const double pbz = from_prev_lane_z(bz[P - 1]);
  double bz_dn[P], dbz[P], flx[P], bzz[P], adv[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    bz_dn[p] = (p > 0) ? bz[p > 0 ? p - 1 : 0] : pbz;
    dbz[p] = bz[p] - bz_dn[p];
    const double weff = WEFF ? wA[p] : wA[p] - r.dAk[p];  // column.py:241
    if constexpr (FLUXFMA) {
      // upwind flux (-weff)*bz* (column.py:242-246) without a select: weff is static over
      // the launch, so exactly one of the two factors below is (-weff) and the other 0, and
      // fma(wn, bz, RN(wp*bz_dn)) = RN((-weff)*bz*) -- the unselected product is an exact
      // zero for finite gradients.  (Differences to the select form: the sign of a zero
      // flux, which no later operation can tell apart, and NaN instead of a finite flux
      // next to an infinite gradient, i.e. only inside members that are already blown up.)
      const double wn = (weff < 0.0) ? -weff : 0.0, wp = (weff < 0.0) ? 0.0 : -weff;
      flx[p] = __builtin_fma(wn, bz[p], wp * bz_dn[p]);
    } else {
      const double bzu = (weff < 0.0) ? bz[p] : bz_dn[p];  // column.py:242-243
      flx[p] = (-weff) * bzu;
    }
  }
  if constexpr (DIV == 2) {  // bzz = dbz/dzc (:238) and adv = flx/Area (:246), interleaved
    double r1[P], r2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      r1[p] = dbz[p] * g.rdzc_l[p];
      r2[p] = flx[p] * (UA ? r.rarea_lu : r.rarea_l[p]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = __builtin_fma(dbz[p], g.rdzc[p], r1[p]);
      adv[p] = __builtin_fma(flx[p], (UA ? r.rarea_u : r.rarea[p]), r2[p]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      r1[p] = __builtin_fma(-g.dzc[p], bzz[p], dbz[p]);
      r2[p] = __builtin_fma(-(UA ? r.area_u : r.area[p]), adv[p], flx[p]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = __builtin_fma(r1[p], g.rdzc[p], bzz[p]);
      adv[p] = __builtin_fma(r2[p], (UA ? r.rarea_u : r.rarea[p]), adv[p]);
    }
  } else if constexpr (DIV == 6) {  // both by the proven 3-instruction quotient, interleaved
    double r1[P], r2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = dbz[p] * g.rdzc[p];
      adv[p] = flx[p] * (UA ? r.rarea_u : r.rarea[p]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      r1[p] = __builtin_fma(-g.dzc[p], bzz[p], dbz[p]);
      r2[p] = __builtin_fma(-(UA ? r.area_u : r.area[p]), adv[p], flx[p]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = __builtin_fma(r1[p], g.rdzc[p], bzz[p]);
      adv[p] = __builtin_fma(r2[p], (UA ? r.rarea_u : r.rarea[p]), adv[p]);
    }
  } else if constexpr (DIV == 5) {  // grid as DIV == 3; Area (UA: one number) by its double-double reciprocal
    double r1[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = dbz[p] * g.rdzc[p];
      adv[p] = div_by_recip2(flx[p], r.area_u, r.rarea_u, r.rarea_lu);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int p = 0; p < P; ++p) r1[p] = __builtin_fma(-g.dzc[p], bzz[p], dbz[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) bzz[p] = __builtin_fma(r1[p], g.rdzc[p], bzz[p]);
    }
  } else if constexpr (DIV == 3) {  // grid division by reciprocal, area by IEEE division
    double r1[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = dbz[p] * g.rdzc[p];
      adv[p] = flx[p] / (UA ? r.area_u : r.area[p]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int p = 0; p < P; ++p) r1[p] = __builtin_fma(-g.dzc[p], bzz[p], dbz[p]);
#pragma unroll
      for (int p = 0; p < P; ++p) bzz[p] = __builtin_fma(r1[p], g.rdzc[p], bzz[p]);
    }
  } else if constexpr (DIV == 1) {
    double r1[P], r2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = dbz[p] * g.rdzc[p];
      adv[p] = flx[p] * (UA ? r.rarea_u : r.rarea[p]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        r1[p] = __builtin_fma(-g.dzc[p], bzz[p], dbz[p]);
        r2[p] = __builtin_fma(-(UA ? r.area_u : r.area[p]), adv[p], flx[p]);
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        bzz[p] = __builtin_fma(r1[p], g.rdzc[p], bzz[p]);
        adv[p] = __builtin_fma(r2[p], (UA ? r.rarea_u : r.rarea[p]), adv[p]);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bzz[p] = dbz[p] / g.dzc[p];
      adv[p] = flx[p] / (UA ? r.area_u : r.area[p]);
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lvl0 + lg * P + p;
    const bool interior = (i >= 1) && (i <= nz - 2);
    const double db_dt = adv[p] + r.kap[p] * bzz[p];  // column.py:245-248
    // column.py:249.  Boundary / padding slots advance with dt = 0 (b + 0*x == b for the
    // finite x they hold) instead of a select, which the compiler would turn back into
    // one exec-masked block per slot and serialise the chains.
    if constexpr (DIV == 0) {
      const double nb = r.b[p] + dt * db_dt;
      r.b[p] = interior ? nb : r.b[p];
    } else {
      r.b[p] = r.b[p] + (interior ? dt : 0.0) * db_dt;
    }
  }
}

// Column.horadv (column.py:288-313)
template <int P>
__device__ __forceinline__ void col_horadv(ColRegs<P> &r, const double (&vdx)[P],
                                           const double (&bin)[P], double dt, int lg,
                                           int nz) {
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lg * P + p;
    if (i < nz && vdx[p] > 0.0) {
      const double db = bin[p] - r.b[p];
      r.b[p] = r.b[p] + dt * vdx[p] * db / r.area[p];
    }
  }
}

// True when every operand of this lane's levels lies in the window of the exact-division
// shortcuts (common.hip.h: in_fast_div_range): the kernels combine it over the wave and take
// the IEEE form (DIV = 0) otherwise.
template <int P>
__device__ __forceinline__ bool col_inputs_in_fast_range(const ColGrid<P> &g, const ColRegs<P> &r,
                                                         const double (&wA)[P], double dt,
                                                         double bs, double bbot, double bzbot,
                                                         double N2min, int lg, int nz) {
  bool ok = in_fast_div_range(dt) && in_fast_div_range(bs) && in_fast_div_range(bbot) &&
            in_fast_div_range(bzbot) && in_fast_div_range(N2min);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const bool top = lg * P + p >= nz - 1;  // (dz is 0 at and above the top level)
    ok = ok && in_fast_div_range(r.b[p]) && in_fast_div_range(wA[p]) &&
         in_fast_div_range(r.dAk[p]) && in_fast_div_range(wA[p] - r.dAk[p]) &&
         in_fast_div_range(r.kap[p]) && in_fast_div_range(r.area[p]) && r.area[p] != 0.0 &&
         in_fast_div_range(g.z[p]) && (top || (in_fast_div_range(g.dz[p]) && g.dz[p] != 0.0)) &&
         in_fast_div_range(g.dzc[p]) && g.dzc[p] != 0.0;
  }
  return ok;
}

// The part of that test that changes from launch to launch: the state and weff = wA - d(A
// kappa)/dz (wA and d(A kappa)/dz enter the step only through their difference).
template <int P>
__device__ __forceinline__ bool col_state_in_fast_range(const ColRegs<P> &r, const double (&wA)[P]) {
  // (bit operations, not &&: the short-circuit form compiled to a branch per operand)
  int ok = 1;
#pragma unroll
  for (int p = 0; p < P; ++p)
    ok &= (int)in_fast_div_range(r.b[p]) & (int)in_fast_div_range(wA[p] - r.dAk[p]);
  return ok != 0;
}

// grid metrics of the batch into registers
template <int P, int DIV = 1>
__device__ __forceinline__ void col_load_grid(ColGrid<P> &g, const pm_columns &c, int lg,
                                              int lvl0 = 0) {
  const int nz = c.nz;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lvl0 + lg * P + p;
    const int ic = i < nz ? i : nz - 1;
    const int iu = ic + 1 < nz ? ic + 1 : nz - 1;
    const int id = ic > 0 ? ic - 1 : 0;
    const double zc = c.z[ic];
    g.z[p] = zc;
    g.dz[p] = c.z[iu] - zc;
    g.dzc[p] = 0.5 * (g.dz[p] + (zc - c.z[id]));
    if constexpr (DIV != 0 && DIV != 4) {  // reciprocals only where div_by_recip will use them
      g.rdz[p] = (i < nz - 1) ? 1.0 / g.dz[p] : 0.0;  // 0: bz above the top level is 0
      g.rdzc[p] = 1.0 / g.dzc[p];
    } else {
      g.rdz[p] = g.rdzc[p] = 0.0;
    }
    if constexpr (DIV == 2) {
      g.rdz_l[p] = (i < nz - 1) ? recip_lo(g.dz[p], g.rdz[p]) : 0.0;
      g.rdzc_l[p] = recip_lo(g.dzc[p], g.rdzc[p]);
    } else {
      g.rdz_l[p] = g.rdzc_l[p] = 0.0;
    }
  }
}

// static coefficients of one column (coefficient set `sel`) into registers
// `uniform_area` (wave-uniform; the column's PM_COL_UNIFORM_AREA hint): Area(z) is one number --
// one broadcast load and one reciprocal instead of a row from HBM and a division per level;
// `weff_in` (PM_OP_WEFF): the forcing already holds wA - d(A kappa)/dz, so that row is not read.
template <int P, int DIV = 1>
__device__ __forceinline__ void col_load_static(ColRegs<P> &r, const pm_columns &c,
                                                int col, int sel, int lg, int lvl0 = 0,
                                                bool uniform_area = false, bool weff_in = false) {
  const int nz = c.nz;
  const size_t base = (size_t)col * nz;
  const size_t sbase = ((size_t)sel * c.ncols + col) * nz;
  load_levels<P>(r.kap, c.kappa + sbase, lg, nz, lvl0);
  if (!weff_in) {
    load_levels<P>(r.dAk, c.dAkappa + sbase, lg, nz, lvl0);
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) r.dAk[p] = 0.0;  // weff - 0 = weff, exactly
  }
  if (uniform_area) {
    const double a0 = c.area[base];
    const double ra = (DIV != 0 && DIV != 4) ? 1.0 / a0 : 0.0;
    const double ral = DIV == 2 ? recip_lo(a0, ra) : 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      r.area[p] = a0;
      r.rarea[p] = ra;
      r.rarea_l[p] = ral;
    }
    return;
  }
  load_levels<P>(r.area, c.area + base, lg, nz, lvl0);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    r.rarea[p] = (DIV != 0 && DIV != 4) ? 1.0 / r.area[p] : 0.0;
    r.rarea_l[p] = DIV == 2 ? recip_lo(r.area[p], r.rarea[p]) : 0.0;
  }
}

// Coefficients of the contracted step (col_vertadvdiff, DIV == 4) of a wave-owned column:
//   cu_i = dt (kappa_i / (dzc_i dz_i)     + [weff_i <  0] (-weff_i) / (A_i dz_i))      -> r.kap
//   cl_i = dt (-kappa_i / (dzc_i dz_{i-1}) + [weff_i >= 0] (-weff_i) / (A_i dz_{i-1})) -> r.dAk
// (the diffusion and upwind-advection terms of column.py:235-249 collected by the difference they
// multiply); zero on boundary and padding levels, which therefore stay put.
template <int P>
__device__ __forceinline__ void col_make_contracted(const ColGrid<P> &g, ColRegs<P> &r,
                                                    const double (&wA)[P], double dt, int lane,
                                                    int nz) {
  const double dz_prev_lane = from_prev_lane_z(g.dz[P - 1]);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    const bool interior = i >= 1 && i <= nz - 2;
    const double weff = wA[p] - r.dAk[p];
    const double dz_up = interior ? g.dz[p] : 1.0;
    const double dz_dn = interior ? ((p > 0) ? g.dz[p > 0 ? p - 1 : 0] : dz_prev_lane) : 1.0;
    const double dzc = interior ? g.dzc[p] : 1.0, area = interior ? r.area[p] : 1.0;
    const double wn = (weff < 0.0) ? -weff : 0.0, wp = (weff < 0.0) ? 0.0 : -weff;
    const double cu = dt * (r.kap[p] / (dzc * dz_up) + wn / (area * dz_up));
    const double cl = dt * (-r.kap[p] / (dzc * dz_dn) + wp / (area * dz_dn));
    r.kap[p] = interior ? cu : 0.0;
    r.dAk[p] = interior ? cl : 0.0;
  }
}

// Slots (bit p = slot p) in which the cached pattern has convecting lanes.  P <= 2 keeps one
// specialised time loop per slot set; taller lanes only distinguish "none" from "some".
template <int P>
__device__ __forceinline__ unsigned conv_variant(const ConvCache<P> &cc) {
  unsigned need = 0u;
#pragma unroll
  for (int p = 0; p < P; ++p) need |= (cc.mask[p] != 0ull) ? (1u << p) : 0u;
  if constexpr (P <= 2) {
    return need;
  } else {
    return need ? (1u << P) - 1u : 0u;
  }
}

// Speculative convective time loop of a wave-owned column (G = 64), steps [s, nsteps).
// Every step adjusts with the facts of the last ESTABLISHED convecting pattern cc (its zconv,
// hence the adjusted values bs + N2min (z - zconv), column.py:268) and issues the step's
// arithmetic; whether the pattern really was the cached one is checked once per block of UNR
// steps, and only a changed pattern (rare) redoes the block exactly from the saved b and
// re-establishes cc.  Nothing leaves the registers before the check, so steps computed past
// a pattern change are simply discarded.
//   * SEL names the slots whose cached mask is non-zero.  Those slots compare (`b > bs`), select
//     the adjusted value and fold `mask ^ cached` into a scalar accumulator (s_xor / s_or).
//   * In the other slots nothing convects under the cached pattern, so they need no select, and
//     "still nothing convects" is one running v_max per step, compared with bs once per block
//     (a NaN never convects in the reference -- `b > bs` is False -- and v_max ignores it).
// Why blocks: for a lone wave every scalar instruction costs a vector issue slot and a
// conditional branch several; the 3-instruction loop control and the check's compare + branch
// are paid once per UNR steps.  Returns the number of steps done when the new pattern belongs
// to another SEL class (the caller re-dispatches) or at nsteps.
template <int P, int DIV, unsigned SEL, bool UA = false>
__device__ __forceinline__ int conv_spec_run(const ColGrid<P> &g, ColRegs<P> &r,
                                             const double (&wA)[P], double dt, double bs,
                                             double bbot, double N2min, int lane, int nz,
                                             ConvCache<P> &cc, int s, int nsteps) {
  constexpr int UNR = P <= 2 ? 16 : (P <= 4 ? 2 : 1);
  double adjv[P];    // column.py:268 for the cached zconv: changes only with the pattern
  double bs_eff[P];  // bs for real levels, +inf for padding: `b > bs_eff` is the whole test
  // Level 0 always ends a step's convect + boundary condition holding bbot (column.py:232
  // overwrites whatever convect wrote), so its "adjusted" value is bbot itself.
  auto set_adj = [&]() {
#pragma unroll
    for (int p = 0; p < P; ++p)
      adjv[p] = (lane * P + p == 0) ? bbot : bs + N2min * (g.z[p] - cc.zconv);
  };
  set_adj();
#pragma unroll
  for (int p = 0; p < P; ++p) bs_eff[p] = (lane * P + p < nz) ? bs : __builtin_inf();
  double b_blk[P];          // state at the start of the current block
  unsigned long long acc;   // OR of (mask ^ cached) over the SEL slots and steps of the block
  double bmax[P];           // running max of b over the block, slots outside SEL
  auto spec = [&]() {
    unsigned long long im[P];
    bool ind[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (((SEL >> p) & 1u) != 0u) {  // folds after unrolling
        ind[p] = r.b[p] > bs_eff[p];  // column.py:264
        im[p] = __builtin_amdgcn_ballot_w64(ind[p]);
      } else {
        bmax[p] = __builtin_fmax(bmax[p], r.b[p]);
      }
    }
    // all compares first, then the selects: a select issued right behind its own compare costs
    // an s_nop (VALU-written SGPR read as a VALU mask)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (((SEL >> p) & 1u) != 0u) r.b[p] = ind[p] ? adjv[p] : r.b[p];
      // column.py:271 (b[-1] = bs when nothing convects) needs no work here: the step that
      // established the cached pattern imposed it, and neither vertadvdiff (the surface level
      // advances with dt = 0) nor an unchanged pattern alters that level
    }
    col_vertadvdiff<64, P, DIV, false, false, UA>(g, r, wA, dt, true, bs, bbot, false, 0., lane, nz);
    // (mask ^ cached) AFTER the step's arithmetic has been issued (no wait for the compare
    // results).  The xor is asm because the optimiser rewrites the C form of the whole test
    // into compare + select chains (more scalar instructions).
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (((SEL >> p) & 1u) != 0u) {
        unsigned long long x;
        asm("s_xor_b64 %0, %1, %2" : "=s"(x) : "s"(im[p]), "s"(cc.mask[p]) : "scc");
        acc |= x;
      }
    }
  };
  auto block_begin = [&]() {
    acc = 0ull;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      b_blk[p] = r.b[p];
      bmax[p] = -__builtin_inf();
    }
  };
  auto block_bad = [&]() -> bool {
    bool conv = false;  // a lane of a slot outside SEL would convect
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (((SEL >> p) & 1u) == 0u) conv = conv || (bmax[p] > bs_eff[p]);
    if constexpr (SEL != (1u << P) - 1u) {
      // opaque to the optimiser: it would turn `ballot(conv) != 0` back into a DIVERGENT
      // branch on conv, and the structuriser then wraps the whole loop in exec-mask logic
      unsigned long long any = __builtin_amdgcn_ballot_w64(conv);
      asm("" : "+s"(any));
      acc |= any;
    }
    return acc != 0ull;
  };
  // redo the block that started at b_blk exactly (pattern changed somewhere inside it)
  auto redo = [&](int nb) {
#pragma unroll
    for (int p = 0; p < P; ++p) r.b[p] = b_blk[p];
    for (int k = 0; k < nb; ++k) {
      col_convect_cached<P>(r.b, g.z, bs, N2min, lane, nz, cc);
      if (lane == 0) r.b[0] = bbot;
      col_vertadvdiff<64, P, DIV, false, false, UA>(g, r, wA, dt, true, bs, bbot, false, 0., lane, nz);
    }
  };
  // Blocks of UNR steps, then of 4, then of 1.  The only loop-carried scalar of a tier is the count of blocks left;
  // single-exit loops (a `return` inside makes the compiler wrap the hot path in flag logic): a
  // pattern of another SEL class ends the loop through its own counter.
  int pos = s, ret = -1;
  auto tier = [&](auto U) {
    constexpr int N = decltype(U)::value;
    int left = (nsteps - pos) / N;
    const int s_end = pos + left * N;
    for (; left > 0; --left) {
      block_begin();
#pragma unroll
      for (int k = 0; k < N; ++k) spec();
      if (__builtin_expect(block_bad(), 0)) {
        redo(N);
        if (conv_variant<P>(cc) != SEL) {
          ret = s_end - (left - 1) * N;
          left = 1;
        } else {
          set_adj();
        }
      }
    }
    pos = s_end;
  };
  // the long blocks only in long launches: a pattern change redoes the whole block, and the
  // coupled drivers' 24-step launches (new forcing, hence often a new pattern, every launch)
  // came out 9 % slower with them
  if (UNR <= 4 || nsteps - pos >= 4 * UNR) tier(std::integral_constant<int, UNR>{});
  if (ret >= 0) return ret;
  if constexpr (UNR > 4) {
    tier(std::integral_constant<int, 4>{});
    if (ret >= 0) return ret;
  }
  if constexpr (UNR > 1) tier(std::integral_constant<int, 1>{});
  return ret >= 0 ? ret : nsteps;
}

// result of a column into HBM (16-byte stores for P == 2) and its non-finite flag
template <int G, int P>
__device__ __forceinline__ void col_store_result(const pm_columns &c, const ColRegs<P> &r,
                                                 size_t base, int col, bool col_ok, int lg,
                                                 int lane, int nz) {
  bool bad = false;
  bool stored = false;
  if constexpr (P == 2) {
    if ((nz & 1) == 0 && (((unsigned long long)(c.b + base)) & 15ull) == 0ull) {
      if (lg * 2 < nz) {  // even nz: both levels of the pair are valid
        if (col_ok) *reinterpret_cast<double2 *>(c.b + base + lg * 2) = make_double2(r.b[0], r.b[1]);
        bad = !isfinite(r.b[0]) || !isfinite(r.b[1]);
      }
      stored = true;
    }
  }
  if (!stored) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lg * P + p;
      if (i < nz) {
        if (col_ok) c.b[base + i] = r.b[p];
        bad |= !isfinite(r.b[p]);
      }
    }
  }
  if (c.nonfinite) {
    const unsigned long long m = __ballot(bad) & group_mask<G>(lane);
    if (lg == 0 && col_ok) c.nonfinite[col] = (m != 0ull) ? 1 : 0;
  }
}

// Pins a column's state in its registers between convect and the step that follows it: no
// instruction, but the optimiser cannot merge across it (see k_column_stream: hipcc 7.2 lost a
// convected slot in such a merge).  Used on the paths that are not instruction-count critical.
template <int P>
__device__ __forceinline__ void col_pin(double (&b)[P]) {
#pragma unroll
  for (int p = 0; p < P; ++p) asm volatile("" : "+v"(b[p]));
}

// PLAIN: ops == PM_OP_TIMESTEP without horadv inputs -- the time loop then carries no
// loop-invariant branches (they cost a lone wave ~15% of a step).
// UA (launch-level: pm_columns.reserved & PM_COLS_ALL_UNIFORM_AREA): Area and its reciprocal pair
// are three scalars instead of three P-element arrays -- 12 registers at P = 2, which is what
// separates 4 from 5 waves per SIMD (108 -> 96).
template <int G, int P, int FAST, bool PLAIN, bool UA = false>
__global__ __launch_bounds__(256) void k_column_steps(
    pm_columns c, const double *__restrict__ wA_g, const double *__restrict__ vdx_g,
    const double *__restrict__ bin_g, double dt, int nsteps, int ops) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int lg = threadIdx.x % G;
  const int col_raw = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / G);
  const bool col_ok = col_raw < c.ncols;
  const int col = col_ok ? col_raw : c.ncols - 1;  // idle groups shadow the last column
  const int nz = c.nz;
  const size_t base = (size_t)col * nz;

  ColGrid<P> g;
  ColRegs<P> r;
  const int sel = c.ksel ? c.ksel[col] : 0;
  const int flags = c.flags ? c.flags[col] : 0;
  col_load_grid<P, FAST>(g, c, lg);
  // (a wave holds one column when G == 64: the hint is then wave-uniform)
  const bool ua = G == 64 && __builtin_amdgcn_readfirstlane(flags & PM_COL_UNIFORM_AREA) != 0;
  col_load_static<P, FAST>(r, c, col, sel, lg, 0, ua, (ops & PM_OP_WEFF) != 0);
  if constexpr (UA) {  // (the arrays col_load_static filled with the same number die here)
    r.area_u = lane_value(r.area[0], 0);
    r.rarea_u = lane_value(r.rarea[0], 0);
    r.rarea_lu = lane_value(r.rarea_l[0], 0);
  }

  double wA[P], vdx[P], bin[P];
  load_levels<P>(r.b, c.b + base, lg, nz);
#pragma unroll
  for (int p = 0; p < P; ++p) wA[p] = vdx[p] = bin[p] = 0.0;
  if (wA_g) load_levels<P>(wA, wA_g + base, lg, nz);
  if constexpr (PLAIN) {
    if (ops & PM_OP_WA_PSI) {
      // the two-column drivers' forcing from the overturning itself (example_twocol_plusSO.py:
      // 105-106; pm_thermwind_update's wA1 / wA2 epilogue, same operations): basin rows
      // (Psi_iso - Psi_SO) * 1e6, northern rows -Psi_iso * 1e6
      const int half = c.ncols >> 1;
      if (col < half) {
        if (bin_g) {
          double pso[P];
          load_levels<P>(pso, bin_g + base, lg, nz);
#pragma unroll
          for (int p = 0; p < P; ++p) wA[p] = (wA[p] - pso[p]) * 1e6;
        } else {
#pragma unroll
          for (int p = 0; p < P; ++p) wA[p] = wA[p] * 1e6;
        }
      } else {
#pragma unroll
        for (int p = 0; p < P; ++p) wA[p] = (-wA[p]) * 1e6;
      }
    }
  }
  if constexpr (PLAIN) {
    if (ops & PM_OP_WA_TWOBASIN) {
      // the two-basin driver's forcing (twobasin_NadeauJansen.py:103-105; pm_twobasin_forcing's
      // operations): wA_g / vdx_g / bin_g = the AMOC's, the zonal overturning's isopycnal
      // overturnings and the two sectors' Psi_SO, [2 ncols / 3][nz] each; wA holds wA_g's row
      const int third = c.ncols / 3;
      if (col < third) {  // Atlantic: (iso_A + zon_A - SO_A) * 1e6
        double zon[P], pso[P];
        load_levels<P>(zon, vdx_g + base, lg, nz);
        load_levels<P>(pso, bin_g + base, lg, nz);
#pragma unroll
        for (int p = 0; p < P; ++p) wA[p] = (wA[p] + zon[p] - pso[p]) * 1e6;
      } else if (col < 2 * third) {  // north: -iso_N * 1e6
#pragma unroll
        for (int p = 0; p < P; ++p) wA[p] = (-wA[p]) * 1e6;
      } else {  // Pacific: (-zon_P - SO_P) * 1e6 (rows [n, 2n) of the two arrays)
        const size_t bp = (size_t)(col - third) * nz;
        double zon[P], pso[P];
        load_levels<P>(zon, vdx_g + bp, lg, nz);
        load_levels<P>(pso, bin_g + bp, lg, nz);
#pragma unroll
        for (int p = 0; p < P; ++p) wA[p] = (-zon[p] - pso[p]) * 1e6;
      }
    }
  }
  const bool wa_formed = (ops & (PM_OP_WA_PSI | PM_OP_WA_TWOBASIN)) != 0;  // (the slots held overturnings)
  if (vdx_g && !wa_formed) load_levels<P>(vdx, vdx_g + base, lg, nz);
  if (bin_g && !wa_formed) load_levels<P>(bin, bin_g + base, lg, nz);
  const bool do_conv = (flags & PM_COL_DO_CONV) != 0;
  const bool use_bzbot = (flags & PM_COL_BZBOT) != 0 && c.bzbot != nullptr;
  const double bs = c.bs[col];
  const double bbot = c.bbot[col];
  const double bzbot = use_bzbot ? c.bzbot[col] : 0.0;
  const double N2min = c.N2min[col];

  if constexpr (FAST != 0) {
    // operands outside the exact-division window (a column scaled by 2^-1000, an inf level, ...):
    // the whole wave steps in the reference-faithful IEEE form
    // (PM_COL_STATIC_IN_RANGE: the caller vouches for the static operands -- grid, coefficients,
    // boundary values -- so only what changes from launch to launch is tested: the state and the
    // forcing; ~115 of the ~1500 vector instructions of a 24-step launch)
    // (a wave holds one column when G == 64: the hint is then wave-uniform and the choice a branch)
    const bool vouched = G == 64 &&
                         __builtin_amdgcn_readfirstlane(flags & PM_COL_STATIC_IN_RANGE) != 0 &&
                         in_fast_div_range(dt);
    bool fast_ok;
    if (vouched)
      fast_ok = col_state_in_fast_range<P>(r, wA);
    else
      fast_ok = col_inputs_in_fast_range<P>(g, r, wA, dt, bs, bbot, bzbot, N2min, lg, nz);
    if (__builtin_expect(__ballot(!fast_ok) != 0ull, 0)) {
      for (int s = 0; s < nsteps; ++s) {
        if ((ops & PM_OP_CONVECT) && do_conv)
          col_convect<G, P>(r.b, g.z, bs, N2min, lg, lane, nz, c.z);
        col_pin<P>(r.b);
        if (ops & PM_OP_VERTADVDIFF)
          col_vertadvdiff<G, P, 0>(g, r, wA, dt, do_conv, bs, bbot, use_bzbot, bzbot, lg, nz);
        if ((ops & PM_OP_HORADV) && vdx_g) col_horadv<P>(r, vdx, bin, dt, lg, nz);
      }
      col_store_result<G, P>(c, r, base, col, col_ok, lg, lane, nz);
      return;
    }
  }

  if constexpr (FAST == 4) col_make_contracted<P>(g, r, wA, dt, lane, nz);  // (G == 64 only)

  if constexpr (PLAIN) {
    if (do_conv && use_bzbot) {
      for (int s = 0; s < nsteps; ++s) {
        col_convect<G, P>(r.b, g.z, bs, N2min, lg, lane, nz, c.z);
        col_vertadvdiff<G, P, FAST, true, false, UA>(g, r, wA, dt, true, bs, bbot, true, bzbot, lg, nz);
      }
    } else if (do_conv) {
      // b[0] = bbot is constant unless a convection event rewrites level 0: impose it once
      // and again after such an event (the reference re-imposes it every step, column.py:232)
      ConvCache<P> cc;
#pragma unroll
      for (int p = 0; p < P; ++p) cc.mask[p] = 0ull;
      cc.zconv = 0.;
      if constexpr (G == 64) {
        col_convect_cached<P>(r.b, g.z, bs, N2min, lane, nz, cc);  // step 0: establishes cc
        if (lg == 0) r.b[0] = bbot;
        col_vertadvdiff<G, P, FAST, false, false, UA>(g, r, wA, dt, true, bs, bbot, false, 0., lg, nz);
        int s = 1;
        while (s < nsteps) {  // one pass per established pattern class (rarely more than one)
          const unsigned v = conv_variant<P>(cc);
          if constexpr (P <= 2) {
            switch (v) {
              case 0: s = conv_spec_run<P, FAST, 0u, UA>(g, r, wA, dt, bs, bbot, N2min, lane, nz, cc, s, nsteps); break;
              case 1: s = conv_spec_run<P, FAST, 1u, UA>(g, r, wA, dt, bs, bbot, N2min, lane, nz, cc, s, nsteps); break;
              case 2: s = conv_spec_run<P, FAST, 2u, UA>(g, r, wA, dt, bs, bbot, N2min, lane, nz, cc, s, nsteps); break;
              default: s = conv_spec_run<P, FAST, 3u, UA>(g, r, wA, dt, bs, bbot, N2min, lane, nz, cc, s, nsteps); break;
            }
          } else {
            if (v == 0u)
              s = conv_spec_run<P, FAST, 0u, UA>(g, r, wA, dt, bs, bbot, N2min, lane, nz, cc, s, nsteps);
            else
              s = conv_spec_run<P, FAST, (1u << P) - 1u, UA>(g, r, wA, dt, bs, bbot, N2min, lane, nz, cc, s, nsteps);
          }
        }
      } else {
        for (int s = 0; s < nsteps; ++s) {
          col_convect<G, P>(r.b, g.z, bs, N2min, lg, lane, nz, c.z);
          if (lg == 0) r.b[0] = bbot;  // column.py:232 (a convection event may rewrite level 0)
          col_vertadvdiff<G, P, FAST, false, false, UA>(g, r, wA, dt, true, bs, bbot, false, 0., lg, nz);
        }
      }
    } else if (use_bzbot) {
      for (int s = 0; s < nsteps; ++s)
        col_vertadvdiff<G, P, FAST, true, false, UA>(g, r, wA, dt, false, bs, bbot, true, bzbot, lg, nz);
    } else {
      // constant boundary values: impose them once, then run the BC-free step
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (lg * P + p == nz - 1) r.b[p] = bs;
        if (lg * P + p == 0) r.b[p] = bbot;
      }
      // unrolled: the loop control (3 scalar instructions) costs a lone wave as much as 3
      // vector instructions
      int s = 0;
      if constexpr (P <= 2) {
        for (; s + 4 <= nsteps; s += 4) {
          col_vertadvdiff<G, P, FAST, false, false, UA>(g, r, wA, dt, false, bs, bbot, false, 0., lg, nz);
          col_vertadvdiff<G, P, FAST, false, false, UA>(g, r, wA, dt, false, bs, bbot, false, 0., lg, nz);
          col_vertadvdiff<G, P, FAST, false, false, UA>(g, r, wA, dt, false, bs, bbot, false, 0., lg, nz);
          col_vertadvdiff<G, P, FAST, false, false, UA>(g, r, wA, dt, false, bs, bbot, false, 0., lg, nz);
        }
      }
      for (; s < nsteps; ++s)
        col_vertadvdiff<G, P, FAST, false, false, UA>(g, r, wA, dt, false, bs, bbot, false, 0., lg, nz);
    }
  } else {
    for (int s = 0; s < nsteps; ++s) {
      if ((ops & PM_OP_CONVECT) && do_conv)
        col_convect<G, P>(r.b, g.z, bs, N2min, lg, lane, nz, c.z);
      col_pin<P>(r.b);
      if (ops & PM_OP_VERTADVDIFF)
        col_vertadvdiff<G, P, FAST, true, false, UA>(g, r, wA, dt, do_conv, bs, bbot, use_bzbot, bzbot, lg, nz);
      if ((ops & PM_OP_HORADV) && vdx_g) col_horadv<P>(r, vdx, bin, dt, lg, nz);
    }
  }

  col_store_result<G, P>(c, r, base, col, col_ok, lg, lane, nz);
}

// ------------------------------------------------------------------ streaming kernel
// One or two steps per launch on a large ensemble (config 1 style loops, Psi refreshed every
// step): the state cannot stay in registers across launches, every launch streams b, the
// forcing and the coefficient arrays in and b out -- 32 nz ... 48 nz B per column-step (below),
// HBM is the roof.  k_column_steps pays per column what only depends on the shared grid (loading
// z, forming dz / dzc, three IEEE divisions per level ~ 40 quarter-rate instructions); here a
// wave walks through `cpw` consecutive columns and forms the grid metrics and their reciprocals
// ONCE.  What bounds the kernel is the number of bytes in flight: a column's loads take ~4 us
// under load, so the wave keeps the loads of the next D columns in flight (a ring of D stages in
// registers, indices static after unrolling) while it computes the current one; with one stage
// (round 2) the 5-array form reached 4.8 TB/s, the 3-array form only 3.4.  The per-column
// scalars (flags, coefficient set, bs, bbot, bzbot, N2min) are fetched for all the wave's
// columns by ONE vector load each (lane i = i-th column) and handed out by v_readlane: a scalar
// load per column sat between a column's flags and its vector loads.
//   weff_in (PM_OP_WEFF): wA_g holds weff = wA - d(A kappa)/dz, d(A kappa)/dz is not read;
//   PM_COL_UNIFORM_AREA:  Area(z) is one number, read with the scalars;
// together 32 nz instead of 48 nz bytes per column-step.
template <int P>
struct ColStage {
  double b[P], wA[P], kap[P], area[P], dAk[P];
};
// LEAN (forcing precombined, Area one number per column, kappa formed from its two factors): a
// column in flight is b and weff -- 4 P registers per ring stage instead of 10 P
template <int P>
struct ColStageLean {
  double b[P], wA[P];
};

constexpr int STREAM_MAX_CPW = 64;  // columns per wave <= lanes (the scalars' vector load)

// the wave's per-column scalars, lane i = column col0 + i
struct StreamScalars {
  int flags, sel;
  double bs, bbot, bzbot, N2min, area0, kbase;
  double ra, ral;  // RN(1 / area0) and its low part, for ALL the wave's columns by one vector division
  __device__ __forceinline__ void load(const pm_columns &c, int col0, int cend, int lane) {
    const int col = col0 + lane < cend ? col0 + lane : cend - 1;
    flags = c.flags ? c.flags[col] : 0;
    sel = c.ksel ? c.ksel[col] : 0;
    bs = c.bs[col];
    bbot = c.bbot[col];
    N2min = c.N2min[col];
    bzbot = ((flags & PM_COL_BZBOT) != 0 && c.bzbot != nullptr) ? c.bzbot[col] : 0.0;
    area0 = (flags & PM_COL_UNIFORM_AREA) ? c.area[(size_t)col * c.nz] : 0.0;
    kbase = c.kappa_base ? c.kappa_base[col] : 0.0;
    ra = 1.0 / area0;
    ral = recip_lo(area0, ra);
  }
};

// AFF: kappa is formed from kappa_base + kappa_profile (pm_columns), not read
template <int P, bool AFF = false>
__device__ __forceinline__ void col_stage_load(ColStage<P> &s, const pm_columns &c,
                                               const double *__restrict__ wA_g, int col,
                                               int lane, bool weff_in, int flags, int sel,
                                               double area0) {
  const int nz = c.nz;
  const size_t base = (size_t)col * nz;
  const size_t sbase = ((size_t)sel * c.ncols + col) * nz;
  load_levels<P>(s.b, c.b + base, lane, nz);
  load_levels<P>(s.wA, wA_g + base, lane, nz);
  if constexpr (!AFF) load_levels<P>(s.kap, c.kappa + sbase, lane, nz);
  if ((flags & PM_COL_UNIFORM_AREA) == 0) {
    load_levels<P>(s.area, c.area + base, lane, nz);
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) s.area[p] = area0;
  }
  if (!weff_in) {
    load_levels<P>(s.dAk, c.dAkappa + sbase, lane, nz);
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) s.dAk[p] = 0.0;  // weff - 0 = weff, exactly
  }
}

// VEC (lean form, P == 2; the launcher checks: nz even, b and weff 16-byte aligned): a column's
// two levels per lane move as ONE 16-byte access at a loop-invariant lane offset from a scalar
// base.  (load_levels takes that decision per call; inside the ring the compiler turned it
// into both address forms, selects, and two 8-byte loads.)
// D3 (lean form; PM_COLS_DIV3_PROVEN): the step's three quotients by the proven 3-instruction form
// CPWU > 0 (the launcher: every wave has exactly CPWU columns): the wave's columns as
// STRAIGHT-LINE code.  Inside a loop the compiler's wait-count bookkeeping forgets the order of
// the loads issued before the loop header, and the first use of a ring stage waits for every
// load but the two issued since (s_waitcnt vmcnt(2)): the ring drained once per D columns.
// Unrolled, every wait names exactly the loads that lie between.
// WEFFT / UABT (the straight-line non-lean forms): `weff_in` and the batch-wide uniform-Area hint as
// compile-time facts, so that a stage's loads are UNCONDITIONAL (a load under a run-time condition
// is merged into the ring's registers by moves, which wait for it: see the refill above).
template <int P, int D, bool AFF = false, bool LEAN = false, bool VEC = false, bool D3 = false,
          int CPWU = 0, int WEFFT = -1, int UABT = -1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((LEAN && P <= 2 && D <= 5) ? 4 : 1)))
void k_column_stream(pm_columns c,
                                                       const double *__restrict__ wA_g,
                                                       double dt, int nsteps, int cpw, bool dt_ok,
                                                       bool weff_in) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
  const int col0 = __builtin_amdgcn_readfirstlane(wave * cpw);
  const int nz = c.nz;
  if (col0 >= c.ncols) return;  // wave-uniform
  const int cend = col0 + cpw < c.ncols ? col0 + cpw : c.ncols;

  StreamScalars sc;
  sc.load(c, col0, cend, lane);
  static_assert(!LEAN || AFF, "the lean ring needs the affine kappa");
  static_assert(!VEC || (LEAN && P == 2), "16-byte accesses: the lean form with two levels per lane");
  const int off2 = lane * 2 < nz - 2 ? lane * 2 : nz - 2;  // padding lanes re-read the last pair
  using Stage = typename std::conditional<LEAN, ColStageLean<P>, ColStage<P>>::type;
  auto flags_of = [&](int k) { return __builtin_amdgcn_readlane(sc.flags, k); };
  auto issue = [&](Stage &st, int col) {
    const int k = col - col0;
    if constexpr (VEC) {
      const size_t base = (size_t)col * nz;
      const double2 vb = *reinterpret_cast<const double2 *>(c.b + base + off2);
      const double2 vw = *reinterpret_cast<const double2 *>(wA_g + base + off2);
      st.b[0] = vb.x;
      st.b[P - 1] = vb.y;
      st.wA[0] = vw.x;
      st.wA[P - 1] = vw.y;
    } else if constexpr (LEAN) {
      const size_t base = (size_t)col * nz;
      load_levels<P>(st.b, c.b + base, lane, nz);
      load_levels<P>(st.wA, wA_g + base, lane, nz);
    } else if constexpr (CPWU > 0 && WEFFT >= 0 && UABT >= 0) {
      const int sel = __builtin_amdgcn_readlane(sc.sel, k);
      const size_t base = (size_t)col * nz, sbase = ((size_t)sel * c.ncols + col) * nz;
      load_levels<P>(st.b, c.b + base, lane, nz);
      load_levels<P>(st.wA, wA_g + base, lane, nz);
      if constexpr (!AFF) load_levels<P>(st.kap, c.kappa + sbase, lane, nz);
      if constexpr (UABT == 0) load_levels<P>(st.area, c.area + base, lane, nz);
      if constexpr (WEFFT == 0) load_levels<P>(st.dAk, c.dAkappa + sbase, lane, nz);
    } else {
      col_stage_load<P, AFF>(st, c, wA_g, col, lane, weff_in, flags_of(k),
                             __builtin_amdgcn_readlane(sc.sel, k), lane_value(sc.area0, k));
    }
  };
  ColGrid<P> g;
  col_load_grid<P, 1>(g, c, lane);
  double kprof[P];  // AFF: the shared part of kappa, once per wave
#pragma unroll
  for (int p = 0; p < P; ++p) kprof[p] = 0.0;
  if constexpr (AFF) load_levels<P>(kprof, c.kappa_profile, lane, nz);
  Stage ring[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (col0 + d < cend) issue(ring[d], col0 + d);
  unsigned nf_bits = 0u;  // (CPWU: the columns' non-finite flags, written once at the end)
  auto one_column = [&](int col, int d, bool prefetch, int kk) {
      ColRegs<P> r;
      double wA[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        r.b[p] = ring[d].b[p];
        wA[p] = ring[d].wA[p];
        if constexpr (AFF)  // kappa[col][i] = kappa_base[col] + kappa_profile[i]: the caller's identity
          r.kap[p] = lane_value(sc.kbase, col - col0) + kprof[p];
        else
          r.kap[p] = ring[d].kap[p];
        if constexpr (LEAN) {
          r.area[p] = lane_value(sc.area0, col - col0);
          r.area_u = r.area[p];
          r.rarea_u = lane_value(sc.ra, col - col0);
          r.rarea_lu = lane_value(sc.ral, col - col0);
          r.dAk[p] = 0.0;  // weff - 0 = weff, exactly
        } else if constexpr (CPWU > 0 && WEFFT >= 0 && UABT >= 0) {
          if constexpr (UABT == 1)
            r.area[p] = lane_value(sc.area0, col - col0);
          else
            r.area[p] = ring[d].area[p];
          if constexpr (WEFFT == 1)
            r.dAk[p] = 0.0;  // weff - 0 = weff, exactly
          else
            r.dAk[p] = ring[d].dAk[p];
        } else {
          r.area[p] = ring[d].area[p];
          r.dAk[p] = ring[d].dAk[p];
        }
        r.rarea[p] = r.rarea_l[p] = 0.;
      }
      // keep D columns' loads in flight.  UNCONDITIONALLY (the wave's last columns re-request its
      // last one, a cache hit): under `if (col + D < cend)` the loaded values had to be MERGED
      // into the ring's registers, and the compiler waited for every prefetch right after
      // issuing it (s_waitcnt vmcnt(0) behind the loads: the ring never overlapped anything)
      if (CPWU > 0) {
        if (prefetch) issue(ring[d], col + D);  // (compile-time after unrolling: no tail re-reads)
      } else {
        issue(ring[d], col + D < cend ? col + D : cend - 1);
      }
      const int k = col - col0;
      const int flags = flags_of(k);
      const bool do_conv = (flags & PM_COL_DO_CONV) != 0;
      const bool use_bzbot = (flags & PM_COL_BZBOT) != 0 && c.bzbot != nullptr;
      const double bs = lane_value(sc.bs, k), bbot = lane_value(sc.bbot, k),
                   N2min = lane_value(sc.N2min, k), bzbot = lane_value(sc.bzbot, k);
      // operands outside the exact-division window: this column steps in the IEEE form
      // (PM_COL_STATIC_IN_RANGE: the caller vouches for the static operands, dt_ok for dt)
      const bool fast_ok = ((flags & PM_COL_STATIC_IN_RANGE) != 0 && dt_ok)
                               ? col_state_in_fast_range<P>(r, wA)
                               : col_inputs_in_fast_range<P>(g, r, wA, dt, bs, bbot, bzbot, N2min,
                                                             lane, nz);
      const bool slow = __ballot(!fast_ok) != 0ull;
      for (int s = 0; s < nsteps; ++s) {
        if (do_conv) col_convect<64, P>(r.b, g.z, bs, N2min, lane, lane, nz, c.z);
        col_pin<P>(r.b);
        if (__builtin_expect(slow, 0))
          col_vertadvdiff<64, P, 0>(g, r, wA, dt, do_conv, bs, bbot, use_bzbot, bzbot, lane, nz);
        else if constexpr (LEAN && D3)
          col_vertadvdiff<64, P, 6, true, false, true, false>(g, r, wA, dt, do_conv, bs, bbot, use_bzbot,
                                                              bzbot, lane, nz);
        else if constexpr (LEAN)  // (Area one number: its reciprocal came with the wave's scalars)
          col_vertadvdiff<64, P, 5, true, false, true>(g, r, wA, dt, do_conv, bs, bbot, use_bzbot,
                                                       bzbot, lane, nz);
        else
          col_vertadvdiff<64, P, 3>(g, r, wA, dt, do_conv, bs, bbot, use_bzbot, bzbot, lane, nz);
      }
      const size_t base = (size_t)col * nz;
      bool bad = false;
      bool stored = false;
      if constexpr (VEC) {
        if (lane * 2 < nz) {
          *reinterpret_cast<double2 *>(c.b + base + lane * 2) = make_double2(r.b[0], r.b[P - 1]);
          bad = !isfinite(r.b[0]) || !isfinite(r.b[P - 1]);
        }
        stored = true;
      } else if constexpr (P == 2) {
        if ((nz & 1) == 0 && (((unsigned long long)(c.b + base)) & 15ull) == 0ull) {
          if (lane * 2 < nz) {
            *reinterpret_cast<double2 *>(c.b + base + lane * 2) = make_double2(r.b[0], r.b[1]);
            bad = !isfinite(r.b[0]) || !isfinite(r.b[1]);
          }
          stored = true;
        }
      }
      if (!stored) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = lane * P + p;
          if (i < nz) {
            c.b[base + i] = r.b[p];
            bad |= !isfinite(r.b[p]);
          }
        }
      }
      if (CPWU > 0) {
        nf_bits |= (__ballot(bad) != 0ull ? 1u : 0u) << kk;
      } else if (c.nonfinite) {
        const unsigned long long m = __ballot(bad);
        if (lane == 0) c.nonfinite[col] = (m != 0ull) ? 1 : 0;
      }
  };
  if constexpr (CPWU > 0) {
#pragma unroll
    for (int k = 0; k < CPWU; ++k) one_column(col0 + k, k % D, k + D < CPWU, k);
    if (c.nonfinite && lane < CPWU) c.nonfinite[col0 + lane] = (int)((nf_bits >> lane) & 1u);
  } else {
    for (int colb = col0; colb < cend; colb += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int col = colb + d;
        if (col >= cend) break;  // wave-uniform
        one_column(col, d, true, 0);
      }
    }
  }
}

// columns per wave of the streaming kernel: enough waves to fill the chip several times over
inline int stream_cols_per_wave(int ncols) {
  static const int forced = []() {
    const char *e = getenv("PYMOC_STREAM_CPW");  // experiments (profiles/): 0 = automatic
    return e ? atoi(e) : 0;
  }();
  if (forced > 0) return forced > STREAM_MAX_CPW ? STREAM_MAX_CPW : forced;
  // few, long waves: two rounds of four resident waves per SIMD; a wave's ramp of D columns is
  // paid once per cpw columns (measured at 262144 columns: 8 / 16 / 32 per wave -> 187 / 189 /
  // 181 us with the forcing precombined)
  const int waves_wanted = 1024 * 8;
  int cpw = ncols / waves_wanted;
  return cpw < 1 ? 1 : (cpw > 32 ? 32 : cpw);
}

// ------------------------------------------------------------------ dispatch
// The <G,P,FAST> instantiations are compiled in three translation units (column_g16.hip,
// column_g32.hip, column_g64.hip) so that the build parallelises.
inline int pick_levels_per_lane(int G, int need) {
  // narrow groups only up to 8 levels per lane; taller columns go to wider groups
  static const int sup64[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 13, 16};
  static const int sup[] = {1, 2, 3, 4, 5, 6, 7, 8};
  if (G == 64) {
    for (int v : sup64)
      if (v >= need) return v;
  } else {
    for (int v : sup)
      if (v >= need) return v;
  }
  return -1;
}

template <int G, int P>
int launch_column_steps(const pm_columns &c, const double *wA, const double *vdx,
                        const double *bin, double dt, int nsteps, int ops,
                        hipStream_t st) {
  const int cols_per_block = 256 / G;
  const unsigned grid = (unsigned)((c.ncols + cols_per_block - 1) / cols_per_block);
  const bool weff_in = (ops & PM_OP_WEFF) != 0;  // wA holds wA - d(A kappa)/dz
  const bool contracted = (ops & PM_OP_CONTRACTED) != 0;  // tolerance mode (one wave per column)
  // the kernel forms wA from Psi_iso / Psi_SO, or from the two-basin driver's overturnings (>= 3 steps)
  const int wa_psi = ops & (PM_OP_WA_PSI | PM_OP_WA_TWOBASIN);
  const double *const forcing2 = (ops & PM_OP_WA_TWOBASIN) ? vdx : nullptr;  // (horadv's slot)
  if (forcing2) vdx = nullptr;  // no horadv with that modifier: the launch conditions below
  ops &= ~(PM_OP_WEFF | PM_OP_CONTRACTED | PM_OP_WA_PSI | PM_OP_WA_TWOBASIN);
  const double *const vk = forcing2 ? forcing2 : vdx;  // what the PLAIN kernels get in that slot
  if constexpr (G == 64 && P <= 4) {
    const int cpw = stream_cols_per_wave(c.ncols);
    if (nsteps < 3 && ops == PM_OP_TIMESTEP && !vdx && cpw >= 2) {
      const unsigned waves = (unsigned)((c.ncols + cpw - 1) / cpw);
      // dt inside the exact-division window (in_fast_div_range, host side)
      const double adt = dt < 0 ? -dt : dt;
      const bool dt_ok = dt == 0.0 || (adt >= 0x1p-200 && adt <= 0x1p200);
      // stages of the load ring: three arrays per column with the forcing precombined, five
      // without (the ring's registers bound the occupancy)
      const bool aff = c.kappa_base && c.kappa_profile && c.nsel == 1;
      // PM_COL_BATCH_UNIFORM_AREA in reserved: the caller vouches that EVERY column carries
      // PM_COL_UNIFORM_AREA (the kernel cannot branch per column on what its ring holds)
      const bool lean = weff_in && aff && (c.reserved & PM_COLS_ALL_UNIFORM_AREA) != 0;
      static const int lean_d = []() {
        const char *e = getenv("PYMOC_STREAM_LEAN_D");  // experiments: ring depth of the lean form
        return e ? atoi(e) : 0;
      }();
      if (lean && lean_d == 4)
        hipLaunchKernelGGL((k_column_stream<P, 4, true, true>), dim3((waves + 3) / 4), dim3(256), 0,
                           st, c, wA, dt, nsteps, cpw, dt_ok, weff_in);
      else if (lean && lean_d == 5)
        hipLaunchKernelGGL((k_column_stream<P, 5, true, true>), dim3((waves + 3) / 4), dim3(256), 0,
                           st, c, wA, dt, nsteps, cpw, dt_ok, weff_in);
      else if (lean && lean_d == 8)
        hipLaunchKernelGGL((k_column_stream<P, 8, true, true>), dim3((waves + 3) / 4), dim3(256), 0,
                           st, c, wA, dt, nsteps, cpw, dt_ok, weff_in);
      else if (lean && lean_d == 6)
        hipLaunchKernelGGL((k_column_stream<P, 6, true, true>), dim3((waves + 3) / 4), dim3(256), 0,
                           st, c, wA, dt, nsteps, cpw, dt_ok, weff_in);
      else if (lean) {
        // measured at 262144 columns x nz = 100 (profiles/r04/probe_stream_lean.py): ring depth
        // 4 / 5 / 6 / 8 -> 150.5 / 148.8 / 158.6 / 154.8 us at their best columns-per-wave; 16
        // columns per wave beat 32 (the non-lean forms' choice) and 8
        int cl = cpw;
        if (!getenv("PYMOC_STREAM_CPW")) {
          cl = c.ncols / 16384;
          cl = cl < 2 ? 2 : (cl > 16 ? 16 : cl);
        }
        unsigned wl = (unsigned)((c.ncols + cl - 1) / cl);
        bool vec = false;
        if constexpr (P == 2)
          vec = (c.nz & 1) == 0 && ((((unsigned long long)c.b) | ((unsigned long long)wA)) & 15ull) == 0ull &&
                !getenv("PYMOC_STREAM_NO_VEC");
        const bool d3 = (c.reserved & PM_COLS_DIV3_PROVEN) != 0;
        // every wave exactly 8 columns: the straight-line instantiation (CPWU).  Measured at 262144
        // columns (profiles/r05/stream_variants.log): ring depth 3 / 4 / 5 / 6 at 16 columns per wave
        // 132 / 133 / 133 / 134 us -- with a ring that works its depth no longer matters -- and depth
        // 5 at 8 / 16 / 32 columns per wave 127 / 133 / 268 us
        const bool u16 = vec && c.ncols % 8 == 0 && c.ncols >= 8 * 4096 &&
                         !getenv("PYMOC_STREAM_LOOP") && !getenv("PYMOC_STREAM_CPW");
        if (u16) {
          cl = 8;
          wl = (unsigned)(c.ncols / 8);
        }
#ifdef PM_STREAM_VARIANTS  // experiments (profiles/r05/stream_variants.sh): ring depth x columns per wave
        if constexpr (P == 2) {
          const char *ev = getenv("PYMOC_STREAM_VARIANT");
          if (ev && vec && d3) {
            const int dv = atoi(ev), cv = strchr(ev, ',') ? atoi(strchr(ev, ',') + 1) : 16;
            const unsigned wv = (unsigned)((c.ncols + cv - 1) / cv);
#define PM_SV(DD, CC)                                                                              \
  if (dv == DD && cv == CC && c.ncols % CC == 0) {                                                 \
    hipLaunchKernelGGL((k_column_stream<P, DD, true, true, true, true, CC>), dim3((wv + 3) / 4),   \
                       dim3(256), 0, st, c, wA, dt, nsteps, cv, dt_ok, weff_in);                   \
    PM_HIP(hipGetLastError());                                                                     \
    return PM_OK;                                                                                  \
  }
            PM_SV(4, 16) PM_SV(6, 16) PM_SV(5, 32) PM_SV(5, 8) PM_SV(3, 16)
#undef PM_SV
          }
        }
#endif
        if constexpr (P == 2) {
          if (vec && u16 && d3)
            hipLaunchKernelGGL((k_column_stream<P, 5, true, true, true, true, 8>), dim3((wl + 3) / 4),
                               dim3(256), 0, st, c, wA, dt, nsteps, cl, dt_ok, weff_in);
          else if (vec && u16)
            hipLaunchKernelGGL((k_column_stream<P, 5, true, true, true, false, 8>), dim3((wl + 3) / 4),
                               dim3(256), 0, st, c, wA, dt, nsteps, cl, dt_ok, weff_in);
          else if (vec && d3)
            hipLaunchKernelGGL((k_column_stream<P, 5, true, true, true, true>), dim3((wl + 3) / 4),
                               dim3(256), 0, st, c, wA, dt, nsteps, cl, dt_ok, weff_in);
          else if (vec)
            hipLaunchKernelGGL((k_column_stream<P, 5, true, true, true>), dim3((wl + 3) / 4), dim3(256),
                               0, st, c, wA, dt, nsteps, cl, dt_ok, weff_in);
        }
        if (!vec)
          hipLaunchKernelGGL((k_column_stream<P, 5, true, true>), dim3((wl + 3) / 4), dim3(256), 0,
                             st, c, wA, dt, nsteps, cl, dt_ok, weff_in);
      }
      else if (P == 2 && !aff && c.ncols % 8 == 0 && c.ncols >= 8 * 4096 && !getenv("PYMOC_STREAM_LOOP") &&
               !getenv("PYMOC_STREAM_CPW") &&
               ((weff_in && (c.reserved & PM_COLS_ALL_UNIFORM_AREA)) ||
                (!weff_in && !(c.reserved & PM_COLS_ALL_UNIFORM_AREA)))) {
        // the straight-line forms of the two other regimes the bench reports: forcing precombined
        // and every Area one number (32 nz B per column-step), and the C-ABI default (48 nz B)
        const unsigned w8 = (unsigned)(c.ncols / 8);
        if constexpr (P == 2) {
          if (weff_in)
            hipLaunchKernelGGL((k_column_stream<P, 3, false, false, false, false, 8, 1, 1>),
                               dim3((w8 + 3) / 4), dim3(256), 0, st, c, wA, dt, nsteps, 8, dt_ok, weff_in);
          else
            hipLaunchKernelGGL((k_column_stream<P, 2, false, false, false, false, 8, 0, 0>),
                               dim3((w8 + 3) / 4), dim3(256), 0, st, c, wA, dt, nsteps, 8, dt_ok, weff_in);
        }
      }
      else if (weff_in && aff)
        hipLaunchKernelGGL((k_column_stream<P, 4, true>), dim3((waves + 3) / 4), dim3(256), 0, st, c,
                           wA, dt, nsteps, cpw, dt_ok, weff_in);
      else if (weff_in)
        hipLaunchKernelGGL((k_column_stream<P, 3>), dim3((waves + 3) / 4), dim3(256), 0, st, c, wA,
                           dt, nsteps, cpw, dt_ok, weff_in);
      else if (aff)
        hipLaunchKernelGGL((k_column_stream<P, 3, true>), dim3((waves + 3) / 4), dim3(256), 0, st, c,
                           wA, dt, nsteps, cpw, dt_ok, weff_in);
      else
        hipLaunchKernelGGL((k_column_stream<P, 2>), dim3((waves + 3) / 4), dim3(256), 0, st, c, wA,
                           dt, nsteps, cpw, dt_ok, weff_in);
      PM_HIP(hipGetLastError());
      return PM_OK;
    }
  }
  if constexpr (G == 64 && P <= 4) {
    if (contracted && nsteps >= 3 && ops == PM_OP_TIMESTEP && !vdx) {
      hipLaunchKernelGGL((k_column_steps<G, P, 4, true>), dim3(grid), dim3(256), 0, st, c, wA, vk,
                         bin, dt, nsteps, ops | (weff_in ? PM_OP_WEFF : 0) | wa_psi);
      PM_HIP(hipGetLastError());
      return PM_OK;
    }
  }
  // the reciprocal path pays 3 true divisions per level up front: worth it from 3 steps on
  bool launched = false;
  if constexpr (G == 64 && P <= 4) {
    if (nsteps >= 3 && ops == PM_OP_TIMESTEP && !vdx && (c.reserved & PM_COLS_ALL_UNIFORM_AREA)) {
      static const int lds_pad = []() {
        const char *e = getenv("PYMOC_K1_LDS");  // experiments: unused LDS per block caps the occupancy
        return e ? atoi(e) : 0;
      }();
      if ((c.reserved & PM_COLS_DIV3_PROVEN) != 0)  // (the caller's pm_div3_proven verdict)
        hipLaunchKernelGGL((k_column_steps<G, P, 6, true, true>), dim3(grid), dim3(256), lds_pad, st, c, wA,
                           vk, bin, dt, nsteps, ops | (weff_in ? PM_OP_WEFF : 0) | wa_psi);
      else
        hipLaunchKernelGGL((k_column_steps<G, P, 2, true, true>), dim3(grid), dim3(256), lds_pad, st, c, wA,
                           vk, bin, dt, nsteps, ops | (weff_in ? PM_OP_WEFF : 0) | wa_psi);
      launched = true;
    }
  }
  if (launched) {
  } else if (nsteps >= 3 && ops == PM_OP_TIMESTEP && !vdx)
    hipLaunchKernelGGL((k_column_steps<G, P, 2, true>), dim3(grid), dim3(256), 0, st, c, wA,
                       vk, bin, dt, nsteps, ops | (weff_in ? PM_OP_WEFF : 0) | wa_psi);
  else if (nsteps >= 3)
    hipLaunchKernelGGL((k_column_steps<G, P, 1, false>), dim3(grid), dim3(256), 0, st, c, wA,
                       vdx, bin, dt, nsteps, ops | (weff_in ? PM_OP_WEFF : 0));
  else
    hipLaunchKernelGGL((k_column_steps<G, P, 0, false>), dim3(grid), dim3(256), 0, st, c,
                       wA, vdx, bin, dt, nsteps, ops | (weff_in ? PM_OP_WEFF : 0));
  PM_HIP(hipGetLastError());
  return PM_OK;
}

int column_steps_g16(int P, const pm_columns &c, const double *wA, const double *vdx,
                     const double *bin, double dt, int nsteps, int ops, hipStream_t st);
int column_steps_g32(int P, const pm_columns &c, const double *wA, const double *vdx,
                     const double *bin, double dt, int nsteps, int ops, hipStream_t st);
int column_steps_g64(int P, const pm_columns &c, const double *wA, const double *vdx,
                     const double *bin, double dt, int nsteps, int ops, hipStream_t st);

inline int auto_lanes_per_col(int ncols, int nz) {
  // Few columns: one wave per column keeps every SIMD busy (1024 SIMDs on the chip).
  // Many columns: narrower groups raise per-lane ILP and cut idle padding lanes.
  // measured on MI355X (profiles/r01_sweep_columns.txt): one wave per column wins at every
  // ensemble size from 1024 to 65536 columns at nz=100
  (void)ncols;
  int G = 64;
  while (G < 64 && pick_levels_per_lane(G, (nz + G - 1) / G) < 0) G *= 2;
  return G;
}

}  // namespace pm
