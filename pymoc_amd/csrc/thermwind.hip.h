// thermwind.hip.h -- K2/K3: thermal-wind overturning and its isopycnal remap.
//
// Arithmetic restated from the reference (nothing copied):
//   Psi_Thermwind.solve  src/pymoc/modules/psi_thermwind.py:125-135 (+ode :95-123, bc :72-93)
//   Psi_Thermwind.Psib   src/pymoc/modules/psi_thermwind.py:137-185
//   Psi_Thermwind.Psibz  src/pymoc/modules/psi_thermwind.py:187-208
//
// One wavefront per ensemble member; lane l owns the P contiguous levels [l*P, l*P+P).
//  * solve: the reference hands  Psi'' = (b2-b1)/f, Psi(z0)=Psi(zn)=0  to SciPy's
//    4th-order collocation solver.  For np.interp profiles the right-hand side is
//    piecewise linear and the collocation equations reduce exactly to two running
//    Simpson integrals.  Increments are formed in parallel; the two
//    prefix sums are taken in level order (through LDS, every lane redundantly) so the
//    result is bit-identical to the oracle's sequential loop.
//  * Psib: lanes own isopycnal classes, the nz-1 cells are broadcast from LDS; the
//    per-class sum runs in NumPy's pairwise order (8 accumulators, tree, tail), so
//    psib is bit-identical to np.sum in the reference, NaN/inf cases included.
//  * Psibz: np.interp on the uniform bgrid -- direct index + fix-up instead of a search.
#pragma once
#include <type_traits>
#include "common.hip.h"

namespace pm {

constexpr int TW_WAVES_PER_BLOCK = 4;
#ifndef TW_JT_N
#define TW_JT_N 2
#endif
constexpr int TW_JT = TW_JT_N;  // isopycnal classes per lane per pass

__device__ __forceinline__ double np_clip01(double v) {
  // np.clip(v, 0, 1) = minimum(maximum(v, 0), 1); both propagate NaN
  if (v != v) return v;
  v = v < 0. ? 0. : v;
  return v > 1. ? 1. : v;
}

// np.interp(x, bgrid, psib) for one query, bgrid = lin (ascending, uniform), psib in LDS.
// The interval index starts from a guess, (x - start) * rstep with rstep ~ 1 / step, and is
// then moved until bgrid[j] <= x < bgrid[j+1] holds with np.linspace's own node values, so the
// guess only has to be close (a correctly rounded division here cost a sixth of Psibz).
// the interval of x on the uniform grid: largest j with bgrid[j] <= x, and the grid's own values
// at j and j + 1 (callers have dealt with NaN, a single node and x outside [start, stop]); -1:
// no such node
__device__ __forceinline__ int interp_uniform_index(double x, const Linspace &lin, int nb,
                                                    double rstep, double &xj, double &xj1) {
  int j;
  if (lin.step > 0. && lin.step < 1e300) {
    // (nb > 1 and step != 0 here: Linspace::at without its special cases)
    auto at = [&](int i) { return i == nb - 1 ? lin.stop : (double)i * lin.step + lin.start; };
    const double q = (x - lin.start) * rstep;
    j = (int)q;
    j = j < 0 ? 0 : (j > nb - 1 ? nb - 1 : j);
    xj = at(j);
    xj1 = at(j + 1 < nb ? j + 1 : nb - 1);
    while (j > 0 && x < xj) {
      --j;
      xj1 = xj;
      xj = at(j);
    }
    while (j < nb - 1 && x >= xj1) {
      ++j;
      xj = xj1;
      xj1 = at(j + 1 < nb ? j + 1 : nb - 1);
    }
  } else {
    int lo = 0, hi = nb;  // upper bound
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (x >= lin.at(mid))
        lo = mid + 1;
      else
        hi = mid;
    }
    j = lo - 1;
    if (j < 0) return j;
    xj = lin.at(j);
    xj1 = lin.at(j + 1 < nb ? j + 1 : nb - 1);
  }
  return j;
}

// np.interp's value from the interval j = interp_uniform_index(x, ...) >= 0 (x inside the grid)
__device__ __forceinline__ double interp_uniform_at(double x, int j, const Linspace &lin,
                                                    const double *psib, int nb) {
  if (j < 0) return x;  // (a NaN level: the marking pass left it out)
  if (j == nb - 1) return psib[j];
  const double xj = lin.at(j), xj1 = lin.at(j + 1);
  if (xj == x) return psib[j];
  const double fj = psib[j], fj1 = psib[j + 1];
  const double slope = (fj1 - fj) / (xj1 - xj);
  double r = slope * (x - xj) + fj;
  if (r != r) {  // numpy: nan in one direction, try the other
    r = slope * (x - xj1) + fj1;
    if (r != r && fj == fj1) r = fj;
  }
  return r;
}

__device__ __forceinline__ double interp_uniform(double x, const Linspace &lin,
                                                 const double *psib, int nb, double rstep) {
  if (x != x) return x;
  const double lval = psib[0], rval = psib[nb - 1];
  if (nb == 1) return (x < lin.start) ? lval : ((x > lin.start) ? rval : psib[0]);
  if (x > lin.stop) return rval;
  if (x < lin.start) return lval;
  double xj, xj1;
  const int j = interp_uniform_index(x, lin, nb, rstep, xj, xj1);
  if (j < 0) return lval;
  if (j == nb - 1) return psib[j];
  if (xj == x) return psib[j];
  const double fj = psib[j], fj1 = psib[j + 1];
  const double slope = (fj1 - fj) / (xj1 - xj);
  double r = slope * (x - xj) + fj;
  if (r != r) {  // numpy: nan in one direction, try the other
    r = slope * (x - xj1) + fj1;
    if (r != r && fj == fj1) r = fj;
  }
  return r;
}

// Cells of the Psib sum in LDS.  Two layouts:
//   W = 6: 6 doubles per cell, top, d = top - bot | yh = RN(1/d) or NaN, yl = recip_lo(d, yh) | u, bot
//          (three 16-byte broadcast loads; the sixth slot feeds the group ranges of lane shapes
//          whose groups do not coincide with lanes);
//   W = 5 (P = 1, 2, 4, 8: the group ranges come from the lanes' registers): {top, d | yh, yl}[nz]
//          followed by u[nz] -- two 16-byte loads and one 8-byte load per cell, the all-ones groups
//          read u alone.  A sixth less LDS per wave (16 members of nz = 200 then fit a CU together
//          with the fused step loop's tables: coupled_run.hip.h) and a sixth less LDS traffic in
//          the class passes.
// Measured (profiles/r04/run_ab_w5.sh): W = 5 makes k_thermwind<4,1> SLOWER, 77.5 against 69.6 us
// per update of config 5 (the third, narrower load costs the tile loop more than the bytes it
// saves), and changes nothing at P = 2.  The stand-alone kernel keeps W = 6; the persistent run
// kernel uses W = 5 at P = 4, where nothing else fits the LDS.
__host__ __device__ constexpr bool tw_w5_ok(int P) { return P == 1 || P == 2 || P == 4 || P == 8; }
struct PsibCell {
  double top, d, yh, yl, u;
};
template <int W>
struct CellView {
  const double *base;
  int nz;
  __device__ __forceinline__ PsibCell load(int k) const {
    if constexpr (W == 6) {
      const double2 *c = reinterpret_cast<const double2 *>(base + (size_t)k * 6);
      const double2 v0 = c[0], v1 = c[1], v2 = c[2];
      return PsibCell{v0.x, v0.y, v1.x, v1.y, v2.x};
    } else {
      const double2 *c = reinterpret_cast<const double2 *>(base + (size_t)k * 4);
      const double2 v0 = c[0], v1 = c[1];
      return PsibCell{v0.x, v0.y, v1.x, v1.y, base[4 * nz + k]};
    }
  }
  __device__ __forceinline__ double u(int k) const {
    return W == 6 ? base[(size_t)k * 6 + 4] : base[4 * nz + k];
  }
};
// The psib row (nb doubles, read by Psibz) overlays the cells when the class passes' results
// fit in registers (<= TW_HELD passes): they are written once the last pass has read the cells.
// Only where it buys residency: nz > 128 (smaller grids keep 16 waves on a CU anyway).
constexpr int TW_HELD = 8 / TW_JT_N;  // (TW_JT_N = 1, 64-class passes: measured no faster)
__host__ __device__ inline bool tw_overlay(int nz, int nb, int W = 6) {
  return nz > 128 && nb <= 64 * TW_JT * TW_HELD && nb <= W * nz;
}
// LDS doubles per wave: cells, the psib row unless overlaid, two group-range rows; even, so
// that every wave's cells stay 16-byte aligned
// ... the group-range rows double as the bit marks (16 words) and the list (TW_LAZY_CAP class
// numbers, 16 bits each) of the classes Psibz asks for (chain-order members, W = 6)
constexpr int TW_LAZY_CAP = 256, TW_LAZY_DOUBLES = (64 + 2 * TW_LAZY_CAP) / 8;
// (chain = false: the tiles only -- the persistent run kernels, whose phases share the LDS)
__host__ __device__ inline int tw_pad_cells(int nz, int W);
__host__ __device__ inline int tw_lds_doubles(int nz, int nb, int W = 6, bool chain = true) {
  int rows = 2 * ((nz + 7) / 8);
  if (chain && W == 6 && rows < TW_LAZY_DOUBLES) rows = TW_LAZY_DOUBLES;
  return (W * (nz + (chain ? tw_pad_cells(nz, W) : 0)) + (tw_overlay(nz, nb, W) ? 0 : nb) + rows + 1) & ~1;
}
// RN(fma(r, yh, q)) clamped to [0, 1] by the VOP3 clamp modifier (the last Markstein step and
// np.clip in one instruction; finite operands only)
__device__ __forceinline__ double fma_clamp01(double r, double yh, double q) {
  double o;
  asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(o) : "v"(r), "v"(yh), "v"(q));
  return o;
}

// mask_k * u_k of one REGULAR cell for TW_JT classes: clip((top - bg)/(top - bot), 0, 1) * u
// (psi_thermwind.py:183), the quotient correctly rounded in 4 instructions from the
// double-double reciprocal (div_by_recip2), the clip folded into the last of them.
__device__ __forceinline__ void psib_regular_terms(const PsibCell &c, const double (&bg)[TW_JT],
                                                   double (&out)[TW_JT]) {
  double tt[TW_JT], q[TW_JT], rr[TW_JT];
#pragma unroll
  for (int j = 0; j < TW_JT; ++j) tt[j] = c.top - bg[j];
#pragma unroll
  for (int j = 0; j < TW_JT; ++j) rr[j] = tt[j] * c.yl;
#pragma unroll
  for (int j = 0; j < TW_JT; ++j) q[j] = __builtin_fma(tt[j], c.yh, rr[j]);
#pragma unroll
  for (int j = 0; j < TW_JT; ++j) rr[j] = __builtin_fma(-c.d, q[j], tt[j]);
#pragma unroll
  for (int j = 0; j < TW_JT; ++j) out[j] = fma_clamp01(rr[j], c.yh, q[j]) * c.u;
}

// Any cell: regular ones (finite, thickness in [1e-100, 1e100]; yh is NaN otherwise) as above;
// degenerate cells (zero thickness -> +-inf / NaN, hazard H6) take IEEE division and NumPy's
// NaN-propagating clip.  The branch is wave-uniform (cell data is broadcast from LDS).
__device__ __forceinline__ void psib_cell_terms(const PsibCell &c, const double (&bg)[TW_JT],
                                                double (&out)[TW_JT]) {
  if (c.yh == c.yh) {
    psib_regular_terms(c, bg, out);
  } else {
#pragma unroll
    for (int j = 0; j < TW_JT; ++j) out[j] = np_clip01((c.top - bg[j]) / c.d) * c.u;
  }
}

// One isopycnal class: sum_k clip((top_k - bg)/(top_k - bot_k), 0, 1) * u_k over cells
// [k0, k0+n) in NumPy's pairwise order, for TW_JT classes at once.
// Range test of a pass: every class of the pass lies in [gmin, gmax] (wave-uniform scalars);
// gbot[g] / gtop[g] hold min(bot) / max(top) of the 8 cells k = 8g .. 8g+7, or -inf / +inf when
// a cell of the group is degenerate or inverted or carries a non-finite u_k ("irregular
// group").  If gmax <= gbot[g] every mask of the group is exactly 1 for every class of the
// pass (fl(top-g) >= fl(top-bot) > 0, quotient >= 1), so the terms are the u_k themselves; if
// gmin >= gtop[g] every mask is exactly 0 and the group only adds zeros.  Both shortcuts leave
// every partial sum of NumPy's pairwise order bit-identical (x + 0 = x).  A regular group
// that straddles the pass's range runs its 8 cells without any per-cell branch; an irregular
// group tests every cell.
struct PsibRange {
  const double *gbot, *gtop;
  double gmin, gmax;
};

__device__ __forceinline__ double tw_uniform(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}

// terms of the 8 cells of group k8/8 into r (ADD: accumulate, else initialise).
// kind: 0 general and regular, 1 general and irregular, 2 all masks one, 3 all masks zero
template <bool ADD, class CV>
__device__ __forceinline__ void psib_group(const CV &cells, int k8, int kind,
                                           const double (&bg)[TW_JT],
                                           double (&r)[8][TW_JT] PM_TICK_PARAM) {
  double term[TW_JT];
  if (kind == 2) {
    PM_COUNT(12)
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const double uk = cells.u(k8 + a);
#pragma unroll
      for (int j = 0; j < TW_JT; ++j) r[a][j] = ADD ? r[a][j] + uk : uk;
    }
  } else if (kind == 3) {
    PM_COUNT(13)
    if (!ADD) {  // all masks 0: the group adds zeros
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int j = 0; j < TW_JT; ++j) r[a][j] = 0.;
    }
  } else if (kind == 0) {
    PM_COUNT(11)
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      psib_regular_terms(cells.load(k8 + a), bg, term);
#pragma unroll
      for (int j = 0; j < TW_JT; ++j) r[a][j] = ADD ? r[a][j] + term[j] : term[j];
    }
  } else {
    PM_COUNT(14)
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      psib_cell_terms(cells.load(k8 + a), bg, term);
#pragma unroll
      for (int j = 0; j < TW_JT; ++j) r[a][j] = ADD ? r[a][j] + term[j] : term[j];
    }
  }
}

// n <= 128 cells from k0 (a multiple of 8): NumPy's 8-accumulator block.  The groups of the
// block are classified at once -- lane l compares the range of group k0/8 + l with the pass's
// class range, three ballots give the masks -- so the loop below visits only the groups that
// add something, with scalar bit tests instead of an LDS round trip and two compares per group.
template <class CV>
__device__ __forceinline__ void psib_block_sum(const CV &cells, int k0, int n,
                                               const double (&bg)[TW_JT],
                                               double (&res)[TW_JT],
                                               const PsibRange &rg PM_TICK_PARAM) {
  double term[TW_JT];
  if (n < 8) {
#pragma unroll
    for (int j = 0; j < TW_JT; ++j) res[j] = 0.;
    for (int k = k0; k < k0 + n; ++k) {
      psib_cell_terms(cells.load(k), bg, term);
#pragma unroll
      for (int j = 0; j < TW_JT; ++j) res[j] += term[j];
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int ng = n >> 3, g0 = k0 >> 3;  // ng <= 16
  // (the cells after the last full group, n % 8 of them, are the array's last, partial group:
  // bit ng classifies them too)
  const bool mine = lane < ng + ((n & 7) ? 1 : 0);
  const double vgb = rg.gbot[g0 + (mine ? lane : 0)], vgt = rg.gtop[g0 + (mine ? lane : 0)];
  const unsigned long long ones = __ballot(mine && rg.gmax <= vgb);
  const unsigned long long zero = __ballot(mine && rg.gmin >= vgt) & ~ones;
  const unsigned long long irr = __ballot(mine && vgb == -__builtin_inf());
  auto kind_of = [&](int g) -> int {
    return ((ones >> g) & 1ull) ? 2 : (((zero >> g) & 1ull) ? 3 : (((irr >> g) & 1ull) ? 1 : 0));
  };
  double r[8][TW_JT];
  psib_group<false, CV>(cells, k0, kind_of(0), bg, r PM_TICK_ARG);  // initialises the accumulators
  unsigned long long todo = ~zero & ((1ull << ng) - 1ull) & ~1ull;
  while (todo != 0ull) {
    const int g = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    psib_group<true, CV>(cells, k0 + 8 * g, kind_of(g), bg, r PM_TICK_ARG);
  }
#pragma unroll
  for (int j = 0; j < TW_JT; ++j)
    res[j] = ((r[0][j] + r[1][j]) + (r[2][j] + r[3][j])) +
             ((r[4][j] + r[5][j]) + (r[6][j] + r[7][j]));
  if ((n & 7) != 0) {  // NumPy's tail: the remaining cells are added one by one
    const int kt = kind_of(ng);
    if (kt == 2) {
      for (int k = (ng << 3); k < n; ++k) {
        const double uk = cells.u(k0 + k);
#pragma unroll
        for (int j = 0; j < TW_JT; ++j) res[j] += uk;
      }
    } else if (kt != 3) {  // (all masks zero: the tail adds zeros)
      for (int k = (ng << 3); k < n; ++k) {
        psib_cell_terms(cells.load(k0 + k), bg, term);
#pragma unroll
        for (int j = 0; j < TW_JT; ++j) res[j] += term[j];
      }
    }
  }
}

// out-of-line copy for the recursive (nz > 129) path: 16 inlined copies per kernel cost
// minutes of compile time and every register
template <class CV>
__device__ __noinline__ void psib_block_sum_call(const CV &cells, int k0, int n,
                                                 const double (&bg)[TW_JT],
                                                 double (&res)[TW_JT],
                                                 const PsibRange &rg PM_TICK_PARAM) {
  psib_block_sum(cells, k0, n, bg, res, rg PM_TICK_ARG);
}

// np.add.reduce pairwise recursion (blocks of <= 128, left half rounded down to a
// multiple of 8).  D bounds the recursion depth: D=4 covers n <= 128*16.
template <int D, class CV>
__device__ __forceinline__ void psib_pairwise(const CV &cells, int k0, int n,
                                              const double (&bg)[TW_JT], double (&res)[TW_JT],
                                              const PsibRange &rg PM_TICK_PARAM) {
  if constexpr (D == 0) {
    psib_block_sum_call(cells, k0, n, bg, res, rg PM_TICK_ARG);
  } else {
    if (n <= 128) {
      psib_block_sum_call(cells, k0, n, bg, res, rg PM_TICK_ARG);
      return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    double l[TW_JT], r[TW_JT];
    psib_pairwise<D - 1, CV>(cells, k0, n2, bg, l, rg PM_TICK_ARG);
    psib_pairwise<D - 1, CV>(cells, k0 + n2, n - n2, bg, r, rg PM_TICK_ARG);
#pragma unroll
    for (int j = 0; j < TW_JT; ++j) res[j] = l[j] + r[j];
  }
}

// ------------------------------------------------------------------------------------------
// Round 5: the class sums of a member whose cells lie in CHAIN ORDER, class by class.
//
// The tiles above evaluate every (cell, class) pair of a group that the pass's class range
// cuts: 8 x 128 quotients where a class value g cuts ONE cell.  Along BASELINE's runs the
// upstream cells (bot_k, top_k) of psi_thermwind.py:175-181 are in chain order -- top_k <=
// bot_{k+1}, every thickness > 0 -- in every update of configs 3 and 6, 95 % of config 4's and,
// from cell K0 <= 2 on (the bottom cells of a no-flux bottom boundary are degenerate or
// inverted), in every update of config 5 (profiles/r05/probe_psib_structure.py).  For such a
// member a class value g has
//   mask exactly 0 on the cells below the cut cell kz = first k >= K0 with top_k > g
//       (top_k <= g: fl(top - g) <= 0, the quotient is clipped to 0),
//   the quotient of psi_thermwind.py:183 on cell kz itself (1 exactly when g <= bot_kz),
//   mask exactly 1 on every cell above it (g < top_kz <= bot_k: the tiles' all-ones argument),
// so NumPy's pairwise sum of mask * udydz -- 8 accumulators over the cells k = a (mod 8) of a
// block of <= 128 cells, a tree, the tail one by one -- only ever holds, per accumulator,
//   S[k_a] = ((u[k_a] + u[k_a + 8]) + u[k_a + 16]) + ...   (k_a: its first cell above the cut;
//            zeros added before the first non-zero term change nothing), a table per MEMBER, or
//   ((term(kz) + u[kz + 8]) + u[kz + 16]) + ...             on the cut cell's own accumulator.
// Per class: a search for kz, ONE quotient, a chain of <= 11 additions, 8 table reads, the tree
// and the tail -- instead of 7 instructions for each of ~1500 (cell, class) pairs of a first
// pass.  Every partial sum is the one NumPy forms (x + 0 = x), so psib stays bit-identical.
// Cells below K0 <= 2 are taken as they come: a pass whose classes can see them evaluates their
// accumulators cell by cell with the tiles' own term (psib_cell_terms).  Any other member (a
// cell out of order further up, non-finite values, nz - 1 > 256) keeps the tiles.
//
// Layout: S[k] takes the sixth slot of cell k (the tiles' group ranges have been formed by
// then); cell nc = nz - 1, which no level owns, is the SENTINEL {top = +inf, u = -0.0, S = 0}: a
// search may probe it, a chain may add it (x + -0.0 = x for every x), and the cells from the
// last block's first tail cell on (tw_pad_cells() more than the array had) form a row of zero S
// for accumulators that have no cell left.
__host__ __device__ inline int tw_sorted_blocks(int nc) { return nc <= 128 ? 1 : (nc <= 256 ? 2 : 0); }
__host__ __device__ inline int tw_block_split(int nc) {  // NumPy's split of 128 < nc <= 256 cells
  int n2 = nc / 2;
  return n2 - n2 % 8;
}
// first cell of the row of zero S (the last block's first tail cell)
__host__ __device__ inline int tw_zero_row(int nc) {
  const int k0 = tw_sorted_blocks(nc) == 2 ? tw_block_split(nc) : 0;
  return k0 + (((nc - k0) >> 3) << 3);
}
// cells the array needs beyond its nz so that the zero row is 8 cells long
__host__ __device__ inline int tw_pad_cells(int nz, int W) {
  const int nc = nz - 1;
  if (W != 6 || nc < 16 || tw_sorted_blocks(nc) == 0) return 0;
  const int over = tw_zero_row(nc) + 8 - nz;
  return over > 0 ? over : 0;
}

// term of one class on one cell of the chain: psib_regular_terms for one (cell, class) pair.
// DEG (wave-uniform: the member has cells of zero thickness in its chain, yh = NaN there):
// clip((top - g) / 0, 0, 1) is 1 below the cell's buoyancy, 0 above it and NaN on it (H6).
__device__ __forceinline__ double psib_deg_term(const PsibCell &c, double tt, double r) {
  const double m = tt > 0. ? 1. : (tt < 0. ? 0. : __builtin_nan(""));
  return c.d == 0. ? m * c.u : r;
}
__device__ __forceinline__ double psib_term1(const PsibCell &c, double g, bool deg) {
  const double tt = c.top - g;
  double rr = tt * c.yl;
  const double q = __builtin_fma(tt, c.yh, rr);
  rr = __builtin_fma(-c.d, q, tt);
  const double r = fma_clamp01(rr, c.yh, q) * c.u;
  return deg ? psib_deg_term(c, tt, r) : r;
}
// ... and the same quotient from RN(1/d) alone (div_by_recip: two Markstein steps, 5
// instructions), for cells whose `yl` slot holds a table entry (second block of nz - 1 > 128)
__device__ __forceinline__ double psib_term1_y(const PsibCell &c, double g, bool deg) {
  const double tt = c.top - g;
  double q = tt * c.yh;
  double rr = __builtin_fma(-c.d, q, tt);
  q = __builtin_fma(rr, c.yh, q);
  rr = __builtin_fma(-c.d, q, tt);
  const double r = fma_clamp01(rr, c.yh, q) * c.u;
  return deg ? psib_deg_term(c, tt, r) : r;
}

// term of one class on ANY cell (psib_cell_terms for one pair): the tiles' quotient on a regular
// cell, IEEE division and NumPy's NaN-propagating clip on a degenerate or inverted one
__device__ __forceinline__ double psib_cell_term1(const PsibCell &c, double g) {
  if (c.yh == c.yh) {
    const double tt = c.top - g;
    double rr = tt * c.yl;
    const double q = __builtin_fma(tt, c.yh, rr);
    rr = __builtin_fma(-c.d, q, tt);
    return fma_clamp01(rr, c.yh, q) * c.u;
  }
  return np_clip01((c.top - g) / c.d) * c.u;
}

// One pass (JT classes per lane, ascending with the lane, then with the slot) over a member in
// chain order.
// cell: the W = 6 cells; klo: a cell index no class of the pass cuts below (K0, or the cut of
// the previous pass's last class); exc: bit e set = cell e < K0 may carry a non-zero term for a
// class of this pass; kz (out): the cut cells.
// S of block b's cells lives in slot 5 (b = 0) or, for the second block of 128 < nc <= 256
// cells, in slot 3 (`yl`; slot 5 of those cells is zero, so that the first block's reads past
// its end find zeros).
template <int NBLK, int JT>
__device__ __forceinline__ void psib_sorted_pass(const double *cell, int nc, int klo, int K0,
                                                 unsigned exc, bool deg,
                                                 const double (&bg)[JT], double (&res)[JT],
                                                 int (&kz)[JT]) {
  const CellView<6> cv{cell, nc + 1};
  const int SENT = nc, ZROW = tw_zero_row(nc);
  {  // the cut cell: 1 + the last k in [klo - 1, nc) with top_k <= g, by binary lifting (the
     // tops ascend; the sentinel's top is +inf).  Positions are byte offsets of cells.
    int pos[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) pos[j] = (klo - 1) * 48;
    const int R = nc - klo, lim = nc * 48;
    for (int s = R > 0 ? (48 << (31 - __builtin_clz(R))) : 0; s >= 48; s >>= 1) {
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int cand = pos[j] + s;
        const int c = cand < lim ? cand : lim;
        const double t = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(cell) + c);
        pos[j] = (t <= bg[j]) ? cand : pos[j];
      }
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) kz[j] = (int)((unsigned)(pos[j] + 48) / 48u);
  }
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const int n2 = NBLK == 2 ? tw_block_split(nc) : 0;
    const int k0 = b == 0 ? 0 : n2, n = NBLK == 1 ? nc : (b == 0 ? n2 : nc - n2);
    const int ng = n >> 3, full = k0 + 8 * ng;
    const int sslot = (NBLK == 2 && b == 1) ? 3 : 5;
    double v[JT];
    int kst[JT];
    bool cut[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      kst[j] = kz[j] > k0 ? kz[j] : k0;  // the block's first cell that is not below the cut
      cut[j] = kz[j] >= k0 && kz[j] < full;
      const PsibCell c = cv.load(cut[j] ? kz[j] : SENT);
      v[j] = NBLK == 2 ? psib_term1_y(c, bg[j], deg) : psib_term1(c, bg[j], deg);
    }
    // the cut cell's accumulator: its term, then the cells above it, one by one (the cuts ascend
    // with the lane and the slot: lane 0 of a slot bounds the slot's chains)
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int kl = __builtin_amdgcn_readfirstlane(kz[j]);
      const int kfirst = kl > k0 ? kl : k0;
      // (unrolled by four: the reads of four steps go out together, the additions follow)
#pragma unroll 4
      for (int t = 1; t <= ((full - 1 - kfirst) >> 3); ++t) {
        const int kk = kz[j] + 8 * t;
        v[j] += cv.u((cut[j] && kk < full) ? kk : SENT);
      }
    }
    // The other accumulators hold S of the seven cells above kst (accumulator (kst + i) mod 8
    // has cell kst + i first), the tree pairs accumulators (0,1) (2,3) | (4,5) (6,7): with
    // s_i = S[kst + i] the pairs are, for an even kst, (v, s1) (s2, s3) (s4, s5) (s6, s7) and for
    // an odd one (s7, v) (s1, s2) (s3, s4) (s5, s6) -- in accumulator order starting at the cut
    // cell's own pair, number (kst >> 1) & 3 of the four.  Additions commute, so only which
    // pairs share a quad matters.  A block without a cut (all of it above: kst = k0; all of it
    // below: the zero row) takes S[kst] for v.
    double rb[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int row = kst[j] < full ? kst[j] : ZROW;
      const double *sp = cell + (size_t)row * 6 + sslot;
      const double s0 = sp[0], s1 = sp[6], s2 = sp[12], s3 = sp[18], s4 = sp[24], s5 = sp[30],
                   s6 = sp[36], s7 = sp[42];
      const double vv = cut[j] ? v[j] : s0;
      const bool odd = (kst[j] & 1) != 0, podd = (kst[j] & 2) != 0;
      const double p0 = vv + (odd ? s7 : s1);
      const double qa = s1 + s2, qb = s2 + s3, qc = s3 + s4, qd = s4 + s5, qe = s5 + s6, qf = s6 + s7;
      const double p1 = odd ? qa : qb, p2 = odd ? qc : qd, p3 = odd ? qe : qf;
      rb[j] = (p0 + (podd ? p3 : p1)) + (p2 + (podd ? p1 : p3));
    }
    if (b == 0 && exc != 0u) {  // wave-uniform: cells below K0 that this pass's classes can see
      // (rare passes: the eight accumulators again, those of the cells below K0 cell by cell)
      double r[8][JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int base8 = kst[j] & ~7, a0 = kst[j] & 7;
        const int rlo = base8 < full ? base8 : ZROW, rhi = base8 + 8 < full ? base8 + 8 : ZROW;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int row = a < a0 ? rhi : rlo;
          const double s = cell[(size_t)(row + a) * 6 + sslot];
          r[a][j] = (cut[j] && a == a0) ? v[j] : s;
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if ((exc >> e) & 1u) {
          for (int t = 0; t < ng; ++t) {
            const PsibCell c = cv.load(e + 8 * t);
#pragma unroll
            for (int j = 0; j < JT; ++j) {
              const double term = psib_cell_term1(c, bg[j]);
              r[e][j] = (t == 0) ? term : r[e][j] + term;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < JT; ++j)
        rb[j] = ((r[0][j] + r[1][j]) + (r[2][j] + r[3][j])) + ((r[4][j] + r[5][j]) + (r[6][j] + r[7][j]));
    }
    for (int k = full; k < k0 + n; ++k) {  // NumPy's tail (regular cells), one by one
      const PsibCell c = cv.load(k);
#pragma unroll
      for (int j = 0; j < JT; ++j)
        rb[j] += NBLK == 2 ? psib_term1_y(c, bg[j], deg) : psib_term1(c, bg[j], deg);
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) res[j] = (b == 0) ? rb[j] : res[j] + rb[j];
  }
  if (deg) {  // wave-uniform: a class ON a zero-thickness cell's buoyancy (the cell right below
              // the cut: tops ascend) has a NaN mask there, hence a NaN sum (H6)
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const double2 td = *reinterpret_cast<const double2 *>(cell + (size_t)(kz[j] > 0 ? kz[j] - 1 : 0) * 6);
      if (kz[j] > K0 && td.y == 0. && td.x == bg[j]) res[j] = __builtin_nan("");
    }
  }
}

// Exclusive running sum of a sequence held in registers (element i = lane i / P, slot i % P;
// the caller zeroes the slots of elements that do not count): pre[p] <- d_0 + ... + d_{i-1}.
// Order (restated by the oracle's lane_blocked_scan, so the two stay bit-identical): every
// lane sums its slots left to right, the lane totals go through a Hillis-Steele inclusive scan
// (x[l] = x[l-d] + x[l], d = 1, 2, ..., 32), and pre = (total of the lanes to the left) +
// (running sum inside the lane).  6 + P dependent additions; the first version of the solve
// ran two ordered 100-term chains through LDS, a quarter of the kernel.
template <int P>
__device__ __forceinline__ void lane_blocked_scan(const double (&d)[P], double (&pre)[P],
                                                  int lane) {
  double r = 0.;
  double run[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    run[p] = r;
    r = r + d[p];
  }
  double x = r;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const double o = __shfl_up(x, dd, 64);
    if (lane >= dd) x = o + x;
  }
  double e = __shfl_up(x, 1, 64);
  if (lane == 0) e = 0.;
#pragma unroll
  for (int p = 0; p < P; ++p) pre[p] = (p == 0) ? e : e + run[p];
}

// Minimum / maximum over the wave by DPP moves (quad swaps, half-row and row mirrors) and four
// v_readlane -- the shuffles of group_min / group_max are twelve dependent LDS round trips, 5 %
// of a member's update.  v_min_f64 / v_max_f64: NaNs lose (the callers test for NaN apart).
template <int CTRL>
__device__ __forceinline__ double tw_dpp(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tw_lane(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                          __builtin_amdgcn_readlane(__double2loint(x), l));
}
template <bool MAX>
__device__ __forceinline__ double tw_pick(double a, double b) {
  return MAX ? __builtin_fmax(a, b) : __builtin_fmin(a, b);
}
template <bool MAX>
__device__ __forceinline__ double tw_wave_reduce(double x) {
  x = tw_pick<MAX>(x, tw_dpp<0xB1>(x));   // quad_perm [1,0,3,2]
  x = tw_pick<MAX>(x, tw_dpp<0x4E>(x));   // quad_perm [2,3,0,1]
  x = tw_pick<MAX>(x, tw_dpp<0x141>(x));  // row_half_mirror
  x = tw_pick<MAX>(x, tw_dpp<0x140>(x));  // row_mirror: every lane holds its row's result
  return tw_pick<MAX>(tw_pick<MAX>(tw_lane(x, 0), tw_lane(x, 16)),
                      tw_pick<MAX>(tw_lane(x, 32), tw_lane(x, 48)));
}

__device__ __forceinline__ void tw_pass_priority(int pass) {
  if (pass <= 0)
    __builtin_amdgcn_s_setprio(3);
  else if (pass == 1)
    __builtin_amdgcn_s_setprio(2);
  else if (pass == 2)
    __builtin_amdgcn_s_setprio(1);
  else
    __builtin_amdgcn_s_setprio(0);
}

// P <= 2 (nz <= 128): held to 128 registers, i.e. 4 waves per SIMD -- BASELINE's 4096-member
// ensembles then run as ONE batch of resident waves.  (nz = 200 needs 135; capping it at 128
// with 8 spilled registers, and laying the psib row over the cells so that 16 waves fit the
// LDS, both left config 5's 165 us unchanged.)
// One member's update: `s_cell` = the wave's tw_lds_doubles(nz, nb) doubles of LDS.  The body of
// k_thermwind, and the diagnostic phase of the persistent run kernels (coupled_run.hip), which
// call it between two blocks of time steps of the same wave.
template <int P, int BIG, int W = 6, bool CHAIN = true>
__device__ __forceinline__ void tw_member(const pm_thermwind &a, int ops, int m_raw,
                                          double *s_cell, int lane) {
  static_assert(W == 6 || (W == 5 && tw_w5_ok(P)), "cell layout");
  const bool m_ok = m_raw < a.n;
  const int m = m_ok ? m_raw : a.n - 1;
  const int nz = a.nz, nb = a.nb;
  const int ngrp = (nz + 7) >> 3;
  // s_cell: the cells of the Psib sum, layout W (CellView)
  double *s_a = s_cell;                                // [nz]  G of the solve (before Psib)
  double *s_b = s_a + nz;                              // [nz]  I of the solve
  const bool overlay = P >= 3 && tw_overlay(nz, nb, W);
  const int ncell = nz + (CHAIN ? tw_pad_cells(nz, W) : 0);  // (the chain-order path's zero row may need more)
  double *s_psib = overlay ? s_cell : s_cell + W * ncell;  // [nb]
  double *s_gbot = s_cell + W * ncell + (overlay ? 0 : nb);  // [ngrp] min(bot) of each 8-cell group
  double *s_gtop = s_gbot + ngrp;                   // [ngrp] max(top)
  const size_t base = (size_t)m * nz;

  PM_TICK_INIT
  PM_WAVE_BEGIN
  double z[P], zu[P], b1[P], b2[P], b1u[P], b2u[P], Psi[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    const int ic = i < nz ? i : nz - 1;
    const int iu = ic + 1 < nz ? ic + 1 : nz - 1;
    z[p] = a.z[ic];
    zu[p] = a.z[iu];
    b1[p] = a.b1[base + ic];
    b2[p] = a.b2[base + ic];
    b1u[p] = a.b1[base + iu];
    b2u[p] = a.b2[base + iu];
    Psi[p] = 0.;
  }
  // (the scalars of the solve are requested with the profiles, not one memory round trip each
  // where they are used)
  const bool solving = (ops & PM_TW_SOLVE) != 0;
  const double f_m = solving ? a.f[m] : 1.;
  const double z_first = solving ? a.z[0] : 0., z_last = solving ? a.z[nz - 1] : 1.;

  PM_TICK(0)
  if (solving) {
    const double rf = 1. / f_m;  // psi_thermwind.py:123
    double g[P], gu[P], h[P], dG[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      dG[p] = 0.;
      g[p] = rf * (b2[p] - b1[p]);
      gu[p] = rf * (b2u[p] - b1u[p]);
      h[p] = zu[p] - z[p];
      if (i < nz - 1) {
        const double zm = z[p] + 0.5 * h[p];
        const double s1 = (b1u[p] - b1[p]) / h[p], s2 = (b2u[p] - b2[p]) / h[p];
        const double b1m = a.b1_mid ? a.b1_mid[base + i] : s1 * (zm - z[p]) + b1[p];
        const double b2m = a.b2_mid ? a.b2_mid[base + i] : s2 * (zm - z[p]) + b2[p];
        const double gm = rf * (b2m - b1m);
        dG[p] = h[p] / 6. * (g[p] + gu[p] + 4. * gm);  // dG over [z_i, z_i+1]
      }
    }
    PM_TICK(1)
    double Gl[P];
    lane_blocked_scan<P>(dG, Gl, lane);  // G at the lane's levels
    PM_TICK(2)
    const double G_next = from_next_lane(Gl[0]);
    double dI[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      dI[p] = 0.;
      if (i < nz - 1) {
        const double Gu = (p < P - 1) ? Gl[p + 1 < P ? p + 1 : p] : G_next;
        const double Gm = 0.5 * (Gl[p] + Gu) - 0.125 * h[p] * (gu[p] - g[p]);
        dI[p] = h[p] / 6. * (Gl[p] + Gu + 4. * Gm);
      }
    }
    PM_TICK(3)
    double Il[P];
    lane_blocked_scan<P>(dI, Il, lane);  // I at the lane's levels
    PM_TICK(4)
    double Iend = 0.;  // I at the top level nz-1
#pragma unroll
    for (int p = 0; p < P; ++p)
      if ((nz - 1) % P == p) Iend = lane_value(Il[p], (nz - 1) / P);
    const double z0 = z_first, span = z_last - z0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      Psi[p] = (Il[p] - Iend * ((z[p] - z0) / span)) / 1e6;  // Sv
      if (i < nz && m_ok) {
        a.Psi[base + i] = Psi[p];
        if (a.dPsi) a.dPsi[base + i] = Gl[p] - Iend / span;  // d(Psi 1e6)/dz
        // z-space coupling wA = AMOC.Psi * 1e6 (examples/example_timestepping.py:75)
        if ((ops & PM_TW_WA_PSI) && a.wA1) a.wA1[base + i] = Psi[p] * 1e6;
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      Psi[p] = a.Psi[base + (i < nz ? i : nz - 1)];
    }
  }

  PM_TICK(5)
  if (!(ops & PM_TW_PSIB)) {  // wave-uniform
    PM_TICK_FLUSH
    return;
  }

  // ---- Psib (psi_thermwind.py:170-185)
  double mn = b1[0], mx = b1[0];
  bool has_nan = false;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (lane * P + p < nz) {
      has_nan |= (b1[p] != b1[p]) || (b2[p] != b2[p]);
      mn = b1[p] < mn ? b1[p] : mn;
      mn = b2[p] < mn ? b2[p] : mn;
      mx = b1[p] > mx ? b1[p] : mx;
      mx = b2[p] > mx ? b2[p] : mx;
    }
  }
  // lanes past the last level hold copies of b[nz-1]: harmless for min/max
  mn = tw_wave_reduce<false>(mn);
  mx = tw_wave_reduce<true>(mx);
  if (__ballot(has_nan) != 0ull) mn = mx = __builtin_nan("");
  Linspace lin;
  lin.init(mn, mx, nb);

  PM_TICK(6)
  const bool range_ok = __builtin_fabs(mn) < 1e100 && __builtin_fabs(mx) < 1e100;
  const double Psi_up0 = from_next_lane(Psi[0]);
  // min(bot) / max(top) of this lane's cells for the group ranges; (-inf, +inf) = "holds a cell
  // that bars its group from the shortcuts"
  double lane_gb = __builtin_inf(), lane_gt = -__builtin_inf();
  // chain order (psib_sorted_pass): what this lane's cells contribute to the test
  double c_bot[P], c_top[P];
  bool c_plain[P], c_flat[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    c_bot[p] = c_top[p] = __builtin_inf();  // (slots past the last cell: in order with anything)
    c_plain[p] = true;
    c_flat[p] = false;
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int k = lane * P + p;
    if (k < nz - 1) {
      const double Pu = (p < P - 1) ? Psi[p + 1 < P ? p + 1 : p] : Psi_up0;
      const double u = -(Pu - Psi[p]);  // :175
      const bool north = u < 0;         // :179-181
      const double top = north ? b2u[p] : b1u[p];
      const double bot = north ? b2[p] : b1[p];
      const double d = top - bot, ad = __builtin_fabs(d);
      // regular: every product of the reciprocal division stays far from over- and underflow
      const bool regular = range_ok && ad >= 1e-100 && ad <= 1e100 && __builtin_fabs(top) < 1e100;
      const double yh = regular ? 1.0 / d : __builtin_nan("");
      double2 *cell = reinterpret_cast<double2 *>(s_cell + (size_t)k * (W == 6 ? 6 : 4));
      cell[0] = double2{top, d};
      cell[1] = double2{yh, recip_lo(d, yh)};
      // sixth slot: bot for the group ranges, -inf for a cell that bars its group from the
      // shortcuts.  A non-finite u_k (user-assigned Psi) must reach the products: 0 * NaN and 0 * inf are
      // NaN in the reference's `mask * udydz` (psi_thermwind.py:183-184), so such a cell bars
      // its group from both shortcuts like a degenerate cell does
      const bool ufin = __builtin_fabs(u) <= 1.7976931348623157e308;
      const bool plain = regular && d > 0. && ufin;
      if constexpr (W == 6)
        cell[2] = double2{u, plain ? bot : -__builtin_inf()};
      else
        s_cell[4 * nz + k] = u;
      lane_gb = plain ? __builtin_fmin(lane_gb, bot) : -__builtin_inf();
      lane_gt = (plain && lane_gt != __builtin_inf()) ? __builtin_fmax(lane_gt, top) : __builtin_inf();
      c_bot[p] = bot;
      c_top[p] = top;
      // (in chain order a cell of zero thickness is as good as a regular one: psib_deg_term)
      const bool flat = d == 0. && range_ok && __builtin_fabs(top) < 1e100 && ufin;
      c_plain[p] = plain || flat;
      c_flat[p] = flat;
    }
  }
  // K0 = 1 + the highest cell that is not plain or whose top lies above the next cell's bot
  int K0 = 0;
  constexpr int TW_NBLK = (CHAIN && W == 6 && P <= 4 && BIG <= 1) ? (BIG == 1 ? 2 : 1) : 0;
  if constexpr (TW_NBLK != 0) {
    const double bot_nl = from_next_lane(c_bot[0]);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int k = lane * P + p;
      const double nbot = (p < P - 1) ? c_bot[p + 1 < P ? p + 1 : p] : ((lane < 63) ? bot_nl : __builtin_inf());
      const bool bad = k < nz - 1 && !(c_plain[p] && c_top[p] <= nbot);
      const unsigned long long bm = __ballot(bad);
      const int kb = bm ? (63 - __builtin_clzll(bm)) * P + p + 1 : 0;
      K0 = kb > K0 ? kb : K0;
    }
  }
  int deg_hi = -1;  // the highest cell of zero thickness at or above K0
  if constexpr (TW_NBLK != 0) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const unsigned long long fm = __ballot(c_flat[p] && lane * P + p >= K0);
      const int kf = fm ? (63 - __builtin_clzll(fm)) * P + p : -1;
      deg_hi = kf > deg_hi ? kf : deg_hi;
    }
  }
  const bool chain_order = TW_NBLK != 0 && K0 <= 2 && nz - 1 >= 16 &&
                           tw_sorted_blocks(nz - 1) == TW_NBLK && range_ok && !(mn != mn);
  __builtin_amdgcn_wave_barrier();
  PM_TICK(7)
  const int nc = nz - 1;
  if constexpr (P == 1 || P == 2 || P == 4 || P == 8) {
    // a group of 8 cells is 8 / P neighbouring lanes: reduce there (the loop below walks its
    // 16 LDS reads per group one after the other: 5 % of a member's update)
    double gb = lane_gb, gt = lane_gt;
    if constexpr (P <= 4) {
      gb = __builtin_fmin(gb, tw_dpp<0xB1>(gb));  // lanes 2i, 2i+1
      gt = __builtin_fmax(gt, tw_dpp<0xB1>(gt));
    }
    if constexpr (P <= 2) {
      gb = __builtin_fmin(gb, tw_dpp<0x4E>(gb));  // the quad
      gt = __builtin_fmax(gt, tw_dpp<0x4E>(gt));
    }
    if constexpr (P == 1) {
      gb = __builtin_fmin(gb, tw_dpp<0x141>(gb));  // row_half_mirror: the other quad of the eight
      gt = __builtin_fmax(gt, tw_dpp<0x141>(gt));
    }
    constexpr int LPG = 8 / P;  // lanes per group
    const int g = lane / LPG;
    if (lane % LPG == 0 && g < ngrp) {
      const bool ok = nb >= nz;
      s_gbot[g] = ok ? gb : -__builtin_inf();
      s_gtop[g] = ok ? gt : __builtin_inf();
    }
  } else
  for (int g = lane; g < ngrp; g += 64) {
    double gb = __builtin_inf(), gt = -__builtin_inf();
    bool ok = nb >= nz;
    for (int a = 0; a < 8; ++a) {
      const int k = g * 8 + a;
      if (k < nc && ok) {
        const double bk = s_cell[(size_t)k * 6 + 5];  // (W == 6 on this path)
        ok = bk != -__builtin_inf();
        gb = bk < gb ? bk : gb;
        const double tk = s_cell[(size_t)k * 6];
        gt = tk > gt ? tk : gt;
      }
    }
    s_gbot[g] = ok ? gb : -__builtin_inf();  // never "all masks 1"
    s_gtop[g] = ok ? gt : __builtin_inf();   // never "all masks 0"
  }
  __builtin_amdgcn_wave_barrier();
  double exc_top[2] = {__builtin_inf(), __builtin_inf()};
  if constexpr (TW_NBLK != 0) {
    if (chain_order) {  // wave-uniform
      const int nc_ = nz - 1, zrow = tw_zero_row(nc_);
      // sentinel cell and the zero row's sixth slots (the group ranges above were the last
      // readers of `bot` there)
      if (lane == 0) {
        double2 *sc = reinterpret_cast<double2 *>(s_cell + (size_t)nc_ * 6);
        sc[0] = double2{__builtin_inf(), 1.};
        sc[1] = double2{1., 0.};
        sc[2] = double2{-0., 0.};
      }
      if (lane < 8 && zrow + lane != nc_) {
        s_cell[(size_t)(zrow + lane) * 6 + 5] = 0.;
        if (TW_NBLK == 2) s_cell[(size_t)(zrow + lane) * 6 + 3] = 0.;
      }
      __builtin_amdgcn_wave_barrier();
      // S[k] = ((u[k] + u[k+8]) + u[k+16]) + ... over the full 8-cell groups of k's block
      const int n2 = TW_NBLK == 2 ? tw_block_split(nc_) : 0;
      double S[P];
      int sfull[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int k = lane * P + p;
        const int k0 = (TW_NBLK == 2 && k >= n2) ? n2 : 0;
        const int n = TW_NBLK == 1 ? nc_ : (k0 == 0 ? n2 : nc_ - n2);
        sfull[p] = k0 + ((n >> 3) << 3);
        S[p] = (k < sfull[p]) ? s_cell[(size_t)k * 6 + 4] : 0.;
      }
      const int tmax = ((TW_NBLK == 2 ? (n2 > nc_ - n2 ? n2 : nc_ - n2) : nc_) >> 3) - 1;
#pragma unroll 4
      for (int t = 1; t <= tmax; ++t) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int kk = lane * P + p + 8 * t;
          S[p] += s_cell[(size_t)(kk < sfull[p] ? kk : nc_) * 6 + 4];
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int k = lane * P + p;
        if (TW_NBLK == 2 && k >= n2) {  // second block: S in the `yl` slot, zeros in the sixth
          if (k < zrow) s_cell[(size_t)k * 6 + 3] = S[p];
          if (k < nc_) s_cell[(size_t)k * 6 + 5] = 0.;
        } else if (k < zrow) {
          s_cell[(size_t)k * 6 + 5] = S[p];
        }
      }
      // cells below K0: a class above a zero-thickness cell (and only that) sees an exact zero
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e < K0) {
          const double et = s_cell[e * 6], ed = s_cell[e * 6 + 1], eu = s_cell[e * 6 + 4];
          const bool quiet = ed == 0. && __builtin_fabs(et) < 1e100 && __builtin_fabs(eu) <= 1.7976931348623157e308;
          exc_top[e] = quiet ? et : __builtin_inf();
        } else {
          exc_top[e] = -__builtin_inf();
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- chain order, and no caller asks for psib itself: only the classes that Psibz's two
  // np.interp calls read are summed -- the two grid nodes around every level's buoyancy.
  // BASELINE's ensembles: 70-135 of the 500 classes, ONE pass of two or three classes per lane.
  // The levels' owners find their intervals (kept for Psibz at <= 2 levels per lane) and mark
  // the nodes; eight ballots over the marks then list the marked classes in ascending order.
  bool lazy = false;
  int n_need = 0;
  constexpr bool TW_KEEP_J = P <= 4;
  unsigned jkeep[TW_KEEP_J ? P : 1];  // (j + 1 of a level's two profiles in the halves of a word)
#pragma unroll
  for (int p = 0; p < (TW_KEEP_J ? P : 1); ++p) jkeep[p] = 0u;
  const unsigned short *lazy_list = reinterpret_cast<const unsigned short *>(s_gbot + 8);
  if constexpr (TW_NBLK != 0) {
    lazy = chain_order && a.psib == nullptr && a.bgrid == nullptr && (ops & PM_TW_PSIBZ) != 0 &&
           nb >= 2 && nb <= 512 && lin.step > 0. && lin.step < 1e300;
    if (lazy) {  // wave-uniform
      unsigned *marks = reinterpret_cast<unsigned *>(s_gbot);
      unsigned short *list = reinterpret_cast<unsigned short *>(s_gbot + 8);
      if (lane < 16) marks[lane] = 0u;
      __builtin_amdgcn_wave_barrier();
      const double rstep_m = 1.0 / lin.step;
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const double x = c == 0 ? b1[p] : b2[p];
          int j = -1;
          if (lane * P + p < nz && x >= lin.start && x <= lin.stop) {  // (false for a NaN)
            double xj, xj1;
            j = interp_uniform_index(x, lin, nb, rstep_m, xj, xj1);
            const int j1 = j + 1 < nb ? j + 1 : nb - 1;
            atomicOr(&marks[j >> 5], (1u << (j & 31)) | ((j1 >> 5) == (j >> 5) ? 1u << (j1 & 31) : 0u));
            if ((j1 >> 5) != (j >> 5)) atomicOr(&marks[j1 >> 5], 1u << (j1 & 31));
          }
          if constexpr (TW_KEEP_J) jkeep[p] |= (unsigned)(j + 1) << (16 * c);
        }
      }
      __builtin_amdgcn_wave_barrier();
      // class 64 t + lane in round t: its place in the list = the marked classes before it
      // (the 16 words come by ONE read and are handed out as scalars: the rounds do not wait
      // on the LDS; v_mbcnt counts the marks below a lane)
      const unsigned mw = marks[lane & 15];
      int count = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const unsigned wl = (unsigned)__builtin_amdgcn_readlane((int)mw, 2 * t);
        const unsigned wh = (unsigned)__builtin_amdgcn_readlane((int)mw, 2 * t + 1);
        const unsigned long long mk = ((unsigned long long)wh << 32) | wl;  // wave-uniform
        if ((mk >> lane) & 1ull) {
          const int at = count + (int)__builtin_amdgcn_mbcnt_hi(wh, __builtin_amdgcn_mbcnt_lo(wl, 0u));
          if (at < TW_LAZY_CAP) list[at] = (unsigned short)(64 * t + lane);
        }
        count += __builtin_popcountll(mk);
      }
      n_need = count;
      if (n_need > 192 || n_need < 1) lazy = false;  // (two or three classes per lane)
      __builtin_amdgcn_wave_barrier();
    }
  }
  PM_TICK(8)
  const CellView<W> cv{s_cell, nz};
  PsibRange rg;
  rg.gbot = s_gbot;
  rg.gtop = s_gtop;
  // A NaN anywhere in b1 / b2 makes bgrid all NaN (np.min / np.max propagate it), every mask
  // NaN and every class sum NaN -- a blown-up member (the reference loses two of config 5's
  // 4096) would otherwise drag all nb x (nz-1) pairs through the IEEE-division path and decide
  // the kernel's duration: 165 us instead of 120 us in config 5.
  const bool all_nan = mn != mn;  // wave-uniform
  double held[TW_HELD][TW_JT];
#pragma unroll
  for (int q = 0; q < TW_HELD; ++q)
#pragma unroll
    for (int j = 0; j < TW_JT; ++j) held[q][j] = __builtin_nan("");
  if (all_nan && overlay) __builtin_amdgcn_wave_barrier();
  if (all_nan) {
    for (int i = lane; i < nb; i += 64) {
      s_psib[i] = __builtin_nan("");
      if (m_ok && a.psib) a.psib[(size_t)m * nb + i] = __builtin_nan("");
      if (m_ok && a.bgrid) a.bgrid[(size_t)m * nb + i] = lin.at(i);
    }
  }
  int klo = K0;  // (chain order: no class of the pass cuts a cell below this one)
  if constexpr (TW_NBLK != 0) {
    if (lazy && !all_nan) {  // wave-uniform: ONE pass over the listed classes, JT per lane
      auto listed_pass = [&](auto jt_c) {
        constexpr int JT = decltype(jt_c)::value;
        double bg[JT], res[JT];
        int ci[JT], kzc[JT];
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          const int i = j * 64 + lane;
          ci[j] = (int)lazy_list[i < n_need ? i : n_need - 1];
          bg[j] = lin.at(ci[j]);
        }
        const double gmin_p = lin.at((int)lazy_list[0]);
        const unsigned exc = (!(gmin_p > exc_top[0]) ? 1u : 0u) | (!(gmin_p > exc_top[1]) ? 2u : 0u);
        psib_sorted_pass<TW_NBLK, JT>(s_cell, nc, K0, K0, exc, deg_hi >= K0, bg, res, kzc);
        if (overlay) __builtin_amdgcn_wave_barrier();  // the cells are dead: the row takes their place
#pragma unroll
        for (int j = 0; j < JT; ++j)
          if (j * 64 + lane < n_need) s_psib[ci[j]] = res[j];
      };
      if (n_need <= 128)
        listed_pass(std::integral_constant<int, 2>{});
      else
        listed_pass(std::integral_constant<int, 3>{});
    }
  }
  for (int i0 = 0; i0 < ((all_nan || lazy) ? 0 : nb); i0 += 64 * TW_JT) {
    // Issue priority falls with the pass: the SIMD's arbiter favours its oldest wave, so four
    // members of equal cost finish 25 / 29 / 34 / 41 us after the launch and the last one runs
    // alone; a wave that is a pass ahead yields to the ones behind and they finish together.
    tw_pass_priority(i0 / (64 * TW_JT));
    double bg[TW_JT], res[TW_JT];
#pragma unroll
    for (int j = 0; j < TW_JT; ++j) {
      const int i = i0 + j * 64 + lane;
      bg[j] = lin.at(i < nb ? i : nb - 1);
    }
    {  // class range of this pass (bgrid ascends; NaN ranges fail both tests)
      const int ilast = i0 + 64 * TW_JT - 1;
      rg.gmin = lin.at(i0);
      rg.gmax = lin.at(ilast < nb ? ilast : nb - 1);
    }
    bool done = false;
    if constexpr (TW_NBLK != 0) {
      if (chain_order) {  // wave-uniform
        const double gmin_p = rg.gmin;
        const unsigned exc = (!(gmin_p > exc_top[0]) ? 1u : 0u) | (!(gmin_p > exc_top[1]) ? 2u : 0u);
        int kzc[TW_JT];
        // (zero-thickness cells matter to a pass that can cut at or right above one, or whose
        // tail holds one: the cuts ascend with the passes)
        const bool deg = deg_hi + 1 >= klo || deg_hi >= tw_zero_row(nc);
        psib_sorted_pass<TW_NBLK, TW_JT>(s_cell, nc, klo, K0, exc, deg, bg, res, kzc);
        klo = __builtin_amdgcn_readlane(kzc[TW_JT - 1], 63);
        done = true;
      }
    }
    if (done) {
    } else if constexpr (BIG == 2) {
      psib_pairwise<4, CellView<W>>(cv, 0, nc, bg, res, rg PM_TICK_ARG);
    } else if constexpr (BIG == 1) {
      // 128 < nc <= 256: NumPy's recursion is exactly two blocks, both inlined
      int n2 = nc / 2;
      n2 -= n2 % 8;
      double r2[TW_JT];
      psib_block_sum(cv, 0, n2, bg, res, rg PM_TICK_ARG);
      psib_block_sum(cv, n2, nc - n2, bg, r2, rg PM_TICK_ARG);
#pragma unroll
      for (int j = 0; j < TW_JT; ++j) res[j] = res[j] + r2[j];
    } else {
      psib_block_sum(cv, 0, nc, bg, res, rg PM_TICK_ARG);  // nc <= 128: one pairwise block
    }
#pragma unroll
    for (int j = 0; j < TW_JT; ++j) {
      const int i = i0 + j * 64 + lane;
      if (i < nb) {
        if (!overlay) s_psib[i] = res[j];
        if (m_ok && a.psib) a.psib[(size_t)m * nb + i] = res[j];
        if (m_ok && a.bgrid) a.bgrid[(size_t)m * nb + i] = bg[j];
      }
    }
    if (overlay) {  // wave-uniform pass index: a scalar select per slot
      const int ps = i0 / (64 * TW_JT);
#pragma unroll
      for (int q = 0; q < TW_HELD; ++q)
#pragma unroll
        for (int j = 0; j < TW_JT; ++j) held[q][j] = (q == ps) ? res[j] : held[q][j];
    }
  }
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_wave_barrier();
  if (overlay && !all_nan && !lazy) {  // the cells are dead: the psib row takes their place
#pragma unroll
    for (int q = 0; q < TW_HELD; ++q)
#pragma unroll
      for (int j = 0; j < TW_JT; ++j) {
        const int i = (q * TW_JT + j) * 64 + lane;
        if (i < nb) s_psib[i] = held[q][j];
      }
    __builtin_amdgcn_wave_barrier();
  }

  PM_TICK(9)
  if (!(ops & PM_TW_PSIBZ)) {
    PM_TICK_FLUSH
    return;
  }
  // ---- Psibz (psi_thermwind.py:203-208) and the drivers' wA coupling
  const double rstep = 1.0 / lin.step;  // (only a starting guess is taken from it)
  // 3-4 levels per lane (129 <= nz <= 256): the two profiles are read again here instead of
  // living in 4 P registers through the class passes (where the kernel sits at its register cap
  // and spills per pass); taller grids run at 1-2 waves per SIMD anyway and gain nothing
  double b1e[P], b2e[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    const int ic = i < nz ? i : nz - 1;
    b1e[p] = (P >= 3 && BIG <= 1) ? a.b1[base + ic] : b1[p];
    b2e[p] = (P >= 3 && BIG <= 1) ? a.b2[base + ic] : b2[p];
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    if (i < nz && m_ok) {
      double p1, p2;
      if (TW_KEEP_J && lazy) {  // (the intervals found when the classes were marked)
        const unsigned jw = jkeep[TW_KEEP_J ? p : 0];
        p1 = interp_uniform_at(b1e[p], (int)(jw & 0xffffu) - 1, lin, s_psib, nb);
        p2 = interp_uniform_at(b2e[p], (int)(jw >> 16) - 1, lin, s_psib, nb);
      } else {
        p1 = interp_uniform(b1e[p], lin, s_psib, nb, rstep);
        p2 = interp_uniform(b2e[p], lin, s_psib, nb, rstep);
      }
      if (a.psibz1) a.psibz1[base + i] = p1;
      if (a.psibz2) a.psibz2[base + i] = p2;
      if (a.wA1) {  // (Psi_iso_b - SO.Psi) * 1e6   (example_twocol_plusSO.py:105)
        const double v = a.Psi_SO ? (p1 - a.Psi_SO[base + i]) : p1;
        a.wA1[base + i] = v * 1e6;
      }
      if (a.wA2) a.wA2[base + i] = (-p2) * 1e6;  // -Psi_iso_n * 1e6 (:106)
    }
  }
  PM_TICK(10)
  PM_TICK_FLUSH
  PM_WAVE_END(m_raw)
}

#ifndef PM_DIAG_DEVICE_FUNCTIONS_ONLY  // (coupled_run.hip.h shares tw_member only)
template <int P, int BIG>
__global__ __launch_bounds__(64 * TW_WAVES_PER_BLOCK)
__attribute__((amdgpu_waves_per_eu((P <= 4 && BIG != 2) ? 4 : 1))) void k_thermwind(pm_thermwind a,
                                                                       int ops) {
  extern __shared__ double lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m_raw = blockIdx.x * (blockDim.x >> 6) + wave;
  const int per_wave = tw_lds_doubles(a.nz, a.nb);
  tw_member<P, BIG>(a, ops, m_raw, lds_all + (size_t)wave * per_wave, lane);
}

// pm_thermwind_residuals: one thread per interval (a single member's mesh; the host owns
// solve_bvp's mesh loop for CALLABLE profiles).  scipy _bvp.py: create_spline +
// estimate_rms_residuals for y = (y0, y1), f = (y1, g); the mid-point residual of a converged
// collocation solution is zero.
template <int UNUSED = 0>  // (a template: the header is compiled into two translation units)
__global__ void k_thermwind_residuals(int m, const double *x, const double *y0, const double *y1,
                                      const double *g, const double *g_lob, double *rms) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m - 1) return;
  const double h = x[i + 1] - x[i];
  // cubic of component c on [x_i, x_i+1]: ((c0 dx + c1) dx + yp_a) dx + y_a  (create_spline)
  const double ya[2] = {y0[i] * 1e6, y1[i]}, yb[2] = {y0[i + 1] * 1e6, y1[i + 1]};  // y0 arrives in Sv
  const double pa[2] = {y1[i], g[i]}, pb[2] = {y1[i + 1], g[i + 1]};
  double c0[2], c1[2];
  for (int c = 0; c < 2; ++c) {
    const double slope = (yb[c] - ya[c]) / h;
    const double t = (pa[c] + pb[c] - 2 * slope) / h;
    c0[c] = t / h;
    c1[c] = (slope - pa[c]) / h - t;
  }
  const double s = 0.5 * h * 0.6546536707079771;  // sqrt(3/7)
  double acc = 0.;
  for (int side = 0; side < 2; ++side) {
    const double xe = x[i] + 0.5 * h + (side == 0 ? s : -s);
    const double dx = xe - x[i];
    double Y[2], Yp[2];
    for (int c = 0; c < 2; ++c) {
      Y[c] = ((c0[c] * dx + c1[c]) * dx + pa[c]) * dx + ya[c];
      Yp[c] = (3 * c0[c] * dx + 2 * c1[c]) * dx + pa[c];
    }
    const double F0 = Y[1], F1 = g_lob[(size_t)side * (m - 1) + i];
    const double e0 = (Yp[0] - F0) / (1 + fabs(F0)), e1 = (Yp[1] - F1) / (1 + fabs(F1));
    acc += e0 * e0 + e1 * e1;
  }
  rms[i] = sqrt(0.5 * (49. / 90. * acc));
}

template <int P, int BIG>
int launch_thermwind_impl(const pm_thermwind &a, int ops, hipStream_t st) {
  const size_t per_wave = (size_t)tw_lds_doubles(a.nz, a.nb) * sizeof(double);
  // waves per block: as many of TW_WAVES_PER_BLOCK, .../2, 1 as keeps the most waves on a CU
  // (nz = 200: 14 KB per wave -- blocks of 4 would leave room for 8 waves, single waves for 11)
  int wpb = 1, best = 0;
  for (int w = TW_WAVES_PER_BLOCK; w >= 1; w >>= 1) {
    int resident = (int)((160 * 1024) / (per_wave * w)) * w;
    resident = resident > 16 ? 16 : resident;  // 4 waves per SIMD is all the registers allow
    if (resident > best) {
      best = resident;
      wpb = w;
    }
  }
  const size_t lds = per_wave * wpb;
  if (lds > 160 * 1024) return fail(PM_EINVAL, "thermwind needs %zu B of LDS per member", lds);
  if (lds > 64 * 1024)
    PM_HIP(hipFuncSetAttribute((const void *)k_thermwind<P, BIG>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((a.n + wpb - 1) / wpb);
  hipLaunchKernelGGL((k_thermwind<P, BIG>), dim3(grid), dim3(64 * wpb), lds, st, a, ops);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

template <int P>
int launch_thermwind(const pm_thermwind &a, int ops, hipStream_t st) {
  if constexpr (P <= 3) {
    if (a.nz - 1 <= 128) return launch_thermwind_impl<P, 0>(a, ops, st);
  }
  if constexpr (P >= 2 && P <= 5) {
    if (a.nz - 1 > 128 && a.nz - 1 <= 256) return launch_thermwind_impl<P, 1>(a, ops, st);
  }
  return launch_thermwind_impl<P, 2>(a, ops, st);
}

inline int dispatch_thermwind(const pm_thermwind &a, int ops, hipStream_t st) {
  const int P = (a.nz + 63) / 64;
  switch (P) {
#define PM_CASE(PP) \
  case PP:          \
    return launch_thermwind<PP>(a, ops, st);
    PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7)
    PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11) PM_CASE(12) PM_CASE(13) PM_CASE(14)
    PM_CASE(15) PM_CASE(16)
#undef PM_CASE
  }
  return fail(PM_EINVAL, "nz=%d unsupported by thermwind (max 1024)", a.nz);
}

#endif  // PM_DIAG_DEVICE_FUNCTIONS_ONLY

}  // namespace pm
