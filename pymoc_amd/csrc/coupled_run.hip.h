// coupled_run.hip.h -- persistent per-member kernels for WHOLE coupled runs.
//
// The reference's coupled drivers are loops of the form
//     every MOC_up_iters steps: refresh the overturning diagnostics (Psi_SO.solve,
//                               Psi_Thermwind.solve / Psibz), then step the columns
//   examples/example_twocol.py:85-96            (k_twocol_run:  thermal wind + two columns)
//   examples/run_JansenNadeau_2018.py:201-261   (k_jn2018_run:  Psi_SO + thermal wind + the fused
//                                                BC switch / two columns / mixed layer loop)
// Members never interact, so ONE launch can carry a member through many such intervals: a wave
// owns a member, runs the diagnostic phase and the stepping phase of every interval back to
// back, and nothing but that wave ever touches the member's rows.  The phases are the very
// device functions of the stand-alone kernels (so_member, tw_member, jf_member_run /
// tc_member_run): same operations, same order -- the result is bit-identical to the launch
// sequence of the drivers -- exchanging their rows through global memory (they stay in L2) with
// a fence in between.  What the launch sequence paid and this does not: ~3 launch boundaries
// per interval (every one a drain to the slowest wave), 4096 waves pulling their rows at the
// same moment at the top of every stepping launch, and the per-launch grid tables.
//
// Schedule of a launch (pm_run_schedule): n_first steps, then n_updates x [diagnostics, steps],
// the last block stepping n_last (possibly 0) instead of m_steps.
#pragma once
#include "thermwind.hip.h"
#include "psi_so.hip.h"

namespace pm {

// order the phases' global-memory traffic of one wave: the rows a phase wrote (by other lanes
// of the wave) are what the next phase reads.  Workgroup scope: writer and reader are the same
// wave, i.e. the same CU and the same vector L1 -- an agent-scope fence would write back and
// invalidate the XCD's whole L2 at every phase boundary of every wave (measured: 287 us per
// interval of config 3 instead of 66).
__device__ __forceinline__ void run_phase_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// LDS doubles per wave of a run kernel: the phases reuse one region
// (the thermal wind's cells in the 5-double layout at P = 4: nothing else fits; thermwind.hip.h)
template <int P>
constexpr int run_cell_w() { return P == 4 ? 5 : 6; }
template <int P>
inline int run_wave_stride(int nz, int nb, int ny, bool with_so) {
  int w = JfLds<P>::PER_WAVE;
  const int t = tw_lds_doubles(nz, nb, run_cell_w<P>(), false);
  w = t > w ? t : w;
  if (with_so) {
    const int s = so_lds_doubles(nz, ny, false, false);
    w = s > w ? s : w;
  }
  return (w + 1) & ~1;
}

// ------------------------------------------------------------------------ two columns
// nsteps x [basin.timestep, north.timestep] of one member (example_twocol.py:89-90) from the rows
// in HBM and back: pm_column_steps' work for the member's two columns, with the block's grid
// tables (jf_block_tables) and the column step of the fused JN2018 loop (jf_convect,
// jf_vertadvdiff: col_vertadvdiff<64, P, 2> operation by operation).  Columns: rows m (basin) and
// n + m (north) of `c`, uniform Area (verified), nsel = 1 or ksel given, no bzbot.  Each column's
// PM_COL_DO_CONV flag decides between convect() and the b[-1] = bs surface condition.
template <int P, bool VEC>
__device__ __forceinline__ void tc_member_run(const pm_columns &c, const double *wA, double dt,
                                              int nsteps, int m_raw, int n, int32_t *status,
                                              double *lds, int wstride, int wave, int lane) {
  using L = JfLds<P>;
  const bool m_ok = m_raw < n;
  const int m = m_ok ? m_raw : n - 1;
  const int nz = c.nz;
  double *wl = lds + L::WAVE0 + wave * wstride;
  double *ws = wl + L::W_S;
  JfCol<P> cb, cn;
  JfConv<P> vb, vn;
  bool hint_ok = true, range_ok = in_fast_div_range(dt);
  bool conv_b, conv_n;
  double bbot_b, bbot_n;
  {
    auto load_col = [&](JfCol<P> &r, JfConv<P> &v, int col, double *wsc, bool &conv,
                        double &bbot, double *kap_lds) {
      bool same = true;
      const size_t arow = (size_t)col * nz;
      const int sel = c.ksel ? c.ksel[col] : 0;
      const size_t sbase = ((size_t)sel * c.ncols + col) * nz;
      const double a0 = c.area[arow];
      double rb[P], ra_[P], rk[P], rd[P], rw[P];
      jf_load_row<P, VEC>(rb, c.b + arow, lane, nz);
      jf_load_row<P, VEC>(ra_, c.area + arow, lane, nz);
      jf_load_row<P, VEC>(rk, c.kappa + sbase, lane, nz);
      jf_load_row<P, VEC>(rd, c.dAkappa + sbase, lane, nz);
      jf_load_row<P, VEC>(rw, wA + arow, lane, nz);
      const int flags = c.flags ? c.flags[col] : 0;
      conv = (flags & PM_COL_DO_CONV) != 0;
      hint_ok = hint_ok && (flags & PM_COL_BZBOT) == 0;
      bbot = c.bbot[col];
      bool ok = in_fast_div_range(a0) && a0 != 0.0 && in_fast_div_range(c.bs[col]) &&
                in_fast_div_range(c.N2min[col]) && in_fast_div_range(bbot);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = lane * P + p;
        r.b[p] = i < nz ? rb[p] : JF_PAD;
        same = same && ra_[p] == a0;
        r.kap[p] = rk[p];
        if (kap_lds) kap_lds[jf_entry<P>(lane, p)] = rk[p];
        const double w = rw[p] - rd[p];  // column.py:241
        const bool interior = i >= 1 && i <= nz - 2;
        const double we = interior ? w : 0.0;
        r.wn[p] = (we < 0.0) ? -we : 0.0;
        r.wp[p] = (we < 0.0) ? 0.0 : -we;
        ok = ok && (i >= nz || in_fast_div_range(r.b[p])) && in_fast_div_range(we) &&
             in_fast_div_range(rk[p]);
      }
      hint_ok = hint_ok && __ballot(!same) == 0ull;
      range_ok = range_ok && __ballot(!ok) == 0ull;
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
    load_col(cb, vb, m, ws + S_B, conv_b, bbot_b, nullptr);
    load_col(cn, vn, n + m, ws + S_NN, conv_n, bbot_n, wl + L::W_KN);
  }
  {  // the grid's part of the operand window
    bool ok = true;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p, e = jf_entry<P>(lane, p);
      const double zv = lds[L::T_Z + e], dzv = lds[L::T_DZ + e], dzcv = lds[L::T_DZC + e];
      ok = ok && (i >= nz || (in_fast_div_range(zv) &&
                              (i >= nz - 1 || (in_fast_div_range(dzv) && dzv != 0.0)) &&
                              in_fast_div_range(dzcv) && dzcv != 0.0));
    }
    range_ok = range_ok && __ballot(!ok) == 0ull;
  }
  __builtin_amdgcn_wave_barrier();
  const bool lane0 = lane == 0;
  // the surface condition of a column without convective adjustment (column.py:230-231) is a
  // constant: imposed once (the step leaves boundary levels where they are)
  const double bs_b = ws[S_B + S_BS], bs_n = ws[S_NN + S_BS];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (!conv_b && lane * P + p == nz - 1) cb.b[p] = bs_b;
    if (!conv_n && lane * P + p == nz - 1) cn.b[p] = bs_n;
  }
  auto run_leg = [&](auto ieee_c) {
    for (int s = 0; s < (hint_ok ? nsteps : 0); ++s) {
      int lane_o = lane, woff = wave * wstride;
      asm volatile("" : "+v"(lane_o), "+s"(woff));
      double *wl = lds + L::WAVE0 + woff, *ws = wl + L::W_S;
      if (conv_b) jf_convect<P>(cb.b, vb, lds, ws + S_B, lane_o, nz);
      if (conv_n) jf_convect<P>(cn.b, vn, lds, ws + S_NN, lane_o, nz);
      cb.b[0] = lane0 ? bbot_b : cb.b[0];  // column.py:232
      cn.b[0] = lane0 ? bbot_n : cn.b[0];
      if constexpr (decltype(ieee_c)::value) {
        jf_vertadvdiff_ieee<P, true>(cb, lds, ws + S_B, nullptr, lane_o, dt, nz);
        jf_vertadvdiff_ieee<P, false>(cn, lds, ws + S_NN, wl + L::W_KN, lane_o, dt, nz);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        jf_vertadvdiff<P, true>(cb, lds, ws + S_B, nullptr, lane_o, dt);
        __builtin_amdgcn_sched_barrier(0);
        jf_vertadvdiff<P, false>(cn, lds, ws + S_NN, wl + L::W_KN, lane_o, dt);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (__builtin_expect(range_ok, 1))
    run_leg(std::false_type{});
  else
    run_leg(std::true_type{});
  // ---- results
  bool bad = false;
  if (m_ok && hint_ok) {
    jf_store_row<P, VEC>(c.b + (size_t)m * nz, cb.b, lane, nz);
    jf_store_row<P, VEC>(c.b + (size_t)(n + m) * nz, cn.b, lane, nz);
  }
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (lane * P + p < nz) bad |= !isfinite(cb.b[p]) || !isfinite(cn.b[p]);
  const bool anybad = __ballot(bad) != 0ull;
  if (lane == 0 && m_ok) {
    if (c.nonfinite) {
      c.nonfinite[m] = anybad ? 1 : 0;
      c.nonfinite[n + m] = anybad ? 1 : 0;
    }
    // sticky over the intervals of a run: bit 1 non-finite, bit 4 the member's columns are not what
    // this kernel steps (Area varying in z, bzbot: left untouched), bit 5 IEEE leg taken
    if (status) status[m] |= (anybad ? 2 : 0) | (hint_ok ? 0 : 16) | (range_ok ? 0 : 32);
  }
}

template <int P, bool VEC, int BIG>
__global__ __launch_bounds__(64 * JF_WAVES) JF_OCC_ATTR
void k_twocol_run(pm_twocol_loop r, int wstride) {
  using L = JfLds<P>;
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m_raw = blockIdx.x * JF_WAVES + wave;
  {
    pm_jn2018 g;  // (only the grid enters the tables of a two-column run)
    g.cols = r.cols;
    g.ml.ny = 0;
    g.ml.y = nullptr;
    jf_block_tables<P>(g, r.dt, lds, wave, lane);
  }
  __syncthreads();
  const int n = r.tw.n;
  double *wl = lds + L::WAVE0 + wave * wstride;
  // (one call site per phase: k = -1 is the block of n_first steps that precedes the first refresh)
  for (int k = r.sched.n_first > 0 ? -1 : 0; k < r.sched.n_updates; ++k) {
    if (k >= 0) {
      tw_member<P, BIG, run_cell_w<P>(), false>(r.tw, PM_TW_SOLVE | PM_TW_PSIB | PM_TW_PSIBZ, m_raw, wl, lane);
      run_phase_fence();
    }
    const int ns = k < 0 ? r.sched.n_first
                         : ((k == r.sched.n_updates - 1) ? r.sched.n_last : r.sched.m_steps);
    if (ns > 0) {
      tc_member_run<P, VEC>(r.cols, r.wA, r.dt, ns, m_raw, n, r.status, lds, wstride, wave, lane);
      run_phase_fence();
    }
  }
}

// ------------------------------------------------------------------------ Jansen & Nadeau
// (the pm_jn2018 must be the kernel's FIRST argument: jf_member_run re-reads it from there)
// One member from block k0 of the schedule on (k = -1: the n_first steps before the first
// refresh).  skip_diag: block k0's diagnostics are behind the member already (a resumed member).
// s_first: steps of block k0 that are behind the member as well.  Returns the block in which the
// member left this kernel's hands (operands outside the window of the exact-division shortcuts:
// jf_member_run<IEEE = false> stops there and records the steps done), or INT_MAX.
template <int P, bool CT, bool VEC, int BIG, bool IEEE>
__device__ __forceinline__ int jn_run_member(const pm_jn2018 &a, const pm_thermwind &tw,
                                             const pm_psi_so &so, double dt,
                                             const pm_run_schedule &sched, int k0, bool skip_diag,
                                             int s_first, int m_raw, double *lds, int wstride,
                                             int wave, int lane) {
  using L = JfLds<P>;
  double *wl = lds + L::WAVE0 + wave * wstride;
  int s0 = 0;
  for (int k = k0; k < sched.n_updates; ++k) {
    if (k >= 0 && !(skip_diag && k == k0)) {
      so_member<P, false>(so, PM_SO_OP_SOLVE, m_raw, wl, lane);
      run_phase_fence();
      tw_member<P, BIG, run_cell_w<P>(), false>(tw, PM_TW_SOLVE | PM_TW_PSIB | PM_TW_PSIBZ, m_raw, wl, lane);
      run_phase_fence();
    }
    int ns = k < 0 ? sched.n_first : ((k == sched.n_updates - 1) ? sched.n_last : sched.m_steps);
    if (k == k0) ns -= s_first;
    if (ns > 0) {
      const int done = jf_member_run<P, CT, VEC, false, IEEE>(a, dt, ns, s0, m_raw, lds, wstride,
                                                              wave, lane);
      run_phase_fence();
      if (!IEEE && done < ns) return k;
      s0 += ns;
    }
  }
  return 0x7fffffff;
}

// status word of a member that left the main launch: bit 5, the steps done in its last block in
// bits 8..19 (jf_member_run), that block's index + 1 in bits 20..
constexpr int RUN_RESUME_SHIFT = 20;

template <int P, bool CT, bool VEC, int BIG>
__global__ __launch_bounds__(64 * JF_WAVES) JF_OCC_ATTR
void k_jn2018_run(pm_jn2018 a, pm_thermwind tw, pm_psi_so so, double dt, pm_run_schedule sched,
                  int wstride) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m_raw = blockIdx.x * JF_WAVES + wave;
  jf_block_tables<P>(a, dt, lds, wave, lane);
  __syncthreads();
  const int left_at = jn_run_member<P, CT, VEC, BIG, false>(
      a, tw, so, dt, sched, sched.n_first > 0 ? -1 : 0, false, 0, m_raw, lds, wstride, wave, lane);
  if (left_at != 0x7fffffff && lane == 0 && m_raw < a.n)  // (jf_member_run has set bit 5)
    a.ml.status[m_raw] |= (left_at + 1) << RUN_RESUME_SHIFT;
}

// Follow-up launch of k_jn2018_run: the members that left it, resumed at the block they left at
// and carried to the end of the schedule with the step loop in its IEEE form.  One-wave blocks.
template <int P, bool VEC, int BIG>
__global__ __launch_bounds__(64) void k_jn2018_run_ieee(pm_jn2018 a, pm_thermwind tw,
                                                        pm_psi_so so, double dt,
                                                        pm_run_schedule sched, int wstride) {
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
    const int word = a.ml.status[m];
    const int k0 = (word >> RUN_RESUME_SHIFT) - 1, s_first = (word >> 8) & 0xfff;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) a.ml.status[m] = word & 0xff;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    jn_run_member<P, false, VEC, BIG, true>(a, tw, so, dt, sched, k0, true, s_first, m, lds, wstride,
                                            0, lane);
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------ one update
// PsiSO.solve followed by AMOC.solve / Psibz of the same member by the same wave: the two
// diagnostic launches of a Jansen & Nadeau MOC update (run_JansenNadeau_2018.py:206-217) as one
// (one launch boundary and the small Psi_SO launch's drain less; the SO phase of one wave runs
// under the class passes of the others).  `so` without the boundary-value smoother.
template <int P, int BIG>
__global__ __launch_bounds__(64 * TW_WAVES_PER_BLOCK)
__attribute__((amdgpu_waves_per_eu(4))) void k_so_tw_update(pm_psi_so so, pm_thermwind tw,
                                                             int tw_ops, int per_wave) {
  extern __shared__ double lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m_raw = blockIdx.x * (blockDim.x >> 6) + wave;
  double *wl = lds_all + (size_t)wave * per_wave;
  so_member<P, false>(so, PM_SO_OP_SOLVE, m_raw, wl, lane);
  run_phase_fence();
  tw_member<P, BIG>(tw, tw_ops, m_raw, wl, lane);
}

template <int P, int BIG>
static int launch_so_tw_impl(const pm_psi_so &so, const pm_thermwind &tw, int tw_ops, hipStream_t st) {
  int per = tw_lds_doubles(tw.nz, tw.nb);
  const int s = so_lds_doubles(so.nz, so.ny, false, false);
  per = ((s > per ? s : per) + 1) & ~1;
  const size_t per_wave = (size_t)per * sizeof(double);
  int wpb = 1, best = 0;  // as launch_thermwind_impl: the block size that keeps most waves on a CU
  for (int w = TW_WAVES_PER_BLOCK; w >= 1; w >>= 1) {
    int resident = (int)((160 * 1024) / (per_wave * w)) * w;
    resident = resident > 16 ? 16 : resident;
    if (resident > best) {
      best = resident;
      wpb = w;
    }
  }
  const size_t lds = per_wave * wpb;
  if (lds > 160 * 1024) return fail(PM_EINVAL, "pm_so_tw_update needs %zu B of LDS per member", lds);
  if (lds > 64 * 1024)
    PM_HIP(hipFuncSetAttribute((const void *)k_so_tw_update<P, BIG>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((tw.n + wpb - 1) / wpb);
  hipLaunchKernelGGL((k_so_tw_update<P, BIG>), dim3(grid), dim3(64 * wpb), lds, st, so, tw, tw_ops, per);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

// (the lane shapes the stand-alone kernels use for this nz: results are bit-identical to them)
int launch_so_tw_update(const pm_psi_so &so, const pm_thermwind &tw, int tw_ops, hipStream_t st) {
  const int nz = tw.nz, P = (nz + 63) / 64;
  switch (P) {
    case 1: return launch_so_tw_impl<1, 0>(so, tw, tw_ops, st);
    case 2: return launch_so_tw_impl<2, 0>(so, tw, tw_ops, st);
    case 3: return nz - 1 <= 128 ? launch_so_tw_impl<3, 0>(so, tw, tw_ops, st)
                                 : launch_so_tw_impl<3, 1>(so, tw, tw_ops, st);
    case 4: return launch_so_tw_impl<4, 1>(so, tw, tw_ops, st);
  }
  return -1;  // not covered: the caller issues the two launches
}

// ------------------------------------------------------------------------ launchers
size_t run_lds_bytes(int kind, int nz, int nb, int ny);
inline bool run_rows_aligned(const pm_columns &c, const double *wA, int P) {
  auto al = [](const void *q) { return (((unsigned long long)q) & 15ull) == 0ull; };
  return c.nz % P == 0 && al(c.b) && al(c.area) && al(c.kappa) && al(c.dAkappa) && al(wA);
}

template <int P, int BIG>
static int launch_twocol_run_impl(const pm_twocol_loop &r, hipStream_t st) {
  const int wstride = run_wave_stride<P>(r.cols.nz, r.tw.nb, 0, false);
  const size_t lds = (size_t)(JfLds<P>::WAVE0 + JF_WAVES * wstride) * sizeof(double);
  if (lds > 160 * 1024)
    return fail(PM_EINVAL, "pm_twocol_run: %zu B of LDS per block (nz=%d, nb=%d) exceed 160 KB",
                lds, r.cols.nz, r.tw.nb);
  const unsigned grid = (unsigned)((r.tw.n + JF_WAVES - 1) / JF_WAVES);
  const bool vec = run_rows_aligned(r.cols, r.wA, P);
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      PM_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * JF_WAVES), lds, st, r, wstride);
    PM_HIP(hipGetLastError());
    return PM_OK;
  };
  return vec ? go(k_twocol_run<P, true, BIG>) : go(k_twocol_run<P, false, BIG>);
}

int launch_twocol_run(const pm_twocol_loop &r, hipStream_t st) {
  const int nz = r.cols.nz;  // (Psib's pairwise sum is ONE block up to 128 cells: thermwind.hip.h)
  if (run_lds_bytes(0, nz, r.tw.nb, 0) == 0)
    return fail(PM_EINVAL, "pm_twocol_run: nz=%d not supported (65..128 or 193..256)", nz);
  if (nz <= 128) return launch_twocol_run_impl<2, 0>(r, st);
  if (nz - 1 <= 128) return launch_twocol_run_impl<4, 0>(r, st);
  return launch_twocol_run_impl<4, 1>(r, st);
}

template <int P, int BIG>
static int launch_jn2018_run_impl(const pm_jn2018_loop &r, hipStream_t st) {
  const pm_jn2018 &a = r.jn;
  const int wstride = run_wave_stride<P>(a.cols.nz, r.tw.nb, a.ml.ny, true);
  const size_t lds = (size_t)(JfLds<P>::WAVE0 + JF_WAVES * wstride) * sizeof(double);
  if (lds > 160 * 1024)
    return fail(PM_EINVAL, "pm_jn2018_run: %zu B of LDS per block (nz=%d, nb=%d) exceed 160 KB",
                lds, a.cols.nz, r.tw.nb);
  const unsigned grid = (unsigned)((a.n + JF_WAVES - 1) / JF_WAVES);
  auto al = [](const void *q) { return (((unsigned long long)q) & 15ull) == 0ull; };
  const bool vec = run_rows_aligned(a.cols, a.wA, P) && al(a.Psi_SO);
  const bool ct = (a.hints & PM_JN_CONTRACTED) != 0;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      PM_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * JF_WAVES), lds, st, a, r.tw, r.so, r.dt,
                       r.sched, wstride);
    PM_HIP(hipGetLastError());
    return PM_OK;
  };
  if (ct && vec) return go(k_jn2018_run<P, true, true, BIG>);
  if (ct) return go(k_jn2018_run<P, true, false, BIG>);
  const int rc = vec ? go(k_jn2018_run<P, false, true, BIG>) : go(k_jn2018_run<P, false, false, BIG>);
  if (rc != PM_OK) return rc;
  // the IEEE leg of the members that left the launch (operands outside the division window)
  const size_t lds1 = (size_t)(JfLds<P>::WAVE0 + wstride) * sizeof(double);
  const unsigned g1 = (unsigned)((a.n + 63) / 64 < 64 ? (a.n + 63) / 64 : 64);
  if (vec)
    hipLaunchKernelGGL((k_jn2018_run_ieee<P, true, BIG>), dim3(g1), dim3(64), lds1, st, a, r.tw,
                       r.so, r.dt, r.sched, wstride);
  else
    hipLaunchKernelGGL((k_jn2018_run_ieee<P, false, BIG>), dim3(g1), dim3(64), lds1, st, a, r.tw,
                       r.so, r.dt, r.sched, wstride);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

int launch_jn2018_run(const pm_jn2018_loop &r, hipStream_t st) {
  const int nz = r.jn.cols.nz;
  if (run_lds_bytes(1, nz, r.tw.nb, r.jn.ml.ny) == 0)
    return fail(PM_EINVAL, "pm_jn2018_run: nz=%d not supported (65..128 or 193..256)", nz);
  if (nz <= 128) return launch_jn2018_run_impl<2, 0>(r, st);
  if (nz - 1 <= 128) return launch_jn2018_run_impl<4, 0>(r, st);
  return launch_jn2018_run_impl<4, 1>(r, st);
}

size_t run_lds_bytes(int kind, int nz, int nb, int ny) {
  // levels per lane: 2 or 4, AND what the stand-alone diagnostics kernels use for this nz
  // (ceil(nz / 64): the order of the thermal wind's lane-blocked sums is part of its result)
  const int P = (nz + 63) / 64;
  if ((P != 2 && P != 4) || nb < 1 || (kind == 1 && (ny < 3 || ny > 64))) return 0;
  const bool so = kind == 1;
  const size_t d = nz <= 128 ? (size_t)JfLds<2>::WAVE0 + JF_WAVES * run_wave_stride<2>(nz, nb, ny, so)
                             : (size_t)JfLds<4>::WAVE0 + JF_WAVES * run_wave_stride<4>(nz, nb, ny, so);
  return d * sizeof(double);
}

}  // namespace pm
