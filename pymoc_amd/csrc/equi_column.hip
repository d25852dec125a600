// equi_column.hip -- Equi_Column.solve (src/pymoc/modules/equi_column.py:408-435): the
// non-dimensional equilibrium overturning problem
//     y = (Psi, Psi', Psi'', Psi'''),  y4' = alpha(z,H) y4 (y1 - psi_so(z,H) - A dkappa_dz(z,H)/H^2)
// (ode :349-406, bc :286-347) with the upper-cell depth H either given or an unknown
// parameter, which the reference hands to scipy.integrate.solve_bvp.
//
// k_equi_column_newton is ONE mesh iteration of solve_bvp (scipy 1.15.3 integrate/_bvp.py)
// per member, decision for decision:
//   solve_newton      damped Newton on the collocation system: forward-difference Jacobians
//                     (estimate_fun_jac / estimate_bc_jac), construct_global_jac's blocks,
//                     affine-invariant backtracking line search, at most 8 iterations and
//                     4 Jacobians, the same stopping rule;
//   estimate_rms_residuals  5-point Lobatto estimate on the C1 cubic spline;
//   the node-insertion count of solve_bvp's main loop.
// One wavefront owns one member on its own mesh.  Collocation residuals, Jacobian blocks and
// residual estimates are evaluated in parallel over nodes / intervals.  The linear system is
// solved as SciPy's sparse LU would, by Gaussian elimination with partial pivoting, here on
// the band (rows ordered: 2 conditions at z=-1, 4 collocation rows per interval, 2 conditions
// at z=0: lower / upper bandwidth 5 / 5, fill-in 5) with a 6-row sliding window in LDS; the
// unknown-H column and its extra condition Psi'(-1) = 0 border the band and are eliminated
// by two back-substitutions with the same factors.  Profiles (kappa, dkappa_dz, psi_so) are
// scalars or samples on the model's z grid, interpolated on the device exactly as the
// reference's np.interp closures do; mesh insertion and the spline transfer of the solution
// to a refined mesh are solve_bvp's outer loop (pymoc_amd/equi_column.py).
#include "common.hip.h"

namespace pm {

namespace eqc {

constexpr double SQRT_EPS = 1.4901161193847656e-08;  // EPS**0.5, _bvp.py:estimate_fun_jac
constexpr double SQRT_3_7 = 0.6546536707079771;      // (3/7)**0.5
constexpr int KL = 5, KU = 5, BW = 16;               // band of the bordered system's core
constexpr int NSLOT = 8;                             // LDS window rows (6 live + prefetch)

// scratch arrays of one member, each [4][mmax] unless noted (offsets in units of mmax)
enum : int {
  S_YN = 0,     // trial iterate y_new
  S_YM = 4,     // y_middle
  S_FM = 8,     // f_middle
  S_CR = 12,    // collocation residuals
  S_CO = 16,    // [3] alpha, psi_so, A dkappa/H^2 at the nodes (for the current H)
  S_CM = 19,    // [3] the same at the interval middles
  S_JA = 22,    // [2] d f4/d y1, d f4/d y4 at the nodes
  S_JM = 24,    // [2] the same at the middles
  S_FP = 26,    // [2] d f4/d H at nodes, middles
  S_G = 28,     // [4] right-hand side / solution of the band system
  S_V = 32,     // [4] B^-1 c (unknown-H column)
  S_C = 36,     // [4] c
  S_ST = 40,    // [4] y part of the Newton step
  S_SN = 44,    // [4] y part of the trial step
  S_U = 48,     // [44] U rows: 11 entries per band row
  S_L = 92,     // [20] multipliers: 5 per band row
  S_AB = 112,   // [64] band rows: 16 entries per band row
  S_PIV = 176,  // [4] pivot rows (stored as doubles)
  S_TOTAL = 180
};

struct Member {
  double f, A, bs, bb, kappa;  // bb: non-dimensional b_bot, or B_int
  int flags, nzg;
  const double *zg, *kz, *dkz, *pz;
};

// alpha, psi_so, A*dkappa_dz/H^2 at (z, H): equi_column.py:116-185, :231-249
__device__ __forceinline__ void coef(const Member &q, double z, double H, double &al, double &ps,
                                     double &dk) {
  const double zH = z * H;
  double kap;
  if (q.flags & PM_EQ_KAPPA_ARRAY) {
    kap = interp_sorted(zH, q.zg, q.kz, q.nzg) / (H * H * q.f);
    dk = q.A * (interp_sorted(zH, q.zg, q.dkz, q.nzg) / (H * q.f)) / (H * H);
  } else {
    kap = q.kappa / (H * H * q.f);
    dk = 0.0;
  }
  al = H * H / (q.A * kap);
  ps = (q.flags & PM_EQ_PSI_ARRAY) ? interp_sorted(zH, q.zg, q.pz, q.nzg) / (q.f * (H * H * H))
                                   : 0.0;
}

__device__ __forceinline__ double f4(double al, double ps, double dk, double y1, double y4) {
  return al * y4 * (y1 - ps - dk);  // equi_column.py:402-405
}

// boundary residuals in the reference's order (equi_column.py:330-343):
// HFREE: [ya0, yb0, ya1, cond, yb2 - bs/H]   else: [ya0, yb0, cond, yb2 - bs/H]
__device__ __forceinline__ double bc_cond(const Member &q, double ya2, double ya3, double H) {
  if (q.flags & PM_EQ_HAS_BBOT) return ya2 - q.bb / H;
  double kap;  // bz(H), equi_column.py:251-284: kappa(-1, H)
  if (q.flags & PM_EQ_KAPPA_ARRAY)
    kap = interp_sorted(-1.0 * H, q.zg, q.kz, q.nzg) / (H * H * q.f);
  else
    kap = q.kappa / (H * H * q.f);
  return ya3 + q.bb / (q.f * q.f * q.f * (H * H) * q.A * kap);
}

}  // namespace eqc

using namespace eqc;

__global__ __launch_bounds__(64) void k_equi_column_newton(pm_equi_column a) {
  __shared__ double win[NSLOT][BW];
  const int mem = blockIdx.x;
  const int lane = threadIdx.x;
  if (a.active && !a.active[mem]) return;
  const int mmax = a.mmax;
  const int m = a.m[mem];
  const int ni = m - 1;
  const int N = 4 * m;
  Member q;
  q.f = a.f[mem];
  q.A = a.A[mem];
  q.bs = a.bs[mem];
  q.bb = a.bb[mem];
  q.kappa = a.kappa[mem];
  q.flags = a.flags[mem];
  q.nzg = a.nzg;
  q.zg = a.zg;
  q.kz = a.kappa_z ? a.kappa_z + (size_t)mem * a.nzg : nullptr;
  q.dkz = a.dkappa_z ? a.dkappa_z + (size_t)mem * a.nzg : nullptr;
  q.pz = a.psi_z ? a.psi_z + (size_t)mem * a.nzg : nullptr;
  const bool hfree = (q.flags & PM_EQ_HFREE) != 0;
  const double *x = a.x + (size_t)mem * mmax;
  double *Y = a.y + (size_t)mem * 4 * mmax;   // current iterate
  double *F = a.yp + (size_t)mem * 4 * mmax;  // f at the nodes of the last collocation_fun
  double *S = a.scratch + (size_t)mem * S_TOTAL * mmax;
  auto arr = [&](int off, int c) { return S + (size_t)(off + c) * mmax; };
  const double tol = a.tol, bc_tol = a.tol;

  // ---- collocation_fun(y, H) (_bvp.py:276-314) + boundary residuals
  double bcr[5];  // HFREE: ya0, yb0, ya1, cond, yb2-bs/H   (index 2 unused otherwise)
  auto col_fun = [&](const double *y, double H) {
    for (int i = lane; i < m; i += WAVE) {
      double al, ps, dk;
      coef(q, x[i], H, al, ps, dk);
      arr(S_CO, 0)[i] = al;
      arr(S_CO, 1)[i] = ps;
      arr(S_CO, 2)[i] = dk;
      F[i] = y[mmax + i];
      F[mmax + i] = y[2 * mmax + i];
      F[2 * mmax + i] = y[3 * mmax + i];
      F[3 * mmax + i] = f4(al, ps, dk, y[i], y[3 * mmax + i]);
    }
    __syncthreads();
    for (int i = lane; i < ni; i += WAVE) {
      const double h = x[i + 1] - x[i];
      double ym[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ym[c] = 0.5 * (y[c * mmax + i + 1] + y[c * mmax + i]) -
                0.125 * h * (F[c * mmax + i + 1] - F[c * mmax + i]);
        arr(S_YM, c)[i] = ym[c];
      }
      double al, ps, dk;
      coef(q, x[i] + 0.5 * h, H, al, ps, dk);
      arr(S_CM, 0)[i] = al;
      arr(S_CM, 1)[i] = ps;
      arr(S_CM, 2)[i] = dk;
      const double fm[4] = {ym[1], ym[2], ym[3], f4(al, ps, dk, ym[0], ym[3])};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        arr(S_FM, c)[i] = fm[c];
        arr(S_CR, c)[i] = y[c * mmax + i + 1] - y[c * mmax + i] -
                          h / 6 * (F[c * mmax + i] + F[c * mmax + i + 1] + 4 * fm[c]);
      }
    }
    bcr[0] = y[0];
    bcr[1] = y[m - 1];
    bcr[2] = hfree ? y[mmax] : 0.0;
    bcr[3] = bc_cond(q, y[2 * mmax], y[3 * mmax], H);
    bcr[4] = y[2 * mmax + m - 1] - q.bs / H;
    __syncthreads();
  };

  // ---- Jacobian: forward differences (_bvp.py:15-115), blocks of construct_global_jac
  // (:158-273), assembled into band rows; then LU with partial pivoting
  bool singular = false;
  auto jac_and_factor = [&](const double *y, double H) {
    const double hp = SQRT_EPS * (1 + fabs(H));
    const double Hn = H + hp;
    const double hip = Hn - H;
    for (int i = lane; i < m; i += WAVE) {  // nodes
      const double al = arr(S_CO, 0)[i], ps = arr(S_CO, 1)[i], dk = arr(S_CO, 2)[i];
      const double y1 = y[i], y4 = y[3 * mmax + i], f0 = F[3 * mmax + i];
      const double h1 = SQRT_EPS * (1 + fabs(y1)), y1n = y1 + h1;
      const double h4 = SQRT_EPS * (1 + fabs(y4)), y4n = y4 + h4;
      arr(S_JA, 0)[i] = (f4(al, ps, dk, y1n, y4) - f0) / (y1n - y1);
      arr(S_JA, 1)[i] = (f4(al, ps, dk, y1, y4n) - f0) / (y4n - y4);
      if (hfree) {
        double al2, ps2, dk2;
        coef(q, x[i], Hn, al2, ps2, dk2);
        arr(S_FP, 0)[i] = (f4(al2, ps2, dk2, y1, y4) - f0) / hip;
      }
    }
    for (int i = lane; i < ni; i += WAVE) {  // middles
      const double al = arr(S_CM, 0)[i], ps = arr(S_CM, 1)[i], dk = arr(S_CM, 2)[i];
      const double y1 = arr(S_YM, 0)[i], y4 = arr(S_YM, 3)[i], f0 = arr(S_FM, 3)[i];
      const double h1 = SQRT_EPS * (1 + fabs(y1)), y1n = y1 + h1;
      const double h4 = SQRT_EPS * (1 + fabs(y4)), y4n = y4 + h4;
      arr(S_JM, 0)[i] = (f4(al, ps, dk, y1n, y4) - f0) / (y1n - y1);
      arr(S_JM, 1)[i] = (f4(al, ps, dk, y1, y4n) - f0) / (y4n - y4);
      if (hfree) {
        const double h = x[i + 1] - x[i];
        double al2, ps2, dk2;
        coef(q, x[i] + 0.5 * h, Hn, al2, ps2, dk2);
        arr(S_FP, 1)[i] = (f4(al2, ps2, dk2, y1, y4) - f0) / hip;
      }
    }
    __syncthreads();
    double *AB = arr(S_AB, 0);  // [N][16]: entry (r, c) at r*16 + c - r + KL
    double *Cc = arr(S_C, 0);   // [N]
    for (int i = lane; i < ni; i += WAVE) {
      const double h = x[i + 1] - x[i];
      const double a0 = arr(S_JA, 0)[i], d0 = arr(S_JA, 1)[i];
      const double a1 = arr(S_JA, 0)[i + 1], d1 = arr(S_JA, 1)[i + 1];
      const double am = arr(S_JM, 0)[i], dm = arr(S_JM, 1)[i];
      // df_dy = [[0,1,0,0],[0,0,1,0],[0,0,0,1],[a,0,0,d]]
      const double J0[4][4] = {{0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {a0, 0, 0, d0}};
      const double J1[4][4] = {{0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {a1, 0, 0, d1}};
      const double Jm[4][4] = {{0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {am, 0, 0, dm}};
      // T0 = Jm J0, T1 = Jm J1 (zero products dropped: they are exact zeros)
      const double T0[4][4] = {{0, 0, 1, 0}, {0, 0, 0, 1}, {a0, 0, 0, d0}, {dm * a0, am, 0, dm * d0}};
      const double T1[4][4] = {{0, 0, 1, 0}, {0, 0, 0, 1}, {a1, 0, 0, d1}, {dm * a1, am, 0, dm * d1}};
      const double h6 = h / 6, hh12 = h * h / 12;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 2 + 4 * i + e;
        double *row = AB + (size_t)r * BW;
#pragma unroll
        for (int t = 0; t < BW; ++t) row[t] = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v0 = (e == c) ? -1.0 : -0.0;
          v0 -= h6 * (J0[e][c] + 2 * Jm[e][c]);
          v0 -= hh12 * T0[e][c];
          double v1 = (e == c) ? 1.0 : 0.0;
          v1 -= h6 * (J1[e][c] + 2 * Jm[e][c]);
          v1 += hh12 * T1[e][c];
          row[4 * i + c - r + KL] = v0;
          row[4 * i + 4 + c - r + KL] = v1;
        }
      }
      if (hfree) {
        const double p0 = arr(S_FP, 0)[i], p1 = arr(S_FP, 0)[i + 1], pm = arr(S_FP, 1)[i];
        const double dp = p0 - p1;
        const double m2 = 0.0 + 0.125 * h * dp;        // row 3 of df_dp_middle
        const double m3 = pm + 0.125 * h * (dm * dp);  // row 4
        Cc[2 + 4 * i + 0] = -h / 6 * (0.0 + 0.0 + 4 * 0.0);
        Cc[2 + 4 * i + 1] = -h / 6 * (0.0 + 0.0 + 4 * 0.0);
        Cc[2 + 4 * i + 2] = -h / 6 * (0.0 + 0.0 + 4 * m2);
        Cc[2 + 4 * i + 3] = -h / 6 * (p0 + p1 + 4 * m3);
      }
    }
    if (lane == 0) {  // boundary rows by forward differences (_bvp.py:58-115)
      const double ya2 = y[2 * mmax], ya3 = y[3 * mmax], yb2 = y[2 * mmax + m - 1];
      auto fd = [&](double v) { return SQRT_EPS * (1 + fabs(v)); };
      double *r0 = AB, *r1 = AB + BW, *r2 = AB + (size_t)(N - 2) * BW, *r3 = AB + (size_t)(N - 1) * BW;
      for (int t = 0; t < BW; ++t) r0[t] = r1[t] = r2[t] = r3[t] = 0.0;
      // row 0: ya0 -> d/dya0 = ((ya0+h) - ya0)/hi = 1 exactly
      r0[0 - 0 + KL] = 1.0;
      // row 1: cond(ya2, ya3, H)
      {
        const double c0 = bcr[3];
        const double n2 = ya2 + fd(ya2), n3 = ya3 + fd(ya3);
        r1[2 - 1 + KL] = (bc_cond(q, n2, ya3, H) - c0) / (n2 - ya2);
        r1[3 - 1 + KL] = (bc_cond(q, ya2, n3, H) - c0) / (n3 - ya3);
        Cc[1] = hfree ? (bc_cond(q, ya2, ya3, Hn) - c0) / hip : 0.0;
      }
      Cc[0] = 0.0;
      // row N-2: yb0
      r2[(N - 4) - (N - 2) + KL] = 1.0;
      // row N-1: yb2 - bs/H
      {
        const double c0 = bcr[4];
        const double n2 = yb2 + fd(yb2);
        r3[(N - 2) - (N - 1) + KL] = ((n2 - q.bs / H) - c0) / (n2 - yb2);
        Cc[N - 1] = hfree ? ((yb2 - q.bs / Hn) - c0) / hip : 0.0;
      }
      Cc[N - 2] = 0.0;
    }
    __syncthreads();

    // ---- band LU with partial pivoting, 6-row window in LDS
    double *U = arr(S_U, 0), *L = arr(S_L, 0), *PV = arr(S_PIV, 0);
    for (int r = 0; r < 6 && r < N; ++r)
      if (lane < BW) win[r % NSLOT][lane] = AB[(size_t)r * BW + lane];
    __syncthreads();
    int sing = 0;
    for (int j = 0; j < N; ++j) {
      const int nr = (N - j) < 6 ? (N - j) : 6;
      double v = (lane < nr) ? fabs(win[(j + lane) % NSLOT][KL - lane]) : -1.0;
      int pr = lane < nr ? lane : 0;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        const double ov = __shfl_xor(v, o, WAVE);
        const int op = __shfl_xor(pr, o, WAVE);
        if (ov > v || (ov == v && op < pr)) {
          v = ov;
          pr = op;
        }
      }
      pr = __shfl(pr, 0, WAVE);
      if (pr > 0 && lane < 11) {  // swap rows j, j+pr over columns j .. j+10
        double &p0 = win[j % NSLOT][KL + lane];
        double &p1 = win[(j + pr) % NSLOT][KL + lane - pr];
        const double t0 = p0;
        p0 = p1;
        p1 = t0;
      }
      __syncthreads();
      const double piv = win[j % NSLOT][KL];
      if (piv == 0.0) sing = 1;
      if (lane < 50) {
        const int rr = 1 + lane / 10, cc = 1 + lane % 10;
        if (rr < nr) {
          double *row = win[(j + rr) % NSLOT];
          const double l = row[KL - rr] / piv;
          row[KL - rr + cc] -= l * win[j % NSLOT][KL + cc];
          if (cc == 1) L[(size_t)j * 5 + rr - 1] = l;
        } else if (cc == 1) {
          L[(size_t)j * 5 + rr - 1] = 0.0;
        }
      }
      if (lane < 11) U[(size_t)j * 11 + lane] = win[j % NSLOT][KL + lane];
      if (lane == 0) PV[j] = (double)(j + pr);
      __syncthreads();
      if (j + 6 < N && lane < BW) win[(j + 6) % NSLOT][lane] = AB[(size_t)(j + 6) * BW + lane];
      __syncthreads();
    }
    singular = sing != 0;
  };

  // ---- B g = rhs in place (g in global scratch), lane 0 sweeps with register windows
  auto band_solve = [&](double *g) {
    if (lane == 0) {
      const double *U = arr(S_U, 0), *L = arr(S_L, 0), *PV = arr(S_PIV, 0);
      double w[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) w[t] = t < N ? g[t] : 0.0;
      for (int j = 0; j < N; ++j) {  // forward: P, L
        const int pr = (int)PV[j] - j;
        double gj = w[0];
#pragma unroll
        for (int t = 1; t < 6; ++t)
          if (pr == t) {
            gj = w[t];
            w[t] = w[0];
          }
        g[j] = gj;
#pragma unroll
        for (int t = 0; t < 5; ++t) w[t] = w[t + 1] - L[(size_t)j * 5 + t] * gj;
        w[5] = (j + 6 < N) ? g[j + 6] : 0.0;
      }
      double u[10];
#pragma unroll
      for (int t = 0; t < 10; ++t) u[t] = 0.0;
      for (int j = N - 1; j >= 0; --j) {  // backward: U
        const double *ur = U + (size_t)j * 11;
        double s0 = g[j], s1 = 0.0;
#pragma unroll
        for (int t = 0; t < 10; t += 2) {
          s0 -= ur[1 + t] * u[t];
          s1 -= ur[2 + t] * u[t + 1];
        }
        const double xj = (s0 + s1) / ur[0];
        g[j] = xj;
#pragma unroll
        for (int t = 9; t > 0; --t) u[t] = u[t - 1];
        u[0] = xj;
      }
    }
    __syncthreads();
  };

  // ---- step = J^-1 res for the current residual arrays; returns |step|^2
  auto newton_step = [&](double *ystep, double &pstep) -> double {
    double *g = arr(S_G, 0);
    for (int i = lane; i < ni; i += WAVE)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[2 + 4 * i + e] = arr(S_CR, e)[i];
    if (lane == 0) {
      g[0] = bcr[0];
      g[1] = bcr[3];
      g[N - 2] = bcr[1];
      g[N - 1] = bcr[4];
    }
    __syncthreads();
    band_solve(g);
    double ps = 0.0;
    if (hfree) {  // border row: dy[1] = ya1 residual (its dH entry is exactly 0)
      const double *vv = arr(S_V, 0);
      ps = (g[1] - bcr[2]) / vv[1];
    }
    double acc = 0.0;
    for (int i = lane; i < N; i += WAVE) {
      const double s = hfree ? g[i] - arr(S_V, 0)[i] * ps : g[i];
      ystep[(size_t)(i & 3) * mmax + (i >> 2)] = s;
      acc += s * s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
    pstep = ps;
    __syncthreads();
    return acc + ps * ps;
  };

  // ================================================================ solve_newton
  double H = a.p[mem];
  double *YN = arr(S_YN, 0), *ST = arr(S_ST, 0), *SN = arr(S_SN, 0);
  col_fun(Y, H);
  int njev = 0, iters = 0;
  bool recompute = true;
  double pstep = 0.0, cost = 0.0;
  for (int it = 0; it < 8; ++it) {
    ++iters;
    if (recompute) {
      jac_and_factor(Y, H);
      ++njev;
      if (singular) break;
      if (hfree) {
        double *vv = arr(S_V, 0);
        for (int i = lane; i < N; i += WAVE) vv[i] = arr(S_C, 0)[i];
        __syncthreads();
        band_solve(vv);
      }
      cost = newton_step(ST, pstep);
    }
    double alpha = 1.0, pstep_new = 0.0, cost_new = 0.0, Hnew = H;
    for (int trial = 0; trial < 5; ++trial) {
      for (int i = lane; i < m; i += WAVE)
#pragma unroll
        for (int c = 0; c < 4; ++c) YN[c * mmax + i] = Y[c * mmax + i] - alpha * ST[c * mmax + i];
      Hnew = hfree ? H - alpha * pstep : H;
      __syncthreads();
      col_fun(YN, Hnew);
      cost_new = newton_step(SN, pstep_new);
      if (cost_new < (1 - 2 * alpha * 0.2) * cost) break;
      if (trial < 4) alpha *= 0.5;
    }
    for (int i = lane; i < m; i += WAVE)
#pragma unroll
      for (int c = 0; c < 4; ++c) Y[c * mmax + i] = YN[c * mmax + i];
    H = Hnew;
    __syncthreads();
    if (njev == 4) break;
    // converged?  |col_res| < tol_r (1 + |f_middle|) everywhere and |bc| < bc_tol
    bool ok = true;
    for (int i = lane; i < ni; i += WAVE) {
      const double tol_r = 2.0 / 3.0 * (x[i + 1] - x[i]) * 5e-2 * tol;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        ok = ok && (fabs(arr(S_CR, c)[i]) < tol_r * (1 + fabs(arr(S_FM, c)[i])));
    }
    ok = ok && fabs(bcr[0]) < bc_tol && fabs(bcr[1]) < bc_tol && fabs(bcr[3]) < bc_tol &&
         fabs(bcr[4]) < bc_tol && (!hfree || fabs(bcr[2]) < bc_tol);
    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
    if (alpha == 1.0) {
      for (int i = lane; i < m; i += WAVE)
#pragma unroll
        for (int c = 0; c < 4; ++c) ST[c * mmax + i] = SN[c * mmax + i];
      pstep = pstep_new;
      cost = cost_new;
      recompute = false;
      __syncthreads();
    } else {
      recompute = true;
    }
  }

  // ================================================= residual estimate (_bvp.py:525-573)
  // (F, y_middle, f_middle, col_res are those of the final iterate)
  int nadd = 0;
  double rmax = 0.0;
  for (int ib = 0; ib < ni; ib += WAVE) {
    const int i = ib + lane;
    int add = 0;
    if (i < ni) {
      const double h = x[i + 1] - x[i];
      const double xm = x[i] + 0.5 * h;
      const double sh = 0.5 * h * SQRT_3_7;
      const double xs[2] = {xm + sh, xm - sh};
      double rsum[2] = {0., 0.}, r_mid = 0.;
      double yv[2][4], dv[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double ya = Y[c * mmax + i], yb = Y[c * mmax + i + 1];
        const double fa = F[c * mmax + i], fb = F[c * mmax + i + 1];
        const double slope = (yb - ya) / h;
        const double t = (fa + fb - 2 * slope) / h;
        const double q0 = t / h, q1 = (slope - fa) / h - t, q2 = fa, q3 = ya;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const double d = xs[s] - x[i];
          yv[s][c] = q3 + q2 * d + q1 * (d * d) + q0 * (d * d * d);
          dv[s][c] = q2 + q1 * d * 2.0 + q0 * (d * d) * 3.0;
        }
        const double rm = 1.5 * arr(S_CR, c)[i] / h / (1 + fabs(arr(S_FM, c)[i]));
        r_mid += rm * rm;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        double al, ps, dk;
        coef(q, xs[s], H, al, ps, dk);
        const double fs[4] = {yv[s][1], yv[s][2], yv[s][3], f4(al, ps, dk, yv[s][0], yv[s][3])};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double r = (dv[s][c] - fs[c]) / (1 + fabs(fs[c]));
          rsum[s] += r * r;
        }
      }
      const double rms = sqrt(0.5 * (32.0 / 45.0 * r_mid + 49.0 / 90.0 * (rsum[0] + rsum[1])));
      if (a.rms) a.rms[(size_t)mem * mmax + i] = rms;
      add = (rms > tol && rms < 100 * tol) ? 1 : (rms >= 100 * tol ? 2 : 0);
      rmax = rms > rmax ? rms : rmax;
    }
    nadd += __popcll(__builtin_amdgcn_ballot_w64(add == 1)) +
            2 * __popcll(__builtin_amdgcn_ballot_w64(add == 2));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(rmax, o, WAVE);
    rmax = ov > rmax ? ov : rmax;
  }
  if (lane == 0) {
    a.p[mem] = H;
    if (a.nadd) a.nadd[mem] = nadd;
    if (a.status) a.status[mem] = singular ? 2 : 0;
    if (a.niter) a.niter[mem] = iters;
    if (a.info) {
      double bmax = fabs(bcr[0]);
      bmax = fmax(bmax, fabs(bcr[1]));
      bmax = fmax(bmax, fabs(bcr[3]));
      bmax = fmax(bmax, fabs(bcr[4]));
      if (hfree) bmax = fmax(bmax, fabs(bcr[2]));
      a.info[(size_t)mem * 2] = rmax;      // max rms residual
      a.info[(size_t)mem * 2 + 1] = bmax;  // max |bc residual|
    }
  }
}

}  // namespace pm

using namespace pm;

extern "C" {

size_t pm_equi_column_scratch_doubles(int32_t mmax) { return (size_t)eqc::S_TOTAL * (size_t)mmax; }

int pm_equi_column_newton(const pm_equi_column *eq, pm_stream_t stream) {
  PM_REQUIRE(eq, "eq is NULL");
  const pm_equi_column &a = *eq;
  PM_REQUIRE(a.n >= 0 && a.mmax >= 3 && a.mmax <= 4096, "bad shape n=%d mmax=%d", a.n, a.mmax);
  if (a.n == 0) return PM_OK;
  PM_REQUIRE(a.m && a.x && a.y && a.yp && a.p && a.f && a.A && a.bs && a.bb && a.kappa && a.flags &&
                 a.scratch,
             "pm_equi_column has a NULL pointer");
  PM_REQUIRE(a.nzg == 0 || a.zg, "profile grid zg is NULL");
  PM_REQUIRE(a.tol > 0., "tol must be positive");
  hipLaunchKernelGGL(k_equi_column_newton, dim3(a.n), dim3(WAVE), 0, resolve_stream(stream), a);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

}  // extern "C"
