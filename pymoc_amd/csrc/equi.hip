// equi.hip -- Column.solve_equi (src/pymoc/modules/column.py:124-208): the equilibrium
// buoyancy profile  b'' = (wA - d(A kappa)/dz) / (A kappa) * b'  as the reference solves it,
// i.e. with scipy.integrate.solve_bvp's 4th-order Lobatto-IIIA collocation and its
// residual-controlled mesh refinement (scipy 1.15.3, integrate/_bvp.py).
//
// The problem is linear and its second component decouples, so on a given mesh the
// collocation system has the closed form  v[i+1] = g[i] v[i],  S[i+1] = S[i] + w[i] v[i]
// (derivation in DESIGN.md section 3, K6): a prefix product and a prefix
// sum.  One wavefront owns one member: every lane composes the affine maps of a contiguous
// chunk of intervals, a 6-step wave scan composes the chunks, a second sweep writes the nodal
// values; the boundary conditions fix scale / offset; the rms residual of every interval
// (solve_bvp's refinement criterion) is evaluated in parallel.  Mesh refinement itself is a
// host decision (pymoc_amd/equilibrium.py), exactly solve_bvp's.
#include "common.hip.h"

namespace pm {

constexpr double SQRT_3_7 = 0.6546536707079771;  // (3/7)**0.5, _bvp.py:estimate_rms_residuals

struct EquiElem {
  double g, w;
};

// coefficients of interval i of the collocation recurrence
__device__ __forceinline__ EquiElem equi_elem(const double *x, const double *cn,
                                              const double *cm, int i) {
  const double h = x[i + 1] - x[i];
  const double al = 0.5 + h * cn[i] / 8.0;      // weight of v[i]   in v_mid
  const double be = 0.5 - h * cn[i + 1] / 8.0;  // weight of v[i+1] in v_mid
  const double k4 = 4.0 * h / 6.0 * cm[i];
  EquiElem e;
  e.g = (1.0 + h * cn[i] / 6.0 + k4 * al) / (1.0 - h * cn[i + 1] / 6.0 - k4 * be);
  e.w = h / 6.0 * (1.0 + e.g + 4.0 * (al + be * e.g));
  return e;
}

__global__ __launch_bounds__(64) void k_column_equi_pass(pm_column_equi a) {
  extern __shared__ double lds[];
  const int mem = blockIdx.x;
  const int lane = threadIdx.x;
  if (a.active && !a.active[mem]) return;
  const int mmax = a.mmax, nz = a.nz;
  const int m = a.m[mem];
  double *x = lds;                 // [mmax] mesh
  double *c[4] = {lds + mmax, lds + 2 * mmax, lds + 3 * mmax, lds + 4 * mmax};
  double *Y = lds + 5 * mmax;      // b on the mesh
  double *V = lds + 6 * mmax;      // db/dz on the mesh
  const double *xg = a.x + (size_t)mem * mmax;
  for (int i = lane; i < m; i += WAVE) x[i] = xg[i];
  __syncthreads();

  // c = (wA - dAkappa_dz) / Akappa (column.py:161-164) on the four point sets of the mesh
  for (int set = 0; set < 4; ++set) {
    const int cnt = set == 0 ? m : m - 1;
    for (int i = lane; i < cnt; i += WAVE) {
      const size_t o = ((size_t)set * a.n + mem) * mmax + i;
      double w;
      if (a.wA) {
        w = a.wA[o];
      } else {  // wA = make_func(array) = np.interp on the column grid (column.py:201)
        double xp = x[i];
        if (set > 0) {
          const double h = x[i + 1] - x[i];
          const double xm = x[i] + 0.5 * h;
          const double s = 0.5 * h * SQRT_3_7;
          xp = set == 1 ? xm : (set == 2 ? xm + s : xm - s);
        }
        w = interp_sorted(xp, a.z, a.wA_z + (size_t)mem * nz, nz);
      }
      c[set][i] = (w - a.dAk[o]) / a.Ak[o];
    }
  }
  __syncthreads();

  // sweep 1: affine map (v, S) -> (G v, S + T v) of this lane's chunk of intervals
  const int ni = m - 1;
  const int K = (ni + WAVE - 1) / WAVE;
  const int i0 = lane * K < ni ? lane * K : ni;
  const int i1 = i0 + K < ni ? i0 + K : ni;
  double G = 1.0, T = 0.0;
  for (int i = i0; i < i1; ++i) {
    const EquiElem e = equi_elem(x, c[0], c[1], i);
    T = T + e.w * G;
    G = e.g * G;
  }
  // inclusive wave scan of the maps (earlier chunk applied first)
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const double Gp = __shfl_up(G, d, WAVE);
    const double Tp = __shfl_up(T, d, WAVE);
    if (lane >= d) {
      T = Tp + T * Gp;
      G = G * Gp;
    }
  }
  double v = __shfl_up(G, 1, WAVE), S = __shfl_up(T, 1, WAVE);
  if (lane == 0) {
    v = 1.0;
    S = 0.0;
  }
  // sweep 2: nodal values for v[0] = 1, S[0] = 0
  for (int i = i0; i < i1; ++i) {
    V[i] = v;
    Y[i] = S;
    const EquiElem e = equi_elem(x, c[0], c[1], i);
    S = S + e.w * v;
    v = e.g * v;
  }
  if (i1 == ni && i0 < i1) {
    V[ni] = v;
    Y[ni] = S;
  }
  __syncthreads();
  const double Send = Y[m - 1];
  __syncthreads();
  const double bs = a.bs[mem];
  const bool use_bzbot = a.flags && (a.flags[mem] & PM_COL_BZBOT) && a.bzbot;
  if (!use_bzbot) {  // bc: b(-H) = bbot, b(0) = bs (column.py:156-157)
    const double bbot = a.bbot[mem];
    const double v0 = (bs - bbot) / Send;
    for (int i = lane; i < m; i += WAVE) {
      Y[i] = i == m - 1 ? bs : bbot + v0 * Y[i];
      V[i] = v0 * V[i];
    }
  } else {  // bc: b'(-H) = bzbot, b(0) = bs (column.py:158-159)
    const double bzbot = a.bzbot[mem];
    for (int i = lane; i < m; i += WAVE) {
      Y[i] = bs - bzbot * (Send - Y[i]);
      V[i] = bzbot * V[i];
    }
  }
  __syncthreads();
  if (a.y) {
    double *yo = a.y + (size_t)mem * 2 * mmax;
    for (int i = lane; i < m; i += WAVE) {
      yo[i] = Y[i];
      yo[mmax + i] = V[i];
    }
  }

  // rms residual of every interval: _bvp.py:estimate_rms_residuals on the C1 cubic spline of
  // _bvp.py:create_spline, evaluated in PPoly's order
  int nadd = 0;
  for (int ib = 0; ib < ni; ib += WAVE) {
    const int i = ib + lane;
    int add = 0;
    if (i < ni) {
      const double h = x[i + 1] - x[i];
      const double ya[2] = {Y[i], V[i]}, yb[2] = {Y[i + 1], V[i + 1]};
      const double fa[2] = {V[i], c[0][i] * V[i]}, fb[2] = {V[i + 1], c[0][i + 1] * V[i + 1]};
      double ymid[2], fmid[2], r_mid = 0., r1 = 0., r2 = 0.;
#pragma unroll
      for (int k = 0; k < 2; ++k) ymid[k] = 0.5 * (yb[k] + ya[k]) - 0.125 * h * (fb[k] - fa[k]);
      fmid[0] = ymid[1];
      fmid[1] = c[1][i] * ymid[1];
      const double xm = x[i] + 0.5 * h;
      const double sh = 0.5 * h * SQRT_3_7;
      const double t1 = (xm + sh) - x[i], t2 = (xm - sh) - x[i];
      double y1[2], y2[2], d1[2], d2[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double slope = (yb[k] - ya[k]) / h;
        const double t = (fa[k] + fb[k] - 2 * slope) / h;
        const double q0 = t / h, q1 = (slope - fa[k]) / h - t, q2 = fa[k], q3 = ya[k];
        y1[k] = q3 + q2 * t1 + q1 * (t1 * t1) + q0 * (t1 * t1 * t1);
        y2[k] = q3 + q2 * t2 + q1 * (t2 * t2) + q0 * (t2 * t2 * t2);
        d1[k] = q2 + q1 * t1 * 2.0 + q0 * (t1 * t1) * 3.0;
        d2[k] = q2 + q1 * t2 * 2.0 + q0 * (t2 * t2) * 3.0;
        const double col = yb[k] - ya[k] - h / 6 * (fa[k] + fb[k] + 4 * fmid[k]);
        const double rm = 1.5 * col / h / (1 + fabs(fmid[k]));
        r_mid += rm * rm;
      }
      const double f1[2] = {y1[1], c[2][i] * y1[1]}, f2[2] = {y2[1], c[3][i] * y2[1]};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double ra = (d1[k] - f1[k]) / (1 + fabs(f1[k]));
        const double rb = (d2[k] - f2[k]) / (1 + fabs(f2[k]));
        r1 += ra * ra;
        r2 += rb * rb;
      }
      const double rms = sqrt(0.5 * (32.0 / 45.0 * r_mid + 49.0 / 90.0 * (r1 + r2)));
      if (a.rms) a.rms[(size_t)mem * mmax + i] = rms;
      // _bvp.py:solve_bvp: one new node where tol < rms < 100 tol, two where rms >= 100 tol
      add = (rms > a.tol && rms < 100 * a.tol) ? 1 : (rms >= 100 * a.tol ? 2 : 0);
    }
    nadd += __popcll(__builtin_amdgcn_ballot_w64(add == 1)) +
            2 * __popcll(__builtin_amdgcn_ballot_w64(add == 2));
  }
  if (lane == 0 && a.nadd) a.nadd[mem] = nadd;

  // res.sol(self.z): the column grid's levels are mesh nodes (solve_bvp only inserts)
  if (a.b && a.zidx) {
    const int32_t *zi = a.zidx + (size_t)mem * nz;
    for (int k = lane; k < nz; k += WAVE) {
      a.b[(size_t)mem * nz + k] = Y[zi[k]];
      if (a.bz) a.bz[(size_t)mem * nz + k] = V[zi[k]];
    }
  }
}

__global__ __launch_bounds__(256) void k_axpby(size_t count, double al, const double *x,
                                               double be, const double *y, double *out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = al * x[i] + be * y[i];
}

}  // namespace pm

using namespace pm;

extern "C" {

int pm_column_equi_pass(const pm_column_equi *eq, pm_stream_t stream) {
  PM_REQUIRE(eq, "eq is NULL");
  const pm_column_equi &a = *eq;
  PM_REQUIRE(a.n >= 0 && a.nz >= 2 && a.mmax >= a.nz && a.mmax <= 1024,
             "bad shape n=%d nz=%d mmax=%d (mesh rows hold nz..1024 nodes)", a.n, a.nz, a.mmax);
  if (a.n == 0) return PM_OK;
  PM_REQUIRE(a.m && a.x && a.Ak && a.dAk && a.bs && a.bbot, "pm_column_equi has a NULL pointer");
  PM_REQUIRE(a.wA || (a.wA_z && a.z), "pm_column_equi needs wA tables or wA_z + z");
  PM_REQUIRE(!a.b || a.zidx, "pm_column_equi.b needs zidx");
  PM_REQUIRE(a.tol > 0., "tol must be positive");
  const size_t lds = (size_t)7 * a.mmax * sizeof(double);
  hipLaunchKernelGGL(k_column_equi_pass, dim3(a.n), dim3(WAVE), lds, resolve_stream(stream), a);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

int pm_axpby(size_t count, double alpha, const double *x, double beta, const double *y,
             double *out, pm_stream_t stream) {
  PM_REQUIRE(x && y && out, "NULL pointer");
  if (count == 0) return PM_OK;
  const unsigned grid = (unsigned)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_axpby, dim3(grid), dim3(256), 0, resolve_stream(stream), count, alpha, x,
                     beta, y, out);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

}  // extern "C"
