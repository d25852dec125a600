/*
 * pymoc_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT (see pymoc_oracle.h).
 *
 * Plain C99, scalar, one member at a time -- written for clarity and for agreement
 * with the reference's arithmetic (operation order follows the Python expressions
 * cited at each function), not for speed.  Build: gcc -O2 -ffp-contract=off -fPIC -shared.
 */
#include "pymoc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================================
 * NumPy primitives
 * ========================================================================== */

/* numpy/_core/src/multiarray/compiled_base.c: binary_search_with_guess().
 * Returns j with xp[j] <= key < xp[j+1]; -1 / len outside the range.  The guess
 * logic is kept so that even unsorted xp (np.interp does not check) agree. */
static long interp_search(double key, const double *arr, long len, long guess) {
  long imin = 0, imax = len;
  if (key > arr[len - 1]) return len;
  if (key < arr[0]) return -1;
  if (len <= 4) {
    long i = 1;
    while (i < len && key >= arr[i]) ++i;
    return i - 1;
  }
  if (guess > len - 3) guess = len - 3;
  if (guess < 1) guess = 1;
  if (key < arr[guess]) {
    if (key < arr[guess - 1]) {
      imax = guess - 1;
      if (guess > 8 && key >= arr[guess - 8]) imin = guess - 8;
    } else {
      return guess - 1;
    }
  } else {
    if (key < arr[guess + 1]) return guess;
    if (key < arr[guess + 2]) return guess + 1;
    imin = guess + 2;
    if (guess < len - 8 - 1 && key < arr[guess + 8]) imax = guess + 8;
  }
  while (imin < imax) {
    const long imid = imin + ((imax - imin) >> 1);
    if (key >= arr[imid])
      imin = imid + 1;
    else
      imax = imid;
  }
  return imin - 1;
}

/* np.interp(x, xp, fp) with default left/right (= fp[0], fp[-1]). */
void orc_np_interp(const double *x, int nx, const double *xp, const double *fp, int nxp,
                   double *out) {
  const double lval = fp[0], rval = fp[nxp - 1];
  if (nxp == 1) {
    for (int i = 0; i < nx; ++i) {
      const double xv = x[i];
      out[i] = (xv < xp[0]) ? lval : ((xv > xp[0]) ? rval : fp[0]);
    }
    return;
  }
  long j = 0;
  for (int i = 0; i < nx; ++i) {
    const double xv = x[i];
    if (isnan(xv)) {
      out[i] = xv;
      continue;
    }
    j = interp_search(xv, xp, nxp, j);
    if (j == -1) {
      out[i] = lval;
    } else if (j == nxp) {
      out[i] = rval;
    } else if (j == nxp - 1) {
      out[i] = fp[j];
    } else if (xp[j] == xv) {
      out[i] = fp[j]; /* avoids non-finite interpolation */
    } else {
      const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
      double r = slope * (xv - xp[j]) + fp[j];
      if (isnan(r)) { /* nan in one direction: try the other */
        r = slope * (xv - xp[j + 1]) + fp[j + 1];
        if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
      }
      out[i] = r;
    }
  }
}

/* np.gradient(f, x) for 1-D coordinates, edge_order=1
 * (numpy/lib/_function_base_impl.py: gradient). */
void orc_np_gradient(const double *f, const double *x, int n, double *out) {
  if (n < 2) return;
  int uniform = 1;
  const double d0 = x[1] - x[0];
  for (int i = 1; i < n - 1; ++i)
    if ((x[i + 1] - x[i]) != d0) uniform = 0;
  for (int i = 1; i < n - 1; ++i) {
    if (uniform) {
      out[i] = (f[i + 1] - f[i - 1]) / (2. * d0);
    } else {
      const double dx1 = x[i] - x[i - 1], dx2 = x[i + 1] - x[i];
      const double a = -(dx2) / (dx1 * (dx1 + dx2));
      const double b = (dx2 - dx1) / (dx1 * dx2);
      const double c = dx1 / (dx2 * (dx1 + dx2));
      out[i] = a * f[i - 1] + b * f[i] + c * f[i + 1];
    }
  }
  out[0] = (f[1] - f[0]) / (x[1] - x[0]);
  out[n - 1] = (f[n - 1] - f[n - 2]) / (x[n - 1] - x[n - 2]);
}

/* np.add.reduce on a contiguous double vector: pairwise summation
 * (numpy/_core/src/umath/loops_utils.h.src: DOUBLE_pairwise_sum). */
static double pairwise_sum(const double *a, long n) {
  if (n < 8) {
    double res = 0.;
    for (long i = 0; i < n; ++i) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    long i;
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  } else {
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
  }
}
double orc_np_sum(const double *a, int n) { return pairwise_sum(a, n); }

/* np.linspace(start, stop, num) (numpy/_core/function_base.py). */
void orc_np_linspace(double start, double stop, int num, double *out) {
  if (num <= 0) return;
  const int div = num - 1;
  const double delta = stop - start;
  if (div > 0) {
    const double step = delta / div;
    for (int i = 0; i < num; ++i) {
      if (step == 0.)
        out[i] = ((double)i / div) * delta + start;
      else
        out[i] = (double)i * step + start;
    }
    out[num - 1] = stop;
  } else {
    out[0] = start;
  }
}

/* scipy/optimize/Zeros/brentq.c with the Python defaults xtol=2e-12,
 * rtol=4*eps, maxiter=100, applied to g(y) = np.interp(y, xp, fp) - target. */
static double interp1(double xv, const double *xp, const double *fp, int nxp) {
  double r;
  orc_np_interp(&xv, 1, xp, fp, nxp, &r);
  return r;
}
double orc_brentq_interp(const double *xp, const double *fp, int nxp, double target,
                         double xa, double xb, int *status) {
  const double xtol = 2e-12, rtol = 8.881784197001252e-16;
  const int maxiter = 100;
  double xpre = xa, xcur = xb, xblk = 0., fblk = 0., spre = 0., scur = 0.;
  double fpre = interp1(xpre, xp, fp, nxp) - target;
  double fcur = interp1(xcur, xp, fp, nxp) - target;
  if (status) *status = 0;
  if (fpre == 0) return xpre;
  if (fcur == 0) return xcur;
  if (signbit(fpre) == signbit(fcur)) {
    if (status) *status = -1;
    return 0.;
  }
  for (int it = 0; it < maxiter; ++it) {
    if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
      xblk = xpre;
      fblk = fpre;
      spre = scur = xcur - xpre;
    }
    if (fabs(fblk) < fabs(fcur)) {
      xpre = xcur;
      xcur = xblk;
      xblk = xpre;
      fpre = fcur;
      fcur = fblk;
      fblk = fpre;
    }
    const double delta = (xtol + rtol * fabs(xcur)) / 2;
    const double sbis = (xblk - xcur) / 2;
    if (fcur == 0 || fabs(sbis) < delta) return xcur;
    if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
      double stry;
      if (xpre == xblk) {
        stry = -fcur * (xcur - xpre) / (fcur - fpre); /* secant */
      } else {                                        /* inverse quadratic */
        const double dpre = (fpre - fcur) / (xpre - xcur);
        const double dblk = (fblk - fcur) / (xblk - xcur);
        stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
      }
      const double lim1 = fabs(spre), lim2 = 3 * fabs(sbis) - delta;
      if (2 * fabs(stry) < (lim1 < lim2 ? lim1 : lim2)) {
        spre = scur;
        scur = stry;
      } else {
        spre = sbis;
        scur = sbis;
      }
    } else {
      spre = sbis;
      scur = sbis;
    }
    xpre = xcur;
    fpre = fcur;
    if (fabs(scur) > delta)
      xcur += scur;
    else
      xcur += (sbis > 0 ? delta : -delta);
    fcur = interp1(xcur, xp, fp, nxp) - target;
  }
  if (status) *status = -2;
  return xcur;
}

/* ============================================================================
 * Column   (src/pymoc/modules/column.py)
 * ========================================================================== */

/* Column.convect, column.py:251-271 */
void orc_column_convect(const double *z, double *b, int nz, double bs, double N2min) {
  int any = 0, any_not = 0;
  double zconv = z[0];
  for (int i = 0; i < nz; ++i) {
    if (b[i] > bs) {
      any = 1;
    } else {
      if (!any_not || z[i] > zconv) zconv = z[i]; /* np.max(z[~ind]) */
      any_not = 1;
    }
  }
  if (any) {
    if (!any_not) zconv = z[0];
    for (int i = 0; i < nz; ++i)
      if (b[i] > bs) b[i] = bs + N2min * (z[i] - zconv); /* all from the OLD b */
  } else {
    b[nz - 1] = bs;
  }
}

/* Column.vertadvdiff, column.py:210-249; dAkappa_dz (column.py:96-122) is
 * re-evaluated every call exactly as the reference does. */
void orc_column_vertadvdiff(const double *z, const double *kappa, const double *area,
                            double *b, int nz, const double *wA, double dt, int do_conv,
                            double bs, double bbot, int use_bzbot, double bzbot) {
  double *Ak = (double *)malloc(sizeof(double) * nz * 3);
  double *dAk = Ak + nz, *bz = Ak + 2 * nz;
  for (int i = 0; i < nz; ++i) Ak[i] = area[i] * kappa[i]; /* column.py:94 */
  orc_np_gradient(Ak, z, nz, dAk);                         /* column.py:122 */
  if (!do_conv) b[nz - 1] = bs;                            /* column.py:230-231 */
  b[0] = use_bzbot ? (b[1] - bzbot * (z[1] - z[0])) : bbot; /* column.py:232-233 */
  for (int i = 0; i < nz - 1; ++i) bz[i] = (b[i + 1] - b[i]) / (z[i + 1] - z[i]);
  for (int i = 1; i < nz - 1; ++i) {
    const double dzu = z[i + 1] - z[i], dzd = z[i] - z[i - 1];
    const double bzz = (bz[i] - bz[i - 1]) / (0.5 * (dzu + dzd)); /* column.py:238 */
    const double weff = wA[i] - dAk[i];                           /* column.py:241 */
    const double bzu = (weff < 0) ? bz[i] : bz[i - 1];            /* column.py:242-243 */
    const double db_dt = (-weff) * bzu / area[i] + kappa[i] * bzz; /* :245-248 */
    Ak[i] = b[i] + dt * db_dt; /* staged: every tendency uses the old profile */
  }
  for (int i = 1; i < nz - 1; ++i) b[i] = Ak[i];
  free(Ak);
}

/* Column.horadv, column.py:288-313 */
void orc_column_horadv(const double *area, double *b, int nz, const double *vdx_in,
                       const double *b_in, double dt) {
  for (int i = 0; i < nz; ++i)
    if (vdx_in[i] > 0.0) {
      const double db = b_in[i] - b[i];
      b[i] = b[i] + dt * vdx_in[i] * db / area[i];
    }
}

/* Column.timestep, column.py:315-348 */
void orc_column_timestep(const double *z, const double *kappa, const double *area,
                         double *b, int nz, const double *wA, double dt, int do_conv,
                         double bs, double bbot, int use_bzbot, double bzbot,
                         double N2min, const double *vdx_in, const double *b_in) {
  if (do_conv) orc_column_convect(z, b, nz, bs, N2min);
  orc_column_vertadvdiff(z, kappa, area, b, nz, wA, dt, do_conv, bs, bbot, use_bzbot,
                         bzbot);
  if (vdx_in && b_in) orc_column_horadv(area, b, nz, vdx_in, b_in, dt);
}

void orc_column_ensemble_steps(const double *z, const double *kappa, const double *area,
                               double *b, int ncols, int nz, const double *wA, double dt,
                               const int *do_conv, const double *bs, const double *bbot,
                               const double *N2min, int nsteps) {
  for (int s = 0; s < nsteps; ++s)
    for (int c = 0; c < ncols; ++c) {
      const size_t o = (size_t)c * nz;
      orc_column_timestep(z, kappa + o, area + o, b + o, nz, wA + o, dt, do_conv[c],
                          bs[c], bbot[c], 0, 0.0, N2min[c], NULL, NULL);
    }
}

/* ============================================================================
 * Psi_Thermwind   (src/pymoc/modules/psi_thermwind.py)
 * ========================================================================== */

/* Psi_Thermwind.solve, psi_thermwind.py:125-135.  The reference hands
 *   y0' = y1, y1' = (1/f)(b2(z)-b1(z)),  y0(z[0]) = y0(z[-1]) = 0
 * to scipy.integrate.solve_bvp (4th-order Lobatto IIIA collocation on the mesh z).
 * With np.interp profiles the right-hand side is piecewise linear, the collocation
 * residual vanishes identically on the initial mesh (no refinement), and the
 * collocation equations reduce to the two running integrals below. */
/* b1m / b2m: optional b1, b2 at the interval midpoints z[i] + h/2 (callable profiles are
 * evaluated there by solve_bvp); NULL = the np.interp closure of the level values. */
/* Exclusive running sum s[i] = d[0] + ... + d[i-1], i = 0..nz-1, of n = nz-1 increments in the
 * order the engine's wave uses (thermwind.hip.h, lane_blocked_scan): element i sits in lane
 * i / P, slot i % P, P = ceil(nz / 64); every lane sums its slots left to right (r), the 64 lane
 * totals go through a Hillis-Steele inclusive scan (steps 1, 2, 4, ..., 32: x[l] = x[l-d] + x[l]),
 * and s[i] = e + r_before with e the total of the lanes to the left.  Any summation order
 * restates solve_bvp's collocation solution equally well (the reference agrees to 1e-13 either
 * way); this one needs 6 + P dependent additions instead of nz. */
static void lane_blocked_scan(const double *d, int n, int nz, double *s) {
  const int P = (nz + 63) / 64;
  double x[64], y[64];
  for (int l = 0; l < 64; ++l) {
    double r = 0.;
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      r = r + (i < n ? d[i] : 0.);
    }
    x[l] = r;
  }
  for (int dd = 1; dd < 64; dd <<= 1) {
    for (int l = 0; l < 64; ++l) y[l] = l >= dd ? x[l - dd] + x[l] : x[l];
    for (int l = 0; l < 64; ++l) x[l] = y[l];
  }
  for (int l = 0; l < 64; ++l) {
    const double e = l > 0 ? x[l - 1] : 0.;
    double r = 0.;
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      if (i < nz) s[i] = p == 0 ? e : e + r;
      r = r + (i < n ? d[i] : 0.);
    }
  }
}

void orc_thermwind_solve_mid(const double *z, const double *b1, const double *b2, int nz,
                             double f, const double *b1m_in, const double *b2m_in,
                             double *Psi) {
  const double rf = 1. / f; /* psi_thermwind.py:123: 1. / self.f * (...) */
  double *g = (double *)malloc(sizeof(double) * nz * 4);
  double *G = g + nz, *I = g + 2 * nz, *inc = g + 3 * nz;
  for (int i = 0; i < nz; ++i) g[i] = rf * (b2[i] - b1[i]);
  for (int i = 0; i < nz - 1; ++i) {
    const double h = z[i + 1] - z[i];
    const double zm = z[i] + 0.5 * h;
    /* rhs at the collocation midpoint through the same np.interp closures */
    const double s1 = (b1[i + 1] - b1[i]) / h, s2 = (b2[i + 1] - b2[i]) / h;
    const double b1m = b1m_in ? b1m_in[i] : s1 * (zm - z[i]) + b1[i];
    const double b2m = b2m_in ? b2m_in[i] : s2 * (zm - z[i]) + b2[i];
    const double gm = rf * (b2m - b1m);
    inc[i] = h / 6. * (g[i] + g[i + 1] + 4. * gm); /* G[i+1] - G[i] */
  }
  lane_blocked_scan(inc, nz - 1, nz, G);
  for (int i = 0; i < nz - 1; ++i) {
    const double h = z[i + 1] - z[i];
    const double Gm = 0.5 * (G[i] + G[i + 1]) - 0.125 * h * (g[i + 1] - g[i]);
    inc[i] = h / 6. * (G[i] + G[i + 1] + 4. * Gm); /* I[i+1] - I[i] */
  }
  lane_blocked_scan(inc, nz - 1, nz, I);
  const double span = z[nz - 1] - z[0];
  for (int i = 0; i < nz; ++i)
    Psi[i] = (I[i] - I[nz - 1] * ((z[i] - z[0]) / span)) / 1e6; /* Sv, :135 */
  free(g);
}

void orc_thermwind_solve(const double *z, const double *b1, const double *b2, int nz,
                         double f, double *Psi) {
  orc_thermwind_solve_mid(z, b1, b2, nz, f, NULL, NULL, Psi);
}

static double np_clip01(double v) {
  /* np.clip = minimum(maximum(v, 0), 1); both propagate NaN */
  if (isnan(v)) return v;
  v = v < 0. ? 0. : v;
  return v > 1. ? 1. : v;
}

/* Psi_Thermwind.Psib, psi_thermwind.py:137-185 */
void orc_thermwind_psib(const double *b1, const double *b2, const double *Psi, int nz,
                        int nb, double *bgrid, double *psib) {
  double bmin = b1[0], bmax = b1[0];
  /* min(np.min(b1), np.min(b2)); np.min/np.max propagate NaN */
  int has_nan = 0;
  for (int i = 0; i < nz; ++i) {
    if (isnan(b1[i]) || isnan(b2[i])) has_nan = 1;
    if (b1[i] < bmin) bmin = b1[i];
    if (b2[i] < bmin) bmin = b2[i];
    if (b1[i] > bmax) bmax = b1[i];
    if (b2[i] > bmax) bmax = b2[i];
  }
  if (has_nan) bmin = bmax = NAN;
  orc_np_linspace(bmin, bmax, nb, bgrid); /* :174 */
  const int nc = nz - 1;
  double *u = (double *)malloc(sizeof(double) * nc * 4);
  double *bot = u + nc, *top = u + 2 * nc, *w = u + 3 * nc;
  for (int k = 0; k < nc; ++k) {
    u[k] = -(Psi[k + 1] - Psi[k]); /* :175 */
    if (u[k] < 0) {                /* :179-181 upwind: northern column */
      bot[k] = b2[k];
      top[k] = b2[k + 1];
    } else {
      bot[k] = b1[k];
      top[k] = b1[k + 1];
    }
  }
  for (int i = 0; i < nb; ++i) {
    for (int k = 0; k < nc; ++k)
      w[k] = np_clip01((top[k] - bgrid[i]) / (top[k] - bot[k])) * u[k]; /* :183 */
    psib[i] = pairwise_sum(w, nc);                                      /* :184 */
  }
  free(u);
}

/* Psi_Thermwind.Psibz, psi_thermwind.py:187-208 */
void orc_thermwind_psibz(const double *b1, const double *b2, const double *Psi, int nz,
                         int nb, double *bgrid, double *psib, double *psibz1,
                         double *psibz2) {
  orc_thermwind_psib(b1, b2, Psi, nz, nb, bgrid, psib);
  orc_np_interp(b1, nz, bgrid, psib, nb, psibz1);
  orc_np_interp(b2, nz, bgrid, psib, nb, psibz2);
}

/* ============================================================================
 * Psi_SO   (src/pymoc/modules/psi_SO.py)
 * ========================================================================== */

/* Psi_SO.ys, psi_SO.py:106-140 */
double orc_psi_so_ys(const double *y, const double *bs, int ny, double b, int *status) {
  double bsmin = bs[0];
  int minind = 0;
  for (int j = 1; j < ny; ++j)
    if (bs[j] < bsmin) {
      bsmin = bs[j];
      minind = j;
    }
  if (status) *status = 0;
  if (b < bsmin) return y[0] - 1e3;  /* :128-130 */
  if (b > bs[ny - 1]) return y[ny - 1]; /* :131-133 */
  return orc_brentq_interp(y, bs, ny, b, y[minind], y[ny - 1], status); /* :139-140 */
}

static void bottom_taper(int has, double H, const double *z, int nz, double *out) {
  /* psi_SO.py:164-187: 1. - np.maximum(z[0] + H - z, 0.)**2. / H**2. */
  for (int i = 0; i < nz; ++i) {
    if (!has) {
      out[i] = 1.;
    } else {
      double m = z[0] + H - z[i];
      m = m > 0. ? m : 0.;
      out[i] = 1. - (m * m) / (H * H);
    }
  }
}
static void top_taper(int has, double H, const double *z, int nz, int scalar,
                      double *out) {
  /* psi_SO.py:189-216 */
  for (int i = 0; i < nz; ++i) {
    if (has) {
      double m = z[i] + H;
      m = m > 0 ? m : 0;
      out[i] = 1 - (m * m) / (H * H);
    } else {
      out[i] = 1.;
    }
  }
  if (!has && !scalar) out[nz - 1] = 0.;
}

/* 4th-order (Lobatto IIIA / Simpson, the scheme of scipy.integrate.solve_bvp)
 * collocation for  u'' = q(x) (u - T(x)),  u(x0)=ua, u(xn)=ub  with q = N2/c^2 and
 * N2, T the np.interp closures of psi_SO.py:309-316, on the grid z refined R-fold.
 * Eliminating u' interval by interval leaves a tridiagonal system in u. */
static void gm_bvp(const double *z, int nz, const double *N2, const double *T, double c,
                   double ua, double ub, int R, double *out) {
  const int n = (nz - 1) * R + 1;
  const double c2 = c * c;
  double *x = (double *)malloc(sizeof(double) * (size_t)n * 8);
  double *q = x + n, *r = x + 2 * n, *lo = x + 3 * n, *di = x + 4 * n, *up = x + 5 * n,
         *rh = x + 6 * n, *u = x + 7 * n;
  for (int k = 0; k < nz - 1; ++k) {
    const double h = z[k + 1] - z[k];
    const double sN = (N2[k + 1] - N2[k]) / h, sT = (T[k + 1] - T[k]) / h;
    for (int j = 0; j < R; ++j) {
      const int m = k * R + j;
      const double xv = (j == 0) ? z[k] : z[k] + h * ((double)j / R);
      const double n2 = (j == 0) ? N2[k] : sN * (xv - z[k]) + N2[k];
      const double tv = (j == 0) ? T[k] : sT * (xv - z[k]) + T[k];
      x[m] = xv;
      q[m] = n2 / c2;
      r[m] = q[m] * tv;
    }
  }
  x[n - 1] = z[nz - 1];
  q[n - 1] = N2[nz - 1] / c2;
  r[n - 1] = q[n - 1] * T[nz - 1];
  /* per-interval slope relations: S = u'_k + u'_{k+1}, D = u'_{k+1} - u'_k */
  double *sA = (double *)malloc(sizeof(double) * (size_t)n * 6);
  double *sB = sA + n, *sC = sA + 2 * n, *dA = sA + 3 * n, *dB = sA + 4 * n,
         *dC = sA + 5 * n;
  for (int m = 0; m < n - 1; ++m) {
    const double h = x[m + 1] - x[m];
    const int k = m / R;
    const double xm = x[m] + 0.5 * h;
    const double hz = z[k + 1] - z[k];
    const double n2m = (N2[k + 1] - N2[k]) / hz * (xm - z[k]) + N2[k];
    const double tm = (T[k + 1] - T[k]) / hz * (xm - z[k]) + T[k];
    const double qm = n2m / c2, rm = qm * tm;
    const double al = 1. + h * h * qm / 12.;
    sA[m] = -(2. / h) * (1. + h * h * q[m] / 12.);
    sB[m] = (2. / h) * (1. + h * h * q[m + 1] / 12.);
    sC[m] = -(h / 6.) * (r[m + 1] - r[m]);
    dA[m] = (h / 6.) * (q[m] + 2. * qm) / al;
    dB[m] = (h / 6.) * (q[m + 1] + 2. * qm) / al;
    dC[m] = -(h / 6.) * (r[m] + r[m + 1] + 4. * rm) / al;
  }
  di[0] = 1.;
  up[0] = 0.;
  lo[0] = 0.;
  rh[0] = ua;
  for (int m = 1; m < n - 1; ++m) {
    lo[m] = -(sA[m - 1] + dA[m - 1]);
    di[m] = (sA[m] - dA[m]) - (sB[m - 1] + dB[m - 1]);
    up[m] = sB[m] - dB[m];
    rh[m] = -(sC[m] - dC[m]) + (sC[m - 1] + dC[m - 1]);
  }
  di[n - 1] = 1.;
  lo[n - 1] = 0.;
  up[n - 1] = 0.;
  rh[n - 1] = ub;
  /* Thomas */
  for (int m = 1; m < n; ++m) {
    const double w = lo[m] / di[m - 1];
    di[m] -= w * up[m - 1];
    rh[m] -= w * rh[m - 1];
  }
  u[n - 1] = rh[n - 1] / di[n - 1];
  for (int m = n - 2; m >= 0; --m) u[m] = (rh[m] - up[m] * u[m + 1]) / di[m];
  for (int k = 0; k < nz; ++k) out[k] = u[k * R];
  free(sA);
  free(x);
}

/* The same boundary-value problem on the mesh scipy.integrate.solve_bvp itself ends on
 * (scipy 1.15.3, integrate/_bvp.py:solve_bvp with the defaults the reference uses, tol = 1e-3,
 * max_nodes = 1000; psi_SO.py:319-321).  The ODE is linear, so solve_newton lands on the exact
 * solution of the collocation system of the current mesh (verified against SciPy: 6e-16), and
 * the outer loop is: solve -> rms residual of every interval (estimate_rms_residuals: 5-point
 * Lobatto rule on the C1 cubic spline create_spline(y, f), relative residuals r / (1 + |f|),
 * the mid-point residual being zero for a converged collocation solution) -> insert one node
 * where tol < rms < 100 tol, two where rms >= 100 tol (modify_mesh) -> repeat until no node
 * is added (status 0) or the mesh would exceed max_nodes (status 1, solution of the current
 * mesh returned, like SciPy).  Original levels stay in the mesh, and the spline interpolates
 * its nodes, so res.sol(z) is the nodal solution at the original levels.
 * Returns the number of mesh nodes at the end; *iters = mesh iterations. */
int orc_gm_bvp_adaptive(const double *z, int nz, const double *N2, const double *T, double c,
                        double ua, double ub, double tol, int max_nodes, double *out,
                        int *iters, int *status, double *mesh_out) {
  const double c2 = c * c;
  int m = nz, cap = max_nodes + 2 * nz + 8;
  double *x = (double *)malloc(sizeof(double) * (size_t)cap * 12);
  double *q = x + cap, *r = x + 2 * cap, *u = x + 3 * cap, *up_ = x + 4 * cap,
         *lo = x + 5 * cap, *di = x + 6 * cap, *upd = x + 7 * cap, *rh = x + 8 * cap,
         *rms = x + 9 * cap, *xn = x + 10 * cap, *tq = x + 11 * cap;
  int *seg = (int *)malloc(sizeof(int) * (size_t)cap * 2), *segn = seg + cap;
  double *sA = (double *)malloc(sizeof(double) * (size_t)cap * 6);
  double *sB = sA + cap, *sC = sA + 2 * cap, *dA = sA + 3 * cap, *dB = sA + 4 * cap,
         *dC = sA + 5 * cap;
  for (int k = 0; k < nz; ++k) {
    x[k] = z[k];
    seg[k] = k < nz - 1 ? k : nz - 2; /* original interval a node belongs to */
  }
  int it = 0, st = 0;
  (void)tq;
  for (;;) {
    /* coefficient functions at the nodes: np.interp closures of N2 and T (:309-316) */
    for (int i = 0; i < m; ++i) {
      const int k = seg[i];
      double n2, tv;
      if (x[i] == z[k]) {
        n2 = N2[k];
        tv = T[k];
      } else if (x[i] == z[k + 1]) {
        n2 = N2[k + 1];
        tv = T[k + 1];
      } else {
        const double hz = z[k + 1] - z[k];
        n2 = (N2[k + 1] - N2[k]) / hz * (x[i] - z[k]) + N2[k];
        tv = (T[k + 1] - T[k]) / hz * (x[i] - z[k]) + T[k];
      }
      q[i] = n2 / c2;
      r[i] = q[i] * tv;
    }
    for (int i = 0; i < m - 1; ++i) {
      const double h = x[i + 1] - x[i], xm = x[i] + 0.5 * h;
      const int k = seg[i];
      const double hz = z[k + 1] - z[k];
      const double n2m = (N2[k + 1] - N2[k]) / hz * (xm - z[k]) + N2[k];
      const double tm = (T[k + 1] - T[k]) / hz * (xm - z[k]) + T[k];
      const double qm = n2m / c2, rm = qm * tm;
      const double al = 1. + h * h * qm / 12.;
      sA[i] = -(2. / h) * (1. + h * h * q[i] / 12.);
      sB[i] = (2. / h) * (1. + h * h * q[i + 1] / 12.);
      sC[i] = -(h / 6.) * (r[i + 1] - r[i]);
      dA[i] = (h / 6.) * (q[i] + 2. * qm) / al;
      dB[i] = (h / 6.) * (q[i + 1] + 2. * qm) / al;
      dC[i] = -(h / 6.) * (r[i] + r[i + 1] + 4. * rm) / al;
    }
    di[0] = 1.;
    upd[0] = 0.;
    lo[0] = 0.;
    rh[0] = ua;
    for (int i = 1; i < m - 1; ++i) {
      lo[i] = -(sA[i - 1] + dA[i - 1]);
      di[i] = (sA[i] - dA[i]) - (sB[i - 1] + dB[i - 1]);
      upd[i] = sB[i] - dB[i];
      rh[i] = -(sC[i] - dC[i]) + (sC[i - 1] + dC[i - 1]);
    }
    di[m - 1] = 1.;
    lo[m - 1] = 0.;
    upd[m - 1] = 0.;
    rh[m - 1] = ub;
    for (int i = 1; i < m; ++i) {
      const double w = lo[i] / di[i - 1];
      di[i] -= w * upd[i - 1];
      rh[i] -= w * rh[i - 1];
    }
    u[m - 1] = rh[m - 1] / di[m - 1];
    for (int i = m - 2; i >= 0; --i) u[i] = (rh[i] - upd[i] * u[i + 1]) / di[i];
    /* nodal derivative u' (continuous across nodes by construction) */
    for (int i = 0; i < m - 1; ++i) {
      const double S = sA[i] * u[i] + sB[i] * u[i + 1] + sC[i];
      const double D = dA[i] * u[i] + dB[i] * u[i + 1] + dC[i];
      up_[i] = 0.5 * (S - D);
      if (i == m - 2) up_[m - 1] = 0.5 * (S + D);
    }
    ++it;
    /* estimate_rms_residuals */
    int added = 0;
    const double s37 = sqrt(3. / 7.);
    for (int i = 0; i < m - 1; ++i) {
      const double h = x[i + 1] - x[i];
      const int k = seg[i];
      const double hz = z[k + 1] - z[k];
      /* spline of component 0 (y = u, yp = u') and component 1 (y = u', yp = q u - r) */
      const double y0a = u[i], y0b = u[i + 1], p0a = up_[i], p0b = up_[i + 1];
      const double y1a = up_[i], y1b = up_[i + 1];
      const double p1a = q[i] * u[i] - r[i], p1b = q[i + 1] * u[i + 1] - r[i + 1];
      const double sl0 = (y0b - y0a) / h, t0 = (p0a + p0b - 2 * sl0) / h;
      const double sl1 = (y1b - y1a) / h, t1 = (p1a + p1b - 2 * sl1) / h;
      const double c00 = t0 / h, c01 = (sl0 - p0a) / h - t0, c02 = p0a, c03 = y0a;
      const double c10 = t1 / h, c11 = (sl1 - p1a) / h - t1, c12 = p1a, c13 = y1a;
      const double xmid = x[i] + 0.5 * h, s = 0.5 * h * s37;
      double acc = 0.;
      for (int side = 0; side < 2; ++side) {
        const double xe = side == 0 ? xmid + s : xmid - s;
        const double dx = xe - x[i];
        const double Y0 = ((c00 * dx + c01) * dx + c02) * dx + c03;
        const double Y1 = ((c10 * dx + c11) * dx + c12) * dx + c13;
        const double Y0p = (3 * c00 * dx + 2 * c01) * dx + c02;
        const double Y1p = (3 * c10 * dx + 2 * c11) * dx + c12;
        const double n2e = (N2[k + 1] - N2[k]) / hz * (xe - z[k]) + N2[k];
        const double te = (T[k + 1] - T[k]) / hz * (xe - z[k]) + T[k];
        const double F0 = Y1, F1 = n2e / c2 * (Y0 - te);
        const double r0 = (Y0p - F0) / (1 + fabs(F0)), r1 = (Y1p - F1) / (1 + fabs(F1));
        acc += r0 * r0 + r1 * r1;
      }
      rms[i] = sqrt(0.5 * (49. / 90. * acc));
      if (rms[i] > tol && rms[i] < 100 * tol)
        added += 1;
      else if (rms[i] >= 100 * tol)
        added += 2;
    }
    if (m + added > max_nodes) {
      st = 1;
      break;
    }
    if (added == 0) break;
    /* modify_mesh */
    int mn = 0;
    for (int i = 0; i < m - 1; ++i) {
      xn[mn] = x[i];
      segn[mn++] = seg[i];
      if (rms[i] > tol && rms[i] < 100 * tol) {
        xn[mn] = 0.5 * (x[i] + x[i + 1]);
        segn[mn++] = seg[i];
      } else if (rms[i] >= 100 * tol) {
        xn[mn] = (2 * x[i] + x[i + 1]) / 3;
        segn[mn++] = seg[i];
        xn[mn] = (x[i] + 2 * x[i + 1]) / 3;
        segn[mn++] = seg[i];
      }
    }
    xn[mn] = x[m - 1];
    segn[mn++] = seg[m - 1];
    memcpy(x, xn, sizeof(double) * mn);
    memcpy(seg, segn, sizeof(int) * mn);
    m = mn;
  }
  /* the original levels are mesh nodes: pick them out in order */
  for (int i = 0, k = 0; i < m && k < nz; ++i)
    if (x[i] == z[k]) out[k++] = u[i];
  if (mesh_out) memcpy(mesh_out, x, sizeof(double) * m);
  if (iters) *iters = it;
  if (status) *status = st;
  free(sA);
  free(seg);
  free(x);
  return m;
}

/* Psi_SO.solve, psi_SO.py:333-354 = calc_Ekman (:218-243) + calc_GM (:277-331) */
void orc_psi_so_solve(const double *z, int nz, const double *y, int ny, const double *b,
                      const double *bs, const double *tau, const orc_psi_so_par *par,
                      double *Psi, double *Psi_Ek, double *Psi_GM, int *status) {
  double *ysv = (double *)malloc(sizeof(double) * (size_t)nz * 8);
  double *tau_ave = ysv + nz, *tap1 = ysv + 2 * nz, *tap2 = ysv + 3 * nz,
         *dy = ysv + 4 * nz, *temp = ysv + 5 * nz, *N2 = ysv + 6 * nz,
         *ekraw = ysv + 7 * nz;
  double yl[100], tl[100];
  int st_all = 0;
  /* ---- calc_Ekman */
  for (int i = 0; i < nz; ++i) {
    int st = 0;
    ysv[i] = orc_psi_so_ys(y, bs, ny, b[i], &st); /* :238 (b(z[ii]) == b[ii]) */
    if (st) st_all = st;
    orc_np_linspace(ysv[i], y[ny - 1], 100, yl);
    if (tau)
      orc_np_interp(yl, 100, y, tau, ny, tl);
    else
      for (int k = 0; k < 100; ++k) tl[k] = par->tau_scalar + 0 * yl[k];
    tau_ave[i] = pairwise_sum(tl, 100) / 100.; /* np.mean, :239 */
  }
  bottom_taper(par->has_Hsill, par->Hsill, z, nz, tap1);
  top_taper(par->has_HEk, par->HEk, z, nz, 0, tap2);
  for (int i = 0; i < nz; ++i) {
    ekraw[i] = tau_ave[i] / par->f / par->rho * par->L * tap1[i] * tap2[i]; /* :243 */
    Psi_Ek[i] = ekraw[i] / 1e6;                                            /* :349 */
  }
  /* ---- calc_GM */
  const double eps = 0.1;
  for (int i = 0; i < nz; ++i) {
    const double d = y[ny - 1] - ysv[i];
    dy[i] = d > eps ? d : eps; /* max(d, eps), :305 (NaN d -> d, as Python max) */
    if (isnan(d)) dy[i] = d;
  }
  bottom_taper(par->has_Htaperbot, par->Htaperbot, z, nz, tap1);
  top_taper(par->has_Htapertop, par->Htapertop, z, nz, 1, tap2);
  if (par->has_c) {
    for (int i = 0; i < nz; ++i)
      temp[i] = par->KGM * z[i] / dy[i] * par->L * tap2[i] * tap1[i]; /* :310 */
    /* calc_N2, :142-162 */
    for (int i = 1; i < nz - 1; ++i)
      N2[i] = (b[i + 1] - b[i - 1]) / ((z[i + 1] - z[i]) + (z[i] - z[i - 1]));
    N2[0] = (b[1] - b[0]) / (z[1] - z[0]);
    N2[nz - 1] = (b[nz - 1] - b[nz - 2]) / (z[nz - 1] - z[nz - 2]);
    double ua = 0., ub = 0.;
    if (par->bvp_with_Ek) { /* bc_GM, :270-273 */
      ua = -(Psi_Ek[0] * 1e6);
      ub = -(Psi_Ek[nz - 1] * 1e6);
    }
    if (par->bvp_refine <= 0) { /* default: follow solve_bvp's own adaptive mesh */
      double *sol = (double *)malloc(sizeof(double) * nz);
      orc_gm_bvp_adaptive(z, nz, N2, temp, par->c, ua, ub, 1e-3, 1000, sol, NULL, NULL, NULL);
      memcpy(temp, sol, sizeof(double) * nz);
      free(sol);
    } else {
      gm_bvp(z, nz, N2, temp, par->c, ua, ub, par->bvp_refine, temp);
    }
  } else {
    for (int i = 0; i < nz; ++i) {
      double s = z[i] / dy[i];
      if (!(s >= -par->smax) && !isnan(s)) s = -par->smax; /* np.maximum */
      temp[i] = par->KGM * s * par->L * tap2[i] * tap1[i]; /* :325-327 */
    }
  }
  for (int i = 0; i < nz; ++i) { /* :329-330 */
    if (dy[i] > y[ny - 1] - y[0]) {
      const double lim = -Psi_Ek[i] * 1e6;
      if (isnan(lim) || lim > temp[i]) temp[i] = lim;
    }
  }
  for (int i = 0; i < nz; ++i) {
    Psi_GM[i] = temp[i] / 1e6;        /* :350 */
    Psi[i] = Psi_Ek[i] + Psi_GM[i];   /* :351 */
  }
  Psi[0] = 0.; /* :354 */
  if (status) *status = st_all;
  free(ysv);
}

/* ============================================================================
 * SO_ML   (src/pymoc/modules/SO_ML.py)
 * ========================================================================== */

/* M = U^-1 V of SO_ML.py:136-196 column by column: column j solves U x = V e_j by the
 * Thomas algorithm (V e_j has at most three entries).  M[i*ld + j]. */
static void so_ml_propagator(double *M, int ny, int ld, double s) {
  const double a = -s / 2., bd = 1 + s, c = -s / 2.;
  double *cp = (double *)malloc(sizeof(double) * ny * 2), *den = cp + ny;
  cp[0] = 0.; /* Thomas factors of U: rows 0 and ny-1 identity, interior (-s/2, 1+s, -s/2) */
  for (int i = 1; i < ny - 1; ++i) {
    den[i] = bd - a * cp[i - 1];
    cp[i] = c / den[i];
  }
  for (int j = 0; j < ny; ++j) {
    double dp = 0.;
    for (int i = 0; i < ny; ++i) {
      /* r = V[i][j]; V: rows 0 and ny-1 identity, interior (s/2, 1-s, s/2) */
      double r = 0.;
      if (i == 0 || i == ny - 1)
        r = (i == j) ? 1. : 0.;
      else if (j == i - 1 || j == i + 1)
        r = s / 2.;
      else if (j == i)
        r = 1 - s;
      dp = (i == 0 || i == ny - 1) ? r : (r - a * dp) / den[i];
      M[i * ld + j] = dp;
    }
    double x = M[(ny - 1) * ld + j];
    for (int i = ny - 2; i >= 0; --i) {
      x = M[i * ld + j] - cp[i] * x;
      M[i * ld + j] = x;
    }
  }
  free(cp);
}

/* Parallel cyclic reduction of U x = r (U = tridiag(-s/2, 1+s, -s/2) with identity boundary
 * rows, SO_ML.py:155-165) on 64 rows, rows >= ny being identity rows with r = 0: level l (stride
 * k = 2^l) eliminates x[i-k] and x[i+k] from row i.  Written row-parallel exactly like the HIP
 * kernel (one row per lane, every level reads the previous level's values), same expressions,
 * same fma's: the two are bit-identical.  Multipliers first (they depend only on s, ny): */
#define PCR_LEVELS 6
static void so_ml_pcr_tables(int ny, double s, double alpha[PCR_LEVELS][64],
                             double gamma[PCR_LEVELS][64], double bfin[64]) {
  double a[64], b[64], c[64], an[64], bn[64], cn[64];
  for (int i = 0; i < 64; ++i) {
    const int interior = i >= 1 && i <= ny - 2;
    a[i] = interior ? -s / 2. : 0.;
    b[i] = interior ? 1 + s : 1.;
    c[i] = a[i];
  }
  for (int l = 0; l < PCR_LEVELS; ++l) {
    const int k = 1 << l;
    for (int i = 0; i < 64; ++i) {
      const int lo = i >= k ? i - k : i, hi = i + k <= 63 ? i + k : i;
      const double al = (i >= k) ? -a[i] / b[lo] : 0.;
      const double ga = (i + k <= 63) ? -c[i] / b[hi] : 0.;
      alpha[l][i] = al;
      gamma[l][i] = ga;
      bn[i] = fma(ga, a[hi], fma(al, c[lo], b[i]));
      an[i] = al * a[lo];
      cn[i] = ga * c[hi];
    }
    memcpy(a, an, sizeof(a));
    memcpy(b, bn, sizeof(b));
    memcpy(c, cn, sizeof(c));
  }
  memcpy(bfin, b, sizeof(b));
}

static void so_ml_pcr_solve(double *bs, int ny, double s) {
  double alpha[PCR_LEVELS][64], gamma[PCR_LEVELS][64], bfin[64], r[64], rn[64];
  so_ml_pcr_tables(ny, s, alpha, gamma, bfin);
  const double sh = s / 2.;
  for (int i = 0; i < 64; ++i) {
    if (i >= ny)
      r[i] = 0.;
    else if (i >= 1 && i <= ny - 2)
      r[i] = sh * bs[i - 1] + (1 - s) * bs[i] + sh * bs[i + 1]; /* V bs, interior rows */
    else
      r[i] = bs[i];
  }
  for (int l = 0; l < PCR_LEVELS; ++l) {
    const int k = 1 << l;
    for (int i = 0; i < 64; ++i) {
      const int lo = i >= k ? i - k : i, hi = i + k <= 63 ? i + k : i;
      rn[i] = fma(gamma[l][i], r[hi], fma(alpha[l][i], r[lo], r[i]));
    }
    memcpy(r, rn, sizeof(r));
  }
  for (int i = 0; i < ny; ++i) bs[i] = r[i] / bfin[i];
}

/* calc_implicit_diffusion, SO_ML.py:136-196.
 *   mode 0: Thomas sweep on U x = V bs;
 *   mode 1: invert U by Gauss-Jordan with partial pivoting and form (Uinv V) bs like the
 *           reference's np.dot(np.dot(np.linalg.inv(U), V), bs);
 *   mode 2: the propagator M = U^-1 V built once (so_ml_propagator), then bs <- M bs with
 *           four interleaved fma accumulators -- what the HIP kernel does for ny <= 64 (U, V
 *           depend only on s = Ks dt / dy^2, i.e. they are static between time steps);
 *           (what the HIP kernel did in round 1)
 *   mode 3: parallel cyclic reduction (so_ml_pcr_solve) -- what the HIP kernel does for
 *           ny <= 64 now. */
static void so_ml_implicit_diffusion(double *bs, int ny, double s, int mode) {
  const int dense = mode == 1;
  if (mode == 3) {
    so_ml_pcr_solve(bs, ny, s);
    return;
  }
  double *rhs = (double *)malloc(sizeof(double) * ny);
  if (mode == 2) {
    const int ld = ny | 1;
    double *M = (double *)malloc(sizeof(double) * (size_t)ny * ld);
    so_ml_propagator(M, ny, ld, s);
    for (int i = 0; i < ny; ++i) {
      double acc[4] = {0., 0., 0., 0.};
      for (int j = 0; j < ny; ++j) acc[j & 3] = fma(M[i * ld + j], bs[j], acc[j & 3]);
      rhs[i] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    memcpy(bs, rhs, sizeof(double) * ny);
    free(M);
    free(rhs);
    return;
  }
  if (!dense) {
    double *cp = (double *)malloc(sizeof(double) * ny * 2), *dp = cp + ny;
    rhs[0] = bs[0];
    rhs[ny - 1] = bs[ny - 1];
    for (int i = 1; i < ny - 1; ++i)
      rhs[i] = (s / 2.) * bs[i - 1] + (1 - s) * bs[i] + (s / 2.) * bs[i + 1];
    /* U: rows 0 and ny-1 identity; interior (-s/2, 1+s, -s/2) */
    cp[0] = 0.;
    dp[0] = rhs[0];
    for (int i = 1; i < ny - 1; ++i) {
      const double a = -s / 2., bd = 1 + s, c = -s / 2.;
      const double den = bd - a * cp[i - 1];
      cp[i] = c / den;
      dp[i] = (rhs[i] - a * dp[i - 1]) / den;
    }
    dp[ny - 1] = rhs[ny - 1];
    bs[ny - 1] = dp[ny - 1];
    for (int i = ny - 2; i >= 0; --i) bs[i] = dp[i] - cp[i] * bs[i + 1];
    free(cp);
  } else {
    const int n = ny;
    double *U = (double *)calloc((size_t)n * n * 3, sizeof(double));
    double *V = U + (size_t)n * n, *Ui = U + 2 * (size_t)n * n;
    for (int i = 0; i < n; ++i) {
      U[i * n + i] = 1 + s;
      V[i * n + i] = 1 - s;
      if (i > 0) {
        U[i * n + i - 1] = -s / 2.;
        V[i * n + i - 1] = s / 2.;
      }
      if (i < n - 1) {
        U[i * n + i + 1] = -s / 2.;
        V[i * n + i + 1] = s / 2.;
      }
      Ui[i * n + i] = 1.;
    }
    U[0] = 1;
    U[1] = 0;
    U[(n - 1) * n + n - 2] = 0;
    U[(n - 1) * n + n - 1] = 1;
    V[0] = 1;
    V[1] = 0;
    V[(n - 1) * n + n - 2] = 0;
    V[(n - 1) * n + n - 1] = 1;
    for (int col = 0; col < n; ++col) { /* Gauss-Jordan, partial pivoting */
      int piv = col;
      for (int rr = col + 1; rr < n; ++rr)
        if (fabs(U[rr * n + col]) > fabs(U[piv * n + col])) piv = rr;
      if (piv != col)
        for (int k = 0; k < n; ++k) {
          double t = U[col * n + k];
          U[col * n + k] = U[piv * n + k];
          U[piv * n + k] = t;
          t = Ui[col * n + k];
          Ui[col * n + k] = Ui[piv * n + k];
          Ui[piv * n + k] = t;
        }
      const double d = U[col * n + col];
      for (int k = 0; k < n; ++k) {
        U[col * n + k] /= d;
        Ui[col * n + k] /= d;
      }
      for (int rr = 0; rr < n; ++rr)
        if (rr != col && U[rr * n + col] != 0.) {
          const double fct = U[rr * n + col];
          for (int k = 0; k < n; ++k) {
            U[rr * n + k] -= fct * U[col * n + k];
            Ui[rr * n + k] -= fct * Ui[col * n + k];
          }
        }
    }
    /* M = Ui V (reuse U), then M bs */
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double acc = 0.;
        for (int k = 0; k < n; ++k) acc += Ui[i * n + k] * V[k * n + j];
        U[i * n + j] = acc;
      }
    for (int i = 0; i < n; ++i) {
      double acc = 0.;
      for (int j = 0; j < n; ++j) acc += U[i * n + j] * bs[j];
      rhs[i] = acc;
    }
    memcpy(bs, rhs, sizeof(double) * n);
    free(U);
  }
  free(rhs);
}

/* SO_ML.advdiff, SO_ML.py:198-274 (+ set_boundary_conditions :77-98,
 * calc_advective_tendency :100-134) */
int orc_so_ml_advdiff(const double *y, int ny, const double *surflux,
                      const double *rest_mask, const double *b_rest,
                      const orc_so_ml_par *par, double *bs, double *Psi_s,
                      const double *b_basin, const double *Psi_b, int nz, double dt,
                      int diffusion_mode) {
  double *Psi_mod = (double *)malloc(sizeof(double) * (nz + 2 * ny));
  double *flux = Psi_mod + nz, *adv = flux + ny;
  int ind = -1;
  for (int i = 0; i < nz; ++i)
    if (Psi_b[i] != 0.) { /* np.nonzero(Psi_mod)[0][0], :229 (NaN counts) */
      ind = i;
      break;
    }
  if (ind < 0) {
    free(Psi_mod);
    return -1; /* IndexError in the reference */
  }
  for (int i = 0; i < nz; ++i) Psi_mod[i] = (i < ind) ? Psi_b[ind] : Psi_b[i]; /* :230 */
  orc_np_interp(bs, ny, b_basin, Psi_mod, nz, Psi_s);                          /* :232 */
  int amin = 0; /* np.argmin: first minimum; NaN wins */
  for (int j = 1; j < ny; ++j) {
    if (isnan(bs[amin])) break;
    if (isnan(bs[j]) || bs[j] < bs[amin]) amin = j;
  }
  for (int j = 0; j < amin; ++j) Psi_s[j] = 0.; /* :240 */
  Psi_s[0] = 0.;                                /* :241-243 */
  /* set_boundary_conditions, :93-98 */
  int first_pos = -1;
  for (int i = 0; i < nz; ++i)
    if (Psi_b[i] > 0) {
      first_pos = i;
      break;
    }
  if (Psi_s[1] > 0) {
    if (first_pos < 0) {
      free(Psi_mod);
      return -1;
    }
    bs[0] = b_basin[first_pos];
  } else {
    bs[0] = bs[1];
  }
  for (int j = 0; j < ny; ++j) /* :250-252 */
    flux[j] = surflux[j] / par->h + rest_mask[j] * par->v_pist / par->h * (b_rest[j] - bs[j]);
  const double dy = y[1] - y[0]; /* :255 */
  for (int j = 0; j < ny; ++j) adv[j] = 0.;
  for (int j = 1; j < ny - 1; ++j) { /* :126-133 */
    if (Psi_s[j] < 0.)
      adv[j] = -Psi_s[j] * 1e6 * (bs[j + 1] - bs[j]) / par->h / par->L / dy;
    else if (Psi_s[j] > 0.)
      adv[j] = -Psi_s[j] * 1e6 * (bs[j] - bs[j - 1]) / par->h / par->L / dy;
  }
  for (int j = 0; j < ny; ++j) bs[j] = bs[j] + dt * (flux[j] + adv[j]); /* :259 */
  if (Psi_s[1] <= 0) bs[0] = bs[1];                                     /* :264-266 */
  const double s = par->Ks * dt / (dy * dy);                            /* :191 */
  so_ml_implicit_diffusion(bs, ny, s, diffusion_mode);                   /* :269 */
  if (Psi_s[1] > 0) /* :274 */
    bs[0] = b_basin[first_pos];
  else
    bs[0] = bs[1];
  free(Psi_mod);
  return 0;
}


/* ---- Column.solve_equi ---------------------------------------------------------
 * Reference: column.py:161-164 (ode), :124-159 (bc), :187-208 (solve_equi), which hands the
 * problem to scipy.integrate.solve_bvp (scipy 1.15.3, integrate/_bvp.py; not part of
 * /root/reference).  The ODE is linear and its second component decouples, so the collocation
 * system of _bvp.py:collocation_fun
 *     y[i+1] - y[i] - h/6 (f[i] + f[i+1] + 4 f_mid) = 0,
 *     y_mid = (y[i] + y[i+1])/2 - h/8 (f[i+1] - f[i])
 * has the closed-form solution  v[i+1] = g[i] v[i],  S[i+1] = S[i] + w[i] v[i]  below; the
 * boundary conditions then fix the scale (bbot given) or the offset (bzbot given).  Newton
 * in solve_bvp converges to this in one step. */
void orc_column_equi_pass(const double *x, int m, const double *c_n, const double *c_m,
                          const double *c_1, const double *c_2, double bs, double bbot,
                          int use_bzbot, double bzbot, double *y, double *rms) {
  double *v = y + m, *S = y;
  v[0] = 1.0;
  S[0] = 0.0;
  for (int i = 0; i < m - 1; ++i) {
    const double h = x[i + 1] - x[i];
    const double al = 0.5 + h * c_n[i] / 8.0;     /* weight of v[i]   in v_mid */
    const double be = 0.5 - h * c_n[i + 1] / 8.0; /* weight of v[i+1] in v_mid */
    const double k4 = 4.0 * h / 6.0 * c_m[i];
    const double g = (1.0 + h * c_n[i] / 6.0 + k4 * al) / (1.0 - h * c_n[i + 1] / 6.0 - k4 * be);
    const double w = h / 6.0 * (1.0 + g + 4.0 * (al + be * g));
    v[i + 1] = g * v[i];
    S[i + 1] = S[i] + w * v[i];
  }
  const double Send = S[m - 1];
  if (!use_bzbot) { /* bc: y1(a) = bbot, y1(b) = bs   (column.py:156-157) */
    const double v0 = (bs - bbot) / Send;
    for (int i = 0; i < m; ++i) {
      S[i] = bbot + v0 * S[i];
      v[i] = v0 * v[i];
    }
    S[m - 1] = bs;
  } else { /* bc: y2(a) = bzbot, y1(b) = bs   (column.py:158-159) */
    for (int i = 0; i < m; ++i) {
      S[i] = bs - bzbot * (Send - S[i]);
      v[i] = bzbot * v[i];
    }
  }
  if (!rms) return;
  /* _bvp.py:estimate_rms_residuals with sol = create_spline(y, f, x, h) */
  const double s37 = sqrt(3.0 / 7.0);
  for (int i = 0; i < m - 1; ++i) {
    const double h = x[i + 1] - x[i];
    const double ya[2] = {y[i], v[i]}, yb[2] = {y[i + 1], v[i + 1]};
    const double fa[2] = {v[i], c_n[i] * v[i]}, fb[2] = {v[i + 1], c_n[i + 1] * v[i + 1]};
    double ymid[2], fmid[2], r_mid = 0., r1 = 0., r2 = 0.;
    for (int k = 0; k < 2; ++k) ymid[k] = 0.5 * (yb[k] + ya[k]) - 0.125 * h * (fb[k] - fa[k]);
    fmid[0] = ymid[1];
    fmid[1] = c_m[i] * ymid[1];
    const double xm = x[i] + 0.5 * h;
    const double sh = 0.5 * h * s37;
    const double t1 = (xm + sh) - x[i], t2 = (xm - sh) - x[i];
    double y1[2], y2[2], d1[2], d2[2];
    for (int k = 0; k < 2; ++k) {
      const double slope = (yb[k] - ya[k]) / h;
      const double t = (fa[k] + fb[k] - 2 * slope) / h;
      const double q0 = t / h, q1 = (slope - fa[k]) / h - t, q2 = fa[k], q3 = ya[k];
      /* PPoly evaluation order (scipy/interpolate/_ppoly.pyx:evaluate_poly1) */
      y1[k] = q3 + q2 * t1 + q1 * (t1 * t1) + q0 * (t1 * t1 * t1);
      y2[k] = q3 + q2 * t2 + q1 * (t2 * t2) + q0 * (t2 * t2 * t2);
      d1[k] = q2 + q1 * t1 * 2.0 + q0 * (t1 * t1) * 3.0;
      d2[k] = q2 + q1 * t2 * 2.0 + q0 * (t2 * t2) * 3.0;
      const double col = yb[k] - ya[k] - h / 6 * (fa[k] + fb[k] + 4 * fmid[k]);
      const double rm = 1.5 * col / h / (1 + fabs(fmid[k]));
      r_mid += rm * rm;
    }
    const double f1[2] = {y1[1], c_1[i] * y1[1]}, f2[2] = {y2[1], c_2[i] * y2[1]};
    for (int k = 0; k < 2; ++k) {
      const double a = (d1[k] - f1[k]) / (1 + fabs(f1[k]));
      const double b = (d2[k] - f2[k]) / (1 + fabs(f2[k]));
      r1 += a * a;
      r2 += b * b;
    }
    rms[i] = sqrt(0.5 * (32.0 / 45.0 * r_mid + 49.0 / 90.0 * (r1 + r2)));
  }
}
