// pymoc_hip.hip -- the single translation unit of libpymoc_hip.so: runtime plumbing
// (memory, streams, events, graphs) and the extern "C" launchers of every kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <stdarg.h>
#include <math.h>
#include <vector>
#include <dlfcn.h>
#include "common.hip.h"
#include "column.hip.h"
#include "thermwind.hip.h"
#include "psi_so.hip.h"
#include "so_ml.hip.h"
#include "comm.hip.h"
#ifdef PM_PHASE_PROFILE
#define PM_JF_IN_MAIN_TU
#include "jn2018_fast.hip"
#endif

namespace pm {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static hipStream_t g_default_stream = nullptr;

hipStream_t resolve_stream(pm_stream_t s) {
  if (s) return (hipStream_t)s;
  if (!g_default_stream) {
    if (hipStreamCreateWithFlags(&g_default_stream, hipStreamNonBlocking) != hipSuccess)
      g_default_stream = nullptr;  // fall back to the null stream
  }
  return g_default_stream;
}

// ---- lane shift self test -------------------------------------------------------
__global__ void k_selftest_lane_shift(int *mismatch) {
  const int lane = threadIdx.x & 63;
  const double x = 1000.0 * blockIdx.x + lane + 0.25;
  const double n_dpp = from_next_lane(x);
  const double p_dpp = from_prev_lane(x);
  const double n_ref = __shfl_down(x, 1, 64);
  const double p_ref = __shfl_up(x, 1, 64);
  int bad = 0;
  if (lane < 63 && n_dpp != n_ref) bad = 1;
  if (lane > 0 && p_dpp != p_ref) bad = 1;
  if (lane == 63 && n_dpp != x) bad = 1;  // no source: keeps own value
  if (lane == 0 && p_dpp != x) bad = 1;
  const double n_z = from_next_lane_z(x), p_z = from_prev_lane_z(x);
  if (lane < 63 && n_z != n_ref) bad = 1;
  if (lane > 0 && p_z != p_ref) bad = 1;
  if (lane == 63 && n_z != 0.0) bad = 1;  // no source: zero
  if (lane == 0 && p_z != 0.0) bad = 1;
  if (bad) atomicAdd(mismatch, 1);
}

// ---- wave scans of the GM boundary-value solve (psi_so.hip.h) against serial composition ----
// One wave per case: `nhas` lanes carry a random diagonally dominant element (the others are
// empty, as past the end of a mesh).  out[0..2] = largest relative deviation of the DPP prefix
// scan, suffix scan and affine suffix scan from the same composition done serially, lane by
// lane, through LDS (the association differs: 1e-13, not bits); out_int += mismatches of the
// integer prefix sum (exact).
__global__ void k_selftest_so_scans(int nhas, unsigned long long seed, double *out, int *out_int) {
  __shared__ SoElem el[64];
  __shared__ double aff[128];
  const int lane = threadIdx.x & 63;
  unsigned long long x = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull * (unsigned long long)(lane + 1);
  auto rnd = [&]() {  // uniform in [0, 1)
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    return (double)((x * 0x2545F4914F6CDD1Dull) >> 11) * 0x1p-53;
  };
  const bool has = lane < nhas;
  SoElem e;  // shaped like so_sub_element's: a11, a22 ~ -2/h, a12, a21 ~ +2/h, dominant diagonal
  const double h = 0.01 + rnd(), q = rnd();
  e.a11 = -(2. / h + q * h / 3.);
  e.a12 = 2. / h - q * h / 6.;
  e.a21 = 2. / h - q * h / 6.;
  e.a22 = -(2. / h + q * h / 3.);
  e.c1 = rnd() - 0.5;
  e.c2 = rnd() - 0.5;
  el[lane] = e;
  const double A = rnd() - 0.5, B = (lane == nhas - 1 || !has) ? 0. : 0.9 * (rnd() - 0.5);
  aff[2 * lane] = has ? A : 7.;
  aff[2 * lane + 1] = has ? B : 0.;
  __syncthreads();
  const SoElem PL = so_prefix_scan(e, has, lane);
  const SoElem PR = so_suffix_scan(e, nhas, lane);
  const double U = so_affine_suffix_scan(has ? A : 7., has ? B : 0., lane);
  const int cnt = 1 + (int)(rnd() * 3.);
  const int incl = so_prefix_sum(cnt);
  double dev[3] = {0., 0., 0.};
  int ibad = 0;
  if (has) {
    SoElem L = el[0];
    for (int i = 1; i <= lane; ++i) L = so_merge(L, el[i]);
    SoElem R = el[nhas - 1];
    for (int i = nhas - 2; i >= lane; --i) R = so_merge(el[i], R);
    double u = aff[2 * (nhas - 1)];
    for (int i = nhas - 2; i >= lane; --i) u = aff[2 * i] + aff[2 * i + 1] * u;
    auto rel = [](double a, double b) { return __builtin_fabs(a - b) / (1e-300 + __builtin_fabs(b)); };
    dev[0] = fmax(fmax(fmax(rel(PL.a11, L.a11), rel(PL.a12, L.a12)), fmax(rel(PL.a21, L.a21), rel(PL.a22, L.a22))),
                  fmax(rel(PL.c1, L.c1), rel(PL.c2, L.c2)));
    dev[1] = fmax(fmax(fmax(rel(PR.a11, R.a11), rel(PR.a12, R.a12)), fmax(rel(PR.a21, R.a21), rel(PR.a22, R.a22))),
                  fmax(rel(PR.c1, R.c1), rel(PR.c2, R.c2)));
    dev[2] = rel(U, u);
  }
  {
    __shared__ int cs[64];
    cs[lane] = cnt;
    __syncthreads();
    int ref = 0;
    for (int i = 0; i <= lane; ++i) ref += cs[i];
    ibad = ref != incl;
  }
  for (int k = 0; k < 3; ++k) {
    const double m = group_max<64>(dev[k]);
    if (lane == 0) out[k] = m;
  }
  if (ibad) atomicAdd(out_int, 1);
}

// ---- fast exact division self test -----------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long &x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double random_double(unsigned long long &st, int emax) {
  const unsigned long long r = splitmix64(st);
  unsigned long long mant = r & 0xFFFFFFFFFFFFFull;
  const unsigned kind = (unsigned)(r >> 60) & 7u;  // 3/8 of the draws: edge mantissas
  if (kind == 0) mant = 0xFFFFFFFFFFFFFull - ((r >> 52) & 15ull);
  if (kind == 1) mant = (r >> 52) & 15ull;
  if (kind == 2) mant = 0x8000000000000ull + ((r >> 52) & 15ull) - 8ull;
  const unsigned long long r2 = splitmix64(st);
  const int e = (int)(r2 % (unsigned long long)(2 * emax + 1)) - emax;
  const unsigned long long bits = ((r2 >> 63) << 63) | ((unsigned long long)(1023 + e) << 52) | mant;
  return __longlong_as_double((long long)bits);
}
__global__ void k_selftest_fastdiv(unsigned long long seed, int per_thread, int emax,
                                   unsigned long long *mismatch) {
  unsigned long long st = seed + 0x1234567ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x);
  unsigned long long bad = 0;
  for (int i = 0; i < per_thread; ++i) {
    const double d = random_double(st, emax);
    const double y = 1.0 / d;
    const double yl = recip_lo(d, y);
    // several numerators per denominator, as in the kernels (static d, changing a); both
    // reciprocal forms (5 instructions from RN(1/d), 4 from the double-double reciprocal)
    for (int k = 0; k < 4; ++k) {
      const double a = random_double(st, emax);
      const double q_ref = a / d;
      const double q_fast = div_by_recip(a, d, y);
      const double q_fast2 = div_by_recip2(a, d, y, yl);
      bad += (__double_as_longlong(q_ref) != __double_as_longlong(q_fast)) ? 1ull : 0ull;
      bad += (__double_as_longlong(q_ref) != __double_as_longlong(q_fast2)) ? 1ull : 0ull;
    }
  }
  if (bad) atomicAdd(mismatch, bad);
}

// the 3-instruction quotient (col_vertadvdiff's DIV == 6) and the device's own `/` of n operand
// pairs (the host compares both with ITS IEEE quotient)
__global__ void k_selftest_div3(const double *__restrict__ a, const double *__restrict__ d, size_t n,
                                double *__restrict__ q3, double *__restrict__ qd) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const double y = 1.0 / d[i];
    double q = a[i] * y;
    const double r = __builtin_fma(-d[i], q, a[i]);
    q3[i] = __builtin_fma(r, y, q);
    qd[i] = a[i] / d[i];
  }
}
// y[i] = 1.0 / d[i] as every kernel's prologue forms its reciprocals
__global__ void k_recip(const double *__restrict__ d, size_t n, double *__restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] = 1.0 / d[i];
}

__global__ void k_twobasin_forcing(size_t count, const double *__restrict__ iso_A,
                                   const double *__restrict__ zon_A,
                                   const double *__restrict__ so_A,
                                   const double *__restrict__ iso_N,
                                   const double *__restrict__ zon_P,
                                   const double *__restrict__ so_P, double *__restrict__ wA_A,
                                   double *__restrict__ wA_N, double *__restrict__ wA_P) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
       i += (size_t)gridDim.x * blockDim.x) {
    wA_A[i] = (iso_A[i] + zon_A[i] - so_A[i]) * 1e6;  // twobasin_NadeauJansen.py:103
    wA_N[i] = (-iso_N[i]) * 1e6;                       // :104
    wA_P[i] = (-zon_P[i] - so_P[i]) * 1e6;             // :105
  }
}

// ---- the 3-instruction exact quotient by a static denominator, and its proof per denominator.
//   y = RN(1/d);   q0 = RN(a y);   r = RN(a - d q0) (one fma);   q = RN(q0 + r y) (one fma)
// With y = (1/d)(1 + e), |e| <= 2^-53, q0 lies within 1.5 ulp of x = a/d and the last fma rounds
// x + delta with |delta| <= |x - q0| (2 |e| + ...) < 3 * 2^-53 ulp (the residual may itself be
// rounded when q0 is more than an ulp off).  So q = RN(x) unless x lies within 3 * 2^-53 ulp of a
// rounding boundary, a midpoint (2K + 1) 2^(e-53) of two neighbours.  With 53-bit integer
// mantissas A, D of a, d:  x - midpoint = N / (2 D) ulp,  N = A 2^(53+t) - (2K + 1) D  (t = 0 for
// A >= D, 1 for A < D), an integer -- so only numerators with |N| <= 6 can fail, and for a GIVEN
// D these are the few solutions A of  A 2^(53+t) = N (mod D)  with an odd quotient (none at all
// when D has three or more trailing zero bits, as every difference of two grid levels has).
// div3_proof enumerates them and runs the very sequence on each, both signs: all equal to `/` =
// the quotient is correctly rounded for EVERY numerator (finite operands whose quotient and
// residual stay normal: the kernels' operand window, in_fast_div_range).  No denominator has
// failed in 10^8 tried (the cases that could are where q0 is a faithful quotient anyway), but the
// kernels take the 3-instruction form only on this proof, never on statistics.
static double div3_host(double a, double d, double y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-d, q0, a);
  return __builtin_fma(r, y, q0);
}
// 1: proven; 0: not (a candidate fails, or d is zero / subnormal / not finite).  cand: if given,
// receives the candidate numerators (scaled to d's binade), ncand their count.
int div3_proof(double d, std::vector<double> *cand, long long *ncand) {
  const double ad = d < 0 ? -d : d;
  if (!(ad >= 2.2250738585072014e-308 && ad <= 1.7976931348623157e308)) return 0;
  int e;
  const double m = frexp(ad, &e);                      // ad = m 2^e, m in [0.5, 1)
  const uint64_t D = (uint64_t)ldexp(m, 53);           // 53-bit mantissa
  const double dm = ldexp(m, 53);                      // the denominator the tests run on
  const double y = 1.0 / dm;
  const int v = __builtin_ctzll(D);
  if (v >= 3) return 1;                                // |N| <= 6 has no multiple of 2^v: no candidate
  const uint64_t Dp = D >> v;
  if (Dp == 1) return 1;                               // a power of two
  for (int t = 0; t < 2; ++t) {
    const int sh = 53 + t;
    for (int N = -6; N <= 6; ++N) {
      if (N == 0 || (N % (1 << v)) != 0) continue;
      const int64_t Np = N / (1 << v);
      uint64_t x = (uint64_t)(((Np % (int64_t)Dp) + (int64_t)Dp) % (int64_t)Dp);
      for (int k = 0; k < sh - v; ++k) x = (x & 1) ? (x + Dp) / 2 : x / 2;  // Np 2^-(sh-v) mod Dp
      const uint64_t lo = t == 0 ? D : (1ull << 52), hi = t == 0 ? (1ull << 53) : D;
      const uint64_t k0 = lo > x ? (lo - x + Dp - 1) / Dp : 0;
      for (uint64_t A = x + k0 * Dp; A < hi; A += Dp) {
        const __int128 nn = (__int128)((unsigned __int128)A << sh) - N;
        if (nn % (__int128)D != 0 || ((nn / (__int128)D) & 1) == 0) continue;
        const double a = (double)A;
        if (ncand) ++*ncand;
        if (cand) cand->push_back(a);
        if (div3_host(a, dm, y) != a / dm || div3_host(-a, dm, y) != -a / dm) return 0;
      }
    }
  }
  return 1;
}

}  // namespace pm

using namespace pm;

extern "C" {

const char *pm_version(void) { return "pymoc_hip 0.1.0 (gfx950)"; }
const char *pm_last_error(void) { return g_err; }

int pm_device_count(int *count) {
  PM_REQUIRE(count, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(PM_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return PM_OK;
}

int pm_set_device(int device) {
  PM_HIP(hipSetDevice(device));
  return PM_OK;
}

int pm_device_info(char *name, size_t name_len, int *compute_units, size_t *hbm_bytes,
                   int *clock_mhz) {
  int dev = 0;
  PM_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  PM_HIP(hipGetDeviceProperties(&p, dev));
  if (name && name_len) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  if (clock_mhz) *clock_mhz = p.clockRate / 1000;
  return PM_OK;
}

int pm_malloc(void **dptr, size_t bytes) {
  PM_REQUIRE(dptr, "dptr is NULL");
  *dptr = nullptr;
  if (bytes == 0) return PM_OK;
  PM_HIP(hipMalloc(dptr, bytes));
  return PM_OK;
}
int pm_free(void *dptr) {
  if (dptr) PM_HIP(hipFree(dptr));
  return PM_OK;
}
int pm_memset(void *dptr, int value, size_t bytes, pm_stream_t stream) {
  if (bytes) PM_HIP(hipMemsetAsync(dptr, value, bytes, resolve_stream(stream)));
  return PM_OK;
}
int pm_memcpy_h2d(void *dst, const void *src, size_t bytes, pm_stream_t stream) {
  if (!bytes) return PM_OK;
  hipStream_t st = resolve_stream(stream);
  PM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
  PM_HIP(hipStreamSynchronize(st));  // the host buffer is not retained
  return PM_OK;
}
int pm_memcpy_d2h(void *dst, const void *src, size_t bytes, pm_stream_t stream) {
  if (!bytes) return PM_OK;
  hipStream_t st = resolve_stream(stream);
  PM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
  PM_HIP(hipStreamSynchronize(st));
  return PM_OK;
}
int pm_memcpy_d2d(void *dst, const void *src, size_t bytes, pm_stream_t stream) {
  if (bytes)
    PM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice,
                          resolve_stream(stream)));
  return PM_OK;
}

int pm_host_alloc(void **hptr, size_t bytes) {
  PM_REQUIRE(hptr, "hptr is NULL");
  *hptr = nullptr;
  if (bytes == 0) return PM_OK;
  PM_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
  return PM_OK;
}
int pm_host_free(void *hptr) {
  if (hptr) PM_HIP(hipHostFree(hptr));
  return PM_OK;
}
int pm_memcpy_d2h_async(void *dst_pinned, const void *src, size_t bytes, pm_stream_t stream) {
  if (!bytes) return PM_OK;
  PM_REQUIRE(dst_pinned && src, "NULL pointer");
  PM_HIP(hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, resolve_stream(stream)));
  return PM_OK;
}

namespace pm {
struct PackItems {
  pm_row_copy it[PM_PACK_MAX_ITEMS];
};
// one wave per row (rows are 0.4 - 4 KB: whole rows move as coalesced runs), blockIdx.y = item
static __global__ void __launch_bounds__(256) k_rows_pack(PackItems p, const int32_t *__restrict__ sel,
                                                   int nrows) {
  const pm_row_copy c = p.it[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int wave = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  const int nwaves = (int)gridDim.x * 4;
  for (int r = wave; r < nrows; r += nwaves) {
    const size_t sr = sel ? (size_t)sel[r] : (size_t)r;
    const double *__restrict__ s = c.src + sr * (size_t)c.src_stride;
    double *__restrict__ d = c.dst + (size_t)r * (size_t)c.nlev;
    for (int l = lane; l < c.nlev; l += 64) d[l] = s[l];
  }
}
}  // namespace pm
int pm_rows_pack(const pm_row_copy *items, int32_t nitems, const int32_t *sel, int32_t nrows,
                 pm_stream_t stream) {
  PM_REQUIRE(nitems >= 0 && nitems <= PM_PACK_MAX_ITEMS, "nitems %d outside [0, %d]", nitems,
             PM_PACK_MAX_ITEMS);
  PM_REQUIRE(nrows >= 0, "nrows %d < 0", nrows);
  if (nitems == 0 || nrows == 0) return PM_OK;
  PM_REQUIRE(items, "items is NULL");
  pm::PackItems p;
  memset(&p, 0, sizeof(p));
  for (int k = 0; k < nitems; ++k) {
    PM_REQUIRE(items[k].src && items[k].dst, "item %d: NULL pointer", k);
    PM_REQUIRE(items[k].nlev >= 1 && items[k].src_stride >= items[k].nlev,
               "item %d: nlev %d, src_stride %d", k, items[k].nlev, items[k].src_stride);
    p.it[k] = items[k];
  }
  const unsigned gx = (unsigned)((nrows + 3) / 4 < 4096 ? (nrows + 3) / 4 : 4096);
  hipLaunchKernelGGL(pm::k_rows_pack, dim3(gx, (unsigned)nitems), dim3(256), 0,
                     resolve_stream(stream), p, sel, nrows);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

int pm_stream_create(pm_stream_t *stream) {
  PM_REQUIRE(stream, "stream is NULL");
  hipStream_t s;
  PM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (pm_stream_t)s;
  return PM_OK;
}
int pm_stream_create_priority(pm_stream_t *stream, int priority) {
  PM_REQUIRE(stream, "stream is NULL");
  int least = 0, greatest = 0;
  PM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));  // numerically lower = higher
  hipStream_t s;
  PM_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority > 0 ? greatest : 0));
  *stream = (pm_stream_t)s;
  return PM_OK;
}
int pm_stream_destroy(pm_stream_t stream) {
  if (stream) PM_HIP(hipStreamDestroy((hipStream_t)stream));
  return PM_OK;
}
int pm_stream_sync(pm_stream_t stream) {
  PM_HIP(hipStreamSynchronize(resolve_stream(stream)));
  return PM_OK;
}
int pm_device_sync(void) {
  PM_HIP(hipDeviceSynchronize());
  return PM_OK;
}

int pm_event_create(pm_event_t *event) {
  PM_REQUIRE(event, "event is NULL");
  hipEvent_t e;
  PM_HIP(hipEventCreate(&e));
  *event = (pm_event_t)e;
  return PM_OK;
}
int pm_event_destroy(pm_event_t event) {
  if (event) PM_HIP(hipEventDestroy((hipEvent_t)event));
  return PM_OK;
}
int pm_event_record(pm_event_t event, pm_stream_t stream) {
  PM_HIP(hipEventRecord((hipEvent_t)event, resolve_stream(stream)));
  return PM_OK;
}
namespace pm {
static __global__ void k_twocol_forcing(size_t half, const double *__restrict__ psi_iso,
                                 const double *__restrict__ psi_so, double *__restrict__ wA) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= 2 * half) return;
  const double p = psi_iso[i];
  // (the operations of k_thermwind's wA1 / wA2 epilogue and of PM_OP_WA_PSI)
  wA[i] = i < half ? (psi_so ? (p - psi_so[i]) : p) * 1e6 : (-p) * 1e6;
}
}  // namespace pm
int pm_twocol_forcing(int32_t n, int32_t nz, const double *Psi_iso, const double *Psi_SO,
                      double *wA, pm_stream_t stream) {
  PM_REQUIRE(n >= 0 && nz >= 1, "bad shape n=%d nz=%d", n, nz);
  if (n == 0) return PM_OK;
  PM_REQUIRE(Psi_iso && wA, "NULL pointer");
  const size_t half = (size_t)n * nz;
  hipLaunchKernelGGL(pm::k_twocol_forcing, dim3((unsigned)((2 * half + 255) / 256)), dim3(256), 0,
                     resolve_stream(stream), half, Psi_iso, Psi_SO, wA);
  PM_HIP(hipGetLastError());
  return PM_OK;
}
int pm_stream_wait_event(pm_stream_t stream, pm_event_t event) {
  PM_HIP(hipStreamWaitEvent(resolve_stream(stream), (hipEvent_t)event, 0));
  return PM_OK;
}
int pm_event_sync(pm_event_t event) {
  PM_HIP(hipEventSynchronize((hipEvent_t)event));
  return PM_OK;
}
int pm_event_elapsed_ms(pm_event_t start, pm_event_t stop, float *ms) {
  PM_REQUIRE(ms, "ms is NULL");
  PM_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return PM_OK;
}

int pm_graph_begin_capture(pm_stream_t stream) {
  PM_HIP(hipStreamBeginCapture(resolve_stream(stream), hipStreamCaptureModeThreadLocal));
  return PM_OK;
}
int pm_graph_end_capture(pm_stream_t stream, pm_graph_t *graph) {
  PM_REQUIRE(graph, "graph is NULL");
  hipGraph_t g = nullptr;
  PM_HIP(hipStreamEndCapture(resolve_stream(stream), &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess)
    return fail(PM_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  *graph = (pm_graph_t)ge;
  return PM_OK;
}
int pm_graph_launch(pm_graph_t graph, pm_stream_t stream) {
  PM_HIP(hipGraphLaunch((hipGraphExec_t)graph, resolve_stream(stream)));
  return PM_OK;
}
int pm_graph_destroy(pm_graph_t graph) {
  if (graph) PM_HIP(hipGraphExecDestroy((hipGraphExec_t)graph));
  return PM_OK;
}

// -------------------------------------------------------------------- Column
static int column_shape(int ncols, int nz, int lanes_per_col, int *G_out, int *P_out) {
  int G = lanes_per_col ? lanes_per_col : auto_lanes_per_col(ncols, nz);
  PM_REQUIRE(G == 16 || G == 32 || G == 64, "lanes_per_col must be 0, 16, 32 or 64");
  int P = pick_levels_per_lane(G, (nz + G - 1) / G);
  while (P < 0 && G < 64) {
    G *= 2;
    P = pick_levels_per_lane(G, (nz + G - 1) / G);
  }
  PM_REQUIRE(P > 0, "nz=%d does not fit %d lanes", nz, G);
  *G_out = G;
  *P_out = P;
  return PM_OK;
}

int pm_column_kernel_shape(int32_t ncols, int32_t nz, int32_t lanes_per_col, int32_t *lanes,
                           int32_t *levels_per_lane) {
  PM_REQUIRE(lanes && levels_per_lane, "NULL output");
  PM_REQUIRE(nz >= 2 && nz <= 1024, "nz must be in [2,1024]");
  int G = 0, P = 0;
  const int rc = column_shape(ncols, nz, lanes_per_col, &G, &P);
  if (rc != PM_OK) return rc;
  *lanes = G;
  *levels_per_lane = P;
  return PM_OK;
}

int pm_column_kernel_name(int32_t ncols, int32_t nz, int32_t lanes_per_col, int32_t nsteps,
                          int32_t ops, int32_t has_horadv, char *name, size_t name_len) {
  PM_REQUIRE(name && name_len > 0, "NULL output");
  PM_REQUIRE(nz >= 2 && nz <= 1024, "nz must be in [2,1024]");
  int G = 0, P = 0;
  const int rc = column_shape(ncols, nz, lanes_per_col, &G, &P);
  if (rc != PM_OK) return rc;
  const bool plain =
      (ops & ~(PM_OP_WEFF | PM_OP_CONTRACTED | PM_OP_WA_PSI | PM_OP_WA_TWOBASIN)) == PM_OP_TIMESTEP && !has_horadv;
  if (G == 64 && P <= 4 && nsteps < 3 && plain && stream_cols_per_wave(ncols) >= 2)
    snprintf(name, name_len, "k_column_stream<%d>", P);  // (+ ring depth / affine-kappa variants)
  else if (G == 64 && P <= 4 && nsteps >= 3 && plain && (ops & PM_OP_CONTRACTED))
    snprintf(name, name_len, "k_column_steps<64,%d,4,true>", P);
  else  // mirrors launch_column_steps (column.hip.h)
    snprintf(name, name_len, "k_column_steps<%d,%d,%d,%s>", G, P,
             nsteps >= 3 ? (plain ? 2 : 1) : 0, (nsteps >= 3 && plain) ? "true" : "false");
  return PM_OK;
}

int pm_column_steps(const pm_columns *cols, const double *wA, const double *vdx_in,
                    const double *b_in, double dt, int32_t nsteps, int32_t ops,
                    int32_t lanes_per_col, pm_stream_t stream) {
  PM_REQUIRE(cols, "cols is NULL");
  const pm_columns &c = *cols;
  PM_REQUIRE(c.ncols >= 0 && c.nz >= 2 && c.nz <= 1024,
             "bad batch shape ncols=%d nz=%d (need nz in [2,1024])", c.ncols, c.nz);
  PM_REQUIRE(c.nsel >= 1 && c.nsel <= 2, "nsel must be 1 or 2 (got %d)", c.nsel);
  if (c.ncols == 0) return PM_OK;  // empty batch: nothing to do, pointers may be NULL
  PM_REQUIRE(c.z && c.b && c.kappa && c.area && c.dAkappa && c.bs && c.bbot && c.N2min,
             "pm_columns has a NULL required pointer");
  PM_REQUIRE(nsteps >= 0, "nsteps < 0");
  PM_REQUIRE((ops & ~(PM_OP_TIMESTEP | PM_OP_WEFF | PM_OP_CONTRACTED | PM_OP_WA_PSI | PM_OP_WA_TWOBASIN)) == 0,
             "unknown op bits 0x%x", ops);
  if (ops & PM_OP_WA_TWOBASIN) {
    PM_REQUIRE(vdx_in && b_in && !(ops & (PM_OP_WEFF | PM_OP_WA_PSI)) &&
                   (ops & PM_OP_TIMESTEP) == PM_OP_TIMESTEP && nsteps >= 3 && c.ncols % 3 == 0,
               "PM_OP_WA_TWOBASIN: plain timesteps (>= 3 per launch) of a three-column ensemble, the "
               "three overturning arrays given, no PM_OP_WEFF / PM_OP_WA_PSI");
  } else if (ops & PM_OP_WA_PSI) {
    PM_REQUIRE(!vdx_in && !(ops & PM_OP_WEFF) && (ops & PM_OP_TIMESTEP) == PM_OP_TIMESTEP &&
                   nsteps >= 3 && (c.ncols & 1) == 0,
               "PM_OP_WA_PSI: plain timesteps (>= 3 per launch) of a two-column ensemble, no "
               "horadv, no PM_OP_WEFF");
    vdx_in = nullptr;
  } else {
    PM_REQUIRE(!vdx_in || b_in, "b_in is needed if vdx_in is provided");
  }
  PM_REQUIRE(!(ops & PM_OP_VERTADVDIFF) || wA, "wA is NULL");
  if (c.ncols == 0 || nsteps == 0 || (ops & PM_OP_TIMESTEP) == 0) return PM_OK;
  int G = 0, P = 0;
  const int src = column_shape(c.ncols, c.nz, lanes_per_col, &G, &P);
  if (src != PM_OK) return src;
  hipStream_t st = resolve_stream(stream);
  switch (G) {
    case 16: return column_steps_g16(P, c, wA, vdx_in, b_in, dt, nsteps, ops, st);
    case 32: return column_steps_g32(P, c, wA, vdx_in, b_in, dt, nsteps, ops, st);
    default: return column_steps_g64(P, c, wA, vdx_in, b_in, dt, nsteps, ops, st);
  }
}

static __global__ void k_column_weff(pm_columns c, const double *__restrict__ wA,
                              double *__restrict__ weff) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t n = (size_t)c.ncols * c.nz;
  if (i >= n) return;
  const int col = (int)(i / c.nz);
  const int sel = c.ksel ? c.ksel[col] : 0;
  weff[i] = wA[i] - c.dAkappa[(size_t)sel * n + i];
}

int pm_column_weff(const pm_columns *cols, const double *wA, double *weff, pm_stream_t stream) {
  PM_REQUIRE(cols, "cols is NULL");
  const pm_columns &c = *cols;
  PM_REQUIRE(c.ncols >= 0 && c.nz >= 2 && c.nsel >= 1 && c.nsel <= 2, "bad batch shape");
  if (c.ncols == 0) return PM_OK;
  PM_REQUIRE(c.dAkappa && wA && weff, "pm_column_weff has a NULL pointer");
  const size_t n = (size_t)c.ncols * c.nz;
  hipLaunchKernelGGL(k_column_weff, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     resolve_stream(stream), c, wA, weff);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

int pm_thermwind_residuals(int32_t m, const double *x, const double *y0, const double *y1,
                           const double *g, const double *g_lob, double *rms,
                           pm_stream_t stream) {
  PM_REQUIRE(m >= 2 && x && y0 && y1 && g && g_lob && rms, "bad mesh size or NULL pointer");
  hipStream_t st = resolve_stream(stream);
  hipLaunchKernelGGL(k_thermwind_residuals<0>, dim3((unsigned)((m - 1 + 127) / 128)), dim3(128), 0, st,
                     (int)m, x, y0, y1, g, g_lob, rms);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

// ------------------------------------------------------------- Psi_Thermwind
int pm_thermwind_update(const pm_thermwind *tw, int32_t ops, pm_stream_t stream) {
  PM_REQUIRE(tw, "tw is NULL");
  const pm_thermwind &a = *tw;
  PM_REQUIRE(a.n >= 0 && a.nz >= 2 && a.nz <= 1024, "bad shape n=%d nz=%d", a.n, a.nz);
  PM_REQUIRE((ops & ~15) == 0 && ops != 0, "bad ops 0x%x", ops);
  PM_REQUIRE(!(ops & PM_TW_PSIBZ) || (ops & PM_TW_PSIB), "PM_TW_PSIBZ needs PM_TW_PSIB");
  PM_REQUIRE(!(ops & PM_TW_PSIB) || a.nb >= 1, "nb must be >= 1");
  if (a.n == 0) return PM_OK;
  PM_REQUIRE(a.z && a.b1 && a.b2 && a.Psi, "pm_thermwind has a NULL required pointer");
  PM_REQUIRE(!(ops & PM_TW_SOLVE) || a.f, "f is NULL");
  if (a.n == 0) return PM_OK;
  return dispatch_thermwind(a, ops, resolve_stream(stream));
}

// -------------------------------------------------------------------- Psi_SO
int pm_psi_so_update(const pm_psi_so *so, int32_t ops, pm_stream_t stream) {
  PM_REQUIRE(so, "so is NULL");
  const pm_psi_so &a = *so;
  PM_REQUIRE(a.n >= 0 && a.nz >= 2 && a.nz <= 512 && a.ny >= 2 && a.ny <= 2048,
             "bad shape n=%d nz=%d ny=%d", a.n, a.nz, a.ny);
  PM_REQUIRE(ops >= 1 && ops <= 3, "bad ops %d", ops);
  if (a.n == 0) return PM_OK;
  PM_REQUIRE(a.z && a.y && a.b && a.bs && a.tau && a.KGM && a.Psi_Ek,
             "pm_psi_so has a NULL required pointer");
  PM_REQUIRE(!(ops & PM_SO_OP_GM) || (a.Psi && a.Psi_GM), "Psi / Psi_GM is NULL");
  PM_REQUIRE(a.bvp_refine >= -1 && a.bvp_refine <= 256, "bad bvp_refine (<= 0: adaptive mesh, R > 0: fixed R-fold mesh)");
  if (a.n == 0) return PM_OK;
  return dispatch_psi_so(a, ops, resolve_stream(stream));
}

// --------------------------------------------------------------------- SO_ML
int pm_so_ml_step(const pm_so_ml *ml, double dt, pm_stream_t stream) {
  PM_REQUIRE(ml, "ml is NULL");
  const pm_so_ml &a = *ml;
  PM_REQUIRE(a.n >= 0 && a.nz >= 2 && a.ny >= 3 && a.nz <= 4096 && a.ny <= 2048,
             "bad shape n=%d nz=%d ny=%d", a.n, a.nz, a.ny);
  if (a.n == 0) return PM_OK;
  PM_REQUIRE(a.y && a.bs && a.b_basin && a.Psi_b && a.surflux && a.rest_mask && a.b_rest,
             "pm_so_ml has a NULL required pointer");
  if (a.n == 0) return PM_OK;
  return launch_so_ml(a, dt, resolve_stream(stream));
}

int pm_jn2018_bc_switch(const pm_jn2018_bc *bc, pm_stream_t stream) {
  PM_REQUIRE(bc, "bc is NULL");
  const pm_jn2018_bc &a = *bc;
  PM_REQUIRE(a.n >= 0 && a.nz >= 2 && a.ny >= 1, "bad shape n=%d nz=%d ny=%d", a.n, a.nz, a.ny);
  PM_REQUIRE(a.Psi_SO && a.Psi_res_b && a.Psi_res_n && a.b_basin && a.b_north && a.bs_SO &&
                 a.bbot && a.ksel,
             "pm_jn2018_bc has a NULL pointer");
  if (a.n == 0) return PM_OK;
  hipLaunchKernelGGL(k_jn2018_bc_switch, dim3((a.n + 255) / 256), dim3(256), 0,
                     resolve_stream(stream), a);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

int pm_jn2018_steps(const pm_jn2018 *jn, double dt, int32_t nsteps, pm_stream_t stream) {
  PM_REQUIRE(jn, "jn is NULL");
  const pm_jn2018 &a = *jn;
  const pm_columns &c = a.cols;
  PM_REQUIRE(a.n >= 0 && c.ncols == 2 * a.n && a.ml.n == a.n && a.ml.nz == c.nz,
             "inconsistent batch sizes n=%d ncols=%d ml.n=%d", a.n, c.ncols, a.ml.n);
  PM_REQUIRE(c.nz >= 2 && c.nz <= 256 && a.ml.ny >= 3 && a.ml.ny <= 2048,
             "bad shape nz=%d ny=%d (fused loop needs nz <= 256)", c.nz, a.ml.ny);
  PM_REQUIRE(c.nsel == 2 && c.ksel && c.z && c.b && c.kappa && c.area && c.dAkappa && c.bs &&
                 c.bbot && c.N2min,
             "pm_jn2018.cols has a NULL pointer or nsel != 2");
  PM_REQUIRE(a.wA && a.Psi_SO && a.Psi_res_b && a.Psi_res_n, "pm_jn2018 has a NULL pointer");
  PM_REQUIRE(a.ml.y && a.ml.bs && a.ml.surflux && a.ml.rest_mask && a.ml.b_rest,
             "pm_jn2018.ml has a NULL pointer");
  PM_REQUIRE(nsteps >= 0, "nsteps < 0");
  if (a.n == 0 || nsteps == 0) return PM_OK;
  hipStream_t st = resolve_stream(stream);
  const bool force_general = getenv("PYMOC_JN_GENERAL") != nullptr;  // A/B experiments
  if (!force_general && jn2018_fast_applies(a)) return launch_jn2018_fast(a, dt, nsteps, st);
  switch ((c.nz + 63) / 64) {
    case 1: return launch_jn2018_steps<1>(a, dt, nsteps, st);
    case 2: return launch_jn2018_steps<2>(a, dt, nsteps, st);
    case 3: return launch_jn2018_steps<3>(a, dt, nsteps, st);
    default: return launch_jn2018_steps<4>(a, dt, nsteps, st);
  }
}

int pm_so_tw_update(const pm_psi_so *so, const pm_thermwind *tw, int32_t tw_ops,
                    pm_stream_t stream) {
  PM_REQUIRE(so && tw, "NULL argument");
  const pm_psi_so &a = *so;
  const pm_thermwind &t = *tw;
  PM_REQUIRE(a.n == t.n && a.nz == t.nz && a.n >= 0, "inconsistent sizes");
  PM_REQUIRE(t.nz >= 2 && t.nz <= 256 && a.ny >= 2 && a.ny <= 2048 && t.nb >= 1,
             "pm_so_tw_update: nz <= 256 (nz=%d)", t.nz);
  PM_REQUIRE(!(a.flags & PM_SO_HAS_C), "pm_so_tw_update: Psi_SO without the boundary-value smoother");
  PM_REQUIRE((tw_ops & ~15) == 0 && tw_ops != 0 && (!(tw_ops & PM_TW_PSIBZ) || (tw_ops & PM_TW_PSIB)),
             "bad thermal-wind ops 0x%x", tw_ops);
  if (a.n == 0) return PM_OK;
  PM_REQUIRE(a.z && a.y && a.b && a.bs && a.tau && a.KGM && a.Psi && a.Psi_Ek && a.Psi_GM,
             "pm_psi_so has a NULL required pointer");
  PM_REQUIRE(t.z && t.b1 && t.b2 && t.Psi && (!(tw_ops & PM_TW_SOLVE) || t.f),
             "pm_thermwind has a NULL required pointer");
  const int rc = launch_so_tw_update(a, t, tw_ops, resolve_stream(stream));
  if (rc == -1) return fail(PM_EINVAL, "pm_so_tw_update: shape not covered (nz=%d)", t.nz);
  return rc;
}

int pm_run_lds_bytes(int32_t kind, int32_t nz, int32_t nb, int32_t ny, size_t *bytes) {
  PM_REQUIRE(bytes && (kind == 0 || kind == 1), "bad arguments");
  *bytes = run_lds_bytes(kind, nz, nb, ny);
  return PM_OK;
}

int pm_twocol_run(const pm_twocol_loop *run, pm_stream_t stream) {
  PM_REQUIRE(run, "run is NULL");
  const pm_twocol_loop &r = *run;
  const pm_columns &c = r.cols;
  const pm_thermwind &t = r.tw;
  PM_REQUIRE(t.n >= 0 && c.ncols == 2 * t.n && t.nz == c.nz, "inconsistent sizes n=%d ncols=%d",
             t.n, c.ncols);
  PM_REQUIRE(c.nz >= 4 && c.nz <= 256 && t.nb >= 1, "pm_twocol_run needs 4 <= nz <= 256 (nz=%d)", c.nz);
  PM_REQUIRE(c.nsel >= 1 && c.nsel <= 2 && (c.nsel == 1 || c.ksel), "bad nsel / ksel");
  const pm_run_schedule &s = r.sched;
  PM_REQUIRE(s.n_first >= 0 && s.n_updates >= 0 && s.m_steps >= 0 && s.n_last >= 0, "bad schedule");
  if (t.n == 0 || (s.n_first == 0 && s.n_updates == 0)) return PM_OK;
  PM_REQUIRE(c.z && c.b && c.kappa && c.area && c.dAkappa && c.bs && c.bbot && c.N2min && c.flags,
             "pm_twocol_loop.cols has a NULL required pointer");
  PM_REQUIRE(t.z && t.b1 && t.b2 && t.f && t.Psi && t.wA1 && t.wA2 && r.wA,
             "pm_twocol_loop.tw has a NULL required pointer");
  PM_REQUIRE(!t.b1_mid && !t.b2_mid && !t.Psi_SO, "pm_twocol_run: array profiles, no SO channel");
  return launch_twocol_run(r, resolve_stream(stream));
}

int pm_jn2018_run(const pm_jn2018_loop *run, pm_stream_t stream) {
  PM_REQUIRE(run, "run is NULL");
  const pm_jn2018_loop &r = *run;
  const pm_jn2018 &a = r.jn;
  const pm_columns &c = a.cols;
  PM_REQUIRE(a.n >= 0 && c.ncols == 2 * a.n && a.ml.n == a.n && a.ml.nz == c.nz && r.tw.n == a.n &&
                 r.tw.nz == c.nz && r.so.n == a.n && r.so.nz == c.nz && r.so.ny == a.ml.ny,
             "inconsistent batch sizes n=%d", a.n);
  PM_REQUIRE(jn2018_fast_applies(a), "pm_jn2018_run needs PM_JN_UNIFORM_AREA, ny <= 64, 4 <= nz <= 256, ml.status");
  PM_REQUIRE(c.nsel == 2 && c.ksel && c.z && c.b && c.kappa && c.area && c.dAkappa && c.bs &&
                 c.bbot && c.N2min,
             "pm_jn2018_loop.jn.cols has a NULL pointer or nsel != 2");
  PM_REQUIRE(a.wA && a.Psi_SO && a.Psi_res_b && a.Psi_res_n, "pm_jn2018_loop.jn has a NULL pointer");
  PM_REQUIRE(a.ml.y && a.ml.bs && a.ml.surflux && a.ml.rest_mask && a.ml.b_rest && a.ml.ny >= 3,
             "pm_jn2018_loop.jn.ml has a NULL pointer");
  PM_REQUIRE(r.tw.z && r.tw.b1 && r.tw.b2 && r.tw.f && r.tw.Psi && r.tw.nb >= 1 && !r.tw.b1_mid &&
                 !r.tw.b2_mid,
             "pm_jn2018_loop.tw has a NULL required pointer");
  PM_REQUIRE(r.so.z && r.so.y && r.so.b && r.so.bs && r.so.tau && r.so.KGM && r.so.Psi &&
                 r.so.Psi_Ek && r.so.Psi_GM && !(r.so.flags & PM_SO_HAS_C),
             "pm_jn2018_loop.so has a NULL required pointer (or the boundary-value smoother)");
  const pm_run_schedule &s = r.sched;
  PM_REQUIRE(s.n_first >= 0 && s.n_updates >= 0 && s.m_steps >= 0 && s.n_last >= 0, "bad schedule");
  if (a.n == 0 || (s.n_first == 0 && s.n_updates == 0)) return PM_OK;
  return launch_jn2018_run(r, resolve_stream(stream));
}

int pm_twobasin_forcing(int32_t n, int32_t nz, const double *Psi_iso_Atl,
                        const double *Psi_zonal_Atl, const double *SO_Atl,
                        const double *Psi_iso_N, const double *Psi_zonal_Pac,
                        const double *SO_Pac, double *wA_Atl, double *wAN, double *wA_Pac,
                        pm_stream_t stream) {
  PM_REQUIRE(n >= 0 && nz >= 1, "bad shape n=%d nz=%d", n, nz);
  PM_REQUIRE(Psi_iso_Atl && Psi_zonal_Atl && SO_Atl && Psi_iso_N && Psi_zonal_Pac && SO_Pac &&
                 wA_Atl && wAN && wA_Pac,
             "NULL pointer");
  const size_t count = (size_t)n * nz;
  if (count == 0) return PM_OK;
  const unsigned grid = (unsigned)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_twobasin_forcing, dim3(grid), dim3(256), 0, resolve_stream(stream),
                     count, Psi_iso_Atl, Psi_zonal_Atl, SO_Atl, Psi_iso_N, Psi_zonal_Pac,
                     SO_Pac, wA_Atl, wAN, wA_Pac);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

// ---------------------------------------------------------------------- RCCL
int pm_comm_unique_id(void *id128) {
  PM_REQUIRE(id128, "id128 is NULL");
  if (!nccl().ok) return fail(PM_ENCCL, "librccl.so could not be loaded: %s", dlerror());
  NcclId id;
  PM_NCCL(nccl().GetUniqueId(&id));
  memcpy(id128, id.internal, PM_COMM_ID_BYTES);
  return PM_OK;
}

int pm_comm_init(pm_comm_t *comm, int32_t nranks, int32_t rank, const void *id128) {
  PM_REQUIRE(comm && id128, "NULL argument");
  PM_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
  if (!nccl().ok) return fail(PM_ENCCL, "librccl.so could not be loaded: %s", dlerror());
  Comm *c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  NcclId id;
  memcpy(id.internal, id128, PM_COMM_ID_BYTES);
  int r = nccl().CommInitRank(&c->comm, nranks, id, rank);
  if (r != 0) {
    delete c;
    return fail(PM_ENCCL, "ncclCommInitRank failed: %s", nccl().GetErrorString(r));
  }
  hipError_t e = hipMalloc((void **)&c->scratch, 2 * sizeof(double));
  if (e != hipSuccess) {
    nccl().CommDestroy(c->comm);
    delete c;
    return fail(PM_EHIP, "hipMalloc: %s", hipGetErrorString(e));
  }
  *comm = (pm_comm_t)c;
  return PM_OK;
}

int pm_comm_destroy(pm_comm_t comm) {
  if (!comm) return PM_OK;
  Comm *c = (Comm *)comm;
  if (c->scratch) (void)hipFree(c->scratch);
  int r = nccl().CommDestroy(c->comm);
  delete c;
  if (r != 0) return fail(PM_ENCCL, "ncclCommDestroy: %s", nccl().GetErrorString(r));
  return PM_OK;
}

int pm_comm_allgather(pm_comm_t comm, const void *send, void *recv, size_t count,
                      pm_stream_t stream) {
  PM_REQUIRE(comm && send && recv, "NULL argument");
  Comm *c = (Comm *)comm;
  PM_NCCL(nccl().AllGather(send, recv, count, NCCL_FLOAT64, c->comm, resolve_stream(stream)));
  return PM_OK;
}

int pm_comm_gather_root(pm_comm_t comm, const void *send, void *recv, size_t count, int32_t root,
                        int32_t self_loop, pm_stream_t stream) {
  PM_REQUIRE(comm && send, "NULL argument");
  Comm *c = (Comm *)comm;
  PM_REQUIRE(root >= 0 && root < c->nranks, "root %d outside [0, %d)", root, c->nranks);
  PM_REQUIRE(c->rank != root || recv, "recv is NULL on the root");
  hipStream_t st = resolve_stream(stream);
  if (count == 0) return PM_OK;
  const bool loop = self_loop && c->nranks == 1;
  if (c->rank == root && !loop)  // the root's own block never leaves the device
    PM_HIP(hipMemcpyAsync((double *)recv + (size_t)root * count, send, count * sizeof(double),
                          hipMemcpyDeviceToDevice, st));
  if (c->nranks == 1 && !loop) return PM_OK;
  PM_NCCL(nccl().GroupStart());
  int r = 0;
  if (c->rank == root) {
    for (int p = 0; p < c->nranks && r == 0; ++p)
      if (p != root || loop)
        r = nccl().Recv((double *)recv + (size_t)p * count, count, NCCL_FLOAT64, p, c->comm, st);
    if (loop && r == 0) r = nccl().Send(send, count, NCCL_FLOAT64, root, c->comm, st);
  } else {
    r = nccl().Send(send, count, NCCL_FLOAT64, root, c->comm, st);
  }
  const int re = nccl().GroupEnd();
  if (r != 0) return fail(PM_ENCCL, "ncclSend/ncclRecv failed: %s", nccl().GetErrorString(r));
  if (re != 0) return fail(PM_ENCCL, "ncclGroupEnd failed: %s", nccl().GetErrorString(re));
  return PM_OK;
}

int pm_comm_allreduce_max(pm_comm_t comm, const void *send, void *recv, size_t count,
                          pm_stream_t stream) {
  PM_REQUIRE(comm && send && recv, "NULL argument");
  Comm *c = (Comm *)comm;
  PM_NCCL(nccl().AllReduce(send, recv, count, NCCL_FLOAT64, NCCL_MAX, c->comm,
                           resolve_stream(stream)));
  return PM_OK;
}

int pm_comm_barrier(pm_comm_t comm, pm_stream_t stream) {
  PM_REQUIRE(comm, "comm is NULL");
  Comm *c = (Comm *)comm;
  hipStream_t st = resolve_stream(stream);
  PM_HIP(hipMemsetAsync(c->scratch, 0, 2 * sizeof(double), st));
  PM_NCCL(nccl().AllReduce(c->scratch, c->scratch + 1, 1, NCCL_FLOAT64, NCCL_MAX, c->comm, st));
  PM_HIP(hipStreamSynchronize(st));
  return PM_OK;
}

int pm_selftest_fastdiv(uint64_t seed, int32_t blocks, int32_t per_thread, int32_t emax,
                        uint64_t *tested, uint64_t *mismatches) {
  PM_REQUIRE(tested && mismatches, "NULL output");
  PM_REQUIRE(blocks > 0 && per_thread > 0 && emax >= 0 && emax <= 400, "bad sizes");
  unsigned long long *d = nullptr;
  PM_HIP(hipMalloc((void **)&d, sizeof(unsigned long long)));
  hipStream_t st = resolve_stream(nullptr);
  PM_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_selftest_fastdiv, dim3(blocks), dim3(256), 0, st,
                     (unsigned long long)seed, per_thread, emax, d);
  PM_HIP(hipGetLastError());
  unsigned long long h = 0;
  PM_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st));
  PM_HIP(hipStreamSynchronize(st));
  PM_HIP(hipFree(d));
  *mismatches = h;
  *tested = 4ull * (unsigned long long)per_thread * 256ull * (unsigned long long)blocks;
  return PM_OK;
}

int pm_selftest_so_scans(int32_t nhas, uint64_t seed, double *max_rel3, int32_t *sum_mismatches) {
  PM_REQUIRE(max_rel3 && sum_mismatches, "NULL output");
  PM_REQUIRE(nhas >= 1 && nhas <= 64, "nhas must be in [1,64]");
  double *d = nullptr;
  PM_HIP(hipMalloc((void **)&d, 4 * sizeof(double)));
  hipStream_t st = resolve_stream(nullptr);
  PM_HIP(hipMemsetAsync(d, 0, 4 * sizeof(double), st));
  hipLaunchKernelGGL(k_selftest_so_scans, dim3(1), dim3(64), 0, st, (int)nhas,
                     (unsigned long long)seed, d, reinterpret_cast<int *>(d + 3));
  PM_HIP(hipGetLastError());
  double h[4];
  PM_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
  PM_HIP(hipStreamSynchronize(st));
  PM_HIP(hipFree(d));
  for (int k = 0; k < 3; ++k) max_rel3[k] = h[k];
  int32_t bad;
  memcpy(&bad, &h[3], sizeof(bad));
  *sum_mismatches = bad;
  return PM_OK;
}

int pm_selftest_lane_shift(int32_t *mismatches) {
  PM_REQUIRE(mismatches, "mismatches is NULL");
  int *d = nullptr;
  PM_HIP(hipMalloc((void **)&d, sizeof(int)));
  hipStream_t st = resolve_stream(nullptr);
  PM_HIP(hipMemsetAsync(d, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_selftest_lane_shift, dim3(8), dim3(256), 0, st, d);
  PM_HIP(hipGetLastError());
  PM_HIP(hipMemcpyAsync(mismatches, d, sizeof(int), hipMemcpyDeviceToHost, st));
  PM_HIP(hipStreamSynchronize(st));
  PM_HIP(hipFree(d));
  return PM_OK;
}

int pm_div3_proven(const double *d, int64_t n, int32_t *proven, int64_t *candidates) {
  PM_REQUIRE(proven, "proven is NULL");
  PM_REQUIRE(n == 0 || d, "d is NULL");
  long long nc = 0;
  int ok = 1;
  for (int64_t i = 0; i < n && ok; ++i) ok = div3_proof(d[i], nullptr, &nc);
  *proven = ok;
  if (candidates) *candidates = nc;
  return PM_OK;
}

int pm_recip_check(const double *d, int64_t n, int32_t *ok) {
  PM_REQUIRE(ok, "ok is NULL");
  PM_REQUIRE(n == 0 || d, "d is NULL");
  *ok = 1;
  if (n == 0) return PM_OK;
  double *dd = nullptr, *dy = nullptr;
  PM_HIP(hipMalloc((void **)&dd, (size_t)n * sizeof(double)));
  PM_HIP(hipMalloc((void **)&dy, (size_t)n * sizeof(double)));
  hipStream_t s = resolve_stream(nullptr);
  PM_HIP(hipMemcpyAsync(dd, d, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_recip, dim3(256), dim3(256), 0, s, dd, (size_t)n, dy);
  PM_HIP(hipGetLastError());
  std::vector<double> y((size_t)n);
  PM_HIP(hipMemcpyAsync(y.data(), dy, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  PM_HIP(hipStreamSynchronize(s));
  PM_HIP(hipFree(dd));
  PM_HIP(hipFree(dy));
  for (int64_t i = 0; i < n; ++i) {
    const double h = 1.0 / d[i];  // the host's correctly rounded quotient
    if (memcmp(&h, &y[(size_t)i], sizeof(double)) != 0) *ok = 0;
  }
  return PM_OK;
}

int pm_selftest_div3(uint64_t seed, int32_t ndenoms, uint64_t *tested, uint64_t *mismatches,
                     uint64_t *unproven, uint64_t *device_div_off, double *one_bad_pair) {
  PM_REQUIRE(tested && mismatches && unproven && device_div_off, "NULL output");
  PM_REQUIRE(ndenoms >= 1 && ndenoms <= (1 << 22), "ndenoms must be in [1, 2^22]");
  std::vector<double> ha, hd, c;
  unsigned long long st = seed ? seed : 1, nun = 0;
  auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (int i = 0; i < ndenoms; ++i) {
    // mantissas: uniform, or a few units in the last place off 1 or 2 (where the first product
    // is worst), or with one or two trailing zero bits
    const unsigned long long r = next();
    uint64_t D = (1ull << 52) | (next() >> 12);
    const int kind = (int)(r & 7);
    if (kind == 0) D = (1ull << 52) + 1 + (next() & 1023);
    if (kind == 1) D = (1ull << 53) - 1 - (next() & 1023);
    if (kind == 2) D &= ~1ull;
    if (kind == 3) D &= ~3ull;
    const double d = ldexp((double)D, -52 + (int)((r >> 8) % 41) - 20);
    c.clear();
    long long nc = 0;
    if (!div3_proof(d, &c, &nc)) ++nun;
    const double scale = ldexp(1.0, (int)((r >> 16) % 41) - 20);
    for (double a : c) {
      ha.push_back(a * scale);
      hd.push_back(d);
      ha.push_back(-a * scale);
      hd.push_back(d);
    }
    for (int k = 0; k < 2; ++k) {  // and two arbitrary numerators
      ha.push_back(ldexp((double)((1ull << 52) | (next() >> 12)), -40 - (int)(next() % 30)));
      hd.push_back(d);
    }
  }
  const size_t n = ha.size();
  double *da = nullptr, *dd = nullptr, *dq = nullptr;
  PM_HIP(hipMalloc((void **)&da, n * sizeof(double)));
  PM_HIP(hipMalloc((void **)&dd, n * sizeof(double)));
  PM_HIP(hipMalloc((void **)&dq, 2 * n * sizeof(double)));
  hipStream_t s = resolve_stream(nullptr);
  PM_HIP(hipMemcpyAsync(da, ha.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
  PM_HIP(hipMemcpyAsync(dd, hd.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_selftest_div3, dim3(256), dim3(256), 0, s, da, dd, n, dq, dq + n);
  PM_HIP(hipGetLastError());
  std::vector<double> hq(2 * n);
  PM_HIP(hipMemcpyAsync(hq.data(), dq, 2 * n * sizeof(double), hipMemcpyDeviceToHost, s));
  PM_HIP(hipStreamSynchronize(s));
  PM_HIP(hipFree(da));
  PM_HIP(hipFree(dd));
  PM_HIP(hipFree(dq));
  unsigned long long bad = 0, off = 0;
  for (size_t i = 0; i < n; ++i) {
    const double q = ha[i] / hd[i];  // the host's IEEE quotient: the reference
    if (memcmp(&q, &hq[i], sizeof(double)) != 0) {
      if (one_bad_pair && !bad) {
        one_bad_pair[0] = ha[i];
        one_bad_pair[1] = hd[i];
      }
      ++bad;
    }
    if (memcmp(&q, &hq[n + i], sizeof(double)) != 0) ++off;
  }
  *tested = n;
  *mismatches = bad;
  *unproven = nun;
  *device_div_off = off;
  return PM_OK;
}

}  // extern "C"

#ifdef PM_PHASE_PROFILE
// profiling build only (make -B lib EXTRA=-DPM_PHASE_PROFILE): read and clear the phase clocks
extern "C" int pm_debug_prof(unsigned long long *out16) {
  unsigned long long zero[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pm::pm_prof), sizeof(zero)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(pm::pm_prof), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
extern "C" int pm_debug_jf_rare(unsigned long long *out8) {
  unsigned long long zero[72] = {0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pm::jf_rare), sizeof(zero)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(pm::jf_rare), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
extern "C" int pm_debug_wave_times(unsigned long long *out, int n) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pm::pm_wave_times), sizeof(unsigned long long) * 2 * n) != hipSuccess) return -1;
  return 0;
}
#endif
