// fp64_chain.hip -- dependent-issue latency and issue rate of fp64 VALU instructions on
// gfx950 with ONE wave per SIMD (the regime of k_column_steps at 1024 columns).
// Build: hipcc --offload-arch=gfx950 -O3 -o fp64_chain fp64_chain.hip ; run on the GPU box.
// For NCHAIN independent chains of LEN dependent ops each it reports cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NCHAIN, int OP>
__global__ void k_chain(double *out, long long *cyc, int iters, double a, double b) {
  double x[NCHAIN];
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) x[c] = a + c + threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c) {
        if (OP == 0) x[c] = __builtin_fma(x[c], b, a);
        if (OP == 1) x[c] = x[c] * b;
        if (OP == 2) x[c] = x[c] + b;
        if (OP == 3) {  // DPP shift then add (2 movs + 1 add per link)
          int lo = __double2loint(x[c]), hi = __double2hiint(x[c]);
          lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
          hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
          x[c] = __hiloint2double(hi, lo) + b;
        }
        if (OP == 4) {  // compare -> select (v_cmp + 2 v_cndmask) then add
          x[c] = (x[c] > a ? b : x[c]) + b;
        }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NCHAIN, int OP>
void run(const char *name, int waves_per_simd) {
  double *out;
  long long *cyc, h;
  // 256 CUs x 4 SIMDs; blocks of 64 threads; waves_per_simd waves on each SIMD
  const int blocks = 1024 * waves_per_simd;
  hipMalloc(&out, blocks * 64 * sizeof(double));
  hipMalloc(&cyc, sizeof(long long));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_chain<NCHAIN, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0000001, 0.9999999);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<NCHAIN, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0000001, 0.9999999);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double links = (double)iters * 16 * NCHAIN;
  printf("%-22s chains=%d waves/SIMD=%d : %.2f shader-clock cycles per link (%.1f ns per link-row, kernel %.3f ms)\n",
         name, NCHAIN, waves_per_simd, (double)h / links, ms * 1e6 / (iters * 16.0), ms);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<1, 0>("v_fma_f64", w);
    run<2, 0>("v_fma_f64", w);
    run<3, 0>("v_fma_f64", w);
    run<4, 0>("v_fma_f64", w);
    run<6, 0>("v_fma_f64", w);
    run<8, 0>("v_fma_f64", w);
    run<1, 1>("v_mul_f64", w);
    run<2, 1>("v_mul_f64", w);
    run<4, 1>("v_mul_f64", w);
    run<1, 2>("v_add_f64", w);
    run<2, 2>("v_add_f64", w);
    run<4, 2>("v_add_f64", w);
    run<1, 3>("dpp_shift+add", w);
    run<2, 3>("dpp_shift+add", w);
    run<4, 3>("dpp_shift+add", w);
    run<1, 4>("cmp+cndmask+add", w);
    run<2, 4>("cmp+cndmask+add", w);
    run<4, 4>("cmp+cndmask+add", w);
  }
  return 0;
}
