// stream5.hip -- what HBM delivers for the access MIX of the one-step column update: five
// read streams and one write stream (b, wA, kappa, Area, dAkappa -> b), 16 B per lane, no
// arithmetic to speak of.  The ceiling the memory-bound regime of k_column_stream is compared
// with (the float4 copy of MI355X_MICROARCH.md is 1 read : 1 write).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream5 stream5.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k5(double2 *__restrict__ b, const double2 *__restrict__ w,
                                          const double2 *__restrict__ k, const double2 *__restrict__ a,
                                          const double2 *__restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    double2 x = b[i], y = w[i], z = k[i], u = a[i], v = d[i];
    x.x = x.x + y.x * z.x + u.x * v.x;
    x.y = x.y + y.y * z.y + u.y * v.y;
    b[i] = x;
  }
}
__global__ __launch_bounds__(256) void k1(double2 *__restrict__ o, const double2 *__restrict__ in, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    o[i] = in[i];
}

int main() {
  const size_t n = (size_t)262144 * 100 / 2;  // double2 elements per array (= 210 MB per array)
  double2 *p[5];
  for (int i = 0; i < 5; ++i) {
    hipMalloc(&p[i], n * sizeof(double2));
    hipMemset(p[i], 0, n * sizeof(double2));
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {2048, 8192, 32768}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k5, dim3(grid), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], n);
      hipEventRecord(e1);
      hipDeviceSynchronize();
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("5 reads + 1 write, grid %5d: %.1f us, %.2f TB/s\n", grid, ms * 100, 6.0 * n * 16 / (ms / 10 * 1e-3) / 1e12);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k1, dim3(grid), dim3(256), 0, 0, p[0], p[1], n);
      hipEventRecord(e1);
      hipDeviceSynchronize();
    }
    hipEventElapsedTime(&ms, e0, e1);
    printf("1 read  + 1 write, grid %5d: %.1f us, %.2f TB/s\n", grid, ms * 100, 2.0 * n * 16 / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
