// one operand pair through the 3-instruction quotient on the device, every intermediate printed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(double a, double d, double *o) {
  const double y = 1.0 / d;
  const double q0 = a * y;
  const double r = __builtin_fma(-d, q0, a);
  const double q = __builtin_fma(r, y, q0);
  o[0] = y; o[1] = q0; o[2] = r; o[3] = q; o[4] = a / d;
}
int main(int argc, char **argv) {
  const double a = argc > 1 ? strtod(argv[1], 0) : -0x1.6666666666663p+40;
  const double d = argc > 2 ? strtod(argv[2], 0) : 0x1.ffffffffffffbp-12;
  double *o; hipMalloc((void **)&o, 5 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, a, d, o);
  double h[5]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("device: y %a q0 %a r %a q %a a/d %a\n", h[0], h[1], h[2], h[3], h[4]);
  const double y = 1.0 / d, q0 = a * y, r = __builtin_fma(-d, q0, a), q = __builtin_fma(r, y, q0);
  printf("host:   y %a q0 %a r %a q %a a/d %a\n", y, q0, r, q, a / d);
  return 0;
}
