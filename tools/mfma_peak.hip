// Calibration: sustained v_mfma_f64_16x16x4_f64 rate with operands in registers (no memory),
// 1 or 2 waves per SIMD, random (non-zero) data.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double *out, int iters, double seed)
{
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){seed + i, 1.0 - i, 0.5 * i, 0.25};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
  double *d; hipMalloc(&d, 8 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; blocks_per_cu++) {
    int grid = 256 * blocks_per_cu, iters = 20000;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<16>, dim3(grid), dim3(256), 0, 0, d, iters, 1.5);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * 4 * iters * 16 * 2048.0;
      printf("blocks/CU=%d waves/SIMD=%d: %.3f ms  %.2f TFLOP/s\n", blocks_per_cu, blocks_per_cu, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
