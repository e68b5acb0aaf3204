// Latency microbenchmarks for the serial potrf chain (one wave, s_memtime around unrolled loops).
// hipcc -O3 --offload-arch=gfx950 tools/lat_micro.hip -o tools/lat_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 256
#define TICK(t, x) asm volatile("s_nop 4\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x) :: "memory")
__device__ __forceinline__ double bc(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__global__ void k(double *out, unsigned long long *t, double a, double b)
{
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  double x = a + lane * 1e-9;
  unsigned long long t0, t1;
  // 1: dependent fma
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < N; i++) x = fma(x, a, b);
  TICK(t1, x); if (lane == 0) t[0] = t1 - t0;
  // 2: 4 independent chains
  double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3;
  TICK(t0, y0);
#pragma unroll
  for (int i = 0; i < N / 4; i++) { y0 = fma(y0, a, b); y1 = fma(y1, a, b); y2 = fma(y2, a, b); y3 = fma(y3, a, b); }
  y0 += y1 + y2 + y3;
  TICK(t1, y0); if (lane == 0) t[1] = t1 - t0;
  x = y0 + y1 + y2 + y3;
  // 3: dependent rsq
  x = fabs(x) + 1.0;
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; i++) x = __builtin_amdgcn_rsq(x);
  TICK(t1, x); if (lane == 0) t[2] = t1 - t0;
  // 4: readlane round trip + fma
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; i++) x = fma(x, bc(x, (i & 31)), b);
  TICK(t1, x); if (lane == 0) t[3] = t1 - t0;
  // 5: LDS write -> read (same wave), dependent
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; i++) { lds[lane] = x; x = lds[(lane + 1) & 63] + b; }
  TICK(t1, x); if (lane == 0) t[4] = t1 - t0;
  // 6: LDS uniform read dependent on previous value (address chase)
  lds[lane] = (double)((lane * 7 + 3) & 63);
  __syncthreads();
  int idx = 0;
  TICK(t0, idx);
#pragma unroll
  for (int i = 0; i < 64; i++) idx = (int)lds[idx];
  TICK(t1, idx); if (lane == 0) t[5] = t1 - t0;
  // 7: dependent mul
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < N; i++) x = x * a;
  TICK(t1, x); if (lane == 0) t[6] = t1 - t0;
  // 8: 8 independent fma chains
  double z[8];
#pragma unroll
  for (int q = 0; q < 8; q++) z[q] = x + q;
  TICK(t0, z[0]);
#pragma unroll
  for (int i = 0; i < N / 8; i++)
#pragma unroll
    for (int q = 0; q < 8; q++) z[q] = fma(z[q], a, b);
  z[0] += ((z[1] + z[2]) + (z[3] + z[4])) + ((z[5] + z[6]) + z[7]);
  TICK(t1, z[0]); if (lane == 0) t[7] = t1 - t0;
  for (int q = 0; q < 8; q++) x += z[q];
  // 9: dependent f32 fma for reference
  float f = (float)x;
  TICK(t0, f);
#pragma unroll
  for (int i = 0; i < N; i++) f = fmaf(f, (float)a, (float)b);
  TICK(t1, f); if (lane == 0) t[8] = t1 - t0;
  out[lane] = x + idx + f;
}
int main()
{
  double *d; unsigned long long *t; unsigned long long h[16];
  hipMalloc(&d, 64 * 8); hipMalloc(&t, 16 * 8);
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, 0.999999, 1e-7); hipDeviceSynchronize(); }
  hipMemcpy(h, t, 16 * 8, hipMemcpyDeviceToHost);
  const char *nm[] = {"dep fma f64 x256", "4 indep fma chains x256", "dep rsq f64 x64", "readlane pair + fma x64", "lds write->read x64",
                      "lds read chase x64", "dep mul f64 x256", "8 indep fma chains x256", "dep fma f32 x256"};
  const int cnt[] = {256, 256, 64, 64, 64, 64, 256, 256, 256};
  for (int i = 0; i < 9; i++) printf("%-28s %8llu cycles  %.1f / op\n", nm[i], h[i], (double)h[i] / cnt[i]);
  return 0;
}
