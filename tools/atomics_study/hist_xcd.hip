// Developer microbenchmark (round 3): what bounds the one-atomic-per-point histogram of sort.hip (424 us for 10^7 points,
// 156k cells)?  Variants: agent-scope atomics on one shared table (the shipped kernel), agent scope on a table private
// to the XCD of the issuing workgroup (HW_REG_XCC_ID), workgroup scope on the private table (executes in the local L2).
//   hipcc -O3 --offload-arch=gfx950 hist_xcd.hip -o /tmp/hist_xcd && /tmp/hist_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | (3 << 11)) & 7u; }

__device__ __forceinline__ unsigned cell_hash(size_t k, unsigned ncell)
{
  unsigned long long z = (k + 1) * 0x9E3779B97F4A7C15ull;
  z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
  return (unsigned)(z % ncell);
}

template <int MODE>
__global__ void __launch_bounds__(256) hist_kernel(size_t m, unsigned ncell, unsigned *count, unsigned *slot, unsigned *xcd_seen)
{
  const unsigned x = MODE == 0 ? 0u : xcc_id();
  if (threadIdx.x == 0 && xcd_seen) atomicOr(&xcd_seen[blockIdx.x & 1023], 1u << x);
  unsigned *tab = count + (size_t)x * ncell;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
    const unsigned c = cell_hash(k, ncell);
    unsigned s;
    if (MODE == 2) s = __hip_atomic_fetch_add(&tab[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 3) s = __hip_atomic_fetch_add(&tab[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else s = __hip_atomic_fetch_add(&tab[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot[k] = (x << 29) | s;
  }
}

// no atomics at all: the memory traffic of the kernel alone
__global__ void __launch_bounds__(256) noatomic_kernel(size_t m, unsigned ncell, unsigned *slot)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) slot[k] = cell_hash(k, ncell);
}

int main()
{
  const size_t m = 10000000; const unsigned ncell = 156816;
  unsigned *count, *slot, *seen;
  CK(hipMalloc(&count, (size_t)8 * ncell * 4)); CK(hipMalloc(&slot, m * 4)); CK(hipMalloc(&seen, 1024 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned> h((size_t)8 * ncell), hs(m);
  for (int mode = -1; mode < 4; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipMemset(count, 0, (size_t)8 * ncell * 4)); CK(hipMemset(seen, 0, 1024 * 4));
      CK(hipEventRecord(e0));
      if (mode == -1) hipLaunchKernelGGL(noatomic_kernel, dim3(2048), dim3(256), 0, 0, m, ncell, slot);
      if (mode == 0) hipLaunchKernelGGL(hist_kernel<0>, dim3(2048), dim3(256), 0, 0, m, ncell, count, slot, seen);
      if (mode == 1) hipLaunchKernelGGL(hist_kernel<1>, dim3(2048), dim3(256), 0, 0, m, ncell, count, slot, seen);
      if (mode == 2) hipLaunchKernelGGL(hist_kernel<2>, dim3(2048), dim3(256), 0, 0, m, ncell, count, slot, seen);
      if (mode == 3) hipLaunchKernelGGL(hist_kernel<3>, dim3(2048), dim3(256), 0, 0, m, ncell, count, slot, seen);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    if (mode < 0) { printf("no atomics (hash + 4-byte store): %.1f us\n", best * 1e3); continue; }
    CK(hipMemcpy(h.data(), count, h.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), slot, m * 4, hipMemcpyDeviceToHost));
    unsigned long long total = 0; for (unsigned v : h) total += v;
    // every (xcd, cell, slot) triple must be unique: count how many slots >= the table's final count (lost updates show as duplicates / overflow)
    size_t bad = 0;
    std::vector<unsigned> chk((size_t)8 * ncell, 0);
    for (size_t k = 0; k < m; k++) {
      unsigned long long z = (k + 1) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
      const unsigned c = (unsigned)(z % ncell), x = hs[k] >> 29, s = hs[k] & 0x1fffffffu;
      if (s >= h[(size_t)x * ncell + c]) bad++;
      chk[(size_t)x * ncell + c] += 1;
    }
    size_t mism = 0; for (size_t i = 0; i < chk.size(); i++) if (chk[i] != h[i]) mism++;
    std::vector<unsigned> sn(1024); CK(hipMemcpy(sn.data(), seen, 4096, hipMemcpyDeviceToHost));
    unsigned allx = 0; int multi = 0; for (int i = 0; i < 1024; i++) { allx |= sn[i]; if (sn[i] & (sn[i] - 1)) multi++; }
    printf("mode %d (%s): %.1f us  total %llu (want %zu)  slots out of range %zu  tables differing from recount %zu  xcd mask 0x%x, blocks (mod 1024) seen on >1 xcd: %d\n",
           mode, mode == 0 ? "agent scope, shared table" : mode == 1 ? "agent scope, per-XCD table" : mode == 2 ? "workgroup scope, per-XCD table" : "wavefront scope, per-XCD table",
           best * 1e3, total, m, bad, mism, allx, multi);
  }
  return 0;
}
