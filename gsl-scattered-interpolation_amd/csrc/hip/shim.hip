/*
 * shim.hip -- context, memory, timing and synthetic-cloud entry points of the
 * C-ABI (include/gsl_sinterp_hip.h).  The reference has no device boundary at
 * all; this file is that boundary.
 */
#include "common.h"
#include <stdlib.h>

extern "C" int gsl_sinterp_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int gsl_sinterp_hip_ctx_create(gsl_sinterp_hip_ctx **out, int device, void *stream)
{
  if (!out) return ST_EFAULT;
  *out = NULL;
  int n = gsl_sinterp_hip_device_count();
  if (n <= 0 || device < 0 || device >= n) return ST_EFAILED;
  gsl_sinterp_hip_ctx *ctx = new (std::nothrow) gsl_sinterp_hip_ctx();
  if (!ctx) return ST_ENOMEM;
  memset(ctx, 0, sizeof *ctx);
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return ST_EFAILED; }
  /* NULL = the device's default (null) stream: ordered after everything the
     caller already enqueued there (hipMemcpy, torch's default stream, ...). */
  ctx->stream = (hipStream_t)stream;
  ctx->owns_stream = 0;
  {
    const char *e = getenv("GSL_SINTERP_NO_GRAPH");
    ctx->use_graphs = !(e && e[0] == '1');
  }
  ctx->scratch_bytes = 4096;
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipMalloc(&ctx->d_scratch, ctx->scratch_bytes) != hipSuccess) {
    gsl_sinterp_hip_ctx_destroy(ctx);
    return ST_EFAILED;
  }
  *out = ctx;
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_ctx_own_stream(gsl_sinterp_hip_ctx *ctx)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  if (ctx->owns_stream) return ST_SUCCESS;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  hipStream_t s;
  HIP_OK(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  ctx->stream = s;
  ctx->owns_stream = 1;
  return ST_SUCCESS;
}

extern "C" void gsl_sinterp_hip_ctx_destroy(gsl_sinterp_hip_ctx *ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 4; i++)
    if (ctx->graph[i].exec) (void)hipGraphExecDestroy(ctx->graph[i].exec);
  if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->d_work) (void)hipFree(ctx->d_work);
  if (ctx->d_aux) (void)hipFree(ctx->d_aux);
  if (ctx->d_inv) (void)hipFree(ctx->d_inv);
  if (ctx->d_sort) (void)hipFree(ctx->d_sort);
  if (ctx->d_sort2) (void)hipFree(ctx->d_sort2);
  if (ctx->d_cent) (void)hipFree(ctx->d_cent);
  if (ctx->d_walk) (void)hipFree(ctx->d_walk);
  for (int i = 0; i < 4; i++) if (ctx->side_ev[i]) (void)hipEventDestroy(ctx->side_ev[i]);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->d_sk_partial) (void)hipFree(ctx->d_sk_partial);
  if (ctx->d_sk_flags) (void)hipFree(ctx->d_sk_flags);
  if (ctx->d_tf) (void)hipFree(ctx->d_tf);
  if (ctx->d_xq) (void)hipFree(ctx->d_xq);
  if (ctx->d_lu_coop) (void)hipFree(ctx->d_lu_coop);
  if (ctx->d_jumpt) (void)hipFree(ctx->d_jumpt);
  if (ctx->d_lw_a) (void)hipFree(ctx->d_lw_a);
  if (ctx->d_lw_lines) (void)hipFree(ctx->d_lw_lines);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char *gsl_sinterp_hip_last_error(const gsl_sinterp_hip_ctx *ctx)
{
  return ctx ? ctx->err : "no HIP context (no usable gfx950 device?)";
}

extern "C" int gsl_sinterp_hip_ctx_device(const gsl_sinterp_hip_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int gsl_sinterp_hip_sync(gsl_sinterp_hip_ctx *ctx)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_malloc(gsl_sinterp_hip_ctx *ctx, void **d_ptr, size_t bytes)
{
  REQUIRE(ctx, ctx != NULL && d_ptr != NULL, ST_EFAULT);
  *d_ptr = NULL;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  HIP_OK(ctx, hipMalloc(d_ptr, bytes ? bytes : 8));
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_free(gsl_sinterp_hip_ctx *ctx, void *d_ptr)
{
  if (!d_ptr) return ST_SUCCESS;
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipFree(d_ptr));
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_h2d(gsl_sinterp_hip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  if (!bytes) return ST_SUCCESS;
  HIP_OK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));   /* pageable source: make reuse of h_src safe */
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_d2h(gsl_sinterp_hip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  if (!bytes) return ST_SUCCESS;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_timer_start(gsl_sinterp_hip_ctx *ctx)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  HIP_OK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_timer_stop(gsl_sinterp_hip_ctx *ctx, float *h_ms)
{
  REQUIRE(ctx, ctx != NULL && h_ms != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  HIP_OK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_OK(ctx, hipEventSynchronize(ctx->ev1));
  HIP_OK(ctx, hipEventElapsedTime(h_ms, ctx->ev0, ctx->ev1));
  return ST_SUCCESS;
}

#include <mutex>
#define SINTERP_MAX_DEVICES 64
namespace {
struct FuncAttr { const void *func; unsigned long long done; int bytes; };
std::mutex g_attr_mutex;
FuncAttr g_attr[64];
int g_attr_n = 0;

struct DeviceChain { std::mutex mu; hipEvent_t ev; bool have; };
DeviceChain g_chain[SINTERP_MAX_DEVICES];
}

int sinterp_func_lds(gsl_sinterp_hip_ctx *ctx, const void *func, int bytes)
{
  std::lock_guard<std::mutex> lock(g_attr_mutex);
  const int dev = ctx->device;
  FuncAttr *a = NULL;
  for (int i = 0; i < g_attr_n; i++) if (g_attr[i].func == func) { a = &g_attr[i]; break; }
  if (!a && g_attr_n < 64) { a = &g_attr[g_attr_n++]; a->func = func; a->done = 0; a->bytes = 0; }
  if (a && dev < 64 && ((a->done >> dev) & 1ull) && a->bytes >= bytes) return ST_SUCCESS;
  HIP_OK(ctx, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (a && dev < 64) { a->done = (a->bytes == bytes ? a->done : 0ull) | (1ull << dev); a->bytes = bytes; }
  return ST_SUCCESS;
}

int sinterp_exclusive_begin(gsl_sinterp_hip_ctx *ctx)
{
  if (ctx->excl_depth++ > 0) return ST_SUCCESS;
  if (ctx->device < 0 || ctx->device >= SINTERP_MAX_DEVICES) return ST_SUCCESS;
  DeviceChain &c = g_chain[ctx->device];
  c.mu.lock();                                            /* held until _end: launches of one section are not interleaved */
  if (c.have) {
    hipError_t e = hipStreamWaitEvent(ctx->stream, c.ev, 0);
    if (e != hipSuccess) { ctx->excl_depth--; c.mu.unlock(); return sinterp_fail(ctx, ST_EFAILED, "hipStreamWaitEvent", e, __FILE__, __LINE__); }
  }
  return ST_SUCCESS;
}

int sinterp_exclusive_end(gsl_sinterp_hip_ctx *ctx)
{
  if (--ctx->excl_depth > 0) return ST_SUCCESS;
  if (ctx->device < 0 || ctx->device >= SINTERP_MAX_DEVICES) return ST_SUCCESS;
  DeviceChain &c = g_chain[ctx->device];
  hipError_t e = hipSuccess;
  if (!c.have) { e = hipEventCreateWithFlags(&c.ev, hipEventDisableTiming); c.have = (e == hipSuccess); }
  if (c.have) e = hipEventRecord(c.ev, ctx->stream);
  c.mu.unlock();
  if (e != hipSuccess) return sinterp_fail(ctx, ST_EFAILED, "exclusive section: event", e, __FILE__, __LINE__);
  return ST_SUCCESS;
}

int sinterp_graph_try_launch(gsl_sinterp_hip_ctx *ctx, int which, size_t n, size_t lda, const void *p0, const void *p1,
                             int *launched)
{
  *launched = 0;
  if (!ctx->use_graphs) return ST_SUCCESS;
  gsl_sinterp_hip_ctx::GraphSlot &g = ctx->graph[which];
  if (g.exec && g.n == n && g.lda == lda && g.p0 == p0 && g.p1 == p1 && g.work == ctx->d_work) {
    HIP_OK(ctx, hipGraphLaunch(g.exec, ctx->stream));
    *launched = 1;
  }
  return ST_SUCCESS;
}

int sinterp_capture_begin(gsl_sinterp_hip_ctx *ctx, hipStream_t *saved)
{
  *saved = ctx->stream;
  if (!ctx->use_graphs) return ST_SUCCESS;
  if (!ctx->cap_stream) HIP_OK(ctx, hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
  HIP_OK(ctx, hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal));
  ctx->stream = ctx->cap_stream;
  return ST_SUCCESS;
}

int sinterp_capture_end(gsl_sinterp_hip_ctx *ctx, hipStream_t saved, int which, size_t n, size_t lda, const void *p0,
                        const void *p1)
{
  if (!ctx->use_graphs) return ST_SUCCESS;
  ctx->stream = saved;
  hipGraph_t graph = NULL;
  HIP_OK(ctx, hipStreamEndCapture(ctx->cap_stream, &graph));
  gsl_sinterp_hip_ctx::GraphSlot &g = ctx->graph[which];
  if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = NULL; }
  hipError_t e = hipGraphInstantiate(&g.exec, graph, NULL, NULL, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { g.exec = NULL; return sinterp_fail(ctx, ST_EFAILED, "hipGraphInstantiate", e, __FILE__, __LINE__); }
  g.n = n; g.lda = lda; g.p0 = p0; g.p1 = p1; g.work = ctx->d_work;
  HIP_OK(ctx, hipGraphLaunch(g.exec, ctx->stream));
  return ST_SUCCESS;
}

int sinterp_workspace(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out)
{
  if (bytes > ctx->work_bytes) {
    if (ctx->d_work) {
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      HIP_OK(ctx, hipFree(ctx->d_work));
      ctx->d_work = NULL; ctx->work_bytes = 0;
    }
    HIP_OK(ctx, hipMalloc(&ctx->d_work, bytes));
    ctx->work_bytes = bytes;
  }
  *out = ctx->d_work;
  return ST_SUCCESS;
}

int sinterp_aux(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out)
{
  if (bytes > ctx->aux_bytes) {
    if (ctx->d_aux) {
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      HIP_OK(ctx, hipFree(ctx->d_aux));
      ctx->d_aux = NULL; ctx->aux_bytes = 0;
    }
    HIP_OK(ctx, hipMalloc(&ctx->d_aux, bytes));
    ctx->aux_bytes = bytes;
  }
  *out = ctx->d_aux;
  return ST_SUCCESS;
}

int sinterp_invbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out)
{
  if (bytes > ctx->inv_bytes) {
    if (ctx->d_inv) {
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      HIP_OK(ctx, hipFree(ctx->d_inv));
      ctx->d_inv = NULL; ctx->inv_bytes = 0;
      for (int i = 2; i < 4; i++)                      /* cached sweep graphs bake this pointer */
        if (ctx->graph[i].exec) { (void)hipGraphExecDestroy(ctx->graph[i].exec); ctx->graph[i].exec = NULL; }
    }
    HIP_OK(ctx, hipMalloc(&ctx->d_inv, bytes));
    ctx->inv_bytes = bytes;
  }
  *out = ctx->d_inv;
  return ST_SUCCESS;
}

int sinterp_sortbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out)
{
  if (bytes > ctx->sort_bytes) {
    if (ctx->d_sort) {
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      HIP_OK(ctx, hipFree(ctx->d_sort));
      ctx->d_sort = NULL; ctx->sort_bytes = 0;
    }
    HIP_OK(ctx, hipMalloc(&ctx->d_sort, bytes));
    ctx->sort_bytes = bytes;
  }
  *out = ctx->d_sort;
  return ST_SUCCESS;
}

static int grow_buf(gsl_sinterp_hip_ctx *ctx, void **buf, size_t *have, size_t bytes, void **out)
{
  if (bytes > *have) {
    if (*buf) {
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      HIP_OK(ctx, hipFree(*buf));
      *buf = NULL; *have = 0;
    }
    HIP_OK(ctx, hipMalloc(buf, bytes));
    *have = bytes;
  }
  *out = *buf;
  return ST_SUCCESS;
}
int sinterp_sortbuf2(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out) { return grow_buf(ctx, &ctx->d_sort2, &ctx->sort2_bytes, bytes, out); }
int sinterp_walkbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out) { return grow_buf(ctx, &ctx->d_walk, &ctx->walk_bytes, bytes, out); }
int sinterp_centbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out) { return grow_buf(ctx, &ctx->d_cent, &ctx->cent_bytes, bytes, out); }

/* u(k) = (splitmix64(seed ^ k) >> 11) * 2^-53, out[i] = offset + span * u(first + i)
   (same generator as oracle/oracle_synth.c so CPU and GPU see
   identical clouds without a transfer) */
__global__ void synth_unit_kernel(uint64_t seed, uint64_t first, double offset, double span,
                                  double *__restrict__ out, size_t count)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    uint64_t z = (seed ^ (first + i)) + 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z ^= z >> 31;
    double u = (double)(z >> 11) * 0x1.0p-53;
    out[i] = offset + span * u;
  }
}

extern "C" int gsl_sinterp_hip_synth_unit(gsl_sinterp_hip_ctx *ctx, uint64_t seed, uint64_t first,
                                          double offset, double span, double *d_out, size_t count)
{
  REQUIRE(ctx, ctx != NULL && d_out != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  if (!count) return ST_SUCCESS;
  size_t blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(synth_unit_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, seed, first, offset, span,
                     d_out, count);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}


/* Targets of a regular n0 x n1 grid, generated in HBM: row (i * n1 + j) = (min0 + step0 * i, min1 + step1 * j) --
   the loop of interpolation/scattered_interp_example.c:183-197 (x = min[0] + xstep * i, one multiply and one
   add, separately rounded: this file is compiled with -ffp-contract=off), so the coordinates are the bits the
   reference's host loop produces and nothing crosses PCIe on the way in. */
__global__ void grid_targets_kernel(double min0, double step0, size_t n0, double min1, double step1, size_t n1,
                                    double *__restrict__ y)
{
  const size_t total = n0 * n1, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
    const size_t i = k / n1, j = k - i * n1;
    double2 v;
    v.x = min0 + step0 * (double)i;
    v.y = min1 + step1 * (double)j;
    *reinterpret_cast<double2 *>(y + 2 * k) = v;
  }
}

extern "C" int gsl_sinterp_hip_grid_targets(gsl_sinterp_hip_ctx *ctx, double min0, double step0, size_t n0, double min1,
                                            double step1, size_t n1, double *d_y)
{
  REQUIRE(ctx, ctx != NULL && d_y != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, (((uintptr_t)d_y) & 15) == 0, ST_EINVAL);
  if (n0 == 0 || n1 == 0) return ST_SUCCESS;
  size_t blocks = (n0 * n1 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(grid_targets_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, min0, step0, n0, min1, step1, n1, d_y);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}
