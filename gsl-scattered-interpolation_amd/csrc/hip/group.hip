/*
 * group.hip -- device groups: the multi-GPU side of the C-ABI (north star:
 * "evaluation targets shard across the 8 GPUs of one node with an RCCL broadcast of the weight
 * vector over xGMI").
 *
 * The reference is single-process, single-device-less C: nothing to mirror.
 * The path shards by evaluation targets only, so a group is
 *     one gsl_sinterp_hip_ctx per device, each with a private non-blocking stream (one host
 *     thread drives all devices: H2D / sweep / D2H of every shard are enqueued back to back and
 *     run concurrently), plus
 *     ONE collective: replication of the model (RBF: centres + weights; barycentric: packed DAG
 *     records + leaf table) from member 0 -- ncclBroadcast inside ncclGroupStart/End on a
 *     communicator made by ncclCommInitAll (single process, all local devices).
 * No reduction, no all-to-all.  The factorisation runs on member 0 only ("replicas only").
 *
 * RCCL (librccl.so.1, 570 MB) is bound lazily with dlopen the first time a group with more than
 * one distinct device is created, so single-GPU users never load it.  A device list may name the
 * same ordinal more than once (tests on a one-GPU box exercise the shard / gather logic that way);
 * RCCL refuses duplicate devices, so such groups replicate with hipMemcpyPeerAsync instead, ordered
 * by events -- GSL_SINTERP_NO_RCCL=1 forces that transport for any group.
 */
#include "common.h"
#include <new>
#include <dlfcn.h>
#include <stdlib.h>

#define GROUP_MAX 64

/* the slice of rccl.h this file needs (types are ABI-stable: opaque comm pointer, enums) */
typedef struct ncclComm *sk_ncclComm_t;
typedef int sk_ncclResult_t;
enum { SK_NCCL_INT8 = 0 };
typedef sk_ncclResult_t (*fn_ncclCommInitAll)(sk_ncclComm_t *, int, const int *);
typedef sk_ncclResult_t (*fn_ncclCommDestroy)(sk_ncclComm_t);
typedef sk_ncclResult_t (*fn_ncclGroupStart)(void);
typedef sk_ncclResult_t (*fn_ncclGroupEnd)(void);
typedef sk_ncclResult_t (*fn_ncclBroadcast)(const void *, void *, size_t, int, int, sk_ncclComm_t, hipStream_t);
typedef const char *(*fn_ncclGetErrorString)(sk_ncclResult_t);

struct gsl_sinterp_hip_group {
  int n;
  int devices[GROUP_MAX];
  gsl_sinterp_hip_ctx *ctx[GROUP_MAX];
  int use_rccl;
  void *rccl;
  sk_ncclComm_t comm[GROUP_MAX];
  fn_ncclCommInitAll CommInitAll;
  fn_ncclCommDestroy CommDestroy;
  fn_ncclGroupStart GroupStart;
  fn_ncclGroupEnd GroupEnd;
  fn_ncclBroadcast Broadcast;
  fn_ncclGetErrorString GetErrorString;
  hipEvent_t ev_src;            /* peer-copy transport: "model ready / copies enqueued" on member 0's stream */
  char err[256];
};

extern "C" void gsl_sinterp_hip_shard_bounds(size_t m_total, int world, int rank, size_t *first, size_t *count)
{
  /* contiguous ceil-sized shards; the last ones may be short or empty; every target exactly once
     (the rule of gsl-scattered-interpolation_amd/sharding.py, used by bench.py's process-per-GPU runs) */
  const size_t w = world > 0 ? (size_t)world : 1, r = rank > 0 ? (size_t)rank : 0;
  const size_t per = (m_total + w - 1) / w;
  const size_t f = r * per < m_total ? r * per : m_total;
  const size_t c = m_total - f < per ? m_total - f : per;
  if (first) *first = f;
  if (count) *count = c;
}

static int bind_rccl(gsl_sinterp_hip_group *g)
{
  const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  for (size_t i = 0; i < sizeof names / sizeof names[0] && !g->rccl; i++) g->rccl = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!g->rccl) { snprintf(g->err, sizeof g->err, "dlopen(librccl.so.1): %s", dlerror()); return ST_EFAILED; }
  g->CommInitAll = (fn_ncclCommInitAll)dlsym(g->rccl, "ncclCommInitAll");
  g->CommDestroy = (fn_ncclCommDestroy)dlsym(g->rccl, "ncclCommDestroy");
  g->GroupStart = (fn_ncclGroupStart)dlsym(g->rccl, "ncclGroupStart");
  g->GroupEnd = (fn_ncclGroupEnd)dlsym(g->rccl, "ncclGroupEnd");
  g->Broadcast = (fn_ncclBroadcast)dlsym(g->rccl, "ncclBroadcast");
  g->GetErrorString = (fn_ncclGetErrorString)dlsym(g->rccl, "ncclGetErrorString");
  if (!g->CommInitAll || !g->CommDestroy || !g->GroupStart || !g->GroupEnd || !g->Broadcast) {
    snprintf(g->err, sizeof g->err, "librccl: missing symbol");
    return ST_EFAILED;
  }
  return ST_SUCCESS;
}

extern "C" void gsl_sinterp_hip_group_destroy(gsl_sinterp_hip_group *g)
{
  if (!g) return;
  for (int i = 0; i < g->n; i++) {
    if (g->use_rccl && g->comm[i] && g->CommDestroy) { (void)hipSetDevice(g->devices[i]); (void)g->CommDestroy(g->comm[i]); }
  }
  if (g->ev_src) { (void)hipSetDevice(g->devices[0]); (void)hipEventDestroy(g->ev_src); }
  for (int i = 0; i < g->n; i++) gsl_sinterp_hip_ctx_destroy(g->ctx[i]);
  /* the RCCL handle stays open for the life of the process (its teardown at dlclose is not re-entrant) */
  delete g;
}

extern "C" int gsl_sinterp_hip_group_create(gsl_sinterp_hip_group **out, const int *devices, int n)
{
  if (!out) return ST_EFAULT;
  *out = NULL;
  if (n < 1 || n > GROUP_MAX || !devices) return ST_EINVAL;
  const int visible = gsl_sinterp_hip_device_count();
  for (int i = 0; i < n; i++) if (devices[i] < 0 || devices[i] >= visible) return ST_EFAILED;
  gsl_sinterp_hip_group *g = new (std::nothrow) gsl_sinterp_hip_group();
  if (!g) return ST_ENOMEM;
  memset(g, 0, sizeof *g);
  g->n = n;
  bool distinct = true;
  for (int i = 0; i < n; i++) {
    g->devices[i] = devices[i];
    for (int j = 0; j < i; j++) distinct = distinct && devices[j] != devices[i];
  }
  for (int i = 0; i < n; i++) {
    int st = gsl_sinterp_hip_ctx_create(&g->ctx[i], devices[i], NULL);
    if (!st && n > 1) st = gsl_sinterp_hip_ctx_own_stream(g->ctx[i]);   /* devices progress independently */
    if (st) { gsl_sinterp_hip_group_destroy(g); return st; }
  }
  const char *no = getenv("GSL_SINTERP_NO_RCCL");
  const char *force1 = getenv("GSL_SINTERP_RCCL_SINGLE");          /* exercise the RCCL binding on a one-GPU box */
  g->use_rccl = distinct && !(no && no[0] == '1') && (n > 1 || (force1 && force1[0] == '1'));
  if (g->use_rccl) {
    int st = bind_rccl(g);
    if (!st) {
      const sk_ncclResult_t r = g->CommInitAll(g->comm, n, g->devices);
      if (r != 0) {
        snprintf(g->err, sizeof g->err, "ncclCommInitAll: %s", g->GetErrorString ? g->GetErrorString(r) : "error");
        st = ST_EFAILED;
      }
    }
    if (st) {
      /* no silent change of transport: the caller asked for a multi-GPU group and RCCL is the contract */
      fprintf(stderr, "gsl_sinterp: %s\n", g->err);
      gsl_sinterp_hip_group_destroy(g);
      return st;
    }
  } else if (n > 1) {
    (void)hipSetDevice(g->devices[0]);
    if (hipEventCreateWithFlags(&g->ev_src, hipEventDisableTiming) != hipSuccess) { gsl_sinterp_hip_group_destroy(g); return ST_EFAILED; }
    for (int i = 1; i < n; i++) {
      if (g->devices[i] == g->devices[0]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g->devices[0], g->devices[i]) == hipSuccess && can) {
        (void)hipSetDevice(g->devices[0]);
        hipError_t e = hipDeviceEnablePeerAccess(g->devices[i], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
    }
  }
  *out = g;
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_group_size(const gsl_sinterp_hip_group *g) { return g ? g->n : 0; }
extern "C" int gsl_sinterp_hip_group_device(const gsl_sinterp_hip_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->devices[i] : -1; }
extern "C" gsl_sinterp_hip_ctx *gsl_sinterp_hip_group_ctx(gsl_sinterp_hip_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->ctx[i] : NULL; }
extern "C" const char *gsl_sinterp_hip_group_transport(const gsl_sinterp_hip_group *g)
{
  if (!g) return "none";
  return g->use_rccl ? "rccl" : (g->n > 1 ? "peer-copy" : "none");
}
extern "C" const char *gsl_sinterp_hip_group_last_error(const gsl_sinterp_hip_group *g) { return g ? g->err : "no group"; }

/* d_bufs[i] = buffer of `bytes` bytes on member i; member 0's content is replicated into all others.
   Enqueued on the members' streams: ordered after whatever produced d_bufs[0] on member 0's stream and
   before whatever member i enqueues next. */
extern "C" int gsl_sinterp_hip_group_broadcast(gsl_sinterp_hip_group *g, void *const *d_bufs, size_t bytes)
{
  if (!g || !d_bufs) return ST_EFAULT;
  if (g->n == 1 && !g->use_rccl) return ST_SUCCESS;
  if (bytes == 0) return ST_SUCCESS;
  if (g->use_rccl) {
    sk_ncclResult_t r = g->GroupStart();
    for (int i = 0; i < g->n && r == 0; i++) {
      if (hipSetDevice(g->devices[i]) != hipSuccess) { (void)g->GroupEnd(); return ST_EFAILED; }
      r = g->Broadcast(d_bufs[i], d_bufs[i], bytes, SK_NCCL_INT8, 0, g->comm[i], g->ctx[i]->stream);
    }
    const sk_ncclResult_t r2 = g->GroupEnd();
    if (r != 0 || r2 != 0) {
      snprintf(g->err, sizeof g->err, "ncclBroadcast: %s", g->GetErrorString ? g->GetErrorString(r ? r : r2) : "error");
      return ST_EFAILED;
    }
    return ST_SUCCESS;
  }
  /* peer copies on member 0's stream, then every member's stream waits for them */
  hipStream_t s0 = g->ctx[0]->stream;
  if (hipSetDevice(g->devices[0]) != hipSuccess) return ST_EFAILED;
  for (int i = 1; i < g->n; i++) {
    hipError_t e = (g->devices[i] == g->devices[0])
                       ? hipMemcpyAsync(d_bufs[i], d_bufs[0], bytes, hipMemcpyDeviceToDevice, s0)
                       : hipMemcpyPeerAsync(d_bufs[i], g->devices[i], d_bufs[0], g->devices[0], bytes, s0);
    if (e != hipSuccess) { snprintf(g->err, sizeof g->err, "peer copy to member %d: %s", i, hipGetErrorString(e)); return ST_EFAILED; }
  }
  if (hipEventRecord(g->ev_src, s0) != hipSuccess) return ST_EFAILED;
  for (int i = 1; i < g->n; i++) {
    if (hipSetDevice(g->devices[i]) != hipSuccess) return ST_EFAILED;
    if (hipStreamWaitEvent(g->ctx[i]->stream, g->ev_src, 0) != hipSuccess) return ST_EFAILED;
  }
  return ST_SUCCESS;
}

/* asynchronous copies on the context's stream (the group driver overlaps the shards of all devices and
   synchronises once at the end; host buffers must stay untouched until gsl_sinterp_hip_sync) */
extern "C" int gsl_sinterp_hip_h2d_async(gsl_sinterp_hip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  if (!bytes) return ST_SUCCESS;
  HIP_OK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_d2h_async(gsl_sinterp_hip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  if (!bytes) return ST_SUCCESS;
  HIP_OK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return ST_SUCCESS;
}

/* Copy pipe of ONE context (round 4): an upload and a download stream beside the context's stream, so that a host batch
   cut into chunks overlaps  H2D of chunk i+1 | sweep of chunk i | D2H of chunk i-1  (full-duplex PCIe).  upload: the
   copy runs on the upload stream and everything enqueued on the context's stream AFTERWARDS waits for it; download:
   the copy runs on the download stream after everything enqueued on the context's stream SO FAR.  Host buffers must be
   pinned (gsl_sinterp_hip_host_alloc) and stay untouched until gsl_sinterp_hip_pipe_sync. */
#define PIPE_EVENTS 64
struct gsl_sinterp_hip_pipe {
  gsl_sinterp_hip_ctx *ctx;
  hipStream_t up, down;
  hipEvent_t ev[PIPE_EVENTS];
  int n_ev, used;
};

extern "C" void gsl_sinterp_hip_pipe_destroy(gsl_sinterp_hip_pipe *p)
{
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  if (p->up) { (void)hipStreamSynchronize(p->up); (void)hipStreamDestroy(p->up); }
  if (p->down) { (void)hipStreamSynchronize(p->down); (void)hipStreamDestroy(p->down); }
  for (int i = 0; i < p->n_ev; i++) (void)hipEventDestroy(p->ev[i]);
  delete p;
}

extern "C" int gsl_sinterp_hip_pipe_create(gsl_sinterp_hip_ctx *ctx, gsl_sinterp_hip_pipe **out)
{
  if (!ctx || !out) return ST_EFAULT;
  *out = NULL;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  gsl_sinterp_hip_pipe *p = new (std::nothrow) gsl_sinterp_hip_pipe();
  if (!p) return ST_ENOMEM;
  p->ctx = ctx; p->up = p->down = NULL; p->n_ev = 0; p->used = 0;
  if (hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking) != hipSuccess) {
    gsl_sinterp_hip_pipe_destroy(p);
    return sinterp_fail(ctx, ST_EFAILED, "pipe: stream", hipSuccess, __FILE__, __LINE__);
  }
  *out = p;
  return ST_SUCCESS;
}

static int pipe_event(gsl_sinterp_hip_pipe *p, hipEvent_t *ev)
{
  gsl_sinterp_hip_ctx *ctx = p->ctx;
  if (p->used >= PIPE_EVENTS) return sinterp_fail(ctx, ST_EFAILED, "pipe: more than 64 copies between two syncs", hipSuccess, __FILE__, __LINE__);
  if (p->used >= p->n_ev) { HIP_OK(ctx, hipEventCreateWithFlags(&p->ev[p->n_ev], hipEventDisableTiming)); p->n_ev++; }
  *ev = p->ev[p->used++];
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_pipe_upload(gsl_sinterp_hip_pipe *p, void *d_dst, const void *h_src, size_t bytes)
{
  if (!p) return ST_EFAULT;
  gsl_sinterp_hip_ctx *ctx = p->ctx;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  if (!bytes) return ST_SUCCESS;
  hipEvent_t ev;
  int st = pipe_event(p, &ev);
  if (st) return st;
  HIP_OK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, p->up));
  HIP_OK(ctx, hipEventRecord(ev, p->up));
  HIP_OK(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
  return ST_SUCCESS;
}

/* mark: remember "everything enqueued on the context's stream so far" (an event); *mark >= 0 on success */
extern "C" int gsl_sinterp_hip_pipe_mark(gsl_sinterp_hip_pipe *p, int *mark)
{
  if (!p || !mark) return ST_EFAULT;
  gsl_sinterp_hip_ctx *ctx = p->ctx;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  hipEvent_t ev;
  int st = pipe_event(p, &ev);
  if (st) return st;
  HIP_OK(ctx, hipEventRecord(ev, ctx->stream));
  *mark = p->used - 1;
  return ST_SUCCESS;
}

/* mark < 0: behind everything enqueued on the context's stream so far; else behind that mark only */
extern "C" int gsl_sinterp_hip_pipe_download(gsl_sinterp_hip_pipe *p, int mark, void *h_dst, const void *d_src, size_t bytes)
{
  if (!p) return ST_EFAULT;
  gsl_sinterp_hip_ctx *ctx = p->ctx;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  if (!bytes) return ST_SUCCESS;
  if (mark < 0) { int st = gsl_sinterp_hip_pipe_mark(p, &mark); if (st) return st; }
  if (mark >= p->used) return sinterp_fail(ctx, ST_EINVAL, "pipe: stale mark", hipSuccess, __FILE__, __LINE__);
  HIP_OK(ctx, hipStreamWaitEvent(p->down, p->ev[mark], 0));
  HIP_OK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, p->down));
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_pipe_sync(gsl_sinterp_hip_pipe *p)
{
  if (!p) return ST_EFAULT;
  gsl_sinterp_hip_ctx *ctx = p->ctx;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  hipError_t e1 = hipStreamSynchronize(p->up), e2 = hipStreamSynchronize(ctx->stream), e3 = hipStreamSynchronize(p->down);
  p->used = 0;
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
    return sinterp_fail(ctx, ST_EFAILED, "pipe: synchronize", e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3), __FILE__, __LINE__);
  return ST_SUCCESS;
}

/* number of negative entries of d_v[0 .. m) (leaf / triangle indices: -1 = outside) -- one 8-byte read-back instead of a
   host pass over m indices; synchronises the context's stream */
__global__ void __launch_bounds__(256)
count_negative_kernel(const int *__restrict__ v, size_t m, unsigned long long *__restrict__ out)
{
  unsigned long long c = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) c += v[k] < 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

extern "C" int gsl_sinterp_hip_count_negative(gsl_sinterp_hip_ctx *ctx, const int *d_v, size_t m, long long *h_count)
{
  REQUIRE(ctx, ctx != NULL && h_count != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  *h_count = 0;
  if (!m) return ST_SUCCESS;
  unsigned long long *d_c = (unsigned long long *)((char *)ctx->d_scratch + 512);
  HIP_OK(ctx, hipMemsetAsync(d_c, 0, sizeof(unsigned long long), ctx->stream));
  size_t blocks = (m + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(count_negative_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_v, m, d_c);
  LAUNCH_CHECK(ctx);
  unsigned long long h = 0;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(&h, d_c, sizeof h, hipMemcpyDeviceToHost));
  *h_count = (long long)h;
  return ST_SUCCESS;
}

/* pinned host staging (hipHostMalloc): async copies from pageable memory are staged through a bounce
   buffer by the runtime and serialise; the group driver stages its shards here instead */
extern "C" int gsl_sinterp_hip_host_alloc(void **h_ptr, size_t bytes)
{
  if (!h_ptr) return ST_EFAULT;
  *h_ptr = NULL;
  return hipHostMalloc(h_ptr, bytes ? bytes : 8, hipHostMallocPortable) == hipSuccess ? ST_SUCCESS : ST_ENOMEM;
}
extern "C" void gsl_sinterp_hip_host_free(void *h_ptr) { if (h_ptr) (void)hipHostFree(h_ptr); }
