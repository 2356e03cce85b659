// One-shot gradient all-reduce over peer-mapped memory (hipIpc / xGMI) for the latency-bound collectives of the data-parallel PPO
// update: 16 + 2 all-reduces of 1.7 MB / 1.05 MB per 2.6 ms cycle, each on the critical path
//   minibatch gradients -> average over ranks -> global-norm clip + Adam -> next minibatch's forward.
// A library all-reduce is its own launch (or several) plus a separate pass for the gradient norm; here ONE launch per rank
//   1. copies the rank's flat gradient into its exchange slot (fine-grained device memory every peer has mapped through hipIpc),
//      fences at system scope and -- the last workgroup to finish -- publishes the step's epoch in the rank's flag word;
//   2. waits (bounded) until every peer's flag has reached the epoch, acquires at system scope;
//   3. sums the world's exchange slots element-wise IN RANK ORDER (every rank computes bit-identical averages, so the replicas
//      cannot drift), scales by 1 / world, writes the average back over the gradient and leaves the 64 partial sums of squares
//      mansy_clip_grad_adam wants (have_sumsq = 1: the separate norm launch disappears).
// Each rank reads (world - 1) x n floats straight from its peers' HBM over the point-to-point links -- no ring, no intermediate
// hops: 7 x 1.7 MB at 8 ranks, the seven links in parallel.  Two slots alternate with the epoch: a rank overwrites slot e & 1 at
// epoch e + 2 only after it has seen every peer at epoch e + 1, i.e. after every peer's launch of epoch e -- the last reader of
// that slot -- has completed.
// Placement-independent and bounded: the wait gives up after `timeout_ms`, poisons its output with NaN and raises the context's
// sticky error (mansy_xg_status), it never hangs the queue.  Grid = 64 workgroups of 256 threads, so the launches of all ranks
// are co-resident even when they share one GPU (the functional test: several processes on one device).
#include <vector>
#include "mansy_kernels.h"
#include "../../include/mansy_hip.h"

namespace {

constexpr int XG_BLOCKS = MANSY_CLIP_SCRATCH_DOUBLES;      // one partial sum of squares per workgroup
constexpr int XG_MAX_WORLD = 16;
constexpr long long XG_HEADER_FLOATS = 64;                // 256-byte header: word 0 = the published epoch

struct XgPeers { const float* data[XG_MAX_WORLD]; const unsigned* flag[XG_MAX_WORLD]; };

struct XgCtx {
  int kind = MANSY_SYNC_XG;               // first field of every sync context (include/mansy_hip.h)
  int world = 0, rank = 0, imported = 0;
  long long n = 0, n_pad = 0;
  float* own = nullptr;                   // header + 2 slots
  void* peer_base[XG_MAX_WORLD] = {};     // mapped peer allocations (own entry = own)
  unsigned epoch = 0;
  unsigned* counter = nullptr;            // workgroups of this rank that have published (monotonic)
  int* err = nullptr;                     // sticky error word: pinned HOST memory the kernels write (system-scope store) and the host reads without a sync
  int* err_dev = nullptr;                 // its device address
  double timeout_ms = 2000.0;
  int wall_khz = 100000;                  // wall_clock64 ticks at a constant rate (100 MHz on gfx9), read once at create
};

constexpr int XG_THREADS = 256;        // (1 024-thread workgroups measured 1.4-1.9x slower: profiles/r03_xg_timing.txt history)
// COPY = true (mansy_xg_allreduce_avg): the gradient arrives in ordinary device memory `g` and is copied into the rank's exchange slot first.
// COPY = false (mansy_xg_reduce_avg, round 5): the step's gradient kernels wrote straight INTO the slot (the slot IS the flat gradient buffer of
// that step: mansy_xg_slot_ptrs), so this launch only publishes, waits and sums -- no 1.7 MB copy in front of the flag.
template <bool COPY>
__global__ __launch_bounds__(XG_THREADS) void xg_allreduce_kernel(float* __restrict__ g, long long n4, XgPeers peers, float* __restrict__ own_slot,
                                                          unsigned* __restrict__ own_flag, long long slot_off, int rank, int world, unsigned parity,
                                                          float inv_world, double* __restrict__ parts, unsigned* __restrict__ counter,
                                                          int* __restrict__ err, long long timeout_ticks) {
  const long long gtid = (long long)blockIdx.x * XG_THREADS + threadIdx.x, gsize = (long long)gridDim.x * XG_THREADS;
  // The epoch of THIS launch is derived on the device (round 6: a launch replayed from a captured hipGraph has frozen arguments, an epoch passed by
  // value would be the capture's): the rank's own flag word holds the last epoch it published, E - 1, until this very launch stores E, and the
  // slots alternate with the epoch, so E is the one of those two values whose low bit is the slot this launch works on (`parity`, frozen and
  // right: a graph holds an even number of averages per context).  No other launch of this rank runs concurrently on the context (one stream).
  const unsigned seen = __hip_atomic_load(own_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned epoch = seen + (((seen ^ parity) & 1u) ? 1u : 0u);
  // 1. publish this rank's gradient
  if (COPY) {
    for (long long i0 = gtid; i0 < n4; i0 += gsize * 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long long i = i0 + (long long)u * gsize; v[u] = reinterpret_cast<const float4*>(g)[i < n4 ? i : n4 - 1]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long long i = i0 + (long long)u * gsize; if (i < n4) reinterpret_cast<float4*>(own_slot)[i] = v[u]; }
    }
  }
  __shared__ int timed_out;
  if (COPY) {
    // every storing wave drains its stores, the workgroup meets, then ONE lane releases at system scope (the release is cumulative
    // over the barrier's happens-before) -- a system-scope fence in every thread cost a cache write-back per wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      timed_out = 0;
      __atomic_thread_fence(__ATOMIC_RELEASE);                // system scope
      const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if ((prev + 1u) % (unsigned)XG_BLOCKS == 0u) __hip_atomic_store(own_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // the last of this launch's XG_BLOCKS tickets
    }
  } else {
    // The slot was written by EARLIER launches of this stream: the kernel boundary in front of this launch has already written their stores back
    // to the memory side (what makes them visible to this launch's workgroups on the other XCDs is what makes them visible to a peer reading the
    // home memory over a link).  So the publish is ONE flag store -- no copy, no per-workgroup release, no ticket (tools/xg_lab.hip: 8.2 -> 4.4 us
    // per launch at world 1); the publishing lane still releases at system scope first (its own XCD's L2: belt and braces, off the other
    // workgroups' path).
    if (threadIdx.x == 0) {
      timed_out = 0;
      if (blockIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);              // system scope
        __hip_atomic_store(own_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __syncthreads();
  // 2. wait for every peer's epoch: wave 0 polls (one lane per peer, RELAXED loads: an acquire per poll is a cache invalidate per poll), then that
  // ONE wave acquires at system scope and drains it; the workgroup's barrier orders every other wave's loads behind it (round 5: was an acquire in
  // every poll and a fence in every wave)
  if (threadIdx.x < 64) {
    if ((int)threadIdx.x < world && (int)threadIdx.x != rank) {
      const long long t0 = wall_clock64();
      while ((int)(__hip_atomic_load(peers.flag[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        if (wall_clock64() - t0 > timeout_ticks) { timed_out = 1; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                // system scope: the peers' slots as published
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const bool bad = timed_out != 0;
  // 3. reduce in rank order, average, sum of squares
  // UN elements per thread and round, every load of a round requested before the first is used (a peer's HBM over a link is a
  // microsecond away; with one element per round a thread paid that latency ceil(n4 / threads) times in sequence)
  constexpr int UN = 4;
  double sq = 0.0;
  for (long long i0 = gtid; i0 < n4; i0 += gsize * UN) {
    float4 acc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < world; ++p) {
      const float4* src = reinterpret_cast<const float4*>(peers.data[p] + slot_off);
      float4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) { const long long i = i0 + (long long)u * gsize; v[u] = src[i < n4 ? i : n4 - 1]; }
#pragma unroll
      for (int u = 0; u < UN; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long i = i0 + (long long)u * gsize;
      if (i >= n4) continue;
      float4 s = acc[u];
      s.x *= inv_world; s.y *= inv_world; s.z *= inv_world; s.w *= inv_world;
      if (bad) s.x = s.y = s.z = s.w = __builtin_nanf("");
      reinterpret_cast<float4*>(g)[i] = s;
      sq += ((double)s.x * s.x + (double)s.y * s.y) + ((double)s.z * s.z + (double)s.w * s.w);
    }
  }
  if (!parts) return;
  __shared__ double red[XG_THREADS];
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int o = XG_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = red[0];
}

}  // namespace

extern "C" {

int mansy_xg_create(long long n_floats, int world, int rank, void** ctx_out) {
  MANSY_REQUIRE(ctx_out && n_floats >= 4 && n_floats % 4 == 0 && world >= 1 && world <= XG_MAX_WORLD && rank >= 0 && rank < world,
                "xg_create: need n %% 4 == 0, 1 <= world <= %d, 0 <= rank < world", XG_MAX_WORLD);
  static_assert(sizeof(mansy_xg_handle) == sizeof(hipIpcMemHandle_t), "mansy_xg_handle must hold a hipIpcMemHandle_t");
  XgCtx* c = new XgCtx();
  c->world = world; c->rank = rank; c->n = n_floats; c->n_pad = (n_floats + 63) / 64 * 64;
  const size_t bytes = sizeof(float) * (size_t)(XG_HEADER_FLOATS + 2 * c->n_pad);
  // fine-grained: stores and flag updates are visible to the peers while the kernels run (coarse-grained device memory is only
  // coherent across agents at kernel boundaries)
  hipError_t e = hipExtMallocWithFlags((void**)&c->own, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { delete c; mansy_set_error("xg_create: hipExtMallocWithFlags(fine-grained, %zu bytes) -> %s", bytes, hipGetErrorString(e)); return MANSY_EHIP; }
  e = hipMalloc((void**)&c->counter, 256);
  if (e == hipSuccess) e = hipMemset(c->own, 0, bytes);
  if (e == hipSuccess) e = hipMemset(c->counter, 0, 256);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {                                    // nothing of a half-built context survives
    mansy_set_error("xg_create: %s", hipGetErrorString(e));
    (void)hipFree(c->own);
    if (c->counter) (void)hipFree(c->counter);
    delete c;
    return MANSY_EHIP;
  }
  // the error word lives in mapped host memory: mansy_xg_status() reads it WITHOUT synchronising the device (round 5: a device sync at the end
  // of every learn() / train_identifier() drained the queue twice per cycle)
  if (hipHostMalloc((void**)&c->err, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&c->err_dev, c->err, 0) != hipSuccess) {
    mansy_set_error("xg_create: cannot allocate the mapped error word");
    if (c->err) (void)hipHostFree(c->err);
    (void)hipFree(c->own); (void)hipFree(c->counter);
    delete c;
    return MANSY_EHIP;
  }
  *c->err = 0;
  { int dev = 0, khz = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) c->wall_khz = khz; }
  c->peer_base[rank] = c->own;
  if (world == 1) c->imported = 1;
  *ctx_out = c;
  return MANSY_OK;
}

int mansy_xg_export(void* ctx, mansy_xg_handle* out) {
  XgCtx* c = (XgCtx*)ctx;
  MANSY_REQUIRE(c && out, "xg_export: null");
  hipIpcMemHandle_t h;
  MANSY_HIP_CHECK(hipIpcGetMemHandle(&h, c->own));
  memcpy(out->bytes, &h, sizeof(h));
  return MANSY_OK;
}

int mansy_xg_import(void* ctx, const mansy_xg_handle* all) {
  XgCtx* c = (XgCtx*)ctx;
  MANSY_REQUIRE(c && all && !c->imported, "xg_import: null or already imported");
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, all[p].bytes, sizeof(h));
    MANSY_HIP_CHECK(hipIpcOpenMemHandle(&c->peer_base[p], h, hipIpcMemLazyEnablePeerAccess));
  }
  c->imported = 1;
  return MANSY_OK;
}

int mansy_xg_set_timeout_ms(void* ctx, double ms) {
  XgCtx* c = (XgCtx*)ctx;
  MANSY_REQUIRE(c && ms > 0.0, "xg_set_timeout_ms: bad arguments");
  c->timeout_ms = ms;
  return MANSY_OK;
}

// The rank's two exchange slots (n_pad floats each, fine-grained device memory the peers have mapped).  Round-5 form of a data-parallel step:
// the step's gradient kernels use slot (epoch + 1) & 1 as their flat gradient buffer (mansy_xg_next_slot), then mansy_xg_reduce_avg publishes
// it, waits for the peers and leaves the average in ordinary device memory; the clip + Adam launch zeroes the OTHER slot for the next step.
int mansy_xg_slot_ptrs(void* ctx, float** slot0, float** slot1) {
  XgCtx* c = (XgCtx*)ctx;
  MANSY_REQUIRE(c && c->kind == MANSY_SYNC_XG && slot0 && slot1, "xg_slot_ptrs: null or not a peer-memory context");
  *slot0 = c->own + XG_HEADER_FLOATS; *slot1 = c->own + XG_HEADER_FLOATS + c->n_pad;
  return MANSY_OK;
}
int mansy_xg_next_slot(void* ctx) { XgCtx* c = (XgCtx*)ctx; return (c && c->kind == MANSY_SYNC_XG) ? (int)((c->epoch + 1u) & 1u) : -1; }

static int xg_launch(XgCtx* c, float* g, long long n, double* sumsq_parts, hipStream_t stream, bool copy);

int mansy_xg_reduce_avg(void* ctx, float* g_out, long long n, double* sumsq_parts, void* stream) {
  XgCtx* c = (XgCtx*)ctx;
  MANSY_REQUIRE(c && c->kind == MANSY_SYNC_XG && g_out && c->imported, "xg_reduce_avg: not a ready peer-memory context (create -> export -> exchange handles -> import)");
  MANSY_REQUIRE(n == c->n && (reinterpret_cast<uintptr_t>(g_out) & 15) == 0, "xg_reduce_avg: n must be the context's %lld and g_out 16-byte aligned", c->n);
  return xg_launch(c, g_out, n, sumsq_parts, (hipStream_t)stream, false);
}

int mansy_xg_allreduce_avg(void* ctx, float* g, long long n, double* sumsq_parts, void* stream) {
  XgCtx* c = (XgCtx*)ctx;
  MANSY_REQUIRE(c && c->kind == MANSY_SYNC_XG && g && c->imported, "xg_allreduce_avg: not a ready peer-memory context (create -> export -> exchange handles -> import)");
  MANSY_REQUIRE(n == c->n && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "xg_allreduce_avg: n must be the context's %lld and g 16-byte aligned", c->n);
  return xg_launch(c, g, n, sumsq_parts, (hipStream_t)stream, true);
}

static int xg_launch(XgCtx* c, float* g, long long n, double* sumsq_parts, hipStream_t stream, bool copy) {
  c->epoch += 1;          // host copy: only its low bit (which slot) is used; the epoch VALUE is derived on the device
  XgPeers peers;
  for (int p = 0; p < XG_MAX_WORLD; ++p) {
    const float* base = (const float*)c->peer_base[p < c->world ? p : c->rank];
    peers.data[p] = base + XG_HEADER_FLOATS;
    peers.flag[p] = reinterpret_cast<const unsigned*>(base);
  }
  const long long slot_off = (long long)(c->epoch & 1u) * c->n_pad;
  const long long ticks = (long long)(c->timeout_ms * (double)c->wall_khz);      // (the rate is read once at create: two runtime calls per average were host time on a latency-bound cycle)
  if (copy)
    MANSY_LAUNCH(xg_allreduce_kernel<true>, dim3(XG_BLOCKS), dim3(XG_THREADS), 0, stream, g, n / 4, peers, c->own + XG_HEADER_FLOATS + slot_off,
                       reinterpret_cast<unsigned*>(c->own), slot_off, c->rank, c->world, c->epoch & 1u, 1.0f / (float)c->world, sumsq_parts, c->counter,
                       c->err_dev, ticks);
  else
    MANSY_LAUNCH(xg_allreduce_kernel<false>, dim3(XG_BLOCKS), dim3(XG_THREADS), 0, stream, g, n / 4, peers, c->own + XG_HEADER_FLOATS + slot_off,
                       reinterpret_cast<unsigned*>(c->own), slot_off, c->rank, c->world, c->epoch & 1u, 1.0f / (float)c->world, sumsq_parts, c->counter,
                       c->err_dev, ticks);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// 0: every wait of the launches that have COMPLETED so far met its peers; MANSY_EHIP: one timed out (the outputs of that call are NaN and stay
// NaN through the optimiser: nothing trains on silently).  Does not synchronise: launches still in flight are covered by the next call (the
// word is sticky) -- call it after a synchronisation point of your own for a final verdict.
int mansy_xg_status(void* ctx) {
  XgCtx* c = (XgCtx*)ctx;
  MANSY_REQUIRE(c, "xg_status: null");
  const int e = __atomic_load_n(c->err, __ATOMIC_ACQUIRE);
  if (e) { mansy_set_error("xg: a peer did not publish its gradient within %.0f ms", c->timeout_ms); return MANSY_EHIP; }
  return MANSY_OK;
}

int mansy_xg_destroy(void* ctx) {
  XgCtx* c = (XgCtx*)ctx;
  if (!c) return MANSY_OK;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && c->peer_base[p]) (void)hipIpcCloseMemHandle(c->peer_base[p]);
  if (c->own) (void)hipFree(c->own);
  if (c->counter) (void)hipFree(c->counter);
  if (c->err) (void)hipHostFree(c->err);
  delete c;
  return MANSY_OK;
}

}  // extern "C"
