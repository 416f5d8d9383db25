// Context, error text, raw device memory, timers and the host-only parameter helpers of the C ABI.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace a3d {

static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

static inline uint32_t f32_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float bits_f32(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// Monotone map f32 -> u32 (total order of finite floats).
static inline uint32_t ordered(float f) {
  uint32_t u = f32_bits(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float unordered(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return bits_f32(u);
}

float acos_gate_threshold(float thr, bool strict) {
  auto rejects = [&](float d) {
    float a = fabsf(acosf(d));
    return strict ? (a > thr) : (a >= thr);
  };
  if (!rejects(-1.0f)) return -2.0f;  // acos is largest at -1: nothing is rejected (also thr = NaN)
  if (rejects(1.0f)) return 1.0f;
  // invariant: rejects(lo) && !rejects(hi); acosf is non-increasing
  uint32_t lo = ordered(-1.0f), hi = ordered(1.0f);
  while (hi - lo > 1) {
    uint32_t mid = lo + (hi - lo) / 2;
    if (rejects(unordered(mid)))
      lo = mid;
    else
      hi = mid;
  }
  return unordered(lo);
}

a3d_status ctx_scratch(a3d_context* ctx, int which, size_t bytes, void** out) {
  if (ctx->scratch_size[which] < bytes) {
    A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));  // nothing enqueued may still use the old region
    if (ctx->scratch[which]) A3D_HIP_TRY(hipFree(ctx->scratch[which]));
    ctx->scratch[which] = nullptr;
    ctx->scratch_size[which] = 0;
    if (which == 1) ctx->grid_clean = a3d_context::GridLayoutKey{};
    const size_t grown = bytes + bytes / 4;
    A3D_HIP_TRY(hipMalloc(&ctx->scratch[which], grown));
    ctx->scratch_size[which] = grown;
  }
  *out = ctx->scratch[which];
  return A3D_OK;
}

// The idle kd-tree / Icp blocks a context keeps (up to 8 blocks, 512 MiB) go back to the device when an allocation fails
// (round 5 advisor: a process with several contexts could fail hipMalloc while holding gigabytes of them).  What used a
// block was enqueued on the context's stream: synchronise it first.
static void drop_spare_blocks(a3d_context* ctx) {
  std::vector<a3d_context::SpareBlock> spare;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    spare.swap(ctx->spare_blocks);
  }
  if (spare.empty()) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (const auto& b : spare) (void)hipFree(b.p);
}

a3d_status ctx_block_alloc(a3d_context* ctx, size_t bytes, void** out, size_t* out_bytes) {
  bytes = std::max<size_t>(bytes, 256);
  {  // (a handle may be freed — its block released — from another thread than the one that builds: same mutex as the arena pool)
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < ctx->spare_blocks.size(); ++i)
      if (ctx->spare_blocks[i].bytes >= bytes && ctx->spare_blocks[i].bytes <= 2 * bytes &&
          (best == SIZE_MAX || ctx->spare_blocks[i].bytes < ctx->spare_blocks[best].bytes))
        best = i;
    if (best != SIZE_MAX) {
      *out = ctx->spare_blocks[best].p, *out_bytes = ctx->spare_blocks[best].bytes;
      ctx->spare_blocks.erase(ctx->spare_blocks.begin() + (long)best);
      return A3D_OK;
    }
  }
  if (hipMalloc(out, bytes) != hipSuccess) {  // out of memory: give the idle spare blocks back and try once more
    (void)hipGetLastError();
    drop_spare_blocks(ctx);
    A3D_HIP_TRY(hipMalloc(out, bytes));
  }
  *out_bytes = bytes;
  return A3D_OK;
}

void ctx_block_release(a3d_context* ctx, void* p, size_t bytes) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    size_t held = bytes;
    for (const auto& b : ctx->spare_blocks) held += b.bytes;
    if (ctx->spare_blocks.size() < 8 && held <= ((size_t)512 << 20)) {
      ctx->spare_blocks.push_back({p, bytes});
      return;
    }
  }
  hipStreamSynchronize(ctx->stream);  // (what used the block was enqueued on the context's stream)
  hipFree(p);
}

a3d_status ctx_cached_table(a3d_context* ctx, const uint32_t key[4], const void* host, size_t bytes, void** out) {
  for (const auto& t : ctx->tables)
    if (!memcmp(t.key, key, sizeof(t.key))) {
      *out = t.d;
      return A3D_OK;
    }
  void* d = nullptr;
  A3D_HIP_TRY(hipMalloc(&d, bytes ? bytes : 1));
  A3D_HIP_TRY(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
  a3d_context::CachedTable t;
  memcpy(t.key, key, sizeof(t.key));
  t.d = d;
  ctx->tables.push_back(t);
  *out = d;
  return A3D_OK;
}

// Pool entries: (base, bytes); slab slices are recognised by address (inside one of the context's slabs).
static bool in_slab(const a3d_context* ctx, const void* p, size_t slab_stride_unused = 0) {
  (void)slab_stride_unused;
  for (size_t i = 0; i < ctx->arena_slabs.size(); ++i) {
    const char* b = (const char*)ctx->arena_slabs[i];
    if ((const char*)p >= b && (const char*)p < b + ctx->slab_sizes[i]) return true;
  }
  return false;
}

// Frees every pooled arena that is a single allocation (slab slices stay): called when a hipMalloc fails, so that
// memory parked in the pool is not the reason for an out-of-memory status.
static void forget_single_locked(a3d_context* ctx, void* p) {
  for (size_t i = 0; i < ctx->single_arenas.size(); ++i)
    if (ctx->single_arenas[i] == p) {
      ctx->single_arenas.erase(ctx->single_arenas.begin() + (long)i);
      return;
    }
}

static void trim_pool_locked(a3d_context* ctx) {
  for (size_t i = 0; i < ctx->arena_pool.size();) {
    if (!in_slab(ctx, ctx->arena_pool[i].first)) {
      forget_single_locked(ctx, ctx->arena_pool[i].first);
      hipFree(ctx->arena_pool[i].first);
      ctx->arena_pool.erase(ctx->arena_pool.begin() + (long)i);
    } else {
      ++i;
    }
  }
}

a3d_status ctx_arena_acquire(a3d_context* ctx, size_t bytes, DeviceArena* out) {
  // A slab (several arenas from one hipMalloc, i.e. one device-wide synchronisation) is only taken for a size that
  // has been asked for SLAB_AFTER times — one-shot builds and odd sizes pay for exactly what they use — and is capped
  // at A3D_ARENA_SLAB_MB (default 512 MiB; 0 disables slabs): a 640x480 pyramid then gets 16-arena slabs of 216 MB,
  // a 4K frame none.
  constexpr size_t SLAB_ARENAS = 16, SLAB_BUDGET = 8ull << 30;
  constexpr uint32_t SLAB_AFTER = 3;
  static const size_t slab_cap = [] {
    const char* v = getenv("A3D_ARENA_SLAB_MB");
    return (size_t)(v ? std::max(0, atoi(v)) : 512) << 20;
  }();
  const size_t padded = ((bytes + 255) / 256) * 256;
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  A3D_REQUIRE(!ctx->zombie, A3D_INVALID_PARAMETER, "this context has been destroyed");
  for (size_t i = 0; i < ctx->arena_pool.size(); ++i) {
    const size_t have = ctx->arena_pool[i].second;
    if (have >= bytes && have <= bytes + bytes / 4) {
      out->base = ctx->arena_pool[i].first;
      out->bytes = have;
      out->slab_slice = in_slab(ctx, out->base);
      ctx->arena_pool.erase(ctx->arena_pool.begin() + (long)i);
      ++ctx->live_arenas;
      return A3D_OK;
    }
  }
  uint32_t asked = 0;
  for (auto& r : ctx->arena_requests)
    if (r.first == padded) asked = ++r.second;
  if (!asked) ctx->arena_requests.emplace_back(padded, asked = 1);
  const size_t per_slab = std::min(SLAB_ARENAS, padded ? slab_cap / padded : 0);
  if (asked > SLAB_AFTER && per_slab >= 2 && ctx->slab_bytes_total + per_slab * padded <= SLAB_BUDGET) {
    void* slab = nullptr;
    if (hipMalloc(&slab, per_slab * padded) == hipSuccess) {
      ctx->arena_slabs.push_back(slab);
      ctx->slab_sizes.push_back(per_slab * padded);
      ctx->slab_bytes_total += per_slab * padded;
      for (size_t k = 1; k < per_slab; ++k) ctx->arena_pool.emplace_back((char*)slab + k * padded, padded);
      out->base = slab, out->bytes = padded, out->slab_slice = true;
      ++ctx->live_arenas;
      return A3D_OK;
    }
    (void)hipGetLastError();  // fall back to a single allocation
  }
  if (hipMalloc(&out->base, bytes) != hipSuccess) {
    (void)hipGetLastError();
    trim_pool_locked(ctx);  // give back what the pool holds (and the idle kd-tree blocks), then try once more
    if (!ctx->spare_blocks.empty()) {
      (void)hipStreamSynchronize(ctx->stream);
      for (const auto& b : ctx->spare_blocks) (void)hipFree(b.p);
      ctx->spare_blocks.clear();
    }
    A3D_HIP_TRY(hipMalloc(&out->base, bytes));
  }
  ctx->single_arenas.push_back(out->base);
  out->bytes = bytes;
  out->slab_slice = false;
  ++ctx->live_arenas;
  return A3D_OK;
}

a3d_status ctx_side_stream(a3d_context* ctx, uint32_t index, hipStream_t* out) {
  std::lock_guard<std::mutex> lock(ctx->stream_mutex);
  if (index == 2) {  // a fourth pair group takes the context's fourth stream (the copy stream: the fourth pipe)
    *out = ctx->copy_stream;
    return A3D_OK;
  }
  while (ctx->side_streams.size() <= index) {
    hipStream_t s = nullptr;
    A3D_HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, ctx->stream_priority));
    ctx->side_streams.push_back(s);
  }
  *out = ctx->side_streams[index];
  return A3D_OK;
}

void attach_fence(const a3d_device_image* im, const std::shared_ptr<UseFence>& fence) {
  DeviceArena* a = im ? im->arena : nullptr;
  if (!a) return;  // individually allocated arrays are released with hipFree, which synchronises the device
  std::lock_guard<std::mutex> lock(a->fence_mutex);
  for (const auto& f : a->fences)
    if (f == fence) return;
  a->fences.push_back(fence);
}

void ctx_arena_release(a3d_context* ctx, DeviceArena* arena) {
  {  // consumers on other streams that were enqueued without a host synchronisation must be done with the arena
    std::lock_guard<std::mutex> lock(arena->fence_mutex);
    for (auto& f : arena->fences) f->wait();
    arena->fences.clear();
  }
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    size_t pooled = 0;
    for (const auto& a : ctx->arena_pool) pooled += a.second;
    // slab slices always return to the pool; single allocations are kept up to 128 arenas / 4 GiB per context
    if (arena->slab_slice || (ctx->arena_pool.size() < 128 && pooled + arena->bytes <= (4ull << 30))) {
      ctx->arena_pool.emplace_back(arena->base, arena->bytes);
      arena->base = nullptr;
    }
  }
  if (arena->base) {
    {
      std::lock_guard<std::mutex> lock(ctx->pool_mutex);
      forget_single_locked(ctx, arena->base);
    }
    hipFree(arena->base);
  }
  arena->base = nullptr;
  bool last_of_a_destroyed_context = false;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    last_of_a_destroyed_context = --ctx->live_arenas == 0 && ctx->zombie;
  }
  // a3d_context_destroy was called while images were alive: it deferred to the last of them (this one)
  if (last_of_a_destroyed_context) ctx_destroy_now(ctx);
}

}  // namespace a3d

using namespace a3d;

extern "C" {

uint32_t a3d_abi_version(void) { return A3D_ABI_VERSION; }
const char* a3d_last_error(void) { return a3d::g_last_error; }
const char* a3d_status_string(a3d_status s) {
  switch (s) {
    case A3D_OK: return "A3D_OK";
    case A3D_INVALID_PARAMETER: return "A3D_INVALID_PARAMETER";
    case A3D_MISSING_FIELD: return "A3D_MISSING_FIELD";
    case A3D_SOLVE_FAILED: return "A3D_SOLVE_FAILED";
    case A3D_HIP_ERROR: return "A3D_HIP_ERROR";
    case A3D_NAN_IN_INPUT: return "A3D_NAN_IN_INPUT";
    case A3D_CAST_OVERFLOW: return "A3D_CAST_OVERFLOW";
  }
  return "A3D_UNKNOWN";
}

a3d_status a3d_device_count(int32_t* out_count) {
  A3D_REQUIRE(out_count, A3D_INVALID_PARAMETER, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *out_count = n;
  return A3D_OK;
}

a3d_status a3d_context_create(int32_t device_index, a3d_context** out_ctx) {
  return a3d_context_create_with_priority(device_index, 0, out_ctx);
}

a3d_status a3d_context_create_with_priority(int32_t device_index, int32_t priority, a3d_context** out_ctx) {
  return a3d_context_create_on_pipe(device_index, priority, -1, out_ctx);
}

a3d_status a3d_context_create_on_pipe(int32_t device_index, int32_t priority, int32_t main_slot, a3d_context** out_ctx) {
  A3D_REQUIRE(out_ctx, A3D_INVALID_PARAMETER, "out_ctx is null");
  // A batch runs its pair groups on three streams, a context has four, an aligner + builder pair eight, and a host
  // framework in the same process (PyTorch + RCCL) brings its own.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES
  // hardware queues (default 4) and serialises streams that share one: with RCCL in the process the three
  // groups ran one after the other (5.35 instead of 3.85 ms per step).  Only effective if this is the
  // process's first HIP call; hosts that initialise HIP earlier should export it themselves.
  setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return A3D_HIP_ERROR;
  }
  A3D_REQUIRE(device_index >= 0 && device_index < count, A3D_INVALID_PARAMETER, "device index out of range");
  A3D_HIP_TRY(hipSetDevice(device_index));
  // on any failure below, what has been created so far is released again (a3d_context_destroy tolerates the gaps)
  std::unique_ptr<a3d_context, void (*)(a3d_context*)> guard(new a3d_context(), [](a3d_context* c) { a3d_context_destroy(c); });
  a3d_context* ctx = guard.get();
  ctx->device = device_index;
  hipDeviceProp_t prop;
  A3D_HIP_TRY(hipGetDeviceProperties(&prop, device_index));
  ctx->num_cus = prop.multiProcessorCount;
  {  // priority < 0: the device's highest stream priority, > 0: its lowest, 0: the default
    int least = 0, greatest = 0;
    A3D_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = priority < 0 ? greatest : priority > 0 ? least : 0;
    ctx->stream_priority = prio;
    // A context always creates FOUR streams, in an order chosen for the hardware: the runtime gives every new stream the
    // next hardware queue, queues go round the four compute pipes in that order, and a pipe dispatches one big grid
    // at a time — two busy streams on one pipe take turns instead of overlapping.
    //  * an aligning context (priority >= 0): main, side 0, side 1 — the three pair-group streams of a batch
    //    alignment — then the copy stream: pipes 0, 1, 2 | 3.  (Created lazily, the side streams landed behind
    //    whatever the process had created in between — a second context was enough — and two groups shared a pipe:
    //    14.6 k instead of 22.5 k pairs/s.)
    //  * a builder context (priority < 0, the context that builds frames beside an aligning one): copy, side 0, side 1,
    //    then the main stream, which thereby sits on the pipe of the aligning context's (mostly idle) copy stream:
    //    its chain of short kernels is not queued behind a pair group's grids (streaming 13.8 k against 11.4 k
    //    pairs/s with the main stream on any other pipe).
    // This holds while the process's streams are created by this library, context by context; other creators shift it.
    hipStream_t four[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; ++k) {
      const hipError_t se = hipStreamCreateWithPriority(&four[k], hipStreamNonBlocking, prio);
      if (se != hipSuccess) {
        for (int j = 0; j < k; ++j) hipStreamDestroy(four[j]);
        set_error("hipStreamCreateWithPriority failed: %s", hipGetErrorString(se));
        return A3D_HIP_ERROR;
      }
    }
    //  * a further aligning context on the same GPU (a3d_context_create_on_pipe: run_odometry with alignments in flight)
    //    names the slot of its main stream, so that two lone-pair launch chains do not take turns on one pipe: with both
    //    main streams on pipe 0, two concurrent alignments took 1.84 ms each instead of 0.64.
    //  * by default the k-th ALIGNING context a process creates on a device takes slot k % 3 (0, 1, 2, 0, ...): a host
    //    that gives each of its threads a context of its own (the reference's objects are re-entrant, here a thread that
    //    wants a concurrent `align` creates its own context) then does not stack their launch chains on pipe 0.
    static std::atomic<int> aligners_created[64];
    int m = main_slot >= 0 && main_slot < 4 ? main_slot : (priority < 0 ? 3 : 0);
    if (!(main_slot >= 0 && main_slot < 4) && priority >= 0) m = aligners_created[device_index & 63]++ % 3;
    const int c = m == 3 ? 0 : 3;
    ctx->stream = four[m], ctx->copy_stream = four[c];
    for (int k = 0; k < 4; ++k)
      if (k != m && k != c) ctx->side_streams.push_back(four[k]);
  }
  A3D_HIP_TRY(hipEventCreate(&ctx->ev_start));
  A3D_HIP_TRY(hipEventCreate(&ctx->ev_stop));
  A3D_HIP_TRY(hipEventCreate(&ctx->kd_ev[0]));
  A3D_HIP_TRY(hipEventCreate(&ctx->kd_ev[1]));
  A3D_HIP_TRY(hipHostMalloc((void**)&ctx->pinned_words, (a3d_context::PINNED_WORDS + a3d_context::PINNED_EXTRA) * sizeof(uint32_t), hipHostMallocDefault));
  *out_ctx = guard.release();
  return A3D_OK;
}

a3d_status a3d_context_create_pair(int32_t device_index, a3d_context** out_aligner, a3d_context** out_builder) {
  A3D_REQUIRE(out_aligner && out_builder, A3D_INVALID_PARAMETER, "null argument");
  // created back to back on purpose: see the stream order in a3d_context_create_with_priority
  // The aligner of a pair always has its main stream in slot 0 (not the process-wide rotation of unpaired aligners): the
  // builder's main stream (slot 3) is placed against THAT, so every pair a process creates has the same pipe relation.
  A3D_TRY(a3d_context_create_on_pipe(device_index, 0, 0, out_aligner));
  const a3d_status st = a3d_context_create_with_priority(device_index, -1, out_builder);
  if (st != A3D_OK) {
    a3d_context_destroy(*out_aligner);
    *out_aligner = nullptr;
  }
  return st;
}

a3d_status a3d_context_destroy(a3d_context* ctx) {
  if (!ctx) return A3D_OK;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);  // (a context whose creation failed half-way has gaps)
  if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
  {  // Images (pyramid arenas) created on this context are still alive: their free path needs the context's pool and
     // stream, so the context lives on, unusable for new work, until the last of them is freed (ctx_arena_release).
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    if (ctx->live_arenas > 0) {
      ctx->zombie = true;
      return A3D_OK;
    }
  }
  a3d::ctx_destroy_now(ctx);
  return A3D_OK;
}

a3d_status a3d_context_set_tiling(a3d_context* ctx, uint32_t tiles_per_pair) {
  A3D_REQUIRE(ctx && tiles_per_pair <= 65535, A3D_INVALID_PARAMETER, "bad argument");
  ctx->tiles_per_pair = tiles_per_pair;
  if (ctx->icp_engine && ctx->icp_engine_free) {  // the cached single-pair engine was planned under the other tiling
    hipStreamSynchronize(ctx->stream);
    ctx->icp_engine_free(ctx->icp_engine);
    ctx->icp_engine = nullptr;
  }
  return A3D_OK;
}

}  // extern "C"

namespace a3d {
void ctx_destroy_now(a3d_context* ctx) {
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  for (const auto& b : ctx->spare_blocks) hipFree(b.p);
  ctx->spare_blocks.clear();
  if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
  for (hipEvent_t e : ctx->copy_events) hipEventDestroy(e);
  for (hipEvent_t e : ctx->build_events) hipEventDestroy(e);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  if (ctx->icp_engine && ctx->icp_engine_free) ctx->icp_engine_free(ctx->icp_engine);
  for (hipStream_t st : ctx->side_streams) {
    hipStreamSynchronize(st);
    hipStreamDestroy(st);
  }

  for (void* a : ctx->single_arenas) hipFree(a);  // pooled ones and those of images that are still alive
  for (void* slab : ctx->arena_slabs) hipFree(slab);
  for (auto& t : ctx->tables) hipFree(t.d);
  if (ctx->pinned_words) hipHostFree(ctx->pinned_words);
  hipFree(ctx->scratch[0]);
  hipFree(ctx->scratch[1]);
  hipFree(ctx->scratch[2]);
  if (ctx->ev_start) hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) hipEventDestroy(ctx->ev_stop);
  for (hipEvent_t e : ctx->kd_ev)
    if (e) hipEventDestroy(e);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}
}  // namespace a3d

extern "C" {

a3d_status a3d_context_synchronize(a3d_context* ctx) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return A3D_OK;
}

void* a3d_context_stream(a3d_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int32_t a3d_context_device(a3d_context* ctx) { return ctx ? (int32_t)ctx->device : -1; }

a3d_status a3d_timer_start(a3d_context* ctx) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  A3D_HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
  return A3D_OK;
}
a3d_status a3d_timer_stop(a3d_context* ctx, float* out_ms) {
  A3D_REQUIRE(ctx && out_ms, A3D_INVALID_PARAMETER, "null argument");
  A3D_HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
  A3D_HIP_TRY(hipEventSynchronize(ctx->ev_stop));
  A3D_HIP_TRY(hipEventElapsedTime(out_ms, ctx->ev_start, ctx->ev_stop));
  return A3D_OK;
}

a3d_status a3d_malloc(a3d_context* ctx, size_t bytes, void** out) {
  A3D_REQUIRE(ctx && out, A3D_INVALID_PARAMETER, "null argument");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  A3D_HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
  return A3D_OK;
}
a3d_status a3d_free(a3d_context* ctx, void* p) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  if (p) A3D_HIP_TRY(hipFree(p));
  return A3D_OK;
}
a3d_status a3d_host_alloc(a3d_context* ctx, size_t bytes, void** out) {
  A3D_REQUIRE(ctx && out, A3D_INVALID_PARAMETER, "null argument");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  A3D_HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return A3D_OK;
}
a3d_status a3d_host_free(a3d_context* ctx, void* p) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  if (p) A3D_HIP_TRY(hipHostFree(p));
  return A3D_OK;
}
a3d_status a3d_memcpy_h2d(a3d_context* ctx, void* dst, const void* src, size_t bytes) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  A3D_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return A3D_OK;
}
a3d_status a3d_memcpy_d2h(a3d_context* ctx, void* dst, const void* src, size_t bytes) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  A3D_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return A3D_OK;
}
a3d_status a3d_memcpy_d2d(a3d_context* ctx, void* dst, const void* src, size_t bytes) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  A3D_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return A3D_OK;
}

// IcpParams::default() (src/icp/icp_params.rs:33-43)
void a3d_icp_params_default(a3d_icp_params* out) {
  out->max_iterations = 15;
  out->weight = 1.0f;
  out->color_weight = 1.0e-1f;
  out->max_point_to_plane_distance = 0.1f;
  out->max_distance = 0.5f;
  out->max_normal_angle = 18.0f * (3.14159265358979323846f / 180.0f);  // f32::to_radians
  out->max_color_distance = 0.25f;
}

// MsIcpParams::default() (src/icp/icp_params.rs:112-133)
void a3d_ms_icp_params_default(a3d_icp_params out[3]) {
  const uint64_t iters[3] = {20, 20, 30};
  for (int l = 0; l < 3; ++l) {
    a3d_icp_params_default(&out[l]);
    out[l].weight = 1.0f;
    out[l].color_weight = 1.0f;
    out[l].max_normal_angle = 3.14159265358979323846f / 10.0f;
    out[l].max_color_distance = 2.75f;
    out[l].max_distance = 0.5f;
    out[l].max_iterations = iters[l];
  }
}

// BilateralFilter::default() (src/bilateral/edge_aware_filter.rs:30-36)
void a3d_bilateral_default_sigmas(double* ss, double* sc) {
  if (ss) *ss = 4.50000000225;
  if (sc) *sc = 29.9999880000072;
}

}  // extern "C"
