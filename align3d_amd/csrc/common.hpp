// Internal definitions shared by the translation units of libalign3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/align3d_hip.h"
#include "devmath.hpp"

struct a3d_context;

// Two builds of these sources (csrc/Makefile): the product library libalign3d_hip.so, and the diagnostics build
// libalign3d_hip_diag.so (-DA3D_DIAGNOSTICS) which also holds what only tests and probes use: the environment knobs
// that pick launch geometries and cross-check paths, the kernel variants that were measured slower (last-block
// hand-off, one launch per level, MFMA / merged accumulation, G = 2 / 4 pipelines), the exact-arithmetic accumulate
// kernel, the rocPRIM cross-check sort, the host kd-tree build, the pass-per-launch bilateral grid.  In the product
// build a knob's name does not even reach the object file: A3D_DIAG_ENV(name) is a null constant there.
#ifdef A3D_DIAGNOSTICS
#define A3D_DIAG_ENV(name) getenv(name)
#else
#define A3D_DIAG_ENV(name) ((const char*)nullptr)
#endif

namespace a3d {

void set_error(const char* fmt, ...);

#define A3D_HIP_TRY(expr)                                                                 \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      a3d::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return A3D_HIP_ERROR;                                                               \
    }                                                                                     \
  } while (0)

#define A3D_TRY(expr)                  \
  do {                                 \
    a3d_status _s = (expr);            \
    if (_s != A3D_OK) return _s;       \
  } while (0)

#define A3D_REQUIRE(cond, status, msg) \
  do {                                 \
    if (!(cond)) {                     \
      a3d::set_error("%s", msg);       \
      return status;                   \
    }                                  \
  } while (0)

inline Pose pose_from_c(const a3d_pose* p) {
  return Pose{{p->t[0], p->t[1], p->t[2]}, {p->q[0], p->q[1], p->q[2], p->q[3]}};
}
inline void pose_to_c(const Pose& p, a3d_pose* o) {
  o->t[0] = p.t.x, o->t[1] = p.t.y, o->t[2] = p.t.z;
  o->q[0] = p.q.i, o->q[1] = p.q.j, o->q[2] = p.q.k, o->q[3] = p.q.w;
}

// Largest f32 d in [-1, 1] with  acosf(d) >= thr  (strict = false)  or  acosf(d) > thr  (strict =
// true), found by bisection over the f32 ordering with the host libm (the libm the reference's
// f32::acos resolves to).  Returns -2.0f when no d qualifies.  The kernels then test
// "d >= -1 && d <= d_star" instead of calling a device acosf: the same decisions, one compare.
float acos_gate_threshold(float thr, bool strict);

}  // namespace a3d

namespace a3d {
// BilateralFilter::filter on device-resident u16 images (bilateral.hip).
a3d_status bilateral_filter_device(a3d_context* ctx, const uint16_t* d_img, uint16_t* d_out, uint32_t w, uint32_t h,
                                   double sigma_space, double sigma_color, uint64_t out_grid_dims[3]);
// a3d_range_image_upload_pyramid, optionally overlapped (image.hip): see there.
a3d_status upload_pyramid(a3d_context* ctx, const a3d_range_image_view* views, uint64_t n_levels,
                          a3d_device_image** out_images, hipEvent_t* level_events);
// RangeImage::compute_normals on device-resident arrays (image.hip).
a3d_status compute_normals_device(a3d_context* ctx, const float* d_points, const uint8_t* d_mask, float* d_normals,
                                  uint32_t w, uint32_t h);
}  // namespace a3d

namespace a3d {
struct UseFence;
}
struct a3d_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  hipEvent_t kd_ev[2] = {nullptr, nullptr};  // around a kd-tree build's launches (a3d_kdtree_build_ms): made once
  int num_cus = 0;
  // Grow-only scratch regions for per-call temporaries (all work on a context is ordered on its one stream,
  // so successive calls may reuse them): [0] frame builder temporaries, [1] bilateral grids, [2] kd-tree build.
  void* scratch[3] = {nullptr, nullptr, nullptr};
  size_t scratch_size[3] = {0, 0, 0};
  // Device blocks handed back by freed kd-trees / Icp objects (ctx_block_release), kept for the next one of about the
  // same size (ctx_block_alloc): Icp::new per frame then costs no hipMalloc / hipFree (each a device-wide
  // synchronisation, ~0.1 ms).  Reuse is stream-ordered: everything on a context runs on its one stream.
  struct SpareBlock {
    void* p;
    size_t bytes;
  };
  std::vector<SpareBlock> spare_blocks;
  // Set by a kd-tree build that met an oversized median bucket (a cloud with thousands of equal coordinates): later
  // builds of this context add the chip-wide placement launches for such buckets (kdtree_select.hip).
  // kd-tree selection build: placement launches for oversized median buckets ON for a new context (its first depth-image cloud
  // does not pay a lone block's streaming rounds), OFF after KD_QUIET_BUILDS builds in a row without such a bucket, ON again with
  // the next one (kdtree_select.hip)
  std::atomic<bool> kd_wide_place{true};
  std::atomic<int> kd_quiet_builds{0};
  // Pyramid arenas handed back by a3d_range_image_free, kept for the next frame of the same size: a frame
  // stream then costs no hipMalloc / hipFree (each of which synchronises the whole device) per frame.
  // Guarded by a mutex because an image may be freed from another thread than the one building frames.
  std::mutex pool_mutex;
  std::vector<std::pair<void*, size_t>> arena_pool;
  // every arena that is its own hipMalloc (pooled or held by live images): all released with the context, so that
  // images still alive at a3d_context_destroy do not leak device memory (they must not be used afterwards)
  std::vector<void*> single_arenas;
  // Arenas are carved out of slabs of SLAB_ARENAS at a time (one hipMalloc, i.e. one device-wide synchronisation,
  // per 16 frames instead of per frame); slices return to the pool and the slabs live as long as the context.
  std::vector<void*> arena_slabs;
  std::vector<size_t> slab_sizes;
  size_t slab_bytes_total = 0;
  // The single-pair ICP engine (image_icp.hip) kept between calls: MultiscaleAlign::align and ImageIcp::align
  // reuse its small device state instead of allocating and freeing it per alignment.
  void* icp_engine = nullptr;
  void (*icp_engine_free)(void*) = nullptr;
  uint32_t* pinned_words = nullptr;  // PINNED_WORDS page-locked words: scalar results copied back asynchronously
  static constexpr size_t PINNED_WORDS = 16384;
  static constexpr size_t PINNED_EXTRA = 16;  // behind them: the kd-tree build's flag words (its own slot: nothing else lands there)
  // Second stream of the frame builder: the uploads of chunk k + 1 run under the kernels of chunk k.
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> copy_events;
  // Cells per frame the bilateral grids of the frame builder are given in the grid scratch region (grown on demand).
  unsigned long long grid_capacity = 0;
  // The grid scratch region keeps an invariant between filter enqueues: every packed cell and every tile flag of the
  // layout described here is ZERO (the enqueue's last kernel puts back the zeros its splat replaced), so no enqueue
  // clears whole grids.  region == nullptr: unknown state, the next enqueue clears once.
  struct GridLayoutKey {
    const void* region = nullptr;
    unsigned long long capacity = 0;
    uint32_t cell_bytes = 0, flags_stride = 0, frames = 0, columns = 0;
  } grid_clean;
  uint32_t grid_layout_frames = 0;  // most frames one enqueue has asked for (the layout is sized for it)
  // How often each arena size has been asked for: a slab is only taken for a size that keeps coming back.
  std::vector<std::pair<size_t, uint32_t>> arena_requests;
  // Small read-only tables uploaded once and kept (the blur tap tables of the pyramid builder), keyed by four words.
  struct CachedTable {
    uint32_t key[4];
    void* d;
  };
  std::vector<CachedTable> tables;
  // Side streams (the pair groups of a batch alignment run on them), created on demand with the context's priority
  // and SHARED by every batch of the context: batches of one context are ordered on its main stream anyway (fork /
  // join events), and the runtime maps streams onto a handful of hardware queues — a process that gave every batch
  // its own side streams ended up with two batches' groups serialised on one queue.  Destroyed with the context.
  int stream_priority = 0;
  std::mutex stream_mutex;
  std::vector<hipStream_t> side_streams;
  // What the most recent a3d_range_image_build_pyramids call processed (a3d_context_last_build_stats): frames, cells of
  // their bilateral grids, blur tiles marked by the splat, first-channel tiles written as zeros.
  uint64_t build_stats[4] = {0, 0, 0, 0};
  // a3d_context_set_build_profiling: every chunk of a build is bracketed by a hipEvent pair on the context stream (behind
  // the wait for its upload), and last_build_kernel_ms is the sum of those brackets: the builder's kernels without PCIe.
  bool build_profiling = false;
  std::vector<hipEvent_t> build_events;
  float last_build_kernel_ms = 0.f;
  // a3d_context_set_tiling: 0 = the tiling of the ICP pixel pass follows the batch size (throughput); otherwise every
  // (pair, level) is cut into this many blocks whatever the batch: a pair's pose bits no longer depend on its batch.
  uint32_t tiles_per_pair = 0;
  // Pyramid arenas held by live images.  a3d_context_destroy with arenas outstanding only marks the context (zombie);
  // the release of the last arena destroys it (both under pool_mutex).
  int live_arenas = 0;
  bool zombie = false;
};

namespace a3d {
// Side stream `index` of the context (created on first use; shared, never released).
a3d_status ctx_side_stream(a3d_context* ctx, uint32_t index, hipStream_t* out);
// Returns a scratch region of at least `bytes` (256-byte aligned); growing one synchronises the stream first.
a3d_status ctx_scratch(a3d_context* ctx, int which, size_t bytes, void** out);
// A device block of at least `bytes` that outlives the call (a kd-tree's arrays, an Icp's state): a spare one of
// bytes <= size <= 2 * bytes if the context holds one, else hipMalloc.  *out_bytes = the block's real size (pass it to
// ctx_block_release).  Released blocks are kept (at most 8, 512 MiB) for later calls on the same context / stream.
a3d_status ctx_block_alloc(a3d_context* ctx, size_t bytes, void** out, size_t* out_bytes);
void ctx_block_release(a3d_context* ctx, void* p, size_t bytes);

// One hipMalloc shared by the arrays of several device images (a pyramid); freed with its last user.
struct DeviceArena {
  void* base = nullptr;
  size_t bytes = 0;
  bool slab_slice = false;  // carved out of a context slab: goes back to the pool, never to hipFree
  std::atomic<int> refs{0};
  // Consumers on OTHER streams that may still be reading this arena after their host call returned (a batch
  // alignment enqueued without host outputs): the arena waits for them before it is recycled.
  std::mutex fence_mutex;
  std::vector<std::shared_ptr<struct UseFence>> fences;
  // enqueue-only work of the arena's own context on its images (a3d_range_image_compute_normals): recorded right
  // behind each such launch
  std::shared_ptr<struct UseFence> self_fence;
};

// "Everything enqueued so far by this consumer": an event the consumer re-records after each enqueue.
struct UseFence {
  std::mutex m;
  hipEvent_t ev = nullptr;
  bool recorded = false;
  hipStream_t deferred = nullptr;  // record_later(): the stream the event is recorded on when somebody first waits
  int deferred_device = -1;        // ... and the device that stream lives on
  void record_locked(hipStream_t s) {
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      ev = nullptr;
      return;
    }
    recorded = hipEventRecord(ev, s) == hipSuccess;
  }
  void record(hipStream_t s) {
    std::lock_guard<std::mutex> lock(m);
    record_locked(s);
  }
  // "Whatever is on `s` by the time somebody waits": no event record on the enqueue path (a single-frame
  // compute_normals is a 1.3 us kernel: the record cost as much as the launch); the first waiter records it, which
  // covers this work and whatever the stream was given since — the right trade where the caller synchronises anyway.
  // `device`: the stream's device.  The first waiter may be a thread whose current device is another one (a process that
  // drives several GPUs frees an image while the last context it used is current): an event created there cannot be recorded
  // on this stream (round 5 advisor: hipErrorInvalidHandle, `recorded` stayed false, wait() returned at once and the arena
  // went back to the pool under a running kernel).  settle_locked() switches to the stream's device for the record and back;
  // if the record fails all the same it waits for the stream itself rather than call the fence satisfied.
  void record_later(hipStream_t s, int device) {
    std::lock_guard<std::mutex> lock(m);
    deferred = s, deferred_device = device;
  }
  void settle_locked() {
    if (deferred && !recorded) {
      int current = -1;
      const bool switched = deferred_device >= 0 && hipGetDevice(&current) == hipSuccess && current != deferred_device &&
                            hipSetDevice(deferred_device) == hipSuccess;
      record_locked(deferred);
      if (!recorded) (void)hipStreamSynchronize(deferred);  // (no event: the stream's own completion is the fence)
      if (switched) (void)hipSetDevice(current);
    }
    deferred = nullptr;
  }
  void wait() {  // host wait: the arena is about to be handed to another stream's build
    std::lock_guard<std::mutex> lock(m);
    settle_locked();
    if (ev && recorded) hipEventSynchronize(ev);
  }
  bool wait_on(hipStream_t s) {  // device-side wait: `s` continues after everything recorded so far
    std::lock_guard<std::mutex> lock(m);
    // the waiting stream is the one the work is on: stream order is the fence — for THIS waiter; `deferred` stays set, so a
    // later wait() or wait_on(another stream) still records and waits (round 5 advisor, low)
    if (deferred == s && !recorded) return true;
    settle_locked();
    return !(ev && recorded) || hipStreamWaitEvent(s, ev, 0) == hipSuccess;
  }
  void retire() {  // the consumer is gone (it synchronised its streams first)
    std::lock_guard<std::mutex> lock(m);
    if (ev) hipEventDestroy(ev);
    ev = nullptr, recorded = false;
  }
  ~UseFence() {
    if (ev) hipEventDestroy(ev);
  }
};
#ifdef __HIPCC__
// Workgroups of a 1-D grid go to the eight XCDs round-robin (workgroup b on XCD b % 8, scripts/xcd_probe.hip) and each XCD
// has its own L2.  Virtual index of workgroup `b` of `n` such that every XCD owns one CONTIGUOUS range of virtual indices:
// work items that share cache lines (the tiles of one frame) are given neighbouring virtual indices and meet in one L2.
__device__ __forceinline__ uint32_t xcd_contiguous_index(uint32_t b, uint32_t n) {
  const uint32_t g = b & 7u, k = b >> 3, q = n >> 3, r = n & 7u;
  return (g < r ? g * (q + 1u) : r * (q + 1u) + (g - r) * q) + k;
}
#endif

// Device copy of a small host table, uploaded on first use and kept until the context dies.
a3d_status ctx_cached_table(a3d_context* ctx, const uint32_t key[4], const void* host, size_t bytes, void** out);

// An arena of at least `bytes` from the context's pool, or a fresh hipMalloc; release returns it to the pool
// (up to 128 arenas / 4 GiB per context are kept) or frees it.
a3d_status ctx_arena_acquire(a3d_context* ctx, size_t bytes, DeviceArena* out);
void ctx_arena_release(a3d_context* ctx, DeviceArena* arena);
// Tears the context down (streams, pools, scratch); called by a3d_context_destroy, or by the release of the last arena
// of a context that was destroyed while images were alive.
void ctx_destroy_now(a3d_context* ctx);
}  // namespace a3d

struct a3d_device_image;
namespace a3d {
// Registers a consumer's (or an enqueue-only producer's) fence with the arena an image lives in: the arena is not
// recycled before the fence has passed.  Individually allocated images are released with hipFree, which waits for
// the device by itself.
void attach_fence(const a3d_device_image* im, const std::shared_ptr<UseFence>& fence);
}  // namespace a3d

// One RangeImage in HBM, in the reference's own standard layout (DESIGN.md "Data layout in HBM").
struct a3d_device_image {
  a3d_context* ctx = nullptr;
  uint32_t width = 0, height = 0;
  double fx64 = 0, fy64 = 0, cx64 = 0, cy64 = 0;  // CameraIntrinsics as given (pyramid levels scale these)
  float fx = 0, fy = 0, cx = 0, cy = 0;  // cast f64 -> f32 once, as the reference does at each use
  float* points = nullptr;               // [h][w][3]
  uint8_t* mask = nullptr;               // [h][w]
  float* normals = nullptr;              // [h][w][3] or null
  uint8_t* intensities = nullptr;        // [h*w] or null
  float* imap = nullptr;                 // [(h+2)][(w+2)] or null
  uint8_t* colors = nullptr;             // [h][w][3] or null (kept by the device-side builder for the pyramid)
  bool has_normals = false, has_intensities = false, has_imap = false;
  a3d::DeviceArena* arena = nullptr;     // when set, the arrays above are carved out of it and not freed one by one
  bool built = false;        // made by the device frame builder (its arena layout is fixed when it is planned)
  bool mask_is_z = false;    // built with a depth scale for which mask == (z != 0) on every pixel (frame.hip)
  bool own_normals = false;  // `normals` is its own hipMalloc although the image lives in an arena (uploaded without
                             // normals, a3d_range_image_compute_normals called later)
};
