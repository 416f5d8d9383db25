/*
 * align3d_hip.h — C ABI of the MI355X (gfx950) implementation of align3d's ICP hot path.
 *
 * The reference (otaviog/align3d, Rust) has no FFI of its own: the hot path sits behind ordinary
 * public Rust API.  Every entry point below names the Rust item it stands in for (file:line under
 * the reference checkout).  A Rust shim that keeps `MultiscaleAlign::new/align` and `MsIcpParams`
 * binds exactly these symbols (INTEGRATION.md shows the `extern "C"` block).
 *
 * Conventions
 *  - plain pointers and sizes only; all structs are POD with fixed-width members;
 *  - every function returns an a3d_status; nothing throws or aborts across the boundary.  Where the
 *    reference panics (`expect`, `unwrap`) the status says why and the shim re-raises;
 *  - "host" pointers are ordinary process memory, "device" pointers are HIP device memory of the
 *    context's GPU;
 *  - a context owns one HIP stream; calls on one context are serialised on that stream and the
 *    functions that return results to host memory synchronise it before returning.
 *  - poses are nalgebra `Isometry3<f32>` storage: translation xyz + unit quaternion (i, j, k, w)
 *    (src/transform.rs:18).
 */
#ifndef ALIGN3D_HIP_H
#define ALIGN3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3D_ABI_VERSION 1

typedef enum a3d_status {
  A3D_OK = 0,
  /* MultiscaleAlign::new length mismatch (src/icp/multiscale.rs:30-34) and any malformed argument. */
  A3D_INVALID_PARAMETER = 1,
  /* target without intensity map / normals, source without intensities (src/icp/image_icp.rs:44-57),
     point cloud without normals (src/icp/pcl_icp.rs:50-58): the reference `expect`s. */
  A3D_MISSING_FIELD = 2,
  /* GaussNewton::solve() == None (count == 0 or Cholesky failed; src/optim/gaussnewton.rs:84-93):
     the reference `unwrap`s (src/icp/image_icp.rs:152, src/icp/pcl_icp.rs:96). */
  A3D_SOLVE_FAILED = 3,
  /* any HIP runtime failure, including "no device"; a3d_last_error() has the text. */
  A3D_HIP_ERROR = 4,
  /* NaN coordinate while building the kd-tree (partial_cmp().unwrap(), src/kdtree.rs:43). */
  A3D_NAN_IN_INPUT = 5,
  /* bilateral slice result not representable as u16 (num::cast().unwrap(), src/bilateral/grid.rs:129). */
  A3D_CAST_OVERFLOW = 6
} a3d_status;

/* IcpParams (src/icp/icp_params.rs:8-23), field for field. */
typedef struct a3d_icp_params {
  uint64_t max_iterations;
  float weight;
  float color_weight;
  float max_point_to_plane_distance; /* never read by the reference either */
  float max_distance;
  float max_normal_angle;
  float max_color_distance;
} a3d_icp_params;

/* Transform = Isometry3<f32> (src/transform.rs:18). */
typedef struct a3d_pose {
  float t[3];
  float q[4]; /* i, j, k, w */
} a3d_pose;

/* Borrowed view of a RangeImage (src/range_image/structure.rs:20-36) exactly as the Rust struct
 * holds it in standard layout, so a shim passes `as_ptr()` with no copies.  Host pointers. */
typedef struct a3d_range_image_view {
  const float* points;        /* [height][width][3], 12-byte stride (Array2<Vector3<f32>>) */
  const uint8_t* mask;        /* [height][width] */
  const float* normals;       /* [height][width][3] or NULL (Option) */
  const uint8_t* intensities; /* [height*width] or NULL (Option) */
  const float* intensity_map; /* [(height+2)][(width+2)] or NULL (IntensityMap.map, src/intensity_map.rs:8) */
  double fx, fy, cx, cy;      /* CameraIntrinsics (src/camera.rs:7-20); cast to f32 at use */
  uint64_t width, height;     /* array dims (the reference never reads intrinsics.width/height on this path) */
} a3d_range_image_view;

/* Borrowed view of a PointCloud (src/pointcloud.rs:8-12). Host pointers. */
typedef struct a3d_point_cloud_view {
  const float* points;  /* [len][3] */
  const float* normals; /* [len][3] or NULL */
  uint64_t len;
} a3d_point_cloud_view;

/* One GaussNewton<6> accumulator as read back for tests (src/optim/gaussnewton.rs:9-14). */
typedef struct a3d_gn_state {
  float hessian[36]; /* row-major 6x6, both triangles filled */
  float gradient[6];
  float squared_residual_sum;
  uint64_t count;
} a3d_gn_state;

/* RangeImageBuilder (src/range_image/builder.rs:7-14). */
typedef struct a3d_builder_params {
  uint32_t with_normals;     /* RangeImage::compute_normals on level 0 */
  uint32_t with_intensity;   /* compute_intensity + compute_intensity_map on every level */
  uint32_t use_bilateral;    /* with_bilateral_filter(Some(BilateralFilter::new(sigma_space, sigma_color))) */
  uint32_t pad;
  double sigma_space, sigma_color;
  uint64_t pyramid_levels;
  float blur_sigma;
  uint32_t pad2;
} a3d_builder_params;

typedef struct a3d_context a3d_context;
typedef struct a3d_device_image a3d_device_image;       /* one RangeImage resident in HBM */
typedef struct a3d_multiscale a3d_multiscale;           /* MultiscaleAlign */
typedef struct a3d_multiscale_batch a3d_multiscale_batch; /* P independent MultiscaleAlign jobs */
typedef struct a3d_multi_context a3d_multi_context;     /* one a3d_context per device of a device list */
typedef struct a3d_multiscale_multi_batch a3d_multiscale_multi_batch; /* P MultiscaleAlign jobs over several GPUs */
typedef struct a3d_kdtree a3d_kdtree;                   /* R3dTree */
typedef struct a3d_pcl_icp a3d_pcl_icp;                 /* Icp */

/* ---- library / context -------------------------------------------------------------------
 * Threading: a context owns one HIP stream, its scratch regions, the pool of pyramid arenas and the cached
 * single-pair ICP engine; calls that take the same context (directly or through a handle created on it) must not
 * run concurrently.  Different contexts, also on the same GPU, may be used from different threads at the same time
 * (this is how frame builds overlap alignments); freeing an image from another thread than the one using its
 * context is allowed.  The reference's objects are re-entrant because they borrow host memory; here a thread that
 * wants its own concurrent `align` creates its own context.  Results are complete when a call returns, except
 * a3d_multiscale_batch_align with no host outputs, which only enqueues (a3d_context_synchronize waits).
 * Lifetime of images: the objects that read images (a3d_multiscale, a3d_multiscale_batch) borrow them like the
 * reference's `&'a Vec<RangeImage>`; an image must not be freed while a host-synchronous call on it is running.
 * After an enqueue-only batch_align the images MAY be freed right away: a3d_range_image_free waits for the batch's
 * launches (also when batch and image live on different contexts / streams) before the memory is recycled.
 * Lifetime of contexts: every handle created on a context (a3d_multiscale, batches, kd-trees, Icp objects) must be
 * freed before a3d_context_destroy.  Images are the exception — they are data and tend to outlive the code that made
 * them: a3d_context_destroy with images of that context still alive waits for the context's work, refuses new work
 * on it, and the context's memory (the pyramid arenas those images live in, its streams) is released when the last
 * such image is freed; a3d_range_image_free on them stays valid. */

uint32_t a3d_abi_version(void);
/* HIP devices visible to the process (0 without a GPU): the device list a3d_multi_context_create is given. */
a3d_status a3d_device_count(int32_t* out_count);
/* Text of the most recent failure on the calling thread ("" if none). */
const char* a3d_last_error(void);
const char* a3d_status_string(a3d_status s);

/* Binds HIP device `device_index`, creates the context's stream. A3D_HIP_ERROR if there is no GPU. */
a3d_status a3d_context_create(int32_t device_index, a3d_context** out_ctx);
/* The same with a stream priority: < 0 the device's highest, 0 the default, > 0 its lowest.  A frame-builder context
 * that shares the GPU with a batch alignment wants the highest: its many short kernels are then dispatched ahead of
 * the alignment's long ones instead of queueing behind them (the build is the dependent chain of the two). */
a3d_status a3d_context_create_with_priority(int32_t device_index, int32_t priority, a3d_context** out_ctx);
/* How the ICP pixel pass of the alignments created on `ctx` is cut into blocks (no reference counterpart: the
 * reference's own sums depend on the order rayon delivers its 75 chunks in, src/icp/image_icp.rs:96,143-148).
 *   tiles_per_pair == 0 (default): throughput tiling — the number of blocks a (pair, level) is cut into follows the
 *     batch size, so that every batch fills the chip; the f32 sums of a pair are then associated differently in a
 *     batch of 64, a batch of 32 and alone, and its pose can differ in the last bits (more where the reference's
 *     parameters are not contractive).
 *   tiles_per_pair  > 0: pinned tiling — every pair's level is cut into that many blocks (fewer when the level is
 *     small), from the pair's OWN size: a pair's pose is bit-identical alone, in any batch, at any position of a
 *     batch, next to images of other sizes, on one stream group or three.  24 is what the throughput tiling gives a
 *     64-pair batch of 640x480 images at level 0, so a 64-pair batch loses nothing at level 0; small batches lose
 *     parallelism (a lone pair runs on 24 blocks instead of ~120).
 * Takes effect for batches created or re-bound and single alignments started after the call. */
a3d_status a3d_context_set_tiling(a3d_context* ctx, uint32_t tiles_per_pair);
/* The same, naming which of the context's four streams (created in order: consecutive hardware queues, i.e. compute pipes
 * 0..3 when the process's streams are created by this library context by context) is its main stream; -1 = the default
 * order of a3d_context_create_with_priority.  For a SECOND aligning context on one GPU (alignments in flight on separate
 * streams, align3d_amd/odometry.py): main_slot 1 or 2 keeps its launch chain off the first context's pipe. */
a3d_status a3d_context_create_on_pipe(int32_t device_index, int32_t priority, int32_t main_slot, a3d_context** out_ctx);
/* An aligning context and the builder context (highest priority) that feeds it, created back to back.  Which compute
 * pipe of the GPU a HIP stream lands on follows the order in which the process creates its streams, and a pipe
 * dispatches one big grid at a time: created as a pair, the builder's kernel stream sits on the pipe of the aligner's
 * idle copy stream, beside (not behind) the three streams a batch alignment's pair groups run on.  Created at
 * unrelated moments, it lands wherever the process's stream count happens to point (measured: 11 k instead of 14 k
 * frame pairs/s in the streaming loop).  Destroy both with a3d_context_destroy. */
a3d_status a3d_context_create_pair(int32_t device_index, a3d_context** out_aligner, a3d_context** out_builder);
/* Waits for the context's work and releases its streams and scratch regions.  Every other handle created on the context
 * must have been freed.  Images may still be alive ("Lifetime of contexts" above): the context then only refuses new
 * work, and its pyramid arenas and streams go when the last of those images is freed. */
a3d_status a3d_context_destroy(a3d_context* ctx);
a3d_status a3d_context_synchronize(a3d_context* ctx);
/* The context's hipStream_t, for callers that want to order their own work after ours. */
void* a3d_context_stream(a3d_context* ctx);
/* The HIP device the context was created on (-1 for a null context): what a rank of a multi-process job checks against
 * its LOCAL_RANK before it does any work (bench.py). */
int32_t a3d_context_device(a3d_context* ctx);

/* hipEvent pair on the context stream: start, ..., stop -> elapsed milliseconds (stop synchronises). */
a3d_status a3d_timer_start(a3d_context* ctx);
a3d_status a3d_timer_stop(a3d_context* ctx, float* out_ms);

/* Raw device memory on the context's GPU (bench / tests keep inputs resident with these). */
a3d_status a3d_malloc(a3d_context* ctx, size_t bytes, void** out_device_ptr);
a3d_status a3d_free(a3d_context* ctx, void* device_ptr);
/* Page-locked host memory (hipHostMalloc): frames handed to a3d_range_image_build_pyramid from such a buffer are
 * copied by DMA at the PCIe rate instead of through the runtime's pageable staging path. */
a3d_status a3d_host_alloc(a3d_context* ctx, size_t bytes, void** out_host_ptr);
a3d_status a3d_host_free(a3d_context* ctx, void* host_ptr);
a3d_status a3d_memcpy_h2d(a3d_context* ctx, void* dst_device, const void* src_host, size_t bytes);
a3d_status a3d_memcpy_d2h(a3d_context* ctx, void* dst_host, const void* src_device, size_t bytes);
a3d_status a3d_memcpy_d2d(a3d_context* ctx, void* dst_device, const void* src_device, size_t bytes);

/* ---- parameters (host only, no GPU needed) ---------------------------------------------- */

/* IcpParams::default() (src/icp/icp_params.rs:33-43). */
void a3d_icp_params_default(a3d_icp_params* out);
/* MsIcpParams::default() (src/icp/icp_params.rs:112-133): writes 3 entries, index 0 = finest level. */
void a3d_ms_icp_params_default(a3d_icp_params out[3]);

/* ---- range images resident on the device ------------------------------------------------- */

/* Copies a RangeImage into HBM in the kernels' layout.  `view->normals`, `intensities`,
 * `intensity_map` may be NULL; what is missing only matters to the call that needs it. */
a3d_status a3d_range_image_upload(a3d_context* ctx, const a3d_range_image_view* view,
                                  a3d_device_image** out_image);
/* The same for a whole pyramid (`&[RangeImage]` as MultiscaleAlign::align receives it, src/icp/multiscale.rs:51):
 * `n_levels` views -> `n_levels` resident images that share ONE arena from the context's pool, one asynchronous copy
 * per array straight from the caller's memory (DMA when it is page-locked, a3d_host_alloc), one synchronisation.
 * A steady stream of upload / align / free calls allocates nothing.  Free each image with a3d_range_image_free. */
a3d_status a3d_range_image_upload_pyramid(a3d_context* ctx, const a3d_range_image_view* views, uint64_t n_levels,
                                          a3d_device_image** out_images);
a3d_status a3d_range_image_free(a3d_device_image* image);

/* RangeImage::compute_normals (src/range_image/structure.rs:184-262) on a resident image;
 * afterwards the image "has normals". */
a3d_status a3d_range_image_compute_normals(a3d_device_image* image);
/* Reads the resident normals back: [height][width][3] f32. */
a3d_status a3d_range_image_download_normals(a3d_device_image* image, float* out_normals);

/* RangeImageBuilder::default() (src/range_image/builder.rs:16-26): normals, intensity, no bilateral
 * filter, 3 levels, blur sigma 1; the sigmas are BilateralFilter::default()'s. */
void a3d_builder_params_default(a3d_builder_params* out);
/* RangeImageBuilder::build(frame) (src/range_image/builder.rs:74-91) on the device: depth u16 [h][w] and
 * rgb u8 [h][w][3] come from host memory, every level of the pyramid stays resident.  out_levels receives
 * params->pyramid_levels handles (index 0 = full resolution); free each with a3d_range_image_free.
 * The pyramid's RGB blur restates image-0.24.7's imageops::blur (parity unpinned). */
a3d_status a3d_range_image_build_pyramid(a3d_context* ctx, const a3d_builder_params* params,
                                         const uint16_t* depth, const uint8_t* rgb, uint64_t width,
                                         uint64_t height, double fx, double fy, double cx, double cy,
                                         double depth_scale, a3d_device_image** out_levels);
/* The same for n_frames frames of one stream (same size, intrinsics and depth scale) in ONE launch sequence: every
 * kernel of the builder has a frame dimension, so up to 48 frames cost the 10 launches one frame costs (a frame stream is
 * launch-bound otherwise).  depth_frames / rgb_frames: n_frames host pointers (page-locked buffers from a3d_host_alloc
 * are copied by DMA).  out_levels: [n_frames][params->pyramid_levels] handles, frame-major.  All or nothing: on
 * failure no handle is returned.  The call returns when every pyramid is complete. */
a3d_status a3d_range_image_build_pyramids(a3d_context* ctx, const a3d_builder_params* params, uint64_t n_frames,
                                          const uint16_t* const* depth_frames, const uint8_t* const* rgb_frames,
                                          uint64_t width, uint64_t height, double fx, double fy, double cx, double cy,
                                          double depth_scale, a3d_device_image** out_levels);
/* Instrumentation for the roofline of the frame builder: what the most recent a3d_range_image_build_pyramids call on
 * this context processed — out_stats = {frames, cells of their bilateral grids (GH x GW x GD, src/bilateral/grid.rs:37-56),
 * 12^3-cell blur tiles the splat marked, first-channel tiles written as zeros}. */
a3d_status a3d_context_last_build_stats(a3d_context* ctx, uint64_t out_stats[4]);
/* Instrumentation: when on, every chunk (up to 48 frames) of a3d_range_image_build_pyramids is bracketed by a hipEvent
 * pair on the context's stream, recorded behind the wait for the chunk's upload: a3d_context_last_build_kernel_ms is the
 * sum of those brackets for the most recent build — the device time of the builder's kernels without the PCIe copies
 * (a live figure for the builder's roofline; rocprofv3 --kernel-trace shows the same kernels one by one). */
a3d_status a3d_context_set_build_profiling(a3d_context* ctx, int32_t on);
a3d_status a3d_context_last_build_kernel_ms(a3d_context* ctx, float* out_ms);
a3d_status a3d_range_image_size(const a3d_device_image* image, uint64_t* out_width, uint64_t* out_height);
/* Reads resident arrays back (each pointer nullable): points [h][w][3], mask [h][w], normals [h][w][3],
 * intensities [h*w], intensity_map [(h+2)][(w+2)], colors [h][w][3] u8, intrinsics fx fy cx cy. */
a3d_status a3d_range_image_download(a3d_device_image* image, float* points, uint8_t* mask, float* normals,
                                    uint8_t* intensities, float* intensity_map, uint8_t* colors,
                                    double out_intrinsics[4]);

/* RangeImage::compute_normals (src/range_image/structure.rs:184-262) on n resident images of one size and one context
 * in ONE launch per 64 images; enqueue-only like a3d_range_image_compute_normals (results are ordered on the context's
 * stream; a3d_range_image_download_normals / a3d_context_synchronize wait).  An odometry or mapping host that keeps its
 * range images resident recomputes the normals of a whole window of frames at the stencil's HBM rate instead of paying
 * one launch per frame. */
a3d_status a3d_range_image_compute_normals_batch(a3d_device_image* const* images, uint64_t n);

/* RangeImage::compute_normals, host in / host out convenience form. */
a3d_status a3d_compute_normals(a3d_context* ctx, const float* points, const uint8_t* mask,
                               uint64_t width, uint64_t height, float* out_normals);

/* ---- ImageIcp (src/icp/image_icp.rs:19-165) ---------------------------------------------- */

/* ImageIcp::new(params, target) + initial_transform + align(source): all iterations run on the
 * device; returns best_transform.  init_pose NULL = Transform::eye(). */
a3d_status a3d_image_icp_align(a3d_context* ctx, const a3d_icp_params* params,
                               const a3d_device_image* target, const a3d_device_image* source,
                               const a3d_pose* init_pose, a3d_pose* out_pose);

/* One pass of the per-pixel body (src/icp/image_icp.rs:101-139) from `pose`, returning the two
 * merged accumulators before add_weighted (the state GaussNewton holds at image_icp.rs:148).  Used for per-iteration
 * parity and by bench.py's live-pixel count. */
a3d_status a3d_image_icp_accumulate(a3d_context* ctx, const a3d_icp_params* params,
                                    const a3d_device_image* target, const a3d_device_image* source,
                                    const a3d_pose* pose, a3d_gn_state* out_geom,
                                    a3d_gn_state* out_color);

#ifdef A3D_DIAGNOSTICS /* exported by the diagnostics build only (csrc/Makefile `diag`: libalign3d_hip_diag.so) */
/* a3d_image_icp_accumulate through a cross-check kernel in which EVERY per-pixel value — the Jacobians
 * (src/icp/cost_function.rs:33-57), CameraIntrinsics::project_grad (src/camera.rs:82-89) and the products of
 * GaussNewton::step (src/optim/gaussnewton.rs:47-77) — is computed with the reference's own unfused operations and
 * IEEE divisions; only the order of the additions differs.  The product kernel fuses those (they only feed the sums). */
a3d_status a3d_image_icp_accumulate_exact(a3d_context* ctx, const a3d_icp_params* params, const a3d_device_image* target,
                                          const a3d_device_image* source, const a3d_pose* pose,
                                          a3d_gn_state* out_geom, a3d_gn_state* out_color);
/* The same pass through the opt-in merged-accumulator kernel (A3D_ICP_ACCUM=merged: a thread sums
 * geom.add_weighted(color, weight, color_weight) (src/optim/gaussnewton.rs:115-121) directly, from the weighted
 * Jacobians), returning that merged accumulator: H, g, the weighted residual sum and the combined count.  Test hook. */
a3d_status a3d_image_icp_accumulate_weighted(a3d_context* ctx, const a3d_icp_params* params,
                                             const a3d_device_image* target, const a3d_device_image* source,
                                             const a3d_pose* pose, a3d_gn_state* out_state);
#endif /* A3D_DIAGNOSTICS */

/* ---- instrumentation that SHIPS in the product library -------------------------------------------------------
 * The four entry points below (a3d_image_icp_align_trace, a3d_selftest_transform, a3d_selftest_division and, further
 * down, a3d_multiscale_batch_persistent_levels) and the timing getters (…_last_kernel_ms, …_last_level_ms, …_last_timing,
 * a3d_context_last_build_*) have no reference counterpart.  They are part of the product library on purpose: they run or
 * time the PRODUCT's own device functions (the iteration tail's pose arithmetic, the shared-reciprocal division, the
 * launches of an alignment), so the parity tests and bench.py check and time the code that ships — no second kernel sits
 * behind any of them.  What only exists to cross-check the product (exact-arithmetic accumulate, merged accumulators,
 * every A3D_* environment knob and the kernel variants behind them) is compiled into the diagnostics build alone
 * (#ifdef A3D_DIAGNOSTICS above). */
/* Instrumentation (no reference counterpart): a3d_image_icp_align that also writes, per iteration,
 * [residual, t(3), q_ijkw(4)] of the transform after that iteration's update; out_trace holds
 * 8 * params->max_iterations floats. */
a3d_status a3d_image_icp_align_trace(a3d_context* ctx, const a3d_icp_params* params,
                                     const a3d_device_image* target, const a3d_device_image* source,
                                     const a3d_pose* init_pose, a3d_pose* out_pose, float* out_trace);

/* Instrumentation: the pose arithmetic of the iteration tail run ON THE DEVICE for n items (host arrays in and out):
 *   out_composed[i] = Transform::exp(&LieGroup::Se3(updates6[i])) * poses[i]   (src/transform.rs:44-108, 205-220;
 *                     poses == NULL: identity), evaluated by the very device functions the ICP kernels' tail calls
 *                     (minimax sin / cos up to theta = pi/4, libm above; the theta^2 < 1e-16 Taylor branch);
 *   out_points3[i]  = Transform::transform_vector(out_composed[i], points3[i])  (src/transform.rs:138-145);
 *   out_normals3[i] = Transform::transform_normal(out_composed[i], points3[i])  (src/transform.rs:147-153).
 * Lets the reference's own known answers (src/transform.rs:321-411) be checked against the device code. */
a3d_status a3d_selftest_transform(a3d_context* ctx, const float* updates6, const a3d_pose* poses, const float* points3,
                                  uint64_t n, a3d_pose* out_composed, float* out_points3, float* out_normals3);
/* Instrumentation: the ICP kernels divide with a reciprocal shared between the quotients of one pixel
 * (same arithmetic as a correctly rounded f32 division); this runs that routine on n caller-drawn
 * (numerator, denominator) pairs on the device and counts results that differ from IEEE `/`. */
a3d_status a3d_selftest_division(a3d_context* ctx, const float* numerators, const float* denominators,
                                 uint64_t n, uint64_t* out_mismatches);

/* ---- MultiscaleAlign (src/icp/multiscale.rs:7-68) ---------------------------------------- */

/* MultiscaleAlign::new(params, &target_pyramid): A3D_INVALID_PARAMETER unless
 * n_params == n_levels (multiscale.rs:30-34).  Index 0 = finest level.  Borrows the images. */
a3d_status a3d_multiscale_new(a3d_context* ctx, const a3d_icp_params* params, uint64_t n_params,
                              const a3d_device_image* const* target_pyramid, uint64_t n_levels,
                              a3d_multiscale** out);
/* MultiscaleAlign::align(&source_pyramid): coarsest level first, each level starts from the
 * previous level's result; a shorter source pyramid truncates like izip! (multiscale.rs:54-64). */
a3d_status a3d_multiscale_align(a3d_multiscale* ms, const a3d_device_image* const* source_pyramid,
                                uint64_t n_source_levels, a3d_pose* out_pose);
/* The same with the source pyramid as the reference holds it: `&[RangeImage]` in HOST memory (src/icp/multiscale.rs:51).
 * One call uploads and aligns: the arrays go up on the context's copy stream, coarsest level first, and the launches of
 * a level wait for that level's arrays only, so the coarse levels iterate under the upload of the fine ones (0.65 against
 * 0.84 ms for upload-then-align at 640x480, 3 levels x 15 iterations, page-locked arrays).  Nothing stays resident.
 * At most 16 levels (A3D_INVALID_PARAMETER beyond — a 17-level pyramid needs an image side of 2^17 pixels; a3d_multiscale_align
 * over resident images has no such limit). */
a3d_status a3d_multiscale_align_host(a3d_multiscale* ms, const a3d_range_image_view* source_pyramid,
                                     uint64_t n_source_levels, a3d_pose* out_pose);
a3d_status a3d_multiscale_free(a3d_multiscale* ms);

/* P independent MultiscaleAlign::new(params, target_p).align(source_p) jobs run as one launch
 * sequence (grid = pairs x tiles).  target/source are [n_pairs][n_levels] row-major handle tables. */
a3d_status a3d_multiscale_batch_new(a3d_context* ctx, const a3d_icp_params* params,
                                    uint64_t n_params, uint64_t n_pairs, uint64_t n_levels,
                                    const a3d_device_image* const* target_pyramids,
                                    const a3d_device_image* const* source_pyramids,
                                    a3d_multiscale_batch** out);
/* Runs every pair.  out_poses_host: [n_pairs] or NULL.  out_matrices_device: [n_pairs][16] f32
 * row-major 4x4 in device memory (the buffer the RCCL gather sends) or NULL.
 * out_status_host: per-pair A3D_OK / A3D_SOLVE_FAILED, [n_pairs] or NULL.  With
 * out_poses_host == NULL and out_status_host == NULL the call only enqueues (no host sync). */
a3d_status a3d_multiscale_batch_align(a3d_multiscale_batch* batch, a3d_pose* out_poses_host,
                                      float* out_matrices_device, int32_t* out_status_host);
/* Results of the most recent pass of `batch` (the way to read an enqueue-only batch_align): host-synchronous, but
 * waits for that pass only, not for work enqueued on the context since (another batch's pass: two batches
 * alternating over a stream of rounds keep the GPU busy while the host reads the previous round). */
a3d_status a3d_multiscale_batch_results(a3d_multiscale_batch* batch, a3d_pose* out_poses_host, int32_t* out_status_host);
/* Points an existing batch at other pyramids (same pair and level counts, same layout as _new): the next
 * batch_align runs on them.  A stream of batches reuses one object and allocates nothing per batch.  Waits for this
 * batch's own earlier passes only. */
a3d_status a3d_multiscale_batch_rebind(a3d_multiscale_batch* batch, const a3d_device_image* const* target_pyramids,
                                       const a3d_device_image* const* source_pyramids);
a3d_status a3d_multiscale_batch_free(a3d_multiscale_batch* batch);
/* Instrumentation: when on, every launch of the per-pixel kernel is bracketed by its own hipEvent pair
 * on the stream it is launched on, and a3d_multiscale_batch_last_kernel_ms returns the sum of those durations
 * for the most recent batch_align (divide by the launch count for the average launch).  A batch splits its
 * pairs into a3d_multiscale_batch_concurrency() groups whose launches run on separate HIP streams at the same
 * time, so that sum can exceed the wall time reported by a3d_multiscale_batch_last_timing. */
a3d_status a3d_multiscale_batch_set_profiling(a3d_multiscale_batch* batch, int32_t on);
a3d_status a3d_multiscale_batch_last_kernel_ms(a3d_multiscale_batch* batch, float* out_kernel_ms);
/* The same sum for one pyramid level.  Levels that ran inside the persistent kernel (one launch per stream group
 * for all their iterations) are timed by that kernel's own clock stamps: per group, from the first pair entering the
 * level to the last pair leaving it; their `launches` are iterations x groups. */
a3d_status a3d_multiscale_batch_last_level_ms(a3d_multiscale_batch* batch, uint32_t level, float* out_ms,
                                              uint32_t* out_launches);
/* Which levels the most recent batch_align ran inside the persistent kernel (bit l = level l): always 0 in the product
 * library (the persistent kernel is a diagnostics-build variant, measured slower); see "instrumentation that ships". */
a3d_status a3d_multiscale_batch_persistent_levels(a3d_multiscale_batch* batch, uint32_t* out_mask);
a3d_status a3d_multiscale_batch_concurrency(a3d_multiscale_batch* batch, uint32_t* out_streams);
/* Time of the most recent batch_align on the device, between hipEvents recorded on the context
 * stream around its launches, and the share of it spent in the per-pixel kernel (sum of that
 * kernel's launches / number of launches). */
a3d_status a3d_multiscale_batch_last_timing(a3d_multiscale_batch* batch, float* out_total_ms,
                                            uint64_t* out_pixel_kernel_launches);

/* ---- P independent MultiscaleAlign jobs over a device list (one host process, several GPUs) ---------------
 * The path shards over independent frame pairs only (no intra-pair sharding): pair j of P goes to device
 * floor(j D / P) — contiguous blocks, 512 pairs over 8 GPUs = 64 each — every device runs its block as one
 * a3d_multiscale_batch on its own context, enqueued by its own host thread, and ONE gather collects the 4x4 poses
 * (16 f32 per pair) into a buffer on the first device (device-to-device copies over xGMI).  The images of a pair
 * must be resident on the device that owns the pair: build or upload them through a3d_multi_context_device(mc, d). */

/* Block [begin, end) of n_items owned by `device` of n_devices (remainders go to the first devices).  Host only. */
a3d_status a3d_multi_shard_range(uint64_t n_items, uint64_t n_devices, uint64_t device, uint64_t* out_begin,
                                 uint64_t* out_end);
/* One context per entry of device_ids (an id may repeat: several contexts = streams on one GPU). */
a3d_status a3d_multi_context_create(const int32_t* device_ids, uint64_t n_devices, a3d_multi_context** out);
a3d_status a3d_multi_context_destroy(a3d_multi_context* mc);
uint64_t a3d_multi_context_size(const a3d_multi_context* mc);
/* The context of entry `index` (borrowed; NULL if out of range): frames of the pairs that entry owns are built on it. */
a3d_context* a3d_multi_context_device(a3d_multi_context* mc, uint64_t index);
/* MultiscaleAlign::new for every pair, as a3d_multiscale_batch_new: [n_pairs][n_levels] handle tables in global pair
 * order.  A3D_INVALID_PARAMETER if an image lives on another device than the owner of its pair. */
a3d_status a3d_multiscale_batch_new_multi(a3d_multi_context* mc, const a3d_icp_params* params, uint64_t n_params,
                                          uint64_t n_pairs, uint64_t n_levels,
                                          const a3d_device_image* const* target_pyramids,
                                          const a3d_device_image* const* source_pyramids,
                                          a3d_multiscale_multi_batch** out);
/* Runs every pair on its device and gathers.  Each output is nullable: out_poses_host [n_pairs], out_matrices_host
 * [n_pairs][16] f32 row-major 4x4, out_status_host [n_pairs], *out_matrices_device0 = the gathered [n_pairs][16]
 * buffer on the first device (owned by the batch, overwritten by the next call).  Returns when all are complete. */
a3d_status a3d_multiscale_multi_batch_align(a3d_multiscale_multi_batch* batch, a3d_pose* out_poses_host,
                                            float* out_matrices_host, int32_t* out_status_host,
                                            const float** out_matrices_device0);
a3d_status a3d_multiscale_multi_batch_free(a3d_multiscale_multi_batch* batch);

/* ---- R3dTree (src/kdtree.rs:19-106) ------------------------------------------------------- */

/* R3dTree::new(&points): `points` [n][3] f32 in host memory are uploaded and the tree is built ON THE DEVICE
 * (kdtree_select.hip: per level the point of rank len / 2 in the reference's own order and a partition around it, the
 * last levels sorted in LDS; leaf <= 16, mid = len / 2 — the same tree, bit for bit, as the reference's recursive build
 * with a stable sort per node). */
a3d_status a3d_kdtree_new(a3d_context* ctx, const float* points, uint64_t n, a3d_kdtree** out);
/* The same over points that are already resident: d_points [n][3] f32 in device memory of the context's GPU (the
 * layout of PointCloud::points, src/pointcloud.rs:8-12).  Read during the call only; the same tree, bit for bit. */
a3d_status a3d_kdtree_new_device(a3d_context* ctx, const void* d_points, uint64_t n, a3d_kdtree** out);
/* Instrumentation: which build made the tree: 1 selection build (kdtree_select.hip: the product's), 3 the same with one
 * more launch per upper level that places oversized median buckets chip-wide (a cloud from a depth image: a wall of
 * tens of thousands of equal coordinates; a new context's builds have it, drop it after four builds in a row without such a
 * bucket and take it up again with the next one — the tree is the same either way); diagnostics build only: 0 host build,
 * 2 sorting build (the cross-checks). */
a3d_status a3d_kdtree_build_path(a3d_kdtree* tree, int32_t* out_path);
/* Instrumentation: device time (ms) of the build's launches (first kernel to last, hipEvents on the context's stream;
 * without the upload of host points and without the allocation of the tree's arrays). */
a3d_status a3d_kdtree_build_ms(a3d_kdtree* tree, float* out_ms);
/* R3dTree::nearest for m queries (leaf-only search, no backtracking).  Host pointers.
 * out_indices are indices into the `points` given to a3d_kdtree_new. */
a3d_status a3d_kdtree_nearest(a3d_kdtree* tree, const float* queries, uint64_t m,
                              uint64_t* out_indices, float* out_sqr_distances);
/* Same with queries and results resident: d_queries [m][3] f32, d_indices [m] u32, d_sqr [m] f32. */
a3d_status a3d_kdtree_nearest_device(a3d_kdtree* tree, const void* d_queries, uint64_t m,
                                     void* d_indices, void* d_sqr_distances);
a3d_status a3d_kdtree_free(a3d_kdtree* tree);
/* Instrumentation: tree shape {leaves, internal nodes, depth of the deepest leaf}. */
a3d_status a3d_kdtree_stats(a3d_kdtree* tree, uint64_t out3[3]);
/* Instrumentation: the tree as laid out in HBM.  out_split: [2^depth - 1] f32 split values in heap order;
 * out_leaves: [2^depth * 16][4] f32 {x, y, z, index bits}, +inf in unused slots.  Either may be null; the
 * element counts are returned in out_counts {splits, leaf slots} (call with null arrays to size the buffers). */
a3d_status a3d_kdtree_download(a3d_kdtree* tree, float* out_split, float* out_leaves, uint64_t out_counts[2]);

/* ---- Icp (src/icp/pcl_icp.rs:15-108) ------------------------------------------------------ */

/* Icp::new(params, &target): builds the kd-tree over target.points; A3D_MISSING_FIELD is
 * deferred to align like the reference. */
a3d_status a3d_pcl_icp_new(a3d_context* ctx, const a3d_icp_params* params,
                           const a3d_point_cloud_view* target, a3d_pcl_icp** out);
/* Icp::align(&source): starts from Transform::eye() (initial_transform is ignored, pcl_icp.rs:59). */
a3d_status a3d_pcl_icp_align(a3d_pcl_icp* icp, const a3d_point_cloud_view* source,
                             a3d_pose* out_pose);
/* Icp::new / Icp::align over clouds that are already resident: the view's `points` / `normals` are DEVICE pointers
 * (same layout: [len][3] f32, src/pointcloud.rs:8-12) on the context's GPU.  The target's arrays are read during
 * a3d_pcl_icp_new_device only (tree and leaf normals are copies); the source's during a3d_pcl_icp_align_device, which
 * is host-synchronous like a3d_pcl_icp_align.  Same tree, same pose bits as the host-pointer forms; no PCIe traffic
 * except the 32-byte result (benches/bench_icp.rs:9-39 without the copies). */
a3d_status a3d_pcl_icp_new_device(a3d_context* ctx, const a3d_icp_params* params,
                                  const a3d_point_cloud_view* d_target, a3d_pcl_icp** out);
a3d_status a3d_pcl_icp_align_device(a3d_pcl_icp* icp, const a3d_point_cloud_view* d_source,
                                    a3d_pose* out_pose);
/* One pass of the per-point body (pcl_icp.rs:68-92) from `pose`: test hook. */
a3d_status a3d_pcl_icp_accumulate(a3d_pcl_icp* icp, const a3d_point_cloud_view* source,
                                  const a3d_pose* pose, a3d_gn_state* out_state);
/* Instrumentation: device time (ms) of the iteration launches of the most recent a3d_pcl_icp_align. */
a3d_status a3d_pcl_icp_last_device_ms(a3d_pcl_icp* icp, float* out_ms);
a3d_status a3d_pcl_icp_free(a3d_pcl_icp* icp);

/* ---- BilateralFilter<u16> (src/bilateral/edge_aware_filter.rs:30-135, grid.rs:32-162) ----- */

/* BilateralFilter::default() sigmas (edge_aware_filter.rs:30-36). */
void a3d_bilateral_default_sigmas(double* out_sigma_space, double* out_sigma_color);
/* BilateralFilter::new(sigma_space, sigma_color).filter(&image): u16 [height][width] in and out,
 * host pointers.  out_grid_dims (nullable) receives GH, GW, GD. */
a3d_status a3d_bilateral_filter_u16(a3d_context* ctx, const uint16_t* image, uint64_t width,
                                    uint64_t height, double sigma_space, double sigma_color,
                                    uint16_t* out_image, uint64_t out_grid_dims[3]);
/* The same filter (edge_aware_filter.rs:126-135) for images that are already in DEVICE memory: n_images images
 * [n][height][width] u16 in, the same layout out (d_out may not alias d_images); the shape of benches/bench_bilateral.rs
 * without PCIe in it.  Images below 2^24 pixels.  With a3d_context_set_build_profiling on, the device time of the call's
 * launch sequences is left in a3d_context_last_build_kernel_ms. */
a3d_status a3d_bilateral_filter_u16_device(a3d_context* ctx, const uint16_t* d_images, uint64_t n_images,
                                           uint64_t width, uint64_t height, double sigma_space, double sigma_color,
                                           uint16_t* d_out);

#ifdef __cplusplus
}
#endif
#endif /* ALIGN3D_HIP_H */
