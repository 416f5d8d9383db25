/* TEST INFRASTRUCTURE — C API of the CPU oracle (a restatement of the reference's algorithm for the
 * ICP hot path; see oracle/README.md).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (align3d_amd/, libalign3d_hip.so) never does.
 * The POD types are the product's (include/align3d_hip.h) so the same ctypes structs serve both. */
#ifndef A3D_ORACLE_H
#define A3D_ORACLE_H
#include "../include/align3d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- math KAT hooks -------------------------------------------------------------------------- */
void orc_pose_eye(a3d_pose* out);
void orc_exp_se3(const float xyz_so3[6], a3d_pose* out);                 /* src/transform.rs:83-108 */
void orc_compose(const a3d_pose* a, const a3d_pose* b, a3d_pose* out);   /* src/transform.rs:205-220 */
void orc_inverse(const a3d_pose* a, a3d_pose* out);
void orc_transform_points(const a3d_pose* p, const float* in, uint64_t n, float* out); /* :138-140 */
void orc_transform_normals(const a3d_pose* p, const float* in, uint64_t n, float* out); /* :151-153 */
void orc_pose_to_matrix(const a3d_pose* p, float out16[16]);             /* Isometry3 -> Matrix4, row-major */
void orc_pose_from_matrix(const float m16[16], a3d_pose* out);           /* Transform::from_matrix4 :112-118 */
/* TransformMetrics::new(lhs, rhs) (src/metrics.rs:23-31): angle and translation norm of lhs^-1 rhs */
void orc_transform_metrics(const a3d_pose* lhs, const a3d_pose* rhs, float* angle, float* translation);
void orc_project(double fx, double fy, double cx, double cy, const float p[3], float uv[2]); /* camera.rs:64-70 */
void orc_project_grad(double fx, double fy, const float p[3], float out4[4]);                /* camera.rs:82-89 */
void orc_backproject(double fx, double fy, double cx, double cy, float x, float y, float z, float out[3]);

/* GaussNewton<6>: n steps with residuals r[n], jacobians J[n][6] (src/optim/gaussnewton.rs:47-77). */
void orc_gn_steps(const float* residuals, const float* jacobians, uint64_t n, a3d_gn_state* out);
void orc_gn_add_weighted(a3d_gn_state* self, const a3d_gn_state* other, float w1, float w2);
void orc_gn_weight(a3d_gn_state* self, float w);
float orc_gn_mean_squared_residual(const a3d_gn_state* s);
int32_t orc_gn_solve(const a3d_gn_state* s, float out[6]); /* 1 = Some, 0 = None */

/* ---- IntensityMap (src/intensity_map.rs) ------------------------------------------------------ */
/* from_luma_image: out is [(h+2)][(w+2)] */
void orc_intensity_map_fill(const uint8_t* luma, uint64_t width, uint64_t height, float* out_map);
void orc_intensity_map_bilinear_grad(const float* map, uint64_t width, uint64_t height, float u,
                                     float v, float out3[3]);

/* ---- ImageIcp / MultiscaleAlign ----------------------------------------------------------------- */
/* One pass of image_icp.rs:76-148 from `pose`.  accum_f64 = 0: f32 accumulators, 4096-pixel chunks
 * merged in chunk order (one of the orders the reference can produce); 1: same samples summed in f64. */
a3d_status orc_image_icp_accumulate(const a3d_icp_params* params, const a3d_range_image_view* target,
                                    const a3d_range_image_view* source, const a3d_pose* pose,
                                    int32_t accum_f64, a3d_gn_state* out_geom, a3d_gn_state* out_color);
/* Order in which the following passes ON THE CALLING THREAD merge their 4096-pixel chunks: 0 (default) = chunk
 * order; any other value seeds a fresh pseudo-random permutation per pass.  The reference merges in whatever order
 * rayon's par_bridge() delivered the chunks (image_icp.rs:96,143-148), so every permutation is a legitimate
 * reference result; tests use this to measure the reference's own run-to-run envelope. */
void orc_set_chunk_merge_order(uint64_t seed);
/* ImageIcp::align.  threads >= 1 spreads the 4096-pixel chunks over threads (merge stays in chunk
 * order).  trace (nullable) receives per iteration [residual, t(3), q(4)] = 8 floats of the
 * transform after that iteration's update. */
a3d_status orc_image_icp_align(const a3d_icp_params* params, const a3d_range_image_view* target,
                               const a3d_range_image_view* source, const a3d_pose* init_pose,
                               int32_t threads, a3d_pose* out_pose, float* trace);
/* MultiscaleAlign::new + align (src/icp/multiscale.rs:26-67). */
a3d_status orc_multiscale_align(const a3d_icp_params* params, uint64_t n_params,
                                const a3d_range_image_view* target_pyramid, uint64_t n_target,
                                const a3d_range_image_view* source_pyramid, uint64_t n_source,
                                int32_t threads, a3d_pose* out_pose);

/* ---- R3dTree / Icp --------------------------------------------------------------------------------- */
typedef struct orc_kdtree orc_kdtree;
a3d_status orc_kdtree_new(const float* points, uint64_t n, orc_kdtree** out);
void orc_kdtree_nearest(const orc_kdtree* t, const float* queries, uint64_t m, uint64_t* out_idx,
                        float* out_sqr);
/* tree shape, for cross-checking the product's implicit layout: number of leaves, internal nodes, max depth */
void orc_kdtree_stats(const orc_kdtree* t, uint64_t out3[3]);
void orc_kdtree_free(orc_kdtree* t);

a3d_status orc_pcl_icp_accumulate(const a3d_icp_params* params, const orc_kdtree* tree,
                                  const a3d_point_cloud_view* target, const a3d_point_cloud_view* source,
                                  const a3d_pose* pose, int32_t accum_f64, a3d_gn_state* out);
a3d_status orc_pcl_icp_align(const a3d_icp_params* params, const orc_kdtree* tree,
                             const a3d_point_cloud_view* target, const a3d_point_cloud_view* source,
                             a3d_pose* out_pose, float* trace);

/* ---- frame preparation ------------------------------------------------------------------------------ */
/* RangeImage::compute_normals (src/range_image/structure.rs:184-262) */
void orc_compute_normals(const float* points, const uint8_t* mask, uint64_t width, uint64_t height,
                         float* out_normals);
/* the same pixel loop spread over `threads` threads in the reference's 1024-pixel chunks (structure.rs:193-201) */
void orc_compute_normals_mt(const float* points, const uint8_t* mask, uint64_t width, uint64_t height,
                            int32_t threads, float* out_normals);
/* BilateralFilter::filter (src/bilateral/edge_aware_filter.rs:126-135) */
a3d_status orc_bilateral_filter_u16(const uint16_t* image, uint64_t width, uint64_t height,
                                    double sigma_space, double sigma_color, uint16_t* out,
                                    uint64_t out_grid_dims[3]);
/* BilateralGrid::from_image + normalize + slice without the blur (src/bilateral/grid.rs:188-194 shape) */
a3d_status orc_bilateral_grid_slice_u16(const uint16_t* image, uint64_t width, uint64_t height,
                                        double sigma_space, double sigma_color, uint16_t* out,
                                        uint64_t out_grid_dims[3]);
/* RangeImage::from_rgbd_image (src/range_image/structure.rs:56-95): points, mask; returns valid count */
uint64_t orc_backproject_depth(const uint16_t* depth, uint64_t width, uint64_t height, double fx,
                               double fy, double cx, double cy, double depth_scale, float* out_points,
                               uint8_t* out_mask);
/* rgb_to_luma_u8 over an [h][w][3] image (src/image/luma.rs:81-83) */
void orc_rgb_to_luma_u8(const uint8_t* rgb, uint64_t n_pixels, uint8_t* out);
/* resize_range_points / resize_range_normals (src/range_image/resize.rs:42-104) */
void orc_resize_range_points(const float* src_points, const uint8_t* src_mask, uint64_t src_w,
                             uint64_t src_h, uint64_t dst_w, uint64_t dst_h, float* dst_points,
                             uint8_t* dst_mask);
void orc_resize_range_normals(const float* src_normals, const uint8_t* src_mask, uint64_t src_w,
                              uint64_t src_h, uint64_t dst_w, uint64_t dst_h, float* dst_normals);
/* py_scale_down2 (src/range_image/structure.rs:38-47): image-0.24.7 gaussian blur restated from its
 * published algorithm — PARITY UNPINNED (no reference test pins its values) — then 2x subsample. */
void orc_rgb_pyr_down(const uint8_t* rgb, uint64_t width, uint64_t height, float sigma, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
