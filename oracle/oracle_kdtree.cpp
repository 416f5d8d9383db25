// TEST INFRASTRUCTURE — CPU oracle, part 2: R3dTree (src/kdtree.rs) and Icp (src/icp/pcl_icp.rs).
// The tree is kept as an explicit node graph like the reference's Box<Node> (the product uses an
// implicit layout; two different constructions agreeing is part of the check).
// PARITY PINNING: tests/test_oracle_kat.py replays src/kdtree.rs:121-139 (exact) and :142-170
// (property form: every point finds its own index, any permutation).
#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "a3d_oracle.h"
#include "oracle_math.hpp"

using namespace orc;

struct orc_kdtree {
  struct Node {
    bool leaf = false;
    float middle_value = 0.0f;
    int32_t left = -1, right = -1;
    std::vector<V3> points;         // Leaf.points
    std::vector<uint64_t> indices;  // Leaf.indices
  };
  std::vector<Node> nodes;  // nodes[0] is the root
  uint64_t n_leaves = 0, n_internal = 0, max_depth = 0;
};

namespace {

inline V3 load3(const float* base, uint64_t idx) {
  return {base[3 * idx], base[3 * idx + 1], base[3 * idx + 2]};
}
inline float coord(const float* pts, uint64_t idx, int k) { return pts[3 * idx + k]; }

// R3dTree::new::rec (src/kdtree.rs:30-52)
int32_t build_rec(orc_kdtree& t, const float* pts, std::vector<uint64_t> indices, uint64_t depth,
                  bool* nan_seen) {
  int32_t me = (int32_t)t.nodes.size();
  t.nodes.emplace_back();
  t.max_depth = std::max(t.max_depth, depth);
  if (indices.size() <= 16) {
    orc_kdtree::Node& nd = t.nodes[me];
    nd.leaf = true;
    nd.points.reserve(indices.size());
    for (uint64_t i : indices) nd.points.push_back(load3(pts, i));
    nd.indices = std::move(indices);
    t.n_leaves++;
    return me;
  }
  const int k = (int)(depth % 3);
  for (uint64_t i : indices)
    if (std::isnan(coord(pts, i, k))) *nan_seen = true;  // partial_cmp().unwrap() would panic
  // slice::sort_by is a stable sort; partial_cmp orders -0.0 == +0.0
  std::stable_sort(indices.begin(), indices.end(),
                   [&](uint64_t a, uint64_t b) { return coord(pts, a, k) < coord(pts, b, k); });
  const size_t mid = indices.size() / 2;
  const float middle_value = coord(pts, indices[mid], k);
  std::vector<uint64_t> left(indices.begin(), indices.begin() + mid);
  std::vector<uint64_t> right(indices.begin() + mid, indices.end());
  std::vector<uint64_t>().swap(indices);
  int32_t l = build_rec(t, pts, std::move(left), depth + 1, nan_seen);
  int32_t r = build_rec(t, pts, std::move(right), depth + 1, nan_seen);
  orc_kdtree::Node& nd = t.nodes[me];
  nd.middle_value = middle_value;
  nd.left = l;
  nd.right = r;
  t.n_internal++;
  return me;
}

// R3dTree::nearest (src/kdtree.rs:69-105)
inline void nearest(const orc_kdtree& t, V3 point, uint64_t* out_idx, float* out_dist) {
  const orc_kdtree::Node* cur = &t.nodes[0];
  int dim = 0;
  const float pc[3] = {point.x, point.y, point.z};
  while (!cur->leaf) {
    cur = (pc[dim] < cur->middle_value) ? &t.nodes[cur->left] : &t.nodes[cur->right];
    dim = (dim + 1) % 3;
  }
  float min_dist = std::numeric_limits<float>::max();
  size_t min_idx = 0;
  for (size_t i = 0; i < cur->points.size(); ++i) {
    float d = norm_squared(point - cur->points[i]);
    if (d < min_dist) {
      min_dist = d;
      min_idx = i;
    }
  }
  // indices[min_idx] panics on an empty leaf in the reference; report u64::MAX instead.
  *out_idx = cur->indices.empty() ? UINT64_MAX : cur->indices[min_idx];
  *out_dist = min_dist;
}

inline Pose from_c(const a3d_pose* p) {
  return Pose{{p->t[0], p->t[1], p->t[2]}, {p->q[0], p->q[1], p->q[2], p->q[3]}};
}

// The per-point body of Icp::align (src/icp/pcl_icp.rs:68-92), sequential like the reference.
template <typename Acc>
void pcl_pass(const a3d_icp_params& prm, const orc_kdtree& tree, const a3d_point_cloud_view& tgt,
              const a3d_point_cloud_view& src, const Pose& T, GaussNewton6<Acc>& opt) {
  const float max_distance_sqr = prm.max_distance * prm.max_distance;
  for (uint64_t i = 0; i < src.len; ++i) {
    V3 sp = transform_vector(T, load3(src.points, i));
    V3 sn = transform_normal(T, load3(src.normals, i));
    uint64_t found;
    float d2;
    nearest(tree, sp, &found, &d2);
    if (d2 > max_distance_sqr) continue;
    V3 tn = load3(tgt.normals, found);
    float ang = std::fabs(std::acos(dot(sn, tn)));
    if (ang > prm.max_normal_angle) continue;
    V3 tp = load3(tgt.points, found);
    float residual = dot(tp - sp, tn);
    V3 tw = cross(sp, tn);
    float J[6] = {tn.x, tn.y, tn.z, tw.x, tw.y, tw.z};
    opt.step(residual, J);
  }
}

void gn_to_c(const GnF32& s, a3d_gn_state* o) {
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) o->hessian[i * 6 + j] = s.H[i][j];
    o->gradient[i] = s.g[i];
  }
  o->squared_residual_sum = s.ssq;
  o->count = s.count;
}

}  // namespace

extern "C" {

a3d_status orc_kdtree_new(const float* points, uint64_t n, orc_kdtree** out) {
  if (!out || (n && !points)) return A3D_INVALID_PARAMETER;
  auto t = std::make_unique<orc_kdtree>();
  std::vector<uint64_t> indices(n);
  for (uint64_t i = 0; i < n; ++i) indices[i] = i;
  bool nan_seen = false;
  build_rec(*t, points, std::move(indices), 0, &nan_seen);
  if (nan_seen) return A3D_NAN_IN_INPUT;
  *out = t.release();
  return A3D_OK;
}

void orc_kdtree_nearest(const orc_kdtree* t, const float* queries, uint64_t m, uint64_t* out_idx,
                        float* out_sqr) {
  for (uint64_t i = 0; i < m; ++i) nearest(*t, load3(queries, i), &out_idx[i], &out_sqr[i]);
}

void orc_kdtree_stats(const orc_kdtree* t, uint64_t out3[3]) {
  out3[0] = t->n_leaves;
  out3[1] = t->n_internal;
  out3[2] = t->max_depth;
}

void orc_kdtree_free(orc_kdtree* t) { delete t; }

a3d_status orc_pcl_icp_accumulate(const a3d_icp_params* prm, const orc_kdtree* tree,
                                  const a3d_point_cloud_view* target, const a3d_point_cloud_view* source,
                                  const a3d_pose* pose, int32_t accum_f64, a3d_gn_state* out) {
  if (!target->normals || !source->normals) return A3D_MISSING_FIELD;  // pcl_icp.rs:50-58
  Pose T = pose ? from_c(pose) : pose_eye();
  if (accum_f64) {
    GaussNewton6<double> opt;
    pcl_pass<double>(*prm, *tree, *target, *source, T, opt);
    gn_to_c(to_f32(opt), out);
  } else {
    GaussNewton6<float> opt;
    pcl_pass<float>(*prm, *tree, *target, *source, T, opt);
    gn_to_c(to_f32(opt), out);
  }
  return A3D_OK;
}

a3d_status orc_pcl_icp_align(const a3d_icp_params* prm, const orc_kdtree* tree,
                             const a3d_point_cloud_view* target, const a3d_point_cloud_view* source,
                             a3d_pose* out_pose, float* trace) {
  if (!target->normals || !source->normals) return A3D_MISSING_FIELD;
  Pose optim = pose_eye();  // initial_transform is ignored (pcl_icp.rs:59)
  float best_residual = std::numeric_limits<float>::infinity();
  Pose best = optim;
  for (uint64_t it = 0; it < prm->max_iterations; ++it) {
    GaussNewton6<float> opt;
    pcl_pass<float>(*prm, *tree, *target, *source, optim, opt);
    GnF32 g = to_f32(opt);
    float residual = mean_squared_residual(g);  // :94
    weight(g, prm->weight);                     // :95
    float update[6];
    if (!solve(g, update)) return A3D_SOLVE_FAILED;  // :96
    optim = compose(exp_se3(update), optim);         // :97
    if (trace) {
      float* tr = trace + 8 * it;
      tr[0] = residual;
      tr[1] = optim.t.x, tr[2] = optim.t.y, tr[3] = optim.t.z;
      tr[4] = optim.q.i, tr[5] = optim.q.j, tr[6] = optim.q.k, tr[7] = optim.q.w;
    }
    if (residual < best_residual) {  // :100-103
      best_residual = residual;
      best = optim;
    }
  }
  out_pose->t[0] = best.t.x, out_pose->t[1] = best.t.y, out_pose->t[2] = best.t.z;
  out_pose->q[0] = best.q.i, out_pose->q[1] = best.q.j, out_pose->q[2] = best.q.k;
  out_pose->q[3] = best.q.w;
  return A3D_OK;
}

}  // extern "C"
