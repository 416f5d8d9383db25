// Exercises include/align3d.hpp (the C++ host-side mirror of the reference API) against libalign3d_hip.so.
//   ./host_mirror_test        CPU-only checks (parameters, error mapping without a GPU)
//   ./host_mirror_test gpu    + the kd-tree KAT of src/kdtree.rs:121-139, MultiscaleAlign::new's length check
//                             (src/icp/multiscale.rs:30-34), one tiny ImageIcp, the device RangeImageBuilder and
//                             a two-pair MultiscaleAlignBatch
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/align3d.hpp"

#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

using namespace align3d;

int main(int argc, char** argv) {
  // IcpParams::default() / MsIcpParams::default() (src/icp/icp_params.rs:33-43, :112-133)
  IcpParams p;
  EXPECT(p.max_iterations == 15 && p.weight == 1.0f && p.color_weight == 0.1f && p.max_distance == 0.5f);
  EXPECT(std::fabs(p.max_normal_angle - 18.0f * 3.14159265f / 180.0f) < 1e-7f && p.max_color_distance == 0.25f);
  MsIcpParams ms = MsIcpParams::default_();
  EXPECT(ms.len() == 3 && ms[0].max_iterations == 20 && ms[1].max_iterations == 20 && ms[2].max_iterations == 30);
  EXPECT(ms[0].color_weight == 1.0f && ms[2].max_color_distance == 2.75f);
  MsIcpParams rep = MsIcpParams::repeat(3, IcpParams()).customize([](size_t i, IcpParams& q) { q.max_iterations = 5 + i; });
  EXPECT(rep[0].max_iterations == 5 && rep[2].max_iterations == 7 && !rep.is_empty());
  BilateralFilter bf;
  EXPECT(bf.sigma_space == 4.50000000225 && bf.sigma_color == 29.9999880000072);
  const bool want_gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
  if (!want_gpu) {
    try {
      Context ctx(0);
      std::printf("a GPU is present; run with `gpu` for the device checks\n");
    } catch (const Panic& e) {
      EXPECT(e.status == A3D_HIP_ERROR);  // no device: a loud error, never a CPU fallback
    }
    std::printf("host mirror CPU checks OK\n");
    return 0;
  }
  Context ctx(0);
  {  // src/kdtree.rs:121-139
    const float pts[] = {1, 2, 3, 2, 3, 4, 5, 6, 7, 8, 9, 1};
    R3dTree tree(ctx, pts, 4);
    EXPECT(tree.nearest({8.f, 9.1f, 1.3f}).first == 3 && tree.nearest({5.1f, 6.4f, 7.f}).first == 2);
    EXPECT(tree.nearest({1.5f, 2.1f, 3.3f}).first == 0 && tree.nearest({2.2f, 3.1f, 4.2f}).first == 1);
  }
  // a 32x24 fronto-parallel plane at z = 2 with a gradient texture, seen twice
  const int W = 32, H = 24;
  std::vector<float> pts(W * H * 3), nrm(W * H * 3), imap((W + 2) * (H + 2), 0.f);
  std::vector<uint8_t> mask(W * H, 1), inten(W * H);
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      int i = r * W + c;
      pts[3 * i] = (c - 16.f) * 2.f / 40.f, pts[3 * i + 1] = (r - 12.f) * 2.f / 40.f, pts[3 * i + 2] = 2.f;
      nrm[3 * i] = 0, nrm[3 * i + 1] = 0, nrm[3 * i + 2] = -1;
      inten[i] = (uint8_t)(4 * c + 3 * r);
      imap[r * (W + 2) + c] = inten[i] / 255.0f;
    }
  a3d_range_image_view v{pts.data(), mask.data(), nrm.data(), inten.data(), imap.data(), 40, 40, 16, 12, (uint64_t)W, (uint64_t)H};
  std::vector<RangeImage> target, source;
  target.emplace_back(ctx, v);
  source.emplace_back(ctx, v);
  try {  // MultiscaleAlign::new: Err(InvalidParameter) on a length mismatch
    MultiscaleAlign bad(ctx, MsIcpParams::repeat(2, IcpParams()), target);
    EXPECT(false);
  } catch (const InvalidParameter& e) {
    EXPECT(std::strstr(e.what(), "must be equal") != nullptr);
  }
  IcpParams one = IcpParams().with_max_iterations(2);
  one.color_weight = 1.0f;
  Transform T = MultiscaleAlign(ctx, MsIcpParams::repeat(1, one), target).align(source);
  // identical frames: the estimate stays at the identity to rounding
  for (int i = 0; i < 3; ++i) EXPECT(std::fabs(T.translation[i]) < 1e-4f && std::fabs(T.rotation_ijkw[i]) < 1e-4f);
  EXPECT(std::fabs(T.rotation_ijkw[3] - 1.0f) < 1e-6f);
  ImageIcp icp(ctx, one, target[0]);
  Transform T2 = icp.align(source[0]);
  EXPECT(std::fabs(T2.translation[0] - T.translation[0]) < 1e-6f);
  {  // RangeImageBuilder on the device + MultiscaleAlignBatch: two identical synthetic frames, two pairs
    const int BW = 64, BH = 48;
    std::vector<uint16_t> depth(BW * BH);
    std::vector<uint8_t> rgb(BW * BH * 3);
    for (int r = 0; r < BH; ++r)
      for (int c = 0; c < BW; ++c) {
        depth[r * BW + c] = (uint16_t)(1500 + 4 * c + 3 * r + ((r * 7 + c * 13) % 5));
        for (int k = 0; k < 3; ++k) rgb[(r * BW + c) * 3 + k] = (uint8_t)((c * 5 + r * 3 + 40 * k) & 255);
      }
    CameraIntrinsics k{60.0, 60.0, 32.0, 24.0, (uint64_t)BW, (uint64_t)BH};
    BilateralFilter bf2;
    RangeImageBuilder builder(ctx);
    builder.with_bilateral_filter(&bf2).pyramid_levels(2);
    std::vector<RangeImage> a = builder.build(k, depth.data(), rgb.data(), 0.001);
    std::vector<RangeImage> b2 = builder.build(k, depth.data(), rgb.data(), 0.001);
    EXPECT(a.size() == 2 && a[0].width() == 64 && a[0].height() == 48 && a[1].width() == 32 && a[1].height() == 24);
    MsIcpParams prm = MsIcpParams::repeat(2, IcpParams().with_max_iterations(3));
    Transform single = MultiscaleAlign(ctx, prm, a).align(b2);
    MultiscaleAlignBatch batch(ctx, prm, {&a, &b2}, {&b2, &a});
    std::vector<Transform> Ts = batch.align();
    EXPECT(Ts.size() == 2 && batch.status()[0] == 0 && batch.status()[1] == 0);
    for (int i = 0; i < 3; ++i) EXPECT(std::fabs(Ts[0].translation[i] - single.translation[i]) < 2e-6f);
    for (int i = 0; i < 4; ++i) EXPECT(std::fabs(Ts[0].rotation_ijkw[i] - single.rotation_ijkw[i]) < 2e-6f);
    batch.rebind({&b2, &a}, {&a, &b2});  // swapped roles: pair 0 now is what pair 1 was
    std::vector<Transform> Tr = batch.align();
    for (int i = 0; i < 3; ++i) EXPECT(Tr[0].translation[i] == Ts[1].translation[i] && Tr[1].translation[i] == Ts[0].translation[i]);
    {  // an aligner + builder context pair, two batches alternating: enqueue the second before the first is read
      auto pair = Context::pair(0);
      Context& aligner = pair.first;
      RangeImageBuilder pb(pair.second);  // frames are built on the builder context, aligned on the aligner
      pb.with_bilateral_filter(&bf2).pyramid_levels(2);
      std::vector<RangeImage> pa = pb.build(k, depth.data(), rgb.data(), 0.001), pc = pb.build(k, depth.data(), rgb.data(), 0.001);
      MultiscaleAlignBatch first(aligner, prm, {&pa, &pc}, {&pc, &pa}), second(aligner, prm, {&pc, &pa}, {&pa, &pc});
      first.enqueue();
      second.enqueue();
      std::vector<Transform> r1 = first.results(), r2 = second.results();
      EXPECT(first.status()[0] == 0 && second.status()[1] == 0);
      for (int i = 0; i < 3; ++i) EXPECT(r1[0].translation[i] == Ts[0].translation[i] && r2[0].translation[i] == Ts[1].translation[i]);
    }
    try {
      RangeImageBuilder(ctx).pyramid_levels(9).build(k, depth.data(), rgb.data(), 0.001);  // 64x48 has no 9 levels
      EXPECT(false);
    } catch (const InvalidParameter&) {
    }
    // the batched builder and the device-list batch: three frames (frame 1 shifted), two pairs over the list {0, 0}
    std::vector<uint16_t> depth1(depth);
    for (auto& d : depth1) d = (uint16_t)(d + 7);
    MultiContext mc({0, 0});
    EXPECT(mc.size() == 2 && mc.shard(2, 0).first == 0 && mc.shard(2, 0).second == 1 && mc.shard(2, 1).first == 1);
    Context c0 = mc.device(0), c1 = mc.device(1);
    RangeImageBuilder b0(c0), b1(c1);
    b0.with_bilateral_filter(&bf2).pyramid_levels(2);
    b1.with_bilateral_filter(&bf2).pyramid_levels(2);
    // pair 0 = frames (0, 1) on entry 0, pair 1 = frames (1, 2) on entry 1
    auto f01 = b0.build_many(k, {depth.data(), depth1.data()}, {rgb.data(), rgb.data()}, 0.001);
    auto f12 = b1.build_many(k, {depth1.data(), depth.data()}, {rgb.data(), rgb.data()}, 0.001);
    EXPECT(f01.size() == 2 && f01[0].size() == 2 && f12[1][1].width() == 32);
    MultiscaleAlignMultiBatch mb(mc, prm, {&f01[0], &f12[0]}, {&f01[1], &f12[1]});
    std::vector<float> mats;
    std::vector<Transform> Tm = mb.align(&mats);
    EXPECT(Tm.size() == 2 && mb.status()[0] == 0 && mb.status()[1] == 0 && mats.size() == 32);
    auto many = builder.build_many(k, {depth.data(), depth1.data()}, {rgb.data(), rgb.data()}, 0.001);
    Transform ref01 = MultiscaleAlign(ctx, prm, many[0]).align(many[1]);
    for (int i = 0; i < 3; ++i) EXPECT(std::fabs(Tm[0].translation[i] - ref01.translation[i]) < 2e-6f);
    EXPECT(std::fabs(mats[3] - Tm[0].translation[0]) < 1e-6f && std::fabs(mats[16 + 7] - Tm[1].translation[1]) < 1e-6f);
  }
  std::printf("host mirror GPU checks OK\n");
  return 0;
}
