// A C host below Python: P frame pairs per rank aligned through the C ABI (include/align3d_hip.h), then ONE RCCL
// all-gather of the 4x4 poses (16 f32 per pair) on the context's own stream — the multi-GPU layout of SURVEY §8e /
// BASELINE configs[4] as a Rust or C program would write it (bench.py does the same through torch.distributed).
// The library itself does not link librccl: it hands out the device buffer (out_matrices_device) and the stream
// (a3d_context_stream), which is all a collective needs.
//
//   hipcc -O2 examples/rccl_gather.cpp -Iinclude -Lalign3d_amd/csrc -lalign3d_hip -lrccl -o examples/rccl_gather
//   ./examples/rccl_gather                       one rank (RANK / WORLD_SIZE / LOCAL_RANK unset)
//   RANK=r WORLD_SIZE=n LOCAL_RANK=r A3D_NCCL_ID_FILE=/tmp/id ./examples/rccl_gather     n ranks, one per GPU
// Plain C calls only (compiled as C++ because the HIP / RCCL headers are); error handling by exit code.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/align3d_hip.h"

#define A3D(x) do { a3d_status s_ = (x); if (s_ != A3D_OK) { std::printf("%s -> %s: %s\n", #x, a3d_status_string(s_), a3d_last_error()); return 2; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::printf("%s -> %s\n", #x, ncclGetErrorString(r_)); return 4; } } while (0)

int main() {
  const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
  const int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
  const int device = std::getenv("LOCAL_RANK") ? std::atoi(std::getenv("LOCAL_RANK")) : 0;
  const int P = 4, W = 160, H = 120, L = 2;  // pairs per rank, frame size, pyramid levels

  a3d_context* ctx = nullptr;
  A3D(a3d_context_create(device, &ctx));
  hipStream_t stream = (hipStream_t)a3d_context_stream(ctx);

  // RCCL communicator: rank 0 makes the id and writes it to a file every rank can see; EVERY rank (rank 0 too) then reads
  // the id from that file — so a one-rank run with A3D_NCCL_ID_FILE set goes through the whole N-rank rendezvous and only
  // the communicator's size differs from the 8-GPU launch.  Without the variable (one rank only): no file.
  ncclUniqueId id;
  const char* id_file = std::getenv("A3D_NCCL_ID_FILE");
  if (world > 1 && !id_file) {
    std::printf("WORLD_SIZE > 1 needs A3D_NCCL_ID_FILE\n");
    return 5;
  }
  if (rank == 0) {
    NCCL(ncclGetUniqueId(&id));
    if (id_file) {  // written under a temporary name and renamed: a reader never sees half an id
      const std::string tmp = std::string(id_file) + ".tmp";
      FILE* f = std::fopen(tmp.c_str(), "wb");
      if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return 5;
      std::fclose(f);
      if (std::rename(tmp.c_str(), id_file) != 0) return 5;
    }
  }
  if (id_file) {
    std::memset(&id, 0, sizeof(id));
    FILE* f = nullptr;
    for (int tries = 0; tries < 600 && !f; ++tries) {
      f = std::fopen(id_file, "rb");
      if (!f) usleep(100000);
    }
    if (!f || std::fread(&id, sizeof(id), 1, f) != 1) return 5;
    std::fclose(f);
  }
  ncclComm_t comm;
  NCCL(ncclCommInitRank(&comm, world, id, rank));

  // P + 1 synthetic frames of this rank's stream (a wavy surface with a texture, drifting sideways), built on the device
  a3d_builder_params bp;
  a3d_builder_params_default(&bp);
  bp.pyramid_levels = L;
  std::vector<std::vector<uint16_t>> depth(P + 1, std::vector<uint16_t>(W * H));
  std::vector<std::vector<uint8_t>> rgb(P + 1, std::vector<uint8_t>(W * H * 3));
  std::vector<const uint16_t*> dptr;
  std::vector<const uint8_t*> cptr;
  for (int f = 0; f <= P; ++f) {
    const float shift = 0.35f * (float)f + 3.0f * (float)rank;
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < W; ++c) {
        const float x = (float)c + shift, y = (float)r;
        depth[f][r * W + c] = (uint16_t)(2000.0f + 250.0f * std::sin(x * 0.045f) * std::cos(y * 0.06f) + 1.5f * y);
        for (int k = 0; k < 3; ++k)
          rgb[f][(r * W + c) * 3 + k] = (uint8_t)(128.0f + 60.0f * std::sin(x * 0.21f + (float)k) * std::cos(y * 0.17f));
      }
    dptr.push_back(depth[f].data()), cptr.push_back(rgb[f].data());
  }
  std::vector<a3d_device_image*> levels((P + 1) * L);
  A3D(a3d_range_image_build_pyramids(ctx, &bp, P + 1, dptr.data(), cptr.data(), W, H, 140.0, 140.0, W / 2.0, H / 2.0, 0.001,
                                     levels.data()));
  std::vector<const a3d_device_image*> targets, sources;  // pair p: frame p (target) <- frame p + 1 (source)
  for (int p = 0; p < P; ++p)
    for (int l = 0; l < L; ++l) targets.push_back(levels[p * L + l]), sources.push_back(levels[(p + 1) * L + l]);
  a3d_icp_params prm[2];
  a3d_icp_params_default(&prm[0]);
  prm[0].max_iterations = 6;
  prm[1] = prm[0];
  a3d_multiscale_batch* batch = nullptr;
  A3D(a3d_multiscale_batch_new(ctx, prm, L, P, L, targets.data(), sources.data(), &batch));

  // this rank's poses land in d_local (device), the gather collects every rank's block in global pair order
  float *d_local = nullptr, *d_all = nullptr;
  HIP(hipMalloc((void**)&d_local, (size_t)P * 16 * sizeof(float)));
  HIP(hipMalloc((void**)&d_all, (size_t)world * P * 16 * sizeof(float)));
  std::vector<a3d_pose> poses(P);
  std::vector<int32_t> status(P);
  A3D(a3d_multiscale_batch_align(batch, nullptr, d_local, nullptr));  // enqueue only: no host synchronisation
  NCCL(ncclAllGather(d_local, d_all, (size_t)P * 16, ncclFloat, comm, stream));  // ordered behind the kernels
  A3D(a3d_multiscale_batch_results(batch, poses.data(), status.data()));
  std::vector<float> all((size_t)world * P * 16), local((size_t)P * 16);
  HIP(hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  HIP(hipMemcpyAsync(local.data(), d_local, local.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));

  int bad = 0;
  for (int p = 0; p < P; ++p) bad += status[p] != A3D_OK;
  bad += std::memcmp(all.data() + (size_t)rank * P * 16, local.data(), local.size() * sizeof(float)) != 0;  // own block
  for (int p = 0; p < P; ++p) {  // a rigid transform with the pose's translation in the last column
    const float* m = all.data() + ((size_t)rank * P + p) * 16;
    bad += !(m[12] == 0.f && m[13] == 0.f && m[14] == 0.f && m[15] == 1.f && m[3] == poses[p].t[0] && m[7] == poses[p].t[1]);
  }
  if (rank == 0)
    std::printf("%s: %d ranks x %d pairs gathered over RCCL; pair 0 of rank 0: t = (%.5f, %.5f, %.5f)\n",
                bad ? "FAILED" : "rccl gather OK", world, P, poses[0].t[0], poses[0].t[1], poses[0].t[2]);
  a3d_multiscale_batch_free(batch);
  for (a3d_device_image* im : levels) a3d_range_image_free(im);
  (void)hipFree(d_local), (void)hipFree(d_all);
  ncclCommDestroy(comm);
  a3d_context_destroy(ctx);
  return bad ? 1 : 0;
}
