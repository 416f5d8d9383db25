// P independent MultiscaleAlign jobs over SEVERAL GPUs from one host process (SURVEY §8b "batch variant taking P pairs
// and a device list", §8e): contiguous blocks of pairs per device, one a3d_context + a3d_multiscale_batch per device,
// every device enqueued by its own host thread, and ONE gather of the 4x4 poses into a buffer on the first device
// (hipMemcpyPeerAsync: device-to-device over xGMI) — the single-process counterpart of bench.py's
// one-process-per-GPU + RCCL all-gather layout, for hosts (the Rust crate's callers) that are not launched per GPU.
// Built on the single-device C ABI only.
#include <memory>
#include <thread>

#include "common.hpp"

using namespace a3d;

struct a3d_multi_context {
  std::vector<a3d_context*> ctxs;
  std::vector<int32_t> devices;
};

struct a3d_multiscale_multi_batch {
  a3d_multi_context* mc = nullptr;
  uint64_t n_pairs = 0, n_levels = 0;
  std::vector<a3d_multiscale_batch*> batches;  // one per device that owns at least one pair (else null)
  std::vector<float*> d_mats;                  // per device: [pairs of that device][16]
  float* d_gathered = nullptr;                 // on device 0: [n_pairs][16]
};

extern "C" {

// Pair j of n_items belongs to device floor(j * n_devices / n_items) when n_devices divides n_items (512 pairs over
// 8 GPUs -> 64 each); remainders go to the first devices.  Host only.
a3d_status a3d_multi_shard_range(uint64_t n_items, uint64_t n_devices, uint64_t device, uint64_t* out_begin,
                                 uint64_t* out_end) {
  A3D_REQUIRE(out_begin && out_end && n_devices > 0 && device < n_devices, A3D_INVALID_PARAMETER, "bad shard query");
  const uint64_t base = n_items / n_devices, extra = n_items % n_devices;
  *out_begin = device * base + std::min(device, extra);
  *out_end = *out_begin + base + (device < extra ? 1 : 0);
  return A3D_OK;
}

a3d_status a3d_multi_context_create(const int32_t* device_ids, uint64_t n_devices, a3d_multi_context** out) {
  A3D_REQUIRE(device_ids && out && n_devices >= 1 && n_devices <= 64, A3D_INVALID_PARAMETER,
              "a3d_multi_context_create needs 1..64 device ids");
  auto mc = std::make_unique<a3d_multi_context>();
  for (uint64_t i = 0; i < n_devices; ++i) {
    a3d_context* c = nullptr;
    const a3d_status st = a3d_context_create(device_ids[i], &c);
    if (st != A3D_OK) {
      for (a3d_context* made : mc->ctxs) a3d_context_destroy(made);
      return st;
    }
    mc->ctxs.push_back(c);
    mc->devices.push_back(device_ids[i]);
  }
  // peer access for the pose gather (a no-op between contexts of one device; ignored where the platform refuses it:
  // hipMemcpyPeerAsync then stages through the host)
  for (uint64_t i = 1; i < n_devices; ++i)
    if (device_ids[i] != device_ids[0]) {
      if (hipSetDevice(device_ids[0]) == hipSuccess) (void)hipDeviceEnablePeerAccess(device_ids[i], 0);
      (void)hipGetLastError();
    }
  *out = mc.release();
  return A3D_OK;
}

a3d_status a3d_multi_context_destroy(a3d_multi_context* mc) {
  if (!mc) return A3D_OK;
  for (a3d_context* c : mc->ctxs) a3d_context_destroy(c);
  delete mc;
  return A3D_OK;
}

uint64_t a3d_multi_context_size(const a3d_multi_context* mc) { return mc ? mc->ctxs.size() : 0; }

a3d_context* a3d_multi_context_device(a3d_multi_context* mc, uint64_t index) {
  return (mc && index < mc->ctxs.size()) ? mc->ctxs[index] : nullptr;
}

a3d_status a3d_multiscale_multi_batch_free(a3d_multiscale_multi_batch* mb) {
  if (!mb) return A3D_OK;
  for (size_t d = 0; d < mb->batches.size(); ++d) {
    if (mb->batches[d]) a3d_multiscale_batch_free(mb->batches[d]);
    if (mb->d_mats[d]) a3d_free(mb->mc->ctxs[d], mb->d_mats[d]);
  }
  if (mb->d_gathered) a3d_free(mb->mc->ctxs[0], mb->d_gathered);
  delete mb;
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_new_multi(a3d_multi_context* mc, const a3d_icp_params* params, uint64_t n_params,
                                          uint64_t n_pairs, uint64_t n_levels,
                                          const a3d_device_image* const* target_pyramids,
                                          const a3d_device_image* const* source_pyramids,
                                          a3d_multiscale_multi_batch** out) {
  A3D_REQUIRE(mc && params && target_pyramids && source_pyramids && out, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(n_params == n_levels, A3D_INVALID_PARAMETER,
              "The number of range images pyramid levels and ICP parameters must be equal.");
  A3D_REQUIRE(n_pairs > 0 && n_levels > 0 && n_levels <= 16, A3D_INVALID_PARAMETER, "bad batch shape");
  const uint64_t D = mc->ctxs.size();
  // every pair's images must already live on the device that owns the pair (they were built / uploaded through
  // a3d_multi_context_device(mc, owner)): nothing is copied between devices here
  for (uint64_t d = 0; d < D; ++d) {
    uint64_t lo, hi;
    a3d_multi_shard_range(n_pairs, D, d, &lo, &hi);
    for (uint64_t k = lo * n_levels; k < hi * n_levels; ++k) {
      A3D_REQUIRE(target_pyramids[k] && source_pyramids[k], A3D_INVALID_PARAMETER, "null image handle");
      A3D_REQUIRE(target_pyramids[k]->ctx->device == mc->devices[d] && source_pyramids[k]->ctx->device == mc->devices[d],
                  A3D_INVALID_PARAMETER,
                  "an image is resident on another device than the one that owns its pair (a3d_multi_shard_range)");
    }
  }
  auto mb = std::unique_ptr<a3d_multiscale_multi_batch>(new a3d_multiscale_multi_batch());
  mb->mc = mc, mb->n_pairs = n_pairs, mb->n_levels = n_levels;
  mb->batches.assign(D, nullptr);
  mb->d_mats.assign(D, nullptr);
  a3d_status st = a3d_malloc(mc->ctxs[0], n_pairs * 64, (void**)&mb->d_gathered);
  for (uint64_t d = 0; d < D && st == A3D_OK; ++d) {
    uint64_t lo, hi;
    a3d_multi_shard_range(n_pairs, D, d, &lo, &hi);
    if (lo == hi) continue;
    st = a3d_multiscale_batch_new(mc->ctxs[d], params, n_params, hi - lo, n_levels, target_pyramids + lo * n_levels,
                                  source_pyramids + lo * n_levels, &mb->batches[d]);
    if (st == A3D_OK) st = a3d_malloc(mc->ctxs[d], (hi - lo) * 64, (void**)&mb->d_mats[d]);
  }
  if (st != A3D_OK) {
    a3d_multiscale_multi_batch_free(mb.release());
    return st;
  }
  *out = mb.release();
  return A3D_OK;
}

a3d_status a3d_multiscale_multi_batch_align(a3d_multiscale_multi_batch* mb, a3d_pose* out_poses_host,
                                            float* out_matrices_host, int32_t* out_status_host,
                                            const float** out_matrices_device0) {
  A3D_REQUIRE(mb, A3D_INVALID_PARAMETER, "batch is null");
  const uint64_t D = mb->batches.size();
  std::vector<a3d_status> st(D, A3D_OK);
  std::vector<std::string> err(D);
  // one host thread per device: a device's launch sequence costs ~0.7 ms of host time, so eight devices enqueued from
  // one thread would be bound by the host, not by the GPUs
  auto work = [&](uint64_t d) {
    uint64_t lo, hi;
    a3d_multi_shard_range(mb->n_pairs, D, d, &lo, &hi);
    st[d] = a3d_multiscale_batch_align(mb->batches[d], out_poses_host ? out_poses_host + lo : nullptr, mb->d_mats[d],
                                       out_status_host ? out_status_host + lo : nullptr);
    if (st[d] == A3D_OK && !(out_poses_host || out_status_host)) st[d] = a3d_context_synchronize(mb->mc->ctxs[d]);
    if (st[d] != A3D_OK) err[d] = a3d_last_error();
  };
  std::vector<std::thread> threads;
  for (uint64_t d = 1; d < D; ++d)
    if (mb->batches[d]) threads.emplace_back(work, d);
  if (mb->batches[0]) work(0);
  for (auto& t : threads) t.join();
  for (uint64_t d = 0; d < D; ++d)
    if (st[d] != A3D_OK) {
      set_error("device %d: %s", (int)mb->mc->devices[d], err[d].c_str());
      return st[d];
    }
  // the gather: every device's block of 4x4 poses into the global-order buffer on the first device
  a3d_context* c0 = mb->mc->ctxs[0];
  A3D_HIP_TRY(hipSetDevice(c0->device));
  for (uint64_t d = 0; d < D; ++d) {
    uint64_t lo, hi;
    a3d_multi_shard_range(mb->n_pairs, D, d, &lo, &hi);
    if (lo == hi) continue;
    A3D_HIP_TRY(hipMemcpyPeerAsync(mb->d_gathered + lo * 16, c0->device, mb->d_mats[d], mb->mc->devices[d], (hi - lo) * 64,
                                   c0->stream));
  }
  if (out_matrices_host)
    A3D_HIP_TRY(hipMemcpyAsync(out_matrices_host, mb->d_gathered, mb->n_pairs * 64, hipMemcpyDeviceToHost, c0->stream));
  A3D_HIP_TRY(hipStreamSynchronize(c0->stream));
  if (out_matrices_device0) *out_matrices_device0 = mb->d_gathered;
  return A3D_OK;
}

}  // extern "C"
