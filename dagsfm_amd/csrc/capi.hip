// capi.hip -- the C-ABI (include/dagsfm_mi355x.h) over the HIP kernels.
//
// Host-side orchestration only: HBM layout, pair scheduling, kernel launches, result
// fetches.  No arithmetic of the hot path is done on the host; if no HIP device is
// present dsm_ctx_create fails with DSM_ERR_NO_DEVICE (there is no CPU fallback).
//
// HBM layout (one context = one GPU):
//   desc_s8   [total_rows][128] int8   all images back to back, each padded with zero rows
//                                      to a multiple of 256 rows (zero rows never match)
//   rterm     [total_rows]      int32  128 * sum(row as s8)
//   kp        [total_rows][2]   double keypoint (x,y) as FeatureKeypointsToPointsVector
//                                      (/root/reference/src/feature/utils.cc:38-46) makes them
//   m         [sum rows(a)]     int32  K1 output per directed pair (scratch, per chunk)
//   matches   [total][2]        uint32 compact FeatureMatches of all pairs, list order
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/dagsfm_mi355x.h"
#include "ctx.h"
#include "kernels.h"
#include "verify_camera.h"

namespace {
thread_local std::string g_create_error;
}  // namespace

static int fail(dsm_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

// ---------------------------------------------------------------------------------- hand-over of host buffers
// The reference's FeatureDescriptors / FeatureKeypoints are Eigen matrices and std::vectors in PAGEABLE memory
// (/root/reference/src/feature/types.h:102-104), one allocation per image.  hipMemcpy from pageable memory goes through the
// runtime's own staging at 4 - 5 GB/s (measured: 278 MB of config 2 in 57 ms).  staged_upload moves `total` bytes that a
// callback can produce for any byte range: host threads fill one pinned slot while the DMA engine drains the other (and,
// for descriptors, k0_prepare converts the slot that just arrived), all device work on ctx->stream.  Buffers that are pinned
// already (the shim's FeatureMatcherCache slabs) skip this and are copied where they lie.
static const uint64_t kStageSlot = 16ull << 20;
static const uint64_t kStageSmall = 1ull << 20;  // at and below: one plain copy (the staging machinery costs a leaf call more than it saves)
static const unsigned kStageThreads = 8;  // (images that are cold in the host's caches: 4 threads read them at 8 GB/s, hot ones at 32 GB/s)

static bool host_pointer_is_pinned(const void* ptr) {
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory is "invalid value" to the runtime: not an error of ours
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// fill(dst, begin, end) writes bytes [begin, end) of the stream to dst; arrived(slot_dev, begin, end) enqueues what follows a
// slot's copy on the stream (nullptr: the copy went straight to dev_dst + begin).
template <typename Fill, typename Arrived>
static int staged_upload(dsm_ctx* ctx, uint64_t total, uint8_t* dev_dst, bool through_device_slot, Fill fill, Arrived arrived) {
  if (!total) return DSM_OK;
  if (total <= kStageSmall) {
    // a leaf call's two images (dsm_match_sift_features, dsm_estimate_two_view_geometry) or a handful of keypoints: no pinned slots,
    // no device mirror of 32 MB each, no helper threads -- gather into one host buffer, one copy
    std::vector<uint8_t> tmp;
    try {
      tmp.resize(total);
    } catch (const std::exception&) {
      return dsm_fail(ctx, DSM_ERR_HIP, "staged_upload: out of host memory");
    }
    fill(tmp.data(), 0, total);
    uint8_t* dev = dev_dst;
    if (through_device_slot) {
      HIPCHK(ctx, ctx->d_stage.reserve(total));
      dev = ctx->d_stage.as<uint8_t>();
    }
    HIPCHK(ctx, hipMemcpyAsync(dev, tmp.data(), total, hipMemcpyHostToDevice, ctx->stream));
    if (through_device_slot) {
      const int rc = arrived(dev, 0, total);
      if (rc != DSM_OK) return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (tmp is read until here)
    return DSM_OK;
  }
  if (!ctx->h_stage) HIPCHK(ctx, hipHostMalloc(&ctx->h_stage, 2 * kStageSlot, hipHostMallocDefault));
  for (hipEvent_t& e : ctx->stage_ev)
    if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (through_device_slot) HIPCHK(ctx, ctx->d_stage.reserve(2 * kStageSlot));
  const unsigned hw = std::thread::hardware_concurrency();
  const unsigned n_thr = std::max(1u, std::min(kStageThreads, hw / 4));  // (the shim hands over to every device of gpu_index at once)
  uint64_t begin = 0;
  for (uint32_t piece = 0; begin < total; ++piece) {
    const uint32_t slot = piece & 1u;
    const uint64_t end = std::min(total, begin + kStageSlot);
    uint8_t* host = static_cast<uint8_t*>(ctx->h_stage) + slot * kStageSlot;
    if (piece >= 2) HIPCHK(ctx, hipEventSynchronize(ctx->stage_ev[slot]));  // the copy out of this slot two pieces ago is done
    {
      // equal shares of the piece, cut at 4 KB; the caller's thread takes the first
      const uint64_t share = ((end - begin + n_thr - 1) / n_thr + 4095) / 4096 * 4096;
      std::vector<std::thread> helpers;
      struct JoinAll {  // whatever leaves this scope -- an exception of fill or of the vector included -- finds no joinable thread behind
        std::vector<std::thread>& v;
        ~JoinAll() {
          for (std::thread& th : v)
            if (th.joinable()) th.join();
        }
      } join_all{helpers};
      helpers.reserve(n_thr);
      for (unsigned t = 1; t < n_thr; ++t) {
        const uint64_t b = begin + t * share, e = std::min(end, b + share);
        if (b >= end) break;
        try {
          helpers.emplace_back([&fill, host, begin, b, e]() { fill(host + (b - begin), b, e); });
        } catch (const std::exception&) {  // no thread (or no memory for one) to be had: this one does the share itself
          fill(host + (b - begin), b, e);
        }
      }
      fill(host, begin, std::min(end, begin + share));
    }
    uint8_t* dev = through_device_slot ? ctx->d_stage.as<uint8_t>() + slot * kStageSlot : dev_dst + begin;
    HIPCHK(ctx, hipMemcpyAsync(dev, host, end - begin, hipMemcpyHostToDevice, ctx->stream));
    if (through_device_slot) {
      const int rc = arrived(dev, begin, end);
      if (rc != DSM_OK) return rc;
    }
    HIPCHK(ctx, hipEventRecord(ctx->stage_ev[slot], ctx->stream));
    begin = end;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return DSM_OK;
}

extern "C" {

void dsm_default_match_options(dsm_match_options* o) {
  if (!o) return;
  o->max_ratio = 0.8;       // sift.h:129
  o->max_distance = 0.7;    // sift.h:132
  o->cross_check = 1;       // sift.h:135
  o->max_num_matches = 32768;  // sift.h:138
}

void dsm_default_two_view_options(dsm_two_view_options* o) {
  if (!o) return;
  o->min_num_inliers = 15;              // two_view_geometry.h:107
  o->min_E_F_inlier_ratio = 0.95;       // :112
  o->max_H_inlier_ratio = 0.8;          // :117
  o->watermark_min_inlier_ratio = 0.7;  // :122
  o->watermark_border_size = 0.1;       // :127
  o->detect_watermark = 1;              // :130
  o->multiple_models = 0;               // sift.h:159
  o->multiple_ignore_watermark = 1;     // two_view_geometry.h:140
  o->reserved = 0;
  o->max_error = 4.0;                   // sift.h:141
  o->min_inlier_ratio = 0.25;           // sift.h:152
  o->confidence = 0.999;                // sift.h:144
  o->min_num_trials = 30;               // sift.h:148
  o->max_num_trials = 10000;            // sift.h:149
}

int dsm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n < 0 ? 0 : n;
}

int dsm_ctx_create(int device, dsm_ctx** out_ctx) {
  if (!out_ctx) return DSM_ERR_INVALID_ARGUMENT;
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = "no HIP device visible (hipGetDeviceCount)";
    return DSM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    g_create_error = "device index out of range";
    return DSM_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_create_error = "hipSetDevice failed";
    return DSM_ERR_NO_DEVICE;
  }
  dsm_ctx* c = new dsm_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    g_create_error = "hipStreamCreate failed";
    return DSM_ERR_HIP;
  }
  // The second verification lane's stream, created NOW (lane 0 runs on c->stream): the runtime hands a stream its hardware queue
  // when the stream is created, from a small pool (ROCclr: 4 by default), and two busy streams on one queue serialise.  A host
  // that creates its contexts first -- before RCCL, before worker streams of its own -- gives the two lanes queues of their
  // own; created lazily at the first dsm_verify_pairs they got whatever was left (bench.py --force-collectives: 322 ms of
  // verification against 288, profiles/r05_lanes_hw_queues.txt).  Not fatal if it fails here: the lane creates it when it runs.
  c->lanes[0].stream = c->stream;
  if (hipStreamCreateWithFlags(&c->lanes[1].stream, hipStreamNonBlocking) != hipSuccess) c->lanes[1].stream = nullptr;
  // acos LUT over the integer dot product, built with the host libm so that the float
  // compares of sift.cc:140-155 are reproduced bit for bit.
  std::vector<float> lut(262145);
  const float kDistNorm = 1.0f / (512.0f * 512.0f);  // sift.cc:115
  for (int d = 0; d <= 262144; ++d) lut[d] = acosf(fminf(kDistNorm * (float)d, 1.0f));
  if (c->d_lut.reserve(lut.size() * sizeof(float)) != hipSuccess ||
      hipMemcpy(c->d_lut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      c->d_total.reserve(sizeof(uint64_t)) != hipSuccess) {
    g_create_error = "device allocation failed";
    (void)hipStreamDestroy(c->stream);
    delete c;
    return DSM_ERR_HIP;
  }
  *out_ctx = c;
  return DSM_OK;
}

void dsm_ctx_destroy(dsm_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->leaf) dsm_ctx_destroy(ctx->leaf);
  dsm_retrieval_destroy(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
  DevBuf* bufs[] = {&ctx->d_desc, &ctx->d_rterm, &ctx->d_kp, &ctx->d_img_row0, &ctx->d_img_rows, &ctx->d_lut,
                    &ctx->d_dpairs, &ctx->d_doutoff, &ctx->d_pair_dir, &ctx->d_m, &ctx->d_counts,
                    &ctx->d_offsets, &ctx->d_matches, &ctx->d_total, &ctx->d_cams, &ctx->d_pairs_dev, &ctx->d_seeds,
                    &ctx->d_tvg, &ctx->d_inl, &ctx->d_inl_counts, &ctx->d_inl_off, &ctx->d_inl_compact,
                    &ctx->d_vscratch, &ctx->d_inl_total, &ctx->d_nt_table, &ctx->d_nt_off, &ctx->d_nt_off_t,
                    &ctx->d_pair_state, &ctx->d_pts_px, &ctx->d_pts_norm, &ctx->d_reports, &ctx->d_masks,
                    &ctx->d_fam_state,
                    &ctx->d_sidx, &ctx->d_g_nfeat, &ctx->d_g_dpairs, &ctx->d_g_doff, &ctx->d_g_pdir,
                    &ctx->d_g_params, &ctx->d_g_m, &ctx->d_g_counts, &ctx->d_g_offsets, &ctx->d_g_total, &ctx->d_g_matches, &ctx->d_g_plan,
                    &ctx->d_g_inl, &ctx->d_g_inl_off, &ctx->d_mm_matches[0], &ctx->d_mm_matches[1], &ctx->d_mm_off[0],
                    &ctx->d_mm_off[1], &ctx->d_mm_counts, &ctx->d_mm_state, &ctx->d_mm_first, &ctx->d_mm_acc, &ctx->d_mm_keep,
                    &ctx->d_mm_total, &ctx->d_order, &ctx->d_dpairs2, &ctx->d_ecnt, &ctx->d_eoff, &ctx->d_etotal, &ctx->d_entries,
                    &ctx->d_out2, &ctx->d_ms, &ctx->d_out2s, &ctx->d_lo_inl, &ctx->d_lo_inl_pool, &ctx->d_nt_table_t, &ctx->d_wm_redo, &ctx->d_wm_total,
                    &ctx->d_wm_count, &ctx->d_pose_jobs};
  for (VerifyLane& L : ctx->lanes) {
    for (DevBuf* b : {&L.samples, &L.draws_end, &L.nmodels, &L.vcounts, &L.vsums, &L.models, &L.ework, &L.active, &L.vscratch, &L.lo_queue,
                      &L.lo_work, &L.lo_models, &L.lo_slots, &L.lo_ework, &L.tail_items, &L.tail_n, &L.lo_jobs, &L.job_list, &L.hyp_map})
      b->release();
    if (L.done) (void)hipEventDestroy(L.done);
    if (L.host_ctr) (void)hipHostFree(L.host_ctr);
    if (L.stream && L.stream != ctx->stream) (void)hipStreamDestroy(L.stream);
  }
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (hipEvent_t e : ctx->stage_ev)
    if (e) (void)hipEventDestroy(e);
  ctx->d_stage.release();
  if (ctx->vev0) (void)hipEventDestroy(ctx->vev0);
  if (ctx->vev1) (void)hipEventDestroy(ctx->vev1);
  for (DevBuf* b : bufs) b->release();
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* dsm_last_error(const dsm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int dsm_get_device_info(dsm_ctx* ctx, dsm_device_info* out) {
  if (!ctx || !out) return DSM_ERR_INVALID_ARGUMENT;
  hipDeviceProp_t pr;
  HIPCHK(ctx, hipGetDeviceProperties(&pr, ctx->device));
  memset(out, 0, sizeof(*out));
  snprintf(out->name, sizeof(out->name), "%s", pr.name);
  snprintf(out->arch, sizeof(out->arch), "%s", pr.gcnArchName);
  out->compute_units = pr.multiProcessorCount;
  out->clock_khz = pr.clockRate;
  out->memory_clock_khz = pr.memoryClockRate;
  out->memory_bus_bits = pr.memoryBusWidth;
  out->total_memory = pr.totalGlobalMem;
  out->l2_bytes = pr.l2CacheSize;
  out->lds_per_cu = (int32_t)pr.maxSharedMemoryPerMultiProcessor;
  return DSM_OK;
}

int dsm_sync(dsm_ctx* ctx) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return DSM_OK;
}

int dsm_set_debug_option(dsm_ctx* ctx, const char* key, const char* value) {
  if (!ctx || !key) return DSM_ERR_INVALID_ARGUMENT;
  bool known = false;
  for (const char* k : dsm_product_debug_keys) known = known || strcmp(k, key) == 0;
#ifdef DSM_CHECK_BUILD
  for (const char* k : dsm_check_debug_keys) known = known || strcmp(k, key) == 0;
#endif
  if (!known)
    return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "dsm_set_debug_option: unknown key (the cross-check switches exist in libdagsfm_mi355x_check.so only)");
  if (value)
    ctx->debug_options[key] = value;
  else
    ctx->debug_options.erase(key);
  return DSM_OK;
}

// Both stages keep their scratch between calls (a step re-uses it), so the budget is split: the matcher's chunk outputs get a
// quarter of it (at most their default 8 GiB), the verifier the rest.
static inline uint64_t match_budget_share(uint64_t budget) { return std::min<uint64_t>(8ull << 30, budget / 4); }

static std::vector<DevBuf*> scratch_buffers(dsm_ctx* ctx) {
  std::vector<DevBuf*> v = {&ctx->d_m, &ctx->d_ms, &ctx->d_entries, &ctx->d_out2, &ctx->d_out2s, &ctx->d_dpairs, &ctx->d_dpairs2, &ctx->d_doutoff, &ctx->d_pair_dir,
                            &ctx->d_order, &ctx->d_ecnt, &ctx->d_eoff, &ctx->d_vscratch};
  for (VerifyLane& L : ctx->lanes)
    for (DevBuf* b : {&L.samples, &L.draws_end, &L.nmodels, &L.vcounts, &L.vsums, &L.models, &L.ework, &L.active, &L.vscratch, &L.lo_queue, &L.lo_work,
                      &L.lo_models, &L.lo_slots, &L.lo_ework, &L.tail_items, &L.tail_n, &L.lo_jobs, &L.job_list, &L.hyp_map})
      v.push_back(b);
  return v;
}

int dsm_ctx_set_memory_budget(dsm_ctx* ctx, uint64_t bytes) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->memory_budget = bytes;
  if (bytes) {  // scratch held from an earlier, larger call is given back now: the next call allocates inside the budget
    uint64_t held = 0;
    for (DevBuf* b : scratch_buffers(ctx)) held += b->cap;
    if (held > bytes)
      for (DevBuf* b : scratch_buffers(ctx)) b->release();
  }
  return DSM_OK;
}

int dsm_ctx_memory_footprint(const dsm_ctx* cctx, uint64_t* resident_bytes, uint64_t* scratch_bytes) {
  if (!cctx) return DSM_ERR_INVALID_ARGUMENT;
  dsm_ctx* ctx = const_cast<dsm_ctx*>(cctx);
  uint64_t scratch = 0, resident = 0;
  for (DevBuf* b : scratch_buffers(ctx)) scratch += b->cap;
  for (DevBuf* b : {&ctx->d_stage, &ctx->d_desc, &ctx->d_rterm, &ctx->d_kp, &ctx->d_img_row0, &ctx->d_img_rows, &ctx->d_lut, &ctx->d_counts, &ctx->d_offsets, &ctx->d_matches,
                    &ctx->d_total, &ctx->d_etotal, &ctx->d_cams, &ctx->d_pairs_dev, &ctx->d_seeds, &ctx->d_tvg, &ctx->d_inl, &ctx->d_inl_counts, &ctx->d_inl_off,
                    &ctx->d_inl_compact, &ctx->d_inl_total, &ctx->d_nt_table, &ctx->d_nt_off, &ctx->d_nt_off_t, &ctx->d_pair_state, &ctx->d_pts_px,
                    &ctx->d_pts_norm, &ctx->d_reports, &ctx->d_masks, &ctx->d_fam_state, &ctx->d_sidx, &ctx->d_lo_inl, &ctx->d_nt_table_t, &ctx->d_wm_redo,
                    &ctx->d_wm_total, &ctx->d_wm_count, &ctx->d_lo_inl_pool, &ctx->d_pose_jobs})
    resident += b->cap;
  if (resident_bytes) *resident_bytes = resident;
  if (scratch_bytes) *scratch_bytes = scratch;
  return DSM_OK;
}

// dsm_set_images (append = false: the resident set is replaced) and dsm_append_images (append = true: the images
// already on the device stay where they are, the new ones get the next indices and only THEIR rows cross PCIe).
static int upload_images(dsm_ctx* ctx, bool append, uint32_t n_new, const uint32_t* n_feats, const uint8_t* const* desc,
                         const float* const* kp_xy, uint32_t kp_stride, const dsm_camera* cameras) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  if (n_new && (!n_feats || !desc)) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "null image arrays");
  if (kp_xy && kp_stride < 2) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "kp_stride must be >= 2");
  if (cameras)  // Camera::SetModelId CHECKs ExistsCameraModelWithId (camera.cc:52); never a silent default
    for (uint32_t i = 0; i < n_new; ++i)
      if (!cam_model_exists(cameras[i].model_id)) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "camera model id does not exist (0..10)");
  const uint32_t n_old = append ? ctx->n_images : 0u;
  const uint64_t rows_old = append ? ctx->total_rows : 0ull;
  if (append && n_old) {
    if ((kp_xy != nullptr) != ctx->have_kp) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "appended images must carry keypoints iff the resident ones do");
    if ((cameras != nullptr) != !ctx->cameras.empty()) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "appended images must carry cameras iff the resident ones do");
  }
  std::vector<uint32_t> row0(n_new), rows(n_new);
  uint64_t total = rows_old;
  for (uint32_t i = 0; i < n_new; ++i) {
    if (n_feats[i] > 0 && !desc[i]) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "null descriptor pointer");
    if (kp_xy && n_feats[i] > 0 && !kp_xy[i]) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "null keypoint pointer");
    const uint64_t r = ((uint64_t)n_feats[i] + 255) / 256 * 256;
    if (total + r > 0xffffff00ull) return fail(ctx, DSM_ERR_OUT_OF_RANGE, "too many feature rows for one context");
    row0[i] = (uint32_t)total;
    rows[i] = (uint32_t)r;
    total += r;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->matched = false;
  ctx->verified = false;
  dsm_retrieval_invalidate(ctx);
  const uint32_t n_images = n_old + n_new;
  const uint64_t rows_new = total - rows_old;
  // device buffers first (an allocation failure leaves the resident set as it was)
  if (append) {
    HIPCHK(ctx, ctx->d_desc.grow(std::max<uint64_t>(total, 1) * 128, rows_old * 128, ctx->stream));
    HIPCHK(ctx, ctx->d_rterm.grow(std::max<uint64_t>(total, 1) * 4, rows_old * 4, ctx->stream));
    HIPCHK(ctx, ctx->d_img_row0.grow(std::max<uint32_t>(n_images, 1) * 4, (size_t)n_old * 4, ctx->stream));
    HIPCHK(ctx, ctx->d_img_rows.grow(std::max<uint32_t>(n_images, 1) * 4, (size_t)n_old * 4, ctx->stream));
    if (kp_xy && total) HIPCHK(ctx, ctx->d_kp.grow(total * 16, rows_old * 16, ctx->stream));
  } else {  // replacing: free first, no second copy of the old set in memory
    HIPCHK(ctx, ctx->d_desc.reserve(std::max<uint64_t>(total, 1) * 128));
    HIPCHK(ctx, ctx->d_rterm.reserve(std::max<uint64_t>(total, 1) * 4));
    HIPCHK(ctx, ctx->d_img_row0.reserve(std::max<uint32_t>(n_images, 1) * 4));
    HIPCHK(ctx, ctx->d_img_rows.reserve(std::max<uint32_t>(n_images, 1) * 4));
    if (kp_xy && total) HIPCHK(ctx, ctx->d_kp.reserve(total * 16));
  }
  if (!append) {
    ctx->nfeat.clear();
    ctx->row0.clear();
    ctx->rows.clear();
    ctx->cameras.clear();
  }
  ctx->n_images = n_images;
  ctx->nfeat.insert(ctx->nfeat.end(), n_feats, n_feats + n_new);
  ctx->row0.insert(ctx->row0.end(), row0.begin(), row0.end());
  ctx->rows.insert(ctx->rows.end(), rows.begin(), rows.end());
  ctx->total_rows = total;
  if (cameras) ctx->cameras.insert(ctx->cameras.end(), cameras, cameras + n_new);
  ctx->have_kp = kp_xy != nullptr;
  if (n_new) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_img_row0.as<uint32_t>() + n_old, row0.data(), (size_t)n_new * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_img_rows.as<uint32_t>() + n_old, rows.data(), (size_t)n_new * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  // pinned already (judged by the first image that has rows): copied where they lie
  bool pinned = false;
  for (uint32_t i = 0; i < n_new; ++i)
    if (n_feats[i]) {
      pinned = host_pointer_is_pinned(desc[i]) && (!kp_xy || host_pointer_is_pinned(kp_xy[i]));
      break;
    }
  // first image whose padded rows end behind relative row r
  auto image_of_row = [&](uint64_t r) -> uint32_t {
    uint32_t lo = 0, hi = n_new;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if ((uint64_t)row0[mid] - rows_old + rows[mid] > r) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  if (rows_new && pinned) {
    // stage the padded u8 rows of the new images in a temporary device buffer, convert with K0
    DevBuf tmp;
    HIPCHK(ctx, tmp.reserve(rows_new * 128));
    HIPCHK(ctx, hipMemsetAsync(tmp.p, 0, rows_new * 128, ctx->stream));
    for (uint32_t i = 0; i < n_new; ++i) {
      if (!n_feats[i]) continue;
      HIPCHK(ctx, hipMemcpyAsync(tmp.as<uint8_t>() + ((uint64_t)row0[i] - rows_old) * 128, desc[i], (uint64_t)n_feats[i] * 128,
                                 hipMemcpyHostToDevice, ctx->stream));
    }
    launch_k0(tmp.as<uint8_t>(), ctx->d_desc.as<int8_t>() + rows_old * 128, ctx->d_rterm.as<int32_t>() + rows_old, rows_new, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    tmp.release();
  } else if (rows_new) {
    // pageable: the padded u8 rows stream through the pinned slots; k0_prepare converts every slot as it arrives
    auto fill = [&](uint8_t* dst, uint64_t begin, uint64_t end) {
      uint64_t at = begin;
      for (uint32_t i = image_of_row(begin / 128); i < n_new && at < end; ++i) {
        const uint64_t img0 = ((uint64_t)row0[i] - rows_old) * 128, data_end = img0 + (uint64_t)n_feats[i] * 128, img_end = img0 + (uint64_t)rows[i] * 128;
        if (at < data_end) {
          const uint64_t e = std::min(end, data_end);
          memcpy(dst + (at - begin), desc[i] + (at - img0), e - at);
          at = e;
        }
        if (at < end && at < img_end) {  // the zero rows that pad the image to 256
          const uint64_t e = std::min(end, img_end);
          memset(dst + (at - begin), 0, e - at);
          at = e;
        }
      }
    };
    auto arrived = [&](uint8_t* slot_dev, uint64_t begin, uint64_t end) -> int {
      launch_k0(slot_dev, ctx->d_desc.as<int8_t>() + rows_old * 128 + begin, ctx->d_rterm.as<int32_t>() + rows_old + begin / 128, (end - begin) / 128,
                ctx->stream);
      HIPCHK(ctx, hipGetLastError());
      return DSM_OK;
    };
    const int rc = staged_upload(ctx, rows_new * 128, nullptr, true, fill, arrived);
    if (rc != DSM_OK) return rc;
  }
  if (kp_xy && rows_new) {
    // FeatureKeypoint's float x, y (types.h:44-81) -> the double pairs the verification reads (Vector2d, matching.cc:585-592)
    auto convert = [&](uint8_t* dst8, uint64_t begin, uint64_t end) {
      double* dst = reinterpret_cast<double*>(dst8);
      uint64_t r = begin / 16;
      const uint64_t r_end = end / 16;
      for (uint32_t i = image_of_row(r); i < n_new && r < r_end; ++i) {
        const uint64_t img0 = (uint64_t)row0[i] - rows_old, data_end = img0 + n_feats[i], img_end = img0 + rows[i];
        const float* src = kp_xy[i];
        for (; r < r_end && r < data_end; ++r) {
          const uint64_t k = r - img0;
          dst[2 * (r - begin / 16) + 0] = (double)src[k * kp_stride + 0];
          dst[2 * (r - begin / 16) + 1] = (double)src[k * kp_stride + 1];
        }
        for (; r < r_end && r < img_end; ++r) {
          dst[2 * (r - begin / 16) + 0] = 0.0;
          dst[2 * (r - begin / 16) + 1] = 0.0;
        }
      }
    };
    auto nothing = [](uint8_t*, uint64_t, uint64_t) -> int { return DSM_OK; };
    const int rc = staged_upload(ctx, rows_new * 16, reinterpret_cast<uint8_t*>(ctx->d_kp.as<double>() + rows_old * 2), false, convert, nothing);
    if (rc != DSM_OK) return rc;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return DSM_OK;
}

int dsm_set_images(dsm_ctx* ctx, uint32_t n_images, const uint32_t* n_feats, const uint8_t* const* desc,
                   const float* const* kp_xy, uint32_t kp_stride, const dsm_camera* cameras) {
  return upload_images(ctx, false, n_images, n_feats, desc, kp_xy, kp_stride, cameras);
}

int dsm_append_images(dsm_ctx* ctx, uint32_t n_images, const uint32_t* n_feats, const uint8_t* const* desc,
                      const float* const* kp_xy, uint32_t kp_stride, const dsm_camera* cameras) {
  return upload_images(ctx, true, n_images, n_feats, desc, kp_xy, kp_stride, cameras);
}

int dsm_match_pairs(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* pairs, const dsm_match_options* options) {
  if (!ctx || !options || (n_pairs && !pairs)) return DSM_ERR_INVALID_ARGUMENT;
  // SiftMatchingOptions::Check, sift.cc:236-250
  if (!(options->max_ratio > 0.0) || !(options->max_distance > 0.0))
    return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "max_ratio and max_distance must be > 0");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  ctx->matched = false;
  ctx->verified = false;
  ctx->n_pairs = n_pairs;
  ctx->pairs.assign(pairs, pairs + (size_t)n_pairs * 2);
  ctx->k1_ms = 0.0;
  ctx->k1b_ms = 0.0;
  ctx->k1t_ms = 0.0;
  ctx->k1g_ms = 0.0;
  ctx->k1_launches = 0;
  ctx->total_matches = 0;
  const bool cross = options->cross_check != 0;
  for (uint32_t i = 0; i < n_pairs; ++i) {
    const uint32_t a = pairs[2 * i], b = pairs[2 * i + 1];
    if (a >= ctx->n_images || b >= ctx->n_images) return fail(ctx, DSM_ERR_OUT_OF_RANGE, "image index out of range");
    // max_num_matches is not consulted: MatchSiftFeaturesCPU (sift.cc:810-822), the path this library reproduces,
    // ignores it (only the SiftGPU matcher clamps, sift.cc:200-209); an image of any size is matched in full
  }
  HIPCHK(ctx, ctx->d_counts.reserve(std::max<uint32_t>(n_pairs, 1) * 4));
  HIPCHK(ctx, ctx->d_offsets.reserve(((size_t)n_pairs + 1) * 8));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_total.p, 0, 8, st));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_offsets.p, 0, 8, st));

  // Chunk the pair list so that the K1 output scratch stays below a fixed budget.
  uint64_t budget_rows = (8ull << 30) / 8;  // two int32 per row: the result and K1's second-best value for K1b
  // (dsm_ctx_set_memory_budget: the chunk's scratch is those 8 bytes per row + ~4 bytes per row of entry lists / second-pass outputs)
  if (ctx->memory_budget) budget_rows = std::max<uint64_t>(1, std::min<uint64_t>(budget_rows, match_budget_share(ctx->memory_budget) / 12));
  if (const char* e = ctx->dbg("DSM_MATCH_CHUNK_ROWS")) budget_rows = std::max<uint64_t>(1, strtoull(e, nullptr, 10));  // test hook
  std::vector<uint2> dpairs, dpairs2;
  std::vector<uint64_t> doff;
  std::vector<uint4> pdir;
  std::vector<uint32_t> order, cnt_b(ctx->n_images + 1);
  size_t ev_used = 0;
  uint32_t c0 = 0;
  while (c0 < n_pairs) {
    // chunk extent
    uint64_t rows_acc = 0;
    uint32_t c1 = c0;
    while (c1 < n_pairs) {
      const uint64_t r = ctx->rows[pairs[2 * c1]];
      if (c1 > c0 && rows_acc + r > budget_rows) break;
      rows_acc += r;
      ++c1;
    }
    const uint32_t nc = c1 - c0;
    // Pass 1, image a rows vs image b columns: pair k of the chunk IS directed pair k (its K1 output lives at
    // doff[k]); the launch order is counting-sorted by the column image b so that workgroups running at the same
    // time stream the same B image out of L2.
    dpairs.resize(nc);
    dpairs2.resize(nc);
    pdir.resize(nc);
    doff.resize(nc);
    order.resize(nc);
    std::fill(cnt_b.begin(), cnt_b.end(), 0u);
    uint64_t off = 0;
    uint32_t max_rb = 0;
    for (uint32_t k = 0; k < nc; ++k) {
      const uint32_t a = pairs[2 * (c0 + k)], b = pairs[2 * (c0 + k) + 1];
      dpairs[k] = make_uint2(a, b);
      dpairs2[k] = make_uint2(b, a);  // pass 2: gathered rows of image b vs the columns of image a
      pdir[k] = make_uint4(k, 0, ctx->nfeat[a], ctx->nfeat[b]);
      doff[k] = off;
      off += ctx->rows[a];
      max_rb = std::max(max_rb, ctx->rows[a] / 256);
      cnt_b[b + 1]++;
    }
    for (uint32_t k = 0; k < ctx->n_images; ++k) cnt_b[k + 1] += cnt_b[k];
    for (uint32_t k = 0; k < nc; ++k) order[cnt_b[dpairs[k].y]++] = k;
    HIPCHK(ctx, ctx->d_dpairs.reserve(std::max<uint32_t>(nc, 1) * sizeof(uint2)));
    HIPCHK(ctx, ctx->d_order.reserve(std::max<uint32_t>(nc, 1) * 4));
    HIPCHK(ctx, ctx->d_doutoff.reserve(std::max<uint32_t>(nc, 1) * 8));
    HIPCHK(ctx, ctx->d_pair_dir.reserve(std::max<uint32_t>(nc, 1) * sizeof(uint4)));
    HIPCHK(ctx, ctx->d_m.reserve(std::max<uint64_t>(off, 1) * 4));
    HIPCHK(ctx, ctx->d_ms.reserve(std::max<uint64_t>(off, 1) * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_dpairs.p, dpairs.data(), nc * sizeof(uint2), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_order.p, order.data(), nc * 4, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_doutoff.p, doff.data(), nc * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_pair_dir.p, pdir.data(), nc * sizeof(uint4), hipMemcpyHostToDevice, st));

    K1Params k1;
    k1.desc = ctx->d_desc.as<int8_t>();
    k1.rterm = ctx->d_rterm.as<int32_t>();
    k1.dpairs = ctx->d_dpairs.as<uint2>();
    k1.img_row0 = ctx->d_img_row0.as<uint32_t>();
    k1.img_rows = ctx->d_img_rows.as<uint32_t>();
    k1.d_out_off = ctx->d_doutoff.as<uint64_t>();
    k1.lut = ctx->d_lut.as<float>();
    k1.max_ratio = (float)options->max_ratio;        // narrowed as at sift.cc:164-166
    k1.max_distance = (float)options->max_distance;
    k1.out = ctx->d_m.as<int32_t>();
    k1.out_s = ctx->d_ms.as<int32_t>();
    k1.order = ctx->d_order.as<uint32_t>();
    k1.entries = nullptr;
    k1.e_off = nullptr;
    k1.e_cnt = nullptr;
    while (ctx->ev.size() < ev_used + 6) {
      hipEvent_t e;
      HIPCHK(ctx, hipEventCreate(&e));
      ctx->ev.push_back(e);
    }
    // DSM_K1_DOT4=1: the LDS-tiled v_dot4 variant of pass 1 (comparison runs only, profiles/r02_k1_variants.md)
#ifdef DSM_CHECK_BUILD
    const bool k1_dot4 = ctx->dbg("DSM_K1_DOT4") != nullptr;
#else
    const bool k1_dot4 = false;
#endif
    HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used], st));
#ifdef DSM_CHECK_BUILD
    if (k1_dot4)
      launch_k1_dot4(k1, nc, max_rb, st);
    else
#endif
      launch_k1(k1, nc, max_rb, st);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used + 1], st));
    if (!k1_dot4) launch_k1_resolve(k1, nc, max_rb, st);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used + 2], st));
    if (max_rb) ctx->k1_launches++;

    // one-way matches (i1, matches12[i1]) of every pair, compact and in ascending i1: without the cross-check they
    // are the result (sift.cc:188-196), with it they are the entry list of pass 2
    K2Params k2;
    k2.pair_dir = ctx->d_pair_dir.as<uint4>();
    k2.d_out_off = ctx->d_doutoff.as<uint64_t>();
    k2.m = ctx->d_m.as<int32_t>();
    k2.cross_check = 0;
    k2.matches = nullptr;
    uint64_t total = 0;
    if (!cross) {
      // (no pass 2: the gather interval [3, 4) is empty and the compaction below falls into the tail interval [4, 5), so that
      // the four timers still sum to the call -- ADVICE r05)
      HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used + 3], st));
      HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used + 4], st));
      k2.counts = ctx->d_counts.as<uint32_t>() + c0;
      k2.offsets = ctx->d_offsets.as<uint64_t>() + c0;
      launch_k2(k2, nc, false, st);
      HIPCHK(ctx, hipGetLastError());
      launch_scan(ctx->d_counts.as<uint32_t>() + c0, ctx->d_offsets.as<uint64_t>() + c0, nc, ctx->d_total.as<uint64_t>(), st);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipMemcpyAsync(&total, ctx->d_total.p, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      HIPCHK(ctx, ctx->d_matches.grow(std::max<uint64_t>(total, 1) * 8, ctx->total_matches * 8, st));
      k2.matches = ctx->d_matches.as<uint32_t>();
      launch_k2(k2, nc, true, st);
      HIPCHK(ctx, hipGetLastError());
    } else {
      HIPCHK(ctx, ctx->d_ecnt.reserve(std::max<uint32_t>(nc, 1) * 4));
      HIPCHK(ctx, ctx->d_eoff.reserve(((size_t)nc + 1) * 8));
      HIPCHK(ctx, ctx->d_etotal.reserve(8));
      HIPCHK(ctx, hipMemsetAsync(ctx->d_etotal.p, 0, 8, st));
      k2.counts = ctx->d_ecnt.as<uint32_t>();
      k2.offsets = ctx->d_eoff.as<uint64_t>();
      launch_k2(k2, nc, false, st);
      HIPCHK(ctx, hipGetLastError());
      launch_scan(ctx->d_ecnt.as<uint32_t>(), ctx->d_eoff.as<uint64_t>(), nc, ctx->d_etotal.as<uint64_t>(), st);
      HIPCHK(ctx, hipGetLastError());
      uint64_t etotal = 0;
      HIPCHK(ctx, hipMemcpyAsync(&etotal, ctx->d_etotal.p, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      HIPCHK(ctx, ctx->d_entries.reserve(std::max<uint64_t>(etotal, 1) * 8));
      HIPCHK(ctx, ctx->d_out2.reserve(std::max<uint64_t>(etotal, 1) * 4));
      HIPCHK(ctx, ctx->d_out2s.reserve(std::max<uint64_t>(etotal, 1) * 4));
      k2.matches = ctx->d_entries.as<uint32_t>();
      launch_k2(k2, nc, true, st);
      HIPCHK(ctx, hipGetLastError());
      // Pass 2: FindBestMatchesOneWay(dists.transpose()) (sift.cc:175-176) for the rows matches12 points at
      HIPCHK(ctx, ctx->d_dpairs2.reserve(std::max<uint32_t>(nc, 1) * sizeof(uint2)));
      HIPCHK(ctx, hipMemcpyAsync(ctx->d_dpairs2.p, dpairs2.data(), nc * sizeof(uint2), hipMemcpyHostToDevice, st));
      K1Params g = k1;
      g.dpairs = ctx->d_dpairs2.as<uint2>();
      g.order = nullptr;  // list order: the column image a of consecutive pairs rarely changes in the lists the callers build
      g.entries = ctx->d_entries.as<uint2>();
      g.e_off = ctx->d_eoff.as<uint64_t>();
      g.e_cnt = ctx->d_ecnt.as<uint32_t>();
      g.out = ctx->d_out2.as<int32_t>();
      g.out_s = ctx->d_out2s.as<int32_t>();
      HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used + 3], st));
      if (etotal) {
        launch_k1(g, nc, max_rb, st);  // a pair has at most rows(a) entries
        HIPCHK(ctx, hipGetLastError());
      }
      HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used + 4], st));
      if (etotal) {
        launch_k1_resolve(g, nc, max_rb, st);
        HIPCHK(ctx, hipGetLastError());
      }
      K2eParams ke;
      ke.entries = g.entries;
      ke.e_off = g.e_off;
      ke.e_cnt = g.e_cnt;
      ke.out2 = ctx->d_out2.as<int32_t>();
      ke.counts = ctx->d_counts.as<uint32_t>() + c0;
      ke.offsets = ctx->d_offsets.as<uint64_t>() + c0;
      ke.matches = nullptr;
      launch_k2_entries(ke, nc, false, st);
      HIPCHK(ctx, hipGetLastError());
      launch_scan(ctx->d_counts.as<uint32_t>() + c0, ctx->d_offsets.as<uint64_t>() + c0, nc, ctx->d_total.as<uint64_t>(), st);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipMemcpyAsync(&total, ctx->d_total.p, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      HIPCHK(ctx, ctx->d_matches.grow(std::max<uint64_t>(total, 1) * 8, ctx->total_matches * 8, st));
      ke.matches = ctx->d_matches.as<uint32_t>();
      launch_k2_entries(ke, nc, true, st);
      HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[ev_used + 5], st));
    ev_used += 6;
    ctx->total_matches = total;
    // the scratch of this chunk is reused by the next one
    HIPCHK(ctx, hipStreamSynchronize(st));
    c0 = c1;
  }
  HIPCHK(ctx, hipStreamSynchronize(st));
  for (size_t k = 0; k + 5 < ev_used; k += 6) {
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[k + 4], ctx->ev[k + 5]));
    ctx->k1t_ms += ms;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[k + 2], ctx->ev[k + 3]));  // the entry list of pass 2 (k2 + scan + the host's wait)
    ctx->k1t_ms += ms;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[k], ctx->ev[k + 1]));
    ctx->k1_ms += ms;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[k + 1], ctx->ev[k + 2]));
    ctx->k1b_ms += ms;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[k + 3], ctx->ev[k + 4]));
    ctx->k1g_ms += ms;
  }
  ctx->matched = true;
  return DSM_OK;
}

int dsm_set_matches(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* pairs, const uint64_t* offsets, const uint32_t* matches) {
  if (!ctx || (n_pairs && (!pairs || !offsets))) return DSM_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->matched = false;
  ctx->verified = false;
  const uint64_t total = n_pairs ? offsets[n_pairs] : 0;
  if (total && !matches) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "null matches");
  std::vector<uint32_t> counts(n_pairs);
  for (uint32_t i = 0; i < n_pairs; ++i) {
    const uint32_t a = pairs[2 * i], b = pairs[2 * i + 1];
    if (a >= ctx->n_images || b >= ctx->n_images) return fail(ctx, DSM_ERR_OUT_OF_RANGE, "image index out of range");
    if (offsets[i + 1] < offsets[i]) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    counts[i] = (uint32_t)(offsets[i + 1] - offsets[i]);
    for (uint64_t k = offsets[i]; k < offsets[i + 1]; ++k)
      if (matches[2 * k] >= ctx->nfeat[a] || matches[2 * k + 1] >= ctx->nfeat[b])
        return fail(ctx, DSM_ERR_OUT_OF_RANGE, "match index out of range");
  }
  ctx->n_pairs = n_pairs;
  ctx->pairs.assign(pairs, pairs + (size_t)n_pairs * 2);
  ctx->total_matches = total;
  ctx->k1_ms = 0.0;
  ctx->k1b_ms = 0.0;
  ctx->k1t_ms = 0.0;
  ctx->k1_launches = 0;
  HIPCHK(ctx, ctx->d_counts.reserve(std::max<uint32_t>(n_pairs, 1) * 4));
  HIPCHK(ctx, ctx->d_offsets.reserve(((size_t)n_pairs + 1) * 8));
  HIPCHK(ctx, ctx->d_matches.reserve(std::max<uint64_t>(total, 1) * 8));
  const uint64_t zero = 0;
  HIPCHK(ctx, hipMemcpy(ctx->d_offsets.p, n_pairs ? offsets : &zero, ((size_t)n_pairs + 1) * 8, hipMemcpyHostToDevice));
  if (n_pairs) HIPCHK(ctx, hipMemcpy(ctx->d_counts.p, counts.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice));
  if (total) HIPCHK(ctx, hipMemcpy(ctx->d_matches.p, matches, total * 8, hipMemcpyHostToDevice));
  ctx->matched = true;
  return DSM_OK;
}

int dsm_get_match_counts(dsm_ctx* ctx, uint32_t* counts) {
  if (!ctx || !counts) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->matched) return fail(ctx, DSM_ERR_NOT_READY, "dsm_match_pairs has not run");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->n_pairs) HIPCHK(ctx, hipMemcpy(counts, ctx->d_counts.p, (size_t)ctx->n_pairs * 4, hipMemcpyDefault));
  return DSM_OK;
}

int dsm_get_matches(dsm_ctx* ctx, uint64_t* offsets, uint32_t* matches, uint64_t matches_capacity) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->matched) return fail(ctx, DSM_ERR_NOT_READY, "dsm_match_pairs has not run");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (offsets) HIPCHK(ctx, hipMemcpy(offsets, ctx->d_offsets.p, ((size_t)ctx->n_pairs + 1) * 8, hipMemcpyDefault));
  if (matches) {
    if (matches_capacity < ctx->total_matches) return fail(ctx, DSM_ERR_OUT_OF_RANGE, "matches buffer too small");
    if (ctx->total_matches)
      HIPCHK(ctx, hipMemcpy(matches, ctx->d_matches.p, ctx->total_matches * 8, hipMemcpyDefault));
  }
  return DSM_OK;
}

int dsm_match_sift_features(dsm_ctx* ctx, const dsm_match_options* options, const uint8_t* desc1, uint32_t n1,
                            const uint8_t* desc2, uint32_t n2, uint32_t* matches, uint32_t* n_matches) {
  if (!ctx || !options || !n_matches) return DSM_ERR_INVALID_ARGUMENT;
  *n_matches = 0;
  if (!ctx->leaf) {
    int rc = dsm_ctx_create(ctx->device, &ctx->leaf);
    if (rc != DSM_OK) return fail(ctx, rc, dsm_last_error(nullptr));
  }
  dsm_ctx* lf = ctx->leaf;
  lf->debug_options = ctx->debug_options;
  const uint32_t nf[2] = {n1, n2};
  const uint8_t* dp[2] = {desc1, desc2};
  int rc = dsm_set_images(lf, 2, nf, dp, nullptr, 0, nullptr);
  if (rc != DSM_OK) return fail(ctx, rc, lf->err.c_str());
  const uint32_t pr[2] = {0, 1};
  rc = dsm_match_pairs(lf, 1, pr, options);
  if (rc != DSM_OK) return fail(ctx, rc, lf->err.c_str());
  uint64_t offs[2] = {0, 0};
  rc = dsm_get_matches(lf, offs, nullptr, 0);
  if (rc != DSM_OK) return fail(ctx, rc, lf->err.c_str());
  if (offs[1] && !matches) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "null matches buffer");
  rc = dsm_get_matches(lf, nullptr, matches, options->cross_check ? std::min<uint64_t>(n1, n2) : n1);
  if (rc != DSM_OK) return fail(ctx, rc, lf->err.c_str());
  *n_matches = (uint32_t)offs[1];
  return DSM_OK;
}


// ------------------------------------------------------------------------------------ verification

// RANSAC::ComputeNumTrials, /root/reference/src/optim/ransac.h:150-167, evaluated with the host libm.
// static_cast<size_t> of a non-finite / negative quotient is what x86-64 cvttsd2si makes of it
// (0x8000...): "never stop"; stored saturated.
static uint32_t host_num_trials(uint64_t num_inliers, uint64_t num_samples, double confidence, int min_samples) {
  const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
  const double nom = 1 - confidence;
  if (nom <= 0) return 0xffffffffu;
  const double denom = 1 - pow(inlier_ratio, min_samples);
  if (denom <= 0) return 1;
  const double v = ceil(log(nom) / log(denom));
  if (!(v >= 0.0) || !(v < 4294967295.0)) return 0xffffffffu;
  return (uint32_t)v;
}

static uint32_t ransac_ctor_max_trials(const dsm_two_view_options* o, double min_inlier_ratio, int min_samples) {
  // RANSAC ctor, ransac.h:135-148
  const uint64_t kNumSamples = 100000;
  const uint32_t dyn = host_num_trials((uint64_t)(min_inlier_ratio * kNumSamples), kNumSamples, o->confidence, min_samples);
  const uint64_t mx = o->max_num_trials;
  return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(mx, dyn), 0xffffffffull);
}

static const int kMinSamples[4] = {5, 7, 4, 1};

// makes sure the ComputeNumTrials tables of the E / F / H estimators exist on the device for the given match counts
// (3 x (N + 1) entries per distinct count N; the translation estimator's tables are built on demand, below)
static int ensure_nt_tables(dsm_ctx* ctx, const dsm_two_view_options* o, const std::vector<uint32_t>& counts, uint32_t n_max) {
  if (ctx->nt_confidence != o->confidence) {
    ctx->nt_confidence = o->confidence;
    ctx->nt_table.clear();
    ctx->nt_off.clear();
    ctx->nt_table_t.clear();
    ctx->nt_off_t.clear();
    ctx->nt_dirty = ctx->nt_dirty_t = true;
  }
  if (ctx->nt_off.size() < (size_t)n_max + 1) {
    ctx->nt_off.resize((size_t)n_max + 1, 0);
    ctx->nt_dirty = true;
  }
  if (ctx->nt_off_t.size() < (size_t)n_max + 1) {
    ctx->nt_off_t.resize((size_t)n_max + 1, 0);
    ctx->nt_dirty_t = true;
  }
  for (uint32_t N : counts) {
    if (N > n_max || ctx->nt_off[N] != 0) continue;
    ctx->nt_off[N] = ctx->nt_table.size() + 1;  // +1: 0 means absent
    for (int f = 0; f < 3; ++f)
      for (size_t k = 0; k <= N; ++k) ctx->nt_table.push_back(host_num_trials(k, N, o->confidence, kMinSamples[f]));
    ctx->nt_dirty = true;
  }
  if (ctx->nt_dirty) {
    std::vector<uint64_t> off(ctx->nt_off.size());
    for (size_t i = 0; i < off.size(); ++i) off[i] = ctx->nt_off[i] ? ctx->nt_off[i] - 1 : 0;
    HIPCHK(ctx, ctx->d_nt_table.reserve(std::max<size_t>(ctx->nt_table.size(), 1) * 4));
    HIPCHK(ctx, ctx->d_nt_off.reserve(off.size() * 8));
    HIPCHK(ctx, hipMemcpy(ctx->d_nt_table.p, ctx->nt_table.data(), ctx->nt_table.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_nt_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    ctx->nt_dirty = false;
  }
  return DSM_OK;
}

// The translation estimator's tables (TwoViewGeometry::DetectWatermark, two_view_geometry.cc:541-549: RANSAC over the
// INLIER points, so the sample count is an inlier count nobody knows before the run): only for the counts in `totals`,
// N + 1 entries each.  Round 2 tabulated every N up to the largest match count up front -- O(n^2) entries, 134 MB at
// 8 192 matches, GBs at the reference's max_num_matches -- although a watermark suspect is rare.  Uploads when
// anything changed (or nothing has been uploaded yet: the kernels read nt_off_t[N] == 0 as "not built").
static int ensure_nt_tables_t(dsm_ctx* ctx, const dsm_two_view_options* o, const std::vector<uint32_t>& totals) {
  for (uint32_t N : totals) {
    if (N >= ctx->nt_off_t.size() || ctx->nt_off_t[N] != 0) continue;
    ctx->nt_off_t[N] = ctx->nt_table_t.size() + 1;
    for (size_t k = 0; k <= N; ++k) ctx->nt_table_t.push_back(host_num_trials(k, N, o->confidence, kMinSamples[3]));
    ctx->nt_dirty_t = true;
  }
  if (ctx->nt_dirty_t) {
    HIPCHK(ctx, ctx->d_nt_table_t.reserve(std::max<size_t>(ctx->nt_table_t.size(), 1) * 4));
    HIPCHK(ctx, ctx->d_nt_off_t.reserve(std::max<size_t>(ctx->nt_off_t.size(), 1) * 8));
    HIPCHK(ctx, hipMemcpy(ctx->d_nt_table_t.p, ctx->nt_table_t.data(), ctx->nt_table_t.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_nt_off_t.p, ctx->nt_off_t.data(), ctx->nt_off_t.size() * 8, hipMemcpyHostToDevice));
    ctx->nt_dirty_t = false;
  }
  return DSM_OK;
}

// a lane's counter block: [k_sample's hand-out counters][32 classic words][the replay's][k_lo_prepare's][k_verify_final's] -- every
// area 64 counters on a 128-byte line each (grab_seg, verify_kernels.hip GRAB_*); laid out so that the counters a phase resets are
// contiguous: one fill per phase, as before
#define LANE_CTR_AREA (64 * 128)
#define LANE_CTR_BYTES (LANE_CTR_AREA + 128 + 3 * LANE_CTR_AREA)
// What the lanes of one dsm_verify_pairs call share
struct VerifyPlan {
  uint32_t batch[3] = {0, 0, 0}, bmax = 0;
  uint64_t bm_max = 0;
  uint32_t n_lanes = 1;
  uint32_t begin[DSM_VERIFY_MAX_LANES] = {0}, end[DSM_VERIFY_MAX_LANES] = {0}, chunk[DSM_VERIFY_MAX_LANES] = {0};  // per lane
  int dev_cus = 256;
  bool inline_lo = false;
  uint32_t lo_tail = 0;  // queue length at and below which the batched schedule finishes a round inline
  uint32_t grid_div = 1;
  uint32_t replay_grid_mul = 1;
  bool tail_items = true;   // the tail of a round (<= lo_tail pairs queued) as an item pass; false: the inline tail of round 2
  bool item_mode = false;   // item passes from the start of every round (short pair lists)
  uint32_t item_rounds = 0xffffffffu;  // ... of the first item_rounds rounds of a family only (later rounds: the chain)
  bool hyp_compact = true;  // the lane-per-hypothesis solvers of E / F take later rounds' hypotheses 64 per wave across the pairs (hyp_of_lane)
};

#define LANECHK(L, call)                                              \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      (L).err = std::string(#call) + ": " + hipGetErrorString(e_);    \
      (L).rc = DSM_ERR_HIP;                                           \
      return;                                                         \
    }                                                                 \
  } while (0)

// One lane (host thread + stream) of the phase-split pipeline: its range of the pair list, chunk by chunk.
static void verify_lane_run(dsm_ctx* ctx, uint32_t li, VerifyParams vp, VerifyPlan plan) {
  VerifyLane& L = ctx->lanes[li];
  LANECHK(L, hipSetDevice(ctx->device));
  hipStream_t st = L.stream;
  // the lane's counter fills and read-backs as one-wave kernels, not as the runtime's blit kernels (verify_kernels.hip k_lane_counters:
  // an eight-wave blit workgroup starves for tens of ms behind the other lane's register-filling kernels)
  auto ctr_zero = [st](void* at, size_t bytes) {
    launch_lane_counters(nullptr, nullptr, 0, at, bytes, st);
    return hipGetLastError();
  };
  auto ctr_read = [st](uint32_t* host, const void* src, uint32_t bytes) {
    launch_lane_counters(src, host, bytes, nullptr, 0, st);
    return hipGetLastError();
  };
  LANECHK(L, hipStreamWaitEvent(st, ctx->vev0, 0));
  LANECHK(L, ctr_zero(L.active.p, LANE_CTR_BYTES));
  const uint32_t chunk = plan.chunk[li];
  vp.scratch = L.vscratch.as<double>();
  vp.samples = L.samples.as<uint32_t>();
  vp.draws_end = L.draws_end.as<uint32_t>();
  vp.nmodels = L.nmodels.as<int32_t>();
  vp.counts = L.vcounts.as<int32_t>();
  vp.sums = L.vsums.as<double>();
  vp.models = L.models.as<double>();
  vp.e_work = L.ework.as<double>();
  vp.active_count = L.active.as<uint32_t>() + LANE_CTR_AREA / 4;
  vp.lo_work = L.lo_work.as<double>();
  vp.lo_models = L.lo_models.as<double>();
  vp.lo_slots = L.lo_slots.as<double>();
  vp.lo_ework = L.lo_ework.as<double>();
  char* const actr = static_cast<char*>(L.active.p) + LANE_CTR_AREA;  // the classic words
  for (uint64_t c0 = plan.begin[li]; c0 < plan.end[li]; c0 += chunk) {
    vp.pair0 = (uint32_t)c0;
    vp.n_chunk = (uint32_t)std::min<uint64_t>(chunk, plan.end[li] - c0);
    // persistent wave-per-pair grids: sized for the whole chip when a lane has it to itself, for its share when the
    // lanes are many (DSM_VERIFY_GRID_DIV, experiment: a lane's persistent grid must not starve the short launches of the others)
    const uint32_t nb_light = std::min<uint32_t>(vp.n_chunk, std::max<uint32_t>(64u, (uint32_t)plan.dev_cus * 32u / plan.grid_div));
    const uint32_t nb_heavy = std::min<uint32_t>(vp.n_chunk, std::max<uint32_t>(64u, (uint32_t)plan.dev_cus * 16u / plan.grid_div));
    // the replay scans are latency-bound and light (four waves per SIMD fit): their grid may be larger than the heavy kernels'
    // (a lane's scratch holds dev_cus * 16 workgroups)
    const uint32_t nb_replay = std::min<uint32_t>(vp.n_chunk, std::min<uint32_t>((uint32_t)plan.dev_cus * 4u * DSM_REPLAY_WAVES, std::max<uint32_t>(64u, (uint32_t)plan.dev_cus * 4u * DSM_REPLAY_WAVES * plan.replay_grid_mul / plan.grid_div)));
    vp.batch = 0;
    launch_vp_prep(vp, nb_light, st);
    LANECHK(L, hipGetLastError());
    for (int f = 0; f < 3; ++f) {
      vp.batch = plan.batch[f];
      for (uint32_t round = 0; round < 100000; ++round) {
        // counters of the lane (uint32 at actr): [0] pairs still active after the round, [16] replay work counter, [17]
        // k_verify_final's, [18] k_sample's, [19] k_lo_prepare's, [20] / [21] lengths of the two alternating queues, [22] /
        // [23] problems for the general LO kernels, [24] jobs of an item pass, [25] problems for the larger register prepare.  Every launch below finds its counters zeroed;
        // counters that are dead at that point are zeroed along with them, so that each phase costs ONE fill, not one per
        // counter (a fill is a kernel launch: 33 of them made 0.7 ms of the 8.7 ms a 1 225-pair list takes)
        if (vp.stats) {  // DSM_VERIFY_DEBUG keeps its statistics in [1] - [13] across the rounds
          LANECHK(L, hipMemsetAsync(actr - LANE_CTR_AREA, 0, LANE_CTR_AREA + 4, st));
          LANECHK(L, hipMemsetAsync(actr + 64, 0, 40, st));  // [16] - [25]
        } else {
          LANECHK(L, ctr_zero(actr - LANE_CTR_AREA, LANE_CTR_AREA + 112));  // round start: k_sample's hand-out counters and [0] are live, the rest ([1] - [27]) is dead here
        }
        // E / F after a pair's first round (whole waves): the solvers take the round's hypotheses from k_sample's list, 64 per wave
        // across the pairs (verify_kernels.hip hyp_of_lane); its segment counters are k_sample's hand-out counters, zeroed above
        vp.hyp_map = (plan.hyp_compact && f != 2 && round > 0) ? L.hyp_map.as<uint32_t>() : nullptr;
        vp.hyp_seg_cap = (uint32_t)(((size_t)(chunk + 63u) / 64u) * vp.batch);
        launch_vp_sample(vp, f, nb_light, st);
        launch_vp_solve_score(vp, f, st);
        uint32_t active = 0;
        if (plan.inline_lo) {
          LANECHK(L, ctr_zero(actr + 64, 16));  // k_replay's work counter ([17] - [19] are unused words)
          launch_vp_replay(vp, f, nb_heavy, st);
          LANECHK(L, hipGetLastError());
          LANECHK(L, ctr_read(L.host_ctr, actr, 4));
          LANECHK(L, hipStreamSynchronize(st));
          active = L.host_ctr[0];
        } else {
          // replay until every pair of the chunk has either consumed the batch or stopped; a pair that reaches a
          // local optimisation is suspended onto the queue, the optimisation runs for the whole queue, and the
          // next replay launch works through exactly that queue
          uint32_t* queues = L.lo_queue.as<uint32_t>();
          vp.lo_queue_g = queues + (size_t)2 * chunk;
          vp.worklist = nullptr;
          vp.n_work = vp.n_chunk;
          vp.tail_items = L.tail_items.as<TailItem>();
          vp.tail_n = L.tail_n.as<uint32_t>();
          vp.lo_jobs = nullptr;  // set only in the copies the item passes launch with
          // Two ways through a round's local optimisations (verify_kernels.hip, "item passes"):
          //   chain  k_replay_lo suspends a pair at every step, the batched kernels run for all suspended pairs, the next
          //          replay resumes them: one iteration per step of the slowest pair, no speculative work
          //   items  every step a pair can still reach is computed at once (jobs), the replay looks the outcomes up:
          //          one pass per round, ~2x the local-optimisation work
          // plan.item_mode: item passes from the start of a round (short lists: every iteration of the chain is
          // latency-bound there); otherwise the chain, and item passes once <= lo_tail pairs are left in the queue.
          int mode = 0;
          bool pass_items = plan.item_mode && round < plan.item_rounds;
          for (uint32_t cur = 0;; cur ^= 1u) {
            uint32_t* host_ctr = L.host_ctr;
            if (pass_items) {
              // [22], [23]: problems for the general LO kernels; [24]: jobs; [19]: k_lo_prepare's work counter (used after
              // the two read-backs below); [20], [21]: the queue lengths, already read by the host
              VerifyParams vj = vp;
              vj.lo_jobs = L.lo_jobs.as<LoJob>();
              vj.job_list = L.job_list.as<uint32_t>();
              vj.lo_inl_pool = ctx->d_lo_inl_pool.as<uint32_t>();
              vj.lo_queue_g = queues + (size_t)2 * chunk;  // (the general kernels' list: job slots here)
              LANECHK(L, ctr_zero(actr + 64, 64 + 2 * LANE_CTR_AREA));  // from [16] (dead here, like [17] - [18]) through k_lo_prepare's hand-out counters (the replay's, in between, are dead too)
              launch_vp_items_enum(vj, f, st);
              LANECHK(L, hipGetLastError());
              LANECHK(L, ctr_read(host_ctr, actr, 128));
              LANECHK(L, hipStreamSynchronize(st));
              const uint32_t n_jobs = host_ctr[24];
              if (n_jobs) {
                vj.worklist = vj.job_list;
                vj.n_work = n_jobs;
                launch_vp_items_inliers(vj, f, nb_heavy, st);
                LANECHK(L, hipGetLastError());
                LANECHK(L, ctr_read(host_ctr, actr, 128));
                LANECHK(L, hipStreamSynchronize(st));
                launch_vp_local_opt(vj, f, nb_heavy, host_ctr[22], host_ctr[23], host_ctr[25], st);  // ([19] zeroed above)
                LANECHK(L, hipGetLastError());
                launch_vp_items_outcome(vj, f, nb_heavy, st);
                LANECHK(L, hipGetLastError());
              }
              L.lo_iters[f]++;
              mode = 2;
              pass_items = false;
            }
            uint32_t* cnt_dev = vp.active_count + 20 + cur;
            // this queue's length [20 + cur], the work counter [16], [22] / [23]: queued problems for the general LO kernels;
            // [17] - [19], the other queue's length (read by the host after the launch that filled it) and [24] are dead here
            LANECHK(L, ctr_zero(actr + 64, 64 + LANE_CTR_AREA));  // ... and the replay's hand-out counters behind them: one fill
            vp.lo_queue = queues + (size_t)cur * chunk;
            vp.lo_count = cnt_dev;
            launch_vp_replay_lo(vp, f, std::min<uint32_t>(nb_replay, vp.n_work), mode, st);
            LANECHK(L, hipGetLastError());
            LANECHK(L, ctr_read(host_ctr, actr, 128));
            LANECHK(L, hipStreamSynchronize(st));
            active = host_ctr[0];
            const uint32_t nq = host_ctr[20 + cur];
            if (nq == 0) break;
            vp.worklist = vp.lo_queue;
            vp.n_work = nq;
            if (mode == 2 || (nq <= plan.lo_tail && plan.tail_items)) {  // the pairs still queued: an item pass over them
              pass_items = true;
              continue;
            }
            L.lo_iters[f]++;
            if (nq <= plan.lo_tail && f != 0) {  // DSM_LO_TAIL_MODE=inline (not for E: its inline solver does not fit the registers)
              // round-2 form of the tail: the wave runs its pair's remaining local optimisations inline, one after the other
              LANECHK(L, ctr_zero(actr + 64, 16));  // work counter [16]
              launch_vp_replay_lo(vp, f, std::min<uint32_t>(nb_heavy, nq), 1, st);
              LANECHK(L, hipGetLastError());
              LANECHK(L, ctr_read(host_ctr, actr, 4));
              LANECHK(L, hipStreamSynchronize(st));
              active = host_ctr[0];
              break;
            }
            LANECHK(L, ctr_zero(actr + 128 + LANE_CTR_AREA, LANE_CTR_AREA));  // k_lo_prepare's hand-out counters
            launch_vp_local_opt(vp, f, nb_heavy, host_ctr[22], host_ctr[23], host_ctr[25], st);
            LANECHK(L, hipGetLastError());
          }
        }
        L.rounds[f]++;
        if (active == 0) break;
      }
    }
    LANECHK(L, ctr_zero(actr + 128 + 2 * LANE_CTR_AREA, LANE_CTR_AREA));  // k_verify_final's hand-out counters
    launch_vp_final(vp, nb_heavy, st);
    LANECHK(L, hipGetLastError());
  }
  if (vp.stats) LANECHK(L, hipMemcpyAsync(L.dbg, actr, 128, hipMemcpyDeviceToHost, st));
  LANECHK(L, hipEventRecord(L.done, st));
  LANECHK(L, hipStreamSynchronize(st));
}

// core: verifies n_pairs pairs whose matches/keypoints/cameras are already on the device
static int verify_core(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* d_pairs, const uint64_t* d_match_off,
                       const uint32_t* d_matches, uint64_t total_matches, const std::vector<uint32_t>& counts,
                       const double* d_kp, const uint32_t* d_img_row0, const dsm_camera* d_cams,
                       const dsm_two_view_options* o, const uint32_t* d_seeds, int stage_filter, bool reseed = true,
                       bool keep_generator = false, bool accumulate_time = false) {
  hipStream_t st = ctx->stream;
  uint32_t n_max = 1;
  for (uint32_t c : counts) n_max = std::max(n_max, c);
  int rc = ensure_nt_tables(ctx, o, counts, n_max);
  if (rc == DSM_OK) rc = ensure_nt_tables_t(ctx, o, {});
  if (rc != DSM_OK) return rc;
  HIPCHK(ctx, ctx->d_wm_redo.reserve(std::max<uint32_t>(n_pairs, 1) * 4));
  HIPCHK(ctx, ctx->d_wm_total.reserve(std::max<uint32_t>(n_pairs, 1) * 4));
  HIPCHK(ctx, ctx->d_wm_count.reserve(4));
  HIPCHK(ctx, ctx->d_pose_jobs.reserve(std::max<uint32_t>(n_pairs, 1) * sizeof(PoseJob)));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_wm_count.p, 0, 4, st));
  HIPCHK(ctx, ctx->d_tvg.reserve(std::max<uint32_t>(n_pairs, 1) * sizeof(dsm_two_view_geometry)));
  HIPCHK(ctx, ctx->d_inl.reserve(std::max<uint64_t>(total_matches, 1) * 8));
  HIPCHK(ctx, ctx->d_inl_counts.reserve(std::max<uint32_t>(n_pairs, 1) * 4));
  HIPCHK(ctx, ctx->d_inl_off.reserve(((size_t)n_pairs + 1) * 8));
  HIPCHK(ctx, ctx->d_inl_total.reserve(8));
  int dev_cus = 256;
  (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const uint32_t n_blocks = std::min<uint32_t>(n_pairs, (uint32_t)dev_cus * 16u);
  (void)n_blocks;  // the grid of the legacy schedule (check build)
  const uint64_t tm = std::max<uint64_t>(total_matches, 1);
  HIPCHK(ctx, ctx->d_pair_state.reserve(std::max<size_t>(n_pairs, 1) * 1280 * 4));
  HIPCHK(ctx, ctx->d_pts_px.reserve(tm * 32));
  HIPCHK(ctx, ctx->d_pts_norm.reserve(tm * 32));
  HIPCHK(ctx, ctx->d_reports.reserve(std::max<size_t>(n_pairs, 1) * 3 * sizeof(RansacReport)));
  HIPCHK(ctx, ctx->d_masks.reserve(tm * 3));
  VerifyParams vp;
  vp.pair_state = ctx->d_pair_state.as<uint32_t>();
  vp.pts_px = ctx->d_pts_px.as<double>();
  vp.pts_norm = ctx->d_pts_norm.as<double>();
  vp.reports = ctx->d_reports.as<RansacReport>();
  vp.masks = ctx->d_masks.as<unsigned char>();
  vp.mask_stride = tm;
  vp.pairs = d_pairs;
  vp.match_off = d_match_off;
  vp.matches = d_matches;
  vp.kp = d_kp;
  vp.img_row0 = d_img_row0;
  vp.cams = d_cams;
  vp.opt = *o;
  vp.seeds = d_seeds;
  vp.nt_table = ctx->d_nt_table.as<uint32_t>();
  vp.nt_off = ctx->d_nt_off.as<uint64_t>();
  vp.nt_off_t = ctx->d_nt_off_t.as<uint64_t>();
  vp.nt_table_t = ctx->d_nt_table_t.as<uint32_t>();
  vp.wm_redo = ctx->d_wm_redo.as<uint32_t>();
  vp.wm_total = ctx->d_wm_total.as<uint32_t>();
  vp.wm_count = ctx->d_wm_count.as<uint32_t>();
  vp.final_list = nullptr;
  vp.n_final = 0;
  vp.pose_jobs = ctx->d_pose_jobs.as<PoseJob>();
  vp.max_trials[0] = ransac_ctor_max_trials(o, o->min_inlier_ratio, kMinSamples[0]);
  vp.max_trials[1] = ransac_ctor_max_trials(o, o->min_inlier_ratio, kMinSamples[1]);
  vp.max_trials[2] = ransac_ctor_max_trials(o, o->min_inlier_ratio, kMinSamples[2]);
  vp.max_trials[3] = ransac_ctor_max_trials(o, o->watermark_min_inlier_ratio, kMinSamples[3]);  // two_view_geometry.cc:541-542
  vp.tvg = ctx->d_tvg.as<dsm_two_view_geometry>();
  vp.inlier_matches = ctx->d_inl.as<uint32_t>();
  vp.inl_counts = ctx->d_inl_counts.as<uint32_t>();
  vp.scratch = nullptr;
  vp.n_pairs = n_pairs;
  vp.n_max = n_max;
  vp.stage_filter = stage_filter;
  vp.sampler_serial = ctx->dbg("DSM_SAMPLER_SERIAL") ? 1 : 0;
  vp.reseed = reseed ? 1 : 0;
  vp.keep_generator = keep_generator ? 1 : 0;
  if (!ctx->vev0) {
    HIPCHK(ctx, hipEventCreate(&ctx->vev0));
    HIPCHK(ctx, hipEventCreate(&ctx->vev1));
  }
  vp.pair0 = 0;
  vp.n_chunk = n_pairs;
  vp.batch = 0;
  vp.fam_state = nullptr;
  vp.samples = nullptr;
  vp.draws_end = nullptr;
  vp.nmodels = nullptr;
  vp.counts = nullptr;
  vp.sums = nullptr;
  vp.first_batch[0] = vp.first_batch[1] = vp.first_batch[2] = 0;
  vp.stats = ctx->dbg("DSM_VERIFY_DEBUG") ? 1 : 0;
  vp.lo_reg_prepare = ctx->dbg("DSM_LO_PREPARE_WAVE") ? 0 : 1;
  vp.rp_cap = std::min<uint32_t>(n_max, 256u);  // RP_CAP (verify_kernels.hip)
  vp.spec_margin[0] = 8;
  vp.spec_margin[1] = vp.spec_margin[2] = 4;
  if (const char* e = ctx->dbg("DSM_SPEC_MARGIN")) {  // check build: "e,f"
    unsigned a = 8, b = 4;
    if (sscanf(e, "%u,%u", &a, &b) == 2) {
      vp.spec_margin[0] = a;
      vp.spec_margin[1] = b;
    }
  }
  vp.hyp_map = nullptr;  // (set per round by verify_lane_run)
  vp.hyp_seg_cap = 0;
  vp.replay_legacy = ctx->dbg("DSM_REPLAY_LEGACY") ? 1 : 0;
  vp.dbg_elu_lds = ctx->dbg("DSM_ELU_LDS") ? 1 : 0;
  vp.dbg_jacobi_groups = ctx->dbg("DSM_LO_JACOBI_GROUPS") ? 1 : 0;  // the 8-lane-group Jacobi kernel for every problem (round-2 form)
  vp.dbg_roots_lds = ctx->dbg("DSM_ROOTS_LDS") ? 1 : 0;              // k_roots_e_lds instead of the register form
  vp.dbg_final_waves = ctx->dbg("DSM_FINAL_WAVES") ? atoi(ctx->dbg("DSM_FINAL_WAVES")) : 0;
  vp.score_prefilter = ctx->dbg("DSM_SCORE_PREFILTER") ? atoi(ctx->dbg("DSM_SCORE_PREFILTER")) : 1;
  // "check": every slot is scored exactly AND held against its bounds (counters [14] violations, [15] slots the filter
  // would have skipped; dsm_debug_verify_counters); the statistics counters stay alive across the rounds
  const bool prefilter_check = ctx->dbg("DSM_SCORE_PREFILTER") && strcmp(ctx->dbg("DSM_SCORE_PREFILTER"), "check") == 0;
  if (prefilter_check) {
    vp.score_prefilter = 5;
    vp.stats = 1;
  }  // =1: the round-2 kernel (matrix in global scratch) for every problem
  vp.models = nullptr;
  vp.e_work = nullptr;
  vp.sidx_g = nullptr;
  vp.active_count = nullptr;
  vp.lo_inl = nullptr;
  vp.worklist = nullptr;
  vp.n_work = 0;
  vp.lo_queue_g = nullptr;
  vp.lo_queue = nullptr;
  vp.lo_count = nullptr;
  vp.lo_work = vp.lo_models = vp.lo_slots = vp.lo_ework = nullptr;
  vp.tail_items = nullptr;
  vp.tail_n = nullptr;
  vp.lo_jobs = nullptr;
  vp.lo_inl_pool = nullptr;
  vp.job_list = nullptr;
#ifdef DSM_CHECK_BUILD
  const bool legacy = ctx->dbg("DSM_VERIFY_LEGACY") != nullptr;  // single-kernel-per-family schedule (check build only)
#else
  const bool legacy = false;
#endif
  if (legacy) {
#ifdef DSM_CHECK_BUILD
    HIPCHK(ctx, ctx->d_vscratch.reserve(std::max<size_t>(1, (size_t)n_blocks * verify_scratch_bytes_per_block(n_max))));
    vp.scratch = ctx->d_vscratch.as<double>();
    HIPCHK(ctx, ctx->lanes[0].active.reserve(LANE_CTR_BYTES));
    HIPCHK(ctx, hipMemsetAsync(ctx->lanes[0].active.p, 0, LANE_CTR_BYTES, st));
    vp.active_count = ctx->lanes[0].active.as<uint32_t>() + LANE_CTR_AREA / 4;
    HIPCHK(ctx, hipEventRecord(ctx->vev0, st));
    launch_verify(vp, n_blocks, st);
    HIPCHK(ctx, hipGetLastError());
#endif
  } else {
    // phase-split pipeline: per family, rounds of sample -> solve+score -> replay until no pair is active
    VerifyPlan plan;
    for (int f = 0; f < 3; ++f) {
      plan.batch[f] = vp_batch(f, vp.max_trials[f], (uint32_t)std::min<uint64_t>(o->min_num_trials, 0xffffffffull));
      vp.first_batch[f] = plan.batch[f];
      // (measured, round 3: smaller first rounds cost more in extra rounds than they save in unused hypotheses -- E 56 / 48 / 32: +10 / +19 / +24 ms, F 96: +5 ms)
    }
    auto per_pair_bytes = [&](const uint32_t* b, uint32_t* bmax_out, uint64_t* bm_out) {
      uint32_t bmax = 0;
      uint64_t bm = 0;
      for (int f = 0; f < 3; ++f) {
        bmax = std::max(bmax, b[f]);
        bm = std::max<uint64_t>(bm, (uint64_t)b[f] * vp_maxm(f));
      }
      if (bmax_out) *bmax_out = bmax;
      if (bm_out) *bm_out = bm;
      return (uint64_t)bmax * (7 * 4 + 4 + 4) + bm * (4 + 8 + 72) + (uint64_t)b[0] * 200 * 8 + (LO_WORK_DOUBLES + 90 + 90 + 200) * 8 + 12 + (uint64_t)std::max(b[0], b[1]) * 4;
    };
    // Lanes: the pair list is dealt out in chunks to up to DSM_VERIFY_MAX_LANES lanes that run concurrently (own
    // stream, own host thread, own scratch; see VerifyLane).  A pair's three families cannot overlap -- F starts from
    // the generator state E ends with -- but different pairs can, and the replay / local-optimisation launches of
    // one lane are latency chains that leave most of the chip idle.
    // (two lanes from 1 024 pairs on: config 1's 1 225 pairs 7.97 vs 8.27 ms per step, a 15 594-pair shard 78.4 vs 85.3; three or four lanes
    // lose on both, profiles/r05_lanes_short_lists.txt)
    uint32_t n_lanes = n_pairs >= 1024 ? 2 : 1;
    if (const char* e = ctx->dbg("DSM_VERIFY_LANES")) n_lanes = (uint32_t)std::max(1, std::min(DSM_VERIFY_MAX_LANES, atoi(e)));
    n_lanes = std::max<uint32_t>(1, std::min<uint32_t>(n_lanes, n_pairs));
    // Scratch of the speculated trials: as much of the pair list per chunk as memory allows (every chunk pays the
    // latency tail of its sequential rounds, so fewer chunks are faster): up to 40 % of what is free now, at
    // least 4 GiB, at most 96 GiB (config 2 needs 38 GiB for the whole list; an MI355X has 288 GB).
    uint64_t budget = 16ull << 30;
    {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        uint64_t have = 0;  // buffers of an earlier call are reused, not allocated again
        for (VerifyLane& L : ctx->lanes)
          for (DevBuf* b : {&L.samples, &L.draws_end, &L.nmodels, &L.vcounts, &L.vsums, &L.models, &L.ework, &L.lo_work, &L.lo_models, &L.lo_slots,
                            &L.lo_ework})
            have += b->cap;
        budget = std::min<uint64_t>(96ull << 30, std::max<uint64_t>(4ull << 30, (uint64_t)((free_b + have) * 0.4)));
      }
    }
    if (ctx->memory_budget) budget = ctx->memory_budget - match_budget_share(ctx->memory_budget);  // dsm_ctx_set_memory_budget: the host application's word
    // Round buffers: a pair's FIRST round speculates first_batch trials (sized for the easy regime: E / F stop after
    // ~60 / ~150 trials at a 64 % inlier ratio), every later round what its dynamic stop still asks for.  At a 25 %
    // ratio that is thousands of trials (E 5 400, F 10 000: measured), i.e. a hundred rounds of 64 -- so the buffers
    // grow (E up to 512, F up to 1 024 trials per round) as long as the whole list still fits a quarter of the budget.
    if (!ctx->dbg("DSM_VERIFY_FIXED_BATCH")) {
      for (bool grew = true; grew;) {
        grew = false;
        for (int f = 1; f >= 0; --f) {
          const uint32_t cap = f == 0 ? 512u : 1024u;
          if (plan.batch[f] * 2 > cap) continue;
          uint32_t trial[3] = {plan.batch[0], plan.batch[1], plan.batch[2]};
          trial[f] *= 2;
          if (per_pair_bytes(trial, nullptr, nullptr) * (uint64_t)n_pairs <= budget / 4) {
            plan.batch[f] = trial[f];
            grew = true;
          }
        }
      }
    }
    uint64_t bm_max = 0;
    const uint64_t per_pair = per_pair_bytes(plan.batch, &plan.bmax, &bm_max);
    plan.bm_max = bm_max;
    // Equal shares (measured: giving the first lane 0.6 - 0.8 of the list to push the lanes out of phase is 1 - 2 %
    // slower than 0.5; three lanes are slower than two).  DSM_VERIFY_LANE_SPLIT = share of the first lane.
    double first_share = 1.0 / n_lanes;
    if (const char* e = ctx->dbg("DSM_VERIFY_LANE_SPLIT")) first_share = std::min(0.95, std::max(0.05, atof(e)));
    const char* cp = ctx->dbg("DSM_VERIFY_CHUNK_PAIRS");  // test hook: force several chunks on a small pair list
    uint32_t at = 0;
    for (uint32_t li = 0; li < n_lanes; ++li) {
      uint32_t cnt = (li == 0 && n_lanes > 1) ? (uint32_t)(n_pairs * first_share + 0.5) : (n_pairs - at) / (n_lanes - li);
      cnt = std::min<uint32_t>(std::max<uint32_t>(cnt, 1), n_pairs - at - (n_lanes - 1 - li));
      plan.begin[li] = at;
      plan.end[li] = at + cnt;
      at += cnt;
      const uint64_t lane_budget = (uint64_t)(budget * ((double)cnt / n_pairs));
      plan.chunk[li] = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(cnt, (1u << 21) - 64u), lane_budget / per_pair));  // (< 2^21 pairs: an entry of the hypothesis list holds pl in 21 bits)
      if (cp) plan.chunk[li] = std::max<uint32_t>(1, std::min<uint32_t>(plan.chunk[li], (uint32_t)atoi(cp)));
    }
    plan.n_lanes = n_lanes;
    plan.dev_cus = dev_cus;
    // persistent grids sized for a lane's share of the chip: two lanes 362.5 vs 366.3 ms at config 2, 193.5 vs 205.1 on its half
    // -- when the lanes' launches are long.  A short list's launches are tails more than bodies, and a lane then gains more from
    // waves of its own on every CU than it loses to the other lane's: 15 594 pairs (an eighth of config 2) 46.7 vs 48.3 ms with
    // whole-chip grids, the whole list 306.0 vs 297.8.
    plan.grid_div = n_pairs <= 40000u ? 1u : n_lanes;
    if (const char* e = ctx->dbg("DSM_VERIFY_GRID_DIV")) plan.grid_div = (uint32_t)std::max(1, atoi(e));
    if (const char* e = ctx->dbg("DSM_VERIFY_REPLAY_GRID")) plan.replay_grid_mul = (uint32_t)std::max(1, atoi(e));
    // Local optimisation: batched kernels (k_replay_lo + k_lo_*) or inline in the replay (k_replay).  The batched form
    // wins on throughput (config 2, 124 750 pairs: 416 vs 702 ms) but every LO iteration costs a kernel round trip; it
    // hands the last <= lo_tail queued pairs of a round (F, H) to an inline finish, which keeps it ahead or level down
    // to ~1 000 pairs (7 140 pairs: 42 vs 50 ms; 4 950: 34.5 vs 36.4; 1 225: 9.5 vs 9.3).  The inline form remains for
    // lists shorter than lo_tail, where the batched one would only add a launch.
    // DSM_VERIFY_INLINE_LO=1 / =0 forces one or the other (tests cover both).
    plan.inline_lo = n_pairs < 256u;  // (round 2: below 2 048; with item passes the batched kernels are ahead from ~1 000 pairs on: 1 225 pairs 14.7 vs 19.8 ms)
    if (const char* e = ctx->dbg("DSM_VERIFY_INLINE_LO")) plan.inline_lo = atoi(e) != 0;
    // queue length at which the chain hands the rest of a round to an item pass.  Measured (round 3, item-pass tail): 2 048 /
    // 4 096 / 8 192 / 16 384 / 32 768 -> 370.6 / 373.2 / 366.3 / 373.1 / 375.9 ms at config 2, 61.7 / 61.0 / 57.5 / 57.8 / 57.7 ms on its 1/8 shard
    plan.lo_tail = 8192;
    if (const char* e = ctx->dbg("DSM_LO_TAIL")) plan.lo_tail = (uint32_t)std::max(0, atoi(e));
    if (const char* e = ctx->dbg("DSM_LO_TAIL_MODE")) plan.tail_items = strcmp(e, "inline") != 0;
    // item passes from the start of every round for short lists: 4 950 pairs 24.3 vs 28.5 ms, 15 593 pairs (1/8 of config 2)
    // 55.8 vs 57.5, 31 187 pairs level, 124 750 pairs 395 vs 366 (the speculative half of the items is throughput there)
    plan.item_mode = n_pairs <= 24000u;
    if (const char* e = ctx->dbg("DSM_VERIFY_ITEM_MODE")) {  // 0: never, 1: every round, 1 + k: the first k rounds of a family only
      plan.item_mode = atoi(e) != 0;
      if (atoi(e) > 1) plan.item_rounds = (uint32_t)atoi(e) - 1u;
    }
    if (const char* e = ctx->dbg("DSM_HYP_GRID")) plan.hyp_compact = strcmp(e, "pair") != 0;  // check build: "pair" = the (pair, 64 trials) grid in every round
    HIPCHK(ctx, ctx->d_fam_state.reserve(std::max<size_t>(n_pairs, 1) * 3 * sizeof(FamState)));
    HIPCHK(ctx, ctx->d_sidx.reserve(tm * 4));
    HIPCHK(ctx, ctx->d_lo_inl.reserve(tm * 4));
    HIPCHK(ctx, ctx->d_lo_inl_pool.reserve(tm * TAIL_KMAX * 4));  // inlier lists of the item passes' jobs
    vp.lo_inl = ctx->d_lo_inl.as<uint32_t>();
    vp.fam_state = ctx->d_fam_state.as<FamState>();
    vp.sidx_g = ctx->d_sidx.as<uint32_t>();
    // The lane scratch was sized from what hipMemGetInfo reported a moment ago; other contexts of this process (bench
    // --contexts, several host threads) may have taken that memory since.  An allocation failure here is not an
    // error: release the lanes' scratch, halve the chunks, try again (a chunk of one pair always fits or nothing does).
    // what a lane holds for a chunk of `chunk` pairs, buffer by buffer (reserve_lanes below allocates exactly these)
    struct LaneBytes {
      size_t v[19];
      size_t total() const {
        size_t t = 0;
        for (size_t x : v) t += x + x / 8 + 256;  // (DevBuf::reserve rounds up by an eighth)
        return t;
      }
    };
    // the list of a round's hypotheses for the compact solver grids: 64 segments of ceil(chunk / 64) pairs x the larger of the E / F round sizes
    auto hyp_seg_cap = [&](uint32_t chunk, uint32_t batch) { return (size_t)((chunk + 63u) / 64u) * batch; };
    auto hyp_entries = [&](uint32_t chunk) { return 64 * hyp_seg_cap(chunk, std::max(plan.batch[0], plan.batch[1])); };
    auto lane_bytes = [&](uint32_t chunk) {
      LaneBytes z;
      const uint32_t lane_blocks = std::min<uint32_t>(chunk, (uint32_t)dev_cus * 4u * DSM_REPLAY_WAVES);
      // item passes: over the whole chunk (item_mode) or over a queue of <= lo_tail pairs
      const size_t item_pairs = plan.item_mode ? chunk : std::min<uint32_t>(chunk, std::max<uint32_t>(plan.lo_tail, 1));
      const size_t slots = std::max<size_t>(chunk, item_pairs * TAIL_KMAX);  // records of the batched LO kernels: per pair or per job
      z.v[0] = LANE_CTR_BYTES;
      z.v[1] = std::max<size_t>(1, (size_t)lane_blocks * verify_scratch_bytes_per_block(n_max));
      z.v[2] = (size_t)chunk * plan.bmax * 7 * 4;
      z.v[3] = (size_t)chunk * plan.bmax * 4;
      z.v[4] = (size_t)chunk * plan.bmax * 4;
      z.v[5] = (size_t)chunk * bm_max * 4;
      z.v[6] = (size_t)chunk * bm_max * 8;
      z.v[7] = (size_t)chunk * bm_max * 72;
      z.v[8] = (size_t)chunk * plan.batch[0] * 200 * 8;
      z.v[9] = ((size_t)chunk * 2 + std::max<size_t>(chunk, item_pairs * TAIL_KMAX)) * 4;  // two alternating queues + the general-kernel list (pairs or job slots)
      z.v[10] = item_pairs * TAIL_KMAX * sizeof(TailItem);
      z.v[11] = item_pairs * 4;
      z.v[12] = item_pairs * TAIL_KMAX * sizeof(LoJob);
      z.v[13] = item_pairs * TAIL_KMAX * 4;
      z.v[14] = slots * LO_WORK_DOUBLES * 8;
      z.v[15] = slots * 90 * 8;
      z.v[16] = slots * 90 * 8;
      z.v[17] = slots * 200 * 8;
      z.v[18] = hyp_entries(chunk) * 4;
      return z;
    };
    if (ctx->memory_budget) {
      // the host application's budget is a promise about what is ALLOCATED: shrink the chunks until the lanes' buffers -- fixed
      // parts, item-pass records and DevBuf's rounding included -- fit it (a chunk of one pair always runs)
      for (;;) {
        size_t total = 0;
        uint32_t largest = 0;
        for (uint32_t li = 0; li < n_lanes; ++li) {
          total += lane_bytes(plan.chunk[li]).total();
          largest = std::max(largest, plan.chunk[li]);
        }
        if (total <= ctx->memory_budget - match_budget_share(ctx->memory_budget) || largest <= 1) break;
        for (uint32_t li = 0; li < n_lanes; ++li) plan.chunk[li] = std::max<uint32_t>(1, (uint32_t)(plan.chunk[li] * 0.9));
      }
    }
    for (;;) {
      auto reserve_lanes = [&]() -> hipError_t {
#define LRES(call)                          \
  do {                                      \
    const hipError_t e_ = (call);           \
    if (e_ != hipSuccess) return e_;        \
  } while (0)
        for (uint32_t li = 0; li < n_lanes; ++li) {
          VerifyLane& L = ctx->lanes[li];
          const uint32_t chunk = plan.chunk[li];
          // Lane 0 runs on the context's own stream, the others get one each.  The runtime maps a process's streams onto a small
          // pool of hardware queues (ROCclr: GPU_MAX_HW_QUEUES = 4 by default) and streams that share a queue serialise: with
          // the null stream, the context's stream and one stream per lane, THREE lanes made five streams -- measured as "three
          // lanes are slower than one" until the queue pool was raised (profiles/r05_lanes_hw_queues.txt).  One stream fewer
          // keeps three lanes inside the default pool.
          if (!L.stream) {
            if (li == 0)
              L.stream = ctx->stream;
            else
              LRES(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
          }
          if (!L.done) LRES(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
          if (!L.host_ctr) LRES(hipHostMalloc(reinterpret_cast<void**>(&L.host_ctr), 128, hipHostMallocDefault));
          const LaneBytes z = lane_bytes(chunk);
          LRES(L.active.reserve(z.v[0]));
          LRES(L.vscratch.reserve(z.v[1]));
          LRES(L.samples.reserve(z.v[2]));
          LRES(L.draws_end.reserve(z.v[3]));
          LRES(L.nmodels.reserve(z.v[4]));
          LRES(L.vcounts.reserve(z.v[5]));
          LRES(L.vsums.reserve(z.v[6]));
          LRES(L.models.reserve(z.v[7]));
          LRES(L.ework.reserve(z.v[8]));
          LRES(L.lo_queue.reserve(z.v[9]));
          LRES(L.tail_items.reserve(z.v[10]));
          LRES(L.tail_n.reserve(z.v[11]));
          LRES(L.lo_jobs.reserve(z.v[12]));
          LRES(L.job_list.reserve(z.v[13]));
          LRES(L.lo_work.reserve(z.v[14]));
          LRES(L.lo_models.reserve(z.v[15]));
          LRES(L.lo_slots.reserve(z.v[16]));
          LRES(L.lo_ework.reserve(z.v[17]));
          LRES(L.hyp_map.reserve(z.v[18]));
        }
#undef LRES
        return hipSuccess;
      };
      const hipError_t e = reserve_lanes();
      if (e == hipSuccess) break;
      (void)hipGetLastError();
      uint32_t largest = 0;
      for (uint32_t li = 0; li < n_lanes; ++li) largest = std::max(largest, plan.chunk[li]);
      if (e != hipErrorOutOfMemory || largest <= 1) {
        ctx->err = std::string("verification scratch: ") + hipGetErrorString(e);
        return DSM_ERR_HIP;
      }
      for (VerifyLane& L : ctx->lanes)
        for (DevBuf* b : {&L.samples, &L.draws_end, &L.nmodels, &L.vcounts, &L.vsums, &L.models, &L.ework, &L.vscratch, &L.lo_queue, &L.lo_work,
                          &L.lo_models, &L.lo_slots, &L.lo_ework, &L.tail_items, &L.tail_n, &L.lo_jobs, &L.job_list, &L.hyp_map})
          b->release();
      for (uint32_t li = 0; li < n_lanes; ++li) plan.chunk[li] = std::max<uint32_t>(1, plan.chunk[li] / 2);
    }
    for (uint32_t li = 0; li < n_lanes; ++li) {
      VerifyLane& L = ctx->lanes[li];
      L.rc = DSM_OK;
      L.err.clear();
      for (int f = 0; f < 3; ++f) L.rounds[f] = L.lo_iters[f] = 0;
      memset(L.dbg, 0, sizeof(L.dbg));
    }
    ctx->verify_lanes = n_lanes;
    HIPCHK(ctx, hipEventRecord(ctx->vev0, st));
    if (n_lanes == 1) {
      verify_lane_run(ctx, 0, vp, plan);
    } else {
      // lane 0 on the calling thread, the others on their own; a lane whose thread cannot be created runs here afterwards
      std::vector<std::thread> th;
      std::vector<uint32_t> here;
      for (uint32_t li = 1; li < n_lanes; ++li) {
        try {
          th.emplace_back(verify_lane_run, ctx, li, vp, plan);
        } catch (const std::system_error&) {
          here.push_back(li);
        }
      }
      verify_lane_run(ctx, 0, vp, plan);
      for (uint32_t li : here) verify_lane_run(ctx, li, vp, plan);
      for (std::thread& t : th) t.join();
    }
    for (uint32_t li = 0; li < n_lanes; ++li) {
      VerifyLane& L = ctx->lanes[li];
      if (L.rc != DSM_OK) {
        ctx->err = L.err;
        (void)hipDeviceSynchronize();
        return L.rc;
      }
      HIPCHK(ctx, hipStreamWaitEvent(st, L.done, 0));
    }
    for (int f = 0; f < 3; ++f) {
      ctx->verify_rounds[f] = ctx->verify_lo_iters[f] = 0;
      for (uint32_t li = 0; li < n_lanes; ++li) {
        ctx->verify_rounds[f] = std::max(ctx->verify_rounds[f], ctx->lanes[li].rounds[f]);
        ctx->verify_lo_iters[f] = std::max(ctx->verify_lo_iters[f], ctx->lanes[li].lo_iters[f]);
      }
    }
    if (ctx->dbg("DSM_VERIFY_DEBUG")) {
      uint32_t dbg[32] = {0};
      unsigned long long cyc[3] = {0, 0, 0};
      for (uint32_t li = 0; li < n_lanes; ++li) {
        for (int k = 0; k < 8; ++k) dbg[k] += ctx->lanes[li].dbg[k];
        const unsigned long long* c = reinterpret_cast<const unsigned long long*>(ctx->lanes[li].dbg + 8);
        for (int k = 0; k < 3; ++k) cyc[k] += c[k];
      }
      fprintf(stderr, "[dsm verify] replay cycles (all families): candidates %llu  local-opt %llu  whole-pair loop %llu\n", cyc[0], cyc[1], cyc[2]);
#ifdef DSM_PROFILE_SECTIONS
      {
        unsigned long long prof[16];
        debug_read_prof(prof);
        fprintf(stderr, "[dsm verify] LO sections (cycles): qr %llu  jacobi %llu  finish8pt %llu  finish5pt %llu\n", prof[4], prof[5], prof[6], prof[7]);
        fprintf(stderr, "[dsm verify] 5pt per-lane sections (Mcycles): build %llu  lu %llu  det %llu  roots %llu  models %llu | score E %llu  score F/H %llu\n",
                prof[8] >> 20, prof[9] >> 20, prof[10] >> 20, prof[11] >> 20, prof[12] >> 20, prof[13] >> 20, prof[14] >> 20);
      }
#endif
      fprintf(stderr, "[dsm verify] lanes %u (chunk %u pairs); per lane at most: local-optimisation iterations E/F/H %u/%u/%u\n", n_lanes, plan.chunk[0],
              ctx->verify_lo_iters[0], ctx->verify_lo_iters[1], ctx->verify_lo_iters[2]);
      fprintf(stderr, "[dsm verify] pairs %u rounds E/F/H %u/%u/%u candidates E/F/H %u/%u/%u LO calls E/F/H %u/%u/%u\n", n_pairs,
              ctx->verify_rounds[0], ctx->verify_rounds[1], ctx->verify_rounds[2], dbg[1], dbg[3], dbg[5], dbg[2], dbg[4], dbg[6]);
    }
  }
  // Pairs whose watermark test waited for a translation table (k_verify_final): build the tables of their inlier
  // counts, visit exactly those pairs again.  Normally none.
  {
    uint32_t n_redo = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n_redo, ctx->d_wm_count.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (n_redo) {
      std::vector<uint32_t> totals(n_redo);
      HIPCHK(ctx, hipMemcpy(totals.data(), ctx->d_wm_total.p, (size_t)n_redo * 4, hipMemcpyDeviceToHost));
      rc = ensure_nt_tables_t(ctx, o, totals);
      if (rc != DSM_OK) return rc;
      vp.nt_table_t = ctx->d_nt_table_t.as<uint32_t>();
      vp.nt_off_t = ctx->d_nt_off_t.as<uint64_t>();
      vp.final_list = ctx->d_wm_redo.as<uint32_t>();
      vp.n_final = n_redo;
      vp.pair0 = 0;
      vp.n_chunk = n_pairs;
      vp.active_count = nullptr;
      // the lanes are done: the per-block work area of lane 0 (legacy schedule: the context's) is free
      DevBuf& sb = legacy ? ctx->d_vscratch : ctx->lanes[0].vscratch;
      const size_t per_block = std::max<size_t>(1, verify_scratch_bytes_per_block(n_max));
      const uint32_t nb = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_redo, sb.cap / per_block));
      vp.scratch = sb.as<double>();
      HIPCHK(ctx, hipMemsetAsync(ctx->d_wm_count.p, 0, 4, st));
      launch_vp_final(vp, nb, st);
      HIPCHK(ctx, hipGetLastError());
    }
  }
  HIPCHK(ctx, hipEventRecord(ctx->vev1, st));
  // compact inlier matches in list order
  HIPCHK(ctx, hipMemsetAsync(ctx->d_inl_total.p, 0, 8, st));
  launch_scan(ctx->d_inl_counts.as<uint32_t>(), ctx->d_inl_off.as<uint64_t>(), n_pairs, ctx->d_inl_total.as<uint64_t>(), st);
  HIPCHK(ctx, hipGetLastError());
  uint64_t total = 0;
  HIPCHK(ctx, hipMemcpyAsync(&total, ctx->d_inl_total.p, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->total_inliers = total;
  HIPCHK(ctx, ctx->d_inl_compact.reserve(std::max<uint64_t>(total, 1) * 8));
  launch_compact_inliers(d_match_off, ctx->d_inl_off.as<uint64_t>(), ctx->d_inl_counts.as<uint32_t>(), ctx->d_inl.as<uint32_t>(),
                         ctx->d_inl_compact.as<uint32_t>(), n_pairs, st);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(st));
  float ms = 0.f;
  HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->vev0, ctx->vev1));
  ctx->verify_ms = accumulate_time ? ctx->verify_ms + ms : ms;
  return DSM_OK;
}

static int verify_core_plain(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* d_pairs, const uint64_t* d_match_off,
                             const uint32_t* d_matches, uint64_t total_matches, const std::vector<uint32_t>& counts,
                             const double* d_kp, const uint32_t* d_img_row0, const dsm_camera* d_cams,
                             const dsm_two_view_options* o, const uint32_t* d_seeds, int stage_filter) {
  return verify_core(ctx, n_pairs, d_pairs, d_match_off, d_matches, total_matches, counts, d_kp, d_img_row0, d_cams, o, d_seeds,
                     stage_filter);
}

// TwoViewGeometry::EstimateMultiple (/root/reference/src/estimators/two_view_geometry.cc:128-167) for all pairs:
// repeated passes of verify_core over the matches that are not inliers of the geometries found so far (a
// finished pair carries zero matches and costs nothing), the pair's generator stream continuing from pass to
// pass.  Results land where verify_core leaves them (d_tvg, d_inl_off, d_inl_compact).
static int verify_multiple(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* d_pairs, const uint64_t* d_match_off,
                           const uint32_t* d_matches, uint64_t total_matches, const std::vector<uint32_t>& counts,
                           const double* d_kp, const uint32_t* d_img_row0, const dsm_camera* d_cams,
                           const dsm_two_view_options* o, const uint32_t* d_seeds, int stage_filter) {
  hipStream_t st = ctx->stream;
  const uint64_t tm = std::max<uint64_t>(total_matches, 1);
  const size_t np1 = std::max<uint32_t>(n_pairs, 1);
  HIPCHK(ctx, ctx->d_mm_matches[0].reserve(tm * 8));
  HIPCHK(ctx, ctx->d_mm_matches[1].reserve(tm * 8));
  HIPCHK(ctx, ctx->d_mm_off[0].reserve((np1 + 1) * 8));
  HIPCHK(ctx, ctx->d_mm_off[1].reserve((np1 + 1) * 8));
  HIPCHK(ctx, ctx->d_mm_counts.reserve(np1 * 4));
  HIPCHK(ctx, ctx->d_mm_state.reserve(np1 * sizeof(MultiState)));
  HIPCHK(ctx, ctx->d_mm_first.reserve(np1 * sizeof(dsm_two_view_geometry)));
  HIPCHK(ctx, ctx->d_mm_acc.reserve(tm * 8));
  HIPCHK(ctx, ctx->d_mm_keep.reserve(tm));
  HIPCHK(ctx, ctx->d_mm_total.reserve(16));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_mm_state.p, 0, np1 * sizeof(MultiState), st));
  dsm_two_view_options pass_opt = *o;
  pass_opt.multiple_models = 0;
  const uint64_t* cur_off = d_match_off;
  const uint32_t* cur_matches = d_matches;
  std::vector<uint32_t> cur_counts = counts;
  uint64_t cur_total = total_matches;
  MultiParams mp;
  mp.orig_off = d_match_off;
  mp.state = ctx->d_mm_state.as<MultiState>();
  mp.first = ctx->d_mm_first.as<dsm_two_view_geometry>();
  mp.acc = ctx->d_mm_acc.as<uint32_t>();
  mp.keep = ctx->d_mm_keep.as<unsigned char>();
  mp.next_count = ctx->d_mm_counts.as<uint32_t>();
  mp.active = ctx->d_mm_total.as<uint32_t>() + 2;
  mp.ignore_watermark = o->multiple_ignore_watermark ? 1 : 0;
  mp.stage_filter = stage_filter;
  mp.min_num_inliers = o->min_num_inliers;
  mp.n_pairs = n_pairs;
  double ms_total = 0.0;
  for (uint32_t pass = 0;; ++pass) {
    int rc = verify_core(ctx, n_pairs, d_pairs, cur_off, cur_matches, cur_total, cur_counts, d_kp, d_img_row0, d_cams,
                         &pass_opt, d_seeds, /*stage_filter=*/0, /*reseed=*/pass == 0, /*keep_generator=*/true);
    if (rc != DSM_OK) return rc;
    ms_total += ctx->verify_ms;
    const int nxt = (int)(pass & 1u);
    mp.cur_off = cur_off;
    mp.cur_matches = cur_matches;
    mp.tvg = ctx->d_tvg.as<dsm_two_view_geometry>();
    mp.inl = ctx->d_inl.as<uint32_t>();
    mp.inl_counts = ctx->d_inl_counts.as<uint32_t>();
    mp.next_off = ctx->d_mm_off[nxt].as<uint64_t>();
    mp.next_matches = ctx->d_mm_matches[nxt].as<uint32_t>();
    HIPCHK(ctx, hipMemsetAsync(ctx->d_mm_total.p, 0, 16, st));
    launch_multi_accumulate(mp, st);
    HIPCHK(ctx, hipGetLastError());
    launch_scan(mp.next_count, ctx->d_mm_off[nxt].as<uint64_t>(), n_pairs, ctx->d_mm_total.as<uint64_t>(), st);
    launch_multi_scatter(mp, st);
    HIPCHK(ctx, hipGetLastError());
    uint32_t tot[4] = {0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(tot, ctx->d_mm_total.p, 16, hipMemcpyDeviceToHost, st));
    if (n_pairs) HIPCHK(ctx, hipMemcpyAsync(cur_counts.data(), ctx->d_mm_counts.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (tot[2] == 0) break;  // every pair has returned DEGENERATE
    cur_off = ctx->d_mm_off[nxt].as<uint64_t>();
    cur_matches = ctx->d_mm_matches[nxt].as<uint32_t>();
    cur_total = (uint64_t)tot[0] | ((uint64_t)tot[1] << 32);
  }
  // final records + inlier matches (accumulated at the original offsets) in list order
  mp.out_tvg = ctx->d_tvg.as<dsm_two_view_geometry>();
  mp.out_inl_counts = ctx->d_inl_counts.as<uint32_t>();
  launch_multi_finalize(mp, st);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemsetAsync(ctx->d_inl_total.p, 0, 8, st));
  launch_scan(ctx->d_inl_counts.as<uint32_t>(), ctx->d_inl_off.as<uint64_t>(), n_pairs, ctx->d_inl_total.as<uint64_t>(), st);
  HIPCHK(ctx, hipGetLastError());
  uint64_t total = 0;
  HIPCHK(ctx, hipMemcpyAsync(&total, ctx->d_inl_total.p, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->total_inliers = total;
  HIPCHK(ctx, ctx->d_inl_compact.reserve(std::max<uint64_t>(total, 1) * 8));
  launch_compact_inliers(d_match_off, ctx->d_inl_off.as<uint64_t>(), ctx->d_inl_counts.as<uint32_t>(), ctx->d_mm_acc.as<uint32_t>(),
                         ctx->d_inl_compact.as<uint32_t>(), n_pairs, st);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->verify_ms = ms_total;
  return DSM_OK;
}

uint32_t dsm_pair_seed(uint32_t image_id1, uint32_t image_id2, uint32_t user_seed) {
  // pair_id as Database::ImagePairToPairId (/root/reference/src/base/database.h:336-347), then a 32-bit mix
  const uint64_t kMaxNumImages = 2147483647ull;
  const uint64_t a = std::min(image_id1, image_id2), b = std::max(image_id1, image_id2);
  uint64_t h = kMaxNumImages * a + b;
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return (uint32_t)h ^ user_seed;
}

int dsm_verify_pairs(dsm_ctx* ctx, const dsm_two_view_options* options, const uint32_t* seeds, uint32_t user_seed,
                     int32_t stage_filter) {
  if (!ctx || !options) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->matched) return fail(ctx, DSM_ERR_NOT_READY, "dsm_match_pairs has not run");
  if (!ctx->have_kp || ctx->cameras.size() != ctx->n_images)
    return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "dsm_set_images was called without keypoints/cameras");
  // RANSACOptions::Check, ransac.h:63-71; TwoViewGeometry::Options::Check
  if (!(options->max_error > 0) || options->min_inlier_ratio < 0 || options->min_inlier_ratio > 1 ||
      options->confidence < 0 || options->confidence > 1 || options->min_num_trials > options->max_num_trials)
    return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "invalid RANSAC options");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->verified = false;
  const uint32_t np = ctx->n_pairs;
  std::vector<uint32_t> counts(np), sd(np);
  if (np) HIPCHK(ctx, hipMemcpy(counts.data(), ctx->d_counts.p, (size_t)np * 4, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < np; ++i)
    sd[i] = seeds ? seeds[i] : dsm_pair_seed(ctx->pairs[2 * i], ctx->pairs[2 * i + 1], user_seed);
  HIPCHK(ctx, ctx->d_cams.reserve(std::max<uint32_t>(ctx->n_images, 1) * sizeof(dsm_camera)));
  HIPCHK(ctx, ctx->d_pairs_dev.reserve(std::max<uint32_t>(np, 1) * 8));
  HIPCHK(ctx, ctx->d_seeds.reserve(std::max<uint32_t>(np, 1) * 4));
  if (ctx->n_images)
    HIPCHK(ctx, hipMemcpy(ctx->d_cams.p, ctx->cameras.data(), ctx->n_images * sizeof(dsm_camera), hipMemcpyHostToDevice));
  if (np) {
    HIPCHK(ctx, hipMemcpy(ctx->d_pairs_dev.p, ctx->pairs.data(), (size_t)np * 8, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_seeds.p, sd.data(), (size_t)np * 4, hipMemcpyHostToDevice));
  }
  int rc = (options->multiple_models ? verify_multiple : verify_core_plain)(
      ctx, np, ctx->d_pairs_dev.as<uint32_t>(), ctx->d_offsets.as<uint64_t>(), ctx->d_matches.as<uint32_t>(), ctx->total_matches,
      counts, ctx->d_kp.as<double>(), ctx->d_img_row0.as<uint32_t>(), ctx->d_cams.as<dsm_camera>(), options,
      ctx->d_seeds.as<uint32_t>(), stage_filter);
  if (rc != DSM_OK) return rc;
  ctx->verified = true;
  return DSM_OK;
}

// Guided matching over the pairs of the last dsm_verify_pairs: GuidedSiftCPUFeatureMatcher::Run
// (/root/reference/src/feature/matching.cc:441-470) + MatchGuidedSiftFeaturesCPU (src/feature/sift.cc:824-875).
int dsm_guided_match_pairs(dsm_ctx* ctx, const dsm_match_options* mo, const dsm_two_view_options* to, int32_t stage_filter) {
  if (!ctx || !mo || !to) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->verified) return fail(ctx, DSM_ERR_NOT_READY, "dsm_verify_pairs has not run");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint32_t np = ctx->n_pairs;
  std::vector<dsm_two_view_geometry> tv(np);
  std::vector<uint64_t> old_off((size_t)np + 1, 0);
  if (np) {
    HIPCHK(ctx, hipMemcpy(tv.data(), ctx->d_tvg.p, (size_t)np * sizeof(dsm_two_view_geometry), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(old_off.data(), ctx->d_inl_off.p, ((size_t)np + 1) * 8, hipMemcpyDeviceToHost));
  }
  // which pairs are re-matched, and against which matrix (sift.cc:838-863)
  std::vector<uint32_t> G;
  std::vector<int> mode(np, 0);
  for (uint32_t i = 0; i < np; ++i) {
    if ((uint64_t)tv[i].num_inliers < to->min_num_inliers) continue;  // matching.cc:449-453
    const int c = tv[i].config;
    if (c == DSM_CONFIG_CALIBRATED || c == DSM_CONFIG_UNCALIBRATED) mode[i] = 1;
    else if (c == DSM_CONFIG_PLANAR || c == DSM_CONFIG_PANORAMIC || c == DSM_CONFIG_PLANAR_OR_PANORAMIC) mode[i] = 2;
    if (mode[i]) G.push_back(i);
  }
  const bool cross = mo->cross_check != 0;
  const uint32_t ng = (uint32_t)G.size();
  std::vector<uint32_t> gcount(ng, 0);
  std::vector<uint64_t> goff((size_t)ng + 1, 0);
  HIPCHK(ctx, ctx->d_g_nfeat.reserve(std::max<uint32_t>(ctx->n_images, 1) * 4));
  if (ctx->n_images) HIPCHK(ctx, hipMemcpyAsync(ctx->d_g_nfeat.p, ctx->nfeat.data(), ctx->n_images * 4, hipMemcpyHostToDevice, st));
  HIPCHK(ctx, ctx->d_g_counts.reserve(std::max<uint32_t>(ng, 1) * 4));
  HIPCHK(ctx, ctx->d_g_offsets.reserve(((size_t)ng + 1) * 8));
  HIPCHK(ctx, ctx->d_g_total.reserve(8));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_g_total.p, 0, 8, st));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_g_offsets.p, 0, 8, st));
  uint64_t g_total = 0;
  const uint64_t budget_rows = (6ull << 30) / 4;
  std::vector<uint2> dpairs;
  std::vector<uint64_t> doff;
  std::vector<uint4> pdir;
  std::vector<float> gpar;
  const float max_residual = (float)(to->max_error * to->max_error);  // sift.cc:833
  uint32_t c0 = 0;
  while (c0 < ng) {
    uint64_t rows_acc = 0;
    uint32_t c1 = c0;
    while (c1 < ng) {
      const uint32_t a = ctx->pairs[2 * G[c1]], b = ctx->pairs[2 * G[c1] + 1];
      const uint64_t r = (uint64_t)ctx->rows[a] + (cross ? ctx->rows[b] : 0);
      if (c1 > c0 && rows_acc + r > budget_rows) break;
      rows_acc += r;
      ++c1;
    }
    const uint32_t nc = c1 - c0, nd = cross ? 2 * nc : nc;
    dpairs.resize(nd);
    doff.resize(nd);
    pdir.resize(nc);
    gpar.assign((size_t)nd * 12, 0.0f);
    uint64_t off = 0;
    uint32_t max_rb = 0;
    for (uint32_t k = 0; k < nc; ++k) {
      const uint32_t pi = G[c0 + k];
      const uint32_t a = ctx->pairs[2 * pi], b = ctx->pairs[2 * pi + 1];
      const double* Md = mode[pi] == 1 ? tv[pi].F : tv[pi].H;
      const uint32_t dab = cross ? 2 * k : k, dba = 2 * k + 1;
      for (int dir = 0; dir < (cross ? 2 : 1); ++dir) {
        const uint32_t dd = dir ? dba : dab;
        dpairs[dd] = dir ? make_uint2(b, a) : make_uint2(a, b);
        doff[dd] = off;
        off += ctx->rows[dir ? b : a];
        max_rb = std::max(max_rb, ctx->rows[dir ? b : a] / 256);
        float* g = gpar.data() + (size_t)dd * 12;
        for (int e = 0; e < 9; ++e) g[e] = (float)Md[e];  // F.cast<float>() / H.cast<float>(), sift.cc:835-836
        g[9] = (float)mode[pi];
        g[10] = dir ? 1.0f : 0.0f;
      }
      pdir[k] = make_uint4(dab, cross ? dba : 0, ctx->nfeat[a], ctx->nfeat[b]);
    }
    HIPCHK(ctx, ctx->d_g_dpairs.reserve(std::max<uint32_t>(nd, 1) * sizeof(uint2)));
    HIPCHK(ctx, ctx->d_g_doff.reserve(std::max<uint32_t>(nd, 1) * 8));
    HIPCHK(ctx, ctx->d_g_pdir.reserve(std::max<uint32_t>(nc, 1) * sizeof(uint4)));
    HIPCHK(ctx, ctx->d_g_params.reserve(std::max<size_t>(gpar.size(), 1) * 4));
    HIPCHK(ctx, ctx->d_g_m.reserve(std::max<uint64_t>(off, 1) * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_g_dpairs.p, dpairs.data(), nd * sizeof(uint2), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_g_doff.p, doff.data(), nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_g_pdir.p, pdir.data(), nc * sizeof(uint4), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_g_params.p, gpar.data(), gpar.size() * 4, hipMemcpyHostToDevice, st));
    KgParams kg;
    kg.desc = ctx->d_desc.as<int8_t>();
    kg.rterm = ctx->d_rterm.as<int32_t>();
    kg.kp = ctx->d_kp.as<double>();
    kg.dpairs = ctx->d_g_dpairs.as<uint2>();
    kg.img_row0 = ctx->d_img_row0.as<uint32_t>();
    kg.img_rows = ctx->d_img_rows.as<uint32_t>();
    kg.img_nfeat = ctx->d_g_nfeat.as<uint32_t>();
    kg.d_out_off = ctx->d_g_doff.as<uint64_t>();
    kg.lut = ctx->d_lut.as<float>();
    kg.max_ratio = (float)mo->max_ratio;
    kg.max_distance = (float)mo->max_distance;
    kg.max_residual = max_residual;
    kg.gparams = ctx->d_g_params.as<float>();
    kg.out = ctx->d_g_m.as<int32_t>();
    launch_kg(kg, nd, max_rb, st);
    HIPCHK(ctx, hipGetLastError());
    K2Params k2;
    k2.pair_dir = ctx->d_g_pdir.as<uint4>();
    k2.d_out_off = ctx->d_g_doff.as<uint64_t>();
    k2.m = ctx->d_g_m.as<int32_t>();
    k2.cross_check = cross ? 1 : 0;
    k2.counts = ctx->d_g_counts.as<uint32_t>() + c0;
    k2.offsets = ctx->d_g_offsets.as<uint64_t>() + c0;
    k2.matches = nullptr;
    launch_k2(k2, nc, false, st);
    launch_scan(ctx->d_g_counts.as<uint32_t>() + c0, ctx->d_g_offsets.as<uint64_t>() + c0, nc, ctx->d_g_total.as<uint64_t>(), st);
    HIPCHK(ctx, hipGetLastError());
    uint64_t total = 0;
    HIPCHK(ctx, hipMemcpyAsync(&total, ctx->d_g_total.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, ctx->d_g_matches.grow(std::max<uint64_t>(total, 1) * 8, g_total * 8, st));
    k2.matches = ctx->d_g_matches.as<uint32_t>();
    launch_k2(k2, nc, true, st);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));
    g_total = total;
    c0 = c1;
  }
  if (ng) {
    HIPCHK(ctx, hipMemcpy(gcount.data(), ctx->d_g_counts.p, (size_t)ng * 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(goff.data(), ctx->d_g_offsets.p, ((size_t)ng + 1) * 8, hipMemcpyDeviceToHost));
  }
  // final inlier lists: guided pairs take the new matches, the others keep theirs; then Match()'s post-filter
  struct Plan { uint64_t src_off; uint64_t dst_off; uint32_t count; uint32_t from_guided; };
  std::vector<Plan> plan(np);
  std::vector<uint64_t> new_off((size_t)np + 1, 0);
  std::vector<int> gslot(np, -1);
  for (uint32_t k = 0; k < ng; ++k) gslot[G[k]] = (int)k;
  uint64_t acc = 0;
  for (uint32_t i = 0; i < np; ++i) {
    uint32_t cnt = gslot[i] >= 0 ? gcount[gslot[i]] : tv[i].num_inliers;
    const uint64_t src = gslot[i] >= 0 ? goff[gslot[i]] : old_off[i];
    if (stage_filter && (uint64_t)cnt < to->min_num_inliers) {  // matching.cc:824-831
      const dsm_two_view_geometry keep = tv[i];
      memset(&tv[i], 0, sizeof(tv[i]));
      tv[i].num_matches = keep.num_matches;
      for (int e = 0; e < 4; ++e) {
        tv[i].num_trials[e] = keep.num_trials[e];
        tv[i].num_models[e] = keep.num_models[e];
      }
      cnt = 0;
    }
    tv[i].num_inliers = cnt;
    plan[i] = Plan{src, acc, cnt, gslot[i] >= 0 ? 1u : 0u};
    new_off[i] = acc;
    acc += cnt;
  }
  new_off[np] = acc;
  HIPCHK(ctx, ctx->d_g_plan.reserve(std::max<size_t>(np, 1) * sizeof(Plan)));
  HIPCHK(ctx, ctx->d_g_inl.reserve(std::max<uint64_t>(acc, 1) * 8));
  if (np) HIPCHK(ctx, hipMemcpyAsync(ctx->d_g_plan.p, plan.data(), (size_t)np * sizeof(Plan), hipMemcpyHostToDevice, st));
  launch_guided_assemble(ctx->d_g_plan.p, ctx->d_inl_compact.as<uint32_t>(), ctx->d_g_matches.as<uint32_t>(),
                         ctx->d_g_inl.as<uint32_t>(), np, st);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(st));
  std::swap(ctx->d_inl_compact, ctx->d_g_inl);
  HIPCHK(ctx, ctx->d_inl_off.reserve(((size_t)np + 1) * 8));
  HIPCHK(ctx, hipMemcpy(ctx->d_inl_off.p, new_off.data(), ((size_t)np + 1) * 8, hipMemcpyHostToDevice));
  if (np) HIPCHK(ctx, hipMemcpy(ctx->d_tvg.p, tv.data(), (size_t)np * sizeof(dsm_two_view_geometry), hipMemcpyHostToDevice));
  ctx->total_inliers = acc;
  return DSM_OK;
}

int dsm_get_two_view_geometries(dsm_ctx* ctx, dsm_two_view_geometry* out) {
  if (!ctx || !out) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->verified) return fail(ctx, DSM_ERR_NOT_READY, "dsm_verify_pairs has not run");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->n_pairs)
    HIPCHK(ctx, hipMemcpy(out, ctx->d_tvg.p, (size_t)ctx->n_pairs * sizeof(dsm_two_view_geometry), hipMemcpyDefault));
  return DSM_OK;
}

int dsm_get_inlier_matches(dsm_ctx* ctx, uint64_t* offsets, uint32_t* inlier_matches, uint64_t capacity) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->verified) return fail(ctx, DSM_ERR_NOT_READY, "dsm_verify_pairs has not run");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (offsets) HIPCHK(ctx, hipMemcpy(offsets, ctx->d_inl_off.p, ((size_t)ctx->n_pairs + 1) * 8, hipMemcpyDefault));
  if (inlier_matches) {
    if (capacity < ctx->total_inliers) return fail(ctx, DSM_ERR_OUT_OF_RANGE, "inlier_matches buffer too small");
    if (ctx->total_inliers)
      HIPCHK(ctx, hipMemcpy(inlier_matches, ctx->d_inl_compact.p, ctx->total_inliers * 8, hipMemcpyDefault));
  }
  return DSM_OK;
}

int dsm_get_verify_kernel_time(dsm_ctx* ctx, double* total_ms) {
  if (!ctx || !total_ms) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->verified) return fail(ctx, DSM_ERR_NOT_READY, "dsm_verify_pairs has not run");
  *total_ms = ctx->verify_ms;
  return DSM_OK;
}

int dsm_estimate_two_view_geometry(dsm_ctx* ctx, const dsm_camera* camera1, const double* points1, uint32_t n1,
                                   const dsm_camera* camera2, const double* points2, uint32_t n2,
                                   const uint32_t* matches, uint32_t n_matches, const dsm_two_view_options* options,
                                   uint32_t seed, dsm_two_view_geometry* out, uint32_t* inlier_matches) {
  if (!ctx || !camera1 || !camera2 || !options || !out || (n_matches && (!matches || !points1 || !points2)))
    return DSM_ERR_INVALID_ARGUMENT;
  if (!cam_model_exists(camera1->model_id) || !cam_model_exists(camera2->model_id))
    return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "camera model id does not exist (0..10)");
  for (uint32_t i = 0; i < n_matches; ++i)
    if (matches[2 * i] >= n1 || matches[2 * i + 1] >= n2) return fail(ctx, DSM_ERR_OUT_OF_RANGE, "match index out of range");
  if (!ctx->leaf) {
    int rc = dsm_ctx_create(ctx->device, &ctx->leaf);
    if (rc != DSM_OK) return fail(ctx, rc, dsm_last_error(nullptr));
  }
  dsm_ctx* lf = ctx->leaf;
  lf->debug_options = ctx->debug_options;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf kp, row0, cams, pr, off, mt, sd;
  const uint32_t rows0[2] = {0, n1};
  const dsm_camera cc[2] = {*camera1, *camera2};
  const uint32_t prs[2] = {0, 1};
  const uint64_t offs[2] = {0, n_matches};
  int rc = DSM_OK;
#define LCHK(call)                                                      \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess && rc == DSM_OK) {                             \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);     \
      rc = DSM_ERR_HIP;                                                 \
    }                                                                   \
  } while (0)
  LCHK(kp.reserve(std::max<size_t>((size_t)(n1 + n2) * 16, 16)));
  LCHK(row0.reserve(8));
  LCHK(cams.reserve(2 * sizeof(dsm_camera)));
  LCHK(pr.reserve(8));
  LCHK(off.reserve(16));
  LCHK(mt.reserve(std::max<size_t>((size_t)n_matches * 8, 8)));
  LCHK(sd.reserve(4));
  if (rc == DSM_OK) {
    if (n1) LCHK(hipMemcpy(kp.p, points1, (size_t)n1 * 16, hipMemcpyHostToDevice));
    if (n2) LCHK(hipMemcpy(kp.as<double>() + 2 * (size_t)n1, points2, (size_t)n2 * 16, hipMemcpyHostToDevice));
    LCHK(hipMemcpy(row0.p, rows0, 8, hipMemcpyHostToDevice));
    LCHK(hipMemcpy(cams.p, cc, sizeof(cc), hipMemcpyHostToDevice));
    LCHK(hipMemcpy(pr.p, prs, 8, hipMemcpyHostToDevice));
    LCHK(hipMemcpy(off.p, offs, 16, hipMemcpyHostToDevice));
    if (n_matches) LCHK(hipMemcpy(mt.p, matches, (size_t)n_matches * 8, hipMemcpyHostToDevice));
    LCHK(hipMemcpy(sd.p, &seed, 4, hipMemcpyHostToDevice));
  }
  if (rc == DSM_OK) {
    std::vector<uint32_t> counts(1, n_matches);
    lf->n_pairs = 1;
    rc = (options->multiple_models ? verify_multiple : verify_core_plain)(
        lf, 1, pr.as<uint32_t>(), off.as<uint64_t>(), mt.as<uint32_t>(), n_matches, counts, kp.as<double>(),
        row0.as<uint32_t>(), cams.as<dsm_camera>(), options, sd.as<uint32_t>(), 0);
    if (rc != DSM_OK) ctx->err = lf->err;
  }
  if (rc == DSM_OK) {
    LCHK(hipMemcpy(out, lf->d_tvg.p, sizeof(dsm_two_view_geometry), hipMemcpyDeviceToHost));
    if (rc == DSM_OK && inlier_matches && out->num_inliers)
      LCHK(hipMemcpy(inlier_matches, lf->d_inl_compact.p, (size_t)out->num_inliers * 8, hipMemcpyDeviceToHost));
  }
#undef LCHK
  DevBuf* bufs[] = {&kp, &row0, &cams, &pr, &off, &mt, &sd};
  for (DevBuf* b : bufs) b->release();
  return rc;
}

int dsm_debug_verify_counters(dsm_ctx* ctx, uint32_t* out16) {
  if (!ctx || !out16) return DSM_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < 16; ++k) out16[k] = 0;
  for (uint32_t li = 0; li < ctx->verify_lanes && li < DSM_VERIFY_MAX_LANES; ++li)
    for (int k = 0; k < 16; ++k) out16[k] += ctx->lanes[li].dbg[k];
  return DSM_OK;
}

int dsm_debug_sample_sequence(dsm_ctx* ctx, uint32_t seed, uint32_t k, uint32_t total, uint32_t n_draws, uint32_t* out) {
  if (!ctx || !out || k == 0 || k > total) return DSM_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf o, idx, tmp7;
  HIPCHK(ctx, o.reserve((size_t)k * n_draws * 4 + 4));
  HIPCHK(ctx, idx.reserve((size_t)total * 4));
  HIPCHK(ctx, tmp7.reserve((size_t)n_draws * 7 * 4 + 4));
  // 1 (default): the wave sampler of the product path; 0: lane-0 reference loop; 2: the sampler's serial replay path
  const char* mode_env = ctx->dbg("DSM_DEBUG_SAMPLER_MODE");
  int mode = mode_env ? atoi(mode_env) : 1;
  if (!(k == 1 || k == 4 || k == 5 || k == 7)) mode = 0;
  launch_debug_samples(seed, k, total, n_draws, o.as<uint32_t>(), idx.as<uint32_t>(), tmp7.as<uint32_t>(), mode, ctx->stream);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(out, o.p, (size_t)k * n_draws * 4, hipMemcpyDeviceToHost));
  o.release();
  idx.release();
  tmp7.release();
  return DSM_OK;
}

int dsm_debug_image_to_world(dsm_ctx* ctx, const dsm_camera* camera, uint32_t n, const double* xy, double* out_uv) {
  if (!ctx || !camera || (n && (!xy || !out_uv))) return DSM_ERR_INVALID_ARGUMENT;
  if (!cam_model_exists(camera->model_id)) return fail(ctx, DSM_ERR_INVALID_ARGUMENT, "camera model id does not exist (0..10)");
  if (!n) return DSM_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf in, out;
  HIPCHK(ctx, in.reserve((size_t)n * 16));
  HIPCHK(ctx, out.reserve((size_t)n * 16));
  HIPCHK(ctx, hipMemcpy(in.p, xy, (size_t)n * 16, hipMemcpyHostToDevice));
  launch_debug_image_to_world(*camera, n, in.as<double>(), out.as<double>(), ctx->stream);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(out_uv, out.p, (size_t)n * 16, hipMemcpyDeviceToHost));
  in.release();
  out.release();
  return DSM_OK;
}

int dsm_ctx_device(const dsm_ctx* ctx) { return ctx ? ctx->device : -1; }
uint32_t dsm_ctx_num_pairs(const dsm_ctx* ctx) { return (ctx && ctx->matched) ? ctx->n_pairs : 0u; }

int dsm_get_match_kernel_time(dsm_ctx* ctx, double* total_ms, uint32_t* n_launches) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->matched) return fail(ctx, DSM_ERR_NOT_READY, "dsm_match_pairs has not run");
  if (total_ms) *total_ms = ctx->k1_ms;
  if (n_launches) *n_launches = ctx->k1_launches;
  return DSM_OK;
}

int dsm_get_match_gather_time(dsm_ctx* ctx, double* total_ms) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->matched) return fail(ctx, DSM_ERR_NOT_READY, "dsm_match_pairs has not run");
  if (total_ms) *total_ms = ctx->k1g_ms;
  return DSM_OK;
}

int dsm_get_match_tail_time(dsm_ctx* ctx, double* total_ms) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->matched) return fail(ctx, DSM_ERR_NOT_READY, "dsm_match_pairs has not run");
  if (total_ms) *total_ms = ctx->k1t_ms;
  return DSM_OK;
}

int dsm_get_match_resolve_time(dsm_ctx* ctx, double* total_ms) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  if (!ctx->matched) return fail(ctx, DSM_ERR_NOT_READY, "dsm_match_pairs has not run");
  if (total_ms) *total_ms = ctx->k1b_ms;
  return DSM_OK;
}

}  // extern "C"
