// ctx.h -- the context behind the C-ABI (shared by capi.hip and capi_retrieval.hip; not part of the public header).
#ifndef DAGSFM_AMD_CSRC_CTX_H_
#define DAGSFM_AMD_CSRC_CTX_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dagsfm_mi355x.h"

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      e = hipMalloc(&p, bytes);
      want = bytes;
    }
    if (e == hipSuccess) cap = want;
    return e;
  }
  // reserve keeping the old contents (first `keep` bytes)
  hipError_t grow(size_t bytes, size_t keep, hipStream_t st) {
    if (bytes <= cap) return hipSuccess;
    void* np = nullptr;
    size_t want = bytes + bytes / 2 + 256;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) {
      want = bytes;
      e = hipMalloc(&np, want);
    }
    if (e != hipSuccess) return e;
    if (p && keep) {
      e = hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (p) (void)hipFree(p);
    p = np;
    cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};


// One lane of dsm_verify_pairs: a share of the pair list runs through the whole E -> F -> H pipeline on the lane's own
// stream, driven by its own host thread, with its own chunk-local buffers.  The rounds of one lane are a serial chain
// of small launches and host round trips (its length is set by the slowest pair); lanes fill each other's bubbles.
struct VerifyLane {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  DevBuf samples, draws_end, nmodels, vcounts, vsums, models, ework, active, vscratch;
  DevBuf lo_queue, lo_work, lo_models, lo_slots, lo_ework;  // batched local optimisation
  DevBuf tail_items, tail_n, lo_jobs, job_list;             // item passes (TailItem, LoJob)
  DevBuf hyp_map;                                           // the round's hypotheses of all pairs for the compact solver grids (verify_kernels.hip hyp_of_lane)
  uint32_t rounds[3] = {0, 0, 0}, lo_iters[3] = {0, 0, 0};
  uint32_t dbg[32] = {0};
  uint32_t* host_ctr = nullptr;  // pinned read-back target of the lane's counters
  std::string err;
  int rc = 0;
};
#define DSM_VERIFY_MAX_LANES 4

struct dsm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // resident images
  uint32_t n_images = 0;
  uint64_t total_rows = 0;
  std::vector<uint32_t> nfeat, row0, rows;
  std::vector<dsm_camera> cameras;
  bool have_kp = false;
  DevBuf d_desc, d_rterm, d_kp, d_img_row0, d_img_rows, d_lut;
  // hand-over of pageable host buffers (capi.hip: staged_upload): two pinned host slots, their device mirror for the descriptor
  // rows on their way through k0_prepare, one event per slot; allocated by the first upload that needs them
  void* h_stage = nullptr;
  DevBuf d_stage;
  hipEvent_t stage_ev[2] = {nullptr, nullptr};

  // last dsm_match_pairs
  bool matched = false;
  uint32_t n_pairs = 0;
  std::vector<uint32_t> pairs;  // n_pairs x 2
  DevBuf d_dpairs, d_doutoff, d_pair_dir, d_m, d_counts, d_offsets, d_matches, d_total;
  uint64_t total_matches = 0;
  double k1_ms = 0.0;
  double k1b_ms = 0.0;  // k1_resolve_index
  double k1t_ms = 0.0;  // after pass 2: its k1_resolve_index + the compaction of the mutual matches (host wait for the total included)
  double k1g_ms = 0.0;  // k1_best_rows<GATHER> (pass 2 of the cross-check)
  DevBuf d_order, d_dpairs2, d_ecnt, d_eoff, d_etotal, d_entries, d_out2, d_ms, d_out2s;
  uint32_t k1_launches = 0;
  std::vector<hipEvent_t> ev;

  // last dsm_verify_pairs
  bool verified = false;
  DevBuf d_cams, d_pairs_dev, d_seeds, d_tvg, d_inl, d_inl_counts, d_inl_off, d_inl_compact, d_vscratch, d_inl_total;
  DevBuf d_nt_table, d_nt_off, d_nt_off_t, d_pair_state, d_pts_px, d_pts_norm, d_reports, d_masks;
  DevBuf d_fam_state, d_sidx, d_lo_inl;
  DevBuf d_nt_table_t, d_wm_redo, d_wm_total, d_wm_count, d_lo_inl_pool, d_pose_jobs;
  VerifyLane lanes[DSM_VERIFY_MAX_LANES];
  uint32_t verify_lanes = 1;  // lanes of the last call
  uint32_t verify_lo_iters[3] = {0, 0, 0};
  DevBuf d_g_nfeat, d_g_dpairs, d_g_doff, d_g_pdir, d_g_params, d_g_m, d_g_counts, d_g_offsets, d_g_total, d_g_matches,
      d_g_plan, d_g_inl, d_g_inl_off;  // guided matching
  DevBuf d_mm_matches[2], d_mm_off[2], d_mm_counts, d_mm_state, d_mm_first, d_mm_acc, d_mm_keep, d_mm_total;  // EstimateMultiple
  uint32_t verify_rounds[3] = {0, 0, 0};
  uint64_t total_inliers = 0;
  double verify_ms = 0.0;
  // cache of the tabulated RANSAC::ComputeNumTrials (host libm), keyed by confidence
  double nt_confidence = -1.0;
  std::vector<uint32_t> nt_table;        // E / F / H tables of the match counts seen so far, back to back
  std::vector<uint64_t> nt_off;          // per N (0 = absent; offsets are stored +1)
  std::vector<uint32_t> nt_table_t;      // translation tables of the inlier counts that needed one (built on demand)
  std::vector<uint64_t> nt_off_t;        // per N (0 = absent; offsets are stored +1)
  bool nt_dirty = true, nt_dirty_t = true;
  hipEvent_t vev0 = nullptr, vev1 = nullptr;

  dsm_ctx* leaf = nullptr;  // private context of the one-shot leaf entry points

  // dsm_set_debug_option: switches of THIS context (none changes a result).  The library never reads the process
  // environment: a host application's environment cannot change schedules.
  // Two builds of the same sources (Makefile):
  //   libdagsfm_mi355x.so        the product: only the scheduling knobs a deployer could want (dsm_product_debug_keys);
  //                              the cross-check schedules are not compiled in
  //   libdagsfm_mi355x_check.so  -DDSM_CHECK_BUILD: additionally the independent schedules of the same results that
  //                              tools/check_schedules.py and the schedule-parametrised tests compare the product with
  //                              (legacy one-kernel LO-RANSAC, fused / LDS forms of the 5-point root finder,
  //                              the v_dot4 K1 and word assignment, counters of DSM_VERIFY_DEBUG / DSM_SCORE_PREFILTER=check)
  std::map<std::string, std::string> debug_options;
  const char* dbg(const char* key) const {
    const auto it = debug_options.find(key);
    return it == debug_options.end() ? nullptr : it->second.c_str();
  }

  // dsm_ctx_set_memory_budget: bytes the chunk planners of dsm_match_pairs / dsm_verify_pairs may spend on their transient
  // scratch (0: the default policy -- 8 GiB of K1 output per chunk, 40 % of the free memory, 4 .. 96 GiB, for the verifier)
  uint64_t memory_budget = 0;

  struct RetrievalState* retrieval = nullptr;  // vocabulary-tree retrieval (retrieval.hip), created on first use
};
void dsm_retrieval_destroy(dsm_ctx* ctx);     // retrieval.hip
void dsm_retrieval_invalidate(dsm_ctx* ctx);  // retrieval.hip: the resident images changed

// keys dsm_set_debug_option accepts (an unknown key is an error, so a check-only switch fails loudly on the product build)
static const char* const dsm_product_debug_keys[] = {"DSM_MATCH_CHUNK_ROWS", "DSM_VERIFY_CHUNK_PAIRS", "DSM_VERIFY_LANES", "DSM_VERIFY_INLINE_LO",
                                                     "DSM_VERIFY_ITEM_MODE", "DSM_LO_TAIL", "DSM_LO_TAIL_MODE", "DSM_VERIFY_GRID_DIV"};
#ifdef DSM_CHECK_BUILD
static const char* const dsm_check_debug_keys[] = {"DSM_K1_DOT4", "DSM_VERIFY_DEBUG", "DSM_SAMPLER_SERIAL", "DSM_LO_PREPARE_WAVE", "DSM_LO_JACOBI_GROUPS",
                                                   "DSM_ROOTS_LDS", "DSM_FINAL_WAVES", "DSM_VERIFY_LEGACY", "DSM_VERIFY_FIXED_BATCH", "DSM_VERIFY_LANE_SPLIT",
                                                   "DSM_DEBUG_SAMPLER_MODE", "DSM_VOCAB_ASSIGN_VALU", "DSM_VERIFY_HOST_LOOP", "DSM_SCORE_PREFILTER",
                                                   "DSM_VERIFY_REPLAY_GRID", "DSM_REPLAY_LEGACY", "DSM_FLANN_GROUP", "DSM_FLANN_STATS", "DSM_ELU_LDS", "DSM_HYP_GRID", "DSM_SPEC_MARGIN"};
#endif

#define HIPCHK(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
      return DSM_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)


static inline int dsm_fail(dsm_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#endif  // DAGSFM_AMD_CSRC_CTX_H_
