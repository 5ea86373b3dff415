// kernels.h -- parameter blocks and launchers shared by the HIP kernels and the C-ABI layer.
#ifndef DAGSFM_AMD_CSRC_KERNELS_H_
#define DAGSFM_AMD_CSRC_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dagsfm_mi355x.h"

// K1: one directed matching pass per blockIdx.x, one 256-row block of image a per blockIdx.y.
struct K1Params {
  const int8_t* desc;         // s8 descriptors (u8 ^ 0x80), all images, rows padded per image to 256
  const int32_t* rterm;       // 128 * sum(s8 row)
  const uint2* dpairs;        // directed pairs {image a, image b}
  const uint32_t* img_row0;   // first padded row of every image
  const uint32_t* img_rows;   // padded row count of every image (multiple of 256)
  const uint64_t* d_out_off;  // per directed pair: offset (in rows) into `out`
  const float* lut;           // acosf(min(d / 2^18, 1)), d = 0..262144, built on the host
  float max_ratio;
  float max_distance;
  int32_t* out;               // per row: matched column index or -1 (between K1 and K1b: tile * 4 + column set)
  int32_t* out_s;             // per flagged row, between K1 and K1b: K1's second-best dot product (a lower bound)
  // gathered pass (pass 2 of the cross-check); entries == nullptr: plain pass over an image's rows
  const uint32_t* order;      // optional: blockIdx.x -> directed pair (launch order, sorted by column image)
  const uint2* entries;       // one-way matches (i1, i2) of all pairs; entry k of pair d = row i2 of image dpairs[d].x
  const uint64_t* e_off;      // [n_pairs] first entry of every pair (also the pair's offset into `out`)
  const uint32_t* e_cnt;      // [n_pairs] entries of every pair
};

// Cross-check on the gathered pass' result: keep entry (i1, i2) when out2 == i1.
struct K2eParams {
  const uint2* entries;
  const uint64_t* e_off;
  const uint32_t* e_cnt;
  const int32_t* out2;        // per entry: matches21[i2] (column of image a) or -1
  uint32_t* counts;           // [n_pairs] (count pass)
  const uint64_t* offsets;    // [n_pairs] absolute offsets into matches (write pass)
  uint32_t* matches;          // [total][2]
};

// K2: mutual check + ordered compaction, one workgroup per undirected pair.
struct K2Params {
  const uint4* pair_dir;      // {directed idx a->b, directed idx b->a, n1, n2}
  const uint64_t* d_out_off;
  const int32_t* m;           // K1 output
  int32_t cross_check;
  uint32_t* counts;           // [n_pairs] (count pass)
  const uint64_t* offsets;    // [n_pairs] absolute offsets into matches (write pass)
  uint32_t* matches;          // [total][2]
};

// Guided matching (MatchGuidedSiftFeaturesCPU, sift.cc:824-875): one directed pass with the geometric filter.
struct KgParams {
  const int8_t* desc;         // as K1Params
  const int32_t* rterm;
  const double* kp;           // keypoints (x, y) of all images, exact float values widened to double
  const uint2* dpairs;        // directed pairs {row image, column image}
  const uint32_t* img_row0;
  const uint32_t* img_rows;   // padded row counts
  const uint32_t* img_nfeat;  // true feature counts
  const uint64_t* d_out_off;
  const float* lut;
  float max_ratio, max_distance, max_residual;
  const float* gparams;       // per directed pair: 9 x float matrix (row-major), mode (1 F / 2 H), swap (row image is image 2), pad
  int32_t* out;               // per row: matched column or -1
};
void launch_kg(const KgParams& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st);
// plan: per pair {uint64 src_off, uint64 dst_off, uint32 count, uint32 from_guided}: copies the pair's final inlier matches
void launch_guided_assemble(const void* plan, const uint32_t* old_inl, const uint32_t* guided, uint32_t* dst, uint32_t n_pairs,
                            hipStream_t st);

// Result of one LO-RANSAC family for one pair (RANSAC<>::Report, /root/reference/src/optim/ransac.h:80-97).
struct RansacReport {
  bool success;
  uint32_t num_trials;
  uint32_t num_models;
  uint32_t num_inliers;
  double residual_sum;
  double model[9];
};

// Progress of one LO-RANSAC family for one pair between the rounds of the phase-split pipeline.
struct FamState {
  RansacReport rep;     // best support so far; rep.num_trials = trials consumed so far
  uint32_t dyn_max;     // dyn_max_num_trials
  uint32_t active;      // 1 while more trials are needed
  uint32_t rounds;      // sampling rounds done
  uint32_t nb;          // trials sampled in the current round
  // suspended replay (local optimisation as its own batched kernels): where k_replay_lo resumes
  uint32_t t_pos;       // trial of the current batch to look at next
  uint32_t m_pos;       // model of that trial to look at next
  uint32_t lo_wait;     // 1: the pair waits for the models of a local optimisation (triggered by model m_pos - 1)
  uint32_t lo_ninl;     // inliers handed to that local optimisation
  uint32_t lo_nm;       // models it produced
  uint32_t pad;
};
#define LO_WORK_DOUBLES 176  // per pair: W[81] V[81] n1[3] n2[3] scale dsz, padded

// The tail of a round (few pairs left in the local-optimisation queue): every local optimisation a pair can still
// reach in this round is computed AT ONCE, one wave per (pair, candidate model), instead of one after the other.
// The outcome of LORANSAC's optimisation step (loransac.h:160-178) depends on the candidate model alone -- the model
// has just become the best one, its inliers are the input, the local models are compared with it in order -- so it
// can be computed before the sequential scan knows whether the step will be taken.  An item is that outcome.
#define TAIL_KMAX 8  // candidates per pair and tail iteration (a pair with more is simply queued once more)
// EstimateWithRelativePose (two_view_geometry.cc:232-290) leaves k_verify_final as a job: the candidate poses of the pair
// (the four (R, t) of DecomposeEssentialMatrix, or the solutions of the homography decomposition); k_final_pose checks
// them, a wave per (pair, candidate); k_final_finish picks the winner.
struct PoseJob {
  int32_t ncmb;        // candidates (0: the pair has no pose step)
  int32_t ni;          // inlier points (compacted to the front of the pair's pts_norm rows)
  double Rc[4 * 9];
  double tc[4 * 3];
  int32_t cnt[4];      // k_final_pose: points in front of both cameras, per candidate
  double med[4];       // k_final_pose: median triangulation angle of those points, per candidate
};
struct LoJob {
  uint32_t pl;          // pair (chunk-local)
  uint32_t ninl;        // inliers of the candidate model = size of the local estimator's input
  uint32_t pending;     // 1: the step the pair is suspended at -- its inliers are the pair's lo_inl list
  uint32_t nm;          // local models produced
};
struct TailItem {
  uint32_t t, m;        // trial of the current batch, model of that trial
  uint32_t nlo;         // local models the optimisation produced (they count as scored models)
  uint32_t out_n;       // support after the step: inliers ...
  double out_sum;       // ... residual sum ...
  double out_model[9];  // ... and model (the candidate itself when no local model beat it)
};

// Two-view verification: one 64-lane workgroup per image pair (grid-stride over the pair list).
// waves per SIMD the replay scans (k_replay_lo, modes 0 and 2) are compiled for; their persistent grid and the lanes' per-workgroup
// scratch are sized with it (capi.hip)
#ifndef DSM_REPLAY_WAVES
#define DSM_REPLAY_WAVES 4
#endif

struct VerifyParams {
  const uint32_t* pairs;       // [n_pairs][2] image indices
  const uint64_t* match_off;   // [n_pairs+1] offsets into matches
  const uint32_t* matches;     // [total][2]
  const double* kp;            // [total_rows][2] keypoints (x, y)
  const uint32_t* img_row0;    // first keypoint row of every image
  const dsm_camera* cams;      // per image
  dsm_two_view_options opt;
  const uint32_t* seeds;       // per pair PRNG seed
  const uint32_t* nt_table;    // tabulated RANSAC::ComputeNumTrials (host libm)
  const uint64_t* nt_off;      // [n_max+1]: offset of the E/F/H tables (3 x (N+1)) for N matches
  const uint32_t* nt_table_t;  // translation tables (DetectWatermark's RANSAC), built on demand: see wm_redo
  const uint64_t* nt_off_t;    // [n_max+1]: offset + 1 of the translation table (N+1 entries) for N inliers, 0 = not built
  // a pair whose watermark test needs a translation table that does not exist yet is recorded here and left alone;
  // the host builds the tables and runs k_verify_final again over exactly those pairs (final_list)
  uint32_t* wm_redo;           // [n_pairs] pair indices
  uint32_t* wm_total;          // [n_pairs] their inlier counts (= the table they need)
  uint32_t* wm_count;
  const uint32_t* final_list;  // != nullptr: k_verify_final / k_final_pose / k_final_finish process pairs final_list[0 .. n_final)
  PoseJob* pose_jobs;          // [n_pairs]
  uint32_t n_final;
  uint32_t max_trials[4];      // RANSAC ctor's max_num_trials per family (E, F, H, T)
  uint32_t first_batch[3];     // trials speculated in a pair's first round (<= batch; later rounds draw what the dynamic stop asks for, up to batch)
  dsm_two_view_geometry* tvg;  // [n_pairs]
  uint32_t* inlier_matches;    // [total][2], pair p at match_off[p]
  uint32_t* inl_counts;        // [n_pairs]
  double* scratch;             // per workgroup work area
  // per-pair state that travels between the phase kernels
  uint32_t* pair_state;        // [n_pairs][640]: MT19937 state (624 words) + index
  double* pts_px;              // [total][4] matched pixel points x1 y1 x2 y2
  double* pts_norm;            // [total][4] the same through Camera::ImageToWorld (calibrated pairs)
  RansacReport* reports;       // [n_pairs][3] E, F, H
  unsigned char* masks;        // [3][mask_stride] inlier masks of the three families
  uint64_t mask_stride;
  // phase-split pipeline (sample -> solve+score -> replay), one chunk of pairs [pair0, pair0 + n_chunk)
  struct FamState* fam_state;  // [n_pairs][3]
  uint32_t* samples;           // [n_chunk][batch][7] minimal-sample indices
  uint32_t* draws_end;         // [n_chunk][batch] generator calls consumed up to the end of each trial's sample
  int32_t* nmodels;            // [n_chunk][batch]
  int32_t* counts;             // [n_chunk][batch][maxm]
  double* sums;                // [n_chunk][batch][maxm] in-order residual sums of the inliers (F and H; k_score)
  int lo_reg_prepare;          // k_lo_prepare_reg takes the tall problems, k_lo_prepare only the rest
  int dbg_elu_lds;             // check build, DSM_ELU_LDS: k_solve_e_lu (the 10 x 10 elimination in LDS) instead of k_solve_e_lu_reg
  int dbg_jacobi_groups, dbg_roots_lds, dbg_final_waves;  // dsm_set_debug_option switches the launch helpers read
  int score_prefilter;         // F / H scoring as bound + exact (k_prescore, k_score_needed); 0: plain k_score (DSM_SCORE_PREFILTER=0)
  int stats;                   // DSM_VERIFY_DEBUG: count candidates / local optimisations (one-address atomics) in the replay
  uint32_t spec_margin[3];     // k_sample: trials a later round speculates beyond what the dynamic stop asks for (samples without a model)
  uint32_t* hyp_map;           // E / F rounds after a pair's first: the round's hypotheses of all pairs, listed by k_sample in 64 segments
  uint32_t hyp_seg_cap;        // (verify_kernels.hip hyp_of_lane); nullptr: the solvers run the (pair, block of 64 trials) grid.  Entries per segment
  uint32_t rp_cap;             // k_replay_rp: correspondences of a pair its LDS holds (min(n_max, RP_CAP)); longer pairs read global memory
  int replay_legacy;           // check build, DSM_REPLAY_LEGACY: the replay scans as k_replay_lo<fam, 0 / 2> (points and residuals through memory)
  double* models;              // [n_chunk][batch][maxm][9]
  uint32_t* sidx_g;            // [total] RandomSampler's persistent index array of every pair (at match offsets)
  uint32_t* active_count;      // the lane's 32 classic counters ([0]: pairs that still need trials after a replay round); the segmented
                               // hand-out counters of the persistent grids lie around them (verify_kernels.hip GRAB_*, capi.hip LANE_CTR_*)
  uint32_t pair0, n_chunk;     // chunk of the pair list handled by this launch
  uint32_t batch;              // trials speculated per round for the family being launched
  uint32_t n_pairs;
  uint32_t n_max;              // max matches of any pair in this launch
  int32_t stage_filter;        // apply SiftFeatureMatcher::Match's min_num_inliers post-filter
  double* e_work;              // 5-point solver: the 10 x 20 constraint matrix of every hypothesis of the E batch
  int32_t sampler_serial;      // test hook (DSM_SAMPLER_SERIAL): force the sampler's serial replay path
  int32_t reseed;              // 1: k_verify_prep seeds the pair's generator; 0: it continues (EstimateMultiple passes)
  int32_t keep_generator;      // 1: k_verify_final stores the generator state for a following pass
  // local optimisation as batched kernels: k_replay_lo suspends a pair at every LO, the LO runs for all suspended
  // pairs at once (k_lo_prepare: wave per pair; k_lo_jacobi: 16-lane group per pair; E: the flat 5-point kernels)
  uint32_t* lo_inl;            // [total] ordered inlier indices of the pair's pending LO (at match offsets)
  TailItem* tail_items;        // [n_work][TAIL_KMAX] (see TailItem)
  uint32_t* tail_n;            // [n_work] items of every pair of the pass
  // item passes: the batched local-optimisation kernels work on JOBS (one per item) instead of queued pairs
  struct LoJob* lo_jobs;       // != nullptr: work index -> job (worklist, if given, maps the launch index to the job index)
  uint32_t* lo_inl_pool;       // inlier lists of the items: item k of pair pi at (match_off[pi] * TAIL_KMAX + k * n)
  uint32_t* job_list;          // compact list of the jobs of a pass (k_items_enum)
  const uint32_t* worklist;    // k_replay_lo / LO kernels: chunk-local pair indices to process (nullptr: all pairs)
  uint32_t n_work;
  uint32_t* lo_queue_g;        // k_replay_lo: the queued pairs that need the general LO kernels (count at active_count[22])
  uint32_t* lo_queue;          // out: pairs that k_replay_lo suspended
  uint32_t* lo_count;          // out: their number
  double* lo_work;             // [n_chunk][LO_WORK_DOUBLES]
  double* lo_models;           // [n_chunk][10][9]
  double* lo_slots;            // E: [n_chunk][90] (null-space basis, B(z), determinant polynomial / roots)
  double* lo_ework;            // E: [n_chunk][200] constraint matrix
};

size_t verify_scratch_bytes_per_block(uint32_t n_max);
size_t verify_smem_bytes(uint32_t n_max);
void launch_verify(const VerifyParams& p, uint32_t n_blocks, hipStream_t st);
// phase-split pipeline launches (fam: 0 = E, 1 = F, 2 = H)
void launch_vp_prep(const VerifyParams& p, uint32_t n_blocks, hipStream_t st);
void launch_vp_sample(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st);
void launch_vp_solve_score(const VerifyParams& p, int fam, hipStream_t st);
void launch_vp_replay(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st);
void launch_vp_final(const VerifyParams& p, uint32_t n_blocks, hipStream_t st);
void launch_vp_replay_lo(const VerifyParams& p, int fam, uint32_t n_blocks, int mode, hipStream_t st);  // mode: 0 suspend, 1 inline tail, 2 item lookup
void launch_vp_items_enum(const VerifyParams& p, int fam, hipStream_t st);
void launch_vp_items_inliers(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st);
void launch_vp_items_outcome(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st);
// over p.worklist / p.n_work; n_wave_prepare / n_small_jacobi: how many of the queued problems need the general kernels
// (k_lo_prepare: not register-preparable; k_lo_jacobi: smaller than 9 x 9), counted by k_replay_lo at [22] / [23]
void launch_vp_local_opt(const VerifyParams& p, int fam, uint32_t n_blocks, uint32_t n_wave_prepare, uint32_t n_small_jacobi, uint32_t n_big_prepare, hipStream_t st);
void launch_lane_counters(const void* src, uint32_t* dst_host, uint32_t copy_bytes, void* zero, size_t zero_bytes, hipStream_t st);
uint32_t vp_batch(int fam, uint32_t max_trials, uint32_t min_trials);
uint32_t vp_maxm(int fam);
void debug_read_prof(unsigned long long* out16);
void launch_debug_samples(uint32_t seed, uint32_t k, uint32_t total, uint32_t n_draws, uint32_t* out, uint32_t* idx, uint32_t* tmp7,
                          int mode, hipStream_t st);
void launch_debug_image_to_world(const dsm_camera& cam, uint32_t n, const double* xy, double* uv, hipStream_t st);
void launch_compact_inliers(const uint64_t* match_off, const uint64_t* inl_off, const uint32_t* inl_counts,
                            const uint32_t* src, uint32_t* dst, uint32_t n_pairs, hipStream_t st);

void launch_k0(const uint8_t* in_u8, int8_t* out_s8, int32_t* rterm, uint64_t n_rows, hipStream_t st);
void launch_k1(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st);
void launch_k1_dot4(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st);
void launch_k1_resolve(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st);
void launch_k2(const K2Params& p, uint32_t n_pairs, bool write, hipStream_t st);
void launch_k2_entries(const K2eParams& p, uint32_t n_pairs, bool write, hipStream_t st);
void launch_scan(const uint32_t* counts, uint64_t* offsets, uint32_t n, uint64_t* running_total, hipStream_t st);

// EstimateMultiple (two_view_geometry.cc:128-167): per-pair bookkeeping across the passes
struct MultiState {
  uint32_t ngeo;      // geometries collected so far
  uint32_t done;      // a pass returned DEGENERATE
  uint32_t acc_inl;   // inlier matches accumulated (stored at the pair's ORIGINAL match offset)
  uint32_t pad;
  uint32_t trials[4], models[4];  // counters summed over the passes
};
struct MultiParams {
  const uint64_t* cur_off;       // match offsets of this pass
  const uint32_t* cur_matches;
  const uint64_t* orig_off;      // match offsets of the first pass
  const dsm_two_view_geometry* tvg;  // results of this pass
  const uint32_t* inl;           // inlier matches of this pass (at cur offsets)
  const uint32_t* inl_counts;
  MultiState* state;
  dsm_two_view_geometry* first;  // first collected geometry of every pair
  uint32_t* acc;                 // accumulated inlier matches (at orig offsets)
  unsigned char* keep;           // per match of this pass: stays in the remaining set
  uint32_t* next_count;
  const uint64_t* next_off;
  uint32_t* next_matches;
  uint32_t* active;
  int32_t ignore_watermark;
  int32_t stage_filter;
  uint64_t min_num_inliers;
  uint32_t n_pairs;
  // finalize
  dsm_two_view_geometry* out_tvg;
  uint32_t* out_inl_counts;
};
void launch_multi_accumulate(const MultiParams& p, hipStream_t st);
void launch_multi_scatter(const MultiParams& p, hipStream_t st);
void launch_multi_finalize(const MultiParams& p, hipStream_t st);

#endif
