/*
 * dagsfm_mi355x.h -- C-ABI of the MI355X-native matching + two-view verification path.
 *
 * This is the drop-in boundary for DAGSfM's data-parallel hot path
 * (SURVEY.md section 8b).  Every entry point is `extern "C"`, takes plain pointers
 * and sizes only, returns an int status (0 = DSM_OK) and never aborts.  The
 * reference interfaces each entry point replaces are cited as
 * `/root/reference/<file>:<line>`.
 *
 * Vocabulary follows the reference: images, features (keypoints + 128-D uint8
 * SIFT descriptors), image pairs, FeatureMatches, TwoViewGeometry.
 */
#ifndef DAGSFM_MI355X_H_
#define DAGSFM_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
enum {
  DSM_OK = 0,
  DSM_ERR_INVALID_ARGUMENT = 1, /* reference: CHECK(...) fatal in matching.cc */
  DSM_ERR_NO_DEVICE = 2,        /* reference: Setup() returns false, matching.cc:732-742 */
  DSM_ERR_HIP = 3,              /* a HIP runtime call failed; see dsm_last_error */
  DSM_ERR_OUT_OF_RANGE = 4,
  DSM_ERR_NOT_READY = 5         /* results requested before the producing call */
};

/* ------------------------------------------------------------------ options */

/* Mirrors the matching half of SiftMatchingOptions, src/feature/sift.h:116-165.
 * max_ratio / max_distance are doubles there and are narrowed to float exactly
 * where the reference narrows them (FindBestMatches signature, sift.cc:164-166). */
typedef struct dsm_match_options {
  double max_ratio;        /* default 0.8  */
  double max_distance;     /* default 0.7  */
  int32_t cross_check;     /* default 1    */
  int32_t max_num_matches; /* default 32768; carried for ABI parity only: MatchSiftFeaturesCPU (sift.cc:810-822)
                              ignores it (the SiftGPU matcher alone clamps, sift.cc:200-209), and so does this library */
} dsm_match_options;

/* Mirrors TwoViewGeometry::Options (src/estimators/two_view_geometry.h:105-157)
 * with its embedded RANSACOptions (src/optim/ransac.h:47-72), filled from
 * SiftMatchingOptions exactly as TwoViewGeometryVerifier's ctor does
 * (src/feature/matching.cc:559-568). */
typedef struct dsm_two_view_options {
  uint64_t min_num_inliers;          /* 15   */
  double min_E_F_inlier_ratio;       /* 0.95 */
  double max_H_inlier_ratio;         /* 0.8  */
  double watermark_min_inlier_ratio; /* 0.7  */
  double watermark_border_size;      /* 0.1  */
  int32_t detect_watermark;          /* 1    */
  int32_t multiple_models;           /* 0; != 0: TwoViewGeometry::EstimateMultiple (two_view_geometry.cc:128-167) */
  /* RANSACOptions */
  double max_error;        /* 4.0   */
  double min_inlier_ratio; /* 0.25  */
  double confidence;       /* 0.999 */
  uint64_t min_num_trials; /* 30    */
  uint64_t max_num_trials; /* 10000 */
  int32_t multiple_ignore_watermark; /* 1 (two_view_geometry.h:140); only read when multiple_models != 0 */
  int32_t reserved;
} dsm_two_view_options;

/* Camera as used by the verification path (src/base/camera.h; models
 * src/base/camera_models.h:187-349).  model_id and the parameter order follow the reference:
 *   0 SIMPLE_PINHOLE f,cx,cy            1 PINHOLE fx,fy,cx,cy              2 SIMPLE_RADIAL f,cx,cy,k
 *   3 RADIAL f,cx,cy,k1,k2              4 OPENCV fx,fy,cx,cy,k1,k2,p1,p2   5 OPENCV_FISHEYE fx,fy,cx,cy,k1..k4
 *   6 FULL_OPENCV fx,fy,cx,cy,k1,k2,p1,p2,k3..k6   7 FOV fx,fy,cx,cy,omega
 *   8 SIMPLE_RADIAL_FISHEYE f,cx,cy,k   9 RADIAL_FISHEYE f,cx,cy,k1,k2
 *  10 THIN_PRISM_FISHEYE fx,fy,cx,cy,k1,k2,p1,p2,k3,k4,sx1,sy1
 * Any other model_id is rejected with DSM_ERR_INVALID_ARGUMENT (the reference CHECKs ExistsCameraModelWithId,
 * camera.cc:52).  Models 5, 7, 8, 9, 10 evaluate atan/tan/sin/cos with the device's math library. */
typedef struct dsm_camera {
  int32_t model_id;
  int32_t has_prior_focal_length; /* Camera::HasPriorFocalLength, camera.h:191 */
  uint64_t width;
  uint64_t height;
  double params[12];
} dsm_camera;

/* TwoViewGeometry::ConfigurationType, src/estimators/two_view_geometry.h:83-102 */
enum {
  DSM_CONFIG_UNDEFINED = 0,
  DSM_CONFIG_DEGENERATE = 1,
  DSM_CONFIG_CALIBRATED = 2,
  DSM_CONFIG_UNCALIBRATED = 3,
  DSM_CONFIG_PLANAR = 4,
  DSM_CONFIG_PANORAMIC = 5,
  DSM_CONFIG_PLANAR_OR_PANORAMIC = 6,
  DSM_CONFIG_WATERMARK = 7,
  DSM_CONFIG_MULTIPLE = 8
};

/* Fixed-size part of a TwoViewGeometry (two_view_geometry.h:286-303).  Matrices
 * are row-major 3x3.  The variable-length inlier_matches are fetched separately. */
typedef struct dsm_two_view_geometry {
  int32_t config;
  uint32_t num_inliers;  /* == inlier_matches.size() */
  uint32_t num_matches;  /* putative matches that went into Estimate */
  uint32_t reserved;
  double F[9];
  double E[9];
  double H[9];
  double qvec[4];
  double tvec[3];
  double tri_angle;
  /* bookkeeping for the hypotheses/s metric: LO-RANSAC trials and
   * models scored (minimal-sample models + local-optimisation models). */
  uint32_t num_trials[4];  /* E, F, H, watermark-translation */
  uint32_t num_models[4];
} dsm_two_view_geometry;

/* ------------------------------------------------------------------ context */
typedef struct dsm_ctx dsm_ctx;

/* Number of HIP devices visible to this process (0 when there is none).  SiftMatchingOptions::gpu_index "-1" means
 * "all of them", one matcher per device (src/feature/matching.cc:631-645, doc/faq.rst:322-331). */
int dsm_device_count(void);

/* Creates a context bound to HIP device `device`.  Replaces SiftFeatureMatcher's
 * ctor + Setup() (src/feature/matching.cc:610-675, 713-747): returns
 * DSM_ERR_NO_DEVICE where Setup() would return false. */
int dsm_ctx_create(int device, dsm_ctx** out_ctx);
void dsm_ctx_destroy(dsm_ctx* ctx);
/* The HIP device the context is bound to (-1 for NULL): what a companion library needs to place its own buffers and
 * communicators next to the context (include/dagsfm_gather.h). */
int dsm_ctx_device(const dsm_ctx* ctx);
/* Number of pairs of the context's last dsm_match_pairs / dsm_set_matches (0 before any). */
uint32_t dsm_ctx_num_pairs(const dsm_ctx* ctx);
/* Last error text for this context (or for ctx creation when ctx == NULL). */
const char* dsm_last_error(const dsm_ctx* ctx);
/* Blocks until all work queued by this context has finished. */
int dsm_sync(dsm_ctx* ctx);
/* Scheduling / cross-check switches of one context, for tests and profiling: `key` is one of the names DESIGN.md lists
 * under "Tuning / debugging hooks" (e.g. "DSM_VERIFY_LANES"), `value` its setting; value == NULL removes the key.  None
 * of them changes a result (tools/check_schedules.py).  The library never reads the process environment, so a host
 * application's environment cannot change schedules; the reference has no analogue (its options all travel in
 * SiftMatchingOptions, src/feature/sift.h:139-195).  DSM_ERR_INVALID_ARGUMENT for an unknown key. */
int dsm_set_debug_option(dsm_ctx* ctx, const char* key, const char* value);

/* A memory budget at the boundary.  The reference documents its GPU matcher's footprint and leaves the rest of the device to the
 * host application (doc/faq.rst:353-356: "4 n^2 + 4 n 256 bytes"; the mapper's dense stages run on the same GPU next).  This
 * library's stages cut their pair list into chunks sized by the memory they may use for TRANSIENT scratch -- the matcher's
 * per-row outputs of K1 (default: 8 GiB per chunk), the verifier's speculated trials of a chunk of pairs (default: 40 % of what
 * is free at the call, at least 4 GiB, at most 96 GiB) -- and `bytes` > 0 replaces both defaults: the two stages together then
 * hold at most `bytes` of chunk scratch (a quarter of it, at most 8 GiB, for the matcher; the rest for the verifier's lanes; both
 * keep their scratch between calls).  Smaller chunks cost time, never results
 * (tests/test_memory_budget_gpu.py; profiles/r06_memory_budget.txt has the step time of config 2 at 8 / 16 / 38 GiB).  The
 * RESIDENT set -- descriptors, keypoints, matches, per-pair state and results of the current list -- follows from the inputs and
 * is not chunked; dsm_ctx_memory_footprint reports both.  Scratch already held beyond a new, smaller budget is released at once.
 * bytes == 0 restores the defaults.  A budget too small for one pair's scratch still runs (a chunk is never shorter than one pair). */
int dsm_ctx_set_memory_budget(dsm_ctx* ctx, uint64_t bytes);
/* Device memory the context holds right now: *resident_bytes = images + the last calls' inputs / results / per-pair state,
 * *scratch_bytes = the chunk scratch the budget governs.  Either pointer may be NULL. */
int dsm_ctx_memory_footprint(const dsm_ctx* ctx, uint64_t* resident_bytes, uint64_t* scratch_bytes);

/* What the device reports about itself (hipDeviceProp_t): used by bench.py to derive the roofline peaks from
 * the hardware instead of hard-coding them. */
typedef struct dsm_device_info {
  char name[128];
  char arch[64];            /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
  int32_t compute_units;    /* multiProcessorCount */
  int32_t clock_khz;        /* clockRate: peak engine clock */
  int32_t memory_clock_khz; /* memoryClockRate */
  int32_t memory_bus_bits;  /* memoryBusWidth */
  uint64_t total_memory;    /* totalGlobalMem, bytes */
  int32_t l2_bytes;         /* l2CacheSize (one XCD's L2) */
  int32_t lds_per_cu;       /* maxSharedMemoryPerMultiProcessor */
} dsm_device_info;
int dsm_get_device_info(dsm_ctx* ctx, dsm_device_info* out);

/* Makes `n_images` images resident in HBM.  Replaces FeatureMatcherCache::Setup /
 * GetDescriptors / GetKeypoints (src/feature/matching.cc:221-316) and
 * SiftMatchGPU::SetDescriptors (lib/SiftGPU/SiftGPU.h:303-336).
 *   n_feats[i]      number of features of image i (may be 0)
 *   desc[i]         n_feats[i] x 128 uint8, row-major (FeatureDescriptors, types.h:102)
 *   kp_xy[i]        n_feats[i] x 2 float (x,y) with stride kp_stride floats between
 *                   keypoints (6 for FeatureKeypoint, types.h:44-81); may be NULL when
 *                   only matching is wanted
 *   cameras[i]      camera of image i; may be NULL when only matching is wanted
 * Pointers are host pointers, borrowed for the duration of the call.  Pageable memory (the reference's Eigen
 * matrices and std::vectors) is read by a few host threads of the library's own into pinned staging slots
 * (32 MB per context, allocated by the first such call); memory that is pinned already is copied where it lies. */
int dsm_set_images(dsm_ctx* ctx, uint32_t n_images, const uint32_t* n_feats,
                   const uint8_t* const* desc, const float* const* kp_xy,
                   uint32_t kp_stride, const dsm_camera* cameras);

/* Adds `n_images` images to the resident set without touching the ones already there: the new images
 * get the indices n_resident .. n_resident + n_images - 1 and only their rows are uploaded.  This is
 * FeatureMatcherCache's incremental behaviour (an LRU that loads what a block of the pair list adds,
 * src/feature/matching.cc:245-316) for the device-side copy: ExhaustiveFeatureMatcher visits the
 * database in blocks (matching.cc:870-905) and consecutive blocks share half of their images.
 * Arguments as dsm_set_images; keypoints / cameras must be given iff the resident images have them.
 * Invalidates the results of earlier dsm_match_pairs / dsm_verify_pairs calls, like dsm_set_images. */
int dsm_append_images(dsm_ctx* ctx, uint32_t n_images, const uint32_t* n_feats,
                      const uint8_t* const* desc, const float* const* kp_xy,
                      uint32_t kp_stride, const dsm_camera* cameras);

/* Brute-force matches every listed image pair on the device.  Replaces the
 * matcher stage of SiftFeatureMatcher::Match (src/feature/matching.cc:749-839)
 * = MatchSiftFeaturesCPU per pair (src/feature/sift.cc:810-822).
 *   pairs   n_pairs x 2 image indices (into the dsm_set_images order)
 * Results stay in HBM until fetched or consumed by dsm_verify_pairs. */
int dsm_match_pairs(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* pairs,
                    const dsm_match_options* options);

/* Installs caller-provided matches (host pointers) as if dsm_match_pairs had produced them, so
 * that dsm_verify_pairs can verify them.  Used for SiftFeatureMatcher::Match's resume path: a
 * pair whose `matches` row exists but whose `two_view_geometries` row does not is only
 * re-verified (src/feature/matching.cc:782-812).  offsets has n_pairs+1 entries. */
int dsm_set_matches(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* pairs, const uint64_t* offsets,
                    const uint32_t* matches);

/* Per-pair number of matches of the last dsm_match_pairs; `counts` has n_pairs
 * entries (host or device pointer). */
int dsm_get_match_counts(dsm_ctx* ctx, uint32_t* counts);
/* All matches of the last dsm_match_pairs, pair after pair in list order:
 * `offsets` gets n_pairs+1 prefix offsets (in matches), `matches` gets
 * offsets[n_pairs] x 2 uint32 (point2D_idx1, point2D_idx2) in ascending idx1
 * per pair (FeatureMatches, types.h:86-104).  Either pointer may be NULL.
 * `matches_capacity` is in matches (pairs of uint32). */
int dsm_get_matches(dsm_ctx* ctx, uint64_t* offsets, uint32_t* matches,
                    uint64_t matches_capacity);

/* One-shot leaf with the signature shape of MatchSiftFeaturesCPU
 * (src/feature/sift.h:214-217) / MatchSiftFeaturesGPU (sift.h:229-239):
 * host descriptors in, FeatureMatches out.  `matches` must hold
 * min(n1,n2) x 2 uint32 with cross_check, n1 x 2 uint32 without. */
int dsm_match_sift_features(dsm_ctx* ctx, const dsm_match_options* options,
                            const uint8_t* desc1, uint32_t n1,
                            const uint8_t* desc2, uint32_t n2,
                            uint32_t* matches, uint32_t* n_matches);

/* Device timing of the dominant kernel (descriptor distance + fused top-2) of the
 * last dsm_match_pairs, measured with HIP events on the stream the kernel was
 * launched on: total milliseconds and number of launches. */
int dsm_get_match_kernel_time(dsm_ctx* ctx, double* total_ms, uint32_t* n_launches);
/* Same for the small follow-up kernel that turns the best tile of every accepted row into
 * the exact column index (k1_resolve_index). */
int dsm_get_match_resolve_time(dsm_ctx* ctx, double* total_ms);
/* Same for the gathered second pass of the cross-check (k1_best_rows over the rows matches12 points at). */
int dsm_get_match_gather_time(dsm_ctx* ctx, double* total_ms);
/* Same for everything else on the stream: the entry list of pass 2 (k2 + scan between pass 1's resolve and pass 2), pass 2's
 * k1_resolve_index and the compaction of the mutual matches (k2_entries + scan, the host's waits for the totals included) --
 * without the cross-check: the compaction of the one-way matches -- so that the four match timers sum to the call in both modes. */
int dsm_get_match_tail_time(dsm_ctx* ctx, double* total_ms);

/* ------------------------------------------------------------------ verification */

/* Per-pair PRNG seed used when dsm_verify_pairs gets no explicit seeds: a 32-bit mix of
 * Database::ImagePairToPairId(id1, id2) (src/base/database.h:336-347) xor user_seed.  The
 * reference seeds each verifier thread from the wall clock (src/util/random.cc:40-56) and is
 * not reproducible; one MT19937 stream per pair, consumed E -> F -> H -> watermark in the order of
 * src/estimators/two_view_geometry.cc:325-342, 547-549, is the defined schedule here. */
uint32_t dsm_pair_seed(uint32_t image_id1, uint32_t image_id2, uint32_t user_seed);

/* Geometric verification of every pair of the last dsm_match_pairs on the device.  Replaces
 * the TwoViewGeometryVerifier stage of SiftFeatureMatcher::Match (src/feature/matching.cc:
 * 550-608, 749-839) = TwoViewGeometry::Estimate per pair (two_view_geometry.cc:113-126).
 *   seeds         n_pairs explicit PRNG seeds, or NULL to use dsm_pair_seed(idx1, idx2, user_seed)
 *   stage_filter  non-zero: pairs with fewer than min_num_inliers inliers get a default
 *                 TwoViewGeometry(), as Match() writes them (matching.cc:824-831)
 * options->multiple_models != 0 runs TwoViewGeometry::EstimateMultiple (two_view_geometry.cc:128-167)
 * instead, as the verifier does (matching.cc:596-599): repeated Estimate passes over the matches that are
 * not inliers yet, on ONE generator stream per pair; several geometries -> config MULTIPLE with the inlier
 * matches of all of them (other fields as in a fresh TwoViewGeometry()); num_trials / num_models are summed
 * over the passes. */
int dsm_verify_pairs(dsm_ctx* ctx, const dsm_two_view_options* options, const uint32_t* seeds,
                     uint32_t user_seed, int32_t stage_filter);
/* Results of the last dsm_verify_pairs: n_pairs fixed-size records ... */
/* Guided matching (SiftMatchingOptions::guided_matching, src/feature/sift.h:162) over the pairs of the last
 * dsm_verify_pairs: every pair with at least min_num_inliers inlier matches (matching.cc:449-453) and an F-type
 * (CALIBRATED, UNCALIBRATED) or H-type (PLANAR, PANORAMIC, PLANAR_OR_PANORAMIC) configuration is matched again
 * with the descriptor distance of keypoint pairs that violate F / H (float Sampson / transfer error above
 * max_error^2) set to zero, and the result REPLACES its inlier matches -- MatchGuidedSiftFeaturesCPU
 * (src/feature/sift.cc:824-875), GuidedSiftCPUFeatureMatcher::Run (matching.cc:441-470); config, E, F, H and
 * the pose stay.  Call dsm_verify_pairs with stage_filter = 0 first; stage_filter here is Match()'s
 * post-filter (matching.cc:824-831) on the final inlier counts. */
int dsm_guided_match_pairs(dsm_ctx* ctx, const dsm_match_options* match_options,
                           const dsm_two_view_options* options, int32_t stage_filter);

int dsm_get_two_view_geometries(dsm_ctx* ctx, dsm_two_view_geometry* out);
/* ... and the inlier_matches of all pairs in list order (same conventions as dsm_get_matches). */
int dsm_get_inlier_matches(dsm_ctx* ctx, uint64_t* offsets, uint32_t* inlier_matches,
                           uint64_t capacity);
/* Device time (HIP events on the launch stream) of the verification kernel of the last call. */
int dsm_get_verify_kernel_time(dsm_ctx* ctx, double* total_ms);

/* One-shot leaf with the signature shape of TwoViewGeometry::Estimate
 * (src/estimators/two_view_geometry.h:180-184): host cameras, points (n x 2 doubles, as
 * FeatureKeypointsToPointsVector makes them) and matches in, TwoViewGeometry out.
 * `inlier_matches` must hold n_matches x 2 uint32 (may be NULL). */
int dsm_estimate_two_view_geometry(dsm_ctx* ctx, const dsm_camera* camera1, const double* points1,
                                   uint32_t n1, const dsm_camera* camera2, const double* points2,
                                   uint32_t n2, const uint32_t* matches, uint32_t n_matches,
                                   const dsm_two_view_options* options, uint32_t seed,
                                   dsm_two_view_geometry* out, uint32_t* inlier_matches);

/* Test hook: the first n_draws samples (k indices each) the device sampler draws from
 * `total` items with the given seed (RandomSampler, src/optim/random_sampler.cc:43-62). */
int dsm_debug_sample_sequence(dsm_ctx* ctx, uint32_t seed, uint32_t k, uint32_t total,
                              uint32_t n_draws, uint32_t* out);

/* Test hook: the statistics counters of the last dsm_verify_pairs call that ran with the debug option DSM_VERIFY_DEBUG or
 * DSM_SCORE_PREFILTER=check (16 uint32, summed over the lanes): [1..6] candidates / local optimisations per family, [14]
 * (model, pair) slots whose exact inlier count fell outside the bounds of the scoring's bound step -- must be 0 --, [15] slots
 * the bound step would have skipped. */
int dsm_debug_verify_counters(dsm_ctx* ctx, uint32_t* out16);

/* Test hook: Camera::ImageToWorld (src/base/camera.cc:210-214) of n pixel points (x, y) on the device. */
int dsm_debug_image_to_world(dsm_ctx* ctx, const dsm_camera* camera, uint32_t n, const double* xy, double* out_uv);

/* ------------------------------------------------------------------ vocabulary-tree retrieval (candidate pairs)
 * The step BEFORE matching: VocabSimilarityGraph::Run (src/graph/similarity_graph.cpp:101-199)
 * indexes every image in a retrieval::VisualIndex (src/retrieval/visual_index.h) and queries every image against it;
 * an image and each image retrieved for it become a candidate pair.  Works on the images of dsm_set_images
 * (descriptors only).  Deviations from the reference, all in DESIGN.md: the nearest visual words are EXACT (the
 * reference asks FLANN for approximate ones), float sums run left to right, ties keep first-seen order. */
typedef struct dsm_vocabulary {
  uint32_t num_words;      /* visual words (leaves of the vocabulary tree), VisualIndex::NumVisualWords */
  uint32_t reserved;
  const uint8_t* words;    /* [num_words][128] uint8 centroids, visual_words_ (visual_index.h:176-178) */
  const float* projection; /* [64][128] row-major Hamming-embedding projection, InvertedIndex::proj_matrix_ */
  const float* thresholds; /* [num_words][64] per-word embedding thresholds, InvertedFile::thresholds_ */
} dsm_vocabulary;
int dsm_retrieval_set_vocabulary(dsm_ctx* ctx, const dsm_vocabulary* vocabulary);
/* The reference's OWN word ids instead of the device's exact nearest words: VisualIndex::FindWordIds (visual_index.h:
 * 695-738) asks the flann::AutotunedIndex loaded from the vocabulary file for APPROXIMATE neighbours, once with 1
 * neighbour when a feature is indexed (VisualIndex::Add, :201-243) and once with QueryOptions::num_neighbors when it is
 * queried (:664-693).  The host shim restates that search (dagsfm_amd/host/flann_index.cc, bit for bit against the
 * reference's FLANN) and hands the ids over here: index_ids [features] and query_ids [features][k_query] for the
 * features of all resident images back to back, kInvalidWordId (INT_MAX) where FLANN returned fewer.  Later
 * dsm_retrieval_index / _query / _matches use them (num_neighbors must equal k_query); both NULL: exact search again. */
int dsm_retrieval_set_word_ids(dsm_ctx* ctx, const int32_t* index_ids, uint32_t k_query, const int32_t* query_ids);
/* The reference's word search ON THE DEVICE (round 5): the FLANN index the vocabulary file carries -- what
 * flann::AutotunedIndex::loadIndex reads (visual_index.h:564-574) -- as flat arrays, searched by a lane per feature with
 * FLANN's own visit order, branch heap and result set (csrc/flann_search.hip <- lib/FLANN/algorithms/kdtree_index.h:
 * 543-617, kmeans_index.h:717-833, linear_index.h:130-146; ids and float distances equal the reference's knnSearch bit
 * for bit).  The host shim parses the file (dagsfm_amd/host/flann_index.cc) and hands the trees over here; afterwards
 * dsm_retrieval_index searches with 1 neighbour and dsm_retrieval_query / _matches with num_neighbors, both with
 * `num_checks` (IndexOptions / QueryOptions::num_checks).  Every node, child and point index is validated on upload
 * (DSM_ERR_OUT_OF_RANGE).  NULL: the device's exact search again.  Word ids set with dsm_retrieval_set_word_ids win. */
typedef struct dsm_flann_kd_node {
  int32_t divfeat;        /* inner node: split dimension (0..127); leaf: the word's index */
  float divval;
  int32_t child1, child2; /* node indices, greater than the node's own; -1 / -1 marks a leaf */
} dsm_flann_kd_node;
typedef struct dsm_flann_km_node {
  uint64_t pivot;         /* offset of the node's centre in `pivots`, in floats (a multiple of 128) */
  float radius, variance;
  int32_t size;           /* leaf: number of points */
  uint32_t first_child;   /* inner node: `branching` entries of km_childs from here */
  uint32_t num_childs;    /* 0 for a leaf, else == branching */
  uint32_t reserved;
  uint64_t first_point;   /* leaf: `size` entries of km_points from here */
} dsm_flann_km_node;
typedef struct dsm_flann_index {
  int32_t algorithm;      /* flann_algorithm_t: 0 linear, 1 randomised kd-trees, 2 hierarchical k-means */
  int32_t num_checks;     /* SearchParams::checks, >= 0 on a tree index */
  uint32_t num_words;     /* must equal the vocabulary's */
  int32_t branching;      /* k-means */
  float cb_index;         /* k-means */
  int32_t km_root;        /* k-means: index of the root node */
  uint32_t n_kd_nodes, n_kd_roots;
  const dsm_flann_kd_node* kd_nodes;
  const int32_t* kd_roots;
  uint32_t n_km_nodes, reserved;
  const dsm_flann_km_node* km_nodes;
  uint64_t n_km_childs;
  const int32_t* km_childs;
  uint64_t n_km_points;
  const uint64_t* km_points;
  uint64_t n_pivot_floats;
  const float* pivots;
} dsm_flann_index;
int dsm_retrieval_set_flann_index(dsm_ctx* ctx, const dsm_flann_index* index);
/* The same search for caller-supplied descriptors (n x 128 uint8, host memory): ids [n][k] (INT_MAX where FLANN returned
 * fewer) and, if not NULL, FLANN's squared L2 distances [n][k]; 1 <= k <= 8.  What the parity tests and
 * tools/bench_retrieval.py call; *ms (may be NULL) receives the kernel's device time. */
int dsm_retrieval_flann_search(dsm_ctx* ctx, const uint8_t* descriptors, uint32_t n, uint32_t k, int32_t* ids, float* dists,
                               double* ms);
/* VisualIndex::Add (IndexOptions::num_neighbors = 1) for every resident image in list order, then Prepare()
 * (visual_index.h:201-243, 501-505): inverted files sorted by image, IDF weights, normalisation constants. */
int dsm_retrieval_index(dsm_ctx* ctx);
/* VisualIndex::Query (num_images_after_verification = 0, visual_index.h:664-693) for every resident image:
 * counts[q] image scores (<= max_num_images) for query image q, image_idx / scores at [q * max_num_images + k] in
 * retrieval order (descending score).  The query image itself is among its results, as in the reference.
 * num_neighbors: QueryOptions::num_neighbors (VocabSimilaritySearchOptions::num_nearest_neighbors, default 5, max 8). */
int dsm_retrieval_query(dsm_ctx* ctx, uint32_t num_neighbors, uint32_t max_num_images, uint32_t* counts,
                        uint32_t* image_idx, float* scores);
/* The input of the spatial re-ranking (VisualIndex::Query with geometries, visual_index.h:295-346; QueryOptions::
 * num_images_after_verification > 0): for every resident image as the query and its retrieved images (counts / image_idx as
 * dsm_retrieval_query returned them, same max_num_images), the database features that fall into one of the query feature's
 * words, belong to a retrieved image and lie within HammingDistWeightFunctor::kMaxHammingDistance (InvertedIndex::
 * FindMatches + the Hamming test).  offsets[q] .. offsets[q + 1] (n_images + 1 entries) index the tuples of query q;
 * dsm_get_retrieval_matches copies them: 5 uint32 each = query feature, image, database feature, (word << 8) | Hamming
 * distance, position of the entry in the inverted files (the order the reference's pointer comparison has inside one
 * file), in (query feature, neighbour, entry) order.  The 1-to-1 assignment and VoteAndVerify (vote_and_verify.cc) run on
 * the host (dagsfm_amd/host/spatial_verification.cc): they are sequential per image and use the host's float libm. */
int dsm_retrieval_matches(dsm_ctx* ctx, uint32_t num_neighbors, uint32_t max_num_images, const uint32_t* counts,
                          const uint32_t* image_idx, uint64_t* offsets);
int dsm_get_retrieval_matches(dsm_ctx* ctx, uint32_t* tuples, uint64_t capacity);
/* InvertedFile::IDFWeight of every visual word (inverted_file.h:260-271) after dsm_retrieval_index. */
int dsm_get_retrieval_idf(dsm_ctx* ctx, float* idf, uint32_t capacity);
/* Test hook: the k nearest visual words (ascending distance, ties to the lower id) of every feature of one image. */
int dsm_retrieval_debug_word_ids(dsm_ctx* ctx, uint32_t image, uint32_t k, int32_t* out);
/* Device time (HIP events) of the last dsm_retrieval_index / dsm_retrieval_query. */
int dsm_get_retrieval_time(dsm_ctx* ctx, double* index_ms, double* query_ms);

/* ------------------------------------------------------------------ view-graph ingest + rotation-cycle filter
 * The step AFTER the stage: DistributedMapperController::LoadTwoviewGeometries
 * (src/controllers/distributed_mapper_controller.cpp:585-631) turns every two_view_geometries row into a view-graph
 * edge (rotation = the row's qvec), ViewGraph::FilterViewGraphCyclesByRotation(5.0) (src/graph/view_graph.cpp:115-165)
 * keeps an edge iff it lies on a cycle of length 3 whose loop rotation R23 * R12 * R13^T is below the threshold.
 *   pairs  n_pairs x 2 image ids (any ids; a repeat of an earlier pair is ignored: keep = 0)
 *   qvecs  n_pairs x 4 (w, x, y, z): the relative rotation of the pair as stored, i.e. for image_id1 < image_id2
 *   keep   n_pairs flags out;  n_triplets (optional): cycles of length 3 found
 * Host pointers. */
int dsm_view_graph_filter_cycles(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* pairs, const double* qvecs,
                                 double max_loop_error_degrees, uint8_t* keep, uint64_t* n_triplets);

void dsm_default_match_options(dsm_match_options* o);
void dsm_default_two_view_options(dsm_two_view_options* o);

#ifdef __cplusplus
}
#endif
#endif /* DAGSFM_MI355X_H_ */
