// spatial_verification.h -- spatial re-ranking of the retrieved images (QueryOptions::num_images_after_verification > 0):
// the host half of VisualIndex::Query with geometries (/root/reference/src/retrieval/visual_index.h:259-500) and
// VoteAndVerify (/root/reference/src/retrieval/vote_and_verify.cc:208-418).  The device (csrc/retrieval.hip,
// dsm_retrieval_matches) delivers, per query image, the database features that share a visual word with a query feature
// within the Hamming threshold -- the data-parallel part; the 1-to-1 assignment and the vote-and-verify of one retrieved
// image are sequential and use the host's float libm (sinf / cosf / log2f), so they run here.
#ifndef DAGSFM_AMD_HOST_SPATIAL_VERIFICATION_H_
#define DAGSFM_AMD_HOST_SPATIAL_VERIFICATION_H_
#include <cstddef>
#include <cstdint>
#include <vector>

#include "types.h"

namespace dagsfm_amd {

struct FeatureGeometry {  // retrieval/geometry.h:49-67
  float x = 0.0f, y = 0.0f, scale = 0.0f, orientation = 0.0f;
};
// visual_index.h:229-233 / 303-307: x, y, FeatureKeypoint::ComputeScale(), ComputeOrientation() (feature/types.cc:84-98)
FeatureGeometry GeometryOfKeypoint(const FeatureKeypoint& keypoint);

struct GeometryMatch {  // FeatureGeometryMatch with its single geometries2 entry (geometry.h:70-73)
  FeatureGeometry query, database;
};

struct VoteAndVerifyOptions {  // retrieval/vote_and_verify.h:42-68
  int num_transformations = 30;
  int num_trans_bins = 64;
  int num_scale_bins = 32;
  int num_angle_bins = 8;
  int max_image_size = 4096;
  int min_num_votes = 1;
  double confidence = 0.99;
  double max_transfer_error = 100.0 * 100.0;
  double max_scale_error = 2.0;
};

// Effective inlier count of the best similarity / affine transformation between the matched features.
int VoteAndVerify(const VoteAndVerifyOptions& options, const std::vector<GeometryMatch>& matches);

// AffineTransformEstimator::Estimate (estimators/affine_transform.cc:40-75): least-squares A (row-major 2 x 3) with
// dst ~ A * (src, 1) over n >= 3 points (xy interleaved).
void EstimateAffineTransform(const double* src, const double* dst, size_t n, double A[6]);

// One candidate correspondence of a query image, as dsm_get_retrieval_matches delivers it plus what the host looks up.
struct RetrievalCandidate {
  uint32_t query_feature = 0;
  uint32_t image = 0;             // index of the database image (the context's image order)
  uint32_t database_feature = 0;
  uint32_t entry_position = 0;    // position of the entry in the inverted files: the tie order of equal weights
  float weight = 0.0f;            // HammingDistWeightFunctor(distance) * idf^2 (visual_index.h:328-329)
  FeatureGeometry database_geometry;
};

// Verifies the retrieved images of ONE query and re-ranks them (visual_index.h:366-500): image_idx / scores hold `count`
// retrieved images in retrieval order and come back re-ranked; returns the new count
// (min(count, num_images_after_verification)).  candidates: in any order.
uint32_t SpatialRerank(const std::vector<FeatureGeometry>& query_geometries, const std::vector<RetrievalCandidate>& candidates,
                       int num_images_after_verification, uint32_t count, uint32_t* image_idx, float* scores);

}  // namespace dagsfm_amd
#endif
