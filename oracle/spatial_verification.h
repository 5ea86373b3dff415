// spatial_verification.h -- CPU restatement of the spatial re-ranking of the reference's vocabulary-tree retrieval:
//   FeatureGeometry::TransformFromMatch / GetArea / GetAreaUnderTransform   /root/reference/src/retrieval/geometry.cc:37-86
//   AffineTransformEstimator::Estimate                                      src/estimators/affine_transform.cc:40-75
//   VoteAndVerify (+ TwoWayTransform, VotingBin, ComputeScaleError, ComputeTransferError, ComputeInliers,
//   ComputeEffectiveInlierCount)                                            src/retrieval/vote_and_verify.cc:46-418
//
// TEST INFRASTRUCTURE ONLY (see oracle/Makefile): the checker of dagsfm_amd/host/spatial_verification.cc.
//
// PARITY UNPINNED for VoteAndVerify: the reference holds no test vector for it (there is no vote_and_verify_test.cc),
// Eigen is absent here, and its result depends on things the C++ standard leaves open.  What stands in their place:
//   * the reference walks `std::unordered_map` bins and `std::partial_sort`s them by score: bins of equal score come
//     out in an order that depends on the hash table's history and on the library.  Here: the same containers fed the
//     same sequence of insertions, i.e. the order of THIS toolchain's libstdc++ (GCC 11) -- like the PRNG mapping of
//     std::uniform_int_distribution in the two-view oracle.  (An order of our own -- equal scores by ascending bin index --
//     changes the result of 26 of 3 000 random scenes, mostly by one or two effective inliers.)
//   * Eigen's fixed-size float products (Matrix2f * Vector2f, A^T * M * A) are written out coefficient by coefficient,
//     sums left to right; the least-squares solve (JacobiSVD of the 2N x 6 system, ColPivHouseholderQR preconditioner,
//     solve() = V * S^-1 * U^T b over the numerical rank) uses oracle/linalg.h, dot products left to right.
//   * a float -> int conversion of NaN / out-of-range values (ComputeEffectiveInlierCount with a single inlier:
//     0 * inf) is undefined behaviour in C++; here it is what x86-64 cvttss2si returns (INT_MIN).
// The leaf functions are pinned to the reference's own tests: geometry_test.cc (identity / translation / scale /
// orientation) and affine_transform_test.cc (tests/test_retrieval.py).
#pragma once
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

#include "linalg.h"

namespace oracle_sv {

struct FeatureGeometry {  // retrieval/geometry.h:49-67
  float x = 0.0f, y = 0.0f, scale = 0.0f, orientation = 0.0f;
};
struct FeatureGeometryTransform {  // geometry.h:42-47
  float scale = 0.0f, angle = 0.0f, tx = 0.0f, ty = 0.0f;
};
struct FeatureGeometryMatch {  // geometry.h:70-73 (the retrieval only ever produces 1-to-1 matches)
  FeatureGeometry geometry1;
  std::vector<FeatureGeometry> geometries2;
};

inline int FloatToInt(float v) {  // static_cast<int>(float) as x86-64 executes it (cvttss2si)
  if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT_MIN;
  return static_cast<int>(v);
}

// FeatureGeometry::TransformFromMatch, geometry.cc:37-57
inline FeatureGeometryTransform TransformFromMatch(const FeatureGeometry& feature1, const FeatureGeometry& feature2) {
  FeatureGeometryTransform tform;
  tform.scale = feature2.scale / feature1.scale;
  tform.angle = feature2.orientation - feature1.orientation;
  const float sin_angle = std::sin(tform.angle);
  const float cos_angle = std::cos(tform.angle);
  // t = (x2, y2) - scale * R * (x1, y1): (scale * R) first, then the product with the vector
  const float r00 = tform.scale * cos_angle, r01 = tform.scale * -sin_angle;
  const float r10 = tform.scale * sin_angle, r11 = tform.scale * cos_angle;
  tform.tx = feature2.x - (r00 * feature1.x + r01 * feature1.y);
  tform.ty = feature2.y - (r10 * feature1.x + r11 * feature1.y);
  return tform;
}

// FeatureGeometry::GetArea, geometry.cc:77-79
inline float GetArea(const FeatureGeometry& f) { return 1.0f / std::sqrt(4.0f / (f.scale * f.scale * f.scale * f.scale)); }

// FeatureGeometry::GetAreaUnderTransform, geometry.cc:81-86.  A: row-major 2 x 2.
inline float GetAreaUnderTransform(const FeatureGeometry& f, const float A[4]) {
  const float m = 1.0f / (f.scale * f.scale);  // M = Identity / (scale * scale)
  const float zero = 0.0f / (f.scale * f.scale);
  // T = A^T * M, N = T * A
  const float T00 = A[0] * m + A[2] * zero, T01 = A[0] * zero + A[2] * m;
  const float T10 = A[1] * m + A[3] * zero, T11 = A[1] * zero + A[3] * m;
  const float N00 = T00 * A[0] + T01 * A[2], N01 = T00 * A[1] + T01 * A[3];
  const float N10 = T10 * A[0] + T11 * A[2], N11 = T10 * A[1] + T11 * A[3];
  const float B = N10 + N01;
  return 1.0f / std::sqrt(4.0f * N00 * N11 - B * B);
}

// TwoWayTransform, vote_and_verify.cc:46-71 (matrices row-major 2 x 2)
struct TwoWayTransform {
  float A12[4] = {0, 0, 0, 0}, t12[2] = {0, 0}, A21[4] = {0, 0, 0, 0}, t21[2] = {0, 0};
  TwoWayTransform() {}
  explicit TwoWayTransform(const FeatureGeometryTransform& tform) {
    const float sin_angle = std::sin(tform.angle);
    const float cos_angle = std::cos(tform.angle);
    const float R[4] = {cos_angle, -sin_angle, sin_angle, cos_angle};
    for (int k = 0; k < 4; ++k) A12[k] = tform.scale * R[k];
    t12[0] = tform.tx;
    t12[1] = tform.ty;
    const float Rt[4] = {R[0], R[2], R[1], R[3]};
    for (int k = 0; k < 4; ++k) A21[k] = Rt[k] / tform.scale;
    t21[0] = (-A21[0]) * t12[0] + (-A21[1]) * t12[1];
    t21[1] = (-A21[2]) * t12[0] + (-A21[3]) * t12[1];
  }
};

// vote_and_verify.cc:105-116
inline float ComputeScaleError(const FeatureGeometry& feature1, const FeatureGeometry& feature2, const TwoWayTransform& tform) {
  const float area_transformed = GetAreaUnderTransform(feature1, tform.A21);
  const float area_measured = GetArea(feature2);
  if (area_transformed > area_measured) return area_transformed / area_measured;
  return area_measured / area_transformed;
}
// vote_and_verify.cc:119-127
inline float ComputeTransferError(const FeatureGeometry& feature1, const FeatureGeometry& feature2, const TwoWayTransform& tform) {
  const float e1x = (feature2.x - (tform.A12[0] * feature1.x + tform.A12[1] * feature1.y)) - tform.t12[0];
  const float e1y = (feature2.y - (tform.A12[2] * feature1.x + tform.A12[3] * feature1.y)) - tform.t12[1];
  const float e2x = (feature1.x - (tform.A21[0] * feature2.x + tform.A21[1] * feature2.y)) - tform.t21[0];
  const float e2y = (feature1.y - (tform.A21[2] * feature2.x + tform.A21[3] * feature2.y)) - tform.t21[1];
  const float error1 = e1x * e1x + e1y * e1y;
  const float error2 = e2x * e2x + e2y * e2y;
  return error1 + error2;
}

// vote_and_verify.cc:130-150
inline void ComputeInliers(const TwoWayTransform& tform, const std::vector<FeatureGeometryMatch>& matches,
                           const float max_transfer_error, const float max_scale_error, std::vector<std::pair<int, int>>* inlier_idxs) {
  inlier_idxs->clear();
  for (size_t i = 0; i < matches.size(); ++i) {
    const auto& match = matches[i];
    for (size_t j = 0; j < match.geometries2.size(); ++j) {
      const auto& geometry2 = match.geometries2[j];
      if (ComputeScaleError(match.geometry1, geometry2, tform) <= max_scale_error &&
          ComputeTransferError(match.geometry1, geometry2, tform) <= max_transfer_error)
        inlier_idxs->emplace_back(static_cast<int>(i), static_cast<int>(j));
    }
  }
}

// vote_and_verify.cc:153-204
inline size_t ComputeEffectiveInlierCount(const TwoWayTransform& tform, const std::vector<FeatureGeometryMatch>& matches,
                                          const float max_transfer_error, const float max_scale_error, const int num_bins) {
  std::vector<std::pair<float, float>> inlier_coords;
  float min_x = std::numeric_limits<float>::max(), min_y = std::numeric_limits<float>::max();
  float max_x = 0, max_y = 0;
  for (const auto& match : matches) {
    for (const auto& geometry2 : match.geometries2) {
      if (ComputeScaleError(match.geometry1, geometry2, tform) <= max_scale_error &&
          ComputeTransferError(match.geometry1, geometry2, tform) <= max_transfer_error) {
        inlier_coords.emplace_back(match.geometry1.x, match.geometry1.y);
        min_x = std::min(min_x, match.geometry1.x);
        min_y = std::min(min_y, match.geometry1.y);
        max_x = std::max(max_x, match.geometry1.x);
        max_y = std::max(max_y, match.geometry1.y);
        break;
      }
    }
  }
  if (inlier_coords.empty()) return 0;
  const float scale_x = num_bins / (max_x - min_x);
  const float scale_y = num_bins / (max_y - min_y);
  std::vector<int> counter(static_cast<size_t>(num_bins) * num_bins, 0);
  for (const auto& coord : inlier_coords) {
    const int c_x = FloatToInt((coord.first - min_x) * scale_x);
    const int c_y = FloatToInt((coord.second - min_y) * scale_y);
    counter[static_cast<size_t>(std::max(0, std::min(num_bins - 1, c_x))) * num_bins + std::max(0, std::min(num_bins - 1, c_y))] = 1;
  }
  size_t sum = 0;
  for (int v : counter) sum += v;
  return sum;
}

// AffineTransformEstimator::Estimate, affine_transform.cc:40-75: A (row-major 2 x 3) with x2 ~ A * (x1, 1), least squares.
inline void EstimateAffine(const std::vector<double>& x1, const std::vector<double>& x2, double A[6]) {
  const int n = static_cast<int>(x1.size() / 2);
  oracle::Mat C(2 * n, 6);
  std::vector<double> b(2 * static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    C(2 * i, 0) = x1[2 * i];
    C(2 * i, 1) = x1[2 * i + 1];
    C(2 * i, 2) = 1.0f;
    b[2 * i] = x2[2 * i];
    C(2 * i + 1, 3) = x1[2 * i];
    C(2 * i + 1, 4) = x1[2 * i + 1];
    C(2 * i + 1, 5) = 1.0f;
    b[2 * i + 1] = x2[2 * i + 1];
  }
  // C.jacobiSvd(ComputeThinU | ComputeThinV).solve(b): SVDBase::_solve_impl over rank() (Eigen/src/SVD/SVDBase.h)
  const oracle::SVD svd = oracle::jacobi_svd(C, true);  // the first 6 columns of the full U are the thin U
  const int diag = static_cast<int>(svd.sv.size());
  int nonzero = 0;
  for (int i = 0; i < diag; ++i) nonzero += svd.sv[i] != 0.0;  // m_nonzeroSingularValues
  int rank = 0;
  if (diag > 0) {
    const double thr = std::max(1, diag) * DBL_EPSILON;
    const double premultiplied = std::max(svd.sv[0] * thr, DBL_MIN);
    int i = nonzero - 1;
    while (i >= 0 && svd.sv[i] < premultiplied) --i;
    rank = i + 1;
  }
  double tmp[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < rank; ++k) {
    double s = 0.0;
    for (int r = 0; r < 2 * n; ++r) s += svd.U(r, k) * b[r];
    tmp[k] = (1.0 / svd.sv[k]) * s;
  }
  for (int j = 0; j < 6; ++j) {
    double s = 0.0;
    for (int k = 0; k < rank; ++k) s += svd.V(j, k) * tmp[k];
    A[j] = s;  // the solution (s0..s5) read as Matrix<3,2> column-major and transposed: rows (s0 s1 s2), (s3 s4 s5)
  }
}

// RANSAC<AffineTransformEstimator>::ComputeNumTrials, optim/ransac.h:150-167 (kMinNumSamples = 3)
inline size_t ComputeNumTrials(size_t num_inliers, size_t num_samples, double confidence) {
  const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
  const double nom = 1 - confidence;
  if (nom <= 0) return std::numeric_limits<size_t>::max();
  const double denom = 1 - std::pow(inlier_ratio, 3);
  if (denom <= 0) return 1;
  return static_cast<size_t>(std::ceil(std::log(nom) / std::log(denom)));
}

struct VoteAndVerifyOptions {  // vote_and_verify.h:42-68
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

struct VotingBin {  // vote_and_verify.cc:75-101
  size_t num_votes = 0;
  FeatureGeometryTransform sum;
  void Vote(const FeatureGeometryTransform& t) {
    num_votes += 1;
    sum.scale += t.scale;
    sum.angle += t.angle;
    sum.tx += t.tx;
    sum.ty += t.ty;
  }
  FeatureGeometryTransform GetTransformation() const {
    const float inv_num_votes = 1.0f / static_cast<float>(num_votes);
    FeatureGeometryTransform t = sum;
    t.scale *= inv_num_votes;
    t.angle *= inv_num_votes;
    t.tx *= inv_num_votes;
    t.ty *= inv_num_votes;
    return t;
  }
};

// n_a + num_angle_bins * (n_s + num_scale_bins * (n_x + num_trans_bins * n_y)), vote_and_verify.cc:271-274: an int
// expression in the reference; a NaN transformation passes the range tests and arrives here with INT_MIN coordinates, where
// the int arithmetic overflows (undefined in C++, a two's-complement wrap on x86 -- which is what this computes).
inline uint64_t BinIndex(const VoteAndVerifyOptions& options, int n_a, int n_s, int n_x, int n_y) {
  const uint32_t k = static_cast<uint32_t>(n_a) +
                     static_cast<uint32_t>(options.num_angle_bins) *
                         (static_cast<uint32_t>(n_s) +
                          static_cast<uint32_t>(options.num_scale_bins) *
                              (static_cast<uint32_t>(n_x) + static_cast<uint32_t>(options.num_trans_bins) * static_cast<uint32_t>(n_y)));
  return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(k)));
}

// VoteAndVerify, vote_and_verify.cc:208-418.  platform_order = true (the definition, and what the product implements): the
// candidate bins in the order THIS platform's libstdc++ gives the reference's own containers -- std::unordered_map<size_t,
// ...> iteration order, std::partial_sort'ed by score; false: descending score, equal scores by ascending bin index -- kept
// to measure how often the choice matters (tests/test_retrieval.py::test_bin_order_of_the_platform).
inline int VoteAndVerify(const VoteAndVerifyOptions& options, const std::vector<FeatureGeometryMatch>& matches,
                         bool platform_order = true) {
  if (matches.size() < 3) return 0;  // AffineTransformEstimator::kMinNumSamples
  const float max_trans = options.max_image_size;
  const float kMaxScale = 10.0f;
  const float max_log_scale = std::log2(kMaxScale);
  const float trans_norm = 1.0f / (2.0f * max_trans);
  const float scale_norm = 1.0f / (2.0f * max_log_scale);
  const float angle_norm = 1.0f / (2.0f * M_PI);

  const int kNumLevels = 6;
  std::map<uint64_t, VotingBin> bins[kNumLevels];
  std::map<uint64_t, int> coords_a, coords_s, coords_x, coords_y;
  std::unordered_map<size_t, int> platform_bins0;  // the keys of bins[0] inserted as the reference inserts them
  for (const auto& match : matches) {
    for (const auto& geometry2 : match.geometries2) {
      const auto T = TransformFromMatch(match.geometry1, geometry2);
      if (std::abs(T.tx) > max_trans || std::abs(T.ty) > max_trans) continue;
      const float log_scale = std::log2(T.scale);
      if (std::abs(log_scale) > max_log_scale) continue;
      const float x = (T.tx + max_trans) * trans_norm;
      const float y = (T.ty + max_trans) * trans_norm;
      const float s = (log_scale + max_log_scale) * scale_norm;
      const float a = (T.angle + M_PI) * angle_norm;
      int n_x = std::min(FloatToInt(x * options.num_trans_bins), static_cast<int>(options.num_trans_bins - 1));
      int n_y = std::min(FloatToInt(y * options.num_trans_bins), static_cast<int>(options.num_trans_bins - 1));
      int n_s = std::min(FloatToInt(s * options.num_scale_bins), static_cast<int>(options.num_scale_bins - 1));
      int n_a = std::min(FloatToInt(a * options.num_angle_bins), static_cast<int>(options.num_angle_bins - 1));
      for (int level = 0; level < kNumLevels; ++level) {
        const uint64_t index = BinIndex(options, n_a, n_s, n_x, n_y);
        if (level == 0) {
          coords_a[index] = n_a;
          coords_s[index] = n_s;
          coords_x[index] = n_x;
          coords_y[index] = n_y;
        }
        if (level == 0) platform_bins0[static_cast<size_t>(index)] += 1;
        bins[level][index].Vote(T);
        n_x >>= 1;
        n_y >>= 1;
        n_s >>= 1;
        n_a >>= 1;
      }
    }
  }

  std::vector<std::pair<int, float>> bin_scores;  // (bin index as the reference stores it: int, score); ascending index
  std::vector<uint64_t> bin_keys;
  for (const auto& bin : bins[0]) {
    if (bin.second.num_votes >= static_cast<size_t>(options.min_num_votes)) {
      int n_a = coords_a.at(bin.first), n_s = coords_s.at(bin.first), n_x = coords_x.at(bin.first), n_y = coords_y.at(bin.first);
      float score = bin.second.num_votes;
      float level_weight = 0.5f;
      for (int level = 1; level < kNumLevels; ++level) {
        n_x >>= 1;
        n_y >>= 1;
        n_s >>= 1;
        n_a >>= 1;
        const uint64_t index = BinIndex(options, n_a, n_s, n_x, n_y);
        score += bins[level][index].num_votes * level_weight;
        level_weight *= 0.5f;
      }
      bin_scores.emplace_back(static_cast<int>(bin.first), score);
      bin_keys.push_back(bin.first);
    }
  }
  const size_t num_transformations = std::min(static_cast<size_t>(options.num_transformations), bin_scores.size());
  // std::partial_sort by descending score; equal scores: ascending bin index (see the header)
  std::vector<size_t> order(bin_scores.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bin_scores[a].second > bin_scores[b].second; });
  if (platform_order) {
    std::map<uint64_t, size_t> position;  // key -> index into bin_scores / bin_keys
    for (size_t i = 0; i < bin_keys.size(); ++i) position[bin_keys[i]] = i;
    std::vector<std::pair<size_t, float>> walk;  // (index, score) in the hash table's iteration order
    for (const auto& kv : platform_bins0) {
      const auto it = position.find(static_cast<uint64_t>(kv.first));
      if (it != position.end()) walk.emplace_back(it->second, bin_scores[it->second].second);
    }
    std::partial_sort(walk.begin(), walk.begin() + num_transformations, walk.end(),
                      [](const std::pair<size_t, float>& a, const std::pair<size_t, float>& b) { return a.second > b.second; });
    for (size_t i = 0; i < num_transformations; ++i) order[i] = walk[i].first;
  }

  size_t max_num_trials = std::numeric_limits<size_t>::max();
  size_t best_num_inliers = 0;
  TwoWayTransform best_tform;
  std::vector<std::pair<int, int>> inlier_idxs;
  std::vector<double> inlier_points1, inlier_points2;
  for (size_t i = 0; i < num_transformations && i < max_num_trials; ++i) {
    const auto& bin = bins[0].at(bin_keys[order[i]]);
    const TwoWayTransform tform(bin.GetTransformation());
    ComputeInliers(tform, matches, options.max_transfer_error, options.max_scale_error, &inlier_idxs);
    if (inlier_idxs.size() < best_num_inliers || inlier_idxs.size() < 3) continue;
    best_num_inliers = inlier_idxs.size();
    best_tform = tform;
    if (best_num_inliers == matches.size()) break;
    inlier_points1.resize(2 * inlier_idxs.size());
    inlier_points2.resize(2 * inlier_idxs.size());
    for (size_t j = 0; j < inlier_idxs.size(); ++j) {
      const auto& match = matches.at(inlier_idxs[j].first);
      const auto& geometry1 = match.geometry1;
      const auto& geometry2 = match.geometries2.at(inlier_idxs[j].second);
      inlier_points1[2 * j] = geometry1.x;
      inlier_points1[2 * j + 1] = geometry1.y;
      inlier_points2[2 * j] = geometry2.x;
      inlier_points2[2 * j + 1] = geometry2.y;
    }
    double A[6];
    EstimateAffine(inlier_points1, inlier_points2, A);
    oracle::Mat3 Ah;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Ah(r, c) = r == c ? 1.0 : 0.0;
    for (int c = 0; c < 3; ++c) {
      Ah(0, c) = A[c];
      Ah(1, c) = A[3 + c];
    }
    const oracle::Mat3 inv = oracle::mat3_inverse(Ah);
    TwoWayTransform local_tform;
    local_tform.A12[0] = static_cast<float>(A[0]);
    local_tform.A12[1] = static_cast<float>(A[1]);
    local_tform.A12[2] = static_cast<float>(A[3]);
    local_tform.A12[3] = static_cast<float>(A[4]);
    local_tform.t12[0] = static_cast<float>(A[2]);
    local_tform.t12[1] = static_cast<float>(A[5]);
    local_tform.A21[0] = static_cast<float>(inv(0, 0));
    local_tform.A21[1] = static_cast<float>(inv(0, 1));
    local_tform.A21[2] = static_cast<float>(inv(1, 0));
    local_tform.A21[3] = static_cast<float>(inv(1, 1));
    local_tform.t21[0] = static_cast<float>(inv(0, 2));
    local_tform.t21[1] = static_cast<float>(inv(1, 2));
    ComputeInliers(local_tform, matches, options.max_transfer_error, options.max_scale_error, &inlier_idxs);
    if (inlier_idxs.size() > best_num_inliers) {
      best_num_inliers = inlier_idxs.size();
      best_tform = local_tform;
      if (best_num_inliers == matches.size()) break;
    }
    max_num_trials = ComputeNumTrials(best_num_inliers, matches.size(), options.confidence);
  }
  if (best_num_inliers == 0) return 0;
  const int kNumBins = 64;
  return static_cast<int>(ComputeEffectiveInlierCount(best_tform, matches, options.max_transfer_error, options.max_scale_error, kNumBins));
}

}  // namespace oracle_sv
