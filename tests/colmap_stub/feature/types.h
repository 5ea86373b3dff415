// /root/reference/src/feature/types.h:44-104 (members only)
#pragma once
#include <vector>
#include <Eigen/Core>
#include "util/types.h"
namespace colmap {
struct FeatureKeypoint {
  FeatureKeypoint() : x(0), y(0), a11(1), a12(0), a21(0), a22(1) {}
  float x;
  float y;
  float a11;
  float a12;
  float a21;
  float a22;
};
struct FeatureMatch {
  FeatureMatch() : point2D_idx1(kInvalidPoint2DIdx), point2D_idx2(kInvalidPoint2DIdx) {}
  FeatureMatch(const point2D_t point2D_idx1, const point2D_t point2D_idx2) : point2D_idx1(point2D_idx1), point2D_idx2(point2D_idx2) {}
  point2D_t point2D_idx1 = kInvalidPoint2DIdx;
  point2D_t point2D_idx2 = kInvalidPoint2DIdx;
};
typedef std::vector<FeatureKeypoint> FeatureKeypoints;
typedef Eigen::Matrix<uint8_t, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> FeatureDescriptors;
typedef std::vector<FeatureMatch> FeatureMatches;
}  // namespace colmap
