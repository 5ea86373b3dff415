// /root/reference/src/estimators/two_view_geometry.h:79-306 (data members)
#pragma once
#include <Eigen/Core>
#include "feature/types.h"
namespace colmap {
struct TwoViewGeometry {
  enum ConfigurationType { UNDEFINED = 0, DEGENERATE = 1, CALIBRATED = 2, UNCALIBRATED = 3, PLANAR = 4, PANORAMIC = 5,
                           PLANAR_OR_PANORAMIC = 6, WATERMARK = 7, MULTIPLE = 8 };
  TwoViewGeometry() : config(ConfigurationType::UNDEFINED), tri_angle(0) {}
  int config;
  Eigen::Matrix3d E;
  Eigen::Matrix3d F;
  Eigen::Matrix3d H;
  Eigen::Vector4d qvec;
  Eigen::Vector3d tvec;
  FeatureMatches inlier_matches;
  double tri_angle;
  size_t E_num_inliers;
  size_t F_num_inliers;
  size_t H_num_inliers;
  size_t T_num_tracks;
};
}  // namespace colmap
