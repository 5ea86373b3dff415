// /root/reference/src/feature/sift.h:116-165
#pragma once
#include <string>
namespace colmap {
struct SiftMatchingOptions {
  int num_threads = -1;
  bool use_gpu = true;
  std::string gpu_index = "-1";
  double max_ratio = 0.8;
  double max_distance = 0.7;
  bool cross_check = true;
  int max_num_matches = 32768;
  double max_error = 4.0;
  double confidence = 0.999;
  int min_num_trials = 30;
  int max_num_trials = 10000;
  double min_inlier_ratio = 0.25;
  int min_num_inliers = 15;
  bool multiple_models = false;
  bool guided_matching = false;
  bool Check() const;
};
}  // namespace colmap
