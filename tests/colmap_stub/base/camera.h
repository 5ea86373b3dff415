// /root/reference/src/base/camera.h:55-102 (the accessors the matcher uses)
#pragma once
#include <cstddef>
#include <vector>
#include "util/types.h"
namespace colmap {
class Camera {
 public:
  inline camera_t CameraId() const { return camera_id_; }
  inline int ModelId() const { return model_id_; }
  inline size_t Width() const { return width_; }
  inline size_t Height() const { return height_; }
  inline bool HasPriorFocalLength() const { return prior_focal_length_; }
  inline const std::vector<double>& Params() const { return params_; }

 private:
  camera_t camera_id_ = kInvalidCameraId;
  int model_id_ = -1;
  size_t width_ = 0, height_ = 0;
  std::vector<double> params_;
  bool prior_focal_length_ = false;
};
}  // namespace colmap
