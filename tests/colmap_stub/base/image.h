// /root/reference/src/base/image.h:70-90
#pragma once
#include "util/types.h"
namespace colmap {
class Image {
 public:
  inline image_t ImageId() const { return image_id_; }
  inline camera_t CameraId() const { return camera_id_; }

 private:
  image_t image_id_ = kInvalidImageId;
  camera_t camera_id_ = kInvalidCameraId;
};
}  // namespace colmap
