// /root/reference/src/feature/matching.h with `class SiftFeatureMatcher` (:334-368) replaced by the include of
// dagsfm_amd/host/colmap_traits.h -- the one-line patch of INTEGRATION.md variant A.  FeatureMatcherCache is the
// reference's declaration (:180-212), signature for signature.
#pragma once
#include <utility>
#include <vector>
#include "base/camera.h"
#include "base/database.h"
#include "base/image.h"
#include "estimators/two_view_geometry.h"
#include "feature/sift.h"
#include "feature/types.h"
namespace colmap {
class FeatureMatcherCache {
 public:
  FeatureMatcherCache(const size_t cache_size, const Database* database);
  void Setup();
  const Camera& GetCamera(const camera_t camera_id) const;
  const Image& GetImage(const image_t image_id) const;
  const FeatureKeypoints& GetKeypoints(const image_t image_id);
  const FeatureDescriptors& GetDescriptors(const image_t image_id);
  FeatureMatches GetMatches(const image_t image_id1, const image_t image_id2);
  std::vector<image_t> GetImageIds() const;
  bool ExistsMatches(const image_t image_id1, const image_t image_id2);
  bool ExistsInlierMatches(const image_t image_id1, const image_t image_id2);
  void WriteMatches(const image_t image_id1, const image_t image_id2, const FeatureMatches& matches);
  void WriteTwoViewGeometry(const image_t image_id1, const image_t image_id2, const TwoViewGeometry& two_view_geometry);
  void DeleteMatches(const image_t image_id1, const image_t image_id2);
  void DeleteInlierMatches(const image_t image_id1, const image_t image_id2);
};
}  // namespace colmap

#include "dagsfm_amd/host/colmap_traits.h"  // class colmap::SiftFeatureMatcher
