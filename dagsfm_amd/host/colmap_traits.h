// colmap_traits.h -- INTEGRATION.md variant A: the reference's own SiftFeatureMatcher, replaced in place.
//
// Include this header from /root/reference/src/feature/matching.h where `class SiftFeatureMatcher` is declared
// (matching.h:334-368), after FeatureMatcherCache (matching.h:180-212), and drop the class from matching.{h,cc}:
// the controllers (src/controllers/distributed_mapper_controller.cpp:506-520, incremental_mapper_controller.cc:
// 450-471) and every *FeatureMatcher::Run (matching.cc:853-915 ...) then compile unchanged against
//     colmap::SiftFeatureMatcher(const SiftMatchingOptions&, Database*, FeatureMatcherCache*); Setup(); Match(pairs);
// and run on the MI355X library.  Uses only the public interface of the reference's types:
//   Camera::ModelId/Width/Height/Params/HasPriorFocalLength (base/camera.h:55-102), Image::CameraId (base/image.h:84),
//   FeatureKeypoint{x,y,a11,a12,a21,a22} (feature/types.h:44-81), FeatureDescriptors = Eigen row-major uint8 matrix
//   (:102-103), FeatureMatch (:86-99), TwoViewGeometry{config,E,F,H,qvec,tvec,inlier_matches,tri_angle}
//   (estimators/two_view_geometry.h:79-306), Database::ImagePairToPairId (base/database.h:336-347) and the
//   FeatureMatcherCache methods.  tests/test_integration_variant_a.py compiles it against tests/colmap_stub, a
//   header set with exactly those signatures (Eigen itself is not available in the build container).
#ifndef DAGSFM_AMD_HOST_COLMAP_TRAITS_H_
#define DAGSFM_AMD_HOST_COLMAP_TRAITS_H_

#include "sift_feature_matcher_impl.h"

namespace colmap {

struct DsmColmapTraits {
  typedef SiftMatchingOptions Options;
  typedef colmap::Database Database;
  typedef FeatureMatcherCache Cache;
  typedef colmap::Camera Camera;
  typedef colmap::Image Image;
  typedef colmap::FeatureKeypoints FeatureKeypoints;
  typedef colmap::FeatureDescriptors FeatureDescriptors;
  typedef colmap::FeatureMatches FeatureMatches;
  typedef colmap::TwoViewGeometry TwoViewGeometry;
  // the reference's cache is a plain LRU (util/cache.h): a reference may be evicted by the next Get, so the matcher
  // copies; it has no asynchronous write-back
  static constexpr bool kCachePinsRequested = false;
  static constexpr bool kAsyncWriteBack = false;
  struct NoLock {};
  static uint64_t PairId(image_t a, image_t b) { return Database::ImagePairToPairId(a, b); }
  // FeatureMatcherCache keeps cache_size_ private; the controllers size it 5 * num_images (all images fit), the
  // matchers 5 * block_size with at most 2 * block_size images per Match() call
  static size_t CacheSize(const Cache*) { return 0; }
  static void ReleasePins(Cache*) {}
  static NoLock LockBatch(const Cache*) { return NoLock(); }
  static void ToDsmCamera(const Camera& c, dsm_camera* out) {
    out->model_id = c.ModelId();
    out->has_prior_focal_length = c.HasPriorFocalLength() ? 1 : 0;
    out->width = c.Width();
    out->height = c.Height();
    const std::vector<double>& p = c.Params();
    for (size_t k = 0; k < p.size() && k < 12; ++k) out->params[k] = p[k];
  }
  static camera_t CameraIdOf(const Image& im) { return im.CameraId(); }
  static const float* KeypointData(const FeatureKeypoints& k, size_t* n, uint32_t* stride) {
    static_assert(sizeof(FeatureKeypoint) == 6 * sizeof(float), "FeatureKeypoint layout (feature/sift.cc:997-1002 asserts the same)");
    *n = k.size();
    *stride = 6;
    return k.empty() ? nullptr : &k[0].x;
  }
  static const uint8_t* DescriptorData(const FeatureDescriptors& d, size_t* rows, size_t* cols) {
    *rows = static_cast<size_t>(d.rows());
    *cols = static_cast<size_t>(d.cols());
    return d.data();  // Eigen::RowMajor: rows are contiguous
  }
  static void AppendFlat(const FeatureMatches& m, std::vector<uint32_t>* flat) {
    for (const FeatureMatch& x : m) {
      flat->push_back(x.point2D_idx1);
      flat->push_back(x.point2D_idx2);
    }
  }
  static FeatureMatches MakeMatches(const uint32_t* flat, size_t n) {
    FeatureMatches m(n);
    for (size_t k = 0; k < n; ++k) m[k] = FeatureMatch(flat[2 * k], flat[2 * k + 1]);
    return m;
  }
  static TwoViewGeometry MakeTwoViewGeometry(const dsm_two_view_geometry* r, const uint32_t* inliers, size_t n) {
    TwoViewGeometry t;  // TwoViewGeometry() (two_view_geometry.h:159-166)
    if (!r) return t;
    t.config = r->config;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        t.E(i, j) = r->E[i * 3 + j];
        t.F(i, j) = r->F[i * 3 + j];
        t.H(i, j) = r->H[i * 3 + j];
      }
    for (int i = 0; i < 4; ++i) t.qvec(i) = r->qvec[i];
    for (int i = 0; i < 3; ++i) t.tvec(i) = r->tvec[i];
    t.tri_angle = r->tri_angle;
    t.inlier_matches = MakeMatches(inliers, n);
    return t;
  }
  static uint32_t RandomSeed(const Options&) { return 0; }
  static bool AsyncWriteBack(const Options&) { return false; }
  static bool AssembleOnDevice(const Options&) { return false; }  // (the reference's options have no such switch)
};

// Same name, constructor and methods as the class it replaces (matching.h:334-368).
class SiftFeatureMatcher : public dagsfm_amd::SiftFeatureMatcherT<DsmColmapTraits> {
 public:
  SiftFeatureMatcher(const SiftMatchingOptions& options, Database* database, FeatureMatcherCache* cache)
      : dagsfm_amd::SiftFeatureMatcherT<DsmColmapTraits>(options, database, cache) {}
};

}  // namespace colmap
#endif  // DAGSFM_AMD_HOST_COLMAP_TRAITS_H_
