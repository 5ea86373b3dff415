// types.h -- host-side data types of the matching + verification path, mirroring the reference:
//   util/types.h:48-77 (id typedefs), feature/types.h:44-104 (FeatureKeypoint/Match/Descriptors),
//   feature/sift.h:116-165 (SiftMatchingOptions), estimators/two_view_geometry.h:79-306 (TwoViewGeometry),
//   base/camera.h / base/image.h (only the members this path reads).
#ifndef DAGSFM_AMD_HOST_TYPES_H_
#define DAGSFM_AMD_HOST_TYPES_H_

#include <cstdint>
#include <limits>
#include <string>
#include <vector>

#include "../../include/dagsfm_mi355x.h"

namespace dagsfm_amd {

typedef uint32_t camera_t;
typedef uint32_t image_t;
typedef uint64_t image_pair_t;
typedef uint32_t point2D_t;
const image_t kInvalidImageId = std::numeric_limits<image_t>::max();
const point2D_t kInvalidPoint2DIdx = std::numeric_limits<point2D_t>::max();

struct FeatureKeypoint {  // feature/types.h:44-81
  float x = 0, y = 0, a11 = 1, a12 = 0, a21 = 0, a22 = 1;
};
typedef std::vector<FeatureKeypoint> FeatureKeypoints;

struct FeatureDescriptors {  // row-major uint8 [rows][cols], feature/types.h:102-103
  size_t rows = 0, cols = 128;
  std::vector<uint8_t> data;
};

struct FeatureMatch {  // feature/types.h:86-99
  point2D_t point2D_idx1 = kInvalidPoint2DIdx;
  point2D_t point2D_idx2 = kInvalidPoint2DIdx;
  FeatureMatch() {}
  FeatureMatch(point2D_t a, point2D_t b) : point2D_idx1(a), point2D_idx2(b) {}
};
typedef std::vector<FeatureMatch> FeatureMatches;

struct Camera {  // base/camera.h (subset)
  camera_t camera_id = 0;
  int model_id = 0;
  size_t width = 0, height = 0;
  std::vector<double> params;
  bool prior_focal_length = false;
  bool HasPriorFocalLength() const { return prior_focal_length; }
};

struct Image {  // base/image.h (subset)
  image_t image_id = 0;
  std::string name;
  camera_t camera_id = 0;
};

struct TwoViewGeometry {  // estimators/two_view_geometry.h:79-306
  int config = DSM_CONFIG_UNDEFINED;
  double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // row-major
  double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double qvec[4] = {0, 0, 0, 0};
  double tvec[3] = {0, 0, 0};
  FeatureMatches inlier_matches;
  double tri_angle = 0;
  void Invert();  // two_view_geometry.cc:98-111
};

struct SiftMatchingOptions {  // feature/sift.h:116-165, same names and defaults
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
  // extension (not a reference option): the rows of slice k of a Match() are written by a writer thread while slice k + 1 is on
  // the devices (SQLite overlaps the device work); Match() still returns only when every row of the call is written -- the
  // reference's contract (matching.cc:819-836 writes before it returns).  The writer opens one transaction per slice, unless
  // the caller holds one on the connection (the reference's Run() does, matching.cc:903): then the rows go into the caller's.
  // false: the caller's thread writes the rows between the slices.  Default on since round 5 (53.6 k -> 78 k pairs/s end to end).
  bool async_write_back = true;
  // extension: with async_write_back, Match() returns while its LAST slice is still being written, so that the write-back also
  // overlaps the NEXT Match() call's device work; Flush() / the destructor waits for it and rethrows its error.  The writer
  // owns the transaction then: the caller must not hold one.  ExhaustiveFeatureMatcher::Run switches it on for its own loop.
  bool defer_write_back = false;
  // extension: a Match() over more pairs than 1.5 x this goes to the devices in slices of about this many pairs (0: never
  // sliced) -- bounded device scratch (the first call of a process allocates it), and with async_write_back slice k's rows are
  // written while slice k + 1 is on the devices
  int match_slice_pairs = 32768;
  // extension: the devices' shares of a Match() are assembled on the devices by RCCL (libdagsfm_gather.so: one communicator over
  // the gpu_index devices, grouped all-gather + broadcasts over xGMI) and fetched from device 0 in one copy per array, instead
  // of one fetch per device merged in host memory.  Off by default: every GPU has its own PCIe link, so N parallel fetches
  // reach the host faster than one N-times-larger fetch behind an all-gather; the switch is for hosts that consume the graph
  // on the device next (dsm_gather_device_arrays) and for exercising the RCCL path.  The gpu_index devices must be distinct.
  bool assemble_on_device = false;
  // extension: Database::SetBulkLoadJournal(true) for the run (ExhaustiveFeatureMatcher::Run restores WAL at its end)
  bool bulk_load_journal = false;
  // not in the reference: seed of the per-pair PRNG schedule (the reference seeds from the clock)
  uint32_t random_seed = 0;
  bool Check() const;  // feature/sift.cc:236-250
};

}  // namespace dagsfm_amd
#endif
