// feature_matching.h -- drop-in host side of the matching + verification stage, above the C-ABI.
//
// Same surface as the reference so that the two call sites compile unchanged
// (/root/reference/src/controllers/distributed_mapper_controller.cpp:506-520,
//  /root/reference/src/controllers/incremental_mapper_controller.cc:450-471):
//     FeatureMatcherCache cache(5 * num_images, &database);
//     SiftFeatureMatcher matcher(options, &database, &cache);
//     matcher.Setup();  cache.Setup();  matcher.Match(image_pairs);
// Classes mirrored: FeatureMatcherCache (src/feature/matching.h:180-212, matching.cc:215-316),
// SiftFeatureMatcher (matching.h:320-391, matching.cc:610-839), ExhaustiveFeatureMatcher
// (matching.h:393-414, matching.cc:841-915).  The reference's thread pipeline (matcher threads,
// verifier threads, JobQueues) is replaced by one pass of the HIP kernels over the pair list.
#ifndef DAGSFM_AMD_HOST_FEATURE_MATCHING_H_
#define DAGSFM_AMD_HOST_FEATURE_MATCHING_H_

#include <string>
#include <exception>
#include <list>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "database.h"
#include "types.h"

namespace dagsfm_amd {

struct ExhaustiveMatchingOptions {  // matching.h:52-60
  int block_size = 50;
  bool Check() const { return block_size > 1; }
};

// All cameras / images resident (matching.cc:221-243); keypoints and descriptors go through an LRU of cache_size
// images like the reference's (matching.h:203-206): Match() asks for the images of ITS pair list only, so a
// database far larger than host RAM is matched block by block.
class FeatureMatcherCache {
 public:
  FeatureMatcherCache(size_t cache_size, const Database* database);
  void Setup();
  size_t CacheSize() const { return cache_size_; }

  const Camera& GetCamera(camera_t camera_id) const { return cameras_cache_.at(camera_id); }
  const Image& GetImage(image_t image_id) const { return images_cache_.at(image_id); }
  std::vector<image_t> GetImageIds() const;
  // The references stay valid until the next Get* call that has to evict (the LRU never evicts an image that was
  // requested since the last ReleasePins()).
  const FeatureKeypoints& GetKeypoints(image_t image_id);
  const FeatureDescriptors& GetDescriptors(image_t image_id);
  void ReleasePins();
  size_t NumCachedImages() const { return features_.size(); }
  FeatureMatches GetMatches(image_t a, image_t b) const {
    std::lock_guard<std::mutex> lock(mutex_);
    return database_->ReadMatches(a, b);
  }
  // The pair ids of both result tables are read in bulk by Setup() and kept current here, so the two existence
  // checks Match() makes for every pair (matching.cc:782-812) cost a hash lookup instead of a SELECT each.
  // One mutex serialises every touch of the (single, NOMUTEX) connection and of the id sets, like the reference's
  // database_mutex_ (matching.h:208): the asynchronous write-back thread and the caller may both be here.
  bool ExistsMatches(image_t a, image_t b) const {
    std::lock_guard<std::mutex> lock(mutex_);
    return have_matches_.count(Database::ImagePairToPairId(a, b)) != 0;
  }
  bool ExistsInlierMatches(image_t a, image_t b) const {
    std::lock_guard<std::mutex> lock(mutex_);
    return have_inliers_.count(Database::ImagePairToPairId(a, b)) != 0;
  }
  // Holds the mutex for a whole batch of the calls below (the *Unlocked variants), so that a batch is not
  // interleaved lock by lock with the write-back thread.
  std::unique_lock<std::mutex> Lock() const { return std::unique_lock<std::mutex>(mutex_); }
  bool ExistsMatchesUnlocked(image_t a, image_t b) const { return have_matches_.count(Database::ImagePairToPairId(a, b)) != 0; }
  bool ExistsInlierMatchesUnlocked(image_t a, image_t b) const { return have_inliers_.count(Database::ImagePairToPairId(a, b)) != 0; }
  FeatureMatches GetMatchesUnlocked(image_t a, image_t b) const { return database_->ReadMatches(a, b); }
  void DeleteMatchesUnlocked(image_t a, image_t b) {
    database_->DeleteMatches(a, b);
    have_matches_.erase(Database::ImagePairToPairId(a, b));
  }
  void DeleteInlierMatchesUnlocked(image_t a, image_t b) {
    database_->DeleteInlierMatches(a, b);
    have_inliers_.erase(Database::ImagePairToPairId(a, b));
  }
  void MarkPendingUnlocked(image_t a, image_t b) {
    have_matches_.insert(Database::ImagePairToPairId(a, b));
    have_inliers_.insert(Database::ImagePairToPairId(a, b));
  }
  // the rows of this pair are on their way (asynchronous write-back): later Match() calls must skip it
  void MarkPending(image_t a, image_t b) {
    std::lock_guard<std::mutex> lock(mutex_);
    have_matches_.insert(Database::ImagePairToPairId(a, b));
    have_inliers_.insert(Database::ImagePairToPairId(a, b));
  }
  void WriteMatches(image_t a, image_t b, const FeatureMatches& m) {
    std::lock_guard<std::mutex> lock(mutex_);
    database_->WriteMatches(a, b, m);
    have_matches_.insert(Database::ImagePairToPairId(a, b));
  }
  void WriteTwoViewGeometry(image_t a, image_t b, const TwoViewGeometry& t) {
    std::lock_guard<std::mutex> lock(mutex_);
    database_->WriteTwoViewGeometry(a, b, t);
    have_inliers_.insert(Database::ImagePairToPairId(a, b));
  }
  void DeleteMatches(image_t a, image_t b) {
    std::lock_guard<std::mutex> lock(mutex_);
    database_->DeleteMatches(a, b);
    have_matches_.erase(Database::ImagePairToPairId(a, b));
  }
  void DeleteInlierMatches(image_t a, image_t b) {
    std::lock_guard<std::mutex> lock(mutex_);
    database_->DeleteInlierMatches(a, b);
    have_inliers_.erase(Database::ImagePairToPairId(a, b));
  }
  void BeginTransaction() const {
    std::lock_guard<std::mutex> lock(mutex_);
    database_->BeginTransaction();
  }
  void EndTransaction() const {
    std::lock_guard<std::mutex> lock(mutex_);
    database_->EndTransaction();
  }

 private:
  const size_t cache_size_;
  const Database* database_;
  std::unordered_map<camera_t, Camera> cameras_cache_;
  std::unordered_map<image_t, Image> images_cache_;
  struct Features {
    FeatureKeypoints keypoints;
    FeatureDescriptors descriptors;
    bool have_kp = false, have_desc = false, pinned = false;
    std::list<image_t>::iterator lru_it;
  };
  Features& Touch(image_t image_id);  // mutex_ held
  std::unordered_map<image_t, Features> features_;
  std::list<image_t> lru_;  // front = most recently used
  std::unordered_set<image_pair_t> have_matches_, have_inliers_;
  mutable std::mutex mutex_;
};

class SiftFeatureMatcher {
 public:
  SiftFeatureMatcher(const SiftMatchingOptions& options, Database* database, FeatureMatcherCache* cache);
  ~SiftFeatureMatcher();

  // Creates the device context; false when no usable GPU is present (matching.cc:732-742).
  bool Setup();
  // Matches + verifies the pairs and writes `matches` / `two_view_geometries` rows, with the
  // reference's dedupe / skip / partial-recompute / post-filter semantics (matching.cc:749-839).
  void Match(const std::vector<std::pair<image_t, image_t>>& image_pairs);

  // Waits for an asynchronous write-back (SiftMatchingOptions::async_write_back) and rethrows its error, if any.
  void Flush();

  const std::string& LastError() const { return last_error_; }
  size_t NumDevices() const { return ctxs_.size(); }
  size_t NumResidentImages() const { return image_ids_.size(); }

 private:
  // Makes the images the pair lists refer to resident on every device (replicated: SURVEY 8e); keeps what is
  // already there when the union fits cache_size images, otherwise replaces it.
  bool EnsureResident(const std::vector<std::pair<image_t, image_t>>& a, const std::vector<std::pair<image_t, image_t>>& b);
  SiftMatchingOptions options_;
  Database* database_;
  FeatureMatcherCache* cache_;
  bool is_setup_ = false;
  std::vector<dsm_ctx*> ctxs_;                         // one per device of gpu_index ("-1": every visible device)
  std::vector<int> devices_;
  size_t max_resident_ = 0;                            // cache_size of the FeatureMatcherCache
  std::vector<image_t> image_ids_;                    // device image index -> image_id
  std::vector<uint32_t> image_nfeat_;
  std::unordered_map<image_t, uint32_t> image_index_;  // image_id -> device image index
  std::string last_error_;
  std::thread writer_;                // at most one write-back in flight
  std::exception_ptr writer_error_;
};

// ExhaustiveFeatureMatcher::Run, matching.cc:853-915: blocks of block_size x block_size images,
// one transaction + one Match() per block.
class ExhaustiveFeatureMatcher {
 public:
  ExhaustiveFeatureMatcher(const ExhaustiveMatchingOptions& options, const SiftMatchingOptions& match_options,
                           const std::string& database_path);
  bool Run();

 private:
  ExhaustiveMatchingOptions options_;
  SiftMatchingOptions match_options_;
  Database database_;
  FeatureMatcherCache cache_;
  SiftFeatureMatcher matcher_;
};

}  // namespace dagsfm_amd
#endif
