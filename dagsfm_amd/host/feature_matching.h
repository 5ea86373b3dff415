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

#include <cstring>

#include "database.h"
#include "sift_feature_matcher_impl.h"
#include "types.h"

namespace dagsfm_amd {

struct ExhaustiveMatchingOptions {  // matching.h:52-60
  int block_size = 50;
  // extension: Run() creates the device contexts on a second thread while this one reads the cache's tables and the first
  // block's features (false: one after the other, the reference's order)
  bool overlap_setup = true;
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
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    return database_->ReadMatches(a, b);
  }
  // The pair ids of both result tables are read in bulk by Setup() and kept current here, so the two existence
  // checks Match() makes for every pair (matching.cc:782-812) cost a hash lookup instead of a SELECT each.
  // One mutex serialises every touch of the (single, NOMUTEX) connection and of the id sets, like the reference's
  // database_mutex_ (matching.h:208): the asynchronous write-back thread and the caller may both be here.
  bool ExistsMatches(image_t a, image_t b) const {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    return have_matches_.count(Database::ImagePairToPairId(a, b)) != 0;
  }
  bool ExistsInlierMatches(image_t a, image_t b) const {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    return have_inliers_.count(Database::ImagePairToPairId(a, b)) != 0;
  }
  // Holds the mutex for a whole batch of the calls below (the *Unlocked variants), so that a batch is not
  // interleaved lock by lock with the write-back thread.
  std::unique_lock<std::recursive_mutex> Lock() const { return std::unique_lock<std::recursive_mutex>(mutex_); }
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
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    have_matches_.insert(Database::ImagePairToPairId(a, b));
    have_inliers_.insert(Database::ImagePairToPairId(a, b));
  }
  void WriteMatches(image_t a, image_t b, const FeatureMatches& m) {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    database_->WriteMatches(a, b, m);
    have_matches_.insert(Database::ImagePairToPairId(a, b));
  }
  void WriteTwoViewGeometry(image_t a, image_t b, const TwoViewGeometry& t) {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    database_->WriteTwoViewGeometry(a, b, t);
    have_inliers_.insert(Database::ImagePairToPairId(a, b));
  }
  void DeleteMatches(image_t a, image_t b) {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    database_->DeleteMatches(a, b);
    have_matches_.erase(Database::ImagePairToPairId(a, b));
  }
  void DeleteInlierMatches(image_t a, image_t b) {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    database_->DeleteInlierMatches(a, b);
    have_inliers_.erase(Database::ImagePairToPairId(a, b));
  }
  void BeginTransaction() const {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    database_->BeginTransaction();
  }
  void EndTransaction() const {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    database_->EndTransaction();
  }
  bool InTransaction() const {  // the connection is inside a BEGIN ... END (whoever opened it)
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    return database_->InTransaction();
  }
  // After a failed batch: drops the open transaction's rows (if one is open) and re-reads which pairs the database
  // holds, so that ExistsMatches / ExistsInlierMatches answer for the database again (not in the reference, whose
  // CHECKs abort the process instead of unwinding).
  void RollbackTransaction(bool transaction_open);

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
  mutable std::recursive_mutex mutex_;  // recursive: Lock() (a whole batch) composes with the per-call methods
};

// The types of this repository's own host side, as SiftFeatureMatcherT sees them (sift_feature_matcher_impl.h).
struct NativeTraits {
  typedef SiftMatchingOptions Options;
  typedef dagsfm_amd::Database Database;
  typedef FeatureMatcherCache Cache;
  typedef dagsfm_amd::Camera Camera;
  typedef dagsfm_amd::Image Image;
  typedef dagsfm_amd::FeatureKeypoints FeatureKeypoints;
  typedef dagsfm_amd::FeatureDescriptors FeatureDescriptors;
  typedef dagsfm_amd::FeatureMatches FeatureMatches;
  typedef dagsfm_amd::TwoViewGeometry TwoViewGeometry;
  static constexpr bool kCachePinsRequested = true;
  static constexpr bool kAsyncWriteBack = true;
  static uint64_t PairId(image_t a, image_t b) { return Database::ImagePairToPairId(a, b); }
  static size_t CacheSize(const Cache* c) { return c->CacheSize(); }
  static void ReleasePins(Cache* c) { c->ReleasePins(); }
  static std::unique_lock<std::recursive_mutex> LockBatch(const Cache* c) { return c->Lock(); }
  static void ToDsmCamera(const Camera& c, dsm_camera* out) {
    out->model_id = c.model_id;
    out->has_prior_focal_length = c.HasPriorFocalLength() ? 1 : 0;
    out->width = c.width;
    out->height = c.height;
    for (size_t p = 0; p < c.params.size() && p < 12; ++p) out->params[p] = c.params[p];
  }
  static camera_t CameraIdOf(const Image& im) { return im.camera_id; }
  static const float* KeypointData(const FeatureKeypoints& k, size_t* n, uint32_t* stride) {
    static_assert(sizeof(FeatureKeypoint) == 6 * sizeof(float), "FeatureKeypoint layout");
    *n = k.size();
    *stride = 6;
    return k.empty() ? nullptr : &k[0].x;
  }
  static const uint8_t* DescriptorData(const FeatureDescriptors& d, size_t* rows, size_t* cols) {
    *rows = d.rows;
    *cols = d.cols;
    return d.data.data();
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
    TwoViewGeometry t;
    if (!r) return t;
    t.config = r->config;
    std::memcpy(t.E, r->E, sizeof(t.E));
    std::memcpy(t.F, r->F, sizeof(t.F));
    std::memcpy(t.H, r->H, sizeof(t.H));
    std::memcpy(t.qvec, r->qvec, sizeof(t.qvec));
    std::memcpy(t.tvec, r->tvec, sizeof(t.tvec));
    t.tri_angle = r->tri_angle;
    t.inlier_matches = MakeMatches(inliers, n);
    return t;
  }
  static uint32_t RandomSeed(const Options& o) { return o.random_seed; }
  static bool AsyncWriteBack(const Options& o) { return o.async_write_back; }
  static bool AssembleOnDevice(const Options& o) { return o.assemble_on_device; }
  static bool DeferWriteBack(const Options& o) { return o.async_write_back && o.defer_write_back; }
  static bool InTransaction(const Cache* c) { return c->InTransaction(); }
  static size_t MatchSlicePairs(const Options& o) { return o.match_slice_pairs > 0 ? static_cast<size_t>(o.match_slice_pairs) : 0; }
};

// SiftFeatureMatcher of this repository's host side: Setup() / Match() / Flush() as documented in
// sift_feature_matcher_impl.h.
class SiftFeatureMatcher : public SiftFeatureMatcherT<NativeTraits> {
 public:
  using SiftFeatureMatcherT<NativeTraits>::SiftFeatureMatcherT;
};

// ExhaustiveFeatureMatcher::Run, matching.cc:853-915: blocks of block_size x block_size images,
// one transaction + one Match() per block.
class ExhaustiveFeatureMatcher {
 public:
  ExhaustiveFeatureMatcher(const ExhaustiveMatchingOptions& options, const SiftMatchingOptions& match_options,
                           const std::string& database_path);
  bool Run();
  // where Run()'s wall time went: SiftFeatureMatcher's stage timers + Run()'s own total
  SiftFeatureMatcher::Timings MatcherTimings() const { return matcher_.GetTimings(); }
  double run_seconds = 0.0;
  double setup_seconds = 0.0;  // before that: device contexts, cache set-up, the first block's features

 private:
  ExhaustiveMatchingOptions options_;
  SiftMatchingOptions match_options_;
  Database database_;
  FeatureMatcherCache cache_;
  SiftFeatureMatcher matcher_;
};

// Candidate pairs by vocabulary-tree retrieval: DAGSfM::VocabSimilarityGraph (src/graph/similarity_graph.h:52-89,
// similarity_graph.cpp:101-199) over the C-ABI's dsm_retrieval_* entry points.  Run() indexes every image of the
// database, queries every image and keeps (image_id, retrieved image_id) for image_id < retrieved, with
// score * 1e3 (similarity_graph.cpp:186-193), in the order of the image ids.
struct VocabSimilaritySearchOptions {  // similarity_graph.h:52-76
  int num_images = 100;
  int num_nearest_neighbors = 5;
  int num_checks = 256;                   // FLANN search effort (QueryOptions::num_checks); the FLANN word searches consult it
  // How a feature's visual words are found.
  //   kAuto (default since round 5)  kFlann when the vocabulary file carries a FLANN index (every file the reference wrote does),
  //                     kExact otherwise (this library's flat file has none)
  //   kFlann            the reference's own answer: the approximate search of the flann::AutotunedIndex stored in the vocabulary
  //                     file (visual_index.h:695-738), ON THE DEVICE (csrc/flann_search.hip: FLANN's visit order, heap and result
  //                     set, a lane per feature; ids equal the reference's knnSearch bit for bit)
  //   kFlannHost        the same search on host threads (flann_index.h) with the ids handed to the device: the cross-check of
  //                     kFlann, and the fallback for a vocabulary whose branch heaps outgrow the device's per-lane capacity
  //   kExact            the TRUE nearest words (int8 MFMA search on the device) -- a deviation from the reference: on tree indices
  //                     it agrees with FLANN's top word for about a fifth of the features (profiles/r04_flann_agreement.json)
  enum WordSearch { kExact = 0, kFlann = 1, kFlannHost = 2, kAuto = 3 };
  WordSearch word_search = kAuto;
  int num_images_after_verification = 0;  // > 0: spatial re-ranking of the retrieved images (spatial_verification.h); 0 = off is the reference's default
  int max_num_features = -1;              // > 0: index and query only the features of largest scale (ExtractTopScaleFeatures)
  int num_threads = 8;
  std::string vocab_tree_path;
  bool Check() const { return num_images > 0 && !vocab_tree_path.empty() && num_images_after_verification >= 0; }
};

// The vocabulary.  Read() takes either layout:
//   * the reference's own vocabulary-tree file, as VisualIndex<>::Write leaves it (retrieval/visual_index.h:586-614):
//       uint64 rows, uint64 cols (= 128), the visual words u8 [rows][cols];
//       FLANN's serialised search index (third-party layout, not needed: the word search is exact here);
//       the inverted index (inverted_index.h:342-381): int32 num_words, int32 64, the projection f32 [64][128], then
//       per word (inverted_file.h:375-392) u8 status, f32 idf, f32 thresholds[64], uint32 entries x 32 B, and at the
//       end int32 num_images x (int32 id, f32 constant).
//     The FLANN blob carries no length, so the inverted index is located by its header -- (num_words, 64) -- at the
//     one offset from which the rest of the file parses to exactly its end.
//   * a flat file of this library (Write()): "DSMVOC1\0", uint32 num_words, uint32 reserved, words u8 [W][128],
//     projection f32 [64][128], thresholds f32 [W][64]
struct VocabularyFile {
  uint32_t num_words = 0;
  std::vector<uint8_t> words;
  std::vector<float> projection, thresholds;
  // reference layout only: the byte range of the serialised FLANN index between the words and the inverted index, and
  // whether it carries FLANN's v1.1 archive framing (then index_end is the offset flann's loadIndex stops at)
  uint64_t index_begin = 0, index_end = 0;
  bool flann_framed = false;
  std::vector<uint8_t> flann_blob;  // the bytes of that range (what FlannIndex::Load parses for word_search = flann)
  bool Read(const std::string& path);
  bool ReadReferenceLayout(const std::string& path);
  bool Write(const std::string& path) const;
};

// ExtractTopScaleFeatures (src/feature/utils.cc:79-113): the indices of the num_features keypoints of largest scale, in
// the order the reference leaves them (std::partial_sort on (index, scale) by descending scale -- the same library
// call, so ties fall the same way); every index in order when there are no more than num_features.
std::vector<uint32_t> TopScaleFeatureOrder(const FeatureKeypoints& keypoints, size_t num_features);

class VocabSimilarityGraph {
 public:
  VocabSimilarityGraph(const VocabSimilaritySearchOptions& options, const Database& database);
  bool Run();
  const std::vector<std::pair<image_t, image_t>>& ImagePairs() const { return image_pairs_; }
  const std::vector<float>& Scores() const { return scores_; }
  const std::string& LastError() const { return last_error_; }
  // what kAuto resolved to in the last Run()
  VocabSimilaritySearchOptions::WordSearch WordSearchUsed() const { return word_search_used_; }

 private:
  VocabSimilaritySearchOptions options_;
  VocabSimilaritySearchOptions::WordSearch word_search_used_ = VocabSimilaritySearchOptions::kExact;
  const Database* database_;
  FeatureMatcherCache cache_;
  std::vector<std::pair<image_t, image_t>> image_pairs_;
  std::vector<float> scores_;
  std::string last_error_;
};

}  // namespace dagsfm_amd
#endif
