// feature_matching.cc -- see feature_matching.h.
#include "feature_matching.h"

#include <chrono>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <unordered_set>

namespace dagsfm_amd {

FeatureMatcherCache::FeatureMatcherCache(size_t cache_size, const Database* database)
    : cache_size_(cache_size), database_(database) {
  if (!database) throw std::invalid_argument("FeatureMatcherCache: null database");
}

void FeatureMatcherCache::Setup() {  // matching.cc:221-243
  for (const Camera& c : database_->ReadAllCameras()) cameras_cache_.emplace(c.camera_id, c);
  for (const Image& im : database_->ReadAllImages()) images_cache_.emplace(im.image_id, im);
  have_matches_.clear();
  have_inliers_.clear();
  for (image_pair_t id : database_->ReadPairIds(false)) have_matches_.insert(id);
  for (image_pair_t id : database_->ReadPairIds(true)) have_inliers_.insert(id);
}

std::vector<image_t> FeatureMatcherCache::GetImageIds() const {
  std::vector<image_t> ids;
  ids.reserve(images_cache_.size());
  for (const auto& kv : images_cache_) ids.push_back(kv.first);
  std::sort(ids.begin(), ids.end());
  return ids;
}

const FeatureKeypoints& FeatureMatcherCache::GetKeypoints(image_t image_id) {
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = keypoints_cache_.find(image_id);
  if (it == keypoints_cache_.end()) it = keypoints_cache_.emplace(image_id, database_->ReadKeypoints(image_id)).first;
  return it->second;
}
const FeatureDescriptors& FeatureMatcherCache::GetDescriptors(image_t image_id) {
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = descriptors_cache_.find(image_id);
  if (it == descriptors_cache_.end()) it = descriptors_cache_.emplace(image_id, database_->ReadDescriptors(image_id)).first;
  return it->second;
}

SiftFeatureMatcher::SiftFeatureMatcher(const SiftMatchingOptions& options, Database* database, FeatureMatcherCache* cache)
    : options_(options), database_(database), cache_(cache) {
  if (!options_.Check()) throw std::invalid_argument("SiftMatchingOptions::Check failed");  // CHECK(options_.Check())
}

SiftFeatureMatcher::~SiftFeatureMatcher() {
  if (writer_.joinable()) writer_.join();  // errors of a never-flushed write-back are lost with the object
  if (ctx_) dsm_ctx_destroy(ctx_);
}

void SiftFeatureMatcher::Flush() {
  if (writer_.joinable()) writer_.join();
  if (writer_error_) {
    std::exception_ptr e = writer_error_;
    writer_error_ = nullptr;
    std::rethrow_exception(e);
  }
}

bool SiftFeatureMatcher::Setup() {
  int device = 0;
  if (options_.gpu_index != "-1" && !options_.gpu_index.empty()) device = std::atoi(options_.gpu_index.c_str());
  const int rc = dsm_ctx_create(device, &ctx_);
  if (rc != DSM_OK) {
    last_error_ = dsm_last_error(nullptr);
    return false;
  }
  is_setup_ = true;
  return true;
}

bool SiftFeatureMatcher::UploadImages() {
  image_ids_ = cache_->GetImageIds();
  image_index_.clear();
  const uint32_t n = static_cast<uint32_t>(image_ids_.size());
  std::vector<uint32_t> nfeat(n);
  std::vector<const uint8_t*> desc(n);
  std::vector<const float*> kp(n);
  std::vector<dsm_camera> cams(n);
  for (uint32_t i = 0; i < n; ++i) {
    const image_t id = image_ids_[i];
    image_index_[id] = i;
    const FeatureDescriptors& d = cache_->GetDescriptors(id);
    const FeatureKeypoints& k = cache_->GetKeypoints(id);
    if (d.rows != k.size() || (d.rows && d.cols != 128)) {
      last_error_ = "keypoints/descriptors mismatch for image " + std::to_string(id);
      return false;
    }
    nfeat[i] = static_cast<uint32_t>(d.rows);
    desc[i] = d.data.data();
    kp[i] = k.empty() ? nullptr : &k[0].x;
    const Camera& c = cache_->GetCamera(cache_->GetImage(id).camera_id);
    std::memset(&cams[i], 0, sizeof(dsm_camera));
    cams[i].model_id = c.model_id;
    cams[i].has_prior_focal_length = c.HasPriorFocalLength() ? 1 : 0;
    cams[i].width = c.width;
    cams[i].height = c.height;
    for (size_t p = 0; p < c.params.size() && p < 12; ++p) cams[i].params[p] = c.params[p];
  }
  // FeatureKeypoint is 6 floats (x, y, a11, a12, a21, a22): stride 6
  static_assert(sizeof(FeatureKeypoint) == 6 * sizeof(float), "FeatureKeypoint layout");
  std::vector<float> dummy(2, 0.f);
  for (uint32_t i = 0; i < n; ++i)
    if (!kp[i]) kp[i] = dummy.data();
  const int rc = dsm_set_images(ctx_, n, nfeat.data(), desc.data(), kp.data(), 6, cams.data());
  if (rc != DSM_OK) {
    last_error_ = dsm_last_error(ctx_);
    return false;
  }
  images_uploaded_ = true;
  return true;
}

void SiftFeatureMatcher::Match(const std::vector<std::pair<image_t, image_t>>& image_pairs) {
  if (!database_ || !cache_ || !is_setup_) throw std::logic_error("SiftFeatureMatcher::Match before Setup");  // CHECKs :751-753
  if (image_pairs.empty()) return;
  if (!images_uploaded_ && !UploadImages()) throw std::runtime_error(last_error_);

  // ---- dedupe, resume semantics (matching.cc:763-813)
  std::unordered_set<image_pair_t> seen;
  std::vector<std::pair<image_t, image_t>> to_match, to_verify_only;
  std::vector<FeatureMatches> existing;
  {
    const auto lock = cache_->Lock();  // one acquisition for the whole list (the write-back thread may be running)
    for (const auto& pr : image_pairs) {
      if (pr.first == pr.second) continue;
      const image_pair_t pair_id = Database::ImagePairToPairId(pr.first, pr.second);
      if (!seen.insert(pair_id).second) continue;
      const bool exists_matches = cache_->ExistsMatchesUnlocked(pr.first, pr.second);
      const bool exists_inlier_matches = cache_->ExistsInlierMatchesUnlocked(pr.first, pr.second);
      if (exists_matches && exists_inlier_matches) continue;
      if (exists_inlier_matches) cache_->DeleteInlierMatchesUnlocked(pr.first, pr.second);
      if (exists_matches) {
        existing.push_back(cache_->GetMatchesUnlocked(pr.first, pr.second));
        cache_->DeleteMatchesUnlocked(pr.first, pr.second);
        to_verify_only.push_back(pr);
      } else {
        to_match.push_back(pr);
      }
      if (options_.async_write_back) cache_->MarkPendingUnlocked(pr.first, pr.second);
    }
  }
  dsm_match_options mo;
  dsm_default_match_options(&mo);
  mo.max_ratio = options_.max_ratio;
  mo.max_distance = options_.max_distance;
  mo.cross_check = options_.cross_check ? 1 : 0;
  mo.max_num_matches = options_.max_num_matches;
  dsm_two_view_options to;
  dsm_default_two_view_options(&to);  // TwoViewGeometryVerifier ctor, matching.cc:559-568
  to.min_num_inliers = static_cast<uint64_t>(options_.min_num_inliers);
  to.max_error = options_.max_error;
  to.confidence = options_.confidence;
  to.min_num_trials = static_cast<uint64_t>(options_.min_num_trials);
  to.max_num_trials = static_cast<uint64_t>(options_.max_num_trials);
  to.min_inlier_ratio = options_.min_inlier_ratio;
  to.multiple_models = options_.multiple_models ? 1 : 0;  // TwoViewGeometry::Options::multiple_ignore_watermark stays at its default (true)

  auto run = [&](const std::vector<std::pair<image_t, image_t>>& prs, const std::vector<FeatureMatches>* given) {
    if (prs.empty()) return;
    const uint32_t np = static_cast<uint32_t>(prs.size());
    std::vector<uint32_t> idx(2 * static_cast<size_t>(np)), seeds(np);
    for (uint32_t i = 0; i < np; ++i) {
      idx[2 * i] = image_index_.at(prs[i].first);
      idx[2 * i + 1] = image_index_.at(prs[i].second);
      seeds[i] = dsm_pair_seed(prs[i].first, prs[i].second, options_.random_seed);
    }
    int rc;
    if (given) {
      std::vector<uint64_t> off(np + 1, 0);
      for (uint32_t i = 0; i < np; ++i) off[i + 1] = off[i] + (*given)[i].size();
      std::vector<uint32_t> flat(2 * off[np]);
      for (uint32_t i = 0; i < np; ++i)
        for (size_t k = 0; k < (*given)[i].size(); ++k) {
          flat[2 * (off[i] + k)] = (*given)[i][k].point2D_idx1;
          flat[2 * (off[i] + k) + 1] = (*given)[i][k].point2D_idx2;
        }
      rc = dsm_set_matches(ctx_, np, idx.data(), off.data(), flat.data());
    } else {
      rc = dsm_match_pairs(ctx_, np, idx.data(), &mo);
    }
    // guided_matching (matching.cc:647-667): verifier -> guided matcher -> output; the post-filter then sees the guided counts
    if (rc == DSM_OK) rc = dsm_verify_pairs(ctx_, &to, seeds.data(), 0, options_.guided_matching ? 0 : 1);
    if (rc == DSM_OK && options_.guided_matching) rc = dsm_guided_match_pairs(ctx_, &mo, &to, 1);
    if (rc != DSM_OK) throw std::runtime_error(std::string("device matching failed: ") + dsm_last_error(ctx_));
    std::vector<uint64_t> moff(np + 1), ioff(np + 1);
    rc = dsm_get_matches(ctx_, moff.data(), nullptr, 0);
    std::vector<uint32_t> m(2 * std::max<uint64_t>(moff[np], 1));
    if (rc == DSM_OK) rc = dsm_get_matches(ctx_, nullptr, m.data(), moff[np]);
    std::vector<dsm_two_view_geometry> tv(np);
    if (rc == DSM_OK) rc = dsm_get_two_view_geometries(ctx_, tv.data());
    if (rc == DSM_OK) rc = dsm_get_inlier_matches(ctx_, ioff.data(), nullptr, 0);
    std::vector<uint32_t> im(2 * std::max<uint64_t>(ioff[np], 1));
    if (rc == DSM_OK) rc = dsm_get_inlier_matches(ctx_, nullptr, im.data(), ioff[np]);
    if (rc != DSM_OK) throw std::runtime_error(std::string("result fetch failed: ") + dsm_last_error(ctx_));
    // ---- write results (matching.cc:819-836), on this thread or handed to the write-back thread
    const int min_num_inliers = options_.min_num_inliers;
    FeatureMatcherCache* cache = cache_;
    auto write = [cache, min_num_inliers, np, prs, moff = std::move(moff), m = std::move(m), tv = std::move(tv),
                  ioff = std::move(ioff), im = std::move(im)]() {
      for (uint32_t i = 0; i < np; ++i) {
        FeatureMatches matches(moff[i + 1] - moff[i]);
        for (size_t k = 0; k < matches.size(); ++k) matches[k] = FeatureMatch(m[2 * (moff[i] + k)], m[2 * (moff[i] + k) + 1]);
        if (matches.size() < static_cast<size_t>(min_num_inliers)) matches.clear();
        TwoViewGeometry t;  // stays TwoViewGeometry() when the device post-filter zeroed the pair
        if (tv[i].num_inliers >= static_cast<uint32_t>(min_num_inliers) && tv[i].num_inliers > 0) {
          t.config = tv[i].config;
          std::memcpy(t.E, tv[i].E, sizeof(t.E));
          std::memcpy(t.F, tv[i].F, sizeof(t.F));
          std::memcpy(t.H, tv[i].H, sizeof(t.H));
          std::memcpy(t.qvec, tv[i].qvec, sizeof(t.qvec));
          std::memcpy(t.tvec, tv[i].tvec, sizeof(t.tvec));
          t.tri_angle = tv[i].tri_angle;
          t.inlier_matches.resize(ioff[i + 1] - ioff[i]);
          for (size_t k = 0; k < t.inlier_matches.size(); ++k)
            t.inlier_matches[k] = FeatureMatch(im[2 * (ioff[i] + k)], im[2 * (ioff[i] + k) + 1]);
        }
        cache->WriteMatches(prs[i].first, prs[i].second, matches);
        cache->WriteTwoViewGeometry(prs[i].first, prs[i].second, t);
      }
    };
    if (!options_.async_write_back) {
      write();
    } else {
      Flush();  // one write-back in flight
      writer_ = std::thread([this, cache, write = std::move(write)]() {
        try {
          cache->BeginTransaction();
          write();
          cache->EndTransaction();
        } catch (...) {
          writer_error_ = std::current_exception();
        }
      });
    }
  };
  run(to_match, nullptr);
  run(to_verify_only, &existing);
}

ExhaustiveFeatureMatcher::ExhaustiveFeatureMatcher(const ExhaustiveMatchingOptions& options,
                                                   const SiftMatchingOptions& match_options, const std::string& database_path)
    : options_(options),
      match_options_(match_options),
      database_(database_path),
      cache_(5 * options_.block_size, &database_),
      matcher_(match_options, &database_, &cache_) {
  if (!options_.Check()) throw std::invalid_argument("ExhaustiveMatchingOptions::Check failed");
}

bool ExhaustiveFeatureMatcher::Run() {
  if (!matcher_.Setup()) {
    std::cerr << "ERROR: " << matcher_.LastError() << std::endl;
    return false;
  }
  cache_.Setup();
  const std::vector<image_t> image_ids = cache_.GetImageIds();
  const size_t block_size = static_cast<size_t>(options_.block_size);
  const size_t num_blocks = (image_ids.size() + block_size - 1) / block_size;
  std::vector<std::pair<image_t, image_t>> image_pairs;
  for (size_t start_idx1 = 0; start_idx1 < image_ids.size(); start_idx1 += block_size) {
    const size_t end_idx1 = std::min(image_ids.size(), start_idx1 + block_size) - 1;
    for (size_t start_idx2 = 0; start_idx2 < image_ids.size(); start_idx2 += block_size) {
      const size_t end_idx2 = std::min(image_ids.size(), start_idx2 + block_size) - 1;
      (void)num_blocks;
      image_pairs.clear();
      for (size_t idx1 = start_idx1; idx1 <= end_idx1; ++idx1) {
        for (size_t idx2 = start_idx2; idx2 <= end_idx2; ++idx2) {
          const size_t block_id1 = idx1 % block_size;
          const size_t block_id2 = idx2 % block_size;
          if ((idx1 > idx2 && block_id1 <= block_id2) || (idx1 < idx2 && block_id1 < block_id2)) {  // matching.cc:899-901
            image_pairs.emplace_back(image_ids[idx1], image_ids[idx2]);
          }
        }
      }
      if (match_options_.async_write_back) {  // the write-back thread owns the transaction of its rows
        matcher_.Match(image_pairs);
      } else {
        DatabaseTransaction database_transaction(&database_);
        matcher_.Match(image_pairs);
      }
    }
  }
  matcher_.Flush();
  return true;
}

}  // namespace dagsfm_amd

// ---------------------------------------------------------------------------------------- flat C API
// Thin C exports over the classes above for the Python tests and the CLI.
using namespace dagsfm_amd;

extern "C" {

// Runs ExhaustiveFeatureMatcher over database_path.  Returns 0 on success.
int dsm_host_exhaustive_matcher_ex(const char* database_path, int block_size, int use_prior_defaults, uint32_t random_seed,
                                   double max_ratio, double max_distance, int cross_check, int min_num_inliers,
                                   int guided_matching, int multiple_models) {
  try {
    ExhaustiveMatchingOptions eo;
    eo.block_size = block_size;
    SiftMatchingOptions mo;
    if (!use_prior_defaults) {
      mo.max_ratio = max_ratio;
      mo.max_distance = max_distance;
      mo.cross_check = cross_check != 0;
      mo.min_num_inliers = min_num_inliers;
    }
    mo.guided_matching = guided_matching != 0;
    mo.multiple_models = multiple_models != 0;
    mo.async_write_back = std::getenv("DSM_ASYNC_WRITE_BACK") != nullptr;  // CLI / tests: overlap SQLite with the device
    mo.random_seed = random_seed;
    ExhaustiveFeatureMatcher m(eo, mo, database_path);
    return m.Run() ? 0 : 2;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return 1;
  }
}

int dsm_host_exhaustive_matcher(const char* database_path, int block_size, int use_prior_defaults, uint32_t random_seed,
                                double max_ratio, double max_distance, int cross_check, int min_num_inliers) {
  return dsm_host_exhaustive_matcher_ex(database_path, block_size, use_prior_defaults, random_seed, max_ratio, max_distance,
                                        cross_check, min_num_inliers, 0, 0);
}

// Write-back micro-benchmark: n_pairs synthetic matches + two_view_geometries rows in ONE
// transaction through the same Database calls SiftFeatureMatcher::Match uses.  Returns pairs per second.
double dsm_host_db_bulk_write_bench(const char* database_path, uint32_t n_pairs, uint32_t n_matches, uint32_t n_inliers) {
  try {
    Database db(database_path);
    FeatureMatches m(n_matches), inl(n_inliers);
    for (uint32_t i = 0; i < n_matches; ++i) m[i] = FeatureMatch(i, n_matches - 1 - i);
    for (uint32_t i = 0; i < n_inliers; ++i) inl[i] = m[i];
    TwoViewGeometry t;
    t.config = 2;
    t.qvec[0] = 1;
    t.inlier_matches = inl;
    FeatureMatcherCache cache(100, &db);
    cache.Setup();
    const auto t0 = std::chrono::steady_clock::now();
    {
      DatabaseTransaction tr(&db);
      for (uint32_t k = 0; k < n_pairs; ++k) {
        const image_t a = 1 + k / 1000, b = 2000 + k % 1000;
        if (cache.ExistsMatches(a, b) || cache.ExistsInlierMatches(a, b)) continue;
        cache.WriteMatches(a, b, m);
        cache.WriteTwoViewGeometry(a, b, t);
      }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return n_pairs / dt;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return -1.0;
  }
}

// Database round trip used by the CPU-only tests (mirrors base/database_test.cc:283-360).
int dsm_host_db_write_pair(const char* database_path, uint32_t image_id1, uint32_t image_id2, const uint32_t* matches,
                           uint32_t n_matches, int config, const double* qvec, const double* tvec, const uint32_t* inliers,
                           uint32_t n_inliers) {
  try {
    Database db(database_path);
    FeatureMatches m(n_matches);
    for (uint32_t i = 0; i < n_matches; ++i) m[i] = FeatureMatch(matches[2 * i], matches[2 * i + 1]);
    TwoViewGeometry t;
    t.config = config;
    for (int i = 0; i < 4; ++i) t.qvec[i] = qvec[i];
    for (int i = 0; i < 3; ++i) t.tvec[i] = tvec[i];
    t.inlier_matches.resize(n_inliers);
    for (uint32_t i = 0; i < n_inliers; ++i) t.inlier_matches[i] = FeatureMatch(inliers[2 * i], inliers[2 * i + 1]);
    db.WriteMatches(image_id1, image_id2, m);
    db.WriteTwoViewGeometry(image_id1, image_id2, t);
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return 1;
  }
}

int dsm_host_db_read_pair(const char* database_path, uint32_t image_id1, uint32_t image_id2, uint32_t* matches,
                          uint32_t* n_matches, int* config, double* qvec, double* tvec, uint32_t* inliers, uint32_t* n_inliers,
                          uint32_t capacity) {
  try {
    Database db(database_path);
    const FeatureMatches m = db.ReadMatches(image_id1, image_id2);
    const TwoViewGeometry t = db.ReadTwoViewGeometry(image_id1, image_id2);
    if (m.size() > capacity || t.inlier_matches.size() > capacity) return 3;
    *n_matches = static_cast<uint32_t>(m.size());
    for (size_t i = 0; i < m.size(); ++i) {
      matches[2 * i] = m[i].point2D_idx1;
      matches[2 * i + 1] = m[i].point2D_idx2;
    }
    *config = t.config;
    for (int i = 0; i < 4; ++i) qvec[i] = t.qvec[i];
    for (int i = 0; i < 3; ++i) tvec[i] = t.tvec[i];
    *n_inliers = static_cast<uint32_t>(t.inlier_matches.size());
    for (size_t i = 0; i < t.inlier_matches.size(); ++i) {
      inliers[2 * i] = t.inlier_matches[i].point2D_idx1;
      inliers[2 * i + 1] = t.inlier_matches[i].point2D_idx2;
    }
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return 1;
  }
}

uint64_t dsm_host_image_pair_to_pair_id(uint32_t a, uint32_t b) { return Database::ImagePairToPairId(a, b); }

}  // extern "C"
