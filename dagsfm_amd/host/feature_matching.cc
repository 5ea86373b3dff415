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

FeatureMatcherCache::Features& FeatureMatcherCache::Touch(image_t image_id) {
  auto it = features_.find(image_id);
  if (it != features_.end()) {
    lru_.splice(lru_.begin(), lru_, it->second.lru_it);
  } else {
    // LRUCache of cache_size images (util/cache.h), except that an image requested since the last ReleasePins()
    // is never evicted: Match() holds pointers into the cache while it uploads
    if (features_.size() >= std::max<size_t>(cache_size_, 1)) {
      for (auto rit = lru_.end(); rit != lru_.begin();) {
        --rit;
        auto victim = features_.find(*rit);
        if (!victim->second.pinned) {
          lru_.erase(rit);
          features_.erase(victim);
          break;
        }
      }
    }
    lru_.push_front(image_id);
    it = features_.emplace(image_id, Features()).first;
    it->second.lru_it = lru_.begin();
  }
  it->second.pinned = true;
  return it->second;
}

void FeatureMatcherCache::ReleasePins() {
  std::lock_guard<std::mutex> lock(mutex_);
  for (auto& kv : features_) kv.second.pinned = false;
  while (features_.size() > std::max<size_t>(cache_size_, 1) && !lru_.empty()) {
    features_.erase(lru_.back());
    lru_.pop_back();
  }
}

const FeatureKeypoints& FeatureMatcherCache::GetKeypoints(image_t image_id) {
  std::lock_guard<std::mutex> lock(mutex_);
  Features& f = Touch(image_id);
  if (!f.have_kp) {
    f.keypoints = database_->ReadKeypoints(image_id);
    f.have_kp = true;
  }
  return f.keypoints;
}
const FeatureDescriptors& FeatureMatcherCache::GetDescriptors(image_t image_id) {
  std::lock_guard<std::mutex> lock(mutex_);
  Features& f = Touch(image_id);
  if (!f.have_desc) {
    f.descriptors = database_->ReadDescriptors(image_id);
    f.have_desc = true;
  }
  return f.descriptors;
}

SiftFeatureMatcher::SiftFeatureMatcher(const SiftMatchingOptions& options, Database* database, FeatureMatcherCache* cache)
    : options_(options), database_(database), cache_(cache) {
  if (!options_.Check()) throw std::invalid_argument("SiftMatchingOptions::Check failed");  // CHECK(options_.Check())
}

SiftFeatureMatcher::~SiftFeatureMatcher() {
  if (writer_.joinable()) writer_.join();  // errors of a never-flushed write-back are lost with the object
  for (dsm_ctx* c : ctxs_) dsm_ctx_destroy(c);
}

void SiftFeatureMatcher::Flush() {
  if (writer_.joinable()) writer_.join();
  if (writer_error_) {
    std::exception_ptr e = writer_error_;
    writer_error_ = nullptr;
    std::rethrow_exception(e);
  }
}

// gpu_index: "-1" = one matcher per visible device, otherwise a comma-separated device list
// (matching.cc:631-645: one SiftGPUFeatureMatcher thread per CUDA device; CSVToVector<int>(gpu_index)).
bool SiftFeatureMatcher::Setup() {
  devices_.clear();
  if (options_.gpu_index.empty() || options_.gpu_index == "-1") {
    const int n = dsm_device_count();
    for (int d = 0; d < n; ++d) devices_.push_back(d);
  } else {
    size_t pos = 0;
    while (pos <= options_.gpu_index.size()) {
      const size_t comma = options_.gpu_index.find(',', pos);
      const std::string tok = options_.gpu_index.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
      if (!tok.empty()) devices_.push_back(std::atoi(tok.c_str()));
      if (comma == std::string::npos) break;
      pos = comma + 1;
    }
  }
  if (devices_.empty()) {
    last_error_ = "no HIP device visible";
    return false;  // matching.cc:732-742
  }
  for (int d : devices_) {
    dsm_ctx* c = nullptr;
    if (dsm_ctx_create(d, &c) != DSM_OK) {
      last_error_ = dsm_last_error(nullptr);
      for (dsm_ctx* x : ctxs_) dsm_ctx_destroy(x);
      ctxs_.clear();
      return false;
    }
    ctxs_.push_back(c);
  }
  max_resident_ = cache_ ? cache_->CacheSize() : 0;
  is_setup_ = true;
  return true;
}

bool SiftFeatureMatcher::EnsureResident(const std::vector<std::pair<image_t, image_t>>& a,
                                        const std::vector<std::pair<image_t, image_t>>& b) {
  std::vector<image_t> needed;
  bool all_there = true;
  {
    std::unordered_set<image_t> seen;
    for (const auto* list : {&a, &b})
      for (const auto& pr : *list)
        for (image_t id : {pr.first, pr.second})
          if (seen.insert(id).second) {
            needed.push_back(id);
            if (!image_index_.count(id)) all_there = false;
          }
  }
  if (all_there) return true;
  // keep the resident images too while the union stays within the cache size (consecutive blocks of the
  // exhaustive / sequential matchers share most of their images)
  std::vector<image_t> ids = needed;
  if (image_ids_.size() + needed.size() <= std::max(max_resident_, needed.size())) {
    std::unordered_set<image_t> in(needed.begin(), needed.end());
    for (image_t id : image_ids_)
      if (in.insert(id).second) ids.push_back(id);
  }
  std::sort(ids.begin(), ids.end());
  const uint32_t n = static_cast<uint32_t>(ids.size());
  std::vector<uint32_t> nfeat(n);
  std::vector<const uint8_t*> desc(n);
  std::vector<const float*> kp(n);
  std::vector<dsm_camera> cams(n);
  for (uint32_t i = 0; i < n; ++i) {
    const image_t id = ids[i];
    const FeatureDescriptors& d = cache_->GetDescriptors(id);
    const FeatureKeypoints& k = cache_->GetKeypoints(id);
    if (d.rows != k.size() || (d.rows && d.cols != 128)) {
      last_error_ = "keypoints/descriptors mismatch for image " + std::to_string(id);
      cache_->ReleasePins();
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
  // every device gets every image of the list (the pair list is what is sharded), in parallel
  std::vector<int> rcs(ctxs_.size(), DSM_OK);
  std::vector<std::thread> th;
  for (size_t d = 0; d < ctxs_.size(); ++d)
    th.emplace_back([&, d]() { rcs[d] = dsm_set_images(ctxs_[d], n, nfeat.data(), desc.data(), kp.data(), 6, cams.data()); });
  for (auto& t : th) t.join();
  cache_->ReleasePins();
  for (size_t d = 0; d < ctxs_.size(); ++d)
    if (rcs[d] != DSM_OK) {
      last_error_ = dsm_last_error(ctxs_[d]);  // e.g. a camera model id the reference does not know either
      image_ids_.clear();
      image_index_.clear();
      return false;
    }
  image_ids_ = ids;
  image_nfeat_ = nfeat;
  image_index_.clear();
  for (uint32_t i = 0; i < n; ++i) image_index_[ids[i]] = i;
  return true;
}

void SiftFeatureMatcher::Match(const std::vector<std::pair<image_t, image_t>>& image_pairs) {
  if (!database_ || !cache_ || !is_setup_) throw std::logic_error("SiftFeatureMatcher::Match before Setup");  // CHECKs :751-753
  if (image_pairs.empty()) return;

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
    }
  }
  if (!EnsureResident(to_match, to_verify_only)) throw std::runtime_error(last_error_);
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

  // One share of the pair list on one device: match (or install the given matches), verify, fetch.
  struct Share {
    uint32_t begin = 0, end = 0;
    std::vector<uint64_t> moff, ioff;
    std::vector<uint32_t> m, im;
    std::vector<dsm_two_view_geometry> tv;
    std::string error;
  };
  auto run_share = [&](dsm_ctx* ctx, const std::vector<std::pair<image_t, image_t>>& prs, const std::vector<FeatureMatches>* given,
                       Share* sh) {
    const uint32_t np = sh->end - sh->begin;
    if (np == 0) return;
    std::vector<uint32_t> idx(2 * static_cast<size_t>(np)), seeds(np);
    for (uint32_t i = 0; i < np; ++i) {
      const auto& pr = prs[sh->begin + i];
      idx[2 * i] = image_index_.at(pr.first);
      idx[2 * i + 1] = image_index_.at(pr.second);
      seeds[i] = dsm_pair_seed(pr.first, pr.second, options_.random_seed);
    }
    int rc;
    if (given) {
      std::vector<uint64_t> off(np + 1, 0);
      for (uint32_t i = 0; i < np; ++i) off[i + 1] = off[i] + (*given)[sh->begin + i].size();
      std::vector<uint32_t> flat(2 * off[np]);
      for (uint32_t i = 0; i < np; ++i) {
        const FeatureMatches& g = (*given)[sh->begin + i];
        for (size_t k = 0; k < g.size(); ++k) {
          flat[2 * (off[i] + k)] = g[k].point2D_idx1;
          flat[2 * (off[i] + k) + 1] = g[k].point2D_idx2;
        }
      }
      rc = dsm_set_matches(ctx, np, idx.data(), off.data(), flat.data());
    } else {
      rc = dsm_match_pairs(ctx, np, idx.data(), &mo);
    }
    // guided_matching (matching.cc:647-667): verifier -> guided matcher -> output; the post-filter then sees the guided counts
    if (rc == DSM_OK) rc = dsm_verify_pairs(ctx, &to, seeds.data(), 0, options_.guided_matching ? 0 : 1);
    if (rc == DSM_OK && options_.guided_matching) rc = dsm_guided_match_pairs(ctx, &mo, &to, 1);
    if (rc != DSM_OK) {
      sh->error = std::string("device matching failed: ") + dsm_last_error(ctx);
      return;
    }
    sh->moff.assign(np + 1, 0);
    sh->ioff.assign(np + 1, 0);
    rc = dsm_get_matches(ctx, sh->moff.data(), nullptr, 0);
    sh->m.assign(2 * std::max<uint64_t>(sh->moff[np], 1), 0);
    if (rc == DSM_OK) rc = dsm_get_matches(ctx, nullptr, sh->m.data(), sh->moff[np]);
    sh->tv.resize(np);
    if (rc == DSM_OK) rc = dsm_get_two_view_geometries(ctx, sh->tv.data());
    if (rc == DSM_OK) rc = dsm_get_inlier_matches(ctx, sh->ioff.data(), nullptr, 0);
    sh->im.assign(2 * std::max<uint64_t>(sh->ioff[np], 1), 0);
    if (rc == DSM_OK) rc = dsm_get_inlier_matches(ctx, nullptr, sh->im.data(), sh->ioff[np]);
    if (rc != DSM_OK) sh->error = std::string("result fetch failed: ") + dsm_last_error(ctx);
  };

  auto run = [&](const std::vector<std::pair<image_t, image_t>>& prs, const std::vector<FeatureMatches>* given) {
    if (prs.empty()) return;
    const uint32_t np = static_cast<uint32_t>(prs.size());
    // Contiguous blocks of the list, one per device, cut by cost (descriptor-matrix size + a per-pair term for
    // the verification): the reference lets its per-GPU matcher threads pull pairs from one queue
    // (matching.cc:640-645); a static cut by cost gives the same balance without a queue and keeps list order.
    const size_t nd = std::min<size_t>(ctxs_.size(), np);
    std::vector<double> cum(np + 1, 0.0);
    for (uint32_t i = 0; i < np; ++i) {
      const double n1 = image_nfeat_[image_index_.at(prs[i].first)], n2 = image_nfeat_[image_index_.at(prs[i].second)];
      cum[i + 1] = cum[i] + (given ? 0.0 : n1 * n2) + 4096.0 * 1024.0;
    }
    std::vector<Share> shares(nd);
    uint32_t at = 0;
    for (size_t d = 0; d < nd; ++d) {
      shares[d].begin = at;
      const double target = cum[np] * static_cast<double>(d + 1) / static_cast<double>(nd);
      while (at < np && (cum[at + 1] <= target || d + 1 == nd)) ++at;
      if (d + 1 < nd && at == shares[d].begin && at < np) ++at;  // never an empty share while pairs are left
      shares[d].end = at;
    }
    shares[nd - 1].end = np;
    if (nd == 1) {
      run_share(ctxs_[0], prs, given, &shares[0]);
    } else {
      std::vector<std::thread> th;
      for (size_t d = 0; d < nd; ++d) th.emplace_back([&, d]() { run_share(ctxs_[d], prs, given, &shares[d]); });
      for (auto& t : th) t.join();
    }
    for (const Share& sh : shares)
      if (!sh.error.empty()) throw std::runtime_error(sh.error);
    // merge the shares in list order
    std::vector<uint64_t> moff(np + 1, 0), ioff(np + 1, 0);
    std::vector<dsm_two_view_geometry> tv(np);
    uint64_t mt = 0, it = 0;
    for (const Share& sh : shares) {
      for (uint32_t i = sh.begin; i < sh.end; ++i) {
        moff[i] = mt + sh.moff[i - sh.begin];
        ioff[i] = it + sh.ioff[i - sh.begin];
        tv[i] = sh.tv[i - sh.begin];
      }
      if (sh.end > sh.begin) {
        mt += sh.moff[sh.end - sh.begin];
        it += sh.ioff[sh.end - sh.begin];
      }
    }
    moff[np] = mt;
    ioff[np] = it;
    std::vector<uint32_t> m(2 * std::max<uint64_t>(mt, 1)), im(2 * std::max<uint64_t>(it, 1));
    for (const Share& sh : shares) {
      if (sh.end == sh.begin) continue;
      std::copy(sh.m.begin(), sh.m.begin() + 2 * sh.moff[sh.end - sh.begin], m.begin() + 2 * moff[sh.begin]);
      std::copy(sh.im.begin(), sh.im.begin() + 2 * sh.ioff[sh.end - sh.begin], im.begin() + 2 * ioff[sh.begin]);
    }
    if (options_.async_write_back) {
      // the rows are on their way: later Match() calls must skip these pairs.  Marked only now, with the device
      // results on the host -- a failed device call above leaves the cache saying what the database says.
      const auto lock = cache_->Lock();
      for (const auto& pr : prs) cache_->MarkPendingUnlocked(pr.first, pr.second);
    }
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
        database_transaction.Commit();
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
    if (const char* g = std::getenv("DSM_GPU_INDEX")) mo.gpu_index = g;    // SiftMatchingOptions::gpu_index ("-1": all devices)
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

// CPU-only probe of FeatureMatcherCache's LRU: touches the given image ids in order (releasing the pins after every
// `pin_batch` requests, like one Match() call does) and returns the largest number of images the cache ever held.
int dsm_host_cache_lru_probe(const char* database_path, uint32_t cache_size, const uint32_t* image_ids, uint32_t n,
                             uint32_t pin_batch) {
  try {
    Database db(database_path);
    FeatureMatcherCache cache(cache_size, &db);
    cache.Setup();
    size_t peak = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const FeatureDescriptors& d = cache.GetDescriptors(image_ids[i]);
      const FeatureKeypoints& k = cache.GetKeypoints(image_ids[i]);
      if (d.rows != k.size()) return -2;
      peak = std::max(peak, cache.NumCachedImages());
      if ((i + 1) % std::max<uint32_t>(pin_batch, 1) == 0) cache.ReleasePins();
    }
    cache.ReleasePins();
    if (cache.NumCachedImages() > std::max<uint32_t>(cache_size, 1)) return -3;
    return static_cast<int>(peak);
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return -1;
  }
}

}  // extern "C"
