// feature_matching.cc -- see feature_matching.h.
#include "feature_matching.h"
#include "flann_index.h"
#include "spatial_verification.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>
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

void FeatureMatcherCache::RollbackTransaction(bool transaction_open) {
  std::lock_guard<std::recursive_mutex> lock(mutex_);
  if (transaction_open) database_->RollbackTransaction();
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
  std::lock_guard<std::recursive_mutex> lock(mutex_);
  for (auto& kv : features_) kv.second.pinned = false;
  while (features_.size() > std::max<size_t>(cache_size_, 1) && !lru_.empty()) {
    features_.erase(lru_.back());
    lru_.pop_back();
  }
}

const FeatureKeypoints& FeatureMatcherCache::GetKeypoints(image_t image_id) {
  std::lock_guard<std::recursive_mutex> lock(mutex_);
  Features& f = Touch(image_id);
  if (!f.have_kp) {
    f.keypoints = database_->ReadKeypoints(image_id);
    f.have_kp = true;
  }
  return f.keypoints;
}
const FeatureDescriptors& FeatureMatcherCache::GetDescriptors(image_t image_id) {
  std::lock_guard<std::recursive_mutex> lock(mutex_);
  Features& f = Touch(image_id);
  if (!f.have_desc) {
    f.descriptors = database_->ReadDescriptors(image_id);
    f.have_desc = true;
  }
  return f.descriptors;
}

// Run() owns its block loop and Flush()es at its end, so its matcher may leave the last slice of a block in flight while the
// next block is on the devices (SiftMatchingOptions::defer_write_back); the transactions are then the writer thread's.
static SiftMatchingOptions RunLoopOptions(SiftMatchingOptions o) {
  o.defer_write_back = o.async_write_back;
  return o;
}

ExhaustiveFeatureMatcher::ExhaustiveFeatureMatcher(const ExhaustiveMatchingOptions& options,
                                                   const SiftMatchingOptions& match_options, const std::string& database_path)
    : options_(options),
      match_options_(match_options),
      database_(database_path),
      cache_(5 * options_.block_size, &database_),
      matcher_(RunLoopOptions(match_options), &database_, &cache_) {
  if (!options_.Check()) throw std::invalid_argument("ExhaustiveMatchingOptions::Check failed");
}

bool ExhaustiveFeatureMatcher::Run() {
  // The reference's order is matcher_.Setup(); cache_.Setup() (matching.cc:858-863).  Setup() here creates the device
  // contexts (HIP runtime start-up, code objects: a few hundred ms of a process that matches 500 images in about one second),
  // the cache's set-up and the first block's features are SQLite reads: the two run side by side.
  const auto t_setup = std::chrono::steady_clock::now();
  bool setup_ok = false;
  if (!options_.overlap_setup) setup_ok = matcher_.Setup();
  std::thread setup_thread([this, &setup_ok]() {
    try {  // (an exception must not leave a thread: Setup() reports through its return value, this is for std::bad_alloc and the like)
      if (options_.overlap_setup) setup_ok = matcher_.Setup();
    } catch (...) {
      setup_ok = false;
    }
  });
  struct Joiner {
    std::thread* t;
    ~Joiner() {
      if (t->joinable()) t->join();
    }
  } setup_joiner{&setup_thread};
  cache_.Setup();
  const std::vector<image_t> image_ids = cache_.GetImageIds();
  {
    const size_t first_block = std::min<size_t>({image_ids.size(), static_cast<size_t>(std::max(options_.block_size, 1)), cache_.CacheSize()});
    for (size_t i = 0; i < first_block; ++i) {  // what the first Match() asks for first; stays cached (LRU), pinned only until here
      cache_.GetKeypoints(image_ids[i]);
      cache_.GetDescriptors(image_ids[i]);
    }
    cache_.ReleasePins();
  }
  setup_thread.join();
  setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup).count();
  if (!setup_ok) {
    std::cerr << "ERROR: " << matcher_.LastError() << std::endl;
    return false;
  }
  const auto t_run = std::chrono::steady_clock::now();
  struct JournalGuard {  // WAL again however Run() ends
    const Database* db;
    SiftFeatureMatcher* matcher;
    ~JournalGuard() {
      if (db) {
        // on the exception path a writer thread may still hold an open transaction on this connection: leaving the
        // rollback-journal mode rewrites the file header and fails inside a transaction, so the writer is
        // joined first; its own error, if any, has been or will be reported by whoever unwinds
        try {
          matcher->Flush();
        } catch (...) {
        }
        try {
          db->SetBulkLoadJournal(false);
        } catch (...) {
        }
      }
    }
  } journal_guard{match_options_.bulk_load_journal ? &database_ : nullptr, &matcher_};
  if (match_options_.bulk_load_journal) database_.SetBulkLoadJournal(true);
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
      if (match_options_.async_write_back) {
        // the write-back thread owns the transaction of its rows, one per slice: a failure leaves the earlier slices of
        // the block committed (complete rows of pairs a re-run skips) and rolls back the failing slice only
        matcher_.Match(image_pairs);
      } else {
        DatabaseTransaction database_transaction(&database_);
        try {
          matcher_.Match(image_pairs);
        } catch (...) {
          try {
            database_transaction.Rollback();  // nothing of a failed block is committed ...
          } catch (...) {  // (a failing ROLLBACK must neither hide the original error nor skip the cache)
          }
          try {
            cache_.RollbackTransaction(false);  // ... and the cache's pair-id sets follow the database
          } catch (...) {
          }
          throw;
        }
        database_transaction.Commit();
      }
    }
  }
  matcher_.Flush();
  run_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run).count();
  return true;
}

// ---------------------------------------------------------------------------------------- retrieval
bool VocabularyFile::ReadReferenceLayout(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long fsize = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> buf(fsize > 0 ? static_cast<size_t>(fsize) : 0);
  const bool read_ok = !buf.empty() && std::fread(buf.data(), 1, buf.size(), f) == buf.size();
  std::fclose(f);
  if (!read_ok || buf.size() < 16) return false;
  uint64_t rows = 0, cols = 0;
  std::memcpy(&rows, buf.data(), 8);
  std::memcpy(&cols, buf.data() + 8, 8);
  if (cols != 128 || rows == 0 || rows > 0x7fffffffull || 16 + rows * cols > buf.size()) return false;
  const size_t words_end = 16 + static_cast<size_t>(rows * cols);
  // parse the inverted index from `at`; true when it ends exactly at the end of the file
  auto parse = [&](size_t at, std::vector<float>* proj, std::vector<float>* thr) -> bool {
    const size_t proj_bytes = 64 * 128 * 4;
    if (at + 8 + proj_bytes > buf.size()) return false;
    if (proj) std::memcpy(proj->data(), buf.data() + at + 8, proj_bytes);
    size_t pos = at + 8 + proj_bytes;
    for (uint64_t w = 0; w < rows; ++w) {
      if (pos + 1 + 4 + 64 * 4 + 4 > buf.size()) return false;
      if (thr) std::memcpy(thr->data() + w * 64, buf.data() + pos + 5, 64 * 4);
      uint32_t n_entries = 0;
      std::memcpy(&n_entries, buf.data() + pos + 5 + 256, 4);
      pos += 1 + 4 + 256 + 4;
      const uint64_t entry_bytes = static_cast<uint64_t>(n_entries) * 32;  // int32 image, int32 feature, 4 f32 geometry, u64 signature
      if (entry_bytes > buf.size() - pos) return false;
      pos += static_cast<size_t>(entry_bytes);
    }
    if (pos + 4 > buf.size()) return false;
    int32_t num_images = 0;
    std::memcpy(&num_images, buf.data() + pos, 4);
    if (num_images < 0) return false;
    return pos + 4 + static_cast<uint64_t>(num_images) * 8 == buf.size();
  };
  auto accept = [&](size_t at) -> bool {
    num_words = static_cast<uint32_t>(rows);
    words.assign(buf.begin() + 16, buf.begin() + words_end);
    projection.resize(64 * 128);
    thresholds.resize(static_cast<size_t>(rows) * 64);
    index_begin = words_end;
    index_end = at;
    flann_blob.assign(buf.begin() + words_end, buf.begin() + at);
    return parse(at, &projection, &thresholds);
  };
  const int32_t key[2] = {static_cast<int32_t>(rows), 64};
  // The middle section is what flann::AutotunedIndex::saveIndex wrote (visual_index.h:600-607): two FLANN archives back to
  // back -- the autotuned index's own record, then the index it chose (lib/FLANN/algorithms/autotuned_index.h:209-217) --
  // each framed as header, first compressed block, (size, block)*, 0 (lib/FLANN/util/serialization.h:412-479).  Walking
  // that framing lands on the byte loadIndex stops at (:564-574), which is where the reference reads the inverted index.
  size_t at = words_end;
  flann_framed = FlannSkipArchive(buf.data(), buf.size(), &at) && FlannSkipArchive(buf.data(), buf.size(), &at);
  if (flann_framed)
    return at + 8 <= buf.size() && std::memcmp(buf.data() + at, key, 8) == 0 && parse(at, nullptr, nullptr) && accept(at);
  // not FLANN's v1.1 framing (a blob this reader cannot walk): the inverted index is located by its header --
  // (num_words, 64) -- at the one offset from which the rest of the file parses to exactly its end
  for (at = words_end; at + 8 <= buf.size(); ++at) {
    if (std::memcmp(buf.data() + at, key, 8) != 0) continue;
    if (!parse(at, nullptr, nullptr)) continue;
    return accept(at);
  }
  return false;
}

std::vector<uint32_t> TopScaleFeatureOrder(const FeatureKeypoints& keypoints, size_t num_features) {
  std::vector<uint32_t> order;
  if (keypoints.size() <= num_features || num_features == 0) {
    order.resize(keypoints.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<uint32_t>(i);
    return order;
  }
  std::vector<std::pair<size_t, float>> scales;
  scales.reserve(keypoints.size());
  for (size_t i = 0; i < keypoints.size(); ++i) {
    const FeatureKeypoint& k = keypoints[i];  // FeatureKeypoint::ComputeScale, src/feature/types.cc:84-94
    const float sx = std::sqrt(k.a11 * k.a11 + k.a21 * k.a21), sy = std::sqrt(k.a12 * k.a12 + k.a22 * k.a22);
    scales.emplace_back(i, (sx + sy) / 2.0f);
  }
  std::partial_sort(scales.begin(), scales.begin() + num_features, scales.end(),
                    [](const std::pair<size_t, float> scale1, const std::pair<size_t, float> scale2) { return scale1.second > scale2.second; });
  order.resize(num_features);
  for (size_t i = 0; i < num_features; ++i) order[i] = static_cast<uint32_t>(scales[i].first);
  return order;
}

bool VocabularyFile::Read(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[8];
  uint32_t hdr[2];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "DSMVOC1", 8) != 0) {  // not this library's flat file: the reference's layout
    std::fclose(f);
    return ReadReferenceLayout(path);
  }
  bool ok = std::fread(hdr, 4, 2, f) == 2 && hdr[0] > 0;
  if (ok) {  // the header's word count must be what the file holds: nothing is allocated on the word of a damaged file
    const long at = std::ftell(f);
    ok = at >= 0 && std::fseek(f, 0, SEEK_END) == 0;
    const long end = ok ? std::ftell(f) : -1;
    ok = ok && end >= at && static_cast<uint64_t>(end - at) == static_cast<uint64_t>(hdr[0]) * (128 + 64 * 4) + 64ull * 128 * 4 &&
         std::fseek(f, at, SEEK_SET) == 0;
  }
  if (ok) {
    num_words = hdr[0];
    words.resize(static_cast<size_t>(num_words) * 128);
    projection.resize(64 * 128);
    thresholds.resize(static_cast<size_t>(num_words) * 64);
    ok = std::fread(words.data(), 1, words.size(), f) == words.size() &&
         std::fread(projection.data(), 4, projection.size(), f) == projection.size() &&
         std::fread(thresholds.data(), 4, thresholds.size(), f) == thresholds.size();
  }
  std::fclose(f);
  return ok;
}
bool VocabularyFile::Write(const std::string& path) const {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const uint32_t hdr[2] = {num_words, 0};
  const bool ok = std::fwrite("DSMVOC1", 1, 8, f) == 8 && std::fwrite(hdr, 4, 2, f) == 2 &&
                  std::fwrite(words.data(), 1, words.size(), f) == words.size() &&
                  std::fwrite(projection.data(), 4, projection.size(), f) == projection.size() &&
                  std::fwrite(thresholds.data(), 4, thresholds.size(), f) == thresholds.size();
  std::fclose(f);
  return ok;
}

VocabSimilarityGraph::VocabSimilarityGraph(const VocabSimilaritySearchOptions& options, const Database& database)
    : options_(options), database_(&database), cache_(5 * static_cast<size_t>(options.num_images), &database) {}

bool VocabSimilarityGraph::Run() {
  image_pairs_.clear();
  scores_.clear();
  if (!options_.Check()) {
    last_error_ = "VocabSimilaritySearchOptions::Check failed (num_images > 0, a vocabulary path)";
    return false;
  }
  VocabularyFile voc;
  if (!voc.Read(options_.vocab_tree_path)) {
    last_error_ = "cannot read vocabulary " + options_.vocab_tree_path;
    return false;
  }
  cache_.Setup();
  const std::vector<image_t> ids = cache_.GetImageIds();
  const uint32_t n = static_cast<uint32_t>(ids.size());
  if (n == 0) return true;
  dsm_ctx* ctx = nullptr;
  if (dsm_ctx_create(0, &ctx) != DSM_OK) {
    last_error_ = dsm_last_error(nullptr);
    return false;
  }
  // descriptors only; copied, because the cache is an LRU (all images are indexed at once, like the reference's
  // IndexImagesInVisualIndex, similarity_graph.cpp:56-85)
  std::vector<std::vector<uint8_t>> copies(n);
  std::vector<uint32_t> nfeat(n);
  std::vector<const uint8_t*> desc(n);
  const bool verify = options_.num_images_after_verification > 0;
  std::vector<std::vector<FeatureGeometry>> geometries(verify ? n : 0);  // of the features as indexed, for the spatial re-ranking
  for (uint32_t i = 0; i < n; ++i) {
    const FeatureDescriptors& d = cache_.GetDescriptors(ids[i]);
    const bool top_scale = options_.max_num_features > 0 && d.rows > static_cast<size_t>(options_.max_num_features);
    if (top_scale || verify) {
      // ExtractTopScaleFeatures CHECK_EQs the two tables (src/feature/utils.cc:84); the indices of the keypoints address
      // rows of the descriptor blob below, and the geometries must belong to the indexed rows
      const size_t num_kps = cache_.GetKeypoints(ids[i]).size();
      if (num_kps != d.rows) {
        last_error_ = "image " + std::to_string(ids[i]) + ": " + std::to_string(num_kps) + " keypoints but " +
                      std::to_string(d.rows) + " descriptors";
        cache_.ReleasePins();
        dsm_ctx_destroy(ctx);
        return false;
      }
    }
    if (top_scale) {
      // similarity_graph.cpp:77-79, 137-141: index and query only the features of largest scale, in that order
      const FeatureKeypoints& kps = cache_.GetKeypoints(ids[i]);
      const std::vector<uint32_t> order = TopScaleFeatureOrder(kps, static_cast<size_t>(options_.max_num_features));
      copies[i].resize(order.size() * 128);
      for (size_t k = 0; k < order.size(); ++k) std::memcpy(copies[i].data() + k * 128, d.data.data() + static_cast<size_t>(order[k]) * 128, 128);
      nfeat[i] = static_cast<uint32_t>(order.size());
      if (verify)
        for (size_t k = 0; k < order.size(); ++k) geometries[i].push_back(GeometryOfKeypoint(kps[order[k]]));
    } else {
      copies[i] = d.data;
      nfeat[i] = static_cast<uint32_t>(d.rows);
      if (verify) {
        const FeatureKeypoints& kps = cache_.GetKeypoints(ids[i]);
        for (size_t k = 0; k < d.rows; ++k) geometries[i].push_back(GeometryOfKeypoint(kps[k]));
      }
    }
    desc[i] = copies[i].data();
    cache_.ReleasePins();
  }
  dsm_vocabulary v;
  v.num_words = voc.num_words;
  v.reserved = 0;
  v.words = voc.words.data();
  v.projection = voc.projection.data();
  v.thresholds = voc.thresholds.data();
  const uint32_t max_images = std::min<uint32_t>(static_cast<uint32_t>(options_.num_images), n);
  std::vector<uint32_t> counts(n), idx(static_cast<size_t>(n) * max_images);
  std::vector<float> sc(static_cast<size_t>(n) * max_images);
  int rc = dsm_set_images(ctx, n, nfeat.data(), desc.data(), nullptr, 0, nullptr);
  if (rc == DSM_OK) rc = dsm_retrieval_set_vocabulary(ctx, &v);
  VocabSimilaritySearchOptions::WordSearch word_search = options_.word_search;
  if (word_search == VocabSimilaritySearchOptions::kAuto)
    word_search = voc.flann_framed ? VocabSimilaritySearchOptions::kFlann : VocabSimilaritySearchOptions::kExact;
  word_search_used_ = word_search;
  FlannIndex flann;  // (outlives the dsm_retrieval_set_flann_index call that reads its arrays)
  if (rc == DSM_OK && (word_search == VocabSimilaritySearchOptions::kFlann || word_search == VocabSimilaritySearchOptions::kFlannHost)) {
    // the reference's own word ids: FindWordIds with 1 neighbour for VisualIndex::Add (IndexOptions::num_neighbors,
    // visual_index.h:62-73, 201-243; IndexImagesInVisualIndex passes the same num_checks, similarity_graph.cpp:56-85) and
    // with num_nearest_neighbors for the query, over the index loaded from the vocabulary file
    size_t at = 0;
    if (!voc.flann_framed || !flann.Load(voc.flann_blob.data(), voc.flann_blob.size(), &at, voc.words.data(), voc.num_words)) {
      last_error_ = "word_search = flann: " + (voc.flann_framed ? flann.error() : std::string("the vocabulary file carries no FLANN index"));
      dsm_ctx_destroy(ctx);
      return false;
    }
    const uint32_t kq = static_cast<uint32_t>(options_.num_nearest_neighbors);
    if (word_search == VocabSimilaritySearchOptions::kFlann) {
      // on the device: the trees go over once, dsm_retrieval_index / _query search them (csrc/flann_search.hip)
      dsm_flann_index flat;
      if (kq < 1 || !flann.Export(options_.num_checks, &flat)) {
        last_error_ = "word_search = flann: the search refused num_nearest_neighbors / num_checks";
        dsm_ctx_destroy(ctx);
        return false;
      }
      rc = dsm_retrieval_set_flann_index(ctx, &flat);
    } else {
      uint64_t total = 0;
      for (uint32_t i = 0; i < n; ++i) total += nfeat[i];
      std::vector<int32_t> index_ids(std::max<uint64_t>(total, 1)), query_ids(std::max<uint64_t>(total, 1) * kq);
      uint64_t f0 = 0;
      bool ok = kq >= 1;
      for (uint32_t i = 0; ok && i < n; ++i) {
        ok = flann.FindWordIds(desc[i], nfeat[i], 1, options_.num_checks, options_.num_threads, index_ids.data() + f0, nullptr) &&
             flann.FindWordIds(desc[i], nfeat[i], kq, options_.num_checks, options_.num_threads, query_ids.data() + f0 * kq, nullptr);
        f0 += nfeat[i];
      }
      if (!ok) {
        last_error_ = "word_search = flann: the search refused num_nearest_neighbors / num_checks";
        dsm_ctx_destroy(ctx);
        return false;
      }
      rc = dsm_retrieval_set_word_ids(ctx, index_ids.data(), kq, query_ids.data());
    }
  }
  if (rc == DSM_OK) rc = dsm_retrieval_index(ctx);
  if (rc == DSM_OK)
    rc = dsm_retrieval_query(ctx, static_cast<uint32_t>(options_.num_nearest_neighbors), max_images, counts.data(), idx.data(), sc.data());
  // spatial verification of the retrieved images and re-ranking (VisualIndex::Query with geometries, visual_index.h:259-500):
  // the device lists, per query, the database features that share a word with a query feature within the Hamming
  // threshold; the 1-to-1 assignment and VoteAndVerify of each retrieved image run here (spatial_verification.cc)
  if (rc == DSM_OK && verify) {
    std::vector<uint64_t> offsets(static_cast<size_t>(n) + 1, 0);
    rc = dsm_retrieval_matches(ctx, static_cast<uint32_t>(options_.num_nearest_neighbors), max_images, counts.data(), idx.data(), offsets.data());
    std::vector<uint32_t> tuples;
    std::vector<float> idf;
    std::string host_error;  // what a worker could not do (allocation, a tuple outside the indexed features)
    if (rc == DSM_OK) {
      try {
        tuples.resize(static_cast<size_t>(offsets[n]) * 5 + 1);
        idf.resize(voc.num_words);
      } catch (const std::exception& e) {
        host_error = std::string("spatial re-ranking: ") + e.what();
      }
    }
    if (rc == DSM_OK && host_error.empty()) rc = dsm_get_retrieval_matches(ctx, tuples.data(), offsets[n]);
    if (rc == DSM_OK && host_error.empty()) rc = dsm_get_retrieval_idf(ctx, idf.data(), voc.num_words);
    if (rc == DSM_OK && host_error.empty()) {
      float lut[65];  // HammingDistWeightFunctor<64, 16>, retrieval/utils.h:47-78
      for (int h = 0; h <= 64; ++h) {
        const float hamming_dist = static_cast<float>(h);
        lut[h] = hamming_dist <= 24 ? std::exp(-hamming_dist * hamming_dist / (16.0f * 16.0f)) : 0.0f;
      }
      // one query is independent of the next (the reference verifies on its retrieval thread pool, similarity_graph.cpp:
      // 116-160): options_.num_threads workers take the queries in turn
      std::mutex error_mutex;
      auto rerank_range = [&](uint32_t first, uint32_t step) {
       try {
        std::vector<RetrievalCandidate> candidates;
        for (uint32_t q = first; q < n; q += step) {
          candidates.clear();
          for (uint64_t m = offsets[q]; m < offsets[q + 1]; ++m) {
            const uint32_t* t = tuples.data() + m * 5;
            if (t[1] >= n || t[2] >= geometries[t[1]].size() || t[0] >= geometries[q].size() || (t[3] >> 8) >= idf.size())
              throw std::runtime_error("candidate tuple outside the indexed features");
            RetrievalCandidate c;
            c.query_feature = t[0];
            c.image = t[1];
            c.database_feature = t[2];
            c.entry_position = t[4];
            const float idf_weight = idf[t[3] >> 8];
            c.weight = lut[t[3] & 255u] * (idf_weight * idf_weight);
            c.database_geometry = geometries[c.image][c.database_feature];
            candidates.push_back(c);
          }
          counts[q] = SpatialRerank(geometries[q], candidates, options_.num_images_after_verification, counts[q],
                                    idx.data() + static_cast<size_t>(q) * max_images, sc.data() + static_cast<size_t>(q) * max_images);
        }
       } catch (const std::exception& e) {  // a worker thread must not std::terminate the host process
        std::lock_guard<std::mutex> lock(error_mutex);
        if (host_error.empty()) host_error = std::string("spatial re-ranking: ") + e.what();
       }
      };
      const uint32_t workers = std::max<uint32_t>(1, std::min<uint32_t>(n, static_cast<uint32_t>(std::max(1, options_.num_threads))));
      std::vector<std::thread> pool;
      for (uint32_t w = 1; w < workers; ++w) pool.emplace_back(rerank_range, w, workers);
      rerank_range(0, workers);
      for (std::thread& t : pool) t.join();
    }
    if (!host_error.empty()) {
      last_error_ = host_error;
      dsm_ctx_destroy(ctx);
      return false;
    }
  }
  if (rc != DSM_OK) last_error_ = dsm_last_error(ctx);
  dsm_ctx_destroy(ctx);
  if (rc != DSM_OK) return false;
  for (uint32_t q = 0; q < n; ++q)  // similarity_graph.cpp:183-194
    for (uint32_t k = 0; k < counts[q]; ++k) {
      const image_t other = ids[idx[static_cast<size_t>(q) * max_images + k]];
      if (ids[q] < other) {
        image_pairs_.emplace_back(ids[q], other);
        scores_.push_back(sc[static_cast<size_t>(q) * max_images + k] * 1e3f);
      }
    }
  return true;
}

}  // namespace dagsfm_amd

// ---------------------------------------------------------------------------------------- flat C API
// Thin C exports over the classes above for the Python tests and the CLI.
using namespace dagsfm_amd;

extern "C" {

// Runs ExhaustiveFeatureMatcher over database_path.  Returns 0 on success.
// gpu_index: SiftMatchingOptions::gpu_index ("-1" or null: all devices).
// async_write_back: SiftMatchingOptions::async_write_back -- 0 off, > 0 on, < 0 the option's default (on).
// flags: DSM_HOST_FLAG_* below, each bit one thing (the old entry points packed them into async_write_back).
// Nothing here reads the process environment: the CLI parses its own flags (exhaustive_matcher_main.cc).
// match_slice_pairs: SiftMatchingOptions::match_slice_pairs (< 0: its default).
enum {
  DSM_HOST_FLAG_PRINT_TIMING = 1u,       // one line of stage timers on stderr
  DSM_HOST_FLAG_BULK_LOAD_JOURNAL = 2u,  // SiftMatchingOptions::bulk_load_journal
  DSM_HOST_FLAG_SERIAL_SETUP = 4u,       // ExhaustiveMatchingOptions::overlap_setup = false
  DSM_HOST_FLAG_ASSEMBLE_ON_DEVICE = 8u  // SiftMatchingOptions::assemble_on_device (RCCL, libdagsfm_gather.so)
};
int dsm_host_exhaustive_matcher_ex4(const char* database_path, int block_size, int use_prior_defaults, uint32_t random_seed,
                                    double max_ratio, double max_distance, int cross_check, int min_num_inliers,
                                    int guided_matching, int multiple_models, const char* gpu_index, int async_write_back,
                                    unsigned flags, int match_slice_pairs) {
  const bool print_timing = flags & DSM_HOST_FLAG_PRINT_TIMING;
  const bool bulk_load_journal = flags & DSM_HOST_FLAG_BULK_LOAD_JOURNAL;
  const bool serial_setup = flags & DSM_HOST_FLAG_SERIAL_SETUP;
  try {
    ExhaustiveMatchingOptions eo;
    eo.block_size = block_size;
    eo.overlap_setup = !serial_setup;
    SiftMatchingOptions mo;
    if (!use_prior_defaults) {
      mo.max_ratio = max_ratio;
      mo.max_distance = max_distance;
      mo.cross_check = cross_check != 0;
      mo.min_num_inliers = min_num_inliers;
    }
    mo.guided_matching = guided_matching != 0;
    mo.multiple_models = multiple_models != 0;
    if (async_write_back >= 0) mo.async_write_back = async_write_back != 0;  // overlap SQLite with the device (default: on)
    mo.bulk_load_journal = bulk_load_journal;
    mo.assemble_on_device = (flags & DSM_HOST_FLAG_ASSEMBLE_ON_DEVICE) != 0;
    if (match_slice_pairs >= 0) mo.match_slice_pairs = match_slice_pairs;
    if (gpu_index && *gpu_index) mo.gpu_index = gpu_index;
    mo.random_seed = random_seed;
    ExhaustiveFeatureMatcher m(eo, mo, database_path);
    const bool ok = m.Run();
    if (ok && print_timing) {  // the CLI's --timing 1: one line on stderr, what tools/bench_cli.py keeps
      const SiftFeatureMatcher::Timings t = m.MatcherTimings();
      std::fprintf(stderr, "[dsm_exhaustive_matcher] set-up (device contexts || cache + first block's features) %.3f s;  "
                           "pairs %llu  run %.3f s  =  features from database.db -> device %.3f s  +  device (match + verify + fetch) "
                           "%.3f s (match %.3f, verify %.3f, fetch %.3f)  +  SQLite write-back %.3f s%s  +  other %.3f s\n",
                   m.setup_seconds, static_cast<unsigned long long>(t.pairs), m.run_seconds, t.resident_s, t.device_s, t.match_s, t.verify_s, t.fetch_s, t.write_s,
                   mo.async_write_back ? " (on the write-back thread: overlaps the device time)" : "",
                   m.run_seconds - t.resident_s - t.device_s - (mo.async_write_back ? 0.0 : t.write_s));
    }
    return ok ? 0 : 2;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return 1;
  }
}

// The older entry points: async_write_back is a boolean here (any non-zero value = on, nothing else is read from it).
int dsm_host_exhaustive_matcher_ex3(const char* database_path, int block_size, int use_prior_defaults, uint32_t random_seed,
                                    double max_ratio, double max_distance, int cross_check, int min_num_inliers,
                                    int guided_matching, int multiple_models, const char* gpu_index, int async_write_back,
                                    int match_slice_pairs) {
  return dsm_host_exhaustive_matcher_ex4(database_path, block_size, use_prior_defaults, random_seed, max_ratio, max_distance, cross_check,
                                         min_num_inliers, guided_matching, multiple_models, gpu_index, async_write_back != 0, 0u,
                                         match_slice_pairs);
}

int dsm_host_exhaustive_matcher_ex2(const char* database_path, int block_size, int use_prior_defaults, uint32_t random_seed,
                                    double max_ratio, double max_distance, int cross_check, int min_num_inliers,
                                    int guided_matching, int multiple_models, const char* gpu_index, int async_write_back) {
  return dsm_host_exhaustive_matcher_ex3(database_path, block_size, use_prior_defaults, random_seed, max_ratio, max_distance, cross_check,
                                         min_num_inliers, guided_matching, multiple_models, gpu_index, async_write_back, -1);
}

int dsm_host_exhaustive_matcher_ex(const char* database_path, int block_size, int use_prior_defaults, uint32_t random_seed,
                                   double max_ratio, double max_distance, int cross_check, int min_num_inliers,
                                   int guided_matching, int multiple_models) {
  return dsm_host_exhaustive_matcher_ex2(database_path, block_size, use_prior_defaults, random_seed, max_ratio, max_distance, cross_check,
                                         min_num_inliers, guided_matching, multiple_models, nullptr, 0);
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

// VocabSimilarityGraph::Run over database_path with the vocabulary file; writes up to `capacity` pairs (image ids) and
// their scores.  Returns the number of pairs, or < 0 on error.
int64_t dsm_host_vocab_candidate_pairs2(const char* database_path, const char* vocab_path, int num_images, int num_nearest_neighbors,
                                        int max_num_features, uint32_t* pairs, float* scores, uint64_t capacity);
int64_t dsm_host_vocab_candidate_pairs(const char* database_path, const char* vocab_path, int num_images, int num_nearest_neighbors,
                                       uint32_t* pairs, float* scores, uint64_t capacity) {
  return dsm_host_vocab_candidate_pairs2(database_path, vocab_path, num_images, num_nearest_neighbors, -1, pairs, scores, capacity);
}
// the vocabulary of a file in either layout, for the tests: returns num_words (0: unreadable); arrays may be null
uint32_t dsm_host_read_vocabulary(const char* path, uint8_t* words, float* projection, float* thresholds, uint32_t capacity_words) {
  VocabularyFile v;
  if (!v.Read(path)) return 0;
  if (v.num_words <= capacity_words) {
    if (words) std::memcpy(words, v.words.data(), v.words.size());
    if (projection) std::memcpy(projection, v.projection.data(), v.projection.size() * 4);
    if (thresholds) std::memcpy(thresholds, v.thresholds.data(), v.thresholds.size() * 4);
  }
  return v.num_words;
}
// FlannIndex (flann_index.h) over the FLANN section of a vocabulary file in the reference's layout, for
// tests/test_retrieval_flann.py: the k nearest words of n descriptors as the reference's own search returns them.
// Returns the algorithm the file's index uses (0 linear, 1 kd-trees, 2 k-means), < 0 on error.
int dsm_host_flann_find_word_ids(const char* vocab_path, const uint8_t* descriptors, uint32_t n, uint32_t k, int num_checks, int num_threads,
                                 int32_t* out_ids, float* out_dists, uint64_t* end_offset) {
  VocabularyFile v;
  if (!v.ReadReferenceLayout(vocab_path) || !v.flann_framed) return -1;
  FlannIndex index;
  size_t at = 0;
  if (!index.Load(v.flann_blob.data(), v.flann_blob.size(), &at, v.words.data(), v.num_words)) {
    std::cerr << "ERROR: " << index.error() << std::endl;
    return -2;
  }
  if (end_offset) *end_offset = v.index_begin + at;
  if (n && !index.FindWordIds(descriptors, n, k, num_checks, num_threads, out_ids, out_dists)) return -3;
  return index.algorithm();
}
// The same search ON THE DEVICE (csrc/flann_search.hip through dsm_retrieval_set_flann_index / dsm_retrieval_flann_search): the
// file's index parsed here, its trees handed to a context on `device`.  Returns the algorithm (>= 0) or a negative error;
// *kernel_ms (may be null) = the search kernel's device time.
int dsm_host_flann_device_search(const char* vocab_path, int device, const uint8_t* descriptors, uint32_t n, uint32_t k, int num_checks,
                                 int32_t* out_ids, float* out_dists, double* kernel_ms) {
  VocabularyFile v;
  if (!v.ReadReferenceLayout(vocab_path) || !v.flann_framed) return -1;
  FlannIndex index;
  size_t at = 0;
  if (!index.Load(v.flann_blob.data(), v.flann_blob.size(), &at, v.words.data(), v.num_words)) {
    std::cerr << "ERROR: " << index.error() << std::endl;
    return -2;
  }
  dsm_flann_index flat;
  if (!index.Export(num_checks, &flat)) return -3;
  dsm_ctx* ctx = nullptr;
  if (dsm_ctx_create(device, &ctx) != DSM_OK) return -4;
  dsm_vocabulary voc;
  voc.num_words = v.num_words;
  voc.reserved = 0;
  voc.words = v.words.data();
  voc.projection = v.projection.data();
  voc.thresholds = v.thresholds.data();
  int rc = dsm_retrieval_set_vocabulary(ctx, &voc);
  if (rc == DSM_OK) rc = dsm_retrieval_set_flann_index(ctx, &flat);
  if (rc == DSM_OK) rc = dsm_retrieval_flann_search(ctx, descriptors, n, k, out_ids, out_dists, kernel_ms);
  if (rc != DSM_OK) std::cerr << "ERROR: " << dsm_last_error(ctx) << std::endl;
  dsm_ctx_destroy(ctx);
  return rc == DSM_OK ? index.algorithm() : -5;
}
// Hands the FLANN index of a vocabulary file (reference layout) to an EXISTING context whose vocabulary is that file's:
// dsm_retrieval_set_flann_index over the parsed trees.  For tools that drive the C-ABI themselves (tools/bench_retrieval.py);
// `ctx` is the dsm_ctx* of the same libdagsfm_mi355x.so instance.  Returns the algorithm (>= 0) or a negative error.
int dsm_host_flann_attach(const char* vocab_path, void* ctx, int num_checks) {
  VocabularyFile v;
  if (!ctx || !v.ReadReferenceLayout(vocab_path) || !v.flann_framed) return -1;
  FlannIndex index;
  size_t at = 0;
  if (!index.Load(v.flann_blob.data(), v.flann_blob.size(), &at, v.words.data(), v.num_words)) return -2;
  dsm_flann_index flat;
  if (!index.Export(num_checks, &flat)) return -3;
  if (dsm_retrieval_set_flann_index(static_cast<dsm_ctx*>(ctx), &flat) != DSM_OK) return -5;
  return index.algorithm();
}
// where ReadReferenceLayout found the FLANN index of a vocabulary file in the reference's layout: begin / end offsets,
// *framed = 1 when it walked FLANN's archive framing (0: located the inverted index by its header).  Returns num_words.
uint32_t dsm_host_vocabulary_index_range(const char* path, uint64_t* begin, uint64_t* end, int* framed) {
  VocabularyFile v;
  if (!v.ReadReferenceLayout(path)) return 0;
  if (begin) *begin = v.index_begin;
  if (end) *end = v.index_end;
  if (framed) *framed = v.flann_framed ? 1 : 0;
  return v.num_words;
}
int64_t dsm_host_vocab_candidate_pairs3(const char* database_path, const char* vocab_path, int num_images, int num_nearest_neighbors,
                                        int max_num_features, int num_images_after_verification, uint32_t* pairs, float* scores,
                                        uint64_t capacity);
int64_t dsm_host_vocab_candidate_pairs2(const char* database_path, const char* vocab_path, int num_images, int num_nearest_neighbors,
                                        int max_num_features, uint32_t* pairs, float* scores, uint64_t capacity) {
  return dsm_host_vocab_candidate_pairs3(database_path, vocab_path, num_images, num_nearest_neighbors, max_num_features, 0, pairs, scores,
                                         capacity);
}
// leaf hooks of the spatial re-ranking for tests/test_retrieval.py (geometries: [n][4] = x, y, scale, orientation)
int dsm_host_sv_vote_and_verify(uint32_t n, const float* g1, const float* g2) {
  std::vector<GeometryMatch> m(n);
  for (uint32_t i = 0; i < n; ++i) {
    m[i].query.x = g1[4 * i]; m[i].query.y = g1[4 * i + 1]; m[i].query.scale = g1[4 * i + 2]; m[i].query.orientation = g1[4 * i + 3];
    m[i].database.x = g2[4 * i]; m[i].database.y = g2[4 * i + 1]; m[i].database.scale = g2[4 * i + 2]; m[i].database.orientation = g2[4 * i + 3];
  }
  return VoteAndVerify(VoteAndVerifyOptions(), m);
}
// HammingDistWeightFunctor<64, 16>()(h), retrieval/utils.h:47-78
float dsm_host_sv_hamming_weight(uint32_t h) {
  const float hamming_dist = static_cast<float>(h);
  return hamming_dist <= 24 ? std::exp(-hamming_dist * hamming_dist / (16.0f * 16.0f)) : 0.0f;
}
void dsm_host_sv_estimate_affine(const double* x1, const double* x2, uint32_t n, double* A6) { EstimateAffineTransform(x1, x2, n, A6); }
void dsm_host_sv_keypoint_geometry(const float* kp6, uint32_t n, float* out4) {  // kp6: x, y, a11, a12, a21, a22
  for (uint32_t i = 0; i < n; ++i) {
    FeatureKeypoint k;
    k.x = kp6[6 * i]; k.y = kp6[6 * i + 1]; k.a11 = kp6[6 * i + 2]; k.a12 = kp6[6 * i + 3]; k.a21 = kp6[6 * i + 4]; k.a22 = kp6[6 * i + 5];
    const FeatureGeometry g = GeometryOfKeypoint(k);
    out4[4 * i] = g.x; out4[4 * i + 1] = g.y; out4[4 * i + 2] = g.scale; out4[4 * i + 3] = g.orientation;
  }
}
// SpatialRerank of one query from flat arrays: candidates as 5 uint32 (dsm_get_retrieval_matches) + weight + geometry
uint32_t dsm_host_spatial_rerank(uint32_t n_query_features, const float* query_geom, uint64_t n_candidates, const uint32_t* tuples,
                                 const float* weights, const float* database_geom, int num_images_after_verification, uint32_t count,
                                 uint32_t* image_idx, float* scores) {
  std::vector<FeatureGeometry> qg(n_query_features);
  for (uint32_t i = 0; i < n_query_features; ++i) {
    qg[i].x = query_geom[4 * i]; qg[i].y = query_geom[4 * i + 1]; qg[i].scale = query_geom[4 * i + 2]; qg[i].orientation = query_geom[4 * i + 3];
  }
  std::vector<RetrievalCandidate> c(n_candidates);
  for (uint64_t m = 0; m < n_candidates; ++m) {
    c[m].query_feature = tuples[5 * m];
    c[m].image = tuples[5 * m + 1];
    c[m].database_feature = tuples[5 * m + 2];
    c[m].entry_position = tuples[5 * m + 4];
    c[m].weight = weights[m];
    c[m].database_geometry.x = database_geom[4 * m]; c[m].database_geometry.y = database_geom[4 * m + 1];
    c[m].database_geometry.scale = database_geom[4 * m + 2]; c[m].database_geometry.orientation = database_geom[4 * m + 3];
  }
  return SpatialRerank(qg, c, num_images_after_verification, count, image_idx, scores);
}
int64_t dsm_host_vocab_candidate_pairs4(const char* database_path, const char* vocab_path, int num_images, int num_nearest_neighbors,
                                        int max_num_features, int num_images_after_verification, int word_search_flann, int num_checks,
                                        uint32_t* pairs, float* scores, uint64_t capacity);
int64_t dsm_host_vocab_candidate_pairs3(const char* database_path, const char* vocab_path, int num_images, int num_nearest_neighbors,
                                        int max_num_features, int num_images_after_verification, uint32_t* pairs, float* scores,
                                        uint64_t capacity) {
  return dsm_host_vocab_candidate_pairs4(database_path, vocab_path, num_images, num_nearest_neighbors, max_num_features,
                                         num_images_after_verification, 0, 256, pairs, scores, capacity);  // (this older entry point: the exact search)
}
// word_search_flann: VocabSimilaritySearchOptions::WordSearch (0 exact, 1 FLANN on the device, 2 FLANN on host threads, 3 auto) with `num_checks`
int64_t dsm_host_vocab_candidate_pairs4(const char* database_path, const char* vocab_path, int num_images, int num_nearest_neighbors,
                                        int max_num_features, int num_images_after_verification, int word_search_flann, int num_checks,
                                        uint32_t* pairs, float* scores, uint64_t capacity) {
  try {
    Database db(database_path);
    VocabSimilaritySearchOptions o;
    o.num_images = num_images;
    o.num_nearest_neighbors = num_nearest_neighbors;
    o.max_num_features = max_num_features;
    o.num_images_after_verification = num_images_after_verification;
    o.vocab_tree_path = vocab_path;
    o.word_search = static_cast<VocabSimilaritySearchOptions::WordSearch>(word_search_flann < 0 || word_search_flann > 3 ? 3 : word_search_flann);
    o.num_checks = num_checks;
    VocabSimilarityGraph g(o, db);
    if (!g.Run()) {
      std::cerr << "ERROR: " << g.LastError() << std::endl;
      return -2;
    }
    const uint64_t n = std::min<uint64_t>(capacity, g.ImagePairs().size());
    for (uint64_t k = 0; k < n; ++k) {
      pairs[2 * k] = g.ImagePairs()[k].first;
      pairs[2 * k + 1] = g.ImagePairs()[k].second;
      scores[k] = g.Scores()[k];
    }
    return static_cast<int64_t>(g.ImagePairs().size());
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return -1;
  }
}

int dsm_host_write_vocabulary(const char* path, uint32_t num_words, const uint8_t* words, const float* projection, const float* thresholds) {
  VocabularyFile v;
  v.num_words = num_words;
  v.words.assign(words, words + static_cast<size_t>(num_words) * 128);
  v.projection.assign(projection, projection + 64 * 128);
  v.thresholds.assign(thresholds, thresholds + static_cast<size_t>(num_words) * 64);
  return v.Write(path) ? 0 : 1;
}

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

// Probe of SiftFeatureMatcher::Match's contract under the DEFAULT options (async_write_back on): the caller -- like the
// reference's ExhaustiveFeatureMatcher::Run, matching.cc:903 -- holds one DatabaseTransaction around Match() over all
// image pairs, in slices of `match_slice_pairs`.  When Match() has returned every row must already be written INTO THAT
// transaction (no writer in flight, nothing committed behind the caller's back): rows_inside = what the connection sees
// before the transaction ends; then the transaction is committed (commit != 0) or rolled back.  Returns 0 on success.
int dsm_host_probe_match_in_callers_transaction(const char* database_path, int match_slice_pairs, int commit, uint32_t random_seed,
                                                uint64_t* rows_inside_matches, uint64_t* rows_inside_geometries, int* still_in_transaction) {
  try {
    Database db(database_path);
    FeatureMatcherCache cache(1000, &db);
    SiftMatchingOptions mo;  // the defaults: async_write_back on, defer_write_back off
    if (match_slice_pairs >= 0) mo.match_slice_pairs = match_slice_pairs;
    mo.random_seed = random_seed;
    if (!mo.async_write_back || mo.defer_write_back) return 3;
    SiftFeatureMatcher matcher(mo, &db, &cache);
    if (!matcher.Setup()) return 2;
    cache.Setup();
    const std::vector<image_t> ids = cache.GetImageIds();
    std::vector<std::pair<image_t, image_t>> pairs;
    for (size_t i = 0; i < ids.size(); ++i)
      for (size_t j = i + 1; j < ids.size(); ++j) pairs.emplace_back(ids[i], ids[j]);
    DatabaseTransaction transaction(&db);
    matcher.Match(pairs);
    if (still_in_transaction) *still_in_transaction = db.InTransaction() ? 1 : 0;
    if (rows_inside_matches) *rows_inside_matches = db.NumMatchedImagePairs();
    if (rows_inside_geometries) *rows_inside_geometries = db.ReadPairIds(true).size();
    if (commit)
      transaction.Commit();
    else
      transaction.Rollback();
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return 1;
  }
}

}  // extern "C"
