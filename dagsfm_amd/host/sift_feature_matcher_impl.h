// sift_feature_matcher_impl.h -- SiftFeatureMatcher (/root/reference/src/feature/matching.h:334-368, matching.cc:610-839)
// over the C-ABI, as a template over the HOST PROJECT'S OWN TYPES.
//
// The reference's controllers construct `SiftFeatureMatcher(options, &database, &cache)` with colmap::Database* and
// colmap::FeatureMatcherCache* (src/controllers/distributed_mapper_controller.cpp:506-520), so a drop-in must take
// exactly those types.  Everything the matcher does with them goes through `Traits`:
//
//   types      Options, Database, Cache, Camera, Image, FeatureKeypoints, FeatureDescriptors, FeatureMatches,
//              TwoViewGeometry
//   constants  kCachePinsRequested   references returned by Cache::GetKeypoints / GetDescriptors stay valid until
//                                    Traits::ReleasePins (else the matcher copies before the next Get: an LRU may evict)
//              kAsyncWriteBack       the Options / Cache carry this repository's asynchronous write-back extension
//   functions  PairId(a, b); CacheSize(cache); ReleasePins(cache); LockBatch(cache) (RAII guard, may be empty);
//              ToDsmCamera(camera, &out); CameraIdOf(image); KeypointData(kps, &n, &stride_in_floats);
//              DescriptorData(desc, &rows, &cols); AppendFlat(matches, &flat); MakeMatches(flat, n);
//              MakeTwoViewGeometry(record or nullptr, inliers, n); RandomSeed(options); AsyncWriteBack(options);
//              MatchSlicePairs(options) (only with kAsyncWriteBack: this repository's Options);
//              AssembleOnDevice(options): the shares of the devices are assembled by RCCL (libdagsfm_gather.so) instead of
//              fetched one by one (false for the reference's Options)
//
// feature_matching.h instantiates it with this repository's own types (NativeTraits); colmap_traits.h with the
// reference's (compile-checked against tests/colmap_stub, which carries the reference's exact signatures).
#ifndef DAGSFM_AMD_HOST_SIFT_FEATURE_MATCHER_IMPL_H_
#define DAGSFM_AMD_HOST_SIFT_FEATURE_MATCHER_IMPL_H_

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include <dlfcn.h>

#include "../../include/dagsfm_mi355x.h"

namespace dagsfm_amd {

// A result buffer that is NOT zero-filled on allocation (std::vector<uint32_t>(n) writes every page once before the device
// copy writes it again: the matches of config 2 are 460 MB per call) and moves without copying.
struct RawU32Buffer {
  std::unique_ptr<uint32_t[]> p;
  size_t n;
  RawU32Buffer() : n(0) {}
  void Allocate(size_t count) {
    p.reset(new uint32_t[count]);
    n = count;
  }
  uint32_t* data() { return p.get(); }
  const uint32_t* data() const { return p.get(); }
  void swap(RawU32Buffer& o) {
    p.swap(o.p);
    std::swap(n, o.n);
  }
};

// libdagsfm_gather.so (include/dagsfm_gather.h), loaded on first use from the directory of libdagsfm_mi355x.so: only a host that
// asks for the on-device assembly maps librccl (0.5 s of load time).  Plain function pointers: this header stays free of RCCL.
struct DeviceGatherApi {
  void* handle;
  int (*create)(dsm_ctx* const*, uint32_t, void**);
  void (*destroy)(void*);
  const char* (*last_error)(const void*);
  int (*match_graph)(void*, const uint32_t*, int32_t);
  int (*sizes)(const void*, uint64_t*, uint64_t*, uint64_t*);
  int (*fetch)(void*, uint32_t, uint64_t*, uint32_t*, dsm_two_view_geometry*, uint64_t*, uint32_t*);
  DeviceGatherApi() : handle(nullptr), create(nullptr), destroy(nullptr), last_error(nullptr), match_graph(nullptr), sizes(nullptr), fetch(nullptr) {}
  bool Load(std::string* error) {
    if (handle) return true;
    std::string path = "libdagsfm_gather.so";
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&dsm_ctx_create), &info) && info.dli_fname) {
      const std::string lib(info.dli_fname);
      const size_t slash = lib.rfind('/');
      if (slash != std::string::npos) path = lib.substr(0, slash + 1) + path;
    }
    handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!handle) {
      const char* e = dlerror();
      *error = std::string("cannot load ") + path + ": " + (e ? e : "?");
      return false;
    }
    create = reinterpret_cast<int (*)(dsm_ctx* const*, uint32_t, void**)>(dlsym(handle, "dsm_gather_create"));
    destroy = reinterpret_cast<void (*)(void*)>(dlsym(handle, "dsm_gather_destroy"));
    last_error = reinterpret_cast<const char* (*)(const void*)>(dlsym(handle, "dsm_gather_last_error"));
    match_graph = reinterpret_cast<int (*)(void*, const uint32_t*, int32_t)>(dlsym(handle, "dsm_gather_match_graph"));
    sizes = reinterpret_cast<int (*)(const void*, uint64_t*, uint64_t*, uint64_t*)>(dlsym(handle, "dsm_gather_sizes"));
    fetch = reinterpret_cast<int (*)(void*, uint32_t, uint64_t*, uint32_t*, dsm_two_view_geometry*, uint64_t*, uint32_t*)>(dlsym(handle, "dsm_gather_fetch"));
    if (!create || !destroy || !last_error || !match_graph || !sizes || !fetch) {
      *error = path + " lacks a dsm_gather_* entry point";
      dlclose(handle);
      handle = nullptr;
      return false;
    }
    return true;
  }
};

template <typename Traits>
class SiftFeatureMatcherT {
 public:
  typedef uint32_t image_id_t;
  typedef std::vector<std::pair<image_id_t, image_id_t>> PairList;

  SiftFeatureMatcherT(const typename Traits::Options& options, typename Traits::Database* database, typename Traits::Cache* cache)
      : options_(options), database_(database), cache_(cache) {
    if (!options_.Check()) throw std::invalid_argument("SiftMatchingOptions::Check failed");  // CHECK(options_.Check()), matching.cc:614
  }
  ~SiftFeatureMatcherT() {
    if (writer_.joinable()) writer_.join();
    if (writer_error_) {  // never Flush()ed: a destructor cannot throw, so at least say it (the rows were rolled back)
      try {
        std::rethrow_exception(writer_error_);
      } catch (const std::exception& e) {
        std::fprintf(stderr, "dagsfm_amd::SiftFeatureMatcher: asynchronous write-back failed and was rolled back: %s\n", e.what());
      } catch (...) {
        std::fprintf(stderr, "dagsfm_amd::SiftFeatureMatcher: asynchronous write-back failed and was rolled back\n");
      }
    }
    if (gather_) gather_api_.destroy(gather_);  // (before the contexts it was created over)
    for (dsm_ctx* c : ctxs_) dsm_ctx_destroy(c);
  }
  SiftFeatureMatcherT(const SiftFeatureMatcherT&) = delete;
  SiftFeatureMatcherT& operator=(const SiftFeatureMatcherT&) = delete;

  // Creates one device context per entry of gpu_index ("-1": every visible device; matching.cc:631-645, doc/faq.rst:
  // 322-331); false when a device cannot be set up (matching.cc:732-742).
  bool Setup() {
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
      return false;
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
    max_resident_ = cache_ ? Traits::CacheSize(cache_) : 0;
    is_setup_ = true;
    return true;
  }

  // Waits for an asynchronous write-back and rethrows its error, if any.
  void Flush() {
    if (writer_.joinable()) writer_.join();
    if (writer_error_) {
      std::exception_ptr e = writer_error_;
      writer_error_ = nullptr;
      std::rethrow_exception(e);
    }
  }

  // Where a run's wall time went (seconds, summed over all Match() calls of this object): reading features from the cache /
  // database and making them resident on the devices; the device calls (matching, verification, fetch of the results); the
  // write-back into the `matches` / `two_view_geometries` tables -- on the caller's thread, or on the write-back thread
  // (then it overlaps the next call's device work and is NOT part of the caller's wall time).
  struct Timings {
    double resident_s, device_s, write_s;
    uint64_t pairs;
    double fetch_s;  // the part of device_s that copies the results to the host (the slowest device's, per call)
    double match_s, verify_s;  // likewise: dsm_match_pairs / dsm_set_matches, dsm_verify_pairs (+ guided matching)
  };
  Timings GetTimings() const { return timings_; }
  const std::string& LastError() const { return last_error_; }
  size_t NumDevices() const { return ctxs_.size(); }
  size_t NumResidentImages() const { return image_ids_.size(); }

  // Matches + verifies the pairs and writes `matches` / `two_view_geometries` rows, with the reference's dedupe /
  // skip / partial-recompute / post-filter semantics (matching.cc:749-839).
  void Match(const PairList& image_pairs) {
    if (!database_ || !cache_ || !is_setup_) throw std::logic_error("SiftFeatureMatcher::Match before Setup");  // CHECKs :751-753
    if (image_pairs.empty()) return;

    // ---- dedupe, resume semantics (matching.cc:763-813).  The reference deletes the stale rows right here and
    // CHECK-aborts on any later failure; this class reports device failures as exceptions, so a row must not vanish
    // before its replacement exists: the deletes are only RECORDED here and applied next to the new rows (write()
    // below), i.e. after the device results are on the host and inside the same transaction.
    std::unordered_set<uint64_t> seen;
    PairList to_match, to_verify_only;
    std::vector<char> stale_inliers_match, stale_inliers_verify;  // the pair has a two_view_geometries row to replace
    std::vector<typename Traits::FeatureMatches> existing;
    {
      const auto lock = Traits::LockBatch(cache_);  // one acquisition for the whole list where the cache offers it
      (void)lock;
      for (const auto& pr : image_pairs) {
        if (pr.first == pr.second) continue;
        if (!seen.insert(Traits::PairId(pr.first, pr.second)).second) continue;
        const bool exists_matches = cache_->ExistsMatches(pr.first, pr.second);
        const bool exists_inlier_matches = cache_->ExistsInlierMatches(pr.first, pr.second);
        if (exists_matches && exists_inlier_matches) continue;
        if (exists_matches) {
          existing.push_back(cache_->GetMatches(pr.first, pr.second));
          to_verify_only.push_back(pr);
          stale_inliers_verify.push_back(exists_inlier_matches ? 1 : 0);
        } else {
          to_match.push_back(pr);
          stale_inliers_match.push_back(exists_inlier_matches ? 1 : 0);
        }
      }
    }
    {
      const Clock::time_point t0 = Clock::now();
      const bool resident = EnsureResident(to_match, to_verify_only);
      timings_.resident_s += Seconds(t0);
      if (!resident) throw std::runtime_error(last_error_);
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
    to.multiple_models = options_.multiple_models ? 1 : 0;  // multiple_ignore_watermark stays at its default (true)
    Run(to_match, nullptr, stale_inliers_match, mo, to);
    Run(to_verify_only, &existing, stale_inliers_verify, mo, to);
    // every row of this call is written when Match() returns (unless the caller asked for the deferred form and Flush()es)
    FlushUnlessDeferred(std::integral_constant<bool, Traits::kAsyncWriteBack>());
  }

 private:
  // One share of the pair list on one device: match (or install the given matches), verify, fetch.
  struct Share {
    uint32_t begin = 0, end = 0;
    std::vector<uint64_t> moff, ioff;
    RawU32Buffer m, im;
    std::vector<dsm_two_view_geometry> tv;
    std::string error;
    double fetch_s = 0.0, match_s = 0.0, verify_s = 0.0;
  };

  void RunShare(dsm_ctx* ctx, const PairList& prs, const std::vector<typename Traits::FeatureMatches>* given,
                const dsm_match_options& mo, const dsm_two_view_options& to, Share* sh, bool fetch) const {
    const uint32_t np = sh->end - sh->begin;
    if (np == 0) return;
    std::vector<uint32_t> idx(2 * static_cast<size_t>(np)), seeds(np);
    for (uint32_t i = 0; i < np; ++i) {
      const auto& pr = prs[sh->begin + i];
      idx[2 * i] = image_index_.at(pr.first);
      idx[2 * i + 1] = image_index_.at(pr.second);
      seeds[i] = dsm_pair_seed(pr.first, pr.second, Traits::RandomSeed(options_));
    }
    int rc;
    const Clock::time_point t_match = Clock::now();
    if (given) {
      std::vector<uint64_t> off(np + 1, 0);
      std::vector<uint32_t> flat;
      for (uint32_t i = 0; i < np; ++i) {
        Traits::AppendFlat((*given)[sh->begin + i], &flat);
        off[i + 1] = flat.size() / 2;
      }
      if (flat.empty()) flat.resize(2);
      rc = dsm_set_matches(ctx, np, idx.data(), off.data(), flat.data());
    } else {
      rc = dsm_match_pairs(ctx, np, idx.data(), &mo);
    }
    sh->match_s = Seconds(t_match);
    const Clock::time_point t_verify = Clock::now();
    // guided_matching (matching.cc:647-667): verifier -> guided matcher -> output; the post-filter then sees the guided counts
    if (rc == DSM_OK) rc = dsm_verify_pairs(ctx, &to, seeds.data(), 0, options_.guided_matching ? 0 : 1);
    if (rc == DSM_OK && options_.guided_matching) rc = dsm_guided_match_pairs(ctx, &mo, &to, 1);
    if (rc != DSM_OK) {
      sh->error = std::string("device matching failed: ") + dsm_last_error(ctx);
      return;
    }
    sh->verify_s = Seconds(t_verify);
    if (!fetch) return;  // the shares are assembled on the devices (AssembleShares)
    const Clock::time_point t_fetch = Clock::now();
    sh->moff.assign(np + 1, 0);
    sh->ioff.assign(np + 1, 0);
    rc = dsm_get_matches(ctx, sh->moff.data(), nullptr, 0);
    sh->m.Allocate(2 * std::max<uint64_t>(sh->moff[np], 1));
    if (rc == DSM_OK) rc = dsm_get_matches(ctx, nullptr, sh->m.data(), sh->moff[np]);
    sh->tv.resize(np);
    if (rc == DSM_OK) rc = dsm_get_two_view_geometries(ctx, sh->tv.data());
    if (rc == DSM_OK) rc = dsm_get_inlier_matches(ctx, sh->ioff.data(), nullptr, 0);
    sh->im.Allocate(2 * std::max<uint64_t>(sh->ioff[np], 1));
    if (rc == DSM_OK) rc = dsm_get_inlier_matches(ctx, nullptr, sh->im.data(), sh->ioff[np]);
    if (rc != DSM_OK) sh->error = std::string("result fetch failed: ") + dsm_last_error(ctx);
    sh->fetch_s = Seconds(t_fetch);
  }

  // A long list goes to the devices slice by slice (SiftMatchingOptions::match_slice_pairs).  The device scratch of a call is
  // sized by its pair list -- 38 GiB for 124 750 pairs -- and the FIRST call of a process has to allocate it: 0.1 - 1.2 s,
  // erratic, measured through the CLI's stage timers; a slice of 32 768 pairs needs a quarter of it and every later slice
  // re-uses it.  With the asynchronous write-back the rows of slice k are written (one transaction of the writer thread per
  // slice) while slice k + 1 is on the devices, so one Match() over a block of 500 images overlaps its own write-back instead
  // of only the next block's; without it they are written by the caller between the slices.  Per-pair seeds and the matching
  // depend on the pair alone, so the rows are those of the unsliced call.
  size_t MatchSlicePairs(std::true_type) const { return Traits::MatchSlicePairs(options_); }
  size_t MatchSlicePairs(std::false_type) const { return 32768; }  // (the reference's Options: the default)

  void Run(const PairList& prs, const std::vector<typename Traits::FeatureMatches>* given, const std::vector<char>& stale_inliers,
           const dsm_match_options& mo, const dsm_two_view_options& to) {
    const size_t slice = MatchSlicePairs(std::integral_constant<bool, Traits::kAsyncWriteBack>());
    if (slice == 0 || prs.size() <= slice + slice / 2) {  // (no slice shorter than half the nominal length; 0: never sliced)
      RunSlice(prs, given, stale_inliers, mo, to);
      return;
    }
    const size_t n_slices = (prs.size() + slice - 1) / slice;
    for (size_t k = 0; k < n_slices; ++k) {
      const size_t b = prs.size() * k / n_slices, e = prs.size() * (k + 1) / n_slices;
      const PairList part(prs.begin() + b, prs.begin() + e);
      const std::vector<char> stale(stale_inliers.begin() + b, stale_inliers.begin() + e);
      if (given) {
        const std::vector<typename Traits::FeatureMatches> given_part(given->begin() + b, given->begin() + e);
        RunSlice(part, &given_part, stale, mo, to);
      } else {
        RunSlice(part, nullptr, stale, mo, to);
      }
    }
  }

  void RunSlice(const PairList& prs, const std::vector<typename Traits::FeatureMatches>* given, const std::vector<char>& stale_inliers,
                const dsm_match_options& mo, const dsm_two_view_options& to) {
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
    const bool on_device = Traits::AssembleOnDevice(options_);
    const Clock::time_point t_device = Clock::now();
    if (nd == 1) {
      RunShare(ctxs_[0], prs, given, mo, to, &shares[0], !on_device);
    } else {
      std::vector<std::thread> th;
      for (size_t d = 0; d < nd; ++d) th.emplace_back([&, d]() { RunShare(ctxs_[d], prs, given, mo, to, &shares[d], !on_device); });
      for (auto& t : th) t.join();
    }
    timings_.device_s += Seconds(t_device);
    timings_.pairs += np;
    double fetch_s = 0.0, match_s = 0.0, verify_s = 0.0;
    for (const Share& sh : shares) {
      fetch_s = std::max(fetch_s, sh.fetch_s);
      match_s = std::max(match_s, sh.match_s);
      verify_s = std::max(verify_s, sh.verify_s);
    }
    timings_.fetch_s += fetch_s;
    timings_.match_s += match_s;
    timings_.verify_s += verify_s;
    for (const Share& sh : shares)
      if (!sh.error.empty()) throw std::runtime_error(sh.error);
    // merge the shares in list order (one device: its buffers ARE the result, nothing is copied)
    std::vector<uint64_t> moff, ioff;
    std::vector<dsm_two_view_geometry> tv;
    RawU32Buffer m, im;
    if (on_device) {
      const Clock::time_point t_gather = Clock::now();
      AssembleShares(shares, nd, &moff, &m, &tv, &ioff, &im);
      const double s_gather = Seconds(t_gather);
      timings_.device_s += s_gather;
      timings_.fetch_s += s_gather;
    } else if (nd == 1) {
      moff.swap(shares[0].moff);
      ioff.swap(shares[0].ioff);
      tv.swap(shares[0].tv);
      m.swap(shares[0].m);
      im.swap(shares[0].im);
    } else {
      moff.assign(np + 1, 0);
      ioff.assign(np + 1, 0);
      tv.resize(np);
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
      m.Allocate(2 * std::max<uint64_t>(mt, 1));
      im.Allocate(2 * std::max<uint64_t>(it, 1));
      for (const Share& sh : shares) {
        if (sh.end == sh.begin) continue;
        std::copy(sh.m.data(), sh.m.data() + 2 * sh.moff[sh.end - sh.begin], m.data() + 2 * moff[sh.begin]);
        std::copy(sh.im.data(), sh.im.data() + 2 * sh.ioff[sh.end - sh.begin], im.data() + 2 * ioff[sh.begin]);
      }
    }
    // ---- write results (matching.cc:819-836), on this thread or handed to the write-back thread
    std::shared_ptr<WriteBatch> batch = std::make_shared<WriteBatch>();
    batch->cache = cache_;
    batch->min_num_inliers = options_.min_num_inliers;
    batch->replace_matches = given != nullptr;  // resume path: the pair's `matches` row exists and is rewritten
    batch->prs = prs;
    batch->stale_inliers = stale_inliers;
    batch->moff.swap(moff);
    batch->m.swap(m);
    batch->tv.swap(tv);
    batch->ioff.swap(ioff);
    batch->im.swap(im);
    if (WriteBackAsync(batch, std::integral_constant<bool, Traits::kAsyncWriteBack>())) return;
    const Clock::time_point t_write = Clock::now();
    batch->Write();
    timings_.write_s += Seconds(t_write);
  }

  // SiftMatchingOptions::assemble_on_device: the devices' shares are assembled ON the devices by RCCL -- one communicator over
  // the gpu_index devices (ncclCommInitAll), grouped all-gather of the fixed-size records, exact-size broadcasts of the
  // match lists (include/dagsfm_gather.h) -- and the host fetches the whole graph from device 0 with one copy per array,
  // instead of one fetch per device merged through host memory.  Also taken with ONE device (a one-rank communicator), which
  // is how a one-GPU box exercises the path.
  void AssembleShares(const std::vector<Share>& shares, size_t nd, std::vector<uint64_t>* moff, RawU32Buffer* m,
                      std::vector<dsm_two_view_geometry>* tv, std::vector<uint64_t>* ioff, RawU32Buffer* im) {
    std::string error;
    if (!gather_api_.Load(&error)) throw std::runtime_error(error);
    if (gather_ && gather_ranks_ != nd) {  // (a slice shorter than the device count uses fewer devices)
      gather_api_.destroy(gather_);
      gather_ = nullptr;
    }
    if (!gather_) {
      if (gather_api_.create(ctxs_.data(), static_cast<uint32_t>(nd), &gather_) != DSM_OK || !gather_)
        throw std::runtime_error("dsm_gather_create failed (RCCL needs the gpu_index devices to be distinct)");
      gather_ranks_ = nd;
    }
    std::vector<uint32_t> np(nd);
    uint64_t total = 0;
    for (size_t d = 0; d < nd; ++d) {
      np[d] = shares[d].end - shares[d].begin;
      total += np[d];
    }
    if (gather_api_.match_graph(gather_, np.data(), 1) != DSM_OK)
      throw std::runtime_error(std::string("dsm_gather_match_graph: ") + gather_api_.last_error(gather_));
    uint64_t n_pairs = 0, n_matches = 0, n_inliers = 0;
    if (gather_api_.sizes(gather_, &n_pairs, &n_matches, &n_inliers) != DSM_OK || n_pairs != total)
      throw std::runtime_error("dsm_gather_sizes: the assembled graph does not cover the pair list");
    moff->assign(n_pairs + 1, 0);
    ioff->assign(n_pairs + 1, 0);
    tv->resize(n_pairs);
    m->Allocate(2 * std::max<uint64_t>(n_matches, 1));
    im->Allocate(2 * std::max<uint64_t>(n_inliers, 1));
    if (gather_api_.fetch(gather_, 0, moff->data(), m->data(), tv->data(), ioff->data(), im->data()) != DSM_OK)
      throw std::runtime_error(std::string("dsm_gather_fetch: ") + gather_api_.last_error(gather_));
  }

  typedef std::chrono::steady_clock Clock;
  static double Seconds(const Clock::time_point& since) { return std::chrono::duration<double>(Clock::now() - since).count(); }

  // The rows of one Run(): what the write-back needs, owned by whoever writes them (this thread or the writer thread).
  // (A struct behind a shared_ptr instead of a lambda with init-captures: the reference builds with -std=c++11,
  // /root/reference/src/CMakeLists.txt:37, and this header compiles with its flags.)
  struct WriteBatch {
    typename Traits::Cache* cache;
    int min_num_inliers;
    bool replace_matches;
    PairList prs;
    std::vector<char> stale_inliers;
    std::vector<uint64_t> moff, ioff;
    RawU32Buffer m, im;
    std::vector<dsm_two_view_geometry> tv;

    void Write() const {
      const uint32_t np = static_cast<uint32_t>(prs.size());
      for (uint32_t i = 0; i < np; ++i) {
        if (stale_inliers[i]) cache->DeleteInlierMatches(prs[i].first, prs[i].second);  // matching.cc:797-799
        if (replace_matches) cache->DeleteMatches(prs[i].first, prs[i].second);         // matching.cc:806-808
        size_t nm = moff[i + 1] - moff[i];
        if (nm < static_cast<size_t>(min_num_inliers)) nm = 0;  // matching.cc:824-826
        const typename Traits::FeatureMatches matches = Traits::MakeMatches(m.data() + 2 * moff[i], nm);
        // TwoViewGeometry() when the pair fails the post-filter (matching.cc:828-831: `< min_num_inliers` only, so with
        // min_num_inliers = 0 an estimated geometry without inliers -- e.g. DEGENERATE -- is kept, as in the reference)
        const bool keep = tv[i].num_inliers >= static_cast<uint32_t>(min_num_inliers);
        const typename Traits::TwoViewGeometry t =
            Traits::MakeTwoViewGeometry(keep ? &tv[i] : nullptr, im.data() + 2 * ioff[i], keep ? ioff[i + 1] - ioff[i] : 0);
        cache->WriteMatches(prs[i].first, prs[i].second, matches);
        cache->WriteTwoViewGeometry(prs[i].first, prs[i].second, t);
      }
    }
  };

  void FlushUnlessDeferred(std::false_type) {}
  void FlushUnlessDeferred(std::true_type) {
    if (!Traits::DeferWriteBack(options_)) Flush();
  }

  // host projects without this repository's asynchronous write-back extension (the reference's own Options / Cache)
  bool WriteBackAsync(const std::shared_ptr<WriteBatch>&, std::false_type) { return false; }

  // SiftMatchingOptions::async_write_back: hands the batch to the writer thread; false when the option is off
  bool WriteBackAsync(const std::shared_ptr<WriteBatch>& batch, std::true_type) {
    if (!Traits::AsyncWriteBack(options_)) return false;
    Flush();  // one write-back in flight; rethrows the previous one's error BEFORE this batch is marked (its pairs stay unmarked then)
    {
      // the rows are on their way: later Match() calls must skip these pairs.  Marked only now, with the device
      // results on the host -- a failed device call above leaves the cache saying what the database says.
      const auto lock = Traits::LockBatch(cache_);
      (void)lock;
      for (const auto& pr : batch->prs) cache_->MarkPending(pr.first, pr.second);
    }
    SiftFeatureMatcherT* const self = this;
    // a caller that holds a transaction on the connection (the reference's Run(): one DatabaseTransaction around Match(),
    // matching.cc:903) keeps owning it: the rows go into it, and Match() joins the writer before it returns.  Read here, on
    // the caller's thread, with no writer in flight.  The deferred form outlives Match(): its writer must own the transaction.
    const bool callers_transaction = Traits::InTransaction(cache_);
    if (callers_transaction && Traits::DeferWriteBack(options_))
      throw std::logic_error("SiftFeatureMatcher: defer_write_back needs the connection outside a transaction (the writer thread owns its own)");
    writer_ = std::thread([self, batch, callers_transaction]() {
      typename Traits::Cache* const cache = batch->cache;
      bool open = false;
      try {
        const Clock::time_point t_write = Clock::now();
        if (!callers_transaction) {
          cache->BeginTransaction();
          open = true;
        }
        batch->Write();
        if (!callers_transaction) {
          cache->EndTransaction();  // a failing COMMIT (SQLITE_BUSY, SQLITE_FULL) leaves the transaction open:
          open = false;             // only a COMMIT that returned has closed it
        }
        self->timings_.write_s += Seconds(t_write);  // (one write-back in flight: Flush() joins before the next starts)
      } catch (...) {
        self->writer_error_ = std::current_exception();
        // close the transaction without its rows and make the cache say what the database says again: the batch's
        // pairs were marked as present above and must not be skipped by a later Match().  In a caller's transaction the
        // rows are the caller's to roll back (Match() rethrows this error from its Flush()) -- no ROLLBACK is issued here, but
        // the marks are undone all the same: the cache re-reads which pairs the connection really holds (a caller
        // who rolls back and calls Match() again found the never-written pairs still marked, and they were skipped).
        try {
          cache->RollbackTransaction(callers_transaction ? false : open);
        } catch (...) {
        }
      }
    });
    return true;
  }

  // Makes the images the pair lists refer to resident on every device (replicated: SURVEY 8e); keeps what is
  // already there when the union fits cache_size images, otherwise replaces it.
  bool EnsureResident(const PairList& a, const PairList& b) {
    std::vector<image_id_t> needed;
    bool all_there = true;
    {
      std::unordered_set<image_id_t> seen;
      for (const PairList* list : {&a, &b})
        for (const auto& pr : *list)
          for (image_id_t id : {pr.first, pr.second})
            if (seen.insert(id).second) {
              needed.push_back(id);
              if (!image_index_.count(id)) all_there = false;
            }
    }
    if (all_there) return true;
    // Union fits the cache: the resident images stay where they are and only the missing ones are uploaded
    // (dsm_append_images) -- consecutive blocks of ExhaustiveFeatureMatcher share half of their images.  Otherwise the
    // set is replaced by what this call needs.
    const bool append = !image_ids_.empty() && image_ids_.size() + needed.size() <= std::max(max_resident_, needed.size()) + CountResident(needed);
    std::vector<image_id_t> ids;
    for (image_id_t id : needed)
      if (!append || !image_index_.count(id)) ids.push_back(id);
    std::sort(ids.begin(), ids.end());
    const uint32_t n = static_cast<uint32_t>(ids.size());
    std::vector<uint32_t> nfeat(n);
    std::vector<const uint8_t*> desc(n);
    std::vector<const float*> kp(n);
    std::vector<dsm_camera> cams(n);
    // staging copies when the cache cannot promise stable references (a plain LRU evicts on the next Get)
    std::vector<std::vector<uint8_t>> desc_copy(Traits::kCachePinsRequested ? 0 : n);
    std::vector<std::vector<float>> kp_copy(Traits::kCachePinsRequested ? 0 : n);
    uint32_t stride = 6;
    for (uint32_t i = 0; i < n; ++i) {
      const image_id_t id = ids[i];
      size_t rows = 0, cols = 128, nk = 0;
      uint32_t st = 6;
      const uint8_t* dptr = Traits::DescriptorData(cache_->GetDescriptors(id), &rows, &cols);
      if (!Traits::kCachePinsRequested) {
        desc_copy[i].assign(dptr, dptr + rows * cols);
        dptr = desc_copy[i].data();
      }
      const float* kptr = Traits::KeypointData(cache_->GetKeypoints(id), &nk, &st);
      if (!Traits::kCachePinsRequested) {
        kp_copy[i].assign(kptr, kptr + nk * st);
        kptr = kp_copy[i].data();
      }
      if (rows != nk || (rows && cols != 128)) {
        last_error_ = "keypoints/descriptors mismatch for image " + std::to_string(id);
        Traits::ReleasePins(cache_);
        return false;
      }
      stride = st;
      nfeat[i] = static_cast<uint32_t>(rows);
      desc[i] = dptr;
      kp[i] = nk ? kptr : nullptr;
      std::memset(&cams[i], 0, sizeof(dsm_camera));
      Traits::ToDsmCamera(cache_->GetCamera(Traits::CameraIdOf(cache_->GetImage(id))), &cams[i]);
    }
    std::vector<float> dummy(8, 0.f);
    for (uint32_t i = 0; i < n; ++i)
      if (!kp[i]) kp[i] = dummy.data();
    // every device gets every image of the list (the pair list is what is sharded), in parallel
    std::vector<int> rcs(ctxs_.size(), DSM_OK);
    std::vector<std::thread> th;
    for (size_t d = 0; d < ctxs_.size(); ++d)
      th.emplace_back([&, d]() {
        rcs[d] = append ? dsm_append_images(ctxs_[d], n, nfeat.data(), desc.data(), kp.data(), stride, cams.data())
                        : dsm_set_images(ctxs_[d], n, nfeat.data(), desc.data(), kp.data(), stride, cams.data());
      });
    for (auto& t : th) t.join();
    Traits::ReleasePins(cache_);
    for (size_t d = 0; d < ctxs_.size(); ++d)
      if (rcs[d] != DSM_OK) {
        last_error_ = dsm_last_error(ctxs_[d]);  // e.g. a camera model id the reference does not know either
        image_ids_.clear();
        image_index_.clear();
        return false;
      }
    if (!append) {
      image_ids_.clear();
      image_nfeat_.clear();
      image_index_.clear();
    }
    for (uint32_t i = 0; i < n; ++i) {
      image_index_[ids[i]] = static_cast<uint32_t>(image_ids_.size());
      image_ids_.push_back(ids[i]);
      image_nfeat_.push_back(nfeat[i]);
    }
    return true;
  }
  size_t CountResident(const std::vector<image_id_t>& ids) const {
    size_t c = 0;
    for (image_id_t id : ids) c += image_index_.count(id);
    return c;
  }

  typename Traits::Options options_;
  typename Traits::Database* database_;
  typename Traits::Cache* cache_;
  bool is_setup_ = false;
  std::vector<dsm_ctx*> ctxs_;  // one per device of gpu_index
  std::vector<int> devices_;
  size_t max_resident_ = 0;     // cache_size of the FeatureMatcherCache
  std::vector<image_id_t> image_ids_;  // device image index -> image_id
  std::vector<uint32_t> image_nfeat_;
  std::unordered_map<image_id_t, uint32_t> image_index_;  // image_id -> device image index
  std::string last_error_;
  Timings timings_ = Timings();
  DeviceGatherApi gather_api_;
  void* gather_ = nullptr;  // dsm_gather over ctxs_[0 .. gather_ranks_)
  size_t gather_ranks_ = 0;
  std::thread writer_;  // at most one write-back in flight
  std::exception_ptr writer_error_;
};

}  // namespace dagsfm_amd
#endif  // DAGSFM_AMD_HOST_SIFT_FEATURE_MATCHER_IMPL_H_
