// ref_flann_shim.cpp -- the reference's OWN visual-word search, compiled from where it lies.
//
// TEST INFRASTRUCTURE ONLY.  Built by `make -C oracle ref` into oracle/_ref/libflann_ref.so, only where /root/reference
// exists; the library travels to the GPU box, the sources never enter this repository.
//
// VisualIndex::FindWordIds (/root/reference/src/retrieval/visual_index.h:695-738) asks a
// flann::AutotunedIndex<flann::L2<uint8_t>> for the nearest visual words.  FLANN is vendored, header-only, under
// /root/reference/lib/FLANN (flann.hpp, algorithms/{autotuned,kmeans,kdtree}_index.h, ext/lz4*.c) and needs nothing but a
// C++ compiler -- unlike the rest of the retrieval code, which includes Eigen and glog.  This shim exports exactly the
// calls visual_index.h makes:
//   Build        :517-521   AutotunedIndexParams["target_precision"], buildIndex(visual_words_)
//   Quantize     :624-665   flann::hierarchicalClustering<L2<uint8_t>> with KMeansIndexParams branching / iterations /
//                           FLANN_CENTERS_KMEANSPP, centers rounded to the descriptor type
//   Write        :600-607   fopen(path, "ab"); saveIndex(fout)
//   Read         :564-574   AutotunedIndex(visual_words_); fseek(fin, file_offset); loadIndex(fin); ftell(fin)
//   FindWordIds  :695-738   knnSearch(query, indices, distances, num_neighbors, SearchParams(num_checks)) with cores
// The search over a LOADED index is deterministic (loadIndex restores the trees; the search itself draws no random
// numbers), which is what makes it usable as a checker: tests/test_retrieval_flann.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeinfo>

// flann_ref_build_forced below needs AutotunedIndex's private state (bestParams_, bestIndex_): the test wants the
// reference's OWN serialisation of a kd-tree / k-means index over a small vocabulary, and the autotuner -- which decides
// by wall-clock timings -- picks linear search for anything small enough to be a unit test.
#define private public
#define protected public
#include "FLANN/flann.hpp"
#undef private
#undef protected

namespace {

typedef flann::L2<uint8_t> Dist;
typedef flann::AutotunedIndex<Dist> AutoIndex;

struct RefIndex {
  std::vector<uint8_t> words;  // the index refers to the matrix it was built / loaded over: keep it alive
  flann::Matrix<uint8_t> matrix;
  AutoIndex* index = nullptr;
  int num_checks = 256;  // retrieval::VisualIndex<>::QueryOptions::num_checks (visual_index.h:88)
  int cores = 1;
  ~RefIndex() { delete index; }
};

}  // namespace

extern "C" {

// flann's generators draw from rand(): a fixed seed makes Build / Quantize repeatable inside one process
void flann_ref_seed(unsigned seed) { flann::seed_random(seed); }

// VisualIndex::Quantize (visual_index.h:624-665).  Returns the number of centers written to out_words ([.][128] uint8).
int flann_ref_quantize(const uint8_t* descriptors, uint32_t n, int num_visual_words, int branching, int num_iterations, uint8_t* out_words) {
  const flann::Matrix<uint8_t> descriptor_matrix(const_cast<uint8_t*>(descriptors), n, 128);
  std::vector<Dist::ResultType> centers_data(static_cast<size_t>(num_visual_words) * 128);
  flann::Matrix<Dist::ResultType> centers(centers_data.data(), num_visual_words, 128);
  flann::KMeansIndexParams index_params;
  index_params["branching"] = branching;
  index_params["iterations"] = num_iterations;
  index_params["centers_init"] = flann::FLANN_CENTERS_KMEANSPP;
  const int num_centers = flann::hierarchicalClustering<Dist>(descriptor_matrix, centers, index_params);
  for (size_t i = 0; i < static_cast<size_t>(num_centers) * 128; ++i) out_words[i] = static_cast<uint8_t>(std::round(centers_data[i]));
  return num_centers;
}

// VisualIndex::Build's index part (visual_index.h:517-521)
void* flann_ref_build(const uint8_t* words, uint32_t num_words, float target_precision) {
  RefIndex* r = new RefIndex();
  r->words.assign(words, words + static_cast<size_t>(num_words) * 128);
  r->matrix = flann::Matrix<uint8_t>(r->words.data(), num_words, 128);
  flann::AutotunedIndexParams index_params;
  index_params["target_precision"] = target_precision;
  r->index = new AutoIndex(index_params);
  r->index->buildIndex(r->matrix);
  return r;
}

// The autotuner's decision replaced by a given one, everything else as AutotunedIndex::buildIndex() does it
// (autotuned_index.h:135-155): algorithm 1 = randomised kd-trees (p1 = trees), 2 = hierarchical k-means (p1 = branching,
// p2 = iterations, centers_init random, cb_index 0.2 -- the grid optimizeKMeans / optimizeKDTree explore), 0 = linear.
// saveIndex then writes exactly what it writes after a tuned build that chose these parameters.
void* flann_ref_build_forced(const uint8_t* words, uint32_t num_words, int algorithm, int p1, int p2, int autotuned_checks) {
  RefIndex* r = new RefIndex();
  r->words.assign(words, words + static_cast<size_t>(num_words) * 128);
  r->matrix = flann::Matrix<uint8_t>(r->words.data(), num_words, 128);
  flann::AutotunedIndexParams index_params;
  r->index = new AutoIndex(r->matrix, index_params);
  flann::IndexParams best;
  if (algorithm == 1) {
    best["algorithm"] = flann::FLANN_INDEX_KDTREE;
    best["trees"] = p1;
  } else if (algorithm == 2) {
    best["algorithm"] = flann::FLANN_INDEX_KMEANS;
    best["centers_init"] = flann::FLANN_CENTERS_RANDOM;
    best["iterations"] = p2;
    best["branching"] = p1;
    best["cb_index"] = 0.2f;
  } else {
    best["algorithm"] = flann::FLANN_INDEX_LINEAR;
  }
  r->index->bestParams_ = best;
  const flann::flann_algorithm_t index_type = flann::get_param<flann::flann_algorithm_t>(best, "algorithm");
  r->index->bestIndex_ = flann::create_index_by_type<Dist>(index_type, r->matrix, best, Dist());
  r->index->bestIndex_->buildIndex();
  r->index->bestSearchParams_.checks = autotuned_checks;
  r->index->speedup_ = 1.0f;
  r->index->bestParams_["search_params"] = r->index->bestSearchParams_;
  r->index->bestParams_["speedup"] = r->index->speedup_;
  return r;
}

// VisualIndex::Write's middle section (visual_index.h:600-607): appends the serialised index to `path`.
// Returns the file size after the append (= the offset at which the inverted index starts), -1 on error.
long flann_ref_save_append(void* h, const char* path) {
  FILE* fout = fopen(path, "ab");
  if (!fout) return -1;
  static_cast<RefIndex*>(h)->index->saveIndex(fout);
  fclose(fout);
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, 0, SEEK_END);
  const long size = ftell(f);
  fclose(f);
  return size;
}

// VisualIndex::Read's middle section (visual_index.h:560-574): the index over `words`, restored from `path` at
// `file_offset`; *end_offset = ftell after loadIndex, i.e. where the reference goes on to read the inverted index.
void* flann_ref_load(const uint8_t* words, uint32_t num_words, const char* path, long file_offset, long* end_offset) {
  RefIndex* r = new RefIndex();
  r->words.assign(words, words + static_cast<size_t>(num_words) * 128);
  r->matrix = flann::Matrix<uint8_t>(r->words.data(), num_words, 128);
  r->index = new AutoIndex(r->matrix);
  FILE* fin = fopen(path, "rb");
  if (!fin) {
    delete r;
    return nullptr;
  }
  fseek(fin, file_offset, SEEK_SET);
  try {
    r->index->loadIndex(fin);
  } catch (...) {
    fclose(fin);
    delete r;
    return nullptr;
  }
  if (end_offset) *end_offset = ftell(fin);
  fclose(fin);
  return r;
}

void flann_ref_destroy(void* h) { delete static_cast<RefIndex*>(h); }

void flann_ref_set_search(void* h, int num_checks, int cores) {
  static_cast<RefIndex*>(h)->num_checks = num_checks;
  static_cast<RefIndex*>(h)->cores = cores <= 0 ? 1 : cores;
}

// VisualIndex::FindWordIds (visual_index.h:695-738): out_ids [n][k] row-major, kInvalidWordId (INT_MAX) where FLANN
// returned fewer than k neighbours (word_ids.setConstant(kInvalidWordId) before the search, size_t -> int cast after);
// out_dists may be null.
void flann_ref_knn(void* h, const uint8_t* descriptors, uint32_t n, uint32_t k, int32_t* out_ids, float* out_dists) {
  RefIndex* r = static_cast<RefIndex*>(h);
  if (n == 0 || k == 0) return;
  std::vector<size_t> word_ids(static_cast<size_t>(n) * k, static_cast<size_t>(2147483647));
  flann::Matrix<size_t> indices(word_ids.data(), n, k);
  std::vector<Dist::ResultType> distance_matrix(static_cast<size_t>(n) * k, 0);
  flann::Matrix<Dist::ResultType> distances(distance_matrix.data(), n, k);
  const flann::Matrix<uint8_t> query(const_cast<uint8_t*>(descriptors), n, 128);
  flann::SearchParams search_params(r->num_checks);
  search_params.cores = r->cores;
  r->index->knnSearch(query, indices, distances, k, search_params);
  for (size_t i = 0; i < word_ids.size(); ++i) {
    out_ids[i] = static_cast<int32_t>(static_cast<int>(word_ids[i]));
    if (out_dists) out_dists[i] = distance_matrix[i];
  }
}

// the same search in the signature oracle_retrieval_set_word_search takes (oracle/retrieval.cc): user = the handle
void flann_ref_find_word_ids(void* user, const uint8_t* descriptors, uint32_t n, uint32_t k, int32_t* out_ids) {
  flann_ref_knn(user, descriptors, n, k, out_ids, nullptr);
}

// what the autotuner chose ("algorithm" of bestParams_: 0 linear, 1 kd-trees, 2 k-means, ...), for the test's report
int flann_ref_algorithm(void* h) {
  const flann::IndexParams p = static_cast<RefIndex*>(h)->index->getParameters();
  const flann::IndexParams::const_iterator it = p.find("algorithm");
  return it == p.end() ? -1 : static_cast<int>(it->second.cast<flann::flann_algorithm_t>());
}

}  // extern "C"
