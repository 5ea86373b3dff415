// flann_index.h -- the FLANN index inside the reference's vocabulary-tree file: where it ends, and the reference's own
// visual-word search over it.
//
// VisualIndex<>::Write (/root/reference/src/retrieval/visual_index.h:586-614) stores, between the visual words and the
// inverted index, whatever flann::AutotunedIndex<flann::L2<uint8_t>>::saveIndex wrote
// (/root/reference/lib/FLANN/algorithms/autotuned_index.h:209-217): two archives of FLANN's serialisation
// (lib/FLANN/util/serialization.h:376-547) -- the autotuned index's own record, then the index the autotuner chose
// (linear, randomised kd-trees or a hierarchical k-means tree).  VisualIndex::FindWordIds (:695-738) then asks that LOADED
// index for approximate nearest words (`num_checks` leaves): its answer is a function of the file, not of any random
// seed.  FlannIndex restates that search -- LZ4 block decoding, the archive layout, KDTreeIndex::getNeighbors
// (algorithms/kdtree_index.h:543-617), KMeansIndex::findNN (algorithms/kmeans_index.h:717-833), LinearIndex
// (algorithms/linear_index.h:130-146), KNNSimpleResultSet (util/result_set.h:101-199), L2<uint8_t> in float
// (algorithms/dist.h:133-178) -- so that VocabTreeMatching can return the reference's word ids bit for bit
// (`word_search = flann`), where the device's exact search (the default) returns the true nearest words.
// tests/test_retrieval_flann.py holds it to the reference's own FLANN compiled from where it lies (oracle/_ref).
#ifndef DAGSFM_AMD_HOST_FLANN_INDEX_H_
#define DAGSFM_AMD_HOST_FLANN_INDEX_H_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dagsfm_mi355x.h"

namespace dagsfm_amd {

// flann::IndexHeaderStruct (lib/FLANN/util/serialization.h:15-24): 80 bytes on the LP64 targets the reference builds for
struct FlannIndexHeader {
  char signature[24];  // "FLANN_INDEX_v1.1"
  char version[16];
  int32_t data_type;   // flann_datatype_t: FLANN_UINT8 = 4 for L2<uint8_t>
  int32_t index_type;  // flann_algorithm_t: 0 linear, 1 kd-trees, 2 k-means, 255 autotuned
  uint64_t rows, cols;
  uint64_t compression;       // 1: LZ4 blocks
  uint64_t first_block_size;  // compressed bytes that follow the header
};
static_assert(sizeof(FlannIndexHeader) == 80, "flann::IndexHeaderStruct layout");

// One archive as SaveArchive frames it (serialization.h:412-479): header, first_block_size bytes, then (u64 size, size
// bytes) per further 64 KiB block, then a u64 zero.  *at: in = the archive's first byte, out = the byte after its
// terminating zero (where LoadArchive's destructor leaves the stream, :721-733).  False when the bytes at *at are not such
// an archive (wrong signature, not LZ4-framed, a block that runs past the end).
bool FlannSkipArchive(const uint8_t* buf, size_t size, size_t* at);

// The same walk, decoding: `out` receives the archive's logical byte stream (the 80 header bytes as stored, then the
// LZ4-decoded payload of every block in order -- LoadArchive::initBlock / loadBlock / preparePtr, :631-719).
bool FlannReadArchive(const uint8_t* buf, size_t size, size_t* at, std::vector<uint8_t>* out);

class FlannIndex {
 public:
  enum Algorithm { kLinear = 0, kKdTree = 1, kKMeans = 2 };
  static const int32_t kInvalidWordId = 2147483647;  // InvertedIndexType::kInvalidWordId (inverted_index.h:69-70)

  // AutotunedIndex::loadIndex (autotuned_index.h:219-230) from the bytes at *at of a vocabulary file, over `words`
  // ([num_words][128], must outlive the object): *at ends where loadIndex leaves the stream.
  bool Load(const uint8_t* buf, size_t size, size_t* at, const uint8_t* words, uint32_t num_words);
  // VisualIndex::FindWordIds (visual_index.h:695-738): out_ids [n][k] row-major, kInvalidWordId where the search returned
  // fewer than k words; out_dists ([n][k] float, FLANN's squared L2) may be null.  k <= 250 (beyond that FLANN switches to
  // another result set, nn_index.h:316-340; VocabTreeMatching asks for 1 and num_nearest_neighbors = 5).
  bool FindWordIds(const uint8_t* descriptors, uint32_t n, uint32_t k, int num_checks, int num_threads, int32_t* out_ids,
                   float* out_dists) const;
  // The loaded trees as the flat arrays dsm_retrieval_set_flann_index takes (pointers into this object: it must outlive the call);
  // num_checks: SearchParams::checks, -2 = the index's own autotuned estimate (FLANN_CHECKS_AUTOTUNED).  False for an
  // unlimited walk (-1) on a tree index, like FindWordIds.
  bool Export(int num_checks, dsm_flann_index* out) const;
  int algorithm() const { return algorithm_; }
  int autotuned_checks() const { return autotuned_checks_; }
  const std::string& error() const { return error_; }

 private:
  struct KdNode {
    int32_t divfeat;  // inner node: the dimension; leaf: the point's index
    float divval;
    int32_t child1, child2;  // node indices, -1 / -1 for a leaf
  };
  struct KmNode {
    uint64_t pivot;  // offset into pivots_ (128 floats)
    float radius, variance;
    int32_t size;
    uint32_t first_child, num_childs;  // into km_childs_
    uint64_t first_point;              // into km_points_ (size entries) when num_childs == 0
  };
  struct ResultSet;
  struct Branch;
  class BranchHeap;
  struct SearchScratch;
  void SearchOne(const uint8_t* vec, int num_checks, ResultSet* result, SearchScratch* scratch) const;
  void KdSearchLevel(ResultSet* result, const uint8_t* vec, int32_t node, float mindist, int* check_count, int max_check,
                     BranchHeap* heap, std::vector<uint64_t>* checked) const;
  void KmFindNN(int32_t node, ResultSet* result, const uint8_t* vec, int* checks, int max_checks, BranchHeap* heap) const;
  bool LoadImpl(const uint8_t* buf, size_t size, size_t* at, const uint8_t* words, uint32_t num_words);
  bool ReadKdNode(const uint8_t* s, size_t n, size_t* at, int32_t* index, int depth);
  bool ReadKmNode(const uint8_t* s, size_t n, size_t* at, int32_t* index, int depth);

  int algorithm_ = -1;
  int autotuned_checks_ = 0;
  const uint8_t* words_ = nullptr;
  uint32_t num_words_ = 0;
  std::vector<uint8_t> own_dataset_;  // save_dataset archives carry their own copy of the points
  // kd-trees
  std::vector<KdNode> kd_nodes_;
  std::vector<int32_t> kd_roots_;
  // k-means tree
  int32_t branching_ = 0;
  float cb_index_ = 0.0f;
  std::vector<KmNode> km_nodes_;
  std::vector<int32_t> km_childs_;
  std::vector<uint64_t> km_points_;
  std::vector<float> pivots_;
  int32_t km_root_ = -1;
  std::string error_;
};

}  // namespace dagsfm_amd

#endif  // DAGSFM_AMD_HOST_FLANN_INDEX_H_
