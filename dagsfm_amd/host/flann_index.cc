// flann_index.cc -- see flann_index.h.  Every routine cites the FLANN code it restates (/root/reference/lib/FLANN/...).
#include "flann_index.h"

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstring>
#include <new>
#include <limits>
#include <thread>

namespace dagsfm_amd {

namespace {
const size_t kBlockBytes = 1024 * 64;  // BLOCK_BYTES, util/serialization.h:374
// LZ4_COMPRESSBOUND(BLOCK_BYTES): LoadArchive::loadBlock refuses larger blocks (serialization.h:677-681)
const uint64_t kMaxCompressedBlock = kBlockBytes + kBlockBytes / 255 + 16;
const int kVecLen = 128;
const size_t kMaxDecodedArchive = size_t(16) << 30;  // 16 GiB: ~ 8 kd-trees / a k-means tree with float centres over 2^24 words
const int kMaxTreeDepth = 4096;  // recursion guard of the loader (FLANN splits at the mean / by cluster: real trees are tens of levels deep; a damaged file must not overflow the stack)

// One LZ4 block (the format of ext/lz4.c's LZ4_decompress_safe_continue): sequences of token, literal run, 2-byte
// little-endian match offset, match length; the last sequence ends after its literals.  Matches may reach back into
// everything decoded so far (`out`: the previous block is the dictionary of the next, serialization.h:459-463, 689-690).
bool Lz4DecodeBlock(const uint8_t* src, size_t n, std::vector<uint8_t>* out) {
  const size_t start = out->size();
  size_t ip = 0;
  for (;;) {
    if (ip >= n) return false;
    const unsigned token = src[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      unsigned b;
      do {
        if (ip >= n) return false;
        b = src[ip++];
        lit += b;
      } while (b == 255);
    }
    if (lit > n - ip || out->size() - start + lit > kBlockBytes) return false;
    out->insert(out->end(), src + ip, src + ip + lit);
    ip += lit;
    if (ip == n) return true;  // the last sequence carries literals only
    if (n - ip < 2) return false;
    const size_t offset = static_cast<size_t>(src[ip]) | (static_cast<size_t>(src[ip + 1]) << 8);
    ip += 2;
    if (offset == 0 || offset > out->size()) return false;
    size_t ml = token & 15u;
    if (ml == 15) {
      unsigned b;
      do {
        if (ip >= n) return false;
        b = src[ip++];
        ml += b;
      } while (b == 255);
    }
    ml += 4;
    if (out->size() - start + ml > kBlockBytes) return false;
    size_t from = out->size() - offset;
    for (size_t i = 0; i < ml; ++i) out->push_back((*out)[from + i]);  // byte by byte: a match may overlap its own output
  }
}

template <typename T>
bool Get(const uint8_t* s, size_t n, size_t* at, T* v) {
  if (*at > n || n - *at < sizeof(T)) return false;
  std::memcpy(v, s + *at, sizeof(T));
  *at += sizeof(T);
  return true;
}

// L2<unsigned char>::operator() with ResultType = float (algorithms/dist.h:133-178), worst_dist = -1 (no early exit):
// four differences per step, their squares summed left to right, one addition into the running sum.
inline float DistU8U8(const uint8_t* a, const uint8_t* b) {
  float result = 0.0f;
  for (int i = 0; i < kVecLen; i += 4) {
    const float d0 = static_cast<float>(static_cast<int>(a[i]) - static_cast<int>(b[i]));
    const float d1 = static_cast<float>(static_cast<int>(a[i + 1]) - static_cast<int>(b[i + 1]));
    const float d2 = static_cast<float>(static_cast<int>(a[i + 2]) - static_cast<int>(b[i + 2]));
    const float d3 = static_cast<float>(static_cast<int>(a[i + 3]) - static_cast<int>(b[i + 3]));
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  return result;
}
// the same functor with a float second operand (a k-means pivot): a[i] - b[i] is a float subtraction
inline float DistU8F32(const uint8_t* a, const float* b) {
  float result = 0.0f;
  for (int i = 0; i < kVecLen; i += 4) {
    const float d0 = static_cast<float>(a[i]) - b[i];
    const float d1 = static_cast<float>(a[i + 1]) - b[i + 1];
    const float d2 = static_cast<float>(a[i + 2]) - b[i + 2];
    const float d3 = static_cast<float>(a[i + 3]) - b[i + 3];
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  return result;
}
}  // namespace

bool FlannSkipArchive(const uint8_t* buf, size_t size, size_t* at) { return FlannReadArchive(buf, size, at, nullptr); }

bool FlannReadArchive(const uint8_t* buf, size_t size, size_t* at, std::vector<uint8_t>* out) {
  size_t pos = *at;
  if (pos > size || size - pos < sizeof(FlannIndexHeader)) return false;
  FlannIndexHeader head;
  std::memcpy(&head, buf + pos, sizeof(head));
  // load_header's check (util/saving.h:114-119) is the signature up to its version suffix; LoadArchive::initBlock reads the
  // v1.1 framing (a "v1.0" archive takes the whole rest of the file as ONE block and cannot be followed by anything)
  if (std::memcmp(head.signature, "FLANN_INDEX_v1.1", 16) != 0 || head.compression != 1) return false;
  if (head.first_block_size == 0 || head.first_block_size >= kMaxCompressedBlock) return false;
  pos += sizeof(head);
  if (size - pos < head.first_block_size) return false;
  std::vector<uint8_t> payload;  // the LZ4 history starts BEHIND the header (SaveArchive::flushBlock, :424-427)
  if (out && !Lz4DecodeBlock(buf + pos, static_cast<size_t>(head.first_block_size), &payload)) return false;
  pos += static_cast<size_t>(head.first_block_size);
  for (;;) {
    if (size - pos < 8) return false;
    uint64_t block = 0;
    std::memcpy(&block, buf + pos, 8);
    pos += 8;
    if (block == 0) break;
    if (block >= kMaxCompressedBlock || size - pos < block) return false;
    if (out && !Lz4DecodeBlock(buf + pos, static_cast<size_t>(block), &payload)) return false;
    // an LZ4 block of a few bytes can expand to 64 KiB: without a bound a crafted file of some MB decodes to tens of GB
    // (ADVICE r04).  A real index is its points' references plus a tree over them: generously below this.
    if (payload.size() > kMaxDecodedArchive) return false;
    pos += static_cast<size_t>(block);
  }
  if (out) {
    out->assign(buf + *at, buf + *at + sizeof(head));
    out->insert(out->end(), payload.begin(), payload.end());
  }
  *at = pos;
  return true;
}

// ------------------------------------------------------------------------------------------------ loading
namespace {
// NNIndex<Distance>::serialize (algorithms/nn_index.h:226-294) behind the 80 header bytes: size_, veclen_, size_at_build_,
// save_dataset (+ the points), last_id_, ids_, removed_ (+ the bitset), removed_count_.
bool ReadBase(const uint8_t* s, size_t n, size_t* at, int expected_type, bool has_points, uint32_t num_words, std::vector<uint8_t>* dataset,
              std::string* error) {
  FlannIndexHeader head;
  if (!Get(s, n, at, &head)) return false;
  if (head.data_type != 4 /* FLANN_UINT8 */) {
    *error = "FLANN index of another data type";
    return false;
  }
  if (head.index_type != expected_type) {
    *error = "FLANN index type does not match the autotuned record";
    return false;
  }
  uint64_t size = 0, veclen = 0, size_at_build = 0, last_id = 0, n_ids = 0, removed_count = 0;
  uint8_t save_dataset = 0, removed = 0;
  if (!Get(s, n, at, &size) || !Get(s, n, at, &veclen) || !Get(s, n, at, &size_at_build) || !Get(s, n, at, &save_dataset)) return false;
  // (the autotuned index's own record holds no points: its size_ and veclen_ are 0; the index it chose holds the words)
  if (has_points ? (size != num_words || veclen != kVecLen) : (size != 0 || save_dataset)) {
    *error = "FLANN index was built over another vocabulary (rows / cols differ)";
    return false;
  }
  if (save_dataset) {
    if (n - *at < size * veclen) return false;
    dataset->assign(s + *at, s + *at + size * veclen);
    *at += static_cast<size_t>(size * veclen);
  }
  if (!Get(s, n, at, &last_id) || !Get(s, n, at, &n_ids)) return false;
  if (n_ids > (n - *at) / 8) return false;
  if (n_ids != 0) {  // ids_ is only filled once points have been removed (nn_index.h: removePoint); VisualIndex never does
    *error = "FLANN index with remapped point ids is not supported";
    return false;
  }
  if (!Get(s, n, at, &removed)) return false;
  if (removed) {
    *error = "FLANN index with removed points is not supported";
    return false;
  }
  return Get(s, n, at, &removed_count);
}
}  // namespace

// KDTreeIndex::Node::serialize (algorithms/kdtree_index.h:318-346): divfeat, divval, leaf flag, then both children
bool FlannIndex::ReadKdNode(const uint8_t* s, size_t n, size_t* at, int32_t* index, int depth) {
  if (depth > kMaxTreeDepth) return false;
  KdNode node;
  uint8_t leaf = 0;
  if (!Get(s, n, at, &node.divfeat) || !Get(s, n, at, &node.divval) || !Get(s, n, at, &leaf)) return false;
  node.child1 = node.child2 = -1;
  if (node.divfeat < 0 || static_cast<uint32_t>(node.divfeat) >= (leaf ? num_words_ : static_cast<uint32_t>(kVecLen))) return false;
  *index = static_cast<int32_t>(kd_nodes_.size());
  kd_nodes_.push_back(node);
  if (!leaf) {
    int32_t c1 = -1, c2 = -1;
    if (!ReadKdNode(s, n, at, &c1, depth + 1) || !ReadKdNode(s, n, at, &c2, depth + 1)) return false;
    kd_nodes_[*index].child1 = c1;
    kd_nodes_[*index].child2 = c2;
  }
  return true;
}

// KMeansIndex::Node::serialize (algorithms/kmeans_index.h:412-447): pivot, radius, variance, size, childs_size, then the
// points (vector<PointInfo>: count + one size_t index each, :354-364) or the children
bool FlannIndex::ReadKmNode(const uint8_t* s, size_t n, size_t* at, int32_t* index, int depth) {
  if (depth > kMaxTreeDepth) return false;
  KmNode node;
  if (n - *at < kVecLen * sizeof(float)) return false;
  node.pivot = pivots_.size();
  pivots_.resize(pivots_.size() + kVecLen);
  std::memcpy(pivots_.data() + node.pivot, s + *at, kVecLen * sizeof(float));
  *at += kVecLen * sizeof(float);
  uint64_t childs_size = 0;
  if (!Get(s, n, at, &node.radius) || !Get(s, n, at, &node.variance) || !Get(s, n, at, &node.size) || !Get(s, n, at, &childs_size)) return false;
  node.first_child = 0;
  node.num_childs = 0;
  node.first_point = km_points_.size();
  *index = static_cast<int32_t>(km_nodes_.size());
  km_nodes_.push_back(node);
  if (childs_size == 0) {
    uint64_t count = 0;
    if (!Get(s, n, at, &count) || count > (n - *at) / 8) return false;
    if (node.size < 0 || static_cast<uint64_t>(node.size) > count) return false;  // findNN walks `size` entries of `points`
    for (uint64_t i = 0; i < count; ++i) {
      uint64_t point = 0;
      if (!Get(s, n, at, &point) || point >= num_words_) return false;
      km_points_.push_back(point);
    }
  } else {
    // exploreNodeBranches indexes childs[0 .. branching_) (:804-830): an inner node has exactly branching_ children
    if (childs_size != static_cast<uint64_t>(branching_)) return false;
    std::vector<int32_t> childs(static_cast<size_t>(childs_size));
    for (size_t i = 0; i < childs.size(); ++i)
      if (!ReadKmNode(s, n, at, &childs[i], depth + 1)) return false;
    km_nodes_[*index].first_child = static_cast<uint32_t>(km_childs_.size());
    km_nodes_[*index].num_childs = static_cast<uint32_t>(childs.size());
    km_childs_.insert(km_childs_.end(), childs.begin(), childs.end());
  }
  return true;
}

bool FlannIndex::Load(const uint8_t* buf, size_t size, size_t* at, const uint8_t* words, uint32_t num_words) {
  try {
    return LoadImpl(buf, size, at, words, num_words);
  } catch (const std::bad_alloc&) {  // a file that asks for more memory than the host has is a bad file, not a crash
    algorithm_ = -1;
    error_ = "out of memory while reading the FLANN index";
    return false;
  }
}

bool FlannIndex::LoadImpl(const uint8_t* buf, size_t size, size_t* at, const uint8_t* words, uint32_t num_words) {
  algorithm_ = -1;
  words_ = words;
  num_words_ = num_words;
  kd_nodes_.clear();
  kd_roots_.clear();
  km_nodes_.clear();
  km_childs_.clear();
  km_points_.clear();
  pivots_.clear();
  own_dataset_.clear();
  error_.clear();
  size_t pos = *at;
  std::vector<uint8_t> a;
  // AutotunedIndex::loadIndex: its own archive ... (autotuned_index.h:180-207: base, target_precision_, build_weight_,
  // memory_weight_, sample_fraction_, index_type, bestSearchParams_.checks)
  if (!FlannReadArchive(buf, size, &pos, &a)) {
    error_ = "not a FLANN v1.1 archive";
    return false;
  }
  size_t p = 0;
  std::vector<uint8_t> dataset;
  float tuning[4];
  int32_t index_type = -1, checks = 0;
  if (!ReadBase(a.data(), a.size(), &p, 255 /* FLANN_INDEX_AUTOTUNED */, false, num_words, &dataset, &error_) || !Get(a.data(), a.size(), &p, &tuning) ||
      !Get(a.data(), a.size(), &p, &index_type) || !Get(a.data(), a.size(), &p, &checks)) {
    if (error_.empty()) error_ = "truncated autotuned-index record";
    return false;
  }
  autotuned_checks_ = checks;
  // ... then bestIndex_->loadIndex(stream) (:225-229)
  if (!FlannReadArchive(buf, size, &pos, &a)) {
    error_ = "the chosen index is not a FLANN v1.1 archive";
    return false;
  }
  p = 0;
  if (index_type != kLinear && index_type != kKdTree && index_type != kKMeans) {
    error_ = "FLANN index type " + std::to_string(index_type) + " is not one the autotuner chooses";
    return false;
  }
  if (!ReadBase(a.data(), a.size(), &p, index_type, true, num_words, &own_dataset_, &error_)) {
    if (error_.empty()) error_ = "truncated index record";
    return false;
  }
  if (!own_dataset_.empty()) words_ = own_dataset_.data();
  bool ok = true;
  if (index_type == kKdTree) {  // KDTreeIndex::serialize, kdtree_index.h:169-193
    int32_t trees = 0;
    ok = Get(a.data(), a.size(), &p, &trees) && trees >= 0 && trees <= 4096;
    for (int32_t t = 0; ok && t < trees; ++t) {
      int32_t root = -1;
      ok = ReadKdNode(a.data(), a.size(), &p, &root, 0);
      kd_roots_.push_back(root);
    }
  } else if (index_type == kKMeans) {  // KMeansIndex::serialize, kmeans_index.h:233-258
    int32_t iterations = 0, memory_counter = 0, centers_init = 0;
    ok = Get(a.data(), a.size(), &p, &branching_) && Get(a.data(), a.size(), &p, &iterations) && Get(a.data(), a.size(), &p, &memory_counter) &&
         Get(a.data(), a.size(), &p, &cb_index_) && Get(a.data(), a.size(), &p, &centers_init) && branching_ >= 2 &&
         ReadKmNode(a.data(), a.size(), &p, &km_root_, 0);
  }
  if (!ok) {
    error_ = "damaged FLANN tree";
    return false;
  }
  algorithm_ = index_type;
  *at = pos;
  return true;
}

// ------------------------------------------------------------------------------------------------ searching
// KNNSimpleResultSet<float> (util/result_set.h:101-199; FLANN_FIRST_MATCH is not defined in the translation units that
// include retrieval/visual_index.h)
struct FlannIndex::ResultSet {
  struct DistIndex {
    float dist;
    uint64_t index;
  };
  explicit ResultSet(size_t capacity) : capacity_(capacity), dist_index_(capacity, DistIndex{std::numeric_limits<float>::max(), static_cast<uint64_t>(-1)}) {
    Clear();
  }
  void Clear() {
    worst_distance_ = std::numeric_limits<float>::max();
    dist_index_[capacity_ - 1].dist = worst_distance_;
    count_ = 0;
  }
  bool Full() const { return count_ == capacity_; }
  float WorstDist() const { return worst_distance_; }
  void AddPoint(float dist, uint64_t index) {
    if (dist >= worst_distance_) return;
    if (count_ < capacity_) ++count_;
    size_t i;
    for (i = count_ - 1; i > 0; --i) {
      if (dist_index_[i - 1].dist > dist)
        dist_index_[i] = dist_index_[i - 1];
      else
        break;
    }
    dist_index_[i].dist = dist;
    dist_index_[i].index = index;
    worst_distance_ = dist_index_[capacity_ - 1].dist;
  }
  size_t capacity_, count_;
  float worst_distance_;
  std::vector<DistIndex> dist_index_;
};

// BranchStruct<NodePtr, float> (util/result_set.h:50-62) and Heap<BranchSt> (util/heap.h:47-167): a std::vector under
// std::push_heap / std::pop_heap with "t_2 < t_1" -- the same library calls, so equal keys leave in the same order
struct FlannIndex::Branch {
  int32_t node;
  float mindist;
  bool operator<(const Branch& rhs) const { return mindist < rhs.mindist; }
};
class FlannIndex::BranchHeap {
 public:
  explicit BranchHeap(int size) : length_(size), count_(0) { heap_.reserve(static_cast<size_t>(std::max(size, 0))); }
  void Insert(const Branch& value) {
    if (count_ == length_) return;  // "If heap is full, then return without adding this element."
    heap_.push_back(value);
    std::push_heap(heap_.begin(), heap_.end(), Compare());
    ++count_;
  }
  void Clear() {  // a fresh heap for the next query, the allocation kept (FLANN allocates one per search: same behaviour, no malloc)
    heap_.clear();
    count_ = 0;
  }
  bool PopMin(Branch* value) {
    if (count_ == 0) return false;
    *value = heap_[0];
    std::pop_heap(heap_.begin(), heap_.end(), Compare());
    heap_.pop_back();
    --count_;
    return true;
  }

 private:
  struct Compare {
    bool operator()(const Branch& t_1, const Branch& t_2) const { return t_2 < t_1; }
  };
  std::vector<Branch> heap_;
  int length_, count_;
};

// KDTreeIndex::searchLevel (kdtree_index.h:568-617), with_removed = false, epsError = 1 + SearchParams::eps = 1
void FlannIndex::KdSearchLevel(ResultSet* result_set, const uint8_t* vec, int32_t node_index, float mindist, int* check_count, int max_check,
                               BranchHeap* heap, std::vector<uint64_t>* checked) const {
  for (;;) {  // (the reference's tail recursion into the best child, as a loop)
    if (result_set->WorstDist() < mindist) return;
    const KdNode& node = kd_nodes_[node_index];
    if (node.child1 < 0 && node.child2 < 0) {
      const int index = node.divfeat;
      uint64_t& cell = (*checked)[static_cast<size_t>(index) / 64];
      const uint64_t bit = uint64_t(1) << (static_cast<size_t>(index) % 64);
      if ((cell & bit) != 0 || (*check_count >= max_check && result_set->Full())) return;
      cell |= bit;
      ++*check_count;
      const float dist = DistU8U8(words_ + static_cast<size_t>(index) * kVecLen, vec);
      result_set->AddPoint(dist, static_cast<uint64_t>(index));
      return;
    }
    const uint8_t val = vec[node.divfeat];
    const float diff = static_cast<float>(val) - node.divval;
    const int32_t best_child = (diff < 0) ? node.child1 : node.child2;
    const int32_t other_child = (diff < 0) ? node.child2 : node.child1;
    const float t = static_cast<float>(val) - node.divval;  // accum_dist: (a - b) * (a - b)
    const float new_distsq = mindist + t * t;
    if ((new_distsq * 1.0f < result_set->WorstDist()) || !result_set->Full()) heap->Insert(Branch{other_child, new_distsq});
    node_index = best_child;
  }
}

// KMeansIndex::findNN + exploreNodeBranches (kmeans_index.h:757-833), with_removed = false
void FlannIndex::KmFindNN(int32_t node_index, ResultSet* result, const uint8_t* vec, int* checks, int max_checks, BranchHeap* heap) const {
  for (;;) {
    const KmNode& node = km_nodes_[node_index];
    {
      const float bsq = DistU8F32(vec, pivots_.data() + node.pivot);
      const float rsq = node.radius;
      const float wsq = result->WorstDist();
      const float val = bsq - rsq - wsq;
      const float val2 = val * val - 4 * rsq * wsq;
      if ((val > 0) && (val2 > 0)) return;
    }
    if (node.num_childs == 0) {
      if (*checks >= max_checks) {
        if (result->Full()) return;
      }
      for (int i = 0; i < node.size; ++i) {
        const uint64_t index = km_points_[node.first_point + static_cast<uint64_t>(i)];
        const float dist = DistU8U8(words_ + static_cast<size_t>(index) * kVecLen, vec);
        result->AddPoint(dist, static_cast<uint64_t>(static_cast<int>(index)));  // `int index = point_info.index;`
        ++*checks;
      }
      return;
    }
    const int32_t* childs = km_childs_.data() + node.first_child;
    std::vector<float> domain_distances(static_cast<size_t>(branching_));
    int best_index = 0;
    domain_distances[0] = DistU8F32(vec, pivots_.data() + km_nodes_[childs[0]].pivot);
    for (int i = 1; i < branching_; ++i) {
      domain_distances[i] = DistU8F32(vec, pivots_.data() + km_nodes_[childs[i]].pivot);
      if (domain_distances[i] < domain_distances[best_index]) best_index = i;
    }
    for (int i = 0; i < branching_; ++i) {
      if (i != best_index) {
        domain_distances[i] -= cb_index_ * km_nodes_[childs[i]].variance;
        heap->Insert(Branch{childs[i], domain_distances[i]});
      }
    }
    node_index = childs[best_index];
  }
}

bool FlannIndex::Export(int num_checks, dsm_flann_index* out) const {
  static_assert(sizeof(KdNode) == sizeof(dsm_flann_kd_node) && offsetof(KdNode, child2) == offsetof(dsm_flann_kd_node, child2), "kd node layout");
  static_assert(sizeof(KmNode) == sizeof(dsm_flann_km_node) && offsetof(KmNode, first_point) == offsetof(dsm_flann_km_node, first_point) &&
                    offsetof(KmNode, num_childs) == offsetof(dsm_flann_km_node, num_childs),
                "k-means node layout");
  if (algorithm_ < 0 || !out) return false;
  if (num_checks == -2) num_checks = autotuned_checks_;
  if (num_checks < 0 && algorithm_ != kLinear) return false;
  std::memset(out, 0, sizeof(*out));
  out->algorithm = algorithm_;
  out->num_checks = num_checks < 0 ? 0 : num_checks;
  out->num_words = num_words_;
  out->branching = branching_;
  out->cb_index = cb_index_;
  out->km_root = km_root_;
  out->n_kd_nodes = static_cast<uint32_t>(kd_nodes_.size());
  out->n_kd_roots = static_cast<uint32_t>(kd_roots_.size());
  out->kd_nodes = reinterpret_cast<const dsm_flann_kd_node*>(kd_nodes_.data());
  out->kd_roots = kd_roots_.data();
  out->n_km_nodes = static_cast<uint32_t>(km_nodes_.size());
  out->km_nodes = reinterpret_cast<const dsm_flann_km_node*>(km_nodes_.data());
  out->n_km_childs = km_childs_.size();
  out->km_childs = km_childs_.data();
  out->n_km_points = km_points_.size();
  out->km_points = km_points_.data();
  out->n_pivot_floats = pivots_.size();
  out->pivots = pivots_.data();
  return true;
}

// What one search needs beyond the result set: the branch heap (FLANN: `new Heap<BranchSt>((int)size_)` per query) and the kd-trees'
// `checked` bitset (DynamicBitset(size_) per query).  Here one of each per WORKER, reset between queries -- the same contents at the
// start of every search without two allocations of num_words entries per descriptor.
struct FlannIndex::SearchScratch {
  BranchHeap heap;
  std::vector<uint64_t> checked;
  SearchScratch(uint32_t num_words, bool kd) : heap(static_cast<int>(num_words)), checked(kd ? num_words / 64 + 1 : 0, 0) {}
};

void FlannIndex::SearchOne(const uint8_t* vec, int num_checks, ResultSet* result, SearchScratch* scratch) const {
  if (algorithm_ == kLinear) {  // LinearIndex::findNeighbors, linear_index.h:130-146
    for (uint32_t i = 0; i < num_words_; ++i) result->AddPoint(DistU8U8(words_ + static_cast<size_t>(i) * kVecLen, vec), i);
    return;
  }
  BranchHeap& heap = scratch->heap;
  heap.Clear();
  Branch branch;
  if (algorithm_ == kKdTree) {  // KDTreeIndex::getNeighbors, kdtree_index.h:543-566
    std::vector<uint64_t>& checked = scratch->checked;
    std::fill(checked.begin(), checked.end(), 0);
    int check_count = 0;
    for (size_t t = 0; t < kd_roots_.size(); ++t) KdSearchLevel(result, vec, kd_roots_[t], 0.0f, &check_count, num_checks, &heap, &checked);
    while (heap.PopMin(&branch) && (check_count < num_checks || !result->Full()))
      KdSearchLevel(result, vec, branch.node, branch.mindist, &check_count, num_checks, &heap, &checked);
  } else {  // KMeansIndex::findNeighborsWithRemoved, kmeans_index.h:717-741
    int checks = 0;
    KmFindNN(km_root_, result, vec, &checks, num_checks, &heap);
    while (heap.PopMin(&branch) && (checks < num_checks || !result->Full())) KmFindNN(branch.node, result, vec, &checks, num_checks, &heap);
  }
}

bool FlannIndex::FindWordIds(const uint8_t* descriptors, uint32_t n, uint32_t k, int num_checks, int num_threads, int32_t* out_ids,
                             float* out_dists) const {
  if (algorithm_ < 0 || k == 0 || k > 250) return false;  // KNN_HEAP_THRESHOLD, nn_index.h:46, 316-320
  // AutotunedIndex::knnSearch (autotuned_index.h:232-246): FLANN_CHECKS_AUTOTUNED takes the stored estimate.  An exhaustive
  // walk of the trees (FLANN_CHECKS_UNLIMITED) is not something VocabTreeMatching asks for.
  if (num_checks == -2) num_checks = autotuned_checks_;
  if (num_checks < 0 && algorithm_ != kLinear) return false;
  for (size_t i = 0; i < static_cast<size_t>(n) * k; ++i) out_ids[i] = kInvalidWordId;  // word_ids.setConstant(kInvalidWordId)
  if (out_dists) std::fill(out_dists, out_dists + static_cast<size_t>(n) * k, 0.0f);
  std::atomic<bool> failed(false);  // (a worker must not std::terminate the host process: an allocation failure is reported as `false`)
  auto run = [&](uint32_t begin, uint32_t end) {
   try {
    ResultSet result(k);
    SearchScratch scratch(algorithm_ == kLinear ? 0u : num_words_, algorithm_ == kKdTree);
    for (uint32_t i = begin; i < end; ++i) {
      result.Clear();
      SearchOne(descriptors + static_cast<size_t>(i) * kVecLen, num_checks, &result, &scratch);
      const size_t m = std::min<size_t>(result.count_, k);
      for (size_t j = 0; j < m; ++j) {
        out_ids[static_cast<size_t>(i) * k + j] = static_cast<int32_t>(static_cast<int>(result.dist_index_[j].index));  // word_ids.cast<int>()
        if (out_dists) out_dists[static_cast<size_t>(i) * k + j] = result.dist_index_[j].dist;
      }
    }
   } catch (...) {
    failed = true;  // (only ever set to true: no ordering between the workers is needed)
   }
  };
  // `cores` threads over the queries (nn_index.h:342-354); every query is searched on its own, so any split gives the
  // reference's answer
  const uint32_t workers = std::max<uint32_t>(1, std::min<uint32_t>(n, static_cast<uint32_t>(std::max(1, num_threads))));
  if (workers == 1) {
    run(0, n);
  } else {
    std::vector<std::thread> pool;
    for (uint32_t w = 0; w < workers; ++w)
      pool.emplace_back(run, static_cast<uint32_t>(static_cast<uint64_t>(n) * w / workers), static_cast<uint32_t>(static_cast<uint64_t>(n) * (w + 1) / workers));
    for (std::thread& t : pool) t.join();
  }
  return !failed;
}

}  // namespace dagsfm_amd
