// retrieval.hip -- vocabulary-tree image retrieval on MI355X: candidate-pair generation for the matcher.
//
// Replaces, for all images at once, what VocabSimilarityGraph::Run (/root/reference/src/graph/similarity_graph.cpp:
// 101-199) does image by image on CPU threads:
//   VisualIndex::Add / FindWordIds        src/retrieval/visual_index.h:201-243, 695-738
//   InvertedIndex::AddEntry / Finalize    src/retrieval/inverted_index.h:163-172, 220-228
//   InvertedFile::ConvertToBinaryDescriptor / SortEntries / ComputeIDFWeight / ComputeImageSelfSimilarities
//                                          src/retrieval/inverted_file.h:223-271, 374-381
//   InvertedIndex::Query / ComputeSelfSimilarity / InvertedFile::ScoreFeature
//                                          inverted_index.h:237-286, 327-339; inverted_file.h:297-361
//   VisualIndex::QueryAndFindWordIds (sort + truncate)   visual_index.h:664-693
// Checked bit for bit against oracle/retrieval.cc, which also lists the three places where third-party arithmetic
// (FLANN's approximate search, Eigen's float GEMV order, unstable std::sort) is replaced by a defined one.
//
// Kernels:
//   k_vocab_assign_mfma  the nearest words from int8 MFMA tiles with a per-lane top-8 epilogue (below); k_vocab_assign is
//                      the VALU form of the same result (DSM_VOCAB_ASSIGN_VALU=1):
//   k_vocab_assign     thread = two feature rows, visual words stream through LDS in 64-word tiles, 32 v_dot4 per
//                      element, key = 2 d.w - |w|^2 in exact int32 (argmax key == argmin squared L2), sorted top-8
//   k_vocab_signature  wave = feature, lane = embedding dimension: the 128-term float sum left to right, one
//                      __ballot per assigned word gives the 64-bit Hamming signature
//   (rocPRIM radix sorts: entries by word -> inverted files; by image -> per-image self similarity order)
//   k_word_image_counts / k_image_self / k_query_self   IDF inputs and normalisation constants, sequential double sums
//   k_vocab_score      wave = query image: its (feature, neighbour) items in the reference's order, 64 entries of the
//                      word's inverted file per step, per-image runs summed in entry order, burstiness + IDF weight,
//                      accumulated per database image in item order
//   k_score_keys + segmented radix sort   descending score, first-seen order on ties
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <vector>

#include "ctx.h"
#include "flann_search.h"

typedef int v4i __attribute__((ext_vector_type(4)));

#define RK_MAX 8          // neighbours kept per feature
#define RK_INVALID 0x7fffffff

struct RetrievalState {
  uint32_t num_words = 0, words_padded = 0;
  bool have_vocab = false, indexed = false;
  uint32_t k_assigned = 0;  // neighbours currently held in d_wid / d_sig
  // dsm_retrieval_set_word_ids: word ids searched by the caller (the host shim's FLANN-compatible search) instead of the
  // device's exact search.  The reference searches twice -- 1 neighbour when it indexes a feature, num_neighbors when it
  // queries -- and an approximate search's first of five need not be its one of one, so both lists are kept.
  std::vector<int32_t> host_index_ids, host_query_ids;  // [features] / [features][host_k_query], images back to back
  uint32_t host_k_query = 0;
  bool host_ids = false;
  bool host_ids_stale = false;  // the resident images changed after dsm_retrieval_set_word_ids: the ids describe other features
  int host_loaded = 0;  // which list d_wid / d_sig hold: 1 index, 2 query (caller's ids or the device FLANN search)
  FlannDevice* flann = nullptr;  // dsm_retrieval_set_flann_index: the reference's word search on the device (flann_search.hip)
  DevBuf d_words, d_cw, d_projT, d_thr, d_lut;
  DevBuf d_row_img, d_wid, d_sig;                       // per feature row
  DevBuf d_keys, d_keys2, d_vals, d_vals2, d_tmp;        // sort scratch
  DevBuf d_file_start, d_e_img, d_e_sig, d_e_row, d_nimg, d_idf;  // inverted files (d_e_row: the entry's feature row, spatial verification)
  DevBuf d_m_counts, d_m_off, d_m_tuples, d_m_cnt_in, d_m_idx_in;  // dsm_retrieval_matches
  uint64_t m_total = 0;
  std::vector<float> idf_host;
  DevBuf d_img_start, d_normc, d_qnorm, d_nfeat, d_wcounts;  // d_nfeat: feature counts of the images (query kernels); d_wcounts: entries per word (index)
  DevBuf d_acc, d_first, d_skeys, d_skeys2, d_svals, d_svals2, d_seg, d_out_cnt, d_out_idx, d_out_score;
  std::vector<uint32_t> img_valid_start;  // prefix sums of the feature counts
  double index_ms = 0.0, query_ms = 0.0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// ------------------------------------------------------------------------------------ word assignment
// words: s8 (u8 ^ 0x80) [Wp][128]; cw[j] = 2 * rterm(w_j) - |w_j|^2 (+ INT_MIN/2 for padding words).  desc: the context's
// s8 descriptors; row_img[r] < 0 marks a padding row.  out: [rows][RK_MAX] word ids, ascending distance.
#ifdef DSM_CHECK_BUILD  // the VALU form of the word assignment (cross-check of k_vocab_assign_mfma): check build only
__global__ __launch_bounds__(256) void k_vocab_assign(const int8_t* __restrict__ desc, const int32_t* __restrict__ row_img,
                                                      uint64_t n_rows, const int8_t* __restrict__ words,
                                                      const int32_t* __restrict__ cw, uint32_t num_words, uint32_t words_padded,
                                                      int k, int32_t* __restrict__ out) {
  const int tid = threadIdx.x;
  const uint64_t row[2] = {(uint64_t)blockIdx.x * 512u + tid, (uint64_t)blockIdx.x * 512u + 256u + tid};
  __shared__ __attribute__((aligned(16))) int8_t sW[2][64 * 128];
  __shared__ int sC[2][64];
  v4i a[2][8];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint64_t rr = row[r] < n_rows ? row[r] : 0;
    const int8_t* arow = desc + rr * 128;
#pragma unroll
    for (int c = 0; c < 8; ++c) a[r][c] = *reinterpret_cast<const v4i*>(arow + c * 16);
  }
  int key[2][RK_MAX], id[2][RK_MAX];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < RK_MAX; ++q) {
      key[r][q] = INT32_MIN;
      id[r][q] = RK_INVALID;
    }
  const uint32_t nsteps = words_padded >> 6;
  v4i st[2];
  int sc = 0;
  auto fetch = [&](uint32_t s) {
#pragma unroll
    for (int u = 0; u < 2; ++u) st[u] = *reinterpret_cast<const v4i*>(words + (size_t)s * 64 * 128 + (size_t)(tid + 256 * u) * 16);
    sc = cw[s * 64 + (tid & 63)];
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<v4i*>(&sW[buf][(tid + 256 * u) * 16]) = st[u];
    if (tid < 64) sC[buf][tid] = sc;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (uint32_t s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < nsteps;
    if (more) fetch(s + 1);
    for (int j = 0; j < 64; ++j) {
      int acc[2] = {0, 0};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const v4i b = *reinterpret_cast<const v4i*>(&sW[cur][j * 128 + c * 16]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] = __builtin_amdgcn_sdot4(a[0][c][e], b[e], acc[0], false);
          acc[1] = __builtin_amdgcn_sdot4(a[1][c][e], b[e], acc[1], false);
        }
      }
      const int cterm = sC[cur][j];
      const int wid = (int)(s * 64 + j);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        // 2 d.w - |w|^2 up to a per-row constant; a larger key is a smaller distance.  Words are visited in ascending
        // id and every comparison is strict, so equal distances keep the lower id first.
        const int kv = 2 * acc[r] + cterm;
        if (kv > key[r][RK_MAX - 1] && (uint32_t)wid < num_words) {  // the padding of the last tile is not a word
          key[r][RK_MAX - 1] = kv;
          id[r][RK_MAX - 1] = wid;
#pragma unroll
          for (int q = RK_MAX - 1; q > 0; --q) {
            if (key[r][q] > key[r][q - 1]) {
              const int tk = key[r][q], ti = id[r][q];
              key[r][q] = key[r][q - 1];
              id[r][q] = id[r][q - 1];
              key[r][q - 1] = tk;
              id[r][q - 1] = ti;
            }
          }
        }
      }
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (row[r] >= n_rows) continue;
    const bool valid = row_img[row[r]] >= 0;
    int32_t* o = out + row[r] * RK_MAX;
#pragma unroll
    for (int q = 0; q < RK_MAX; ++q) o[q] = (valid && q < k && key[r][q] != INT32_MIN) ? id[r][q] : RK_INVALID;
  }
}
#endif  // DSM_CHECK_BUILD

// ------------------------------------------------------------------------------------ Hamming signatures
// proj = P * float(descriptor), each of the 64 sums over the 128 dimensions left to right (oracle/retrieval.cc Project);
// bit i of the signature for word w = proj[i] > thresholds[w][i] (inverted_file.h:248-257).
__global__ __launch_bounds__(256) void k_vocab_signature(const int8_t* __restrict__ desc, uint64_t n_rows,
                                                         const float* __restrict__ projT /*[128][64]*/,
                                                         const float* __restrict__ thr /*[W][64]*/,
                                                         const int32_t* __restrict__ wid, int k, uint64_t* __restrict__ sig) {
  __shared__ float sP[128 * 64];
  __shared__ volatile float sD[4][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 128 * 64; e += 256) sP[e] = projT[e];
  __syncthreads();
  for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < n_rows; r += (uint64_t)gridDim.x * 4) {
    const int32_t* w = wid + r * RK_MAX;
    if (w[0] == RK_INVALID) continue;  // padding row (or an empty vocabulary)
    // this wave's descriptor as floats: lane l converts bytes 2l, 2l + 1 (u8 = s8 ^ 0x80)
    const uint16_t two = *reinterpret_cast<const uint16_t*>(desc + r * 128 + lane * 2);
    sD[wave][2 * lane] = (float)((two & 0xffu) ^ 0x80u);
    sD[wave][2 * lane + 1] = (float)(((two >> 8) & 0xffu) ^ 0x80u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float s = 0.0f;
    for (int j = 0; j < 128; ++j) s = s + sP[j * 64 + lane] * sD[wave][j];
    for (int n = 0; n < k; ++n) {
      const int word = w[n];
      unsigned long long bits = 0ull;
      if (word != RK_INVALID) bits = __ballot(s > thr[(size_t)word * 64 + lane]);
      if (lane == 0) sig[r * RK_MAX + n] = bits;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------ inverted files
__global__ void k_index_keys(const int32_t* __restrict__ wid, uint64_t n_rows, uint32_t num_words, uint32_t* keys, uint32_t* vals,
                             uint32_t* counts) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int w = wid[r * RK_MAX];
  const uint32_t key = (w == RK_INVALID) ? num_words : (uint32_t)w;  // padding rows sort behind every word
  keys[r] = key;
  vals[r] = (uint32_t)r;
  atomicAdd(counts + key, 1u);
}
__global__ void k_gather_entries(const uint32_t* __restrict__ rows_sorted, uint64_t n_entries, const int32_t* __restrict__ row_img,
                                 const uint64_t* __restrict__ sig, int32_t* e_img, uint64_t* e_sig, uint32_t* e_row) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_entries) return;
  const uint32_t r = rows_sorted[p];
  e_img[p] = row_img[r];
  e_sig[p] = sig[(uint64_t)r * RK_MAX];
  e_row[p] = r;
}
// distinct images per inverted file (its entries are sorted by image): InvertedFile::GetImageIds(...).size()
__global__ void k_word_image_counts(const uint32_t* __restrict__ keys_sorted, const int32_t* __restrict__ e_img, uint64_t n_entries,
                                    uint32_t* nimg) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_entries) return;
  const uint32_t w = keys_sorted[p];
  if (p == 0 || keys_sorted[p - 1] != w || e_img[p - 1] != e_img[p]) atomicAdd(nimg + w, 1u);
}
// InvertedFile::ComputeImageSelfSimilarities over all files (inverted_index.h:420-424): per image, the squared IDF
// weights of its entries added one by one in word order; then 1 / sqrt (inverted_index.h:428-438)
__global__ void k_image_self(const uint32_t* __restrict__ words_by_image, const uint32_t* __restrict__ img_start, uint32_t n_images,
                             const float* __restrict__ idf, float* normc) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_images) return;
  double s = 0.0;
  for (uint32_t p = img_start[d]; p < img_start[d + 1]; ++p) {
    const float w = idf[words_by_image[p]];
    s += (double)(w * w);
  }
  normc[d] = s > 0.0 ? (float)(1.0 / sqrt(s)) : 0.0f;
}
// InvertedIndex::ComputeSelfSimilarity (inverted_index.h:327-339; neighbour-major: linear index of a column-major
// matrix) and the query's normalisation weight (:247-251)
__global__ void k_query_self(const int32_t* __restrict__ wid, const uint32_t* __restrict__ img_row0, const uint32_t* __restrict__ img_nfeat,
                             uint32_t n_images, int k, const float* __restrict__ idf, float* qnorm) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_images) return;
  double s = 0.0;
  for (int n = 0; n < k; ++n)
    for (uint32_t i = 0; i < img_nfeat[q]; ++i) {
      const int w = wid[(uint64_t)(img_row0[q] + i) * RK_MAX + n];
      if (w != RK_INVALID) {
        const float v = idf[w];
        s += (double)(v * v);
      }
    }
  const float self_similarity = (float)s;
  qnorm[q] = self_similarity > 0.0f ? 1.0f / sqrtf(self_similarity) : 1.0f;
}

// ------------------------------------------------------------------------------------ scoring
// image_scores[d].score += score (inverted_index.h:268-275), plus the item of the first contribution.  Agent-scope
// relaxed accesses (served by L2): a later item of the same wave must read what an earlier item's lane wrote.
__device__ __forceinline__ void acc_add(float* acc, uint32_t* first, int d, float sc, uint32_t item) {
  const float v = __hip_atomic_load(acc + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(acc + d, v + sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_load(first + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xffffffffu)
    __hip_atomic_store(first + d, item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One wave per query image.  acc / first: this workgroup's accumulators over the database images.
__global__ __launch_bounds__(64) void k_vocab_score(const int32_t* __restrict__ wid, const uint64_t* __restrict__ sig,
                                                    const uint32_t* __restrict__ img_row0, const uint32_t* __restrict__ img_nfeat,
                                                    uint32_t q0, uint32_t n_queries, uint32_t n_images, int k,
                                                    const uint32_t* __restrict__ file_start, const int32_t* __restrict__ e_img,
                                                    const uint64_t* __restrict__ e_sig, const float* __restrict__ idf,
                                                    const float* __restrict__ lut, float* acc_all, uint32_t* first_all) {
  const int lane = threadIdx.x;
  const uint32_t ql = blockIdx.x;
  if (ql >= n_queries) return;
  const uint32_t q = q0 + ql;
  float* acc = acc_all + (size_t)ql * n_images;
  uint32_t* first = first_all + (size_t)ql * n_images;
  for (uint32_t d = lane; d < n_images; d += 64) {
    __hip_atomic_store(acc + d, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(first + d, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const uint32_t nf = img_nfeat[q];
  uint32_t item = 0;
  for (uint32_t i = 0; i < nf; ++i) {
    const uint64_t r = (uint64_t)img_row0[q] + i;
    for (int n = 0; n < k; ++n, ++item) {
      const int w = wid[r * RK_MAX + n];
      if (w == RK_INVALID) continue;
      const uint32_t s = file_start[w], e = file_start[w + 1];
      if (s == e) continue;
      const uint64_t bq = sig[r * RK_MAX + n];
      const float iw = idf[w];
      const float squared_idf_weight = iw * iw;
      // carry of the run that is open at a chunk boundary (uniform)
      int c_img = -1, c_votes = 0;
      float c_sum = 0.0f;
      for (uint32_t base = s; base < e; base += 64) {
        const uint32_t p = base + lane;
        const bool in = p < e;
        const int img = in ? e_img[p] : -2 - lane;  // distinct sentinels: no run continues into the padding
        float wgt = 0.0f;
        bool vote = false;
        if (in) {
          const int h = __popcll(bq ^ e_sig[p]);
          vote = h <= 24;  // HammingDistWeightFunctor::kMaxHammingDistance = 1.5 * 16 (utils.h:54)
          if (vote) wgt = lut[h];
        }
        const int prev = __shfl_up(img, 1);
        const bool head = in && (lane == 0 ? true : prev != img);
        // every head sums its run in entry order; lane 0 continues the carried run if it is the same image
        float sum = 0.0f;
        int votes = 0;
        if (head && lane == 0 && img == c_img) {
          sum = c_sum;
          votes = c_votes;
        }
        if (head && vote) {
          sum += wgt;
          votes += 1;
        }
        int run = 1;  // entries of this head's run inside the chunk
        for (int t = 1; t < 64; ++t) {
          const int img_t = __shfl_down(img, t);
          const float w_t = __shfl_down(wgt, t);
          const bool v_t = __shfl_down((int)vote, t) != 0;
          const bool cont = head && (lane + t < 64) && img_t == img;
          if (cont) {
            run = t + 1;
            if (v_t) {
              sum += w_t;
              votes += 1;
            }
          }
          if (!__ballot(cont)) break;
        }
        // a run that reaches the end of the chunk may continue in the next one: it is not finalised yet
        const bool open = head && (lane + run == 64) && (base + 64 < e);
        if (lane == 0 && head && img != c_img && c_img >= 0 && c_votes > 0) {
          // the carried run ended exactly at the chunk boundary: finalise it first (burstiness + IDF, :315-327)
          float sc = c_sum / sqrtf((float)c_votes);
          sc *= squared_idf_weight;
          acc_add(acc, first, c_img, sc, item);
        }
        if (head && !open && votes > 0) {
          float sc = sum / sqrtf((float)votes);
          sc *= squared_idf_weight;
          acc_add(acc, first, img, sc, item);
        }
        // new carry: the open run of this chunk (at most one), else nothing
        const unsigned long long ob = __ballot(open);
        if (ob) {
          const int ol = __ffsll((long long)ob) - 1;
          c_img = __shfl(img, ol);
          c_sum = __shfl(sum, ol);
          c_votes = __shfl(votes, ol);
        } else {
          c_img = -1;
          c_sum = 0.0f;
          c_votes = 0;
        }
      }
      __syncthreads();  // the next item may update the same database images: order the read-modify-writes
    }
  }
}

// ------------------------------------------------------------------------------------ matches for spatial verification
// VisualIndex::Query with geometries, visual_index.h:295-346: for every query feature and each of its words, the entries of
// the word's inverted file that belong to one of the query's retrieved images (InvertedIndex::FindMatches) and lie within
// kMaxHammingDistance.  One wave per query image; the retrieved images are a bitmap in LDS.  Tuples (query feature, image,
// database feature, word << 8 | Hamming distance, entry position) in (feature, neighbour, entry) order; WRITE = false only
// counts.  A database feature sits in ONE inverted file (IndexOptions::num_neighbors = 1) and a query feature's words are
// distinct, so the reference's "keep the best weight per database feature" (:331-336) never has two candidates.
template <bool WRITE>
__global__ __launch_bounds__(64) void k_vocab_matches(const int32_t* __restrict__ wid, const uint64_t* __restrict__ sig,
                                                      const uint32_t* __restrict__ img_row0, const uint32_t* __restrict__ img_nfeat,
                                                      uint32_t n_images, int k, const uint32_t* __restrict__ file_start,
                                                      const int32_t* __restrict__ e_img, const uint64_t* __restrict__ e_sig,
                                                      const uint32_t* __restrict__ e_row, const uint32_t* __restrict__ top_counts,
                                                      const uint32_t* __restrict__ top_idx, uint32_t max_num_images,
                                                      uint32_t* m_counts, const uint64_t* __restrict__ m_off, uint32_t* tuples) {
  extern __shared__ uint32_t s_member[];  // (n_images + 31) / 32 words
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  if (q >= n_images) return;
  const uint32_t words = (n_images + 31u) / 32u;
  for (uint32_t w = lane; w < words; w += 64) s_member[w] = 0u;
  __syncthreads();
  const uint32_t nt = top_counts[q] < max_num_images ? top_counts[q] : max_num_images;
  for (uint32_t t = lane; t < nt; t += 64) {
    const uint32_t d = top_idx[(size_t)q * max_num_images + t];
    if (d < n_images) atomicOr(&s_member[d >> 5], 1u << (d & 31u));
  }
  __syncthreads();
  const uint32_t nf = img_nfeat[q];
  uint32_t total = 0;
  uint32_t* out = WRITE ? tuples + (size_t)m_off[q] * 5 : nullptr;
  for (uint32_t i = 0; i < nf; ++i) {
    const uint64_t r = (uint64_t)img_row0[q] + i;
    for (int n = 0; n < k; ++n) {
      const int w = wid[r * RK_MAX + n];
      if (w == RK_INVALID) continue;
      const uint32_t s = file_start[w], e = file_start[w + 1];
      if (s == e) continue;
      const uint64_t bq = sig[r * RK_MAX + n];
      for (uint32_t base = s; base < e; base += 64) {
        const uint32_t p = base + lane;
        bool hit = false;
        int img = 0, h = 0;
        if (p < e) {
          img = e_img[p];
          if ((s_member[(uint32_t)img >> 5] >> ((uint32_t)img & 31u)) & 1u) {
            h = __popcll(bq ^ e_sig[p]);
            hit = h <= 24;  // HammingDistWeightFunctor::kMaxHammingDistance (utils.h:54)
          }
        }
        const unsigned long long bal = __ballot(hit);
        if (WRITE && hit) {
          uint32_t* o = out + (size_t)(total + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))) * 5;
          o[0] = i;
          o[1] = (uint32_t)img;
          o[2] = e_row[p] - img_row0[img];
          o[3] = ((uint32_t)w << 8) | (uint32_t)h;
          o[4] = p;
        }
        total += (uint32_t)__popcll(bal);
      }
    }
  }
  if (!WRITE && lane == 0) m_counts[q] = total;
}

// keys for the descending sort: ~score bits (scores are >= 0), then first-seen item; untouched images go last
__global__ void k_score_keys(const float* __restrict__ acc, const uint32_t* __restrict__ first, uint32_t q0, uint32_t n_queries,
                             uint32_t n_images, const float* __restrict__ qnorm, const float* __restrict__ normc, uint64_t* keys,
                             uint32_t* vals, uint32_t* counts) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)n_queries * n_images) return;
  const uint32_t ql = (uint32_t)(t / n_images), d = (uint32_t)(t % n_images);
  vals[t] = d;
  if (first[t] == 0xffffffffu) {
    keys[t] = 0xffffffffffffffffull;
    return;
  }
  float sc = acc[t];
  sc *= qnorm[q0 + ql] * normc[d];  // inverted_index.h:282-285
  keys[t] = ((uint64_t)(~__float_as_uint(sc)) << 32) | first[t];
  atomicAdd(counts + ql, 1u);
}
__global__ void k_score_output(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ counts,
                               uint32_t q0, uint32_t n_queries, uint32_t n_images, uint32_t max_out, uint32_t* out_cnt,
                               uint32_t* out_idx, float* out_score) {
  const uint32_t ql = blockIdx.x;
  if (ql >= n_queries) return;
  const uint32_t c = counts[ql] < max_out ? counts[ql] : max_out;
  if (threadIdx.x == 0) out_cnt[q0 + ql] = c;
  for (uint32_t kk = threadIdx.x; kk < c; kk += blockDim.x) {
    const uint64_t key = keys[(size_t)ql * n_images + kk];
    out_idx[(size_t)(q0 + ql) * max_out + kk] = vals[(size_t)ql * n_images + kk];
    out_score[(size_t)(q0 + ql) * max_out + kk] = __uint_as_float(~(uint32_t)(key >> 32));
  }
}

// ------------------------------------------------------------------------------------ C-ABI
#define RCHK(ctx, call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
      return DSM_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

void dsm_retrieval_invalidate(dsm_ctx* ctx) {  // the resident images changed
  if (!ctx->retrieval) return;
  ctx->retrieval->indexed = false;
  ctx->retrieval->k_assigned = 0;
  ctx->retrieval->host_loaded = 0;
  // the caller's word ids belong to the features that WERE resident: with a feature cap a different image set has the same
  // total row count, so the size check in retrieval_assign cannot see the change -- the ids are refused until
  // dsm_retrieval_set_word_ids is called again
  if (ctx->retrieval->host_ids) ctx->retrieval->host_ids_stale = true;
}

void dsm_retrieval_destroy(dsm_ctx* ctx) {
  RetrievalState* r = ctx->retrieval;
  if (!r) return;
  DevBuf* bufs[] = {&r->d_words, &r->d_cw, &r->d_projT, &r->d_thr, &r->d_lut, &r->d_row_img, &r->d_wid, &r->d_sig, &r->d_keys,
                    &r->d_keys2, &r->d_vals, &r->d_vals2, &r->d_tmp, &r->d_file_start, &r->d_e_img, &r->d_e_sig, &r->d_e_row, &r->d_m_counts, &r->d_m_off, &r->d_m_tuples, &r->d_m_cnt_in, &r->d_m_idx_in, &r->d_nimg,
                    &r->d_idf, &r->d_img_start, &r->d_normc, &r->d_qnorm, &r->d_nfeat, &r->d_wcounts, &r->d_acc, &r->d_first, &r->d_skeys, &r->d_skeys2,
                    &r->d_svals, &r->d_svals2, &r->d_seg, &r->d_out_cnt, &r->d_out_idx, &r->d_out_score};
  for (DevBuf* b : bufs) b->release();
  flann_device_destroy(r->flann);
  r->flann = nullptr;
  if (r->ev0) (void)hipEventDestroy(r->ev0);
  if (r->ev1) (void)hipEventDestroy(r->ev1);
  delete r;
  ctx->retrieval = nullptr;
}

// ------------------------------------------------------------------------------------ word assignment on the matrix pipe
// The same result as k_vocab_assign from int8 MFMA tiles (v_mfma_i32_32x32x32_i8): a workgroup is 4 waves x 128 feature
// rows (4 resident 32-row fragments per wave, like k1_best_rows), the visual words stream through LDS in 64-word steps.
// A lane holds, for ITS row, 16 of a tile's 32 columns (the other half-wave holds the rest), so the row's top-RK_MAX list
// is kept per lane over the lane's own columns and the two halves are merged at the end.  Epilogue per tile: key =
// 2 * dot + cw (one v_lshl_add per element) and a running maximum; only when that maximum beats the lane's current
// RK_MAX-th key are the 16 elements looked at one by one, in ascending word id (strict >: equal keys keep the lower id
// first, as in the scalar scan).  ~1.5 VALU per element against 32 v_dot4 + the scan in the VALU kernel.
typedef int v16i_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int rk_lds_off(int col, int chunk) { return col * 128 + ((chunk ^ ((col >> 1) & 7)) << 4); }

__device__ __forceinline__ void rk_insert(int (&key)[RK_MAX], int (&id)[RK_MAX], int kv, int wid) {
  key[RK_MAX - 1] = kv;
  id[RK_MAX - 1] = wid;
#pragma unroll
  for (int q = RK_MAX - 1; q > 0; --q) {
    const bool up = key[q] > key[q - 1] || (key[q] == key[q - 1] && id[q] < id[q - 1]);
    const int tk = key[q], ti = id[q];
    key[q] = up ? key[q - 1] : tk;
    id[q] = up ? id[q - 1] : ti;
    key[q - 1] = up ? tk : key[q - 1];
    id[q - 1] = up ? ti : id[q - 1];
  }
}

__global__ __launch_bounds__(256, 2) void k_vocab_assign_mfma(const int8_t* __restrict__ desc, const int32_t* __restrict__ row_img,
                                                              uint64_t n_rows, const int8_t* __restrict__ words,
                                                              const int32_t* __restrict__ cw, uint32_t num_words, uint32_t words_padded,
                                                              int k, int32_t* __restrict__ out) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  __shared__ __attribute__((aligned(16))) int8_t sB[2][64 * 128];
  __shared__ __attribute__((aligned(16))) int sC[2][64];
  // resident fragments of this wave's 128 rows (MFMA B operand: lane = row, 16 B of k per half)
  v4i afrag[4][4];
  uint64_t rowg[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    rowg[rt] = (uint64_t)blockIdx.x * 512u + (uint64_t)wave * 128u + (uint64_t)rt * 32u + (uint64_t)l31;
    const uint64_t rr = rowg[rt] < n_rows ? rowg[rt] : 0;  // rows are padded per image; beyond the end: any valid row
    const int8_t* arow = desc + rr * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) afrag[rt][ks] = *reinterpret_cast<const v4i*>(arow + ks * 32 + half * 16);
  }
  int key[4][RK_MAX], id[4][RK_MAX];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int q = 0; q < RK_MAX; ++q) {
      key[rt][q] = INT32_MIN;
      id[rt][q] = RK_INVALID;
    }
  }
  const uint32_t nsteps = words_padded >> 6;
  v4i stage[2];
  int cstage = 0;
  auto fetch = [&](uint32_t s) {
#pragma unroll
    for (int u = 0; u < 2; ++u) stage[u] = *reinterpret_cast<const v4i*>(words + (size_t)s * 64 * 128 + (size_t)(tid + 256 * u) * 16);
    cstage = cw[s * 64 + (tid & 63)];
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + 256 * u;
      *reinterpret_cast<v4i*>(&sB[buf][rk_lds_off(q >> 3, q & 7)]) = stage[u];
    }
    if (tid < 64) sC[buf][tid] = cstage;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (uint32_t s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < nsteps;
    if (more) fetch(s + 1);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      // MFMA A operand: lane = column (word) of the tile
      v4i bf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const v4i*>(&sB[cur][rk_lds_off(ct * 32 + l31, ks * 2 + half)]);
      // column terms of this lane's 16 accumulator registers: register r <-> column 8*(r>>2) + 4*half + (r&3)
      int cterm[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4i c = *reinterpret_cast<const v4i*>(&sC[cur][ct * 32 + 8 * q + 4 * half]);
#pragma unroll
        for (int e = 0; e < 4; ++e) cterm[4 * q + e] = c[e];
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        v16i_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks], afrag[rt][ks], acc, 0, 0, 0);
        int kv[16];
        int mx = INT32_MIN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          kv[r] = (acc[r] << 1) + cterm[r];
          mx = kv[r] > mx ? kv[r] : mx;
        }
        // rare path, ONE insertion site per tile: take the lane's largest remaining element (lowest register = lowest
        // word id on equal keys), insert it, repeat while anything is left above the threshold.  The final list does
        // not depend on the order of insertion (rk_insert orders by key, then id).
        const int wid0 = (int)(s * 64) + ct * 32 + 4 * half;
        while (mx > key[rt][RK_MAX - 1]) {
          int best = kv[0], br = 0;
#pragma unroll
          for (int r = 1; r < 16; ++r) {
            const bool g = kv[r] > best;
            best = g ? kv[r] : best;
            br = g ? r : br;
          }
          const int wid = wid0 + 8 * (br >> 2) + (br & 3);
          if ((uint32_t)wid < num_words) rk_insert(key[rt], id[rt], best, wid);  // the last tile's padding is not a word
          mx = INT32_MIN;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            kv[r] = (r == br) ? INT32_MIN : kv[r];
            mx = kv[r] > mx ? kv[r] : mx;
          }
        }
      }
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
  }
  // the two halves of the wave hold disjoint word subsets of the same rows: merge (key descending, lower id first)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    int ok[RK_MAX], oi[RK_MAX];
#pragma unroll
    for (int q = 0; q < RK_MAX; ++q) {
      ok[q] = __shfl_xor(key[rt][q], 32);
      oi[q] = __shfl_xor(id[rt][q], 32);
    }
#pragma unroll
    for (int q = 0; q < RK_MAX; ++q) {
      const int lk = key[rt][RK_MAX - 1], li = id[rt][RK_MAX - 1];
      if (ok[q] > lk || (ok[q] == lk && oi[q] < li)) rk_insert(key[rt], id[rt], ok[q], oi[q]);
    }
    if (half == 0 && rowg[rt] < n_rows) {
      const bool valid = row_img[rowg[rt]] >= 0;
      int32_t* o = out + rowg[rt] * RK_MAX;
#pragma unroll
      for (int q = 0; q < RK_MAX; ++q) o[q] = (valid && q < k && key[rt][q] != INT32_MIN) ? id[rt][q] : RK_INVALID;
    }
  }
}

// words of every feature of every resident image (k nearest), signatures for them
enum { ASSIGN_FOR_INDEX = 1, ASSIGN_FOR_QUERY = 2 };
static int retrieval_assign(dsm_ctx* ctx, uint32_t k, int purpose) {
  RetrievalState* r = ctx->retrieval;
  hipStream_t st = ctx->stream;
  const uint64_t rows = ctx->total_rows;
  if (r->host_ids) {
    // the caller's word ids: scattered into the padded row layout, signatures computed for exactly those words
    if (r->host_ids_stale)
      return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_word_ids: the resident images changed since the ids were set");
    if (r->host_loaded == purpose && r->d_wid.p) return DSM_OK;
    uint64_t n_feat = 0;
    for (uint32_t i = 0; i < ctx->n_images; ++i) n_feat += ctx->nfeat[i];
    if (n_feat != r->host_index_ids.size()) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_word_ids: the resident images changed since the ids were set");
    const uint32_t kk = purpose == ASSIGN_FOR_INDEX ? 1u : r->host_k_query;
    std::vector<int32_t> wid(std::max<uint64_t>(rows, 1) * RK_MAX, RK_INVALID);
    r->img_valid_start.assign(ctx->n_images + 1, 0);
    uint64_t at = 0;
    for (uint32_t i = 0; i < ctx->n_images; ++i) {
      for (uint32_t f = 0; f < ctx->nfeat[i]; ++f, ++at) {
        int32_t* o = wid.data() + ((uint64_t)ctx->row0[i] + f) * RK_MAX;
        if (purpose == ASSIGN_FOR_INDEX)
          o[0] = r->host_index_ids[at];
        else
          for (uint32_t q = 0; q < kk; ++q) o[q] = r->host_query_ids[at * kk + q];
      }
      r->img_valid_start[i + 1] = r->img_valid_start[i] + ctx->nfeat[i];
    }
    std::vector<int32_t> row_img(std::max<uint64_t>(rows, 1), -1);  // row -> image (padding rows: -1), as below
    for (uint32_t i = 0; i < ctx->n_images; ++i)
      for (uint32_t f = 0; f < ctx->nfeat[i]; ++f) row_img[(uint64_t)ctx->row0[i] + f] = (int32_t)i;
    RCHK(ctx, r->d_row_img.reserve(row_img.size() * 4));
    RCHK(ctx, hipMemcpy(r->d_row_img.p, row_img.data(), row_img.size() * 4, hipMemcpyHostToDevice));
    RCHK(ctx, r->d_wid.reserve(wid.size() * 4));
    RCHK(ctx, r->d_sig.reserve(wid.size() * 8));
    RCHK(ctx, hipMemcpy(r->d_wid.p, wid.data(), wid.size() * 4, hipMemcpyHostToDevice));
    if (rows) {
      hipLaunchKernelGGL(k_vocab_signature, dim3(2048), dim3(256), 0, st, ctx->d_desc.as<int8_t>(), rows, r->d_projT.as<float>(),
                         r->d_thr.as<float>(), r->d_wid.as<int32_t>(), (int)kk, r->d_sig.as<uint64_t>());
      RCHK(ctx, hipGetLastError());
    }
    r->k_assigned = kk;
    r->host_loaded = purpose;
    return DSM_OK;
  }
  if (r->flann) {
    // the reference's approximate search over the file's own FLANN index, a lane per feature (flann_search.hip): 1 neighbour
    // when a feature is indexed, num_neighbors when it is queried -- two searches, as in VisualIndex::Add / ::Query
    const uint32_t kk = purpose == ASSIGN_FOR_INDEX ? 1u : k;
    if (r->host_loaded == purpose && r->k_assigned == kk && r->d_wid.p) return DSM_OK;
    std::vector<int32_t> row_img(std::max<uint64_t>(rows, 1), -1);
    r->img_valid_start.assign(ctx->n_images + 1, 0);
    for (uint32_t i = 0; i < ctx->n_images; ++i) {
      for (uint32_t f = 0; f < ctx->nfeat[i]; ++f) row_img[(uint64_t)ctx->row0[i] + f] = (int32_t)i;
      r->img_valid_start[i + 1] = r->img_valid_start[i] + ctx->nfeat[i];
    }
    RCHK(ctx, r->d_row_img.reserve(row_img.size() * 4));
    RCHK(ctx, hipMemcpy(r->d_row_img.p, row_img.data(), row_img.size() * 4, hipMemcpyHostToDevice));
    RCHK(ctx, r->d_wid.reserve(std::max<uint64_t>(rows, 1) * RK_MAX * 4));
    RCHK(ctx, r->d_sig.reserve(std::max<uint64_t>(rows, 1) * RK_MAX * 8));
    if (rows) {
      const int rc = flann_device_search(ctx, r->flann, r->d_words.as<int8_t>(), ctx->d_desc.as<int8_t>(), r->d_row_img.as<int32_t>(), rows, kk,
                                         r->d_wid.as<int32_t>(), nullptr, RK_MAX, st);
      if (rc != DSM_OK) return rc;
      hipLaunchKernelGGL(k_vocab_signature, dim3(2048), dim3(256), 0, st, ctx->d_desc.as<int8_t>(), rows, r->d_projT.as<float>(),
                         r->d_thr.as<float>(), r->d_wid.as<int32_t>(), (int)kk, r->d_sig.as<uint64_t>());
      RCHK(ctx, hipGetLastError());
    }
    r->k_assigned = kk;
    r->host_loaded = purpose;
    return DSM_OK;
  }
  if (r->k_assigned >= k && r->d_wid.p) return DSM_OK;
  // row -> image (padding rows: -1)
  std::vector<int32_t> row_img(std::max<uint64_t>(rows, 1), -1);
  r->img_valid_start.assign(ctx->n_images + 1, 0);
  for (uint32_t i = 0; i < ctx->n_images; ++i) {
    for (uint32_t f = 0; f < ctx->nfeat[i]; ++f) row_img[(uint64_t)ctx->row0[i] + f] = (int32_t)i;
    r->img_valid_start[i + 1] = r->img_valid_start[i] + ctx->nfeat[i];
  }
  RCHK(ctx, r->d_row_img.reserve(row_img.size() * 4));
  RCHK(ctx, hipMemcpy(r->d_row_img.p, row_img.data(), row_img.size() * 4, hipMemcpyHostToDevice));
  RCHK(ctx, r->d_wid.reserve(std::max<uint64_t>(rows, 1) * RK_MAX * 4));
  RCHK(ctx, r->d_sig.reserve(std::max<uint64_t>(rows, 1) * RK_MAX * 8));
  if (rows) {
#ifdef DSM_CHECK_BUILD
    if (ctx->dbg("DSM_VOCAB_ASSIGN_VALU"))  // the LDS-tiled v_dot4 form (comparison / cross-check)
      hipLaunchKernelGGL(k_vocab_assign, dim3((uint32_t)((rows + 511) / 512)), dim3(256), 0, st, ctx->d_desc.as<int8_t>(),
                         r->d_row_img.as<int32_t>(), rows, r->d_words.as<int8_t>(), r->d_cw.as<int32_t>(), r->num_words, r->words_padded, (int)k,
                         r->d_wid.as<int32_t>());
    else
#endif
      hipLaunchKernelGGL(k_vocab_assign_mfma, dim3((uint32_t)((rows + 511) / 512)), dim3(256), 0, st, ctx->d_desc.as<int8_t>(),
                         r->d_row_img.as<int32_t>(), rows, r->d_words.as<int8_t>(), r->d_cw.as<int32_t>(), r->num_words, r->words_padded, (int)k,
                         r->d_wid.as<int32_t>());
    RCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_vocab_signature, dim3(2048), dim3(256), 0, st, ctx->d_desc.as<int8_t>(), rows, r->d_projT.as<float>(),
                       r->d_thr.as<float>(), r->d_wid.as<int32_t>(), (int)k, r->d_sig.as<uint64_t>());
    RCHK(ctx, hipGetLastError());
  }
  r->k_assigned = k;
  return DSM_OK;
}

extern "C" {

int dsm_retrieval_set_vocabulary(dsm_ctx* ctx, const dsm_vocabulary* v) {
  if (!ctx || !v) return DSM_ERR_INVALID_ARGUMENT;
  if (v->num_words == 0 || !v->words || !v->projection || !v->thresholds)
    return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "vocabulary needs words, projection and thresholds");
  if (v->num_words >= 0x7fffff00u) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "too many visual words");
  RCHK(ctx, hipSetDevice(ctx->device));
  if (!ctx->retrieval) ctx->retrieval = new RetrievalState();
  RetrievalState* r = ctx->retrieval;
  const uint32_t W = v->num_words, Wp = (W + 63u) / 64u * 64u;
  std::vector<int8_t> w8((size_t)Wp * 128, 0);
  std::vector<int32_t> cw(Wp, INT32_MIN / 2);
  for (uint32_t j = 0; j < W; ++j) {
    int64_t sum_s8 = 0, nw = 0;
    for (int c = 0; c < 128; ++c) {
      const int u = v->words[(size_t)j * 128 + c];
      w8[(size_t)j * 128 + c] = (int8_t)(u ^ 0x80);
      sum_s8 += u - 128;
      nw += u * u;
    }
    // 2 d.w = 2 S + 2 rterm(d) + 2 rterm(w) + 2^22 with S = sum d'w' and rterm(x) = 128 sum x' (the signed-int8 identity
    // of match_kernels.hip); per row only 2 S + (2 rterm(w) - |w|^2) matters
    cw[j] = (int32_t)(2 * 128 * sum_s8 - nw);
  }
  std::vector<float> projT((size_t)128 * 64);
  for (int i = 0; i < 64; ++i)
    for (int c = 0; c < 128; ++c) projT[(size_t)c * 64 + i] = v->projection[(size_t)i * 128 + c];
  float lut[65];
  const float sigma_squared = 16 * 16;  // HammingDistWeightFunctor<64, 16>, utils.h:56-69, with the host libm's expf
  for (int n = 0; n <= 64; ++n) {
    const float hd = (float)n;
    lut[n] = hd <= 24.0f ? expf(-hd * hd / sigma_squared) : 0.0f;
  }
  RCHK(ctx, r->d_words.reserve(w8.size()));
  RCHK(ctx, r->d_cw.reserve(cw.size() * 4));
  RCHK(ctx, r->d_projT.reserve(projT.size() * 4));
  RCHK(ctx, r->d_thr.reserve((size_t)W * 64 * 4));
  RCHK(ctx, r->d_lut.reserve(sizeof(lut)));
  RCHK(ctx, hipMemcpy(r->d_words.p, w8.data(), w8.size(), hipMemcpyHostToDevice));
  RCHK(ctx, hipMemcpy(r->d_cw.p, cw.data(), cw.size() * 4, hipMemcpyHostToDevice));
  RCHK(ctx, hipMemcpy(r->d_projT.p, projT.data(), projT.size() * 4, hipMemcpyHostToDevice));
  RCHK(ctx, hipMemcpy(r->d_thr.p, v->thresholds, (size_t)W * 64 * 4, hipMemcpyHostToDevice));
  RCHK(ctx, hipMemcpy(r->d_lut.p, lut, sizeof(lut), hipMemcpyHostToDevice));
  r->num_words = W;
  r->words_padded = Wp;
  r->have_vocab = true;
  flann_device_destroy(r->flann);  // an index belongs to the vocabulary it was built over
  r->flann = nullptr;
  r->host_loaded = 0;
  r->indexed = false;
  r->k_assigned = 0;
  if (!r->ev0) {
    RCHK(ctx, hipEventCreate(&r->ev0));
    RCHK(ctx, hipEventCreate(&r->ev1));
  }
  return DSM_OK;
}

int dsm_retrieval_set_word_ids(dsm_ctx* ctx, const int32_t* index_ids, uint32_t k_query, const int32_t* query_ids) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r || !r->have_vocab) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_vocabulary has not run");
  r->indexed = false;
  r->k_assigned = 0;
  r->host_loaded = 0;
  r->host_ids_stale = false;
  if (!index_ids && !query_ids) {  // back to the device's exact search
    r->host_ids = false;
    r->host_index_ids.clear();
    r->host_query_ids.clear();
    return DSM_OK;
  }
  if (!index_ids || !query_ids || k_query == 0 || k_query > RK_MAX) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "word id lists / k_query (1..8)");
  uint64_t n_feat = 0;
  for (uint32_t i = 0; i < ctx->n_images; ++i) n_feat += ctx->nfeat[i];
  for (uint64_t i = 0; i < n_feat; ++i)  // (a 1-neighbour search over >= 1 words always returns a word: every feature is indexed)
    if (index_ids[i] < 0 || (uint32_t)index_ids[i] >= r->num_words) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "word id outside the vocabulary");
  for (uint64_t i = 0; i < n_feat * k_query; ++i)
    if (query_ids[i] != RK_INVALID && (query_ids[i] < 0 || (uint32_t)query_ids[i] >= r->num_words))
      return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "word id outside the vocabulary");
  r->host_index_ids.assign(index_ids, index_ids + n_feat);
  r->host_query_ids.assign(query_ids, query_ids + n_feat * k_query);
  r->host_k_query = k_query;
  r->host_ids = true;
  return DSM_OK;
}

int dsm_retrieval_set_flann_index(dsm_ctx* ctx, const dsm_flann_index* index) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r || !r->have_vocab) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_vocabulary has not run");
  r->indexed = false;
  r->k_assigned = 0;
  r->host_loaded = 0;
  RCHK(ctx, hipSetDevice(ctx->device));
  const int rc = flann_device_set_index(ctx, &r->flann, index, r->d_words.as<int8_t>(), r->num_words);
  if (rc != DSM_OK) {  // a refused or half-uploaded index must not stay behind: the exact search again
    flann_device_destroy(r->flann);
    r->flann = nullptr;
  }
  return rc;
}

int dsm_retrieval_flann_search(dsm_ctx* ctx, const uint8_t* descriptors, uint32_t n, uint32_t k, int32_t* ids, float* dists, double* ms) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r || !r->have_vocab || !r->flann) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_flann_index has not run");
  const int rc = flann_device_search_host(ctx, r->flann, r->d_words.as<int8_t>(), descriptors, n, k, ids, dists);
  if (ms) *ms = flann_device_last_ms(r->flann);
  return rc;
}

int dsm_retrieval_index(dsm_ctx* ctx) {
  if (!ctx) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r || !r->have_vocab) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_vocabulary has not run");
  RCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t rows = ctx->total_rows;
  const uint32_t W = r->num_words, NI = ctx->n_images;
  RCHK(ctx, hipEventRecord(r->ev0, st));
  r->k_assigned = 0;  // the resident images may have changed
  r->host_loaded = 0;
  int rc = retrieval_assign(ctx, RK_MAX, ASSIGN_FOR_INDEX);
  if (rc != DSM_OK) return rc;
  const uint64_t n_entries = r->img_valid_start[NI];  // one entry per feature (IndexOptions::num_neighbors = 1)
  const uint64_t rows1 = std::max<uint64_t>(rows, 1);
  RCHK(ctx, r->d_keys.reserve(rows1 * 4));
  RCHK(ctx, r->d_keys2.reserve(rows1 * 4));
  RCHK(ctx, r->d_vals.reserve(rows1 * 4));
  RCHK(ctx, r->d_vals2.reserve(rows1 * 4));
  RCHK(ctx, r->d_file_start.reserve(((size_t)W + 2) * 4));
  RCHK(ctx, r->d_nimg.reserve(((size_t)W + 1) * 4));
  RCHK(ctx, r->d_idf.reserve(((size_t)W + 1) * 4));
  RCHK(ctx, r->d_e_img.reserve(rows1 * 4));
  RCHK(ctx, r->d_e_sig.reserve(rows1 * 8));
  RCHK(ctx, r->d_e_row.reserve(rows1 * 4));
  RCHK(ctx, r->d_img_start.reserve(((size_t)NI + 1) * 4));
  RCHK(ctx, r->d_normc.reserve(std::max<uint32_t>(NI, 1) * 4));
  RCHK(ctx, hipMemsetAsync(r->d_nimg.p, 0, ((size_t)W + 1) * 4, st));
  // counts per word -> file_start (exclusive scan); entries sorted by word, stable = (image, feature) order inside a
  // file: InvertedFile::SortEntries sorts by image id (inverted_file.h:223-230)
  std::vector<uint32_t> counts((size_t)W + 1, 0), starts((size_t)W + 2, 0);
  DevBuf& d_counts = r->d_wcounts;  // owned by the state: no leak on the early returns below
  RCHK(ctx, d_counts.reserve(((size_t)W + 1) * 4));
  RCHK(ctx, hipMemsetAsync(d_counts.p, 0, ((size_t)W + 1) * 4, st));
  if (rows) {
    hipLaunchKernelGGL(k_index_keys, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, st, r->d_wid.as<int32_t>(), rows, W,
                       r->d_keys.as<uint32_t>(), r->d_vals.as<uint32_t>(), d_counts.as<uint32_t>());
    RCHK(ctx, hipGetLastError());
    int bits = 1;
    while ((1u << bits) <= W) ++bits;
    size_t tmp_bytes = 0;
    RCHK(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, r->d_keys.as<uint32_t>(), r->d_keys2.as<uint32_t>(), r->d_vals.as<uint32_t>(),
                                        r->d_vals2.as<uint32_t>(), (size_t)rows, 0, (unsigned)bits, st));
    RCHK(ctx, r->d_tmp.reserve(std::max<size_t>(tmp_bytes, 16)));
    RCHK(ctx, rocprim::radix_sort_pairs(r->d_tmp.p, tmp_bytes, r->d_keys.as<uint32_t>(), r->d_keys2.as<uint32_t>(), r->d_vals.as<uint32_t>(),
                                        r->d_vals2.as<uint32_t>(), (size_t)rows, 0, (unsigned)bits, st));
  }
  RCHK(ctx, hipMemcpyAsync(counts.data(), d_counts.p, ((size_t)W + 1) * 4, hipMemcpyDeviceToHost, st));
  RCHK(ctx, hipStreamSynchronize(st));
  for (uint32_t w = 0; w <= W; ++w) starts[w + 1] = starts[w] + counts[w];
  RCHK(ctx, hipMemcpyAsync(r->d_file_start.p, starts.data(), ((size_t)W + 2) * 4, hipMemcpyHostToDevice, st));
  if (n_entries) {
    hipLaunchKernelGGL(k_gather_entries, dim3((uint32_t)((n_entries + 255) / 256)), dim3(256), 0, st, r->d_vals2.as<uint32_t>(), n_entries,
                       r->d_row_img.as<int32_t>(), r->d_sig.as<uint64_t>(), r->d_e_img.as<int32_t>(), r->d_e_sig.as<uint64_t>(), r->d_e_row.as<uint32_t>());
    hipLaunchKernelGGL(k_word_image_counts, dim3((uint32_t)((n_entries + 255) / 256)), dim3(256), 0, st, r->d_keys2.as<uint32_t>(),
                       r->d_e_img.as<int32_t>(), n_entries, r->d_nimg.as<uint32_t>());
    RCHK(ctx, hipGetLastError());
  }
  // IDF weights with the host libm (InvertedFile::ComputeIDFWeight, inverted_file.h:260-271)
  std::vector<uint32_t> nimg((size_t)W + 1, 0);
  RCHK(ctx, hipMemcpyAsync(nimg.data(), r->d_nimg.p, ((size_t)W + 1) * 4, hipMemcpyDeviceToHost, st));
  RCHK(ctx, hipStreamSynchronize(st));
  uint32_t num_total_images = 0;
  for (uint32_t i = 0; i < NI; ++i) num_total_images += ctx->nfeat[i] ? 1u : 0u;
  std::vector<float> idf((size_t)W + 1, 0.0f);
  for (uint32_t w = 0; w < W; ++w)
    if (nimg[w]) idf[w] = (float)log((double)num_total_images / (double)nimg[w]);
  RCHK(ctx, hipMemcpyAsync(r->d_idf.p, idf.data(), ((size_t)W + 1) * 4, hipMemcpyHostToDevice, st));
  r->idf_host.assign(idf.begin(), idf.begin() + W);
  RCHK(ctx, hipMemcpyAsync(r->d_img_start.p, r->img_valid_start.data(), ((size_t)NI + 1) * 4, hipMemcpyHostToDevice, st));
  // per image: its entries in word order = the word-sorted list stably re-sorted by image
  if (n_entries) {
    int bits = 1;
    while ((1u << bits) < std::max<uint32_t>(NI, 2)) ++bits;
    // keys: image of every sorted entry (as unsigned), values: its word
    RCHK(ctx, hipMemcpyAsync(r->d_keys.p, r->d_e_img.p, n_entries * 4, hipMemcpyDeviceToDevice, st));
    size_t tmp_bytes = 0;
    RCHK(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, r->d_keys.as<uint32_t>(), r->d_vals.as<uint32_t>(), r->d_keys2.as<uint32_t>(),
                                        r->d_vals2.as<uint32_t>(), (size_t)n_entries, 0, (unsigned)bits, st));
    RCHK(ctx, r->d_tmp.reserve(std::max<size_t>(tmp_bytes, 16)));
    // NOTE: d_vals2 (sorted rows) is overwritten here with the words by image; nothing needs the rows any more
    RCHK(ctx, rocprim::radix_sort_pairs(r->d_tmp.p, tmp_bytes, r->d_keys.as<uint32_t>(), r->d_vals.as<uint32_t>(), r->d_keys2.as<uint32_t>(),
                                        r->d_vals2.as<uint32_t>(), (size_t)n_entries, 0, (unsigned)bits, st));
  }
  if (NI) {
    hipLaunchKernelGGL(k_image_self, dim3((NI + 63) / 64), dim3(64), 0, st, r->d_vals2.as<uint32_t>(), r->d_img_start.as<uint32_t>(), NI,
                       r->d_idf.as<float>(), r->d_normc.as<float>());
    RCHK(ctx, hipGetLastError());
  }
  RCHK(ctx, hipEventRecord(r->ev1, st));
  RCHK(ctx, hipStreamSynchronize(st));
  float ms = 0.f;
  RCHK(ctx, hipEventElapsedTime(&ms, r->ev0, r->ev1));
  r->index_ms = ms;
  r->indexed = true;
  return DSM_OK;
}

int dsm_retrieval_query(dsm_ctx* ctx, uint32_t num_neighbors, uint32_t max_num_images, uint32_t* counts, uint32_t* image_idx,
                        float* scores) {
  if (!ctx || !counts || !image_idx || !scores) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r || !r->indexed) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_index has not run");
  if (num_neighbors == 0 || num_neighbors > RK_MAX) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "num_neighbors must be 1..8");
  if (max_num_images == 0) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "max_num_images must be > 0");
  RCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint32_t NI = ctx->n_images;
  if (NI == 0) return DSM_OK;
  const int k = (int)num_neighbors;
  if (r->host_ids || r->flann) {  // the query's own word search (num_neighbors words), not the one the index was built with
    if (r->host_ids && num_neighbors != r->host_k_query) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "num_neighbors differs from the k of dsm_retrieval_set_word_ids");
    const int rca = retrieval_assign(ctx, num_neighbors, ASSIGN_FOR_QUERY);
    if (rca != DSM_OK) return rca;
  }
  RCHK(ctx, hipEventRecord(r->ev0, st));
  DevBuf& d_nfeat = r->d_nfeat;  // owned by the state: released with it on every exit path
  RCHK(ctx, d_nfeat.reserve((size_t)NI * 4));
  RCHK(ctx, hipMemcpyAsync(d_nfeat.p, ctx->nfeat.data(), (size_t)NI * 4, hipMemcpyHostToDevice, st));
  RCHK(ctx, r->d_qnorm.reserve((size_t)NI * 4));
  hipLaunchKernelGGL(k_query_self, dim3((NI + 63) / 64), dim3(64), 0, st, r->d_wid.as<int32_t>(), ctx->d_img_row0.as<uint32_t>(),
                     d_nfeat.as<uint32_t>(), NI, k, r->d_idf.as<float>(), r->d_qnorm.as<float>());
  RCHK(ctx, hipGetLastError());
  // queries in batches: accumulators [batch][NI]
  const uint64_t budget = 1ull << 30;
  const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(NI, budget / ((uint64_t)NI * 24)));
  RCHK(ctx, r->d_acc.reserve((size_t)batch * NI * 4));
  RCHK(ctx, r->d_first.reserve((size_t)batch * NI * 4));
  RCHK(ctx, r->d_skeys.reserve((size_t)batch * NI * 8));
  RCHK(ctx, r->d_skeys2.reserve((size_t)batch * NI * 8));
  RCHK(ctx, r->d_svals.reserve((size_t)batch * NI * 4));
  RCHK(ctx, r->d_svals2.reserve((size_t)batch * NI * 4));
  RCHK(ctx, r->d_seg.reserve(((size_t)batch + 1) * 4));
  RCHK(ctx, r->d_out_cnt.reserve((size_t)NI * 4 + (size_t)batch * 4));
  RCHK(ctx, r->d_out_idx.reserve((size_t)NI * max_num_images * 4));
  RCHK(ctx, r->d_out_score.reserve((size_t)NI * max_num_images * 4));
  std::vector<uint32_t> seg((size_t)batch + 1);
  for (uint32_t b = 0; b <= batch; ++b) seg[b] = b * NI;
  RCHK(ctx, hipMemcpyAsync(r->d_seg.p, seg.data(), seg.size() * 4, hipMemcpyHostToDevice, st));
  uint32_t* d_bcounts = r->d_out_cnt.as<uint32_t>() + NI;
  for (uint32_t q0 = 0; q0 < NI; q0 += batch) {
    const uint32_t nq = std::min<uint32_t>(batch, NI - q0);
    hipLaunchKernelGGL(k_vocab_score, dim3(nq), dim3(64), 0, st, r->d_wid.as<int32_t>(), r->d_sig.as<uint64_t>(),
                       ctx->d_img_row0.as<uint32_t>(), d_nfeat.as<uint32_t>(), q0, nq, NI, k, r->d_file_start.as<uint32_t>(),
                       r->d_e_img.as<int32_t>(), r->d_e_sig.as<uint64_t>(), r->d_idf.as<float>(), r->d_lut.as<float>(),
                       r->d_acc.as<float>(), r->d_first.as<uint32_t>());
    RCHK(ctx, hipGetLastError());
    RCHK(ctx, hipMemsetAsync(d_bcounts, 0, (size_t)nq * 4, st));
    const uint64_t total = (uint64_t)nq * NI;
    hipLaunchKernelGGL(k_score_keys, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, r->d_acc.as<float>(), r->d_first.as<uint32_t>(),
                       q0, nq, NI, r->d_qnorm.as<float>(), r->d_normc.as<float>(), r->d_skeys.as<uint64_t>(), r->d_svals.as<uint32_t>(),
                       d_bcounts);
    RCHK(ctx, hipGetLastError());
    size_t tmp_bytes = 0;
    RCHK(ctx, rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, r->d_skeys.as<uint64_t>(), r->d_skeys2.as<uint64_t>(),
                                                  r->d_svals.as<uint32_t>(), r->d_svals2.as<uint32_t>(), (unsigned)total, nq,
                                                  r->d_seg.as<uint32_t>(), r->d_seg.as<uint32_t>() + 1, 0, 64, st));
    RCHK(ctx, r->d_tmp.reserve(std::max<size_t>(tmp_bytes, 16)));
    RCHK(ctx, rocprim::segmented_radix_sort_pairs(r->d_tmp.p, tmp_bytes, r->d_skeys.as<uint64_t>(), r->d_skeys2.as<uint64_t>(),
                                                  r->d_svals.as<uint32_t>(), r->d_svals2.as<uint32_t>(), (unsigned)total, nq,
                                                  r->d_seg.as<uint32_t>(), r->d_seg.as<uint32_t>() + 1, 0, 64, st));
    hipLaunchKernelGGL(k_score_output, dim3(nq), dim3(64), 0, st, r->d_skeys2.as<uint64_t>(), r->d_svals2.as<uint32_t>(), d_bcounts, q0, nq,
                       NI, max_num_images, r->d_out_cnt.as<uint32_t>(), r->d_out_idx.as<uint32_t>(), r->d_out_score.as<float>());
    RCHK(ctx, hipGetLastError());
  }
  RCHK(ctx, hipEventRecord(r->ev1, st));
  RCHK(ctx, hipStreamSynchronize(st));
  float ms = 0.f;
  RCHK(ctx, hipEventElapsedTime(&ms, r->ev0, r->ev1));
  r->query_ms = ms;
  RCHK(ctx, hipMemcpy(counts, r->d_out_cnt.p, (size_t)NI * 4, hipMemcpyDefault));
  RCHK(ctx, hipMemcpy(image_idx, r->d_out_idx.p, (size_t)NI * max_num_images * 4, hipMemcpyDefault));
  RCHK(ctx, hipMemcpy(scores, r->d_out_score.p, (size_t)NI * max_num_images * 4, hipMemcpyDefault));
  return DSM_OK;
}

int dsm_retrieval_matches(dsm_ctx* ctx, uint32_t num_neighbors, uint32_t max_num_images, const uint32_t* counts, const uint32_t* image_idx,
                          uint64_t* offsets) {
  if (!ctx || !counts || !image_idx || !offsets) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r || !r->indexed) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_index has not run");
  if (num_neighbors == 0 || num_neighbors > RK_MAX) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "num_neighbors must be 1..8");
  if (max_num_images == 0) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "max_num_images must be > 0");
  const uint32_t NI = ctx->n_images;
  offsets[0] = 0;
  r->m_total = 0;
  if (NI == 0) return DSM_OK;
  if (r->host_ids || r->flann) {
    if (r->host_ids && num_neighbors != r->host_k_query) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "num_neighbors differs from the k of dsm_retrieval_set_word_ids");
    RCHK(ctx, hipSetDevice(ctx->device));
    const int rca = retrieval_assign(ctx, num_neighbors, ASSIGN_FOR_QUERY);
    if (rca != DSM_OK) return rca;
  }
  const size_t smem = ((size_t)NI + 31) / 32 * 4;
  if (smem > 60000) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "dsm_retrieval_matches: more than 480 000 resident images");
  if (r->num_words >= (1u << 24)) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "dsm_retrieval_matches: more than 2^24 visual words");
  RCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  RCHK(ctx, r->d_nfeat.reserve((size_t)NI * 4));
  RCHK(ctx, hipMemcpyAsync(r->d_nfeat.p, ctx->nfeat.data(), (size_t)NI * 4, hipMemcpyHostToDevice, st));
  RCHK(ctx, r->d_m_cnt_in.reserve((size_t)NI * 4));
  RCHK(ctx, r->d_m_idx_in.reserve((size_t)NI * max_num_images * 4));
  RCHK(ctx, hipMemcpyAsync(r->d_m_cnt_in.p, counts, (size_t)NI * 4, hipMemcpyDefault, st));
  RCHK(ctx, hipMemcpyAsync(r->d_m_idx_in.p, image_idx, (size_t)NI * max_num_images * 4, hipMemcpyDefault, st));
  RCHK(ctx, r->d_m_counts.reserve((size_t)NI * 4));
  RCHK(ctx, r->d_m_off.reserve(((size_t)NI + 1) * 8));
  auto launch = [&](bool write) {
    if (write)
      hipLaunchKernelGGL(k_vocab_matches<true>, dim3(NI), dim3(64), smem, st, r->d_wid.as<int32_t>(), r->d_sig.as<uint64_t>(),
                         ctx->d_img_row0.as<uint32_t>(), r->d_nfeat.as<uint32_t>(), NI, (int)num_neighbors, r->d_file_start.as<uint32_t>(),
                         r->d_e_img.as<int32_t>(), r->d_e_sig.as<uint64_t>(), r->d_e_row.as<uint32_t>(), r->d_m_cnt_in.as<uint32_t>(),
                         r->d_m_idx_in.as<uint32_t>(), max_num_images, r->d_m_counts.as<uint32_t>(), r->d_m_off.as<uint64_t>(),
                         r->d_m_tuples.as<uint32_t>());
    else
      hipLaunchKernelGGL(k_vocab_matches<false>, dim3(NI), dim3(64), smem, st, r->d_wid.as<int32_t>(), r->d_sig.as<uint64_t>(),
                         ctx->d_img_row0.as<uint32_t>(), r->d_nfeat.as<uint32_t>(), NI, (int)num_neighbors, r->d_file_start.as<uint32_t>(),
                         r->d_e_img.as<int32_t>(), r->d_e_sig.as<uint64_t>(), r->d_e_row.as<uint32_t>(), r->d_m_cnt_in.as<uint32_t>(),
                         r->d_m_idx_in.as<uint32_t>(), max_num_images, r->d_m_counts.as<uint32_t>(), r->d_m_off.as<uint64_t>(),
                         (uint32_t*)nullptr);
  };
  launch(false);
  RCHK(ctx, hipGetLastError());
  std::vector<uint32_t> mc(NI);
  RCHK(ctx, hipMemcpyAsync(mc.data(), r->d_m_counts.p, (size_t)NI * 4, hipMemcpyDeviceToHost, st));
  RCHK(ctx, hipStreamSynchronize(st));
  for (uint32_t q = 0; q < NI; ++q) offsets[q + 1] = offsets[q] + mc[q];
  r->m_total = offsets[NI];
  RCHK(ctx, hipMemcpyAsync(r->d_m_off.p, offsets, ((size_t)NI + 1) * 8, hipMemcpyHostToDevice, st));
  RCHK(ctx, r->d_m_tuples.reserve(std::max<uint64_t>(r->m_total, 1) * 20));
  launch(true);
  RCHK(ctx, hipGetLastError());
  RCHK(ctx, hipStreamSynchronize(st));
  return DSM_OK;
}

int dsm_get_retrieval_matches(dsm_ctx* ctx, uint32_t* tuples, uint64_t capacity) {
  if (!ctx || !ctx->retrieval) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (capacity < r->m_total) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "dsm_get_retrieval_matches: capacity below the total of dsm_retrieval_matches");
  if (r->m_total == 0) return DSM_OK;
  if (!tuples) return DSM_ERR_INVALID_ARGUMENT;
  RCHK(ctx, hipSetDevice(ctx->device));
  RCHK(ctx, hipMemcpy(tuples, r->d_m_tuples.p, (size_t)r->m_total * 20, hipMemcpyDefault));
  return DSM_OK;
}

int dsm_get_retrieval_idf(dsm_ctx* ctx, float* idf, uint32_t capacity) {
  if (!ctx || !ctx->retrieval || !idf) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r->indexed) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_index has not run");
  if (capacity < r->num_words) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "dsm_get_retrieval_idf: capacity below the number of visual words");
  std::memcpy(idf, r->idf_host.data(), (size_t)r->num_words * 4);
  return DSM_OK;
}

int dsm_retrieval_debug_word_ids(dsm_ctx* ctx, uint32_t image, uint32_t k, int32_t* out) {
  if (!ctx || !out) return DSM_ERR_INVALID_ARGUMENT;
  RetrievalState* r = ctx->retrieval;
  if (!r || !r->have_vocab) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_vocabulary has not run");
  if (image >= ctx->n_images || k == 0 || k > RK_MAX) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "image / k out of range");
  RCHK(ctx, hipSetDevice(ctx->device));
  r->k_assigned = 0;
  r->host_loaded = 0;
  int rc = retrieval_assign(ctx, r->flann ? k : RK_MAX, ASSIGN_FOR_QUERY);  // (an approximate search's first k of 8 are not its k of k)
  if (rc != DSM_OK) return rc;
  RCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int32_t> all((size_t)ctx->nfeat[image] * RK_MAX);
  if (!all.empty())
    RCHK(ctx, hipMemcpy(all.data(), r->d_wid.as<int32_t>() + (size_t)ctx->row0[image] * RK_MAX, all.size() * 4, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < ctx->nfeat[image]; ++i)
    for (uint32_t n = 0; n < k; ++n) out[(size_t)i * k + n] = all[(size_t)i * RK_MAX + n];
  return DSM_OK;
}

int dsm_get_retrieval_time(dsm_ctx* ctx, double* index_ms, double* query_ms) {
  if (!ctx || !ctx->retrieval) return DSM_ERR_INVALID_ARGUMENT;
  if (index_ms) *index_ms = ctx->retrieval->index_ms;
  if (query_ms) *query_ms = ctx->retrieval->query_ms;
  return DSM_OK;
}

}  // extern "C"
