// flann_search.hip -- the reference's OWN visual-word search on the device (round 5).
//
// VisualIndex::FindWordIds (/root/reference/src/retrieval/visual_index.h:695-738) asks the flann::AutotunedIndex loaded from
// the vocabulary file for APPROXIMATE nearest words: the answer is a function of the stored tree and of FLANN's visit order,
// not of the geometry alone (the device's exact search agrees with it for a fifth of the features on tree indices,
// profiles/r04_flann_agreement.json).  Round 4 restated that search on the host (dagsfm_amd/host/flann_index.cc, pinned bit
// for bit to the reference's FLANN compiled where it lies); this file runs the same search on the GPU, a LANE per query:
//
//   kd-trees   KDTreeIndex::getNeighbors / searchLevel (lib/FLANN/algorithms/kdtree_index.h:543-617): every tree descended
//              once, then the branch heap popped while checks remain; `checked` bitset over the words
//   k-means    KMeansIndex::findNeighborsWithRemoved / findNN / exploreNodeBranches (kmeans_index.h:717-833)
//   linear     LinearIndex::findNeighbors (linear_index.h:130-146)
//   shared     KNNSimpleResultSet::addPoint (util/result_set.h:101-199): k <= 8 entries in registers, sorted insertion;
//              Heap<BranchSt> (util/heap.h) = std::push_heap / std::pop_heap with "t_2 < t_1": libstdc++'s __push_heap /
//              __adjust_heap restated index for index (bits/stl_heap.h), so equal keys leave in the same order;
//              L2<uint8_t> in float (dist.h:133-178): for two uint8 vectors every partial sum is an integer below 2^24, so
//              the float sum is exact whatever the order -- computed as |q|^2 + |w|^2 - 2 q.w with v_dot4 on the s8 rows the
//              context already holds; against a k-means pivot (float) the functor's own order: four differences, their
//              squares summed left to right, one addition into the running sum (-ffp-contract=off: no FMA)
//
// Per-lane state that does not fit registers lives in global memory -- the `checked` bitset, its list and the k-means domain
// distances lane-interleaved (element e of lane L at [e][L]), the branch heap contiguous per lane -- : the
// branch heap (FLANN sizes it num_words and drops inserts when full: the same here up to FLANN_HEAP_CAP entries; a lane that
// would need more raises `overflow` and the call fails -- never a silent difference), the `checked` bitset with the list of
// set bits (cleared bit by bit after the query), the k-means domain distances.  The query's 128 bytes sit in registers
// (distances) and in LDS (the kd-trees read vec[divfeat], a per-lane dynamic index).
//
// Lanes of a wave walk different trees: the wave executes the union of their paths.  The search is latency- and
// divergence-bound, not throughput-bound; it exists so that the reference-identical mode of (f2) is a device component and the
// host's threads are out of the retrieval path (tools/bench_retrieval.py keeps both rates).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "ctx.h"
#include "flann_search.h"

#define FLANN_HEAP_CAP 4096u      /* first pass: every resident lane gets this many branch-heap entries */
#define FLANN_HEAP_CAP_RETRY (1u << 20) /* second pass over the queries that needed more (FLANN's own capacity is num_words) */
#define FLANN_K_MAX 8
#define FLANN_INVALID 0x7fffffff

struct FlannKmNodeDev {
  uint32_t pivot;  // index of the node's centre, in units of 128 floats
  float radius, variance;
  int32_t size;
  uint32_t first_child, num_childs, first_point, pad;
};

struct FlannSearchParams {
  int32_t algorithm, num_checks;
  uint32_t num_words;
  int32_t branching;
  float cb_index;
  int32_t km_root;
  const dsm_flann_kd_node* kd_nodes;
  const int32_t* kd_roots;
  uint32_t n_kd_roots;
  const FlannKmNodeDev* km_nodes;
  const int32_t* km_childs;
  const uint32_t* km_points;
  const float* pivots;
  const int8_t* words;    // s8 rows (u8 ^ 0x80), [num_words][128]
  const int32_t* wnorm;   // sum of squares of the s8 row
  const int8_t* desc;     // queries: s8 rows, [n_rows][128]
  const int32_t* row_img; // null, or < 0 for a padding row (no query)
  uint64_t n_rows;
  uint32_t k;
  uint2* heap;            // [heap_cap][n_lanes]: (node, mindist bits)
  uint32_t heap_cap;
  uint32_t* checked;      // [checked_words][n_lanes]
  uint32_t checked_words;
  uint32_t* checked_list; // [list_cap][n_lanes]
  uint32_t list_cap;
  float* domain;          // [branching][n_lanes]
  uint32_t n_lanes;
  int32_t* out_ids;       // [n_rows][out_stride]
  float* out_dists;       // null, or [n_rows][out_stride]
  uint32_t out_stride;
  uint32_t* overflow;     // number of queries whose heap would have passed heap_cap < num_words ...
  uint32_t* overflow_rows;  // ... and their rows (null: not recorded); such a query writes no result
  const uint32_t* row_list; // null, or the rows to search (the second pass over the overflowed queries); n_rows = its length
  uint32_t* stats;          // null, or 16 counters of the group kernel (DSM_FLANN_STATS, check build): see flann_launch_kd_grp
};

namespace {

struct ResultSet {  // KNNSimpleResultSet<float>
  float d[FLANN_K_MAX];
  int32_t i[FLANN_K_MAX];
  int count, capacity;
  float worst;
  __device__ __forceinline__ void clear(int k) {
    capacity = k;
    count = 0;
    worst = FLT_MAX;
#pragma unroll
    for (int j = 0; j < FLANN_K_MAX; ++j) {
      d[j] = FLT_MAX;
      i[j] = -1;
    }
  }
  __device__ __forceinline__ bool full() const { return count == capacity; }
  __device__ __forceinline__ void add(float dist, int32_t index) {
    if (dist >= worst) return;
    if (count < capacity) ++count;
    // the reference shifts entries from the top while they are greater: the new entry lands behind the sorted prefix <= dist
    int pos = 0;
#pragma unroll
    for (int e = 0; e < FLANN_K_MAX - 1; ++e) pos += (e < count - 1 && d[e] <= dist) ? 1 : 0;
#pragma unroll
    for (int j = FLANN_K_MAX - 1; j > 0; --j) {
      const bool move = j <= count - 1 && j > pos;
      d[j] = move ? d[j - 1] : d[j];
      i[j] = move ? i[j - 1] : i[j];
    }
#pragma unroll
    for (int j = 0; j < FLANN_K_MAX; ++j) {
      d[j] = j == pos ? dist : d[j];
      i[j] = j == pos ? index : i[j];
    }
    float w = FLT_MAX;
#pragma unroll
    for (int j = 0; j < FLANN_K_MAX; ++j) w = j == capacity - 1 ? d[j] : w;
    worst = w;
  }
};

// Heap<BranchSt>: libstdc++'s push_heap / pop_heap over (node, mindist), comp(a, b) = b.mindist < a.mindist
struct BranchHeap {
  uint2* base;  // this lane's column
  uint32_t stride, cap, count;
  bool flann_cap;  // cap == num_words: a full heap drops the insert, as FLANN's does
  __device__ __forceinline__ uint2 at(uint32_t e) const { return base[(size_t)e * stride]; }
  __device__ __forceinline__ void put(uint32_t e, uint2 v) { base[(size_t)e * stride] = v; }
  __device__ __forceinline__ void push_from(uint32_t hole, uint32_t top, uint2 value) {  // std::__push_heap
    const float vk = __uint_as_float(value.y);
    while (hole > top) {
      const uint32_t parent = (hole - 1) / 2;
      const uint2 pv = at(parent);
      if (!(vk < __uint_as_float(pv.y))) break;  // comp(first[parent], value)
      put(hole, pv);
      hole = parent;
    }
    put(hole, value);
  }
  // returns false when the entry could not be stored although FLANN's heap would have taken it
  __device__ __forceinline__ bool insert(int32_t node, float mindist) {
    if (count == cap) return flann_cap;
    push_from(count, 0u, make_uint2((uint32_t)node, __float_as_uint(mindist)));
    ++count;
    return true;
  }
  __device__ __forceinline__ void pop_min(int32_t* node, float* mindist) {  // count > 0
    const uint2 top = at(0);
    *node = (int32_t)top.x;
    *mindist = __uint_as_float(top.y);
    --count;
    if (count == 0) return;  // std::pop_heap on one element moves nothing
    const uint2 value = at(count);  // the last element; __adjust_heap(first, 0, len = count, value)
    const uint32_t len = count;
    uint32_t hole = 0, second = 0;
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      const uint2 a = at(second), b = at(second - 1);
      uint2 pick = a;
      if (__uint_as_float(b.y) < __uint_as_float(a.y)) {  // comp(first[second], first[second - 1])
        --second;
        pick = b;
      }
      put(hole, pick);
      hole = second;
    }
    if ((len & 1u) == 0u && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      put(hole, at(second - 1));
      hole = second - 1;
    }
    push_from(hole, 0u, value);
  }
};

__device__ __forceinline__ int dot_s8_128(const int (&q)[32], const int8_t* w) {
  const int4* w4 = reinterpret_cast<const int4*>(w);
  int acc = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int4 v = w4[u];
    acc = __builtin_amdgcn_sdot4(q[4 * u + 0], v.x, acc, false);
    acc = __builtin_amdgcn_sdot4(q[4 * u + 1], v.y, acc, false);
    acc = __builtin_amdgcn_sdot4(q[4 * u + 2], v.z, acc, false);
    acc = __builtin_amdgcn_sdot4(q[4 * u + 3], v.w, acc, false);
  }
  return acc;
}

// L2<unsigned char>::operator()(a = the query, b = a float pivot): dist.h:133-178 with worst_dist = -1
__device__ __forceinline__ float dist_u8_f32(const int (&q)[32], const float* b) {
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float result = 0.0f;
#pragma unroll 8
  for (int u = 0; u < 32; ++u) {
    const float4 v = b4[u];
    const uint32_t w = (uint32_t)q[u] ^ 0x80808080u;  // s8 -> the uint8 the reference sees
    const float d0 = (float)(w & 0xffu) - v.x;
    const float d1 = (float)((w >> 8) & 0xffu) - v.y;
    const float d2 = (float)((w >> 16) & 0xffu) - v.z;
    const float d3 = (float)(w >> 24) - v.w;
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  return result;
}

}  // namespace

#ifndef FLANN_LANE_WAVES
#define FLANN_LANE_WAVES 4
#endif
template <int ALGO>  // 0 linear, 1 kd-trees, 2 k-means: one instance each (the union of their live state spills scalar registers)
__global__ __launch_bounds__(64, FLANN_LANE_WAVES) void k_flann_search(const FlannSearchParams p) {
  __shared__ uint32_t qs[32][64];  // the queries of the wave, dword d of lane l at [d][l]
  const int lane = threadIdx.x;
  const uint32_t L = blockIdx.x * 64u + (uint32_t)lane;
  BranchHeap heap;
  // a lane's heap is CONTIGUOUS (round 6: the two children of a node share a 64-byte sector, the top three levels sit in one; the
  // lane-interleaved layout of round 5 -- entry e of lane L at [e][L] -- only coalesces while the lanes are at the same entry:
  // kd-trees 3.65 -> 3.84 M searches/s, k-means 3.52 -> 3.90, profiles/r06_flann_group_per_query.txt)
  heap.base = p.heap + (size_t)L * p.heap_cap;
  heap.stride = 1;
  heap.cap = p.heap_cap;
  heap.flann_cap = p.heap_cap >= p.num_words;
  ResultSet rs;
  const int k = (int)p.k;
  const int max_check = p.num_checks;
  for (uint64_t row0 = (uint64_t)blockIdx.x * 64u; row0 < p.n_rows; row0 += (uint64_t)p.n_lanes) {
    const uint64_t slot = row0 + (uint64_t)lane;
    const bool in_list = slot < p.n_rows;
    const uint64_t row = p.row_list ? (uint64_t)p.row_list[in_list ? slot : 0] : slot;
    const bool has_query = in_list && (!p.row_img || p.row_img[row] >= 0);
    int q[32];
    {
      const int4* src = reinterpret_cast<const int4*>(p.desc + (size_t)(in_list ? row : 0) * 128);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int4 v = src[u];
        q[4 * u + 0] = v.x;
        q[4 * u + 1] = v.y;
        q[4 * u + 2] = v.z;
        q[4 * u + 3] = v.w;
      }
    }
    __syncthreads();  // the previous query's readers are done with qs
#pragma unroll
    for (int d = 0; d < 32; ++d) qs[d][lane] = (uint32_t)q[d];
    __syncthreads();
    int qn = 0;
#pragma unroll
    for (int d = 0; d < 32; ++d) qn = __builtin_amdgcn_sdot4(q[d], q[d], qn, false);
    rs.clear(k);
    heap.count = 0;
    bool lost = false;  // an insert FLANN would have kept did not fit
    if (has_query) {
      if constexpr (ALGO == 0) {
        for (uint32_t w = 0; w < p.num_words; ++w) {
          const int dist = qn + p.wnorm[w] - 2 * dot_s8_128(q, p.words + (size_t)w * 128);
          rs.add((float)dist, (int32_t)w);
        }
      } else if constexpr (ALGO == 1) {
        uint32_t* checked = p.checked + L;
        uint32_t* clist = p.checked_list + L;
        uint32_t n_list = 0;
        bool list_lost = false;
        int check_count = 0;
        uint32_t root_i = 0;
        for (;;) {
          int32_t node;
          float mindist;
          if (root_i < p.n_kd_roots) {
            node = p.kd_roots[root_i++];
            mindist = 0.0f;
          } else {
            if (heap.count == 0) break;
            heap.pop_min(&node, &mindist);
            if (!(check_count < max_check || !rs.full())) break;
          }
          for (;;) {  // searchLevel; the recursion into the best child as a loop
            if (rs.worst < mindist) break;
            const dsm_flann_kd_node nd = p.kd_nodes[node];
            if (nd.child1 < 0 && nd.child2 < 0) {
              const uint32_t index = (uint32_t)nd.divfeat;
              const uint32_t cell = checked[(size_t)(index >> 5) * p.n_lanes];
              const uint32_t bit = 1u << (index & 31u);
              if ((cell & bit) != 0u || (check_count >= max_check && rs.full())) break;
              checked[(size_t)(index >> 5) * p.n_lanes] = cell | bit;
              if (n_list < p.list_cap)
                clist[(size_t)n_list * p.n_lanes] = index;
              else
                list_lost = true;
              ++n_list;
              ++check_count;
              const int dist = qn + p.wnorm[index] - 2 * dot_s8_128(q, p.words + (size_t)index * 128);
              rs.add((float)dist, (int32_t)index);
              break;
            }
            const uint32_t word = qs[nd.divfeat >> 2][lane];
            const uint32_t val = ((word >> (8 * (nd.divfeat & 3))) & 0xffu) ^ 0x80u;
            const float diff = (float)val - nd.divval;
            const int32_t best_child = diff < 0 ? nd.child1 : nd.child2;
            const int32_t other_child = diff < 0 ? nd.child2 : nd.child1;
            const float new_distsq = mindist + diff * diff;
            if (new_distsq * 1.0f < rs.worst || !rs.full())
              if (!heap.insert(other_child, new_distsq)) lost = true;
            node = best_child;
          }
        }
        // the bitset goes back to zero for the lane's next query
        if (list_lost) {
          for (uint32_t c = 0; c < p.checked_words; ++c) checked[(size_t)c * p.n_lanes] = 0u;
        } else {
          for (uint32_t c = 0; c < n_list; ++c) checked[(size_t)(clist[(size_t)c * p.n_lanes] >> 5) * p.n_lanes] = 0u;
        }
      } else {
        float* domain = p.domain + L;
        int checks = 0;
        bool first = true;
        for (;;) {
          int32_t node;
          float key;
          if (first) {
            node = p.km_root;
            first = false;
          } else {
            if (heap.count == 0) break;
            heap.pop_min(&node, &key);
            if (!(checks < max_check || !rs.full())) break;
          }
          for (;;) {  // findNN
            const FlannKmNodeDev nd = p.km_nodes[node];
            {
              const float bsq = dist_u8_f32(q, p.pivots + (size_t)nd.pivot * 128);
              const float rsq = nd.radius;
              const float wsq = rs.worst;
              const float val = bsq - rsq - wsq;
              const float val2 = val * val - 4 * rsq * wsq;
              if (val > 0 && val2 > 0) break;
            }
            if (nd.num_childs == 0) {
              if (checks >= max_check && rs.full()) break;
              for (int i = 0; i < nd.size; ++i) {
                const uint32_t index = p.km_points[(size_t)nd.first_point + (uint32_t)i];
                const int dist = qn + p.wnorm[index] - 2 * dot_s8_128(q, p.words + (size_t)index * 128);
                rs.add((float)dist, (int32_t)index);
                ++checks;
              }
              break;
            }
            const int32_t* childs = p.km_childs + nd.first_child;
            int best_index = 0;
            float best_dist = dist_u8_f32(q, p.pivots + (size_t)p.km_nodes[childs[0]].pivot * 128);
            domain[0] = best_dist;
            for (int i = 1; i < p.branching; ++i) {
              const float dd = dist_u8_f32(q, p.pivots + (size_t)p.km_nodes[childs[i]].pivot * 128);
              domain[(size_t)i * p.n_lanes] = dd;
              if (dd < best_dist) {
                best_dist = dd;
                best_index = i;
              }
            }
            for (int i = 0; i < p.branching; ++i) {
              if (i != best_index) {
                const float dd = domain[(size_t)i * p.n_lanes] - p.cb_index * p.km_nodes[childs[i]].variance;
                if (!heap.insert(childs[i], dd)) lost = true;
              }
            }
            node = childs[best_index];
          }
        }
      }
    }
    if (lost) {  // the heap was too small for this query: no result from this pass (the host searches it again with more room)
      const uint32_t at = atomicAdd(p.overflow, 1u);
      if (p.overflow_rows) p.overflow_rows[at] = (uint32_t)row;
    }
    if (in_list && !lost) {
#pragma unroll
      for (int j = 0; j < FLANN_K_MAX; ++j) {
        if (j < (int)p.out_stride) {
          const bool have = has_query && j < k && j < rs.count;
          p.out_ids[(size_t)row * p.out_stride + j] = have ? rs.i[j] : FLANN_INVALID;
          if (p.out_dists) p.out_dists[(size_t)row * p.out_stride + j] = have ? rs.d[j] : 0.0f;
        }
      }
    }
  }
}

#ifdef DSM_CHECK_BUILD
// ------------------------------------------------------------------------------------------------ kd-trees, a 16-lane group per query (round 6)
// MEASURED AND NOT ADOPTED -- check build only (DSM_FLANN_GROUP), kept as the A/B the round-5 verdict asked for with the counters that
// say why (profiles/r06_flann_group_per_query.txt).  A query belongs to a GROUP of 16 lanes (one DPP row), four queries per wave: a tree
// node, the `checked` word and the query byte vec[divfeat] (LDS) are read once per group, a leaf's word is split over the lanes (8 bytes,
// two v_dot4 each, the sixteen integer partial sums added across the row: the float distance is the lane-per-query kernel's bit for
// bit), every lane runs libstdc++'s heap code on the same addresses, the result set is kept redundantly in every lane.  Same visit
// order, same heap order, same capacity semantics as k_flann_search<1>: identical ids and distances (tests/test_retrieval_flann.py
// and tools/bench_flann_search.py passed with it as the default path of both builds before it moved here).
// What the counters of this kernel say about the search itself (kd-trees x 4 over 65 536 words, 256 checks, per query): 1 305 tree nodes
// visited, 256 leaves, 260 pops and 1 042 PUSHES, the branch heap reaching 1 168 entries -- more than 511 for 99.99 % of the queries.
// So (1) the heap does not fit LDS at any useful residency: with 512 entries per query in LDS (the first form) all but 80 of 819 200
// queries overflowed into the second pass (1.67 M searches/s); (2) with the heap in global memory, contiguous per query, a query is
// ~7 000 DEPENDENT memory round trips (every heap level is one) of ~1.8 us each = 12.6 ms, and only 24 576 queries are in flight
// (96 per CU): 2.05 M searches/s against the lane-per-query kernel's 3.6 M/s, which keeps 131 072 queries in flight and is bound by L2
// sector throughput (~13 000 sector accesses per query).  The search is a pointer chase through a priority queue: its rate is
// queries-in-flight / latency, and a group spends 16 lanes on one chase.  10 M searches/s would need ~2 000 round trips per query with
// 20 000 queries in flight, i.e. the whole heap (9 KB) in LDS for each of them: 180 MB.
#define FLANN_GRP 16u
namespace {
struct LdsHeap {  // BranchHeap over an LDS array (all lanes of the group run the same code on the same addresses)
  uint2* base;
  uint32_t cap, count;
  bool flann_cap;
  __device__ __forceinline__ uint2 at(uint32_t e) const { return base[e]; }
  __device__ __forceinline__ void put(uint32_t e, uint2 v) { base[e] = v; }
  __device__ __forceinline__ void push_from(uint32_t hole, uint32_t top, uint2 value) {  // std::__push_heap
    const float vk = __uint_as_float(value.y);
    while (hole > top) {
      const uint32_t parent = (hole - 1) / 2;
      const uint2 pv = at(parent);
      if (!(vk < __uint_as_float(pv.y))) break;
      put(hole, pv);
      hole = parent;
    }
    put(hole, value);
  }
  __device__ __forceinline__ bool insert(int32_t node, float mindist) {
    if (count == cap) return flann_cap;
    push_from(count, 0u, make_uint2((uint32_t)node, __float_as_uint(mindist)));
    ++count;
    return true;
  }
  __device__ __forceinline__ void pop_min(int32_t* node, float* mindist) {  // count > 0
    const uint2 top = at(0);
    *node = (int32_t)top.x;
    *mindist = __uint_as_float(top.y);
    --count;
    if (count == 0) return;
    const uint2 value = at(count);
    const uint32_t len = count;
    uint32_t hole = 0, second = 0;
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      const uint2 a = at(second), b = at(second - 1);
      uint2 pick = a;
      if (__uint_as_float(b.y) < __uint_as_float(a.y)) {
        --second;
        pick = b;
      }
      put(hole, pick);
      hole = second;
    }
    if ((len & 1u) == 0u && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      put(hole, at(second - 1));
      hole = second - 1;
    }
    push_from(hole, 0u, value);
  }
};
__device__ __forceinline__ int row16_sum(int v) {  // the sum over the 16 lanes of the group, in every lane
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
}  // namespace

__global__ __launch_bounds__(64) void k_flann_search_kd_grp(const FlannSearchParams p) {
  __shared__ uint32_t s_q[64 / FLANN_GRP][32];  // the groups' queries (s8 dwords): vec[divfeat] is a dynamic index
  const int lane = threadIdx.x;
  const uint32_t g = (uint32_t)lane / FLANN_GRP, gl = (uint32_t)lane % FLANN_GRP;
  const uint32_t n_groups = gridDim.x * (64u / FLANN_GRP);
  const uint32_t G = blockIdx.x * (64u / FLANN_GRP) + g;
  LdsHeap heap;
  heap.base = p.heap + (size_t)G * p.heap_cap;  // (contiguous per query: an access of the group is ONE address)
  heap.cap = p.heap_cap;
  heap.flann_cap = p.heap_cap >= p.num_words;
  uint32_t* checked = p.checked + (size_t)G * p.checked_words;  // (this kernel's layout: a query's bitset and list are contiguous)
  uint32_t* clist = p.checked_list + (size_t)G * p.list_cap;
  ResultSet rs;
  const int k = (int)p.k;
  const int max_check = p.num_checks;
  for (uint64_t slot = G; slot < p.n_rows; slot += n_groups) {
    const uint64_t row = p.row_list ? (uint64_t)p.row_list[slot] : slot;
    const bool has_query = !p.row_img || p.row_img[row] >= 0;
    // this lane's 8 bytes of the query
    const int2 qv = reinterpret_cast<const int2*>(p.desc + (size_t)row * 128)[gl];
    __syncthreads();  // (one-wave workgroup: a fence; the previous query's readers of s_q are this wave's earlier instructions)
    s_q[g][2 * gl] = (uint32_t)qv.x;
    s_q[g][2 * gl + 1] = (uint32_t)qv.y;
    __syncthreads();
    const int qn = row16_sum(__builtin_amdgcn_sdot4(qv.y, qv.y, __builtin_amdgcn_sdot4(qv.x, qv.x, 0, false), false));
    rs.clear(k);
    heap.count = 0;
    bool lost = false;
    uint32_t st_nodes = 0, st_pops = 0, st_push = 0, st_maxheap = 0;
    if (has_query) {
      uint32_t n_list = 0;
      bool list_lost = false;
      int check_count = 0;
      uint32_t root_i = 0;
      for (;;) {
        int32_t node;
        float mindist;
        if (root_i < p.n_kd_roots) {
          node = p.kd_roots[root_i++];
          mindist = 0.0f;
        } else {
          if (heap.count == 0) break;
          heap.pop_min(&node, &mindist);
          ++st_pops;
          if (!(check_count < max_check || !rs.full())) break;
        }
        for (;;) {  // searchLevel; the recursion into the best child as a loop
          if (rs.worst < mindist) break;
          ++st_nodes;
          const dsm_flann_kd_node nd = p.kd_nodes[node];
          if (nd.child1 < 0 && nd.child2 < 0) {
            const uint32_t index = (uint32_t)nd.divfeat;
            const uint32_t cell = checked[index >> 5];
            const uint32_t bit = 1u << (index & 31u);
            if ((cell & bit) != 0u || (check_count >= max_check && rs.full())) break;
            checked[index >> 5] = cell | bit;
            if (n_list < p.list_cap)
              clist[n_list] = index;
            else
              list_lost = true;
            ++n_list;
            ++check_count;
            const int2 wv = reinterpret_cast<const int2*>(p.words + (size_t)index * 128)[gl];
            const int dot = row16_sum(__builtin_amdgcn_sdot4(qv.y, wv.y, __builtin_amdgcn_sdot4(qv.x, wv.x, 0, false), false));
            const int dist = qn + p.wnorm[index] - 2 * dot;
            rs.add((float)dist, (int32_t)index);
            break;
          }
          const uint32_t word = s_q[g][nd.divfeat >> 2];
          const uint32_t val = ((word >> (8 * (nd.divfeat & 3))) & 0xffu) ^ 0x80u;
          const float diff = (float)val - nd.divval;
          const int32_t best_child = diff < 0 ? nd.child1 : nd.child2;
          const int32_t other_child = diff < 0 ? nd.child2 : nd.child1;
          const float new_distsq = mindist + diff * diff;
          if (new_distsq * 1.0f < rs.worst || !rs.full()) {
            if (!heap.insert(other_child, new_distsq)) lost = true;
            ++st_push;
            st_maxheap = heap.count > st_maxheap ? heap.count : st_maxheap;
          }
          node = best_child;
        }
      }
      if (p.stats && gl == 0) {  // (debug) [0] queries [1] nodes visited [2] leaves checked [3] pops [4] pushes [5] max heap; [6..11] queries whose heap passed 32 / 64 / 128 / 256 / 384 / 511
        atomicAdd(p.stats + 0, 1u);
        atomicAdd(p.stats + 1, st_nodes);
        atomicAdd(p.stats + 2, (uint32_t)check_count);
        atomicAdd(p.stats + 3, st_pops);
        atomicAdd(p.stats + 4, st_push);
        atomicMax(p.stats + 5, st_maxheap);
        const uint32_t lim[6] = {32, 64, 128, 256, 384, 511};
        for (int b = 0; b < 6; ++b)
          if (st_maxheap > lim[b]) atomicAdd(p.stats + 6 + b, 1u);
      }
      // the bitset goes back to zero for the group's next query
      if (list_lost) {
        for (uint32_t c = gl; c < p.checked_words; c += FLANN_GRP) checked[c] = 0u;
      } else {
        for (uint32_t c = gl; c < n_list; c += FLANN_GRP) checked[clist[c] >> 5] = 0u;
      }
      __threadfence_block();
    }
    if (lost && gl == 0) {  // the LDS heap was too small for this query: no result from this pass
      const uint32_t at = atomicAdd(p.overflow, 1u);
      if (p.overflow_rows) p.overflow_rows[at] = (uint32_t)row;
    }
    if (!lost && gl < p.out_stride) {
      // lane j of the group writes entry j (the result set is the same in every lane)
      float dj = 0.0f;
      int32_t ij = FLANN_INVALID;
#pragma unroll
      for (int j = 0; j < FLANN_K_MAX; ++j) {
        if ((uint32_t)j == gl && has_query && j < k && j < rs.count) {
          dj = rs.d[j];
          ij = rs.i[j];
        }
      }
      p.out_ids[(size_t)row * p.out_stride + gl] = ij;
      if (p.out_dists) p.out_dists[(size_t)row * p.out_stride + gl] = dj;
    }
  }
}

#endif  // DSM_CHECK_BUILD

__global__ void k_flann_word_norms(const int8_t* words, uint32_t n, int32_t* out) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  const int* r = reinterpret_cast<const int*>(words + (size_t)w * 128);
  int acc = 0;
  for (int d = 0; d < 32; ++d) acc = __builtin_amdgcn_sdot4(r[d], r[d], acc, false);
  out[w] = acc;
}

__global__ void k_flann_u8_to_s8(const uint32_t* in, uint32_t* out, uint64_t n_words) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) out[i] = in[i] ^ 0x80808080u;
}

// ------------------------------------------------------------------------------------------------ host side
struct FlannDevice {
  int32_t algorithm = -1, num_checks = 0, branching = 0, km_root = -1;
  float cb_index = 0.f;
  uint32_t num_words = 0, n_kd_roots = 0;
  DevBuf kd_nodes, kd_roots, km_nodes, km_childs, km_points, pivots, wnorm;
  DevBuf heap, checked, clist, domain, overflow, overflow_rows, overflow_rows2, q_s8, q_u8, out_ids, out_dists;
  DevBuf stats;
  DevBuf checked_g, clist_g;  // the group-per-query kernel's bitsets and lists ([group][word]: its own layout)
  uint32_t n_groups = 0;
  uint32_t n_lanes = 0;
  double last_ms = 0.0;
  uint32_t last_retried = 0;  // queries of the last search that took the second pass
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

void flann_device_destroy(FlannDevice* f) {
  if (!f) return;
  for (DevBuf* b : {&f->kd_nodes, &f->kd_roots, &f->km_nodes, &f->km_childs, &f->km_points, &f->pivots, &f->wnorm, &f->heap, &f->checked, &f->clist,
                    &f->domain, &f->overflow, &f->overflow_rows, &f->overflow_rows2, &f->checked_g, &f->clist_g, &f->stats, &f->q_s8, &f->q_u8, &f->out_ids,
                    &f->out_dists})
    b->release();
  if (f->ev0) (void)hipEventDestroy(f->ev0);
  if (f->ev1) (void)hipEventDestroy(f->ev1);
  delete f;
}

#define FCHK(ctx, call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
      return DSM_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

// Every index the kernel will follow is checked here, once: a damaged or hostile index is an error, never an out-of-bounds read.
int flann_device_set_index(dsm_ctx* ctx, FlannDevice** slot, const dsm_flann_index* ix, const int8_t* d_words_s8, uint32_t num_words) {
  if (*slot) {
    flann_device_destroy(*slot);
    *slot = nullptr;
  }
  if (!ix) return DSM_OK;
  if (ix->num_words != num_words) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "dsm_retrieval_set_flann_index: the index is over another number of words");
  if (ix->algorithm < 0 || ix->algorithm > 2) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "dsm_retrieval_set_flann_index: algorithm must be linear / kd-trees / k-means");
  if (ix->algorithm != 0 && ix->num_checks < 0) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "dsm_retrieval_set_flann_index: num_checks must be >= 0 on a tree index");
  std::vector<FlannKmNodeDev> km;
  std::vector<uint32_t> kmp;
  if (ix->algorithm == 1) {
    if (!ix->kd_nodes || !ix->kd_roots || ix->n_kd_roots == 0 || ix->n_kd_nodes == 0) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "kd-tree index without nodes");
    for (uint32_t r = 0; r < ix->n_kd_roots; ++r)
      if (ix->kd_roots[r] < 0 || (uint32_t)ix->kd_roots[r] >= ix->n_kd_nodes) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "kd-tree root outside the node array");
    for (uint32_t i = 0; i < ix->n_kd_nodes; ++i) {
      const dsm_flann_kd_node& nd = ix->kd_nodes[i];
      const bool leaf = nd.child1 < 0 && nd.child2 < 0;
      if (leaf) {
        if (nd.divfeat < 0 || (uint32_t)nd.divfeat >= num_words) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "kd-tree leaf outside the vocabulary");
      } else {
        // children lie behind their parent in the loader's pre-order array: the descent terminates
        if (nd.child1 <= (int32_t)i || nd.child2 <= (int32_t)i || (uint32_t)nd.child1 >= ix->n_kd_nodes || (uint32_t)nd.child2 >= ix->n_kd_nodes)
          return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "kd-tree child index");
        if (nd.divfeat < 0 || nd.divfeat >= 128) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "kd-tree split dimension");
      }
    }
  } else if (ix->algorithm == 2) {
    if (!ix->km_nodes || ix->n_km_nodes == 0 || !ix->pivots || ix->branching < 2 || ix->km_root < 0 || (uint32_t)ix->km_root >= ix->n_km_nodes)
      return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "k-means index without nodes");
    // (the offsets are 64-bit in the file and 32-bit on the device: sums are never formed -- an offset near 2^64 must not wrap past the
    // check -- and everything that is narrowed is shown to fit first)
    if (ix->n_pivot_floats < 128 || ix->n_pivot_floats / 128 > 0xffffffffull || ix->n_km_points > 0xffffffffull)
      return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "k-means index: pivot / point arrays beyond 32-bit device offsets");
    km.resize(ix->n_km_nodes);
    for (uint32_t i = 0; i < ix->n_km_nodes; ++i) {
      const dsm_flann_km_node& nd = ix->km_nodes[i];
      if (nd.pivot % 128 != 0 || nd.pivot > ix->n_pivot_floats - 128) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "k-means pivot offset");
      if (nd.num_childs == 0) {
        if (nd.size < 0 || (uint64_t)nd.size > ix->n_km_points || nd.first_point > ix->n_km_points - (uint64_t)nd.size || (nd.size > 0 && !ix->km_points))
          return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "k-means leaf points");
      } else {
        if (nd.num_childs != (uint32_t)ix->branching || (uint64_t)nd.first_child + nd.num_childs > ix->n_km_childs || !ix->km_childs)
          return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "k-means child list");
        for (uint32_t c = 0; c < nd.num_childs; ++c) {
          const int32_t ch = ix->km_childs[nd.first_child + c];
          if (ch <= (int32_t)i || (uint32_t)ch >= ix->n_km_nodes) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "k-means child index");
        }
      }
      km[i].pivot = (uint32_t)(nd.pivot / 128);
      km[i].radius = nd.radius;
      km[i].variance = nd.variance;
      km[i].size = nd.size;
      km[i].first_child = nd.first_child;
      km[i].num_childs = nd.num_childs;
      km[i].first_point = (uint32_t)nd.first_point;
      km[i].pad = 0;
    }
    kmp.resize(std::max<uint64_t>(ix->n_km_points, 1), 0);
    for (uint64_t i = 0; i < ix->n_km_points; ++i) {
      if (ix->km_points[i] >= num_words) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "k-means point outside the vocabulary");
      kmp[i] = (uint32_t)ix->km_points[i];
    }
  }
  FlannDevice* f = new FlannDevice();
  *slot = f;
  f->algorithm = ix->algorithm;
  f->num_checks = ix->num_checks;
  f->branching = ix->branching;
  f->km_root = ix->km_root;
  f->cb_index = ix->cb_index;
  f->num_words = num_words;
  f->n_kd_roots = ix->n_kd_roots;
  FCHK(ctx, hipSetDevice(ctx->device));
  if (ix->algorithm == 1) {
    FCHK(ctx, f->kd_nodes.reserve((size_t)ix->n_kd_nodes * sizeof(dsm_flann_kd_node)));
    FCHK(ctx, f->kd_roots.reserve((size_t)ix->n_kd_roots * 4));
    FCHK(ctx, hipMemcpy(f->kd_nodes.p, ix->kd_nodes, (size_t)ix->n_kd_nodes * sizeof(dsm_flann_kd_node), hipMemcpyHostToDevice));
    FCHK(ctx, hipMemcpy(f->kd_roots.p, ix->kd_roots, (size_t)ix->n_kd_roots * 4, hipMemcpyHostToDevice));
  } else if (ix->algorithm == 2) {
    FCHK(ctx, f->km_nodes.reserve(km.size() * sizeof(FlannKmNodeDev)));
    FCHK(ctx, f->km_childs.reserve(std::max<uint64_t>(ix->n_km_childs, 1) * 4));
    FCHK(ctx, f->km_points.reserve(kmp.size() * 4));
    FCHK(ctx, f->pivots.reserve((size_t)ix->n_pivot_floats * 4));
    FCHK(ctx, hipMemcpy(f->km_nodes.p, km.data(), km.size() * sizeof(FlannKmNodeDev), hipMemcpyHostToDevice));
    if (ix->n_km_childs) FCHK(ctx, hipMemcpy(f->km_childs.p, ix->km_childs, (size_t)ix->n_km_childs * 4, hipMemcpyHostToDevice));
    FCHK(ctx, hipMemcpy(f->km_points.p, kmp.data(), kmp.size() * 4, hipMemcpyHostToDevice));
    FCHK(ctx, hipMemcpy(f->pivots.p, ix->pivots, (size_t)ix->n_pivot_floats * 4, hipMemcpyHostToDevice));
  }
  FCHK(ctx, f->wnorm.reserve((size_t)num_words * 4));
  hipLaunchKernelGGL(k_flann_word_norms, dim3((num_words + 255) / 256), dim3(256), 0, ctx->stream, d_words_s8, num_words, f->wnorm.as<int32_t>());
  FCHK(ctx, hipGetLastError());
  FCHK(ctx, hipEventCreate(&f->ev0));
  FCHK(ctx, hipEventCreate(&f->ev1));
  FCHK(ctx, hipStreamSynchronize(ctx->stream));
  return DSM_OK;
}

// One launch: `n_items` queries (rows 0 .. n_items, or row_list's) with heap_cap branch-heap entries per resident lane.
static int flann_launch(dsm_ctx* ctx, FlannDevice* f, const int8_t* d_words_s8, const int8_t* desc, const int32_t* row_img, uint64_t n_items,
                        const uint32_t* row_list, uint32_t k, int32_t* out_ids, float* out_dists, uint32_t out_stride, uint32_t heap_cap,
                        uint32_t* overflow_rows, hipStream_t st, uint32_t* overflow_out) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const uint32_t checked_words = f->algorithm == 1 ? (f->num_words + 31) / 32 : 0;
  const uint32_t list_cap = f->algorithm == 1 ? std::min<uint32_t>(f->num_words, (uint32_t)std::min<int64_t>((int64_t)f->num_checks + 64, 1 << 20)) : 0;
  const uint64_t per_lane = (uint64_t)heap_cap * 8 + (uint64_t)checked_words * 4 + (uint64_t)list_cap * 4 + (uint64_t)std::max(f->branching, 1) * 4;
  // resident lanes: 16 waves per CU (the kernel's 8 KB of LDS and ~120 VGPRs allow about that), fewer when the per-lane scratch
  // would pass 8 GiB (a large vocabulary's bitset, the second pass's deep heaps)
  uint64_t blocks = std::min<uint64_t>((n_items + 63) / 64, (uint64_t)cus * 16);
  while (blocks > 1 && blocks * 64 * per_lane > (8ull << 30)) blocks = (blocks + 1) / 2;
  const uint32_t n_lanes = (uint32_t)blocks * 64u;
  FCHK(ctx, f->heap.reserve((size_t)heap_cap * n_lanes * 8));
  if (checked_words) {
    const size_t bytes = (size_t)checked_words * n_lanes * 4;
    const bool fresh = f->checked.cap < bytes || f->n_lanes != n_lanes;
    FCHK(ctx, f->checked.reserve(bytes));
    if (fresh) FCHK(ctx, hipMemsetAsync(f->checked.p, 0, f->checked.cap, st));  // every query leaves its lane's bits cleared again
    FCHK(ctx, f->clist.reserve((size_t)std::max<uint32_t>(list_cap, 1) * n_lanes * 4));
  }
  FCHK(ctx, f->domain.reserve((size_t)std::max(f->branching, 1) * n_lanes * 4));
  FCHK(ctx, f->overflow.reserve(4));
  FCHK(ctx, hipMemsetAsync(f->overflow.p, 0, 4, st));
  f->n_lanes = n_lanes;
  FlannSearchParams p;
  p.algorithm = f->algorithm;
  p.num_checks = f->num_checks;
  p.num_words = f->num_words;
  p.branching = f->branching;
  p.cb_index = f->cb_index;
  p.km_root = f->km_root;
  p.kd_nodes = f->kd_nodes.as<dsm_flann_kd_node>();
  p.kd_roots = f->kd_roots.as<int32_t>();
  p.n_kd_roots = f->n_kd_roots;
  p.km_nodes = f->km_nodes.as<FlannKmNodeDev>();
  p.km_childs = f->km_childs.as<int32_t>();
  p.km_points = f->km_points.as<uint32_t>();
  p.pivots = f->pivots.as<float>();
  p.words = d_words_s8;
  p.wnorm = f->wnorm.as<int32_t>();
  p.desc = desc;
  p.row_img = row_img;
  p.n_rows = n_items;
  p.k = k;
  p.heap = f->heap.as<uint2>();
  p.heap_cap = heap_cap;
  p.checked = f->checked.as<uint32_t>();
  p.checked_words = checked_words;
  p.checked_list = f->clist.as<uint32_t>();
  p.list_cap = list_cap;
  p.domain = f->domain.as<float>();
  p.n_lanes = n_lanes;
  p.out_ids = out_ids;
  p.out_dists = out_dists;
  p.out_stride = out_stride;
  p.overflow = f->overflow.as<uint32_t>();
  p.overflow_rows = overflow_rows;
  p.row_list = row_list;
  FCHK(ctx, hipEventRecord(f->ev0, st));  // (after the allocations above: the events bracket the kernel, not hipMalloc)
  if (f->algorithm == 0) hipLaunchKernelGGL(k_flann_search<0>, dim3((uint32_t)blocks), dim3(64), 0, st, p);
  if (f->algorithm == 1) hipLaunchKernelGGL(k_flann_search<1>, dim3((uint32_t)blocks), dim3(64), 0, st, p);
  if (f->algorithm == 2) hipLaunchKernelGGL(k_flann_search<2>, dim3((uint32_t)blocks), dim3(64), 0, st, p);
  FCHK(ctx, hipGetLastError());
  FCHK(ctx, hipEventRecord(f->ev1, st));
  FCHK(ctx, hipMemcpyAsync(overflow_out, f->overflow.p, 4, hipMemcpyDeviceToHost, st));
  FCHK(ctx, hipStreamSynchronize(st));
  float ms = 0.f;
  FCHK(ctx, hipEventElapsedTime(&ms, f->ev0, f->ev1));
  f->last_ms += ms;
  return DSM_OK;
}

#ifdef DSM_CHECK_BUILD
// check build, DSM_FLANN_GROUP: the kd-trees' first pass as a 16-lane group per query (k_flann_search_kd_grp; measured slower)
static int flann_launch_kd_grp(dsm_ctx* ctx, FlannDevice* f, const int8_t* d_words_s8, const int8_t* desc, const int32_t* row_img, uint64_t n_items,
                               uint32_t k, int32_t* out_ids, float* out_dists, uint32_t out_stride, uint32_t* overflow_rows, hipStream_t st,
                               uint32_t* overflow_out) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const uint32_t per_wave = 64u / FLANN_GRP;
  const uint32_t checked_words = (f->num_words + 31) / 32;
  const uint32_t list_cap = std::min<uint32_t>(f->num_words, (uint32_t)std::min<int64_t>((int64_t)f->num_checks + 64, 1 << 20));
  const uint32_t heap_cap = std::min<uint32_t>(f->num_words, FLANN_HEAP_CAP);
  // resident waves: 24 per CU (78 VGPRs), fewer when the heaps and bitsets would pass 4 GiB
  uint64_t blocks = std::min<uint64_t>((n_items + per_wave - 1) / per_wave, (uint64_t)cus * 24);
  while (blocks > 1 && blocks * per_wave * (((uint64_t)checked_words + list_cap) * 4 + (uint64_t)heap_cap * 8) > (4ull << 30)) blocks = (blocks + 1) / 2;
  const uint32_t n_groups = (uint32_t)blocks * per_wave;
  FCHK(ctx, f->heap.reserve((size_t)heap_cap * n_groups * 8));
  {
    const size_t bytes = (size_t)checked_words * n_groups * 4;
    const bool fresh = f->checked_g.cap < bytes || f->n_groups != n_groups;
    FCHK(ctx, f->checked_g.reserve(bytes));
    if (fresh) FCHK(ctx, hipMemsetAsync(f->checked_g.p, 0, f->checked_g.cap, st));  // every query leaves its group's bits cleared again
    FCHK(ctx, f->clist_g.reserve((size_t)std::max<uint32_t>(list_cap, 1) * n_groups * 4));
    f->n_groups = n_groups;
  }
  FCHK(ctx, f->overflow.reserve(4));
  FCHK(ctx, hipMemsetAsync(f->overflow.p, 0, 4, st));
  FlannSearchParams p = FlannSearchParams();
  p.algorithm = 1;
  p.num_checks = f->num_checks;
  p.num_words = f->num_words;
  p.kd_nodes = f->kd_nodes.as<dsm_flann_kd_node>();
  p.kd_roots = f->kd_roots.as<int32_t>();
  p.n_kd_roots = f->n_kd_roots;
  p.words = d_words_s8;
  p.wnorm = f->wnorm.as<int32_t>();
  p.desc = desc;
  p.row_img = row_img;
  p.n_rows = n_items;
  p.k = k;
  p.heap = f->heap.as<uint2>();
  p.heap_cap = heap_cap;
  p.checked = f->checked_g.as<uint32_t>();
  p.checked_words = checked_words;
  p.checked_list = f->clist_g.as<uint32_t>();
  p.list_cap = list_cap;
  p.out_ids = out_ids;
  p.out_dists = out_dists;
  p.out_stride = out_stride;
  p.overflow = f->overflow.as<uint32_t>();
  p.overflow_rows = overflow_rows;
  p.row_list = nullptr;
  p.stats = nullptr;
#ifdef DSM_CHECK_BUILD
  if (ctx->dbg("DSM_FLANN_STATS")) {
    FCHK(ctx, f->stats.reserve(64));
    FCHK(ctx, hipMemsetAsync(f->stats.p, 0, 64, st));
    p.stats = f->stats.as<uint32_t>();
  }
#endif
  FCHK(ctx, hipEventRecord(f->ev0, st));
  hipLaunchKernelGGL(k_flann_search_kd_grp, dim3((uint32_t)blocks), dim3(64), 0, st, p);
  FCHK(ctx, hipGetLastError());
  FCHK(ctx, hipEventRecord(f->ev1, st));
  FCHK(ctx, hipMemcpyAsync(overflow_out, f->overflow.p, 4, hipMemcpyDeviceToHost, st));
  FCHK(ctx, hipStreamSynchronize(st));
  float ms = 0.f;
  FCHK(ctx, hipEventElapsedTime(&ms, f->ev0, f->ev1));
  f->last_ms += ms;
#ifdef DSM_CHECK_BUILD
  if (p.stats) {
    uint32_t h[16];
    FCHK(ctx, hipMemcpy(h, f->stats.p, 64, hipMemcpyDeviceToHost));
    fprintf(stderr, "[flann kd grp] queries %u  nodes/query %.1f  leaves/query %.1f  pops/query %.1f  pushes/query %.1f  max heap %u  heap > 32/64/128/256/384/511: %u %u %u %u %u %u  overflow %u  %.1f ms\n",
            h[0], (double)h[1] / std::max(h[0], 1u), (double)h[2] / std::max(h[0], 1u), (double)h[3] / std::max(h[0], 1u), (double)h[4] / std::max(h[0], 1u), h[5], h[6], h[7],
            h[8], h[9], h[10], h[11], *overflow_out, ms);
  }
#endif
  return DSM_OK;
}

#endif

// Searches the s8 rows `desc` (padding rows: row_img < 0) for their k words: out_ids [rows][out_stride] on the device.
// Two passes at most: all queries with FLANN_HEAP_CAP heap entries per lane at full residency; the queries whose heap needed
// more (deep, unbalanced trees push one branch per level and descent) once more with FLANN's own capacity (num_words, at most
// FLANN_HEAP_CAP_RETRY) on as many lanes as that leaves room for.
int flann_device_search(dsm_ctx* ctx, FlannDevice* f, const int8_t* d_words_s8, const int8_t* desc, const int32_t* row_img, uint64_t n_rows,
                        uint32_t k, int32_t* out_ids, float* out_dists, uint32_t out_stride, hipStream_t st) {
  if (!f || f->algorithm < 0) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_flann_index has not run");
  if (k == 0 || k > FLANN_K_MAX || out_stride < k || out_stride > FLANN_K_MAX) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "FLANN search: 1 <= k <= 8");
  if (n_rows == 0) return DSM_OK;
  if (n_rows > 0xfffffff0ull) return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "FLANN search: too many rows");
  FCHK(ctx, f->overflow_rows.reserve((size_t)n_rows * 4));
  uint32_t overflow = 0;
  f->last_ms = 0.0;
  int rc;
  const uint32_t* todo = nullptr;  // the rows still to search (nullptr: all of them)
  uint64_t n_todo = n_rows;
#ifdef DSM_CHECK_BUILD
  if (f->algorithm == 1 && ctx->dbg("DSM_FLANN_GROUP") != nullptr) {
    // the A/B: kd-trees as a 16-lane group per query; the queries whose heap needed more than the first pass's capacity go on to the
    // lane-per-query kernel below
    rc = flann_launch_kd_grp(ctx, f, d_words_s8, desc, row_img, n_rows, k, out_ids, out_dists, out_stride, f->overflow_rows.as<uint32_t>(), st, &overflow);
    if (rc != DSM_OK) return rc;
    f->last_retried = overflow;
    if (!overflow) return DSM_OK;
    FCHK(ctx, f->overflow_rows2.reserve((size_t)overflow * 4));
    FCHK(ctx, hipMemcpyAsync(f->overflow_rows2.p, f->overflow_rows.p, (size_t)overflow * 4, hipMemcpyDeviceToDevice, st));
    todo = f->overflow_rows2.as<uint32_t>();
    n_todo = overflow;
    overflow = 0;
  }
#endif
  rc = flann_launch(ctx, f, d_words_s8, desc, row_img, n_todo, todo, k, out_ids, out_dists, out_stride, std::min<uint32_t>(f->num_words, FLANN_HEAP_CAP),
                    f->overflow_rows.as<uint32_t>(), st, &overflow);
  if (rc != DSM_OK) return rc;
  if (!todo) f->last_retried = overflow;
  if (overflow) {
    const uint32_t n_retry = overflow;
    rc = flann_launch(ctx, f, d_words_s8, desc, row_img, n_retry, f->overflow_rows.as<uint32_t>(), k, out_ids, out_dists, out_stride,
                      std::min<uint32_t>(f->num_words, FLANN_HEAP_CAP_RETRY), nullptr, st, &overflow);
    if (rc != DSM_OK) return rc;
  }
  if (overflow)
    return dsm_fail(ctx, DSM_ERR_OUT_OF_RANGE, "FLANN search: a query's branch heap passed 2^20 entries on a vocabulary of more words than that: search it on the host (word_search = flann_host)");
  return DSM_OK;
}

// Host queries (uint8 descriptors) through the same kernel: the test / bench entry point behind dsm_retrieval_flann_search.
int flann_device_search_host(dsm_ctx* ctx, FlannDevice* f, const int8_t* d_words_s8, const uint8_t* queries, uint32_t n, uint32_t k, int32_t* ids,
                             float* dists) {
  if (!f) return dsm_fail(ctx, DSM_ERR_NOT_READY, "dsm_retrieval_set_flann_index has not run");
  if (n == 0) return DSM_OK;
  if (!queries || !ids) return dsm_fail(ctx, DSM_ERR_INVALID_ARGUMENT, "null queries / ids");
  FCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  FCHK(ctx, f->q_u8.reserve((size_t)n * 128));
  FCHK(ctx, f->q_s8.reserve((size_t)n * 128));
  FCHK(ctx, f->out_ids.reserve((size_t)n * k * 4));
  FCHK(ctx, f->out_dists.reserve((size_t)n * k * 4));
  FCHK(ctx, hipMemcpyAsync(f->q_u8.p, queries, (size_t)n * 128, hipMemcpyHostToDevice, st));
  const uint64_t words = (uint64_t)n * 32;
  hipLaunchKernelGGL(k_flann_u8_to_s8, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, st, f->q_u8.as<uint32_t>(), f->q_s8.as<uint32_t>(), words);
  FCHK(ctx, hipGetLastError());
  const int rc = flann_device_search(ctx, f, d_words_s8, f->q_s8.as<int8_t>(), nullptr, n, k, f->out_ids.as<int32_t>(), f->out_dists.as<float>(), k, st);
  if (rc != DSM_OK) return rc;
  FCHK(ctx, hipMemcpy(ids, f->out_ids.p, (size_t)n * k * 4, hipMemcpyDeviceToHost));
  if (dists) FCHK(ctx, hipMemcpy(dists, f->out_dists.p, (size_t)n * k * 4, hipMemcpyDeviceToHost));
  return DSM_OK;
}

double flann_device_last_ms(const FlannDevice* f) { return f ? f->last_ms : 0.0; }
int flann_device_algorithm(const FlannDevice* f) { return f ? f->algorithm : -1; }
