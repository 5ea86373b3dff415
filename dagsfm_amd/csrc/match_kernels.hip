// match_kernels.hip -- brute-force SIFT descriptor matching on MI355X (gfx950).
//
// Replaces, bit-exactly, the reference CPU matcher
//   ComputeSiftDistanceMatrix   /root/reference/src/feature/sift.cc:76-109
//   FindBestMatchesOneWay       /root/reference/src/feature/sift.cc:111-162
//   FindBestMatches             /root/reference/src/feature/sift.cc:164-198
// without ever materialising the N1 x N2 int32 matrix.
//
// K0  k0_prepare        u8 -> s8 (x ^ 0x80) + per-row bias term, once per image upload.
// K1  k1_best_rows      one *directed* pass (image a rows vs image b columns): int8 MFMA 32x32x32
//                       distance tiles whose accumulators start at the column bias, top-2 VALUES
//                       per row fused in the epilogue (3 VALU per 2 elements: v_med3, v_max3,
//                       v_max) plus the 32-column tile in which the best value was reached;
//                       thresholds (acos LUT, ratio) applied before the single int32 store.
//                       Cross-check: FindBestMatches only ever reads matches21[matches12[i1]]
//                       (sift.cc:178-186), so the second pass (GATHER) computes
//                       FindBestMatchesOneWay(dists.transpose()) for exactly those rows of image b --
//                       the one-way matches of pass 1, gathered through an entry list -- against
//                       all columns of image a: ~N/14 rows instead of N at the benchmark shape.
// K1b k1_resolve_index  for the rows that passed the thresholds only: the lowest column of that
//                       tile that attains the best value (32 dot products per row, v_dot4).
// K2  k2_cross_compact  mutual check + ordered compaction (ascending idx1) per pair.
//
// Exactness of the signed-MFMA trick: a = a' + 128, b = b' + 128,
//   dot = S + rterm(i) + rterm(j) + 2^21,  S = sum a'b',  rterm(x) = 128 * sum x'
// all in int32 (|dot| <= 128*255^2 < 2^23).  Within a row only  v = S + rterm(j) + 2^21  matters
// (dot = v + rterm(i)); the MFMA chain of a tile starts from C = rterm(j) + 2^21, so the finished
// accumulator IS v and needs no further arithmetic before the comparisons.
//
// The tile is computed transposed (descriptor columns of image b as the MFMA A operand): a lane
// owns ONE row of image a and its 16 accumulator registers are 16 different columns, so the
// running (best, second) of a row are two registers and two new values x, y update them with
//   second = max(second, med3(best, x, y));  best = max3(best, x, y)
// (the second largest of {best, second, x, y} given second <= best).  That is the reference's
// second-best semantics on values: a duplicate of the best value is the second best
// (sift.cc:126-132).  The reference's best_i2 is the LOWEST column that attains the maximum
// (ascending strict-`>` scan): K1 remembers the first tile in which the final best value was
// reached (a strict increase of the running best), K1b finds the first column inside that tile.
// Zero padding rows/columns have dot == 0 and can never beat the initial
// best_dist = second_best_dist = 0 (sift.cc:122-123).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>());
    static_for<B + 1, E>(f);
  }
}

// ------------------------------------------------------------------------------------ K0
// 8 lanes per descriptor row, 16 bytes per lane.
__global__ __launch_bounds__(256) void k0_prepare(const uint8_t* __restrict__ in_u8,
                                                  int8_t* __restrict__ out_s8,
                                                  int32_t* __restrict__ rterm,
                                                  uint64_t n_rows) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t row = gid >> 3;
  const uint32_t chunk = (uint32_t)gid & 7u;
  if (row >= n_rows) return;  // whole 8-lane groups leave together
  const uint4 v = reinterpret_cast<const uint4*>(in_u8)[row * 8 + chunk];
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  int32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sum += (int32_t)((w[k] & 0xffu) + ((w[k] >> 8) & 0xffu) + ((w[k] >> 16) & 0xffu) + (w[k] >> 24));
    w[k] ^= 0x80808080u;
  }
  uint4 o;
  o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
  reinterpret_cast<uint4*>(out_s8)[row * 8 + chunk] = o;
  sum += __shfl_xor(sum, 1);
  sum += __shfl_xor(sum, 2);
  sum += __shfl_xor(sum, 4);
  if (chunk == 0) rterm[row] = 128 * (sum - 16384);
}

// ------------------------------------------------------------------------------------ K1
// The top-2 update of a finished 32x32 tile for the lane's row (its 16 accumulator registers = 16 columns), in
// four pieces of three VALU instructions, each ONE asm block so that it can be placed between two MFMAs as a unit:
//   t = max of the 16 values (v_max3 chain);  u = max(t, second);  second = min(u, best);  best = max(u, best)
// i.e. (best, second) become the two largest of {best, second, t}: the running pair is the top-2 of the TILE MAXIMA
// of the lane's 16-column sets.  `best` is exact.  `second` misses exactly one candidate -- the second largest value
// INSIDE the set that holds the best -- and k1_resolve_index, which re-reads that set anyway to find the column,
// adds it (rows are flagged with this `second`, which is <= the true one: a superset of the rows that pass the ratio
// test, sift.cc:148-155; the final test runs there).  12 VALU per tile where the exact top-2 tree needed 22: the
// VALU issue port, not the matrix pipe, was what limited this kernel (2 waves per SIMD x (4 MFMA + 22 VALU) issue
// slots per 2 x 128 cycles of matrix-pipe time).
// Piece 3 records the tile index when the best strictly increased during the tile (the reference's best_i2 is the
// lowest column attaining the maximum).
template <int Q>
__device__ __forceinline__ void k1_epilogue_quarter(const v16i& a, int& best, int& second, int& t, int& btile, int tile) {
  if constexpr (Q == 0) {
    asm volatile(
        "v_max3_i32 %[t], %[a0], %[a1], %[a2]\n\t"
        "v_max3_i32 %[t], %[t], %[a3], %[a4]\n\t"
        "v_max3_i32 %[t], %[t], %[a5], %[a6]"
        : [t] "=&v"(t)
        : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]));
  } else if constexpr (Q == 1) {
    asm volatile(
        "v_max3_i32 %[t], %[t], %[a0], %[a1]\n\t"
        "v_max3_i32 %[t], %[t], %[a2], %[a3]\n\t"
        "v_max3_i32 %[t], %[t], %[a4], %[a5]"
        : [t] "+v"(t)
        : [a0] "v"(a[7]), [a1] "v"(a[8]), [a2] "v"(a[9]), [a3] "v"(a[10]), [a4] "v"(a[11]), [a5] "v"(a[12]));
  } else if constexpr (Q == 2) {
    asm volatile(
        "v_max3_i32 %[t], %[t], %[a0], %[a1]\n\t"
        "v_max3_i32 %[t], %[t], %[a2], %[s]\n\t"
        "v_min_i32 %[s], %[t], %[b]"
        : [t] "+v"(t), [s] "+v"(second)
        : [b] "v"(best), [a0] "v"(a[13]), [a1] "v"(a[14]), [a2] "v"(a[15]));
  } else {
    asm volatile(
        "v_cmp_gt_i32 vcc, %[t], %[b]\n\t"
        "v_cndmask_b32 %[bt], %[bt], %[tile], vcc\n\t"
        "v_max_i32 %[b], %[t], %[b]"
        : [b] "+v"(best), [bt] "+v"(btile)
        : [t] "v"(t), [tile] "v"(tile)
        : "vcc");
  }
}

#ifndef K1_LDS_DMA
#define K1_LDS_DMA 0  // 1: B tiles staged by global_load_lds_dwordx4 (A/B'd in round 4)
#endif
// LDS image of a 64-column B tile: column c occupies 128 B; its eight 16-B chunks are
// XOR-swizzled with (c>>1)&7 so that the 16-lane ds_read_b128 groups are conflict free.
__device__ __forceinline__ int lds_off(int col, int chunk) {
  return col * 128 + ((chunk ^ ((col >> 1) & 7)) << 4);
}

// Workgroup = 4 waves x 128 rows of image a (4 resident 32-row fragments per wave); the columns of
// image b stream through LDS in 64-column steps (double buffered) together with their bias terms.
// GATHER: the rows are not an image's rows but the entries [e_off[d], e_off[d] + e_cnt[d]) of the entry list, entry
// k standing for row entries[k].y of image ab.x (pass 2 of the cross-check); out is indexed like the entry list.
template <bool GATHER>
__global__ __launch_bounds__(256, 2) void k1_best_rows(const K1Params p) {
  const uint32_t d = p.order ? p.order[blockIdx.x] : blockIdx.x;
  const uint32_t rb = blockIdx.y;
  const uint2 ab = p.dpairs[d];
  const uint32_t a_rows = GATHER ? p.e_cnt[d] : p.img_rows[ab.x];
  if (rb * 512u >= a_rows) return;
  const uint32_t b_cols = p.img_rows[ab.y];
  const uint32_t a_row0 = (GATHER ? 0u : p.img_row0[ab.x]) + rb * 512u;  // GATHER: first entry of this block
  const uint32_t b_row0 = p.img_row0[ab.y];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  // images are padded to 256 rows: the upper two waves of the last workgroup may have no rows at all
  const bool active = rb * 512u + (uint32_t)wave * 128u < a_rows;

  __shared__ __attribute__((aligned(16))) int8_t sB[2][64 * 128];
  __shared__ __attribute__((aligned(16))) int sC[2][64];

  // Resident fragments of this wave's 128 rows (MFMA B operand: lane = row, 16 B of k per half).
  v4i afrag[4][4];
  int rterm_i[4];
  bool valid[4];  // GATHER: this lane's row of fragment rt is an entry (the last fragments of a pair are ragged)
  {
    const uint2* ent = GATHER ? p.entries + p.e_off[d] : nullptr;
    const uint32_t img0 = GATHER ? p.img_row0[ab.x] : 0u;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const uint32_t r = a_row0 + wave * 128 + rt * 32 + l31;  // row of the image / entry of the pair
      valid[rt] = GATHER ? (r < a_rows) : active;
      uint32_t grow = r;
      if (GATHER) grow = valid[rt] ? img0 + ent[r].y : 0u;
      const int8_t* arow = p.desc + (size_t)grow * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const v4i z = {0, 0, 0, 0};
        afrag[rt][ks] = valid[rt] ? *reinterpret_cast<const v4i*>(arow + ks * 32 + half * 16) : z;
      }
      rterm_i[rt] = valid[rt] ? p.rterm[grow] : 0;
    }
  }
  // dot == 0  <=>  v == -rterm(i): best_dist = second_best_dist = 0 initially (sift.cc:122-123)
  int best[4], second[4], btile[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    best[rt] = -rterm_i[rt];
    second[rt] = -rterm_i[rt];
    btile[rt] = 0;
  }

  const int8_t* bimg = p.desc + (size_t)b_row0 * 128;
  const int32_t* rt_b = p.rterm + b_row0;
  const uint32_t nsteps = b_cols >> 6;

  int cstage = 0;
#if K1_LDS_DMA
  // B tiles go global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  The DMA writes a wave's
  // 64 x 16 B linearly at a wave-uniform LDS base, so the XOR swizzle of lds_off moves to the SOURCE: LDS slot L (column
  // L >> 3, chunk position L & 7) receives the column's chunk (L & 7) ^ ((L >> 4) & 7) -- the same involution the reads
  // apply; the eight chunks of a column are one 128-B line, so the permutation costs no coalescing.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int src_off[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int L = tid + 256 * u;
    src_off[u] = ((L & ~7) | ((L & 7) ^ ((L >> 4) & 7))) * 16;
  }
  auto stage_tile = [&](const int8_t* src, int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + src_off[u]),
                                       (__attribute__((address_space(3))) void*)(&sB[buf][(wave_u * 64 + 256 * u) * 16]), 16, 0, 0);
  };
  stage_tile(bimg, 0);
  cstage = rt_b[tid & 63];
  if (tid < 64) sC[0][tid] = cstage + (1 << 21);
  __syncthreads();  // (its fence waits for the DMA: vmcnt(0))
#else
  v4i stage[2];
  // prologue: step 0
#pragma unroll
  for (int u = 0; u < 2; ++u) stage[u] = *reinterpret_cast<const v4i*>(bimg + (size_t)(tid + 256 * u) * 16);
  cstage = rt_b[tid & 63];  // every wave loads it: no divergent branch around a global load
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = tid + 256 * u;
    *reinterpret_cast<v4i*>(&sB[0][lds_off(q >> 3, q & 7)]) = stage[u];
  }
  if (tid < 64) sC[0][tid] = cstage + (1 << 21);
  __syncthreads();
#endif
  // make sure nothing issued before the loop is still pending at its head: hipcc's waitcnt pass would
  // otherwise keep a conservative vmcnt wait at the top of every iteration (right behind the prefetch)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (uint32_t s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const bool more = (s + 1) < nsteps;
    // prefetch of the next B step (global -> registers); consumed at the end of the step
    if (more) {
      const int8_t* src = bimg + (size_t)(s + 1) * 64 * 128;
#if K1_LDS_DMA
      stage_tile(src, cur ^ 1);  // the buffer of the previous step: every wave left it at the barrier that ended that step
#else
#pragma unroll
      for (int u = 0; u < 2; ++u) stage[u] = *reinterpret_cast<const v4i*>(src + (size_t)(tid + 256 * u) * 16);
#endif
      cstage = rt_b[(s + 1) * 64 + (tid & 63)];
    }
    if (active) {
      // MFMA A operand: lane = column of the tile; C input: register r <-> column 8*(r>>2) + 4*half + (r&3)
      v4i bf0[4], bf1[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf0[ks] = *reinterpret_cast<const v4i*>(&sB[cur][lds_off(l31, ks * 2 + half)]);
        bf1[ks] = *reinterpret_cast<const v4i*>(&sB[cur][lds_off(32 + l31, ks * 2 + half)]);
      }
      v16i ci0, ci1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4i c0 = *reinterpret_cast<const v4i*>(&sC[cur][8 * q + 4 * half]);
        const v4i c1 = *reinterpret_cast<const v4i*>(&sC[cur][32 + 8 * q + 4 * half]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ci0[4 * q + e] = c0[e];
          ci1[4 * q + e] = c1[e];
        }
      }
      // Software pipeline over the eight 32x32 tiles of the step (tile k = column tile k>>2, row
      // fragment k&3, accumulator set k%3).  Slot j issues one MFMA of tile (j>>2)+1 and, one slot behind,
      // one epilogue quarter of tile j>>2: the VALU works on finished accumulators while the matrix pipe
      // runs the next chain, and no epilogue reads an accumulator right behind the MFMA that writes it.
      v16i acc[3];
      int tmax;  // the tile maximum in the making (one tile's epilogue is in flight at a time)
      int tile_v[2];  // the step's two tile indices, one VGPR each (v_cndmask takes no second SGPR)
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(tile_v[0]), "=v"(tile_v[1]) : "s"(2 * s), "s"(2 * s + 1));
#define K1_MFMA(K, KS)                                                                                 \
  acc[(K) % 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(((K) >> 2) ? bf1[KS] : bf0[KS], afrag[(K)&3][KS], \
                                                       (KS) == 0 ? (((K) >> 2) ? ci1 : ci0) : acc[(K) % 3], 0, 0, 0)
#define K1_EPI(E)                                                                                        \
  {                                                                                                      \
    constexpr int ek = (E) >> 2, eq = (E)&3, ert = ek & 3;                                               \
    k1_epilogue_quarter<eq>(acc[ek % 3], best[ert], second[ert], tmax, btile[ert], tile_v[ek >> 2]);          \
  }
      static_for<0, 4>([&](auto KS) { K1_MFMA(0, KS.value); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 28>([&](auto J) {
        constexpr int j = J.value;
        K1_MFMA((j >> 2) + 1, j & 3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j >= 1) {
          K1_EPI(j - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      static_for<27, 32>([&](auto E) { K1_EPI(E.value); });
      __builtin_amdgcn_sched_barrier(0);
#undef K1_MFMA
#undef K1_EPI
    }
    if (more) {
#if !K1_LDS_DMA
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = tid + 256 * u;
        *reinterpret_cast<v4i*>(&sB[cur ^ 1][lds_off(q >> 3, q & 7)]) = stage[u];
      }
#endif
      if (tid < 64) sC[cur ^ 1][tid] = cstage + (1 << 21);
    }
    __syncthreads();
  }

  if (!active) return;
  const uint64_t out0 = (GATHER ? p.e_off[d] : p.d_out_off[d]) + rb * 512u + wave * 128;
  int32_t* out = p.out + out0;
  int32_t* out_s = p.out_s + out0;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    // the two halves of the wave hold disjoint 16-column sets of the same row
    const int oB = __shfl_xor(best[rt], 32);
    const int oS = __shfl_xor(second[rt], 32);
    const int oT = __shfl_xor(btile[rt], 32);
    const int B = max(best[rt], oB);
    const int S = max(min(best[rt], oB), max(second[rt], oS));
    // where the lowest column with the best value lives: a tile, and the half-wave's 16-column set of it (2 = both
    // sets of the same tile reach it: the lowest column could be in either)
    const bool take = (oB > best[rt]) || (oB == best[rt] && oT < btile[rt]);
    const bool both = oB == best[rt] && oT == btile[rt];
    const int T = take ? oT : btile[rt];
    const int set = both ? 2 : ((take ? 1 : 0) ^ half);
    const int best_dot = B + rterm_i[rt];
    const int second_dot = S + rterm_i[rt];  // <= the true second best (see k1_epilogue_quarter)
    int res = -1;
    if (best_dot > 0) {  // best_i2 != -1, sift.cc:136
      const float bn = p.lut[min(best_dot, 262144)];
      if (!(bn > p.max_distance)) {  // sift.cc:144
        const float sn = p.lut[min(second_dot, 262144)];
        const float rhs = __fmul_rn(p.max_ratio, sn);
        if (!(bn >= rhs)) res = T * 4 + set;  // sift.cc:153 with second <= true second; K1b completes the test
      }
    }
    if (half == (rt & 1) && valid[rt]) {
      out[rt * 32 + l31] = res;
      if (res >= 0) out_s[rt * 32 + l31] = second_dot;
    }
  }
}

// ------------------------------------------------------------------------------------ K1b
// out[row] = tile * 4 + set for the rows K1 flagged (best value exact, `second` possibly too small): the best value
// was first reached in that 32-column tile, in the 16-column set of half-wave `set` (columns 8q + 4*set + e; set 2 =
// both).  This kernel re-reads those columns -- 2 KB instead of the 4 KB tile -- and
//   * finds the LOWEST column that attains the best value (the reference's best_i2, strict `>` ascending scan),
//   * finds the second largest value inside the set, the one candidate K1's `second` lacks, and
//   * applies the ratio test of sift.cc:148-155 with the completed second best: column index or -1.
// One workgroup per directed pair (the columns it reads all belong to ONE image b, which stays in L2); a wave takes
// 64 rows at a time and resolves its flagged rows four at a time.
template <bool GATHER>
__global__ __launch_bounds__(256) void k1_resolve_index(const K1Params p) {
  const uint32_t d = blockIdx.x;
  const uint2 ab = p.dpairs[d];
  const uint32_t a_rows = GATHER ? p.e_cnt[d] : p.img_rows[ab.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t b_row0 = p.img_row0[ab.y];
  const uint32_t a_img0 = p.img_row0[ab.x];
  const int8_t* bimg = p.desc + (size_t)b_row0 * 128;
  const int32_t* rt_b = p.rterm + b_row0;
  const uint2* ent = GATHER ? p.entries + p.e_off[d] : nullptr;
  const int g = lane >> 3;  // 8 lanes share one dot product: column group g, 16-byte chunk lane & 7
  for (uint32_t row_w = wave * 64u; row_w < a_rows; row_w += 256u) {  // first row of this wave's chunk
    const uint64_t o0 = (GATHER ? p.e_off[d] : p.d_out_off[d]) + row_w;
    int32_t* out = p.out + o0;
    const int32_t* out_s = p.out_s + o0;
    const bool in_range = !GATHER || row_w + (uint32_t)lane < a_rows;
    const int t = in_range ? out[lane] : -1;
    const int s_appr = t >= 0 ? out_s[lane] : 0;
    const uint32_t my_row = GATHER ? (in_range ? ent[row_w + lane].y : 0u) : row_w + (uint32_t)lane;  // row inside image a
    unsigned long long mask = __ballot(t >= 0);
    // four flagged rows per trip: their loads (row descriptor, column sets, column terms) are issued together, so a
    // trip pays one memory round trip instead of four (a wave resolves ~100 rows one after the other)
    constexpr int RU = 4;
    while (mask) {
      int r_l[RU], tile_l[RU], set_l[RU], sap_l[RU];
      uint32_t grow_l[RU];
      bool on[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        on[u] = mask != 0ull;
        const int r = on[u] ? (__ffsll((long long)mask) - 1) : 0;
        if (on[u]) mask &= mask - 1;
        r_l[u] = r;
        const int code = on[u] ? __builtin_amdgcn_readlane(t, r) : 0;
        tile_l[u] = code >> 2;
        set_l[u] = code & 3;
        sap_l[u] = __builtin_amdgcn_readlane(s_appr, r);
        grow_l[u] = a_img0 + (uint32_t)__builtin_amdgcn_readlane((int)my_row, r);
      }
      // a set's 16 columns are four runs of four consecutive columns (512 B each): load i covers the eight columns
      // 8*(2i + (g>>2)) + 4*set + (g&3); both sets (a tie between the half-waves): four loads of eight consecutive columns
      v4i x[RU], y[RU][4];
      int ct[RU][4], col[RU][4], rti[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        x[u] = *reinterpret_cast<const v4i*>(p.desc + (size_t)grow_l[u] * 128 + (lane & 7) * 16);
        rti[u] = p.rterm[grow_l[u]];
        const bool both = set_l[u] == 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          col[u][i] = both ? i * 8 + g : 8 * (2 * (i & 1) + (g >> 2)) + 4 * set_l[u] + (g & 3);
          if (i < 2 || both) {  // wave-uniform
            const int c = tile_l[u] * 32 + col[u][i];
            y[u][i] = *reinterpret_cast<const v4i*>(bimg + (size_t)c * 128 + (lane & 7) * 16);
            ct[u][i] = rt_b[c];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const bool both = set_l[u] == 2;
        int k1 = INT32_MIN, k2 = INT32_MIN;  // the two largest keys (value << 5 | 31 - column: distinct per column)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < 2 || both) {
            int acc = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_sdot4(x[u][e], y[u][i][e], acc, false);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 4);
            const int key = (int)((uint32_t)(acc + ct[u][i]) << 5) | (31 - col[u][i]);  // |S + rterm(j)| <= 2^22
            k2 = max(k2, min(k1, key));
            k1 = max(k1, key);
          }
        }
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
          const int o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
          k2 = max(min(k1, o1), max(k2, o2));
          k1 = max(k1, o1);
        }
        if (lane == 0 && on[u]) {
          const int bias = rti[u] + (1 << 21);
          const int best_dot = (k1 >> 5) + bias;
          const int second_dot = max(sap_l[u], (k2 >> 5) + bias);
          const float bn = p.lut[min(best_dot, 262144)];
          const float sn = p.lut[min(second_dot, 262144)];
          const float rhs = __fmul_rn(p.max_ratio, sn);
          out[r_l[u]] = (bn >= rhs) ? -1 : tile_l[u] * 32 + (31 - (k1 & 31));  // sift.cc:153
        }
      }
    }
  }
}

#ifdef DSM_CHECK_BUILD  // comparison variant (BASELINE configs[2]): libdagsfm_mi355x_check.so only
// ------------------------------------------------------------------------------------ K1-dot4 (comparison variant)
// The LDS-tiled VALU form of pass 1 that BASELINE.json configs[1]/[2] name ("int8 LDS-tiled distance kernel" vs "MFMA
// int8 distance GEMM"): thread = one row of image a with its 128-byte descriptor in 32 registers, the columns of
// image b stream through LDS in 64-column tiles and every lane reads the same column (LDS broadcast), 32
// v_dot4_i32_i8 per element, then the reference's own scan (strict >, ascending columns: sift.cc:119-133), so the
// exact column index falls out directly and k1_resolve_index is not needed.  Kept behind DSM_K1_DOT4=1 for the
// rocprof comparison in profiles/ (it is ~5x slower than the MFMA kernel: 34 VALU per matrix element against
// 1.4); same results bit for bit.
__global__ __launch_bounds__(256) void k1_best_rows_dot4(const K1Params p) {
  const uint32_t d = p.order ? p.order[blockIdx.x] : blockIdx.x;
  const uint32_t rb = blockIdx.y;
  const uint2 ab = p.dpairs[d];
  const uint32_t a_rows = p.img_rows[ab.x];
  if (rb * 512u >= a_rows) return;
  const uint32_t b_cols = p.img_rows[ab.y];
  const uint32_t a_row0 = p.img_row0[ab.x], b_row0 = p.img_row0[ab.y];
  const int tid = threadIdx.x;
  // two rows per thread (register blocking: one LDS read of a column feeds 64 v_dot4 instead of 32, which moves the
  // kernel from LDS-bound to VALU-bound); images are padded to 256 rows, so the second row may not exist
  const uint32_t row[2] = {rb * 512u + tid, rb * 512u + 256u + tid};
  const bool has2 = row[1] < a_rows;
  __shared__ __attribute__((aligned(16))) int8_t sB[2][64 * 128];
  __shared__ int sR[2][64];
  v4i a[2][8];
  int rterm_i[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint32_t rr = (r == 0 || has2) ? row[r] : row[0];
    const int8_t* arow = p.desc + (size_t)(a_row0 + rr) * 128;
#pragma unroll
    for (int c = 0; c < 8; ++c) a[r][c] = *reinterpret_cast<const v4i*>(arow + c * 16);
    rterm_i[r] = p.rterm[a_row0 + rr];
  }
  int best[2] = {0, 0}, second[2] = {0, 0}, best_j[2] = {-1, -1};
  const uint32_t nsteps = b_cols >> 6;
  v4i st[2];
  int sr = 0;
  auto fetch = [&](uint32_t s) {  // global -> registers (consumed after the step's arithmetic)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      st[u] = *reinterpret_cast<const v4i*>(p.desc + (size_t)(b_row0 + s * 64) * 128 + (size_t)(tid + 256 * u) * 16);
    sr = p.rterm[b_row0 + s * 64 + (tid & 63)];
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<v4i*>(&sB[buf][(tid + 256 * u) * 16]) = st[u];
    if (tid < 64) sR[buf][tid] = sr;
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
        const v4i b = *reinterpret_cast<const v4i*>(&sB[cur][j * 128 + c * 16]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] = __builtin_amdgcn_sdot4(a[0][c][e], b[e], acc[0], false);
          acc[1] = __builtin_amdgcn_sdot4(a[1][c][e], b[e], acc[1], false);
        }
      }
      const int cterm = sR[cur][j] + (1 << 21);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int dist = acc[r] + rterm_i[r] + cterm;
        if (dist > best[r]) {  // sift.cc:126-132
          second[r] = best[r];
          best[r] = dist;
          best_j[r] = (int)(s * 64 + j);
        } else if (dist > second[r]) {
          second[r] = dist;
        }
      }
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r == 1 && !has2) break;
    int res = -1;
    if (best_j[r] >= 0) {
      const float bn = p.lut[min(best[r], 262144)];
      if (!(bn > p.max_distance)) {
        const float sn = p.lut[min(second[r], 262144)];
        const float rhs = __fmul_rn(p.max_ratio, sn);
        if (!(bn >= rhs)) res = best_j[r];
      }
    }
    p.out[p.d_out_off[d] + row[r]] = res;
  }
}

#endif  // DSM_CHECK_BUILD

// ------------------------------------------------------------------------------------ KG (guided matching)
// The guided filters of MatchGuidedSiftFeaturesCPU (sift.cc:838-866) in float, same operation order as
// oracle_guided_filter (oracle/sift_match.c): true = the keypoint pair violates the geometry.
__device__ __forceinline__ bool guided_filter(int mode, const float* M, float x1, float y1, float x2, float y2,
                                              float max_residual) {
  if (mode == 1) {
    const float Fx1_0 = (M[0] * x1 + M[1] * y1) + M[2];
    const float Fx1_1 = (M[3] * x1 + M[4] * y1) + M[5];
    const float Fx1_2 = (M[6] * x1 + M[7] * y1) + M[8];
    const float Ftx2_0 = (M[0] * x2 + M[3] * y2) + M[6];
    const float Ftx2_1 = (M[1] * x2 + M[4] * y2) + M[7];
    const float x2tFx1 = (x2 * Fx1_0 + y2 * Fx1_1) + Fx1_2;
    return x2tFx1 * x2tFx1 / (((Fx1_0 * Fx1_0 + Fx1_1 * Fx1_1) + Ftx2_0 * Ftx2_0) + Ftx2_1 * Ftx2_1) > max_residual;
  }
  const float Hp_0 = (M[0] * x1 + M[1] * y1) + M[2];
  const float Hp_1 = (M[3] * x1 + M[4] * y1) + M[5];
  const float Hp_2 = (M[6] * x1 + M[7] * y1) + M[8];
  const float d0 = Hp_0 / Hp_2 - x2, d1 = Hp_1 / Hp_2 - y2;
  return d0 * d0 + d1 * d1 > max_residual;
}

// One directed pass with the filter: thread = one row of image a, all columns of image b in ascending order
// (64-column tiles of descriptors + bias + keypoints through LDS, every lane reads the same column: broadcast),
// dists(i1, i2) = filtered ? 0 : dot -- the scan of FindBestMatchesOneWay itself (strict >, lowest index wins).
// An optional mode, per element ~70 instructions (32 v_dot4 + the float filter): no MFMA here, the filter
// dominates.
__global__ __launch_bounds__(256) void kg_best_rows(const KgParams p) {
  const uint32_t d = blockIdx.x;
  const uint32_t rb = blockIdx.y;
  const uint2 ab = p.dpairs[d];
  const uint32_t a_rows = p.img_rows[ab.x];
  if (rb * 256u >= a_rows) return;
  const uint32_t b_cols = p.img_rows[ab.y];
  const uint32_t a_feat = p.img_nfeat[ab.x], b_feat = p.img_nfeat[ab.y];
  const uint32_t a_row0 = p.img_row0[ab.x], b_row0 = p.img_row0[ab.y];
  const int tid = threadIdx.x;
  const uint32_t row = rb * 256u + tid;

  __shared__ __attribute__((aligned(16))) int8_t sB[64 * 128];
  __shared__ int sR[64];
  __shared__ float sX[64], sY[64];

  const float* gp = p.gparams + (size_t)d * 12;
  float M[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) M[k] = gp[k];
  const int mode = (int)gp[9];
  const bool swap = gp[10] != 0.0f;  // rows are image 2 of the pair

  v4i a[8];
  const int8_t* arow = p.desc + (size_t)(a_row0 + row) * 128;
#pragma unroll
  for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const v4i*>(arow + c * 16);
  const int rterm_i = p.rterm[a_row0 + row];
  float xr = 0.0f, yr = 0.0f;
  if (row < a_feat) {
    xr = (float)p.kp[(size_t)(a_row0 + row) * 2];
    yr = (float)p.kp[(size_t)(a_row0 + row) * 2 + 1];
  }
  int best = 0, second = 0, best_j = -1;
  for (uint32_t j0 = 0; j0 < b_cols; j0 += 64) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + 256 * u;
      *reinterpret_cast<v4i*>(&sB[q * 16]) = *reinterpret_cast<const v4i*>(p.desc + (size_t)(b_row0 + j0) * 128 + (size_t)q * 16);
    }
    if (tid < 64) {
      const uint32_t col = j0 + tid;
      sR[tid] = p.rterm[b_row0 + col];
      const bool real = col < b_feat;
      sX[tid] = real ? (float)p.kp[(size_t)(b_row0 + col) * 2] : 0.0f;
      sY[tid] = real ? (float)p.kp[(size_t)(b_row0 + col) * 2 + 1] : 0.0f;
    }
    __syncthreads();
    for (int j = 0; j < 64; ++j) {
      int acc = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const v4i b = *reinterpret_cast<const v4i*>(&sB[j * 128 + c * 16]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_sdot4(a[c][e], b[e], acc, false);
      }
      int dist = acc + rterm_i + sR[j] + (1 << 21);
      const float xc = sX[j], yc = sY[j];
      const bool filtered = swap ? guided_filter(mode, M, xc, yc, xr, yr, p.max_residual)
                                 : guided_filter(mode, M, xr, yr, xc, yc, p.max_residual);
      if (filtered) dist = 0;
      if (dist > best) {  // sift.cc:126-132
        second = best;
        best = dist;
        best_j = (int)(j0 + j);
      } else if (dist > second) {
        second = dist;
      }
    }
  }
  int res = -1;
  if (best_j >= 0) {
    const float bn = p.lut[min(best, 262144)];
    if (!(bn > p.max_distance)) {
      const float sn = p.lut[min(second, 262144)];
      const float rhs = __fmul_rn(p.max_ratio, sn);
      if (!(bn >= rhs)) res = best_j;
    }
  }
  if (row < a_rows) p.out[p.d_out_off[d] + row] = res;
}

// ------------------------------------------------------------------------------------ K2
// One workgroup per undirected pair: mutual check + ordered compaction (sift.cc:171-197).
template <bool WRITE>
__global__ __launch_bounds__(256) void k2_cross_compact(const K2Params p) {
  const uint32_t pi = blockIdx.x;
  const uint4 pd = p.pair_dir[pi];  // {d_ab, d_ba, n1, n2}
  const int32_t* m12 = p.m + p.d_out_off[pd.x];
  const int32_t* m21 = p.cross_check ? (p.m + p.d_out_off[pd.y]) : nullptr;
  const uint32_t n1 = pd.z;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  __shared__ uint32_t wsum[4];
  uint32_t running = 0;
  uint2* dst = nullptr;
  if (WRITE) dst = reinterpret_cast<uint2*>(p.matches) + p.offsets[pi];
  for (uint32_t base = 0; base < n1; base += 256) {
    const uint32_t i = base + tid;
    int j = -1;
    bool ok = false;
    if (i < n1) {
      j = m12[i];
      ok = j >= 0;
      if (ok && p.cross_check) ok = (m21[j] == (int32_t)i);
    }
    const unsigned long long bal = __ballot(ok);
    const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = wsum[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (WRITE && ok) dst[running + wbase + before] = make_uint2(i, (uint32_t)j);
    running += total;
    __syncthreads();
  }
  if (!WRITE && tid == 0) p.counts[pi] = running;
}

// Cross-check after the gathered pass: entry k of pair pi is (i1, i2 = matches12[i1]); out2[k] = matches21[i2].
// Keep it when matches21[i2] == i1 (sift.cc:183-186), in entry order = ascending i1.
template <bool WRITE>
__global__ __launch_bounds__(256) void k2_entries_compact(const K2eParams p) {
  const uint32_t pi = blockIdx.x;
  const uint32_t n = p.e_cnt[pi];
  const uint2* ent = p.entries + p.e_off[pi];
  const int32_t* m21 = p.out2 + p.e_off[pi];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  __shared__ uint32_t wsum[4];
  uint32_t running = 0;
  uint2* dst = nullptr;
  if (WRITE) dst = reinterpret_cast<uint2*>(p.matches) + p.offsets[pi];
  for (uint32_t base = 0; base < n; base += 256) {
    const uint32_t k = base + tid;
    uint2 e = make_uint2(0, 0);
    bool ok = false;
    if (k < n) {
      e = ent[k];
      ok = m21[k] == (int32_t)e.x;
    }
    const unsigned long long bal = __ballot(ok);
    const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = wsum[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (WRITE && ok) dst[running + wbase + before] = e;
    running += total;
    __syncthreads();
  }
  if (!WRITE && tid == 0) p.counts[pi] = running;
}

// Single-workgroup exclusive scan of per-pair counts into 64-bit offsets:
// offsets[i] = base + sum_{k<i} counts[k], offsets[n] = total; *running_total updated.
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* __restrict__ counts,
                                                      uint64_t* __restrict__ offsets, uint32_t n,
                                                      uint64_t* __restrict__ running_total) {
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = *running_total;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t i = base + tid;
    const uint64_t v = i < n ? counts[i] : 0;
    uint64_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint64_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) wbase += wsum[w];
      total += wsum[w];
    }
    if (i < n) offsets[i] = carry + wbase + x - v;
    __syncthreads();
    if (tid == 0) carry += total;
    __syncthreads();
  }
  if (tid == 0) {
    offsets[n] = carry;
    *running_total = carry;
  }
}

// ------------------------------------------------------------------------------------ launchers
void launch_k0(const uint8_t* in_u8, int8_t* out_s8, int32_t* rterm, uint64_t n_rows, hipStream_t st) {
  if (n_rows == 0) return;
  const uint64_t threads = n_rows * 8;
  const uint32_t blocks = (uint32_t)((threads + 255) / 256);
  hipLaunchKernelGGL(k0_prepare, dim3(blocks), dim3(256), 0, st, in_u8, out_s8, rterm, n_rows);
}

// max_row_blocks = largest padded row count of an `a` image / 256
void launch_k1(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st) {
  if (n_directed == 0 || max_row_blocks == 0) return;
  if (p.entries)
    hipLaunchKernelGGL(k1_best_rows<true>, dim3(n_directed, (max_row_blocks + 1) / 2), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(k1_best_rows<false>, dim3(n_directed, (max_row_blocks + 1) / 2), dim3(256), 0, st, p);
}
#ifdef DSM_CHECK_BUILD
// comparison variant (DSM_K1_DOT4): pass 1 on the VALU; writes exact column indices, no resolve pass
void launch_k1_dot4(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st) {
  if (n_directed == 0 || max_row_blocks == 0) return;
  hipLaunchKernelGGL(k1_best_rows_dot4, dim3(n_directed, (max_row_blocks + 1) / 2), dim3(256), 0, st, p);
}
#endif
void launch_k1_resolve(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st) {
  if (n_directed == 0 || max_row_blocks == 0) return;
  if (p.entries)
    hipLaunchKernelGGL(k1_resolve_index<true>, dim3(n_directed), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(k1_resolve_index<false>, dim3(n_directed), dim3(256), 0, st, p);
}
void launch_k2_entries(const K2eParams& p, uint32_t n_pairs, bool write, hipStream_t st) {
  if (n_pairs == 0) return;
  if (write)
    hipLaunchKernelGGL(k2_entries_compact<true>, dim3(n_pairs), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(k2_entries_compact<false>, dim3(n_pairs), dim3(256), 0, st, p);
}

void launch_kg(const KgParams& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st) {
  if (n_directed == 0 || max_row_blocks == 0) return;
  hipLaunchKernelGGL(kg_best_rows, dim3(n_directed, max_row_blocks), dim3(256), 0, st, p);
}

struct GuidedPlan {
  uint64_t src_off, dst_off;
  uint32_t count, from_guided;
};
__global__ __launch_bounds__(64) void k_guided_assemble(const GuidedPlan* plan, const uint32_t* old_inl, const uint32_t* guided,
                                                        uint32_t* dst, uint32_t n_pairs) {
  for (uint32_t pi = blockIdx.x; pi < n_pairs; pi += gridDim.x) {
    const GuidedPlan pl = plan[pi];
    const uint2* s = reinterpret_cast<const uint2*>(pl.from_guided ? guided : old_inl) + pl.src_off;
    uint2* d = reinterpret_cast<uint2*>(dst) + pl.dst_off;
    for (uint32_t i = threadIdx.x; i < pl.count; i += 64) d[i] = s[i];
  }
}
void launch_guided_assemble(const void* plan, const uint32_t* old_inl, const uint32_t* guided, uint32_t* dst, uint32_t n_pairs,
                            hipStream_t st) {
  if (!n_pairs) return;
  hipLaunchKernelGGL(k_guided_assemble, dim3(n_pairs < 8192 ? n_pairs : 8192), dim3(64), 0, st,
                     static_cast<const GuidedPlan*>(plan), old_inl, guided, dst, n_pairs);
}

void launch_k2(const K2Params& p, uint32_t n_pairs, bool write, hipStream_t st) {
  if (n_pairs == 0) return;
  if (write)
    hipLaunchKernelGGL(k2_cross_compact<true>, dim3(n_pairs), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(k2_cross_compact<false>, dim3(n_pairs), dim3(256), 0, st, p);
}

void launch_scan(const uint32_t* counts, uint64_t* offsets, uint32_t n, uint64_t* running_total, hipStream_t st) {
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, counts, offsets, n, running_total);
}
