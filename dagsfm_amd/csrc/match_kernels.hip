// match_kernels.hip -- brute-force SIFT descriptor matching on MI355X (gfx950).
//
// Replaces, bit-exactly, the reference CPU matcher
//   ComputeSiftDistanceMatrix   /root/reference/src/feature/sift.cc:76-109
//   FindBestMatchesOneWay       /root/reference/src/feature/sift.cc:111-162
//   FindBestMatches             /root/reference/src/feature/sift.cc:164-198
// without ever materialising the N1 x N2 int32 matrix.
//
// K0  k0_prepare        u8 -> s8 (x ^ 0x80) + per-row bias term, once per image upload.
// K1  k1_best_rows      one *directed* pass (image a rows vs image b columns):
//                       int8 MFMA 32x32x32 distance tiles, top-2 per row fused in the
//                       epilogue as packed (value<<8 | tile) keys (v_lshl_add, v_med3, v_max),
//                       thresholds (acos LUT, ratio) applied before the single int32 store.
//                       Cross-check uses the second directed pass (b rows vs a columns),
//                       which is exactly FindBestMatchesOneWay(dists.transpose()).
// K2  k2_cross_compact  mutual check + ordered compaction (ascending idx1) per pair.
//
// Exactness of the signed-MFMA trick: a = a' + 128, b = b' + 128,
//   dot = S + rterm(i) + rterm(j) + 2^21,  S = sum a'b',  rterm(x) = 128 * sum x'
// all in int32 (|dot| <= 128*255^2 < 2^23).  Per row the comparison value is
//   v' = dot - rterm(i) - 2^22 = S + rterm(j) - 2^21   in (-2^23, 2^23)
// so key = ((v' + 2^23) << 7) + (127 - tile) is an order-preserving key: larger dot wins, equal
// dots resolve to the lower tile.  The key is a positive int32 below 0x7F800000 with a non-zero
// exponent field, i.e. also a finite normal float with the SAME ordering, so the top-2 update runs
// on the full-rate float pipe (v_max_f32 / v_med3_f32); only the key build is an integer op.
// A tile index has 7 bits => columns are swept in segments of 128 tiles (4 096 columns) whose
// results are merged in ascending order; the 32 columns of a tile live in different lanes and
// are merged per segment with a (key, lane) butterfly, which resolves equal dots to the lowest
// column index exactly like the reference's ascending strict-`>` scan.  Top-2 of *distinct* keys
// reproduces the reference's second-best semantics (a duplicate of the best value is the
// second best, sift.cc:126-132).  Zero padding rows/columns have dot == 0 and can never
// beat the initial best_dist = second_best_dist = 0 (sift.cc:122-123).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

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
// keys are positive normal floats bit-wise (see above): float max / med3 order them like integers
__device__ __forceinline__ int key_med3(int a, int b, int c) {
  int d;
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int key_max(int a, int b) {
  int d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}


// LDS image of a 64-column B tile: column c occupies 128 B; its eight 16-B chunks are
// XOR-swizzled with (c>>1)&7 so that the 16-lane ds_read_b128 groups are conflict free.
__device__ __forceinline__ int lds_off(int col, int chunk) {
  return col * 128 + ((chunk ^ ((col >> 1) & 7)) << 4);
}

__global__ __launch_bounds__(256, 2) void k1_best_rows(const K1Params p) {
  const uint32_t d = blockIdx.x;
  const uint32_t rb = blockIdx.y;
  const uint2 ab = p.dpairs[d];
  const uint32_t a_rows = p.img_rows[ab.x];
  if (rb * 256u >= a_rows) return;
  const uint32_t b_cols = p.img_rows[ab.y];
  const uint32_t a_row0 = p.img_row0[ab.x] + rb * 256u;
  const uint32_t b_row0 = p.img_row0[ab.y];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  __shared__ __attribute__((aligned(16))) int8_t sB[2][64 * 128];

  int32_t* out = p.out + p.d_out_off[d] + rb * 256u + wave * 64;

  // Resident A fragments: 64 rows x 128 B per wave.
  const int8_t* arow = p.desc + (size_t)(a_row0 + wave * 64) * 128;
  v4i afrag[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      afrag[rt][ks] = *reinterpret_cast<const v4i*>(arow + (rt * 32 + l31) * 128 + ks * 32 + half * 16);

  const int8_t* bimg = p.desc + (size_t)b_row0 * 128;
  const int32_t* rt_bimg = p.rterm + b_row0;
  const int32_t* rt_a = p.rterm + a_row0 + wave * 64;
  int key0[2][16];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      key0[rt][r] = (int)((uint32_t)((1 << 22) - rt_a[row]) << 7);  // dot 0: v' + 2^23 = 2^22 - rterm(i)
    }
  // running result of this lane's row over the column segments (best_dist = second_best_dist = 0 initially)
  int run_b = 0, run_s = 0;
  uint32_t run_j = 0;

#define K1_MFMA4(ACC, RT, BF)                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                         \
      ACC = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[RT][ks], BF[ks], ACC, 0, 0, 0)
#define K1_EPILOGUE(ACC, RT, KT)                                                           \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
    const int key = (int)((uint32_t)ACC[r] << 7) + KT;                                     \
    second[RT][r] = key_med3(best[RT][r], second[RT][r], key);                             \
    best[RT][r] = key_max(best[RT][r], key);                                               \
  }
// one MFMA followed by 12 VALU of the previous tile's epilogue, four times
#define K1_INTERLEAVE()                                                                    \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);                                    \
  }

  for (uint32_t seg0 = 0; seg0 < b_cols; seg0 += 4096) {
    const uint32_t seg_cols = (b_cols - seg0) < 4096u ? (b_cols - seg0) : 4096u;
    const uint32_t nsteps = seg_cols >> 6;
    const int8_t* bbase = bimg + (size_t)seg0 * 128;
    const int32_t* rt_b = rt_bimg + seg0;
    int best[2][16], second[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        best[rt][r] = key0[rt][r];
        second[rt][r] = key0[rt][r];
      }
    v4i stage[2];
    int colterm[2];
    // prologue: tile 0 of the segment
#pragma unroll
    for (int u = 0; u < 2; ++u) stage[u] = *reinterpret_cast<const v4i*>(bbase + (size_t)(tid + 256 * u) * 16);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) colterm[ct] = rt_b[ct * 32 + l31];
    __syncthreads();  // the previous segment's last tile is no longer being read
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + 256 * u;
      *reinterpret_cast<v4i*>(&sB[0][lds_off(q >> 3, q & 7)]) = stage[u];
    }
    __syncthreads();
    // make sure nothing issued before the loop is still pending at its head: hipcc's waitcnt pass would
    // otherwise keep a conservative vmcnt wait at the top of every iteration (right behind the prefetch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (uint32_t s = 0; s < nsteps; ++s) {
      const int cur = s & 1;
      const bool more = (s + 1) < nsteps;
      // keys of this step's two column tiles from the already-resident column terms
      int kterm0 = (int)((uint32_t)(colterm[0] + (1 << 23) - (1 << 21)) << 7) + (127 - (int)(2 * s));
      int kterm1 = (int)((uint32_t)(colterm[1] + (1 << 23) - (1 << 21)) << 7) + (127 - (int)(2 * s + 1));
      // keep each kterm one opaque VGPR: otherwise the compiler re-associates (acc << 7) + a + b into
      // v_lshlrev + v_add3 (2 VALU per element) instead of one v_lshl_add_u32
      asm volatile("" : "+v"(kterm0));
      asm volatile("" : "+v"(kterm1));
      // prefetch of the next B tile (global -> registers); consumed at the end of the step
      int colterm_next[2] = {0, 0};
      if (more) {
        const int8_t* src = bbase + (size_t)(s + 1) * 64 * 128;
#pragma unroll
        for (int u = 0; u < 2; ++u) stage[u] = *reinterpret_cast<const v4i*>(src + (size_t)(tid + 256 * u) * 16);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) colterm_next[ct] = rt_b[(s + 1) * 64 + ct * 32 + l31];
      }
      v4i bf0[4], bf1[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf0[ks] = *reinterpret_cast<const v4i*>(&sB[cur][lds_off(l31, ks * 2 + half)]);
        bf1[ks] = *reinterpret_cast<const v4i*>(&sB[cur][lds_off(32 + l31, ks * 2 + half)]);
      }
      const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      v16i accA = zero16, accB = zero16;
      // software pipeline over the four 32x32 tiles of the step: the MFMAs of tile i+1 run in the matrix
      // pipe while the VALU does the top-2 epilogue of tile i
      K1_MFMA4(accA, 0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      K1_MFMA4(accB, 1, bf0);
      K1_EPILOGUE(accA, 0, kterm0);
      K1_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      accA = zero16;
      K1_MFMA4(accA, 0, bf1);
      K1_EPILOGUE(accB, 1, kterm0);
      K1_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      accB = zero16;
      K1_MFMA4(accB, 1, bf1);
      K1_EPILOGUE(accA, 0, kterm1);
      K1_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      K1_EPILOGUE(accB, 1, kterm1);
      __builtin_amdgcn_sched_barrier(0);

      if (more) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = tid + 256 * u;
          *reinterpret_cast<v4i*>(&sB[cur ^ 1][lds_off(q >> 3, q & 7)]) = stage[u];
        }
        colterm[0] = colterm_next[0];
        colterm[1] = colterm_next[1];
      }
      __syncthreads();
    }

    // (key, source lane) butterfly over the 32 lanes that hold the 32 columns of every tile.  Equal keys
    // from two lanes are two distinct columns of the same tile: the lower lane (= lower column) keeps
    // "best", the other one becomes the second best -- exactly sift.cc:126-132.
    int myB = 0, myS = 0, myL = 0;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int B = best[rt][r], S = second[rt][r], L = l31;
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
          const int oB = __shfl_xor(B, m);
          const int oS = __shfl_xor(S, m);
          const int oL = __shfl_xor(L, m);
          const bool take = (oB > B) || (oB == B && oL < L);
          const int lo = min(B, oB);
          S = max(lo, max(S, oS));
          B = take ? oB : B;
          L = take ? oL : L;
        }
        if (l31 == rt * 16 + r) {
          myB = B;
          myS = S;
          myL = L;
        }
      }
    {
      const int rt = l31 >> 4, r = l31 & 15;
      const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int rti = rt_a[row];
      const int seg_b = (myB >> 7) - (1 << 23) + rti + (1 << 22);
      const int seg_s = (myS >> 7) - (1 << 23) + rti + (1 << 22);
      const uint32_t seg_j = seg0 + (uint32_t)(127 - (myB & 127)) * 32u + (uint32_t)myL;
      // merge with the earlier segments: earlier columns win ties (ascending strict-`>` scan)
      const int lo = min(run_b, seg_b);
      const int new_s = max(lo, max(run_s, seg_s));
      if (seg_b > run_b) {
        run_b = seg_b;
        run_j = seg_j;
      }
      run_s = new_s;
    }
  }
#undef K1_MFMA4
#undef K1_EPILOGUE
#undef K1_INTERLEAVE

  {
    const int rt = l31 >> 4, r = l31 & 15;
    const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const int best_dot = run_b;
    const int second_dot = run_s;
    const uint32_t j = run_j;
    int res = -1;
    if (best_dot > 0) {  // best_i2 != -1, sift.cc:136
      const float bn = p.lut[min(best_dot, 262144)];
      if (!(bn > p.max_distance)) {  // sift.cc:144
        const float sn = p.lut[min(second_dot, 262144)];
        const float rhs = __fmul_rn(p.max_ratio, sn);
        if (!(bn >= rhs)) res = (int)j;  // sift.cc:153
      }
    }
    out[row] = res;
  }
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

void launch_k1(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st) {
  if (n_directed == 0 || max_row_blocks == 0) return;
  hipLaunchKernelGGL(k1_best_rows, dim3(n_directed, max_row_blocks), dim3(256), 0, st, p);
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
