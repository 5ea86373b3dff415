// kernels.h -- parameter blocks and launchers shared by the HIP kernels and the C-ABI layer.
#ifndef DAGSFM_AMD_CSRC_KERNELS_H_
#define DAGSFM_AMD_CSRC_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

// K1: one directed matching pass per blockIdx.x, one 256-row block of image a per blockIdx.y.
struct K1Params {
  const int8_t* desc;         // s8 descriptors (u8 ^ 0x80), all images, rows padded per image to 256
  const int32_t* rterm;       // 128 * sum(s8 row)
  const uint2* dpairs;        // directed pairs {image a, image b}
  const uint32_t* img_row0;   // first padded row of every image
  const uint32_t* img_rows;   // padded row count of every image (multiple of 256, <= 8192)
  const uint64_t* d_out_off;  // per directed pair: offset (in rows) into `out`
  const float* lut;           // acosf(min(d / 2^18, 1)), d = 0..262144, built on the host
  float max_ratio;
  float max_distance;
  int32_t* out;               // per row: matched column index or -1
};

// K2: mutual check + ordered compaction, one workgroup per undirected pair.
struct K2Params {
  const uint4* pair_dir;      // {directed idx a->b, directed idx b->a, n1, n2}
  const uint64_t* d_out_off;
  const int32_t* m;           // K1 output
  int32_t cross_check;
  uint32_t* counts;           // [n_pairs] (count pass)
  const uint64_t* offsets;    // [n_pairs] absolute offsets into matches (write pass)
  uint32_t* matches;          // [total][2]
};

void launch_k0(const uint8_t* in_u8, int8_t* out_s8, int32_t* rterm, uint64_t n_rows, hipStream_t st);
void launch_k1(const K1Params& p, uint32_t n_directed, uint32_t max_row_blocks, hipStream_t st);
void launch_k2(const K2Params& p, uint32_t n_pairs, bool write, hipStream_t st);
void launch_scan(const uint32_t* counts, uint64_t* offsets, uint32_t n, uint64_t* running_total, hipStream_t st);

#endif
