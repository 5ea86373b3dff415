// flann_search.h -- internal interface of flann_search.hip (the FLANN-compatible word search on the device) to retrieval.hip
#ifndef DAGSFM_AMD_CSRC_FLANN_SEARCH_H_
#define DAGSFM_AMD_CSRC_FLANN_SEARCH_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dagsfm_mi355x.h"

struct dsm_ctx;
struct FlannDevice;

// uploads (and bounds-checks) an index over the context's vocabulary; ix == nullptr drops it.  d_words_s8: [num_words][128] s8 rows
int flann_device_set_index(dsm_ctx* ctx, FlannDevice** slot, const dsm_flann_index* ix, const int8_t* d_words_s8, uint32_t num_words);
void flann_device_destroy(FlannDevice* f);
// out_ids: device, [n_rows][out_stride]; out_dists: null or the same shape
int flann_device_search(dsm_ctx* ctx, FlannDevice* f, const int8_t* d_words_s8, const int8_t* desc, const int32_t* row_img, uint64_t n_rows,
                        uint32_t k, int32_t* out_ids, float* out_dists, uint32_t out_stride, hipStream_t st);
int flann_device_search_host(dsm_ctx* ctx, FlannDevice* f, const int8_t* d_words_s8, const uint8_t* queries, uint32_t n, uint32_t k, int32_t* ids,
                             float* dists);
double flann_device_last_ms(const FlannDevice* f);
int flann_device_algorithm(const FlannDevice* f);

#endif  // DAGSFM_AMD_CSRC_FLANN_SEARCH_H_
