/* dagsfm_gather.h -- multi-GPU assembly of the match graph with RCCL, below the host language.
 *
 * C-ABI of libdagsfm_gather.so, a companion of libdagsfm_mi355x.so (include/dagsfm_mi355x.h) that links librccl.  It is a
 * library of its own so that a single-GPU host never maps RCCL (0.5 s of load time for a stage that runs one second).
 *
 * Replaces, for a host that drives several devices from ONE process: the shared output queue the reference's per-device
 * matcher threads push into (one SiftGPUFeatureMatcher per `gpu_index` device, /root/reference/src/feature/matching.cc:
 * 631-645; results collected by the caller's loop, :814-836).  Here every device matches + verifies a contiguous share of
 * the pair list in its own dsm_ctx, and the shares are assembled ON THE DEVICES: the fixed-size per-pair records (prefix
 * offsets of matches / inlier matches, dsm_two_view_geometry) by one grouped ncclAllGather per array, padded to the
 * largest share; the variable-size lists (matches, inlier matches) by exact-size ncclBroadcast calls, one per source,
 * inside one group.  Afterwards EVERY device holds the whole graph in list order (what the view-graph kernels of the next
 * stage read), and the host fetches it from one device with a single copy per array.
 * One process per GPU (bench.py, torch.distributed) makes the same calls through dagsfm_amd/sharding.py.
 *
 * No torch types, plain pointers and sizes.  Every function returns DSM_OK (0) or a DSM_ERR_* code of dagsfm_mi355x.h;
 * dsm_gather_last_error gives the text. */
#ifndef DAGSFM_GATHER_H_
#define DAGSFM_GATHER_H_

#include <stdint.h>

#include "dagsfm_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsm_gather dsm_gather;

/* One communicator over the devices of `ctxs` (ncclCommInitAll: one process, n_ctx devices, rank k = ctxs[k]) and one
 * stream per device.  The contexts must sit on DISTINCT devices (RCCL refuses a device twice); n_ctx = 1 is allowed and
 * makes every collective a one-rank call (what a one-GPU box can exercise). */
int dsm_gather_create(dsm_ctx* const* ctxs, uint32_t n_ctx, dsm_gather** out);
void dsm_gather_destroy(dsm_gather* g);
const char* dsm_gather_last_error(const dsm_gather* g);

/* Assembles the match graph of a pair list whose share k (n_pairs[k] pairs, shares back to back in list order) was
 * matched -- and, with_geometry != 0, verified -- by ctxs[k] (dsm_match_pairs / dsm_set_matches, dsm_verify_pairs).
 * On return every device holds: match offsets [N + 1], matches, and with geometry the records [N], inlier offsets
 * [N + 1] and inlier matches of all N = sum n_pairs[k] pairs, rebased to list order. */
int dsm_gather_match_graph(dsm_gather* g, const uint32_t* n_pairs, int32_t with_geometry);

/* Sizes of the assembled graph. */
int dsm_gather_sizes(const dsm_gather* g, uint64_t* n_pairs, uint64_t* n_matches, uint64_t* n_inlier_matches);
/* The assembled arrays as device pointers on the device of ctxs[rank] (valid until the next dsm_gather_match_graph /
 * dsm_gather_destroy); any out pointer may be NULL.  For a consumer that stays on the device. */
int dsm_gather_device_arrays(const dsm_gather* g, uint32_t rank, const uint64_t** match_offsets, const uint32_t** matches,
                             const dsm_two_view_geometry** geometries, const uint64_t** inlier_offsets,
                             const uint32_t** inlier_matches);
/* One fetch per array from the copy on ctxs[rank]'s device into host (or device) memory; any pointer may be NULL.
 * match_offsets / inlier_offsets: N + 1 entries; matches / inlier_matches: n_matches / n_inlier_matches x 2 uint32. */
int dsm_gather_fetch(dsm_gather* g, uint32_t rank, uint64_t* match_offsets, uint32_t* matches, dsm_two_view_geometry* geometries,
                     uint64_t* inlier_offsets, uint32_t* inlier_matches);
/* Wall time of the collectives of the last dsm_gather_match_graph (HIP events on rank 0's stream), in milliseconds. */
int dsm_gather_time(const dsm_gather* g, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* DAGSFM_GATHER_H_ */
