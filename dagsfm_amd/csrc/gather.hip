// gather.hip -- libdagsfm_gather.so: multi-GPU assembly of the match graph with RCCL (include/dagsfm_gather.h).
//
// One process drives n devices, one dsm_ctx each (the mode of SiftMatchingOptions::gpu_index = "0,1,...": the reference
// starts one matcher per device and collects their outputs from a queue, /root/reference/src/feature/matching.cc:631-645,
// 814-836).  Device k has matched + verified share k of the pair list; this file assembles the shares on the devices:
//
//   1. offsets     every device writes its LOCAL prefix offsets of matches and inlier matches -- [2][pad + 1] uint64, pad = the
//                  largest share -- into its send buffer (C-ABI getters, device to device); one grouped ncclAllGather
//   2. totals      k_gather_totals picks every share's totals out of the gathered block; the host reads them (one copy of
//                  2 n words -- the only host round trip) and sizes the lists
//   3. lists       device k's matches / inlier matches go straight into slice k of the assembled lists on device k (getter,
//                  device to device) and from there to everybody: ncclBroadcast in place, root k, exact size, all 2 n calls
//                  in ONE group; the dsm_two_view_geometry records (fixed size) travel as a padded ncclAllGather in the same group
//   4. compact     k_gather_compact rebases the offsets to list order and moves the records out of their padded blocks
//
// RCCL over xGMI is point to point: a ring all-gather moves (n - 1) / n of the result over every link once; the padded
// blocks cost at most one share of records more than the exact size.  Everything runs on one stream per device; the
// getters' copies are synchronous, the collectives follow them on the gather's own streams.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dagsfm_gather.h"

namespace {

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Rank {
  dsm_ctx* ctx = nullptr;
  int device = -1;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  Buf off_send, off_recv, tvg_send, tvg_recv, totals;  // staging of the collectives
  Buf match_off, inl_off, matches, inliers, tvg;      // the assembled graph on this device
};

}  // namespace

struct dsm_gather {
  std::vector<Rank> ranks;
  std::string err;
  uint64_t n_pairs = 0, n_matches = 0, n_inliers = 0;
  bool with_geometry = false, assembled = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double ms = 0.0;
};

static int gfail(dsm_gather* g, int code, const std::string& msg) {
  if (g) g->err = msg;
  return code;
}
#define GHIP(g, call)                                                                      \
  do {                                                                                     \
    const hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) return gfail(g, DSM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define GNCCL(g, call)                                                                     \
  do {                                                                                     \
    const ncclResult_t r_ = (call);                                                        \
    if (r_ != ncclSuccess) return gfail(g, DSM_ERR_HIP, std::string(#call) + ": " + ncclGetErrorString(r_)); \
  } while (0)
// inside ncclGroupStart / ncclGroupEnd: a failed call closes the group before it returns, or the next collective of the
// process would be queued into a group nobody ends
#define GNCCL_GROUPED(g, call)                                                             \
  do {                                                                                     \
    const ncclResult_t r_ = (call);                                                        \
    if (r_ != ncclSuccess) {                                                               \
      (void)ncclGroupEnd();                                                                \
      return gfail(g, DSM_ERR_HIP, std::string(#call) + ": " + ncclGetErrorString(r_));    \
    }                                                                                      \
  } while (0)
#define GDSM(g, r, call)                                                                   \
  do {                                                                                     \
    const int rc_ = (call);                                                                \
    if (rc_ != DSM_OK) return gfail(g, rc_, std::string(#call) + ": " + dsm_last_error((r).ctx)); \
  } while (0)

// gathered: [n][2][pad + 1] local offsets; totals[2 k + a] = offsets of share k, array a, at its own length n_pairs[k]
__global__ void k_gather_totals(const uint64_t* gathered, const uint32_t* n_pairs, uint32_t n, uint32_t pad, uint64_t* totals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  const uint32_t k = i >> 1, a = i & 1;
  totals[i] = gathered[((size_t)k * 2 + a) * (pad + 1) + n_pairs[k]];
}

// One workgroup per share and 256-pair tile: offsets rebased to list order (+ the totals of the shares before it), records
// copied out of their padded block (as 8-byte words: sizeof(dsm_two_view_geometry) is a multiple of 8, checked on the host).
__global__ __launch_bounds__(256) void k_gather_compact(const uint64_t* gathered, const uint8_t* tvg_padded, const uint32_t* n_pairs,
                                                        const uint64_t* totals, uint32_t n, uint32_t pad, uint32_t rec_bytes,
                                                        uint64_t* match_off, uint64_t* inl_off, uint8_t* tvg) {
  const uint32_t k = blockIdx.y;
  const uint32_t nk = n_pairs[k];
  const uint32_t i0 = blockIdx.x * 256u;
  if (i0 > nk) return;
  uint64_t begin = 0, base_m = 0, base_i = 0;
  for (uint32_t q = 0; q < k; ++q) {
    begin += n_pairs[q];
    base_m += totals[2 * q];
    base_i += totals[2 * q + 1];
  }
  const uint32_t i = i0 + threadIdx.x;
  const bool last_share = k + 1 == n;
  // entry nk of a share is the next share's entry 0: written once, by the last share
  if (i < nk || (i == nk && last_share)) {
    match_off[begin + i] = base_m + gathered[((size_t)k * 2 + 0) * (pad + 1) + i];
    if (inl_off) inl_off[begin + i] = base_i + gathered[((size_t)k * 2 + 1) * (pad + 1) + i];
  }
  if (tvg) {
    const uint32_t words = rec_bytes / 8u;
    const uint32_t cnt = (nk - i0 < 256u ? nk - i0 : 256u) * words;
    const uint64_t* src = reinterpret_cast<const uint64_t*>(tvg_padded + ((size_t)k * pad + i0) * rec_bytes);
    uint64_t* dst = reinterpret_cast<uint64_t*>(tvg + ((size_t)begin + i0) * rec_bytes);
    for (uint32_t w = threadIdx.x; w < cnt; w += 256u) dst[w] = src[w];
  }
}

extern "C" {

int dsm_gather_create(dsm_ctx* const* ctxs, uint32_t n_ctx, dsm_gather** out) {
  if (!ctxs || !out || n_ctx == 0) return DSM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  dsm_gather* g = new dsm_gather();
  g->ranks.resize(n_ctx);
  std::vector<int> devs(n_ctx);
  for (uint32_t k = 0; k < n_ctx; ++k) {
    if (!ctxs[k]) {
      delete g;
      return DSM_ERR_INVALID_ARGUMENT;
    }
    g->ranks[k].ctx = ctxs[k];
    g->ranks[k].device = devs[k] = dsm_ctx_device(ctxs[k]);
    for (uint32_t q = 0; q < k; ++q)
      if (devs[q] == devs[k]) {  // RCCL refuses a device twice in one communicator
        delete g;
        return DSM_ERR_INVALID_ARGUMENT;
      }
  }
  std::vector<ncclComm_t> comms(n_ctx);
  if (ncclCommInitAll(comms.data(), (int)n_ctx, devs.data()) != ncclSuccess) {
    delete g;
    return DSM_ERR_HIP;
  }
  for (uint32_t k = 0; k < n_ctx; ++k) {
    g->ranks[k].comm = comms[k];
    if (hipSetDevice(devs[k]) != hipSuccess || hipStreamCreateWithFlags(&g->ranks[k].stream, hipStreamNonBlocking) != hipSuccess) {
      dsm_gather_destroy(g);
      return DSM_ERR_HIP;
    }
  }
  (void)hipSetDevice(devs[0]);
  if (hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) {
    dsm_gather_destroy(g);
    return DSM_ERR_HIP;
  }
  *out = g;
  return DSM_OK;
}

void dsm_gather_destroy(dsm_gather* g) {
  if (!g) return;
  for (Rank& r : g->ranks) {
    if (r.device >= 0) (void)hipSetDevice(r.device);
    if (r.stream) (void)hipStreamSynchronize(r.stream);
    if (r.comm) (void)ncclCommDestroy(r.comm);
    for (Buf* b : {&r.off_send, &r.off_recv, &r.tvg_send, &r.tvg_recv, &r.totals, &r.match_off, &r.inl_off, &r.matches, &r.inliers, &r.tvg})
      b->release();
    if (r.stream) (void)hipStreamDestroy(r.stream);
  }
  if (!g->ranks.empty() && g->ranks[0].device >= 0) (void)hipSetDevice(g->ranks[0].device);
  if (g->ev0) (void)hipEventDestroy(g->ev0);
  if (g->ev1) (void)hipEventDestroy(g->ev1);
  delete g;
}

const char* dsm_gather_last_error(const dsm_gather* g) { return g ? g->err.c_str() : "null dsm_gather"; }

int dsm_gather_match_graph(dsm_gather* g, const uint32_t* n_pairs, int32_t with_geometry) {
  if (!g || !n_pairs) return DSM_ERR_INVALID_ARGUMENT;
  static_assert(sizeof(dsm_two_view_geometry) % 8 == 0, "records are copied as 8-byte words");
  const uint32_t n = (uint32_t)g->ranks.size();
  const size_t rec = sizeof(dsm_two_view_geometry);
  g->assembled = false;
  uint64_t N = 0;
  uint32_t pad = 0;
  for (uint32_t k = 0; k < n; ++k) {
    N += n_pairs[k];
    pad = std::max(pad, n_pairs[k]);
  }
  if (N > 0xfffffff0ull) return gfail(g, DSM_ERR_OUT_OF_RANGE, "pair list too long");
  // the getters below write what the CONTEXT holds: a share length that is not the context's would overrun the staging blocks
  for (uint32_t k = 0; k < n; ++k)
    if (n_pairs[k] && dsm_ctx_num_pairs(g->ranks[k].ctx) != n_pairs[k])
      return gfail(g, DSM_ERR_INVALID_ARGUMENT, "dsm_gather_match_graph: n_pairs[k] is not the length of the list context k matched");
  const size_t off_block = 2 * ((size_t)pad + 1);  // uint64 per rank: [matches | inliers][pad + 1]
  // ---- 1. local offsets -> send buffers; all-gather
  for (uint32_t k = 0; k < n; ++k) {
    Rank& r = g->ranks[k];
    GHIP(g, hipSetDevice(r.device));
    GHIP(g, r.off_send.reserve(off_block * 8));
    GHIP(g, r.off_recv.reserve(off_block * 8 * n));
    GHIP(g, r.totals.reserve(((size_t)2 * n + n) * 8));  // totals [2 n] uint64, then n_pairs [n] uint32
    GHIP(g, hipMemsetAsync(r.off_send.p, 0, off_block * 8, r.stream));
    GHIP(g, hipStreamSynchronize(r.stream));
    uint64_t* send = static_cast<uint64_t*>(r.off_send.p);
    if (n_pairs[k]) {  // (an empty share: the context may not have run at all; its offsets are the zeros above)
      GDSM(g, r, dsm_get_matches(r.ctx, send, nullptr, 0));
      if (with_geometry) GDSM(g, r, dsm_get_inlier_matches(r.ctx, send + pad + 1, nullptr, 0));
    }
    uint32_t* np_dev = reinterpret_cast<uint32_t*>(static_cast<uint64_t*>(r.totals.p) + 2 * n);
    GHIP(g, hipMemcpy(np_dev, n_pairs, (size_t)n * 4, hipMemcpyHostToDevice));
    GHIP(g, hipDeviceSynchronize());  // the getters copy device to device on the null stream: done before the collectives start
  }
  GHIP(g, hipSetDevice(g->ranks[0].device));
  GHIP(g, hipEventRecord(g->ev0, g->ranks[0].stream));
  GNCCL(g, ncclGroupStart());
  for (uint32_t k = 0; k < n; ++k) {
    Rank& r = g->ranks[k];
    GNCCL_GROUPED(g, ncclAllGather(r.off_send.p, r.off_recv.p, off_block, ncclUint64, r.comm, r.stream));
  }
  GNCCL(g, ncclGroupEnd());
  // ---- 2. totals of every share (rank 0's copy; all copies are equal)
  std::vector<uint64_t> totals(2 * (size_t)n, 0);
  for (uint32_t k = 0; k < n; ++k) {
    Rank& r = g->ranks[k];
    GHIP(g, hipSetDevice(r.device));
    const uint32_t* np_dev = reinterpret_cast<const uint32_t*>(static_cast<uint64_t*>(r.totals.p) + 2 * n);
    hipLaunchKernelGGL(k_gather_totals, dim3((2 * n + 63) / 64), dim3(64), 0, r.stream, static_cast<const uint64_t*>(r.off_recv.p), np_dev, n, pad,
                       static_cast<uint64_t*>(r.totals.p));
    GHIP(g, hipGetLastError());
  }
  {
    Rank& r = g->ranks[0];
    GHIP(g, hipSetDevice(r.device));
    GHIP(g, hipMemcpyAsync(totals.data(), r.totals.p, totals.size() * 8, hipMemcpyDeviceToHost, r.stream));
    GHIP(g, hipStreamSynchronize(r.stream));
  }
  std::vector<uint64_t> base_m(n + 1, 0), base_i(n + 1, 0);
  for (uint32_t k = 0; k < n; ++k) {
    base_m[k + 1] = base_m[k] + totals[2 * k];
    base_i[k + 1] = base_i[k] + totals[2 * k + 1];
  }
  const uint64_t M = base_m[n], I = base_i[n];
  // ---- 3. every share's lists into its slice of the assembled lists on its own device, then to everybody
  for (uint32_t k = 0; k < n; ++k) {
    Rank& r = g->ranks[k];
    GHIP(g, hipSetDevice(r.device));
    GHIP(g, r.match_off.reserve((N + 1) * 8));
    GHIP(g, r.matches.reserve(std::max<uint64_t>(M, 1) * 8));
    if (with_geometry) {
      GHIP(g, r.inl_off.reserve((N + 1) * 8));
      GHIP(g, r.inliers.reserve(std::max<uint64_t>(I, 1) * 8));
      GHIP(g, r.tvg.reserve(std::max<uint64_t>(N, 1) * rec));
      GHIP(g, r.tvg_send.reserve(std::max<size_t>(pad, 1) * rec));
      GHIP(g, r.tvg_recv.reserve(std::max<size_t>(pad, 1) * rec * n));
    }
    if (n_pairs[k]) {
      if (totals[2 * k]) GDSM(g, r, dsm_get_matches(r.ctx, nullptr, static_cast<uint32_t*>(r.matches.p) + 2 * base_m[k], totals[2 * k]));
      if (with_geometry) {
        if (totals[2 * k + 1])
          GDSM(g, r, dsm_get_inlier_matches(r.ctx, nullptr, static_cast<uint32_t*>(r.inliers.p) + 2 * base_i[k], totals[2 * k + 1]));
        GDSM(g, r, dsm_get_two_view_geometries(r.ctx, static_cast<dsm_two_view_geometry*>(r.tvg_send.p)));
      }
    }
    GHIP(g, hipDeviceSynchronize());
  }
  GNCCL(g, ncclGroupStart());
  for (uint32_t root = 0; root < n; ++root) {
    for (uint32_t k = 0; k < n; ++k) {
      Rank& r = g->ranks[k];
      if (totals[2 * root]) {
        void* slice = static_cast<uint32_t*>(r.matches.p) + 2 * base_m[root];
        GNCCL_GROUPED(g, ncclBroadcast(slice, slice, totals[2 * root], ncclUint64, (int)root, r.comm, r.stream));  // one match = 2 x uint32
      }
      if (with_geometry && totals[2 * root + 1]) {
        void* slice = static_cast<uint32_t*>(r.inliers.p) + 2 * base_i[root];
        GNCCL_GROUPED(g, ncclBroadcast(slice, slice, totals[2 * root + 1], ncclUint64, (int)root, r.comm, r.stream));
      }
    }
  }
  if (with_geometry && pad)
    for (uint32_t k = 0; k < n; ++k) {
      Rank& r = g->ranks[k];
      GNCCL_GROUPED(g, ncclAllGather(r.tvg_send.p, r.tvg_recv.p, (size_t)pad * rec, ncclUint8, r.comm, r.stream));
    }
  GNCCL(g, ncclGroupEnd());
  // ---- 4. offsets in list order, records out of their padded blocks
  for (uint32_t k = 0; k < n; ++k) {
    Rank& r = g->ranks[k];
    GHIP(g, hipSetDevice(r.device));
    const uint32_t* np_dev = reinterpret_cast<const uint32_t*>(static_cast<uint64_t*>(r.totals.p) + 2 * n);
    if (N == 0) {
      GHIP(g, hipMemsetAsync(r.match_off.p, 0, 8, r.stream));
      if (with_geometry) GHIP(g, hipMemsetAsync(r.inl_off.p, 0, 8, r.stream));
    } else {
      hipLaunchKernelGGL(k_gather_compact, dim3(pad / 256 + 1, n), dim3(256), 0, r.stream, static_cast<const uint64_t*>(r.off_recv.p),
                         static_cast<const uint8_t*>(with_geometry ? r.tvg_recv.p : nullptr), np_dev, static_cast<const uint64_t*>(r.totals.p), n, pad,
                         (uint32_t)rec, static_cast<uint64_t*>(r.match_off.p), with_geometry ? static_cast<uint64_t*>(r.inl_off.p) : nullptr,
                         with_geometry ? static_cast<uint8_t*>(r.tvg.p) : nullptr);
      GHIP(g, hipGetLastError());
    }
  }
  GHIP(g, hipSetDevice(g->ranks[0].device));
  GHIP(g, hipEventRecord(g->ev1, g->ranks[0].stream));
  for (uint32_t k = 0; k < n; ++k) {
    GHIP(g, hipSetDevice(g->ranks[k].device));
    GHIP(g, hipStreamSynchronize(g->ranks[k].stream));
  }
  float ms = 0.f;
  GHIP(g, hipSetDevice(g->ranks[0].device));
  GHIP(g, hipEventElapsedTime(&ms, g->ev0, g->ev1));
  g->ms = ms;
  g->n_pairs = N;
  g->n_matches = M;
  g->n_inliers = with_geometry ? I : 0;
  g->with_geometry = with_geometry != 0;
  g->assembled = true;
  return DSM_OK;
}

int dsm_gather_sizes(const dsm_gather* g, uint64_t* n_pairs, uint64_t* n_matches, uint64_t* n_inlier_matches) {
  if (!g) return DSM_ERR_INVALID_ARGUMENT;
  if (!g->assembled) return DSM_ERR_NOT_READY;
  if (n_pairs) *n_pairs = g->n_pairs;
  if (n_matches) *n_matches = g->n_matches;
  if (n_inlier_matches) *n_inlier_matches = g->n_inliers;
  return DSM_OK;
}

int dsm_gather_device_arrays(const dsm_gather* g, uint32_t rank, const uint64_t** match_offsets, const uint32_t** matches,
                             const dsm_two_view_geometry** geometries, const uint64_t** inlier_offsets, const uint32_t** inlier_matches) {
  if (!g || rank >= g->ranks.size()) return DSM_ERR_INVALID_ARGUMENT;
  if (!g->assembled) return DSM_ERR_NOT_READY;
  const Rank& r = g->ranks[rank];
  if (match_offsets) *match_offsets = static_cast<const uint64_t*>(r.match_off.p);
  if (matches) *matches = static_cast<const uint32_t*>(r.matches.p);
  if (geometries) *geometries = g->with_geometry ? static_cast<const dsm_two_view_geometry*>(r.tvg.p) : nullptr;
  if (inlier_offsets) *inlier_offsets = g->with_geometry ? static_cast<const uint64_t*>(r.inl_off.p) : nullptr;
  if (inlier_matches) *inlier_matches = g->with_geometry ? static_cast<const uint32_t*>(r.inliers.p) : nullptr;
  return DSM_OK;
}

int dsm_gather_fetch(dsm_gather* g, uint32_t rank, uint64_t* match_offsets, uint32_t* matches, dsm_two_view_geometry* geometries,
                     uint64_t* inlier_offsets, uint32_t* inlier_matches) {
  if (!g || rank >= g->ranks.size()) return DSM_ERR_INVALID_ARGUMENT;
  if (!g->assembled) return gfail(g, DSM_ERR_NOT_READY, "dsm_gather_match_graph has not run");
  if (!g->with_geometry && (geometries || inlier_offsets || inlier_matches)) return gfail(g, DSM_ERR_NOT_READY, "assembled without geometry");
  Rank& r = g->ranks[rank];
  GHIP(g, hipSetDevice(r.device));
  if (match_offsets) GHIP(g, hipMemcpy(match_offsets, r.match_off.p, (g->n_pairs + 1) * 8, hipMemcpyDefault));
  if (matches && g->n_matches) GHIP(g, hipMemcpy(matches, r.matches.p, g->n_matches * 8, hipMemcpyDefault));
  if (geometries && g->n_pairs) GHIP(g, hipMemcpy(geometries, r.tvg.p, g->n_pairs * sizeof(dsm_two_view_geometry), hipMemcpyDefault));
  if (inlier_offsets) GHIP(g, hipMemcpy(inlier_offsets, r.inl_off.p, (g->n_pairs + 1) * 8, hipMemcpyDefault));
  if (inlier_matches && g->n_inliers) GHIP(g, hipMemcpy(inlier_matches, r.inliers.p, g->n_inliers * 8, hipMemcpyDefault));
  return DSM_OK;
}

int dsm_gather_time(const dsm_gather* g, double* ms) {
  if (!g || !ms) return DSM_ERR_INVALID_ARGUMENT;
  if (!g->assembled) return DSM_ERR_NOT_READY;
  *ms = g->ms;
  return DSM_OK;
}

}  // extern "C"
