// view_graph.hip -- the step AFTER the stage: view-graph ingest of the two-view geometries and the
// rotation-cycle filter, for all pairs at once.
//   DistributedMapperController::LoadTwoviewGeometries   /root/reference/src/controllers/distributed_mapper_controller.cpp:585-631
//   ViewGraph::FilterViewGraphCyclesByRotation           /root/reference/src/graph/view_graph.cpp:115-165 (called with 5.0, :153)
//   ComputeLoopRotationError                             view_graph.cpp:44-69;  TripletExtractor  src/base/triplet_extractor.h
// The reference enumerates the cycles of length 3 through hash maps on one thread; here every edge (a, b), a < b, merges
// the sorted upper-neighbour lists of a and b: each common neighbour c > b is one triplet, whose loop rotation
// R_bc * R_ab * R_ac^T is evaluated on the spot, and a triplet below the threshold marks its three edges valid (plain
// stores of 1: the result is a set, no order involved).  ceres' rotation conversions are restated in oracle/view_graph.cc;
// sin / cos / atan2 are the device's (the decision `error < 5 degrees` is insensitive to their last place).
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <vector>

#include "ctx.h"

namespace {

__device__ void vg_quaternion_to_angle_axis(const double* q, double* aa) {
  const double q1 = q[1], q2 = q[2], q3 = q[3];
  const double sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3;
  double k = 2.0;
  if (sin_squared_theta > 0.0) {
    const double sin_theta = sqrt(sin_squared_theta);
    const double cos_theta = q[0];
    const double two_theta = 2.0 * ((cos_theta < 0.0) ? atan2(-sin_theta, -cos_theta) : atan2(sin_theta, cos_theta));
    k = two_theta / sin_theta;
  }
  aa[0] = q1 * k;
  aa[1] = q2 * k;
  aa[2] = q3 * k;
}
__device__ void vg_angle_axis_to_rotation(const double* aa, double* R) {  // row-major
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double costheta = cos(theta), sintheta = sin(theta);
    R[0] = costheta + wx * wx * (1.0 - costheta);
    R[3] = wz * sintheta + wx * wy * (1.0 - costheta);
    R[6] = -wy * sintheta + wx * wz * (1.0 - costheta);
    R[1] = wx * wy * (1.0 - costheta) - wz * sintheta;
    R[4] = costheta + wy * wy * (1.0 - costheta);
    R[7] = wx * sintheta + wy * wz * (1.0 - costheta);
    R[2] = wy * sintheta + wx * wz * (1.0 - costheta);
    R[5] = -wx * sintheta + wy * wz * (1.0 - costheta);
    R[8] = costheta + wz * wz * (1.0 - costheta);
  } else {
    R[0] = 1.0; R[3] = aa[2]; R[6] = -aa[1];
    R[1] = -aa[2]; R[4] = 1.0; R[7] = aa[0];
    R[2] = aa[1]; R[5] = -aa[0]; R[8] = 1.0;
  }
}
__device__ void vg_rotation_to_quaternion(const double* R, double* q) {
  const double trace = R[0] + R[4] + R[8];
  if (trace >= 0.0) {
    double t = sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j + 1] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k + 1] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}

// ingest: rotation_2 = QuaternionToAngleAxis(qvec) (distributed_mapper_controller.cpp:617-619), kept as the rotation
// matrix every triplet of the edge needs
__global__ void k_vg_edge_rotations(const double* __restrict__ qvec, uint32_t n_edges, double* __restrict__ R) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  double aa[3], Rm[9];
  vg_quaternion_to_angle_axis(qvec + 4 * (size_t)e, aa);
  vg_angle_axis_to_rotation(aa, Rm);
  for (int k = 0; k < 9; ++k) R[(size_t)e * 9 + k] = Rm[k];
}

// adjacency of upper neighbours: vertex a has (nbr[p], eid[p]) for p in [off[a], off[a + 1]), nbr ascending
__global__ void k_vg_triplets(const uint32_t* __restrict__ edge_a, const uint32_t* __restrict__ edge_b, uint32_t n_edges,
                              const uint32_t* __restrict__ off, const uint32_t* __restrict__ nbr, const uint32_t* __restrict__ eid,
                              const double* __restrict__ R, double max_loop_error_degrees, unsigned char* keep,
                              unsigned long long* n_triplets) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const uint32_t a = edge_a[e], b = edge_b[e];
  uint32_t pa = off[a], ea = off[a + 1], pb = off[b], eb = off[b + 1];
  unsigned long long found = 0;
  while (pa < ea && pb < eb) {
    const uint32_t ca = nbr[pa], cb = nbr[pb];
    if (ca < cb) {
      ++pa;
    } else if (cb < ca) {
      ++pb;
    } else {
      // triplet (a, b, c): edges e = (a, b), eid[pa] = (a, c), eid[pb] = (b, c)
      const uint32_t e13 = eid[pa], e23 = eid[pb];
      const double* R12 = R + (size_t)e * 9;
      const double* R13 = R + (size_t)e13 * 9;
      const double* R23 = R + (size_t)e23 * 9;
      double T[9], L[9], q[4], aa[3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = R23[i * 3 + 0] * R12[0 * 3 + j] + R23[i * 3 + 1] * R12[1 * 3 + j] + R23[i * 3 + 2] * R12[2 * 3 + j];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) L[i * 3 + j] = T[i * 3 + 0] * R13[j * 3 + 0] + T[i * 3 + 1] * R13[j * 3 + 1] + T[i * 3 + 2] * R13[j * 3 + 2];
      vg_rotation_to_quaternion(L, q);
      vg_quaternion_to_angle_axis(q, aa);
      const double err = sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]) * 57.29577951308232286464772187173366546630859375;
      if (err < max_loop_error_degrees) {
        keep[e] = 1;
        keep[e13] = 1;
        keep[e23] = 1;
      }
      ++found;
      ++pa;
      ++pb;
    }
  }
  if (found) atomicAdd(n_triplets, found);
}

}  // namespace

#define VCHK(ctx, call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
      return DSM_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

extern "C" int dsm_view_graph_filter_cycles(dsm_ctx* ctx, uint32_t n_pairs, const uint32_t* pairs, const double* qvecs,
                                            double max_loop_error_degrees, uint8_t* keep, uint64_t* n_triplets) {
  if (!ctx || (n_pairs && (!pairs || !qvecs || !keep))) return DSM_ERR_INVALID_ARGUMENT;
  if (n_triplets) *n_triplets = 0;
  if (n_pairs == 0) return DSM_OK;
  VCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // host: unique edges (a < b; a repeated pair is ignored like ViewGraph::AddTwoViewGeometry, view_graph.cpp:85-96),
  // vertices renumbered 0..V-1, CSR of upper neighbours sorted by neighbour
  std::vector<uint32_t> verts;
  verts.reserve(2 * (size_t)n_pairs);
  for (uint32_t e = 0; e < 2 * n_pairs; ++e) verts.push_back(pairs[e]);
  std::sort(verts.begin(), verts.end());
  verts.erase(std::unique(verts.begin(), verts.end()), verts.end());
  auto vid = [&](uint32_t id) { return (uint32_t)(std::lower_bound(verts.begin(), verts.end(), id) - verts.begin()); };
  const uint32_t V = (uint32_t)verts.size();
  struct E { uint32_t a, b, orig; };
  std::vector<E> edges;
  edges.reserve(n_pairs);
  for (uint32_t e = 0; e < n_pairs; ++e) {
    const uint32_t x = vid(pairs[2 * e]), y = vid(pairs[2 * e + 1]);
    if (x == y) continue;
    edges.push_back(E{std::min(x, y), std::max(x, y), e});
  }
  std::stable_sort(edges.begin(), edges.end(), [](const E& l, const E& r) { return l.a != r.a ? l.a < r.a : l.b < r.b; });
  std::vector<E> uniq;
  uniq.reserve(edges.size());
  for (const E& x : edges)  // the first occurrence in list order wins (stable sort)
    if (uniq.empty() || uniq.back().a != x.a || uniq.back().b != x.b) uniq.push_back(x);
  const uint32_t NE = (uint32_t)uniq.size();
  std::vector<uint32_t> ea(NE), eb(NE), off(V + 1, 0), nbr(NE), eid(NE);
  std::vector<double> q((size_t)NE * 4);
  for (uint32_t k = 0; k < NE; ++k) {
    ea[k] = uniq[k].a;
    eb[k] = uniq[k].b;
    off[uniq[k].a + 1]++;
    nbr[k] = uniq[k].b;  // sorted by (a, b): already CSR order
    eid[k] = k;
    for (int c = 0; c < 4; ++c) q[(size_t)k * 4 + c] = qvecs[(size_t)uniq[k].orig * 4 + c];
  }
  for (uint32_t v = 0; v < V; ++v) off[v + 1] += off[v];
  DevBuf d_ea, d_eb, d_off, d_nbr, d_eid, d_q, d_R, d_keep, d_cnt;
  int rc = DSM_OK;
  auto cleanup = [&]() {
    for (DevBuf* b : {&d_ea, &d_eb, &d_off, &d_nbr, &d_eid, &d_q, &d_R, &d_keep, &d_cnt}) b->release();
  };
#define VTRY(call)                                                       \
  do {                                                                   \
    hipError_t e_ = (call);                                              \
    if (e_ != hipSuccess && rc == DSM_OK) {                              \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);      \
      rc = DSM_ERR_HIP;                                                  \
    }                                                                    \
  } while (0)
  const size_t ne1 = std::max<uint32_t>(NE, 1);
  VTRY(d_ea.reserve(ne1 * 4));
  VTRY(d_eb.reserve(ne1 * 4));
  VTRY(d_off.reserve(((size_t)V + 1) * 4));
  VTRY(d_nbr.reserve(ne1 * 4));
  VTRY(d_eid.reserve(ne1 * 4));
  VTRY(d_q.reserve(ne1 * 32));
  VTRY(d_R.reserve(ne1 * 72));
  VTRY(d_keep.reserve(ne1));
  VTRY(d_cnt.reserve(8));
  std::vector<uint8_t> k8(NE, 0);
  unsigned long long cnt = 0;
  if (rc == DSM_OK && NE) {
    VTRY(hipMemcpyAsync(d_ea.p, ea.data(), (size_t)NE * 4, hipMemcpyHostToDevice, st));
    VTRY(hipMemcpyAsync(d_eb.p, eb.data(), (size_t)NE * 4, hipMemcpyHostToDevice, st));
    VTRY(hipMemcpyAsync(d_off.p, off.data(), ((size_t)V + 1) * 4, hipMemcpyHostToDevice, st));
    VTRY(hipMemcpyAsync(d_nbr.p, nbr.data(), (size_t)NE * 4, hipMemcpyHostToDevice, st));
    VTRY(hipMemcpyAsync(d_eid.p, eid.data(), (size_t)NE * 4, hipMemcpyHostToDevice, st));
    VTRY(hipMemcpyAsync(d_q.p, q.data(), (size_t)NE * 32, hipMemcpyHostToDevice, st));
    VTRY(hipMemsetAsync(d_keep.p, 0, NE, st));
    VTRY(hipMemsetAsync(d_cnt.p, 0, 8, st));
    if (rc == DSM_OK) {
      hipLaunchKernelGGL(k_vg_edge_rotations, dim3((NE + 255) / 256), dim3(256), 0, st, d_q.as<double>(), NE, d_R.as<double>());
      hipLaunchKernelGGL(k_vg_triplets, dim3((NE + 127) / 128), dim3(128), 0, st, d_ea.as<uint32_t>(), d_eb.as<uint32_t>(), NE,
                         d_off.as<uint32_t>(), d_nbr.as<uint32_t>(), d_eid.as<uint32_t>(), d_R.as<double>(), max_loop_error_degrees,
                         d_keep.as<unsigned char>(), d_cnt.as<unsigned long long>());
      VTRY(hipGetLastError());
      VTRY(hipMemcpyAsync(k8.data(), d_keep.p, NE, hipMemcpyDeviceToHost, st));
      VTRY(hipMemcpyAsync(&cnt, d_cnt.p, 8, hipMemcpyDeviceToHost, st));
      VTRY(hipStreamSynchronize(st));
    }
  }
#undef VTRY
  cleanup();
  if (rc != DSM_OK) return rc;
  for (uint32_t e = 0; e < n_pairs; ++e) keep[e] = 0;
  for (uint32_t k = 0; k < NE; ++k) keep[uniq[k].orig] = k8[k];
  if (n_triplets) *n_triplets = cnt;
  return DSM_OK;
}
