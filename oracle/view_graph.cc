// view_graph.cc -- CPU restatement of the view-graph ingest and rotation-cycle filter that consume the stage's output
// (SURVEY.md 8f rank 4).  TEST INFRASTRUCTURE ONLY.
//
//   DistributedMapperController::LoadTwoviewGeometries   /root/reference/src/controllers/distributed_mapper_controller.cpp:585-631
//       per two_view_geometries row: rotation_2 = QuaternionToAngleAxis(qvec), position_2 = tvec, visibility = |inliers|
//   ViewGraph::AddTwoViewGeometry / FilterViewGraphCyclesByRotation(5.0)   src/graph/view_graph.cpp:85-167
//   ComputeLoopRotationError                                               src/graph/view_graph.cpp:44-69
//   TripletExtractor::ExtractTriplets (all cycles of length 3)             src/base/triplet_extractor.h
//   RadToDeg                                                               src/util/math.h:207-209
// Third party, absent from /root/reference: ceres-solver (unpinned, CMakeLists.txt find_package(Ceres)) --
// QuaternionToAngleAxis, AngleAxisToRotationMatrix, RotationMatrixToAngleAxis (= RotationMatrixToQuaternion +
// QuaternionToAngleAxis) are restated from the published ceres/rotation.h.  sin / cos / atan2 come from the host libm.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <set>
#include <utility>
#include <vector>

namespace {

// ceres::QuaternionToAngleAxis (rotation.h): q = (w, x, y, z)
void QuaternionToAngleAxis(const double* q, double* aa) {
  const double q1 = q[1], q2 = q[2], q3 = q[3];
  const double sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3;
  if (sin_squared_theta > 0.0) {
    const double sin_theta = std::sqrt(sin_squared_theta);
    const double cos_theta = q[0];
    const double two_theta = 2.0 * ((cos_theta < 0.0) ? std::atan2(-sin_theta, -cos_theta) : std::atan2(sin_theta, cos_theta));
    const double k = two_theta / sin_theta;
    aa[0] = q1 * k;
    aa[1] = q2 * k;
    aa[2] = q3 * k;
  } else {
    const double k = 2.0;
    aa[0] = q1 * k;
    aa[1] = q2 * k;
    aa[2] = q3 * k;
  }
}
// ceres::AngleAxisToRotationMatrix; R row-major here (the reference goes through ColumnMajorAdapter3x3: same matrix)
void AngleAxisToRotationMatrix(const double* aa, double* R) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double costheta = std::cos(theta), sintheta = std::sin(theta);
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
// ceres::RotationMatrixToQuaternion
void RotationMatrixToQuaternion(const double* R, double* q) {
  const double trace = R[0] + R[4] + R[8];
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
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
    double t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j + 1] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k + 1] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}
void Mul3(const double* A, const double* B, double* C) {  // Eigen 3x3 product: sum over k in order
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
// ComputeLoopRotationError, view_graph.cpp:44-69
double LoopRotationErrorDegrees(const double* aa12, const double* aa13, const double* aa23) {
  double R12[9], R13[9], R23[9], R13t[9], T[9], L[9], q[4], aa[3];
  AngleAxisToRotationMatrix(aa12, R12);
  AngleAxisToRotationMatrix(aa13, R13);
  AngleAxisToRotationMatrix(aa23, R23);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R13t[i * 3 + j] = R13[j * 3 + i];
  Mul3(R23, R12, T);  // rotation2_3 * rotation1_2 * rotation1_3.transpose()
  Mul3(T, R13t, L);
  RotationMatrixToQuaternion(L, q);
  QuaternionToAngleAxis(q, aa);
  const double norm = std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  return norm * 57.29577951308232286464772187173366546630859375;
}

}  // namespace

extern "C" {

// pairs [n][2] image ids, qvecs [n][4] (w, x, y, z) of the relative pose of the pair as stored (image id1 < id2 after the
// database's swap); keep[n] = the pair survives FilterViewGraphCyclesByRotation(max_loop_error_degrees).  A pair that
// repeats an earlier one is ignored like ViewGraph::AddTwoViewGeometry does (keep = 0).  Returns the number of triplets.
uint64_t oracle_view_graph_filter_cycles(uint32_t n, const uint32_t* pairs, const double* qvecs, double max_loop_error_degrees,
                                         uint8_t* keep, double* min_error_per_pair) {
  std::map<std::pair<uint32_t, uint32_t>, uint32_t> edge;  // (a < b) -> index
  std::vector<double> aa(static_cast<size_t>(n) * 3);
  std::map<uint32_t, std::set<uint32_t>> adj;
  for (uint32_t e = 0; e < n; ++e) {
    keep[e] = 0;
    if (min_error_per_pair) min_error_per_pair[e] = std::numeric_limits<double>::infinity();
    const uint32_t a = std::min(pairs[2 * e], pairs[2 * e + 1]), b = std::max(pairs[2 * e], pairs[2 * e + 1]);
    if (a == b || edge.count(std::make_pair(a, b))) continue;
    edge[std::make_pair(a, b)] = e;
    QuaternionToAngleAxis(qvecs + 4 * static_cast<size_t>(e), &aa[3 * static_cast<size_t>(e)]);  // LoadTwoviewGeometries :617-619
    adj[a].insert(b);
  }
  uint64_t triplets = 0;
  for (const auto& kv : edge) {
    const uint32_t a = kv.first.first, b = kv.first.second;
    const auto ita = adj.find(a), itb = adj.find(b);
    if (ita == adj.end() || itb == adj.end()) continue;
    for (uint32_t c : itb->second) {  // c > b
      if (!ita->second.count(c)) continue;
      const uint32_t e12 = kv.second, e13 = edge[std::make_pair(a, c)], e23 = edge[std::make_pair(b, c)];
      ++triplets;
      const double err = LoopRotationErrorDegrees(&aa[3 * static_cast<size_t>(e12)], &aa[3 * static_cast<size_t>(e13)], &aa[3 * static_cast<size_t>(e23)]);
      if (min_error_per_pair)
        for (uint32_t e : {e12, e13, e23}) min_error_per_pair[e] = std::min(min_error_per_pair[e], err);
      if (err < max_loop_error_degrees) keep[e12] = keep[e13] = keep[e23] = 1;
    }
  }
  return triplets;
}

}  // extern "C"
