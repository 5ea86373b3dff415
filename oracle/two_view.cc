// oracle/two_view.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle of two-view verification).
//
// CPU restatement of the reference's RANSAC two-view geometry path; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Follows, in order:
//   util/random.{h,cc}            /root/reference/src/util/random.h:90-129, random.cc:38-56
//   optim/random_sampler.cc       /root/reference/src/optim/random_sampler.cc:43-62
//   optim/support_measurement.cc  /root/reference/src/optim/support_measurement.cc:36-60
//   optim/ransac.h                /root/reference/src/optim/ransac.h:135-167
//   optim/loransac.h              /root/reference/src/optim/loransac.h:91-233
//   estimators/utils.cc           /root/reference/src/estimators/utils.cc:38-131
//   estimators/fundamental_matrix.cc   :47-192     estimators/homography_matrix.cc :44-131
//   estimators/essential_matrix.cc     :46-150     estimators/translation_transform.h :81-118
//   base/polynomial.cc            :63-275
//   estimators/two_view_geometry.cc    :113-126, 232-555
//   base/essential_matrix.cc :41-89, base/homography_matrix.cc :45-192, base/pose.cc :70-73,
//   :225-247, base/triangulation.cc :39-52, :183-218, base/projection.cc :47-53, :193-197,
//   base/camera.cc :75-93, :210-219, base/camera_models.h :535-587, :629-757, util/math.h:211-229
//
// The PRNG is the real libstdc++ std::mt19937 + std::uniform_int_distribution<uint32_t>, as
// in the reference; the reference seeds it per verifier thread from the wall clock
// (random.cc:40-56) and is therefore not reproducible, so this oracle (and the GPU path) take
// an explicit per-pair seed and draw E -> F -> H -> watermark from the one stream, in the order
// of two_view_geometry.cc:325-342, 547-549.
//
// The generated straight-line polynomial code of the 5-point solver (estimators/essential_matrix_poly.h,
// _coeffs.h) is evaluated in the reference's own order of sums and products: that order is kept as a term
// table (dagsfm_amd/csrc/fivept_terms.tbl) and expanded at build time (fivept_build_A / fivept_det_poly).
//
// Parity pinning: every known-answer test the reference holds for these functions is restated
// in tests/test_oracle_estimators.py.  TwoViewGeometry::Estimate* itself and LO-RANSAC over the
// geometric estimators have no tests in the reference: "parity unpinned" for those two.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <set>
#include <utility>
#include <vector>

#include "../include/dagsfm_mi355x.h"
#include "linalg.h"
// atan / sin / cos / tan of the camera models: the correctly rounded values, computed by the same plain-double code as
// on the device (dagsfm_amd/csrc/exact_trig.h; tests/test_exact_trig.py checks it against this host's libm)
#define DSM_XT static inline
#define DSM_XT_CONST static const
#include "../dagsfm_amd/csrc/exact_trig.h"

namespace oracle {

struct Vec2 {
  double x, y;
};

// ------------------------------------------------------------------------------------ PRNG
// One stream per image pair (see header comment).
struct PRNG {
  std::mt19937 gen;
  explicit PRNG(uint32_t seed) : gen(seed) {}
  // RandomInteger<uint32_t>, random.h:90-98
  uint32_t RandomInteger(uint32_t mn, uint32_t mx) {
    std::uniform_int_distribution<uint32_t> distribution(mn, mx);
    return distribution(gen);
  }
};

// RandomSampler, random_sampler.cc:43-62 + Shuffle, random.h:122-129
struct RandomSampler {
  size_t num_samples;
  std::vector<size_t> sample_idxs;
  PRNG* prng;
  RandomSampler(size_t k, PRNG* p) : num_samples(k), prng(p) {}
  void Initialize(size_t total) {
    sample_idxs.resize(total);
    std::iota(sample_idxs.begin(), sample_idxs.end(), 0);
  }
  std::vector<size_t> Sample() {
    const uint32_t last_idx = static_cast<uint32_t>(sample_idxs.size() - 1);
    for (uint32_t i = 0; i < static_cast<uint32_t>(num_samples); ++i) {
      const uint32_t j = prng->RandomInteger(i, last_idx);
      std::swap(sample_idxs[i], sample_idxs[j]);
    }
    return std::vector<size_t>(sample_idxs.begin(), sample_idxs.begin() + num_samples);
  }
};

// ------------------------------------------------------------------------------------ support
struct Support {
  size_t num_inliers = 0;
  double residual_sum = std::numeric_limits<double>::max();
};
static Support EvaluateSupport(const std::vector<double>& residuals, double max_residual) {
  Support s;
  s.num_inliers = 0;
  s.residual_sum = 0;
  for (const double r : residuals) {
    if (r <= max_residual) {
      s.num_inliers += 1;
      s.residual_sum += r;
    }
  }
  return s;
}
static bool CompareSupport(const Support& a, const Support& b) {
  if (a.num_inliers > b.num_inliers) return true;
  return a.num_inliers == b.num_inliers && a.residual_sum < b.residual_sum;
}

// RANSAC::ComputeNumTrials, ransac.h:150-167
static size_t ComputeNumTrials(size_t num_inliers, size_t num_samples, double confidence, int min_samples) {
  const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
  const double nom = 1 - confidence;
  if (nom <= 0) return std::numeric_limits<size_t>::max();
  const double denom = 1 - std::pow(inlier_ratio, min_samples);
  if (denom <= 0) return 1;
  return static_cast<size_t>(std::ceil(std::log(nom) / std::log(denom)));
}

// ------------------------------------------------------------------------------------ utils.cc
static void CenterAndNormalizeImagePoints(const std::vector<Vec2>& points, std::vector<Vec2>* normed, Mat3* matrix) {
  double cx = 0, cy = 0;
  for (const Vec2& p : points) {
    cx += p.x;
    cy += p.y;
  }
  cx /= points.size();
  cy /= points.size();
  double rms = 0;
  for (const Vec2& p : points) {
    const double dx = p.x - cx, dy = p.y - cy;
    rms += dx * dx + dy * dy;
  }
  rms = std::sqrt(rms / points.size());
  const double norm_factor = std::sqrt(2.0) / rms;
  Mat3& M = *matrix;
  M(0, 0) = norm_factor; M(0, 1) = 0; M(0, 2) = -norm_factor * cx;
  M(1, 0) = 0; M(1, 1) = norm_factor; M(1, 2) = -norm_factor * cy;
  M(2, 0) = 0; M(2, 1) = 0; M(2, 2) = 1;
  normed->resize(points.size());
  for (size_t i = 0; i < points.size(); ++i) {
    const double p_0 = points[i].x, p_1 = points[i].y;
    const double np_0 = M(0, 0) * p_0 + M(0, 1) * p_1 + M(0, 2);
    const double np_1 = M(1, 0) * p_0 + M(1, 1) * p_1 + M(1, 2);
    const double np_2 = M(2, 0) * p_0 + M(2, 1) * p_1 + M(2, 2);
    const double inv_np_2 = 1.0 / np_2;
    (*normed)[i].x = np_0 * inv_np_2;
    (*normed)[i].y = np_1 * inv_np_2;
  }
}

static void ComputeSquaredSampsonError(const std::vector<Vec2>& points1, const std::vector<Vec2>& points2,
                                       const Mat3& E, std::vector<double>* residuals) {
  residuals->resize(points1.size());
  const double E_00 = E(0, 0), E_01 = E(0, 1), E_02 = E(0, 2);
  const double E_10 = E(1, 0), E_11 = E(1, 1), E_12 = E(1, 2);
  const double E_20 = E(2, 0), E_21 = E(2, 1), E_22 = E(2, 2);
  for (size_t i = 0; i < points1.size(); ++i) {
    const double x1_0 = points1[i].x, x1_1 = points1[i].y;
    const double x2_0 = points2[i].x, x2_1 = points2[i].y;
    const double Ex1_0 = E_00 * x1_0 + E_01 * x1_1 + E_02;
    const double Ex1_1 = E_10 * x1_0 + E_11 * x1_1 + E_12;
    const double Ex1_2 = E_20 * x1_0 + E_21 * x1_1 + E_22;
    const double Etx2_0 = E_00 * x2_0 + E_10 * x2_1 + E_20;
    const double Etx2_1 = E_01 * x2_0 + E_11 * x2_1 + E_21;
    const double x2tEx1 = x2_0 * Ex1_0 + x2_1 * Ex1_1 + Ex1_2;
    (*residuals)[i] = x2tEx1 * x2tEx1 / (Ex1_0 * Ex1_0 + Ex1_1 * Ex1_1 + Etx2_0 * Etx2_0 + Etx2_1 * Etx2_1);
  }
}

// ------------------------------------------------------------------------------------ polynomial.cc
static bool FindLinearPolynomialRoots(const std::vector<double>& c, std::vector<double>* real, std::vector<double>* imag) {
  if (c[0] == 0) return false;
  real->assign(1, -c[1] / c[0]);
  imag->assign(1, 0.0);
  return true;
}
static bool FindQuadraticPolynomialRoots(const std::vector<double>& coeffs, std::vector<double>* real,
                                         std::vector<double>* imag) {
  const double a = coeffs[0];
  if (a == 0) return FindLinearPolynomialRoots(std::vector<double>(coeffs.begin() + 1, coeffs.end()), real, imag);
  const double b = coeffs[1], c = coeffs[2];
  if (b == 0 && c == 0) {
    real->assign(1, 0.0);
    imag->assign(1, 0.0);
    return true;
  }
  const double d = b * b - 4 * a * c;
  if (d >= 0) {
    const double sqrt_d = std::sqrt(d);
    real->resize(2);
    if (b >= 0) {
      (*real)[0] = (-b - sqrt_d) / (2 * a);
      (*real)[1] = (2 * c) / (-b - sqrt_d);
    } else {
      (*real)[0] = (2 * c) / (-b + sqrt_d);
      (*real)[1] = (-b + sqrt_d) / (2 * a);
    }
    imag->assign(2, 0.0);
  } else {
    real->assign(2, -b / (2 * a));
    imag->resize(2);
    (*imag)[0] = std::sqrt(-d) / (2 * a);
    (*imag)[1] = -(*imag)[0];
  }
  return true;
}
static bool FindPolynomialRootsCompanionMatrix(const std::vector<double>& coeffs_all, std::vector<double>* real,
                                               std::vector<double>* imag) {
  size_t lead = 0;
  for (; lead < coeffs_all.size(); ++lead)
    if (coeffs_all[lead] != 0) break;
  std::vector<double> coeffs(coeffs_all.begin() + lead, coeffs_all.end());
  const int degree = static_cast<int>(coeffs.size()) - 1;
  if (degree <= 0) return false;
  if (degree == 1) return FindLinearPolynomialRoots(coeffs, real, imag);
  if (degree == 2) return FindQuadraticPolynomialRoots(coeffs, real, imag);
  size_t trail = 0;
  for (; trail < coeffs.size(); ++trail)
    if (coeffs[coeffs.size() - 1 - trail] != 0) break;
  coeffs.resize(coeffs.size() - trail);
  if (coeffs.size() == 1) {
    real->assign(1, 0.0);
    imag->assign(1, 0.0);
    return true;
  }
  const int n = static_cast<int>(coeffs.size()) - 1;
  Mat C(n, n);
  for (int i = 1; i < n; ++i) C(i, i - 1) = 1;
  for (int j = 0; j < n; ++j) C(0, j) = -coeffs[j + 1] / coeffs[0];
  std::vector<double> re, im;
  if (!real_eigenvalues(C, &re, &im)) return false;
  const int effective_degree = n < degree ? n + 1 : n;
  real->assign(effective_degree, 0.0);
  imag->assign(effective_degree, 0.0);
  for (int i = 0; i < n; ++i) {
    (*real)[i] = re[i];
    (*imag)[i] = im[i];
  }
  return true;
}

// ------------------------------------------------------------------------------------ estimators
struct FundamentalSevenPoint {
  typedef Mat3 M_t;
  static const int kMinNumSamples = 7;
  static std::vector<M_t> Estimate(const std::vector<Vec2>& points1, const std::vector<Vec2>& points2) {
    Mat A(7, 9);
    for (int i = 0; i < 7; ++i) {
      const double x0 = points1[i].x, y0 = points1[i].y, x1 = points2[i].x, y1 = points2[i].y;
      A(i, 0) = x1 * x0; A(i, 1) = x1 * y0; A(i, 2) = x1;
      A(i, 3) = y1 * x0; A(i, 4) = y1 * y0; A(i, 5) = y1;
      A(i, 6) = x0; A(i, 7) = y0; A(i, 8) = 1;
    }
    const SVD svd = jacobi_svd(A, false);
    double f1[9], f2[9];
    for (int k = 0; k < 9; ++k) {
      f1[k] = svd.V(k, 7);
      f2[k] = svd.V(k, 8);
    }
    for (int k = 0; k < 9; ++k) f1[k] -= f2[k];
    const double t0 = f1[4] * f1[8] - f1[5] * f1[7];
    const double t1 = f1[3] * f1[8] - f1[5] * f1[6];
    const double t2 = f1[3] * f1[7] - f1[4] * f1[6];
    const double t3 = f2[4] * f2[8] - f2[5] * f2[7];
    const double t4 = f2[3] * f2[8] - f2[5] * f2[6];
    const double t5 = f2[3] * f2[7] - f2[4] * f2[6];
    std::vector<double> coeffs(4);
    coeffs[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    coeffs[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
                f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
                f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
                f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    coeffs[2] = f1[0] * t3 - f1[1] * t4 + f1[2] * t5 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
                f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
                f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
                f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    coeffs[3] = f2[0] * t3 - f2[1] * t4 + f2[2] * t5;
    std::vector<double> roots_real, roots_imag;
    if (!FindPolynomialRootsCompanionMatrix(coeffs, &roots_real, &roots_imag)) return {};
    std::vector<M_t> models;
    for (size_t i = 0; i < roots_real.size(); ++i) {
      const double kMaxRootImag = 1e-10;
      if (std::abs(roots_imag[i]) > kMaxRootImag) continue;
      const double lambda = roots_real[i];
      const double mu = 1;
      double F[9];
      for (int k = 0; k < 9; ++k) F[k] = lambda * f1[k] + mu * f2[k];
      const double kEps = 1e-10;
      if (std::abs(F[8]) < kEps) continue;
      const double f22 = F[8];
      M_t model;
      for (int k = 0; k < 9; ++k) model.m[k] = F[k] / f22;  // row-major reshape == F.transpose() of the col-major one
      models.push_back(model);
    }
    return models;
  }
  static void Residuals(const std::vector<Vec2>& p1, const std::vector<Vec2>& p2, const M_t& F, std::vector<double>* r) {
    ComputeSquaredSampsonError(p1, p2, F, r);
  }
};

// shared by the 8-point F estimator: constraint matrix + nullspace + rank-2 projection
static Mat3 EightPointCore(const std::vector<Vec2>& points1, const std::vector<Vec2>& points2, bool essential) {
  std::vector<Vec2> n1, n2;
  Mat3 N1, N2;
  CenterAndNormalizeImagePoints(points1, &n1, &N1);
  CenterAndNormalizeImagePoints(points2, &n2, &N2);
  const int n = static_cast<int>(points1.size());
  Mat cm(n, 9);
  for (int i = 0; i < n; ++i) {
    const double h[3] = {n1[i].x, n1[i].y, 1.0};
    for (int k = 0; k < 3; ++k) {
      cm(i, k) = h[k] * n2[i].x;
      cm(i, 3 + k) = h[k] * n2[i].y;
      cm(i, 6 + k) = h[k];
    }
  }
  const SVD svd = jacobi_svd(cm, false);
  Mat3 Et;  // ematrix_t.transpose(): row-major reshape of the null vector
  for (int k = 0; k < 9; ++k) Et.m[k] = svd.V(k, 8);
  const SVD s3 = jacobi_svd(to_mat(Et), true);
  double sv[3] = {s3.sv[0], s3.sv[1], s3.sv[2]};
  if (essential) {
    sv[0] = (sv[0] + sv[1]) / 2.0;
    sv[1] = sv[0];
  }
  sv[2] = 0.0;
  Mat3 U = to_mat3(s3.U), V = to_mat3(s3.V);
  Mat3 US;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) US(i, j) = U(i, j) * sv[j];
  const Mat3 F = mat3_mul(US, mat3_transpose(V));
  return mat3_mul(mat3_mul(mat3_transpose(N2), F), N1);
}

struct FundamentalEightPoint {
  typedef Mat3 M_t;
  static const int kMinNumSamples = 8;
  static std::vector<M_t> Estimate(const std::vector<Vec2>& p1, const std::vector<Vec2>& p2) {
    return {EightPointCore(p1, p2, false)};
  }
  static void Residuals(const std::vector<Vec2>& p1, const std::vector<Vec2>& p2, const M_t& F, std::vector<double>* r) {
    ComputeSquaredSampsonError(p1, p2, F, r);
  }
};

struct HomographyEstimator {
  typedef Mat3 M_t;
  static const int kMinNumSamples = 4;
  static std::vector<M_t> Estimate(const std::vector<Vec2>& points1, const std::vector<Vec2>& points2) {
    const int N = static_cast<int>(points1.size());
    std::vector<Vec2> n1, n2;
    Mat3 N1, N2;
    CenterAndNormalizeImagePoints(points1, &n1, &N1);
    CenterAndNormalizeImagePoints(points2, &n2, &N2);
    Mat A(2 * N, 9);
    for (int i = 0, j = N; i < N; ++i, ++j) {
      const double s_0 = n1[i].x, s_1 = n1[i].y, d_0 = n2[i].x, d_1 = n2[i].y;
      A(i, 0) = -s_0; A(i, 1) = -s_1; A(i, 2) = -1;
      A(i, 6) = s_0 * d_0; A(i, 7) = s_1 * d_0; A(i, 8) = d_0;
      A(j, 3) = -s_0; A(j, 4) = -s_1; A(j, 5) = -1;
      A(j, 6) = s_0 * d_1; A(j, 7) = s_1 * d_1; A(j, 8) = d_1;
    }
    const SVD svd = jacobi_svd(A, false);
    Mat3 Ht;  // H_t.transpose()
    for (int k = 0; k < 9; ++k) Ht.m[k] = svd.V(k, 8);
    return {mat3_mul(mat3_mul(mat3_inverse(N2), Ht), N1)};
  }
  static void Residuals(const std::vector<Vec2>& points1, const std::vector<Vec2>& points2, const M_t& H,
                        std::vector<double>* residuals) {
    residuals->resize(points1.size());
    const double H_00 = H(0, 0), H_01 = H(0, 1), H_02 = H(0, 2);
    const double H_10 = H(1, 0), H_11 = H(1, 1), H_12 = H(1, 2);
    const double H_20 = H(2, 0), H_21 = H(2, 1), H_22 = H(2, 2);
    for (size_t i = 0; i < points1.size(); ++i) {
      const double s_0 = points1[i].x, s_1 = points1[i].y, d_0 = points2[i].x, d_1 = points2[i].y;
      const double pd_0 = H_00 * s_0 + H_01 * s_1 + H_02;
      const double pd_1 = H_10 * s_0 + H_11 * s_1 + H_12;
      const double pd_2 = H_20 * s_0 + H_21 * s_1 + H_22;
      const double inv_pd_2 = 1.0 / pd_2;
      const double dd_0 = d_0 - pd_0 * inv_pd_2;
      const double dd_1 = d_1 - pd_1 * inv_pd_2;
      (*residuals)[i] = dd_0 * dd_0 + dd_1 * dd_1;
    }
  }
};

struct TranslationEstimator {
  typedef Vec2 M_t;
  static const int kMinNumSamples = 1;
  static std::vector<M_t> Estimate(const std::vector<Vec2>& p1, const std::vector<Vec2>& p2) {
    double sx = 0, sy = 0, dx = 0, dy = 0;
    for (size_t i = 0; i < p1.size(); ++i) {
      sx += p1[i].x; sy += p1[i].y;
      dx += p2[i].x; dy += p2[i].y;
    }
    sx /= p1.size(); sy /= p1.size();
    dx /= p2.size(); dy /= p2.size();
    return {Vec2{dx - sx, dy - sy}};
  }
  static void Residuals(const std::vector<Vec2>& p1, const std::vector<Vec2>& p2, const M_t& t, std::vector<double>* r) {
    r->resize(p1.size());
    for (size_t i = 0; i < p1.size(); ++i) {
      const double ex = p2[i].x - p1[i].x - t.x;
      const double ey = p2[i].y - p1[i].y - t.y;
      (*r)[i] = ex * ex + ey * ey;
    }
  }
};

// ---- 5-point (Nister): steps 3 and 4 in the reference's own evaluation order -------------------
// The reference builds the 10 x 20 constraint matrix and the degree-10 determinant polynomial with
// generated straight-line code (essential_matrix.cc:76-77 `#include "estimators/essential_matrix_poly.h"`,
// :101-102 `#include "estimators/essential_matrix_coeffs.h"`).  The order of its sums and products is data, kept
// in dagsfm_amd/csrc/fivept_terms.tbl (tools/gen_fivept_tables.py) and expanded into fivept_poly_gen.inc by
// this directory's Makefile; tests/test_fivept_reference_order.py compares both functions bit for bit with the
// reference's headers compiled behind a shim (oracle/_ref/libfivept_ref.so).
// e = E.data() (9 x 4, column-major), a = A.data() (10 x 20, column-major), b = B.data() (13 x 3, column-major).
static void fivept_build_A(const double* e, double* a) {
  double e2[36];
  double e3[36];
  for (size_t i = 0; i < 36; ++i) {
    e2[i] = e[i] * e[i];
    e3[i] = e2[i] * e[i];
  }
#define FIVEPT_E(k) e[k]
#define FIVEPT_E2(k) e2[k]
#define FIVEPT_E3(k) e3[k]
#define FIVEPT_A(i) a[i]
#define FIVEPT_FENCE(n)
#define FIVEPT_EMIT_A
#include "fivept_poly_gen.inc"
#undef FIVEPT_EMIT_A
#undef FIVEPT_A
#undef FIVEPT_E3
#undef FIVEPT_E2
#undef FIVEPT_E
}
static void fivept_det_poly(const double* b, double* coeffs) {
#define FIVEPT_B(k) b[k]
#define FIVEPT_C(i) coeffs[i]
#define FIVEPT_EMIT_C
#include "fivept_poly_gen.inc"
#undef FIVEPT_EMIT_C
#undef FIVEPT_FENCE
#undef FIVEPT_C
#undef FIVEPT_B
}

struct EssentialFivePoint {
  typedef Mat3 M_t;
  static const int kMinNumSamples = 5;
  static std::vector<M_t> Estimate(const std::vector<Vec2>& points1, const std::vector<Vec2>& points2) {
    const int n = static_cast<int>(points1.size());
    Mat Q(n, 9);
    for (int i = 0; i < n; ++i) {
      const double x1_0 = points1[i].x, x1_1 = points1[i].y, x2_0 = points2[i].x, x2_1 = points2[i].y;
      Q(i, 0) = x1_0 * x2_0; Q(i, 1) = x1_1 * x2_0; Q(i, 2) = x2_0;
      Q(i, 3) = x1_0 * x2_1; Q(i, 4) = x1_1 * x2_1; Q(i, 5) = x2_1;
      Q(i, 6) = x1_0; Q(i, 7) = x1_1; Q(i, 8) = 1;
    }
    const SVD svd = jacobi_svd(Q, false);
    double Ecm[36];  // E = svd.matrixV().block<9, 4>(0, 5), column-major like Eigen's E.data()
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 4; ++c) Ecm[c * 9 + r] = svd.V(r, 5 + c);
#define Eb(r, c) Ecm[(c) * 9 + (r)]
    double Acm[200];  // A.data(): A(r, c) = Acm[c * 10 + r]
    fivept_build_A(Ecm, Acm);
    Mat A1(10, 10), A2(10, 10), AA;
    for (int r = 0; r < 10; ++r)
      for (int c = 0; c < 10; ++c) {
        A1(r, c) = Acm[c * 10 + r];
        A2(r, c) = Acm[(10 + c) * 10 + r];
      }
    partial_piv_lu_solve(A1, A2, &AA);
    double Bcm[39];  // B.data(): B(r, c) = Bcm[c * 13 + r]
#define B(r, c) Bcm[(c) * 13 + (r)]
    for (int i = 0; i < 3; ++i) {
      B(0, i) = 0; B(4, i) = 0; B(8, i) = 0;
      for (int k = 0; k < 3; ++k) {
        B(1 + k, i) = AA(i * 2 + 4, k);
        B(5 + k, i) = AA(i * 2 + 4, 3 + k);
      }
      for (int k = 0; k < 4; ++k) B(9 + k, i) = AA(i * 2 + 4, 6 + k);
      for (int k = 0; k < 3; ++k) {
        B(0 + k, i) -= AA(i * 2 + 5, k);
        B(4 + k, i) -= AA(i * 2 + 5, 3 + k);
      }
      for (int k = 0; k < 4; ++k) B(8 + k, i) -= AA(i * 2 + 5, 6 + k);
    }
    double c11[11];
    fivept_det_poly(Bcm, c11);
    std::vector<double> coeffs(c11, c11 + 11), roots_real, roots_imag;
    if (!FindPolynomialRootsCompanionMatrix(coeffs, &roots_real, &roots_imag)) return {};
    std::vector<M_t> models;
    for (size_t i = 0; i < roots_imag.size(); ++i) {
      const double kMaxRootImag = 1e-10;
      if (std::abs(roots_imag[i]) > kMaxRootImag) continue;
      const double z1 = roots_real[i];
      const double z2 = z1 * z1;
      const double z3 = z2 * z1;
      const double z4 = z3 * z1;
      Mat3 Bz;
      for (int j = 0; j < 3; ++j) {
        Bz(j, 0) = B(0, j) * z3 + B(1, j) * z2 + B(2, j) * z1 + B(3, j);
        Bz(j, 1) = B(4, j) * z3 + B(5, j) * z2 + B(6, j) * z1 + B(7, j);
        Bz(j, 2) = B(8, j) * z4 + B(9, j) * z3 + B(10, j) * z2 + B(11, j) * z1 + B(12, j);
      }
      const SVD s3 = jacobi_svd(to_mat(Bz), false);
      const double X0 = s3.V(0, 2), X1 = s3.V(1, 2), X2 = s3.V(2, 2);
      const double kMaxX3 = 1e-10;
      if (std::abs(X2) < kMaxX3) continue;
      double ev[9];
      const double sx = X0 / X2, sy = X1 / X2;
      for (int k = 0; k < 9; ++k) ev[k] = Eb(k, 0) * sx + Eb(k, 1) * sy + Eb(k, 2) * z1 + Eb(k, 3);
      double nn = 0.0;
      for (int k = 0; k < 9; ++k) nn += ev[k] * ev[k];
      const double norm = std::sqrt(nn);
      M_t model;
      for (int k = 0; k < 9; ++k) model.m[k] = ev[k] / norm;
      models.push_back(model);
    }
#undef B
#undef Eb
    return models;
  }
  static void Residuals(const std::vector<Vec2>& p1, const std::vector<Vec2>& p2, const M_t& E, std::vector<double>* r) {
    ComputeSquaredSampsonError(p1, p2, E, r);
  }
};

// ------------------------------------------------------------------------------------ LO-RANSAC
struct RansacOptions {
  double max_error, min_inlier_ratio, confidence;
  size_t min_num_trials, max_num_trials;
};

template <typename Estimator>
struct Report {
  bool success = false;
  size_t num_trials = 0;
  size_t num_models = 0;  // models scored (sample models + local models), bench bookkeeping only
  Support support;
  std::vector<char> inlier_mask;
  typename Estimator::M_t model;
};

// LORANSAC<Estimator, LocalEstimator>::Estimate, loransac.h:91-233 (+ RANSAC ctor, ransac.h:135-148)
template <typename Estimator, typename LocalEstimator>
static Report<Estimator> LoRansac(RansacOptions options, PRNG* prng, const std::vector<Vec2>& X, const std::vector<Vec2>& Y) {
  {  // RANSAC ctor
    const size_t kNumSamples = 100000;
    const size_t dyn = ComputeNumTrials(static_cast<size_t>(options.min_inlier_ratio * kNumSamples), kNumSamples,
                                        options.confidence, Estimator::kMinNumSamples);
    options.max_num_trials = std::min<size_t>(options.max_num_trials, dyn);
  }
  const size_t num_samples = X.size();
  Report<Estimator> report;
  std::memset(&report.model, 0, sizeof(report.model));  // reference: uninitialised Eigen matrix
  report.success = false;
  report.num_trials = 0;
  if (num_samples < static_cast<size_t>(Estimator::kMinNumSamples)) return report;

  Support best_support;
  typename Estimator::M_t best_model;
  std::memset(&best_model, 0, sizeof(best_model));
  bool best_model_is_local = false;
  bool abort = false;
  const double max_residual = options.max_error * options.max_error;
  std::vector<double> residuals(num_samples);
  std::vector<Vec2> X_inlier, Y_inlier;
  std::vector<Vec2> X_rand(Estimator::kMinNumSamples), Y_rand(Estimator::kMinNumSamples);
  RandomSampler sampler(Estimator::kMinNumSamples, prng);
  sampler.Initialize(num_samples);
  size_t max_num_trials = options.max_num_trials;
  size_t dyn_max_num_trials = max_num_trials;

  for (report.num_trials = 0; report.num_trials < max_num_trials; ++report.num_trials) {
    if (abort) {
      report.num_trials += 1;
      break;
    }
    const std::vector<size_t> idxs = sampler.Sample();
    for (size_t i = 0; i < X_rand.size(); ++i) {
      X_rand[i] = X[idxs[i]];
      Y_rand[i] = Y[idxs[i]];
    }
    const std::vector<typename Estimator::M_t> sample_models = Estimator::Estimate(X_rand, Y_rand);
    for (const auto& sample_model : sample_models) {
      Estimator::Residuals(X, Y, sample_model, &residuals);
      report.num_models += 1;
      const Support support = EvaluateSupport(residuals, max_residual);
      if (CompareSupport(support, best_support)) {
        best_support = support;
        best_model = sample_model;
        best_model_is_local = false;
        if (support.num_inliers > static_cast<size_t>(Estimator::kMinNumSamples) &&
            support.num_inliers >= static_cast<size_t>(LocalEstimator::kMinNumSamples)) {
          X_inlier.clear();
          Y_inlier.clear();
          for (size_t i = 0; i < residuals.size(); ++i) {
            if (residuals[i] <= max_residual) {
              X_inlier.push_back(X[i]);
              Y_inlier.push_back(Y[i]);
            }
          }
          const std::vector<typename LocalEstimator::M_t> local_models = LocalEstimator::Estimate(X_inlier, Y_inlier);
          for (const auto& local_model : local_models) {
            LocalEstimator::Residuals(X, Y, local_model, &residuals);
            report.num_models += 1;
            const Support local_support = EvaluateSupport(residuals, max_residual);
            if (CompareSupport(local_support, best_support)) {
              best_support = local_support;
              best_model = local_model;
              best_model_is_local = true;
            }
          }
        }
        dyn_max_num_trials = ComputeNumTrials(best_support.num_inliers, num_samples, options.confidence,
                                              Estimator::kMinNumSamples);
      }
      if (report.num_trials >= dyn_max_num_trials && report.num_trials >= options.min_num_trials) {
        abort = true;
        break;
      }
    }
  }
  report.support = best_support;
  report.model = best_model;
  if (report.support.num_inliers < static_cast<size_t>(Estimator::kMinNumSamples)) return report;
  report.success = true;
  if (best_model_is_local)
    LocalEstimator::Residuals(X, Y, report.model, &residuals);
  else
    Estimator::Residuals(X, Y, report.model, &residuals);
  report.inlier_mask.resize(num_samples);
  for (size_t i = 0; i < residuals.size(); ++i) report.inlier_mask[i] = residuals[i] <= max_residual ? 1 : 0;
  return report;
}

// ------------------------------------------------------------------------------------ camera
// All eleven camera models of /root/reference/src/base/camera_models.h:187-349.  kTwoFocal = focal_length_idxs has
// two entries (fx, fy, cx, cy, extra from 4); otherwise (f, cx, cy, extra from 3).
static bool CameraModelExists(int id) { return id >= 0 && id <= 10; }
static bool TwoFocal(int id) { return id == 1 || id == 4 || id == 5 || id == 6 || id == 7 || id == 10; }
static void CheckModel(const dsm_camera& cam) {  // ExistsCameraModelWithId is a CHECK in the reference (camera.cc:52)
  if (!CameraModelExists(cam.model_id)) {
    std::fprintf(stderr, "oracle: camera model %d does not exist\n", cam.model_id);
    std::abort();
  }
}

// <Model>::Distortion(extra_params, u, v, &du, &dv)
static void Distortion(int id, const double* e, double u, double v, double* du, double* dv) {
  const double eps = std::numeric_limits<double>::epsilon();
  switch (id) {
    case 2: {  // SimpleRadialCameraModel::Distortion, camera_models.h:747-757
      const double k = e[0];
      const double u2 = u * u, v2 = v * v;
      const double r2 = u2 + v2;
      const double radial = k * r2;
      *du = u * radial;
      *dv = v * radial;
      break;
    }
    case 3: {  // RadialCameraModel::Distortion, :810-822
      const double k1 = e[0], k2 = e[1];
      const double u2 = u * u, v2 = v * v;
      const double r2 = u2 + v2;
      const double radial = k1 * r2 + k2 * r2 * r2;
      *du = u * radial;
      *dv = v * radial;
      break;
    }
    case 4: {  // OpenCVCameraModel::Distortion, :881-897
      const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3];
      const double u2 = u * u, uv = u * v, v2 = v * v;
      const double r2 = u2 + v2;
      const double radial = k1 * r2 + k2 * r2 * r2;
      *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2);
      *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2);
      break;
    }
    case 5: {  // OpenCVFisheyeCameraModel::Distortion, :957-982
      const double k1 = e[0], k2 = e[1], k3 = e[2], k4 = e[3];
      const double r = std::sqrt(u * u + v * v);
      if (r > eps) {
        const double theta = dsm_atan(r);
        const double theta2 = theta * theta;
        const double theta4 = theta2 * theta2;
        const double theta6 = theta4 * theta2;
        const double theta8 = theta4 * theta4;
        const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
        *du = u * thetad / r - u;
        *dv = v * thetad / r - v;
      } else {
        *du = 0;
        *dv = 0;
      }
      break;
    }
    case 6: {  // FullOpenCVCameraModel::Distortion, :1053-1077
      const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3], k3 = e[4], k4 = e[5], k5 = e[6], k6 = e[7];
      const double u2 = u * u, uv = u * v, v2 = v * v;
      const double r2 = u2 + v2;
      const double r4 = r2 * r2;
      const double r6 = r4 * r2;
      const double radial = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6);
      *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) - u;
      *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) - v;
      break;
    }
    case 8: {  // SimpleRadialFisheyeCameraModel::Distortion, :1278-1297
      const double k = e[0];
      const double r = std::sqrt(u * u + v * v);
      if (r > eps) {
        const double theta = dsm_atan(r);
        const double theta2 = theta * theta;
        const double thetad = theta * (1.0 + k * theta2);
        *du = u * thetad / r - u;
        *dv = v * thetad / r - v;
      } else {
        *du = 0;
        *dv = 0;
      }
      break;
    }
    case 9: {  // RadialFisheyeCameraModel::Distortion, :1358-1380
      const double k1 = e[0], k2 = e[1];
      const double r = std::sqrt(u * u + v * v);
      if (r > eps) {
        const double theta = dsm_atan(r);
        const double theta2 = theta * theta;
        const double theta4 = theta2 * theta2;
        const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4);
        *du = u * thetad / r - u;
        *dv = v * thetad / r - v;
      } else {
        *du = 0;
        *dv = 0;
      }
      break;
    }
    case 10: {  // ThinPrismFisheyeCameraModel::Distortion, :1459-1481
      const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3], k3 = e[4], k4 = e[5], sx1 = e[6], sy1 = e[7];
      const double u2 = u * u, uv = u * v, v2 = v * v;
      const double r2 = u2 + v2;
      const double r4 = r2 * r2;
      const double r6 = r4 * r2;
      const double r8 = r6 * r2;
      const double radial = k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8;
      *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) + sx1 * r2;
      *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) + sy1 * r2;
      break;
    }
    default:
      *du = 0;
      *dv = 0;
  }
}
// BaseCameraModel::IterativeUndistortion, camera_models.h:547-587
static void IterativeUndistortion(int id, const double* params, double* u, double* v) {
  const size_t kNumIterations = 100;
  const double kMaxStepNorm = 1e-10;
  const double kRelStepSize = 1e-6;
  const double x0_0 = *u, x0_1 = *v;
  double x_0 = *u, x_1 = *v;
  for (size_t i = 0; i < kNumIterations; ++i) {
    const double step0 = std::max(std::numeric_limits<double>::epsilon(), std::abs(kRelStepSize * x_0));
    const double step1 = std::max(std::numeric_limits<double>::epsilon(), std::abs(kRelStepSize * x_1));
    double dx0, dx1, b00, b01, f00, f01, b10, b11, f10, f11;
    Distortion(id, params, x_0, x_1, &dx0, &dx1);
    Distortion(id, params, x_0 - step0, x_1, &b00, &b01);
    Distortion(id, params, x_0 + step0, x_1, &f00, &f01);
    Distortion(id, params, x_0, x_1 - step1, &b10, &b11);
    Distortion(id, params, x_0, x_1 + step1, &f10, &f11);
    const double J00 = 1 + (f00 - b00) / (2 * step0);
    const double J01 = (f10 - b10) / (2 * step1);
    const double J10 = (f01 - b01) / (2 * step0);
    const double J11 = 1 + (f11 - b11) / (2 * step1);
    // J.inverse() (2x2: adjugate times 1/det), times (x + dx - x0)
    const double invdet = 1.0 / (J00 * J11 - J10 * J01);
    const double i00 = J11 * invdet, i01 = -J01 * invdet, i10 = -J10 * invdet, i11 = J00 * invdet;
    const double r0 = x_0 + dx0 - x0_0, r1 = x_1 + dx1 - x0_1;
    const double s0 = i00 * r0 + i01 * r1;
    const double s1 = i10 * r0 + i11 * r1;
    x_0 -= s0;
    x_1 -= s1;
    if (s0 * s0 + s1 * s1 < kMaxStepNorm) break;
  }
  *u = x_0;
  *v = x_1;
}
// FOVCameraModel::Undistortion, camera_models.h:1179-1218
static void FOVUndistortion(const double* e, double u, double v, double* du, double* dv) {
  const double omega = e[0];
  const double kEpsilon = 1e-4;
  const double radius2 = u * u + v * v;
  const double omega2 = omega * omega;
  double factor;
  if (omega2 < kEpsilon) {
    factor = (omega2 * radius2) / 3.0 - omega2 / 12.0 + 1.0;
  } else if (radius2 < kEpsilon) {
    factor = (omega * (omega * omega * radius2 + 3.0)) / (6.0 * dsm_tan(omega / 2.0));
  } else {
    const double radius = std::sqrt(radius2);
    const double numerator = dsm_tan(radius * omega);
    factor = numerator / (radius * 2.0 * dsm_tan(omega / 2.0));
  }
  *du = u * factor;
  *dv = v * factor;
}

// Camera::ImageToWorld -> <Model>::ImageToWorld
static Vec2 ImageToWorld(const dsm_camera& cam, const Vec2& p) {
  CheckModel(cam);
  Vec2 w;
  const int id = cam.model_id;
  if (TwoFocal(id)) {
    const double f1 = cam.params[0], f2 = cam.params[1], c1 = cam.params[2], c2 = cam.params[3];
    if (id == 7) {  // FOVCameraModel::ImageToWorld, :1126-1141
      const double uu = (p.x - c1) / f1;
      const double vv = (p.y - c2) / f2;
      FOVUndistortion(&cam.params[4], uu, vv, &w.x, &w.y);
      return w;
    }
    w.x = (p.x - c1) / f1;
    w.y = (p.y - c2) / f2;
    if (id == 1) return w;  // PINHOLE, :679-689
    IterativeUndistortion(id, &cam.params[4], &w.x, &w.y);
    if (id == 10) {  // ThinPrismFisheyeCameraModel::ImageToWorld, :1434-1456
      const double theta = std::sqrt(w.x * w.x + w.y * w.y);
      const double theta_cos_theta = theta * dsm_cos(theta);
      if (theta_cos_theta > std::numeric_limits<double>::epsilon()) {
        const double scale = dsm_sin(theta) / theta_cos_theta;
        w.x *= scale;
        w.y *= scale;
      }
    }
    return w;
  }
  const double f = cam.params[0], c1 = cam.params[1], c2 = cam.params[2];
  w.x = (p.x - c1) / f;
  w.y = (p.y - c2) / f;
  if (id == 0) return w;  // SIMPLE_PINHOLE, :629-637
  IterativeUndistortion(id, &cam.params[3], &w.x, &w.y);
  return w;
}
static double ImageToWorldThreshold(const dsm_camera& cam, double threshold) {  // camera_models.h:535-543
  CheckModel(cam);
  double mean_focal_length = 0;
  if (TwoFocal(cam.model_id)) {
    mean_focal_length += cam.params[0];
    mean_focal_length += cam.params[1];
    mean_focal_length /= 2;
  } else {
    mean_focal_length += cam.params[0];
    mean_focal_length /= 1;
  }
  return threshold / mean_focal_length;
}
static Mat3 CalibrationMatrix(const dsm_camera& cam) {  // camera.cc:75-93
  CheckModel(cam);
  Mat3 K;
  for (int i = 0; i < 9; ++i) K.m[i] = 0;
  K(0, 0) = K(1, 1) = K(2, 2) = 1;
  if (TwoFocal(cam.model_id)) {
    K(0, 0) = cam.params[0]; K(1, 1) = cam.params[1];
    K(0, 2) = cam.params[2]; K(1, 2) = cam.params[3];
  } else {
    K(0, 0) = cam.params[0]; K(1, 1) = cam.params[0];
    K(0, 2) = cam.params[1]; K(1, 2) = cam.params[2];
  }
  return K;
}

// ------------------------------------------------------------------------------------ pose
struct P34 {
  double m[12];
  double operator()(int r, int c) const { return m[r * 4 + c]; }
  double& operator()(int r, int c) { return m[r * 4 + c]; }
};
static P34 ComposeProjectionMatrix(const Mat3& R, const double t[3]) {
  P34 P;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) P(r, c) = R(r, c);
    P(r, 3) = t[r];
  }
  return P;
}
static P34 IdentityProjection() {
  P34 P;
  for (int i = 0; i < 12; ++i) P.m[i] = 0;
  P(0, 0) = P(1, 1) = P(2, 2) = 1;
  return P;
}
// TriangulatePoint, triangulation.cc:39-52
static void TriangulatePoint(const P34& P1, const P34& P2, const Vec2& p1, const Vec2& p2, double X[3]) {
  Mat A(4, 4);
  for (int c = 0; c < 4; ++c) {
    A(0, c) = p1.x * P1(2, c) - P1(0, c);
    A(1, c) = p1.y * P1(2, c) - P1(1, c);
    A(2, c) = p2.x * P2(2, c) - P2(0, c);
    A(3, c) = p2.y * P2(2, c) - P2(1, c);
  }
  const SVD svd = jacobi_svd(A, false);
  const double w = svd.V(3, 3);
  X[0] = svd.V(0, 3) / w;
  X[1] = svd.V(1, 3) / w;
  X[2] = svd.V(2, 3) / w;
}
static double CalculateDepth(const P34& P, const double X[3]) {  // projection.cc:193-197
  const double proj_z = P(2, 0) * X[0] + P(2, 1) * X[1] + P(2, 2) * X[2] + P(2, 3) * 1.0;
  return proj_z * std::sqrt(P(0, 2) * P(0, 2) + P(1, 2) * P(1, 2) + P(2, 2) * P(2, 2));
}
// CheckCheirality, pose.cc:225-247
static bool CheckCheirality(const Mat3& R, const double t[3], const std::vector<Vec2>& points1,
                            const std::vector<Vec2>& points2, std::vector<Vec3>* points3D) {
  const P34 P1 = IdentityProjection();
  const P34 P2 = ComposeProjectionMatrix(R, t);
  const double kMinDepth = std::numeric_limits<double>::epsilon();
  double rt[3];
  for (int i = 0; i < 3; ++i) rt[i] = R(0, i) * t[0] + R(1, i) * t[1] + R(2, i) * t[2];
  const double max_depth = 1000.0f * std::sqrt(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]);
  points3D->clear();
  for (size_t i = 0; i < points1.size(); ++i) {
    Vec3 X;
    TriangulatePoint(P1, P2, points1[i], points2[i], X.v);
    const double depth1 = CalculateDepth(P1, X.v);
    if (depth1 > kMinDepth && depth1 < max_depth) {
      const double depth2 = CalculateDepth(P2, X.v);
      if (depth2 > kMinDepth && depth2 < max_depth) points3D->push_back(X);
    }
  }
  return !points3D->empty();
}
// DecomposeEssentialMatrix, base/essential_matrix.cc:41-62
static void DecomposeEssentialMatrix(const Mat3& E, Mat3* R1, Mat3* R2, double t[3]) {
  const SVD svd = jacobi_svd(to_mat(E), true);
  Mat3 U = to_mat3(svd.U);
  Mat3 V = mat3_transpose(to_mat3(svd.V));
  if (mat3_det(U) < 0)
    for (int i = 0; i < 9; ++i) U.m[i] *= -1;
  if (mat3_det(V) < 0)
    for (int i = 0; i < 9; ++i) V.m[i] *= -1;
  Mat3 W;
  const double w[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) W.m[i] = w[i];
  *R1 = mat3_mul(mat3_mul(U, W), V);
  *R2 = mat3_mul(mat3_mul(U, mat3_transpose(W)), V);
  const double n2 = U(0, 2) * U(0, 2) + U(1, 2) * U(1, 2) + U(2, 2) * U(2, 2);
  if (n2 > 0) {
    const double nn = std::sqrt(n2);
    for (int i = 0; i < 3; ++i) t[i] = U(i, 2) / nn;
  } else {
    for (int i = 0; i < 3; ++i) t[i] = U(i, 2);
  }
}
// PoseFromEssentialMatrix, base/essential_matrix.cc:64-89
static void PoseFromEssentialMatrix(const Mat3& E, const std::vector<Vec2>& points1, const std::vector<Vec2>& points2,
                                    Mat3* R, double t[3], std::vector<Vec3>* points3D) {
  Mat3 R1, R2;
  DecomposeEssentialMatrix(E, &R1, &R2, t);
  const Mat3 R_cmbs[4] = {R1, R2, R1, R2};
  const double t_cmbs[4][3] = {{t[0], t[1], t[2]}, {t[0], t[1], t[2]}, {-t[0], -t[1], -t[2]}, {-t[0], -t[1], -t[2]}};
  points3D->clear();
  for (int i = 0; i < 4; ++i) {
    std::vector<Vec3> cmb;
    CheckCheirality(R_cmbs[i], t_cmbs[i], points1, points2, &cmb);
    if (cmb.size() >= points3D->size()) {
      *R = R_cmbs[i];
      for (int k = 0; k < 3; ++k) t[k] = t_cmbs[i][k];
      *points3D = cmb;
    }
  }
}
static double ComputeOppositeOfMinor(const Mat3& M, int row, int col) {  // base/homography_matrix.cc:45-53
  const int col1 = col == 0 ? 1 : 0;
  const int col2 = col == 2 ? 1 : 2;
  const int row1 = row == 0 ? 1 : 0;
  const int row2 = row == 2 ? 1 : 2;
  return (M(row1, col2) * M(row2, col1) - M(row1, col1) * M(row2, col2));
}
static int SignOfNumber(double v) { return (0.0 < v) - (v < 0.0); }  // util/math.h SignOfNumber
static void Normalized3(const double a[3], double out[3]) {
  const double n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (n2 > 0) {
    const double n = std::sqrt(n2);
    for (int i = 0; i < 3; ++i) out[i] = a[i] / n;
  } else {
    for (int i = 0; i < 3; ++i) out[i] = a[i];
  }
}
static Mat3 ComputeHomographyRotation(const Mat3& Hn, const double tstar[3], const double n[3], double v) {
  Mat3 M;
  const double s = 2.0 / v;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M(i, j) = (i == j ? 1.0 : 0.0) - (s * tstar[i]) * n[j];
  return mat3_mul(Hn, M);
}
// DecomposeHomographyMatrix, base/homography_matrix.cc:65-165
static void DecomposeHomographyMatrix(const Mat3& H, const Mat3& K1, const Mat3& K2, std::vector<Mat3>* R,
                                      std::vector<Vec3>* t, std::vector<Vec3>* n) {
  Mat3 Hn = mat3_mul(mat3_mul(mat3_inverse(K2), H), K1);
  const SVD svd = jacobi_svd(to_mat(Hn), false);
  const double s1 = svd.sv[1];
  for (int i = 0; i < 9; ++i) Hn.m[i] /= s1;
  Mat3 S = mat3_mul(mat3_transpose(Hn), Hn);
  S(0, 0) -= 1; S(1, 1) -= 1; S(2, 2) -= 1;
  const double kMinInfinityNorm = 1e-3;
  double inf_norm = 0;
  for (int i = 0; i < 9; ++i) inf_norm = std::max(inf_norm, std::abs(S.m[i]));
  if (inf_norm < kMinInfinityNorm) {
    *R = {Hn};
    *t = {Vec3{{0, 0, 0}}};
    *n = {Vec3{{0, 0, 0}}};
    return;
  }
  const double M00 = ComputeOppositeOfMinor(S, 0, 0), M11 = ComputeOppositeOfMinor(S, 1, 1),
               M22 = ComputeOppositeOfMinor(S, 2, 2);
  const double rtM00 = std::sqrt(M00), rtM11 = std::sqrt(M11), rtM22 = std::sqrt(M22);
  const double M01 = ComputeOppositeOfMinor(S, 0, 1), M12 = ComputeOppositeOfMinor(S, 1, 2),
               M02 = ComputeOppositeOfMinor(S, 0, 2);
  const int e12 = SignOfNumber(M12), e02 = SignOfNumber(M02), e01 = SignOfNumber(M01);
  const double nS[3] = {std::abs(S(0, 0)), std::abs(S(1, 1)), std::abs(S(2, 2))};
  int idx = 0;
  if (nS[1] > nS[idx]) idx = 1;
  if (nS[2] > nS[idx]) idx = 2;
  double np1[3], np2[3];
  if (idx == 0) {
    np1[0] = S(0, 0); np2[0] = S(0, 0);
    np1[1] = S(0, 1) + rtM22; np2[1] = S(0, 1) - rtM22;
    np1[2] = S(0, 2) + e12 * rtM11; np2[2] = S(0, 2) - e12 * rtM11;
  } else if (idx == 1) {
    np1[0] = S(0, 1) + rtM22; np2[0] = S(0, 1) - rtM22;
    np1[1] = S(1, 1); np2[1] = S(1, 1);
    np1[2] = S(1, 2) - e02 * rtM00; np2[2] = S(1, 2) + e02 * rtM00;
  } else {
    np1[0] = S(0, 2) + e01 * rtM11; np2[0] = S(0, 2) - e01 * rtM11;
    np1[1] = S(1, 2) + rtM00; np2[1] = S(1, 2) - rtM00;
    np1[2] = S(2, 2); np2[2] = S(2, 2);
  }
  const double traceS = S(0, 0) + S(1, 1) + S(2, 2);
  const double v = 2.0 * std::sqrt(1.0 + traceS - M00 - M11 - M22);
  const double ESii = SignOfNumber(S(idx, idx));
  const double r_2 = 2 + traceS + v;
  const double nt_2 = 2 + traceS - v;
  const double r = std::sqrt(r_2);
  const double n_t = std::sqrt(nt_2);
  double n1[3], n2[3];
  Normalized3(np1, n1);
  Normalized3(np2, n2);
  const double half_nt = 0.5 * n_t;
  const double esii_t_r = ESii * r;
  double t1_star[3], t2_star[3];
  for (int i = 0; i < 3; ++i) {
    t1_star[i] = half_nt * (esii_t_r * n2[i] - n_t * n1[i]);
    t2_star[i] = half_nt * (esii_t_r * n1[i] - n_t * n2[i]);
  }
  const Mat3 R1 = ComputeHomographyRotation(Hn, t1_star, n1, v);
  const Mat3 R2 = ComputeHomographyRotation(Hn, t2_star, n2, v);
  Vec3 t1, t2;
  for (int i = 0; i < 3; ++i) {
    t1.v[i] = R1(i, 0) * t1_star[0] + R1(i, 1) * t1_star[1] + R1(i, 2) * t1_star[2];
    t2.v[i] = R2(i, 0) * t2_star[0] + R2(i, 1) * t2_star[1] + R2(i, 2) * t2_star[2];
  }
  *R = {R1, R1, R2, R2};
  *t = {t1, Vec3{{-t1.v[0], -t1.v[1], -t1.v[2]}}, t2, Vec3{{-t2.v[0], -t2.v[1], -t2.v[2]}}};
  *n = {Vec3{{-n1[0], -n1[1], -n1[2]}}, Vec3{{n1[0], n1[1], n1[2]}}, Vec3{{-n2[0], -n2[1], -n2[2]}},
        Vec3{{n2[0], n2[1], n2[2]}}};
}
// PoseFromHomographyMatrix, base/homography_matrix.cc:167-192
static void PoseFromHomographyMatrix(const Mat3& H, const Mat3& K1, const Mat3& K2, const std::vector<Vec2>& points1,
                                     const std::vector<Vec2>& points2, Mat3* R, double t[3], std::vector<Vec3>* points3D) {
  std::vector<Mat3> R_cmbs;
  std::vector<Vec3> t_cmbs, n_cmbs;
  DecomposeHomographyMatrix(H, K1, K2, &R_cmbs, &t_cmbs, &n_cmbs);
  points3D->clear();
  for (size_t i = 0; i < R_cmbs.size(); ++i) {
    std::vector<Vec3> cmb;
    CheckCheirality(R_cmbs[i], t_cmbs[i].v, points1, points2, &cmb);
    if (cmb.size() >= points3D->size()) {
      *R = R_cmbs[i];
      for (int k = 0; k < 3; ++k) t[k] = t_cmbs[i].v[k];
      *points3D = cmb;
    }
  }
}
// CalculateTriangulationAnglesWithPM + Median, triangulation.cc:183-218, math.h:211-229
static double MedianTriangulationAngle(const P34& P1, const P34& P2, const std::vector<Vec3>& points3D) {
  double c1[3], c2[3];
  for (int i = 0; i < 3; ++i) {
    c1[i] = -(P1(0, i) * P1(0, 3) + P1(1, i) * P1(1, 3) + P1(2, i) * P1(2, 3));
    c2[i] = -(P2(0, i) * P2(0, 3) + P2(1, i) * P2(1, 3) + P2(2, i) * P2(2, 3));
  }
  double baseline2 = 0;
  for (int i = 0; i < 3; ++i) baseline2 += (c1[i] - c2[i]) * (c1[i] - c2[i]);
  std::vector<double> angles(points3D.size());
  for (size_t i = 0; i < points3D.size(); ++i) {
    double r1 = 0, r2 = 0;
    for (int k = 0; k < 3; ++k) {
      r1 += (points3D[i].v[k] - c1[k]) * (points3D[i].v[k] - c1[k]);
      r2 += (points3D[i].v[k] - c2[k]) * (points3D[i].v[k] - c2[k]);
    }
    const double ray1 = std::sqrt(r1), ray2 = std::sqrt(r2);
    const double angle = std::abs(std::acos((ray1 * ray1 + ray2 * ray2 - baseline2) / (2 * ray1 * ray2)));
    if (std::isnan(angle))
      angles[i] = 0;
    else
      angles[i] = std::min(angle, M_PI - angle);
  }
  std::vector<double> o = angles;
  std::sort(o.begin(), o.end());
  const size_t mid = o.size() / 2;
  if (o.size() % 2 == 0) return (o[mid] + o[mid - 1]) / 2.0;
  return o[mid];
}

// ------------------------------------------------------------------------------------ two-view
struct TwoView {
  int config = DSM_CONFIG_UNDEFINED;
  Mat3 E, F, H;
  double qvec[4] = {0, 0, 0, 0}, tvec[3] = {0, 0, 0}, tri_angle = 0;
  std::vector<uint32_t> inlier_matches;  // pairs
  uint32_t num_trials[4] = {0, 0, 0, 0}, num_models[4] = {0, 0, 0, 0};
  TwoView() {
    for (int i = 0; i < 9; ++i) E.m[i] = F.m[i] = H.m[i] = 0;
  }
};

static RansacOptions ToRansac(const dsm_two_view_options& o) {
  return RansacOptions{o.max_error, o.min_inlier_ratio, o.confidence, static_cast<size_t>(o.min_num_trials),
                       static_cast<size_t>(o.max_num_trials)};
}

static void ExtractInlierMatches(const uint32_t* matches, size_t n, const std::vector<char>& mask, std::vector<uint32_t>* out) {
  out->clear();
  for (size_t i = 0; i < n; ++i)
    if (mask[i]) {
      out->push_back(matches[2 * i]);
      out->push_back(matches[2 * i + 1]);
    }
}

static bool InBox(const Vec2& p, double minx, double maxx, double miny, double maxy) {
  return p.x >= minx && p.x <= maxx && p.y >= miny && p.y <= maxy;
}

// TwoViewGeometry::DetectWatermark, two_view_geometry.cc:491-555
static bool DetectWatermark(const dsm_camera& camera1, const std::vector<Vec2>& points1, const dsm_camera& camera2,
                            const std::vector<Vec2>& points2, size_t num_inliers, const std::vector<char>& inlier_mask,
                            const dsm_two_view_options& options, PRNG* prng, TwoView* tv) {
  const double diagonal1 = std::sqrt(static_cast<double>(camera1.width * camera1.width + camera1.height * camera1.height));
  const double diagonal2 = std::sqrt(static_cast<double>(camera2.width * camera2.width + camera2.height * camera2.height));
  const double minx1 = options.watermark_border_size * diagonal1;
  const double miny1 = minx1;
  const double maxx1 = camera1.width - minx1;
  const double maxy1 = camera1.height - miny1;
  const double minx2 = options.watermark_border_size * diagonal2;
  const double miny2 = minx2;
  const double maxx2 = camera2.width - minx2;
  const double maxy2 = camera2.height - miny2;
  std::vector<Vec2> inlier_points1(num_inliers), inlier_points2(num_inliers);
  size_t num_matches_in_border = 0;
  size_t j = 0;
  for (size_t i = 0; i < inlier_mask.size(); ++i) {
    if (inlier_mask[i]) {
      inlier_points1[j] = points1[i];
      inlier_points2[j] = points2[i];
      j += 1;
      if (!InBox(points1[i], minx1, maxx1, miny1, maxy1) && !InBox(points2[i], minx2, maxx2, miny2, maxy2))
        num_matches_in_border += 1;
    }
  }
  const double matches_in_border_ratio = static_cast<double>(num_matches_in_border) / num_inliers;
  if (matches_in_border_ratio < options.watermark_min_inlier_ratio) return false;
  RansacOptions ro = ToRansac(options);
  ro.min_inlier_ratio = options.watermark_min_inlier_ratio;
  const auto report = LoRansac<TranslationEstimator, TranslationEstimator>(ro, prng, inlier_points1, inlier_points2);
  tv->num_trials[3] = static_cast<uint32_t>(report.num_trials);
  tv->num_models[3] = static_cast<uint32_t>(report.num_models);
  const double inlier_ratio = static_cast<double>(report.support.num_inliers) / num_inliers;
  return inlier_ratio >= options.watermark_min_inlier_ratio;
}

// TwoViewGeometry::EstimateCalibrated, two_view_geometry.cc:292-425
static void EstimateCalibrated(const dsm_camera& camera1, const std::vector<Vec2>& points1, const dsm_camera& camera2,
                               const std::vector<Vec2>& points2, const uint32_t* matches, size_t n_matches,
                               const dsm_two_view_options& options, PRNG* prng, TwoView* tv) {
  if (n_matches < options.min_num_inliers) {
    tv->config = DSM_CONFIG_DEGENERATE;
    return;
  }
  std::vector<Vec2> mp1(n_matches), mp2(n_matches), mp1n(n_matches), mp2n(n_matches);
  for (size_t i = 0; i < n_matches; ++i) {
    mp1[i] = points1[matches[2 * i]];
    mp2[i] = points2[matches[2 * i + 1]];
    mp1n[i] = ImageToWorld(camera1, mp1[i]);
    mp2n[i] = ImageToWorld(camera2, mp2[i]);
  }
  const RansacOptions ro = ToRansac(options);
  RansacOptions E_ro = ro;
  E_ro.max_error = (ImageToWorldThreshold(camera1, ro.max_error) + ImageToWorldThreshold(camera2, ro.max_error)) / 2;
  const auto E_report = LoRansac<EssentialFivePoint, EssentialFivePoint>(E_ro, prng, mp1n, mp2n);
  tv->E = E_report.model;
  const auto F_report = LoRansac<FundamentalSevenPoint, FundamentalEightPoint>(ro, prng, mp1, mp2);
  tv->F = F_report.model;
  const auto H_report = LoRansac<HomographyEstimator, HomographyEstimator>(ro, prng, mp1, mp2);
  tv->H = H_report.model;
  tv->num_trials[0] = static_cast<uint32_t>(E_report.num_trials);
  tv->num_trials[1] = static_cast<uint32_t>(F_report.num_trials);
  tv->num_trials[2] = static_cast<uint32_t>(H_report.num_trials);
  tv->num_models[0] = static_cast<uint32_t>(E_report.num_models);
  tv->num_models[1] = static_cast<uint32_t>(F_report.num_models);
  tv->num_models[2] = static_cast<uint32_t>(H_report.num_models);

  if ((!E_report.success && !F_report.success && !H_report.success) ||
      (E_report.support.num_inliers < options.min_num_inliers && F_report.support.num_inliers < options.min_num_inliers &&
       H_report.support.num_inliers < options.min_num_inliers)) {
    tv->config = DSM_CONFIG_DEGENERATE;
    return;
  }
  const double E_F_inlier_ratio = static_cast<double>(E_report.support.num_inliers) / F_report.support.num_inliers;
  const double H_F_inlier_ratio = static_cast<double>(H_report.support.num_inliers) / F_report.support.num_inliers;
  const double H_E_inlier_ratio = static_cast<double>(H_report.support.num_inliers) / E_report.support.num_inliers;
  const std::vector<char>* best_inlier_mask = nullptr;
  size_t num_inliers = 0;
  if (E_report.success && E_F_inlier_ratio > options.min_E_F_inlier_ratio &&
      E_report.support.num_inliers >= options.min_num_inliers) {
    if (E_report.support.num_inliers >= F_report.support.num_inliers) {
      num_inliers = E_report.support.num_inliers;
      best_inlier_mask = &E_report.inlier_mask;
    } else {
      num_inliers = F_report.support.num_inliers;
      best_inlier_mask = &F_report.inlier_mask;
    }
    if (H_E_inlier_ratio > options.max_H_inlier_ratio) {
      tv->config = DSM_CONFIG_PLANAR_OR_PANORAMIC;
      if (H_report.support.num_inliers > num_inliers) {
        num_inliers = H_report.support.num_inliers;
        best_inlier_mask = &H_report.inlier_mask;
      }
    } else {
      tv->config = DSM_CONFIG_CALIBRATED;
    }
  } else if (F_report.success && F_report.support.num_inliers >= options.min_num_inliers) {
    num_inliers = F_report.support.num_inliers;
    best_inlier_mask = &F_report.inlier_mask;
    if (H_F_inlier_ratio > options.max_H_inlier_ratio) {
      tv->config = DSM_CONFIG_PLANAR_OR_PANORAMIC;
      if (H_report.support.num_inliers > num_inliers) {
        num_inliers = H_report.support.num_inliers;
        best_inlier_mask = &H_report.inlier_mask;
      }
    } else {
      tv->config = DSM_CONFIG_UNCALIBRATED;
    }
  } else if (H_report.success && H_report.support.num_inliers >= options.min_num_inliers) {
    num_inliers = H_report.support.num_inliers;
    best_inlier_mask = &H_report.inlier_mask;
    tv->config = DSM_CONFIG_PLANAR_OR_PANORAMIC;
  } else {
    tv->config = DSM_CONFIG_DEGENERATE;
    return;
  }
  if (best_inlier_mask != nullptr) {
    ExtractInlierMatches(matches, n_matches, *best_inlier_mask, &tv->inlier_matches);
    if (options.detect_watermark &&
        DetectWatermark(camera1, mp1, camera2, mp2, num_inliers, *best_inlier_mask, options, prng, tv)) {
      tv->config = DSM_CONFIG_WATERMARK;
    }
  }
}

// TwoViewGeometry::EstimateUncalibrated, two_view_geometry.cc:427-489
static void EstimateUncalibrated(const dsm_camera& camera1, const std::vector<Vec2>& points1, const dsm_camera& camera2,
                                 const std::vector<Vec2>& points2, const uint32_t* matches, size_t n_matches,
                                 const dsm_two_view_options& options, PRNG* prng, TwoView* tv) {
  if (n_matches < options.min_num_inliers) {
    tv->config = DSM_CONFIG_DEGENERATE;
    return;
  }
  std::vector<Vec2> mp1(n_matches), mp2(n_matches);
  for (size_t i = 0; i < n_matches; ++i) {
    mp1[i] = points1[matches[2 * i]];
    mp2[i] = points2[matches[2 * i + 1]];
  }
  const RansacOptions ro = ToRansac(options);
  const auto F_report = LoRansac<FundamentalSevenPoint, FundamentalEightPoint>(ro, prng, mp1, mp2);
  tv->F = F_report.model;
  const auto H_report = LoRansac<HomographyEstimator, HomographyEstimator>(ro, prng, mp1, mp2);
  tv->H = H_report.model;
  tv->num_trials[1] = static_cast<uint32_t>(F_report.num_trials);
  tv->num_trials[2] = static_cast<uint32_t>(H_report.num_trials);
  tv->num_models[1] = static_cast<uint32_t>(F_report.num_models);
  tv->num_models[2] = static_cast<uint32_t>(H_report.num_models);
  if ((!F_report.success && !H_report.success) || (F_report.support.num_inliers < options.min_num_inliers &&
                                                    H_report.support.num_inliers < options.min_num_inliers)) {
    tv->config = DSM_CONFIG_DEGENERATE;
    return;
  }
  const double H_F_inlier_ratio = static_cast<double>(H_report.support.num_inliers) / F_report.support.num_inliers;
  if (H_F_inlier_ratio > options.max_H_inlier_ratio)
    tv->config = DSM_CONFIG_PLANAR_OR_PANORAMIC;
  else
    tv->config = DSM_CONFIG_UNCALIBRATED;
  // F_report.inlier_mask is empty when F failed (reference indexes it regardless; with
  // num_inliers < kMinNumSamples no entry can be set, so an all-false mask is equivalent)
  std::vector<char> mask = F_report.inlier_mask;
  if (mask.size() != n_matches) mask.assign(n_matches, 0);
  ExtractInlierMatches(matches, n_matches, mask, &tv->inlier_matches);
  if (options.detect_watermark && F_report.support.num_inliers > 0 &&
      DetectWatermark(camera1, mp1, camera2, mp2, F_report.support.num_inliers, mask, options, prng, tv)) {
    tv->config = DSM_CONFIG_WATERMARK;
  }
}

// TwoViewGeometry::EstimateWithRelativePose, two_view_geometry.cc:232-290
static void EstimateWithRelativePose(const dsm_camera& camera1, const std::vector<Vec2>& points1,
                                     const dsm_camera& camera2, const std::vector<Vec2>& points2, const uint32_t* matches,
                                     size_t n_matches, const dsm_two_view_options& options, PRNG* prng, TwoView* tv) {
  EstimateCalibrated(camera1, points1, camera2, points2, matches, n_matches, options, prng, tv);
  // the reference goes on to decompose an uninitialised H for DEGENERATE pairs and
  // Match() later discards the result (matching.cc:828-831); the oracle stops here instead.
  if (tv->config == DSM_CONFIG_DEGENERATE || tv->config == DSM_CONFIG_UNDEFINED) return;
  std::vector<Vec2> ip1, ip2;
  for (size_t i = 0; i + 1 < tv->inlier_matches.size(); i += 2) {
    ip1.push_back(ImageToWorld(camera1, points1[tv->inlier_matches[i]]));
    ip2.push_back(ImageToWorld(camera2, points2[tv->inlier_matches[i + 1]]));
  }
  Mat3 R;
  for (int i = 0; i < 9; ++i) R.m[i] = 0;
  std::vector<Vec3> points3D;
  if (tv->config == DSM_CONFIG_CALIBRATED || tv->config == DSM_CONFIG_UNCALIBRATED) {
    PoseFromEssentialMatrix(tv->E, ip1, ip2, &R, tv->tvec, &points3D);
  } else {
    PoseFromHomographyMatrix(tv->H, CalibrationMatrix(camera1), CalibrationMatrix(camera2), ip1, ip2, &R, tv->tvec,
                             &points3D);
  }
  rotation_to_quaternion(R, tv->qvec);
  const P34 P1 = IdentityProjection();
  const P34 P2 = ComposeProjectionMatrix(R, tv->tvec);
  if (points3D.empty())
    tv->tri_angle = 0;
  else
    tv->tri_angle = MedianTriangulationAngle(P1, P2, points3D);
  if (tv->config == DSM_CONFIG_PLANAR_OR_PANORAMIC) {
    const double tn = std::sqrt(tv->tvec[0] * tv->tvec[0] + tv->tvec[1] * tv->tvec[1] + tv->tvec[2] * tv->tvec[2]);
    if (tn == 0) {
      tv->config = DSM_CONFIG_PANORAMIC;
      tv->tri_angle = 0;
    } else {
      tv->config = DSM_CONFIG_PLANAR;
    }
  }
}

}  // namespace oracle

// ------------------------------------------------------------------------------------ C exports
using namespace oracle;

static std::vector<Vec2> ToVec2(const double* p, int n) {
  std::vector<Vec2> v(n);
  for (int i = 0; i < n; ++i) v[i] = Vec2{p[2 * i], p[2 * i + 1]};
  return v;
}

extern "C" {

// exact_trig.h on the host, for tests/test_exact_trig.py: kind 0 atan, 1 sin, 2 cos, 3 tan
void oracle_exact_trig(int kind, const double* x, int n, double* out) {
  for (int i = 0; i < n; ++i)
    out[i] = kind == 0 ? dsm_atan(x[i]) : kind == 1 ? dsm_sin(x[i]) : kind == 2 ? dsm_cos(x[i]) : dsm_tan(x[i]);
}

// Steps 3 / 4 of the 5-point solver alone, for tests/test_fivept_reference_order.py (layouts of Eigen's .data()).
void oracle_fivept_build_A(const double* e_colmajor_9x4, double* a_colmajor_10x20) {
  oracle::fivept_build_A(e_colmajor_9x4, a_colmajor_10x20);
}
void oracle_fivept_coeffs(const double* b_colmajor_13x3, double* coeffs11) { oracle::fivept_det_poly(b_colmajor_13x3, coeffs11); }

// TwoViewGeometry::Estimate (two_view_geometry.cc:113-126) for one pair with an explicit PRNG seed.
// points: n x 2 doubles (FeatureKeypointsToPointsVector output); matches: n_matches x 2 uint32.
// inlier_matches_out must hold n_matches x 2 uint32.
void oracle_estimate_two_view_geometry(const dsm_camera* camera1, const double* points1, int n1,
                                       const dsm_camera* camera2, const double* points2, int n2,
                                       const uint32_t* matches, int n_matches, const dsm_two_view_options* options,
                                       uint32_t seed, dsm_two_view_geometry* out, uint32_t* inlier_matches_out) {
  const std::vector<Vec2> p1 = ToVec2(points1, n1), p2 = ToVec2(points2, n2);
  PRNG prng(seed);
  TwoView tv;
  auto estimate = [&](const uint32_t* m, size_t nm, TwoView* t) {  // TwoViewGeometry::Estimate, :113-126
    if (camera1->has_prior_focal_length && camera2->has_prior_focal_length)
      EstimateWithRelativePose(*camera1, p1, *camera2, p2, m, nm, *options, &prng, t);
    else
      EstimateUncalibrated(*camera1, p1, *camera2, p2, m, nm, *options, &prng, t);
  };
  if (!options->multiple_models) {
    estimate(matches, n_matches, &tv);
  } else {
    // TwoViewGeometry::EstimateMultiple, two_view_geometry.cc:128-167.  The reference's PRNG is one thread-local
    // stream, so every pass continues the stream of the previous one.  num_trials / num_models (our own
    // counters) are summed over all passes.
    std::vector<uint32_t> remaining(matches, matches + 2 * static_cast<size_t>(n_matches));
    std::vector<TwoView> found;
    uint32_t trials[4] = {0, 0, 0, 0}, models[4] = {0, 0, 0, 0};
    while (true) {
      TwoView t;
      estimate(remaining.data(), remaining.size() / 2, &t);
      for (int k = 0; k < 4; ++k) {
        trials[k] += t.num_trials[k];
        models[k] += t.num_models[k];
      }
      if (t.config == DSM_CONFIG_DEGENERATE) break;
      if (!(options->multiple_ignore_watermark && t.config == DSM_CONFIG_WATERMARK)) found.push_back(t);
      // ExtractOutlierMatches, :67-88
      std::set<std::pair<uint32_t, uint32_t>> inl;
      for (size_t i = 0; i + 1 < t.inlier_matches.size(); i += 2) inl.emplace(t.inlier_matches[i], t.inlier_matches[i + 1]);
      std::vector<uint32_t> next;
      for (size_t i = 0; i + 1 < remaining.size(); i += 2)
        if (inl.count(std::make_pair(remaining[i], remaining[i + 1])) == 0) {
          next.push_back(remaining[i]);
          next.push_back(remaining[i + 1]);
        }
      // A pass that is not DEGENERATE yet removes no match (reachable with min_num_inliers = 0: F fails on fewer than
      // 7 matches, H succeeds, the inlier list comes from F's empty mask) would be repeated on the same matches for as
      // long as the PRNG keeps H succeeding -- in practice forever.  Oracle and product both stop after recording it.
      const bool no_progress = next.size() == remaining.size();
      remaining.swap(next);
      if (no_progress) break;
    }
    if (found.empty()) {
      tv.config = DSM_CONFIG_DEGENERATE;
    } else if (found.size() == 1) {
      tv = found[0];
    } else {
      tv.config = DSM_CONFIG_MULTIPLE;
      for (const TwoView& t : found) tv.inlier_matches.insert(tv.inlier_matches.end(), t.inlier_matches.begin(), t.inlier_matches.end());
    }
    for (int k = 0; k < 4; ++k) {
      tv.num_trials[k] = trials[k];
      tv.num_models[k] = models[k];
    }
  }
  std::memset(out, 0, sizeof(*out));
  out->config = tv.config;
  out->num_inliers = static_cast<uint32_t>(tv.inlier_matches.size() / 2);
  out->num_matches = static_cast<uint32_t>(n_matches);
  std::memcpy(out->E, tv.E.m, sizeof(out->E));
  std::memcpy(out->F, tv.F.m, sizeof(out->F));
  std::memcpy(out->H, tv.H.m, sizeof(out->H));
  std::memcpy(out->qvec, tv.qvec, sizeof(out->qvec));
  std::memcpy(out->tvec, tv.tvec, sizeof(out->tvec));
  out->tri_angle = tv.tri_angle;
  for (int i = 0; i < 4; ++i) {
    out->num_trials[i] = tv.num_trials[i];
    out->num_models[i] = tv.num_models[i];
  }
  if (!tv.inlier_matches.empty())
    std::memcpy(inlier_matches_out, tv.inlier_matches.data(), tv.inlier_matches.size() * sizeof(uint32_t));
}

// InlierSupportMeasurer::Evaluate / Compare (support_measurement.cc:36-62) for tests/test_oracle_estimators.py, which pins
// them to the reference's own file compiled where it lies (oracle/_ref/libmisc_ref.so)
void oracle_inlier_support(const double* residuals, uint64_t n, double max_residual, uint64_t* num_inliers, double* residual_sum) {
  const Support s = EvaluateSupport(std::vector<double>(residuals, residuals + n), max_residual);
  *num_inliers = s.num_inliers;
  *residual_sum = s.residual_sum;
}
int oracle_inlier_support_compare(uint64_t num_inliers1, double residual_sum1, uint64_t num_inliers2, double residual_sum2) {
  Support a, b;
  a.num_inliers = num_inliers1;
  a.residual_sum = residual_sum1;
  b.num_inliers = num_inliers2;
  b.residual_sum = residual_sum2;
  return CompareSupport(a, b) ? 1 : 0;
}

uint64_t oracle_compute_num_trials(uint64_t num_inliers, uint64_t num_samples, double confidence, int min_samples) {
  return ComputeNumTrials(num_inliers, num_samples, confidence, min_samples);
}

// kind: 0 = 7-point F, 1 = 8-point F, 2 = 8-point E, 3 = homography, 4 = 5-point E.  models_out: up to 10 x 9.
int oracle_estimate_model(int kind, const double* points1, const double* points2, int n, double* models_out) {
  const std::vector<Vec2> p1 = ToVec2(points1, n), p2 = ToVec2(points2, n);
  std::vector<Mat3> models;
  switch (kind) {
    case 0: models = FundamentalSevenPoint::Estimate(p1, p2); break;
    case 1: models = FundamentalEightPoint::Estimate(p1, p2); break;
    case 2: models = {EightPointCore(p1, p2, true)}; break;
    case 3: models = HomographyEstimator::Estimate(p1, p2); break;
    default: models = EssentialFivePoint::Estimate(p1, p2); break;
  }
  for (size_t i = 0; i < models.size(); ++i) std::memcpy(models_out + 9 * i, models[i].m, 9 * sizeof(double));
  return static_cast<int>(models.size());
}

// kind: 0 = Sampson (E/F), 1 = homography transfer
void oracle_residuals(int kind, const double* points1, const double* points2, int n, const double* model, double* out) {
  const std::vector<Vec2> p1 = ToVec2(points1, n), p2 = ToVec2(points2, n);
  Mat3 M;
  std::memcpy(M.m, model, sizeof(M.m));
  std::vector<double> r;
  if (kind == 0)
    ComputeSquaredSampsonError(p1, p2, M, &r);
  else
    HomographyEstimator::Residuals(p1, p2, M, &r);
  std::memcpy(out, r.data(), n * sizeof(double));
}

void oracle_center_and_normalize(const double* points, int n, double* normed_out, double* matrix_out) {
  std::vector<Vec2> nn;
  Mat3 M;
  CenterAndNormalizeImagePoints(ToVec2(points, n), &nn, &M);
  for (int i = 0; i < n; ++i) {
    normed_out[2 * i] = nn[i].x;
    normed_out[2 * i + 1] = nn[i].y;
  }
  std::memcpy(matrix_out, M.m, sizeof(M.m));
}

// FindPolynomialRootsCompanionMatrix; returns number of roots or -1.
int oracle_poly_roots(const double* coeffs, int n, double* real_out, double* imag_out) {
  std::vector<double> re, im;
  if (!FindPolynomialRootsCompanionMatrix(std::vector<double>(coeffs, coeffs + n), &re, &im)) return -1;
  for (size_t i = 0; i < re.size(); ++i) {
    real_out[i] = re[i];
    imag_out[i] = im[i];
  }
  return static_cast<int>(re.size());
}

// JacobiSVD of a row-major rows x cols matrix; U (rows x rows) and V (cols x cols) row-major out.
void oracle_jacobi_svd(const double* a, int rows, int cols, double* U, double* S, double* V) {
  Mat A(rows, cols);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) A(i, j) = a[i * cols + j];
  const SVD svd = jacobi_svd(A, true);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < rows; ++j) U[i * rows + j] = svd.U(i, j);
  for (int i = 0; i < cols; ++i)
    for (int j = 0; j < cols; ++j) V[i * cols + j] = svd.V(i, j);
  for (size_t i = 0; i < svd.sv.size(); ++i) S[i] = svd.sv[i];
}

int oracle_eigenvalues(const double* a, int n, double* re_out, double* im_out) {
  Mat A(n, n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A(i, j) = a[i * n + j];
  std::vector<double> re, im;
  if (!real_eigenvalues(A, &re, &im)) return -1;
  for (int i = 0; i < n; ++i) {
    re_out[i] = re[i];
    im_out[i] = im[i];
  }
  return n;
}

// LO-RANSAC over one family with an explicit seed.  family: 0 = E (5pt/5pt), 1 = F (7pt/8pt),
// 2 = H, 3 = translation.  Returns num_inliers; mask_out n chars; model_out 9 doubles (2 for translation).
uint64_t oracle_loransac(int family, const double* points1, const double* points2, int n, double max_error,
                         double min_inlier_ratio, double confidence, uint64_t min_trials, uint64_t max_trials, uint32_t seed,
                         int* success, uint64_t* num_trials, char* mask_out, double* model_out) {
  const std::vector<Vec2> p1 = ToVec2(points1, n), p2 = ToVec2(points2, n);
  PRNG prng(seed);
  const RansacOptions ro{max_error, min_inlier_ratio, confidence, static_cast<size_t>(min_trials), static_cast<size_t>(max_trials)};
  auto fill = [&](bool ok, size_t trials, const std::vector<char>& mask, const double* m, int nm, size_t ninl) {
    *success = ok ? 1 : 0;
    *num_trials = trials;
    for (size_t i = 0; i < mask.size(); ++i) mask_out[i] = mask[i];
    for (int i = 0; i < nm; ++i) model_out[i] = m[i];
    return static_cast<uint64_t>(ninl);
  };
  if (family == 0) {
    const auto r = LoRansac<EssentialFivePoint, EssentialFivePoint>(ro, &prng, p1, p2);
    return fill(r.success, r.num_trials, r.inlier_mask, r.model.m, 9, r.support.num_inliers);
  } else if (family == 1) {
    const auto r = LoRansac<FundamentalSevenPoint, FundamentalEightPoint>(ro, &prng, p1, p2);
    return fill(r.success, r.num_trials, r.inlier_mask, r.model.m, 9, r.support.num_inliers);
  } else if (family == 2) {
    const auto r = LoRansac<HomographyEstimator, HomographyEstimator>(ro, &prng, p1, p2);
    return fill(r.success, r.num_trials, r.inlier_mask, r.model.m, 9, r.support.num_inliers);
  }
  const auto r = LoRansac<TranslationEstimator, TranslationEstimator>(ro, &prng, p1, p2);
  const double m[2] = {r.model.x, r.model.y};
  return fill(r.success, r.num_trials, r.inlier_mask, m, 2, r.support.num_inliers);
}

// Pose helpers for the reference's base/*_test.cc known answers.
void oracle_decompose_essential(const double* E, double* R1, double* R2, double* t) {
  Mat3 Em, r1, r2;
  std::memcpy(Em.m, E, sizeof(Em.m));
  DecomposeEssentialMatrix(Em, &r1, &r2, t);
  std::memcpy(R1, r1.m, sizeof(r1.m));
  std::memcpy(R2, r2.m, sizeof(r2.m));
}
int oracle_pose_from_essential(const double* E, const double* points1, const double* points2, int n, double* R, double* t) {
  Mat3 Em, Rm;
  std::memcpy(Em.m, E, sizeof(Em.m));
  std::vector<Vec3> pts;
  PoseFromEssentialMatrix(Em, ToVec2(points1, n), ToVec2(points2, n), &Rm, t, &pts);
  std::memcpy(R, Rm.m, sizeof(Rm.m));
  return static_cast<int>(pts.size());
}
int oracle_decompose_homography(const double* H, const double* K1, const double* K2, double* R_out, double* t_out, double* n_out) {
  Mat3 Hm, K1m, K2m;
  std::memcpy(Hm.m, H, 72);
  std::memcpy(K1m.m, K1, 72);
  std::memcpy(K2m.m, K2, 72);
  std::vector<Mat3> R;
  std::vector<Vec3> t, n;
  DecomposeHomographyMatrix(Hm, K1m, K2m, &R, &t, &n);
  for (size_t i = 0; i < R.size(); ++i) {
    std::memcpy(R_out + 9 * i, R[i].m, 72);
    std::memcpy(t_out + 3 * i, t[i].v, 24);
    std::memcpy(n_out + 3 * i, n[i].v, 24);
  }
  return static_cast<int>(R.size());
}
void oracle_triangulate_point(const double* P1, const double* P2, const double* p1, const double* p2, double* X) {
  P34 a, b;
  std::memcpy(a.m, P1, sizeof(a.m));
  std::memcpy(b.m, P2, sizeof(b.m));
  TriangulatePoint(a, b, Vec2{p1[0], p1[1]}, Vec2{p2[0], p2[1]}, X);
}
void oracle_rotation_to_quaternion(const double* R, double* q) {
  Mat3 Rm;
  std::memcpy(Rm.m, R, sizeof(Rm.m));
  rotation_to_quaternion(Rm, q);
}
void oracle_image_to_world(const dsm_camera* cam, const double* p, double* w) {
  const Vec2 r = ImageToWorld(*cam, Vec2{p[0], p[1]});
  w[0] = r.x;
  w[1] = r.y;
}
double oracle_image_to_world_threshold(const dsm_camera* cam, double threshold) { return ImageToWorldThreshold(*cam, threshold); }
// Sample sequence of RandomSampler for tests of the device MT19937 + Lemire mapping.
void oracle_sample_sequence(uint32_t seed, uint32_t k, uint32_t total, uint32_t n_draws, uint32_t* out) {
  PRNG prng(seed);
  RandomSampler s(k, &prng);
  s.Initialize(total);
  for (uint32_t d = 0; d < n_draws; ++d) {
    const std::vector<size_t> idx = s.Sample();
    for (uint32_t i = 0; i < k; ++i) out[d * k + i] = static_cast<uint32_t>(idx[i]);
  }
}

}  // extern "C"
