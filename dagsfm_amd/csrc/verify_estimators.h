// verify_estimators.h -- device-side two-view model estimators and residuals.
//
// Minimal solvers run one hypothesis per lane (private memory); the local-optimisation
// estimators (all inliers) run once per wave through the cooperative SVD in verify_linalg.h and
// finish on lane 0.  Each function cites the reference code it replaces.
#ifndef DAGSFM_AMD_CSRC_VERIFY_ESTIMATORS_H_
#define DAGSFM_AMD_CSRC_VERIFY_ESTIMATORS_H_

#include "verify_linalg.h"

// ---------------------------------------------------------------------------------- residuals
// ComputeSquaredSampsonError, /root/reference/src/estimators/utils.cc:87-131 (one correspondence)
DSM_DEV double sampson_residual(const double* E, double x1_0, double x1_1, double x2_0, double x2_1) {
  const double Ex1_0 = E[0] * x1_0 + E[1] * x1_1 + E[2];
  const double Ex1_1 = E[3] * x1_0 + E[4] * x1_1 + E[5];
  const double Ex1_2 = E[6] * x1_0 + E[7] * x1_1 + E[8];
  const double Etx2_0 = E[0] * x2_0 + E[3] * x2_1 + E[6];
  const double Etx2_1 = E[1] * x2_0 + E[4] * x2_1 + E[7];
  const double x2tEx1 = x2_0 * Ex1_0 + x2_1 * Ex1_1 + Ex1_2;
  return x2tEx1 * x2tEx1 / (Ex1_0 * Ex1_0 + Ex1_1 * Ex1_1 + Etx2_0 * Etx2_0 + Etx2_1 * Etx2_1);
}
// HomographyMatrixEstimator::Residuals, /root/reference/src/estimators/homography_matrix.cc:94-131
DSM_DEV double homography_residual(const double* H, double s_0, double s_1, double d_0, double d_1) {
  const double pd_0 = H[0] * s_0 + H[1] * s_1 + H[2];
  const double pd_1 = H[3] * s_0 + H[4] * s_1 + H[5];
  const double pd_2 = H[6] * s_0 + H[7] * s_1 + H[8];
  const double inv_pd_2 = 1.0 / pd_2;
  const double dd_0 = d_0 - pd_0 * inv_pd_2;
  const double dd_1 = d_1 - pd_1 * inv_pd_2;
  return dd_0 * dd_0 + dd_1 * dd_1;
}
// TranslationTransformEstimator<2>::Residuals, /root/reference/src/estimators/translation_transform.h:106-118
DSM_DEV double translation_residual(const double* t, double x1_0, double x1_1, double x2_0, double x2_1) {
  const double ex = x2_0 - x1_0 - t[0];
  const double ey = x2_1 - x1_1 - t[1];
  return ex * ex + ey * ey;
}

// ---------------------------------------------------------------------------------- normalisation
// CenterAndNormalizeImagePoints, utils.cc:38-85, over points pts[idx(i)] for i < n (4 doubles
// per correspondence: x1 y1 x2 y2; `which` selects image 1 or 2).  Sequential sums; returns the
// matrix entries (norm_factor, -nf*cx, -nf*cy).
template <typename IdxFn>
DSM_DEV void center_and_normalize(const double* pts, int which, int n, IdxFn idx, double* nf, double* m02, double* m12) {
  double cx = 0, cy = 0;
  for (int i = 0; i < n; ++i) {
    const double* p = pts + (size_t)idx(i) * 4 + which * 2;
    cx += p[0];
    cy += p[1];
  }
  cx /= n;
  cy /= n;
  double rms = 0;
  for (int i = 0; i < n; ++i) {
    const double* p = pts + (size_t)idx(i) * 4 + which * 2;
    const double dx = p[0] - cx, dy = p[1] - cy;
    rms += dx * dx + dy * dy;
  }
  rms = sqrt(rms / n);
  const double norm_factor = sqrt(2.0) / rms;
  *nf = norm_factor;
  *m02 = -norm_factor * cx;
  *m12 = -norm_factor * cy;
}
// Same, by the whole wave (in-order sums through wv_seq_sum); every lane gets the same result.
template <typename IdxFn>
DSM_DEV void wv_center_and_normalize(const double* pts, int which, int n, IdxFn idx, int lane, double* nf, double* m02,
                                     double* m12) {
  double cx = wv_seq_sum(0.0, n, lane, [pts, which, idx](int i) { return pts[(size_t)idx(i) * 4 + which * 2]; });
  double cy = wv_seq_sum(0.0, n, lane, [pts, which, idx](int i) { return pts[(size_t)idx(i) * 4 + which * 2 + 1]; });
  cx /= n;
  cy /= n;
  double rms = wv_seq_sum(0.0, n, lane, [pts, which, idx, cx, cy](int i) {
    const double* p = pts + (size_t)idx(i) * 4 + which * 2;
    const double dx = p[0] - cx, dy = p[1] - cy;
    return dx * dx + dy * dy;
  });
  rms = sqrt(rms / n);
  const double norm_factor = sqrt(2.0) / rms;
  *nf = norm_factor;
  *m02 = -norm_factor * cx;
  *m12 = -norm_factor * cy;
}
// applies M = [[nf,0,m02],[0,nf,m12],[0,0,1]] exactly like utils.cc:66-84
DSM_DEV void apply_norm(double nf, double m02, double m12, double p_0, double p_1, double* o0, double* o1) {
  const double np_0 = nf * p_0 + 0.0 * p_1 + m02;
  const double np_1 = 0.0 * p_0 + nf * p_1 + m12;
  const double np_2 = 0.0 * p_0 + 0.0 * p_1 + 1.0;
  const double inv_np_2 = 1.0 / np_2;
  *o0 = np_0 * inv_np_2;
  *o1 = np_1 * inv_np_2;
}

// ---------------------------------------------------------------------------------- 7-point F
// FundamentalMatrixSevenPointEstimator::Estimate, /root/reference/src/estimators/fundamental_matrix.cc:47-142
// xs: 7 correspondences (x1 y1 x2 y2).  Returns the number of models written to models[k*9].
// ES > 1: the 9 x 7 working matrix lives in `At_ext` with element stride ES (lane-interleaved LDS of the batch
// kernel: its pivoted QR indexes it dynamically, which would otherwise put it in scratch memory).
template <int ES>
DSM_DEV int seven_point_t(const double* xs, double* models, double* At_ext) {
  double At_loc[ES == 1 ? 63 : 1];  // A^T, 9 x 7 column-major: At[i*9 + c] = A(i, c)
  double* At;
  if constexpr (ES == 1) At = At_loc; else At = At_ext;
  for (int i = 0; i < 7; ++i) {
    const double x0 = xs[i * 4 + 0], y0 = xs[i * 4 + 1], x1 = xs[i * 4 + 2], y1 = xs[i * 4 + 3];
    double* r = At + i * 9 * ES;
    r[0 * ES] = x1 * x0; r[1 * ES] = x1 * y0; r[2 * ES] = x1;
    r[3 * ES] = y1 * x0; r[4 * ES] = y1 * y0; r[5 * ES] = y1;
    r[6 * ES] = x0; r[7 * ES] = y0; r[8 * ES] = 1;
  }
  double nv[18];
  pl_nullspace_9xm<ES>(At, 7, 7, nv);
  double* f1 = nv;
  double* f2 = nv + 9;
  for (int k = 0; k < 9; ++k) f1[k] -= f2[k];
  const double t0 = f1[4] * f1[8] - f1[5] * f1[7];
  const double t1 = f1[3] * f1[8] - f1[5] * f1[6];
  const double t2 = f1[3] * f1[7] - f1[4] * f1[6];
  const double t3 = f2[4] * f2[8] - f2[5] * f2[7];
  const double t4 = f2[3] * f2[8] - f2[5] * f2[6];
  const double t5 = f2[3] * f2[7] - f2[4] * f2[6];
  double coeffs[4];
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
  double rr[4], ri[4];
  const int nroots = pl_poly_roots<4>(coeffs, 4, rr, ri);
  if (nroots < 0) return 0;
  int nm = 0;
  for (int i = 0; i < nroots; ++i) {
    if (fabs(ri[i]) > 1e-10) continue;
    const double lambda = rr[i];
    const double mu = 1;
    double F[9];
    for (int k = 0; k < 9; ++k) F[k] = lambda * f1[k] + mu * f2[k];
    if (fabs(F[8]) < 1e-10) continue;
    const double f22 = F[8];
    for (int k = 0; k < 9; ++k) models[nm * 9 + k] = F[k] / f22;
    ++nm;
  }
  return nm;
}
DSM_DEVN int seven_point(const double* xs, double* models) { return seven_point_t<1>(xs, models, nullptr); }
// seven_point_t with every array in registers (pr_nullspace_9xm, pr_poly_roots): what the batch kernel k_solve<F> runs.
// models: 3 x 9, slots beyond the returned count are zero.
DSM_DEV int seven_point_reg(const double (&xs)[28], double (&models)[27]) {
  double At[63];  // A^T, 9 x 7 column-major: At[i*9 + c] = A(i, c)
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const double x0 = xs[i * 4 + 0], y0 = xs[i * 4 + 1], x1 = xs[i * 4 + 2], y1 = xs[i * 4 + 3];
    At[i * 9 + 0] = x1 * x0; At[i * 9 + 1] = x1 * y0; At[i * 9 + 2] = x1;
    At[i * 9 + 3] = y1 * x0; At[i * 9 + 4] = y1 * y0; At[i * 9 + 5] = y1;
    At[i * 9 + 6] = x0; At[i * 9 + 7] = y0; At[i * 9 + 8] = 1;
  }
  double nv[18];
  pr_nullspace_9xm<7>(At, nv);
  double f1[9], f2[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    f2[k] = nv[9 + k];
    f1[k] = nv[k] - f2[k];
  }
  const double t0 = f1[4] * f1[8] - f1[5] * f1[7];
  const double t1 = f1[3] * f1[8] - f1[5] * f1[6];
  const double t2 = f1[3] * f1[7] - f1[4] * f1[6];
  const double t3 = f2[4] * f2[8] - f2[5] * f2[7];
  const double t4 = f2[3] * f2[8] - f2[5] * f2[6];
  const double t5 = f2[3] * f2[7] - f2[4] * f2[6];
  double coeffs[4];
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
  double rr[4], ri[4];
  const int nroots = pr_poly_roots<4>(coeffs, rr, ri);
  if (nroots < 0) return 0;
  int nm = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) models[k] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i < nroots && !(fabs(ri[i]) > 1e-10)) {
      const double lambda = rr[i];
      const double mu = 1;
      double F[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) F[k] = lambda * f1[k] + mu * f2[k];
      if (!(fabs(F[8]) < 1e-10)) {
        const double f22 = F[8];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const double v = F[k] / f22;
#pragma unroll
          for (int sl = 0; sl < 3; ++sl) models[sl * 9 + k] = (sl == nm) ? v : models[sl * 9 + k];
        }
        ++nm;
      }
    }
  }
  return nm;
}

// Rank-2 projection + de-normalisation of the 8-point estimator (fundamental_matrix.cc:172-191);
// nullvec = cmatrix_svd.matrixV().col(8); N1/N2 given as (nf, m02, m12).
DSM_DEVN void eight_point_finish(const double* nullvec, const double* n1, const double* n2, double* Fout) {
  double U[9], V[9], sv[3];
  pl_jacobi_svd_square<3, true>(nullvec, U, V, sv);  // ematrix_t.transpose() == row-major reshape
  sv[2] = 0.0;
  double US[9], Vrm_t[9], F[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) US[i * 3 + j] = U[j * 3 + i] * sv[j];   // U is column-major
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Vrm_t[i * 3 + j] = V[i * 3 + j];        // (V^T)(i,j) = V(j,i) = V_cm[i*3+j]
  m3_mul(US, Vrm_t, F);
  const double N1[9] = {n1[0], 0, n1[1], 0, n1[0], n1[2], 0, 0, 1};
  const double N2t[9] = {n2[0], 0, 0, 0, n2[0], 0, n2[1], n2[2], 1};
  double T[9];
  m3_mul(N2t, F, T);
  m3_mul(T, N1, Fout);
}

// ---------------------------------------------------------------------------------- homography
// De-normalisation of HomographyMatrixEstimator::Estimate, homography_matrix.cc:86-91
DSM_DEV void homography_finish(const double* nullvec, const double* n1, const double* n2, double* Hout) {
  const double N1[9] = {n1[0], 0, n1[1], 0, n1[0], n1[2], 0, 0, 1};
  const double N2[9] = {n2[0], 0, n2[1], 0, n2[0], n2[2], 0, 0, 1};
  double N2inv[9], T[9];
  m3_inverse(N2, N2inv);
  m3_mul(N2inv, nullvec, T);  // H_t.transpose() == row-major reshape of the null vector
  m3_mul(T, N1, Hout);
}

// HomographyMatrixEstimator::Estimate for the minimal sample (N = 4), homography_matrix.cc:44-92
DSM_DEVN int homography_four_point(const double* xs, double* models) {
  double n1[3], n2[3];
  auto ident = [](int i) { return i; };
  center_and_normalize(xs, 0, 4, ident, &n1[0], &n1[1], &n1[2]);
  center_and_normalize(xs, 1, 4, ident, &n2[0], &n2[1], &n2[2]);
  double At[72];  // A^T, 9 x 8 column-major: row r of A is At[r*9 .. r*9+8]
  for (int i = 0; i < 72; ++i) At[i] = 0.0;
  for (int i = 0, j = 4; i < 4; ++i, ++j) {
    double s_0, s_1, d_0, d_1;
    apply_norm(n1[0], n1[1], n1[2], xs[i * 4 + 0], xs[i * 4 + 1], &s_0, &s_1);
    apply_norm(n2[0], n2[1], n2[2], xs[i * 4 + 2], xs[i * 4 + 3], &d_0, &d_1);
    double* ri = At + i * 9;
    double* rj = At + j * 9;
    ri[0] = -s_0; ri[1] = -s_1; ri[2] = -1;
    ri[6] = s_0 * d_0; ri[7] = s_1 * d_0; ri[8] = d_0;
    rj[3] = -s_0; rj[4] = -s_1; rj[5] = -1;
    rj[6] = s_0 * d_1; rj[7] = s_1 * d_1; rj[8] = d_1;
  }
  double nv[9];
  pl_nullspace_9xm(At, 8, 8, nv);
  homography_finish(nv, n1, n2, models);
  return 1;
}


// ---------------------------------------------------------------------------------- homography, registers
// Same arithmetic, operation for operation, as homography_four_point() above, but with every
// array index a compile-time constant (fully unrolled loops, column pivoting through selects) so
// that the 9 x 8 system lives in VGPRs instead of scratch memory.  This is the hot minimal solver:
// the H family runs to its trial cap (1 765 trials per pair with the default options).
template <int K>
struct H4Step {
  // one step of ColPivHouseholderQR::computeInPlace on a[c][r] (column c, row r), rows = 9, cols = 8
  static DSM_DEV void run(double (&a)[8][9], double (&nu)[8], double (&nd)[8], double (&hco)[8]) {
    int biggest = K;
    double mx = nu[K];
#pragma unroll
    for (int j = K + 1; j < 8; ++j)
      if (nu[j] > mx) {
        mx = nu[j];
        biggest = j;
      }
#pragma unroll
    for (int j = K + 1; j < 8; ++j) {
      const bool sw = (biggest == j);
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const double t = a[K][r], u = a[j][r];
        a[K][r] = sw ? u : t;
        a[j][r] = sw ? t : u;
      }
      const double t1 = nu[K], u1 = nu[j];
      nu[K] = sw ? u1 : t1;
      nu[j] = sw ? t1 : u1;
      const double t2 = nd[K], u2 = nd[j];
      nd[K] = sw ? u2 : t2;
      nd[j] = sw ? t2 : u2;
    }
    // makeHouseholderInPlace on a[K][K..8]
    double tail_sq = 0.0;
#pragma unroll
    for (int i = K + 1; i < 9; ++i) tail_sq += a[K][i] * a[K][i];
    const double c0 = a[K][K];
    double tau, beta;
    if (tail_sq <= DBL_MIN) {
      tau = 0.0;
      beta = c0;
#pragma unroll
      for (int i = K + 1; i < 9; ++i) a[K][i] = 0.0;
    } else {
      double b = sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0.0) b = -b;
      const SharedDivisor hd = shared_divisor(c0 - b);  // (verify_linalg.h: the same quotients, the reciprocal formed once)
      double mn = 0x1p1000;
#pragma unroll
      for (int i = K + 1; i < 9; ++i) mn = fmin(mn, fabs(a[K][i]));
      if (div_shared_group_ok(hd, mn)) {
#pragma unroll
        for (int i = K + 1; i < 9; ++i) a[K][i] = div_shared_fast(a[K][i], hd);
      } else {
#pragma unroll
        for (int i = K + 1; i < 9; ++i) a[K][i] = a[K][i] / (c0 - b);
      }
      tau = (b - c0) / b;
      beta = b;
    }
    hco[K] = tau;
    a[K][K] = beta;
    // applyHouseholderOnTheLeft to columns K+1..7, rows K..8 (nr = 9 - K >= 2 always)
    if (tau != 0.0) {
#pragma unroll
      for (int j = K + 1; j < 8; ++j) {
        double tmp = 0.0;
#pragma unroll
        for (int i = K + 1; i < 9; ++i) tmp += a[K][i] * a[j][i];
        tmp += a[j][K];
        a[j][K] -= tau * tmp;
#pragma unroll
        for (int i = K + 1; i < 9; ++i) a[j][i] -= tau * a[K][i] * tmp;
      }
    }
    // norm downdating
    const double norm_downdate_threshold = sqrt(DBL_EPSILON);
#pragma unroll
    for (int j = K + 1; j < 8; ++j) {
      if (nu[j] != 0.0) {
        double temp = fabs(a[j][K]) / nu[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double ratio = nu[j] / nd[j];
        const double temp2 = temp * (ratio * ratio);
        if (temp2 <= norm_downdate_threshold) {
          double ss = 0.0;
#pragma unroll
          for (int i = K + 1; i < 9; ++i) ss += a[j][i] * a[j][i];
          nd[j] = sqrt(ss);
          nu[j] = nd[j];
        } else {
          nu[j] *= sqrt(temp);
        }
      }
    }
  }
};

DSM_DEV int homography_four_point_reg(const double* xs, double* models) {
  double n1[3], n2[3];
  auto ident = [](int i) { return i; };
  center_and_normalize(xs, 0, 4, ident, &n1[0], &n1[1], &n1[2]);
  center_and_normalize(xs, 1, 4, ident, &n2[0], &n2[1], &n2[2]);
  double a[8][9];  // a[c][r] = At(r, c) = A(c, r): column c of A^T is row c of A
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 9; ++r) a[c][r] = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double s_0, s_1, d_0, d_1;
    apply_norm(n1[0], n1[1], n1[2], xs[i * 4 + 0], xs[i * 4 + 1], &s_0, &s_1);
    apply_norm(n2[0], n2[1], n2[2], xs[i * 4 + 2], xs[i * 4 + 3], &d_0, &d_1);
    a[i][0] = -s_0; a[i][1] = -s_1; a[i][2] = -1;
    a[i][6] = s_0 * d_0; a[i][7] = s_1 * d_0; a[i][8] = d_0;
    a[4 + i][3] = -s_0; a[4 + i][4] = -s_1; a[4 + i][5] = -1;
    a[4 + i][6] = s_0 * d_1; a[4 + i][7] = s_1 * d_1; a[4 + i][8] = d_1;
  }
  // pl_nullspace_9xm: scale, pivoted QR, column 8 of householderQ
  double scale = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const double v = fabs(a[c][r]);
      if (v > scale) scale = v;
    }
  if (scale == 0.0) scale = 1.0;
  // m / scale (JacobiSVD's m_scaledMatrix): an exactly rounded division per entry -- a third of this solver's
  // instructions when done for all 72.  24 entries are structural zeros (0 / scale = 0) and the eight -1 entries share
  // one quotient: 41 divisions, same values.
  const double neg_inv = -1.0 / scale;
  const SharedDivisor sd = shared_divisor(scale);  // (and the 40 remaining quotients share the refined reciprocal)
  double mn = 0x1p1000;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = i < 4 ? 0 : 3;
    mn = fmin(mn, fmin(fmin(fabs(a[i][o]), fabs(a[i][o + 1])), fmin(fmin(fabs(a[i][6]), fabs(a[i][7])), fabs(a[i][8]))));
  }
  if (div_shared_group_ok(sd, mn)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int o = i < 4 ? 0 : 3;
      a[i][o] = div_shared_fast(a[i][o], sd); a[i][o + 1] = div_shared_fast(a[i][o + 1], sd); a[i][o + 2] = neg_inv;
      a[i][6] = div_shared_fast(a[i][6], sd); a[i][7] = div_shared_fast(a[i][7], sd); a[i][8] = div_shared_fast(a[i][8], sd);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int o = i < 4 ? 0 : 3;
      a[i][o] /= scale; a[i][o + 1] /= scale; a[i][o + 2] = neg_inv;
      a[i][6] /= scale; a[i][7] /= scale; a[i][8] /= scale;
    }
  }
  double nu[8], nd[8], hco[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    double ss = 0.0;
#pragma unroll
    for (int r = 0; r < 9; ++r) ss += a[c][r] * a[c][r];
    nd[c] = sqrt(ss);
    nu[c] = nd[c];
  }
  H4Step<0>::run(a, nu, nd, hco);
  H4Step<1>::run(a, nu, nd, hco);
  H4Step<2>::run(a, nu, nd, hco);
  H4Step<3>::run(a, nu, nd, hco);
  H4Step<4>::run(a, nu, nd, hco);
  H4Step<5>::run(a, nu, nd, hco);
  H4Step<6>::run(a, nu, nd, hco);
  H4Step<7>::run(a, nu, nd, hco);
  double q[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) q[i] = (i == 8) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    const double tau = hco[k];
    if (tau != 0.0) {
      double tmp = 0.0;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) tmp += a[k][i] * q[i];
      tmp += q[k];
      q[k] -= tau * tmp;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) q[i] -= tau * a[k][i] * tmp;
    }
  }
  homography_finish(q, n1, n2, models);
  return 1;
}

// ---------------------------------------------------------------------------------- 5-point E
// Steps 3 and 4 of EssentialMatrixFivePointEstimator::Estimate evaluate machine-generated straight-line code
// (/root/reference/src/estimators/essential_matrix.cc:76-77 includes essential_matrix_poly.h, :101-102
// essential_matrix_coeffs.h).  The ORDER of its sums and products decides the rounding, and it is data: the term table
// fivept_terms.tbl (tools/gen_fivept_tables.py), expanded by the Makefile into fivept_poly_gen.inc -- straight-line
// statements for the lane-per-hypothesis kernels (every index a compile-time constant: registers) and packed term
// words for the 16-lane cooperative solver (verify_fivept_coop.h), which interprets them.  Layouts of the reference:
// e = E.data() (9 x 4 column-major), a = A.data() (10 x 20 column-major), b = B.data() (13 x 3 column-major); here
// Eb[r*4 + c], A[r*20 + c], B[r*3 + c].
#define FIVEPT_TABLE_QUAL __device__
#define FIVEPT_EMIT_TABLES
#include "fivept_poly_gen.inc"
#undef FIVEPT_EMIT_TABLES

// Zero-instruction "the value may have changed" marker: the optimiser cannot merge or move arithmetic on x across it.
DSM_DEV void fivept_launder(double& x) { asm volatile("" : "+v"(x)); }

// Step 3: the 10 x 20 constraint matrix, element A(r, c) written to out[(r*20 + c) * ES] (ES = 1: one hypothesis'
// 200 doubles back to back; 64: the hypotheses of a wave interleaved, element q of lane l at out[q * 64 + l]:
// coalesced stores).  Every index is a compile-time constant: e and e2 (144 VGPRs) stay in registers.
// Left to itself the optimiser merges the common sub-products e[i]*e[j] of all 5 600 terms into hundreds of
// long-lived values (968 spilled VGPRs); a fence after every FIVEPT_GROUP statements confines merging and
// scheduling to the group -- the arithmetic of every statement is untouched.
#define FIVEPT_GROUP 4
template <int ES = 1>
DSM_DEV void five_point_build_A(const double (&Eb)[36], double* out) {
  LSEC_BEGIN();
  // e2[i] = e[i] * e[i], e3[i] = e2[i] * e[i] as in the header's preamble; the 38 uses of e3 recompute the product
  // (same value) so that only e and e2 stay live
  double e[36], e2[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    e[k] = Eb[(k % 9) * 4 + k / 9];
    e2[k] = e[k] * e[k];
  }
#define FIVEPT_E(k) e[k]
#define FIVEPT_E2(k) e2[k]
#define FIVEPT_E3(k) (e2[k] * e[k])
#define FIVEPT_A(i) out[(((i) % 10) * 20 + (i) / 10) * ES]
#define FIVEPT_FENCE(n)                               \
  if ((n) % FIVEPT_GROUP == FIVEPT_GROUP - 1) {       \
    _Pragma("unroll") for (int k = 0; k < 36; ++k) {  \
      fivept_launder(e[k]);                           \
      fivept_launder(e2[k]);                          \
    }                                                 \
  }
#define FIVEPT_EMIT_A
#include "fivept_poly_gen.inc"
#undef FIVEPT_EMIT_A
#undef FIVEPT_FENCE
#undef FIVEPT_A
#undef FIVEPT_E3
#undef FIVEPT_E2
#undef FIVEPT_E
  LSEC_END(8);
}

// Step 4: B(z) from rows 4..9 of the eliminated system (S[(r-4)*10 + c] = AA(r, c); essential_matrix.cc:86-97) and
// the determinant polynomial (essential_matrix_coeffs.h), highest degree first.
DSM_DEV void five_point_B_det(const double* S, double* B, double* coeffs) {
  LSEC_BEGIN3();
#define AAe(r, c) S[((r)-4) * 10 + (c)]
  // B[row*3 + col], 39 entries
  for (int i = 0; i < 3; ++i) {
    B[0 * 3 + i] = 0; B[4 * 3 + i] = 0; B[8 * 3 + i] = 0;
    for (int k = 0; k < 3; ++k) {
      B[(1 + k) * 3 + i] = AAe(i * 2 + 4, k);
      B[(5 + k) * 3 + i] = AAe(i * 2 + 4, 3 + k);
    }
    for (int k = 0; k < 4; ++k) B[(9 + k) * 3 + i] = AAe(i * 2 + 4, 6 + k);
    for (int k = 0; k < 3; ++k) {
      B[(0 + k) * 3 + i] -= AAe(i * 2 + 5, k);
      B[(4 + k) * 3 + i] -= AAe(i * 2 + 5, 3 + k);
    }
    for (int k = 0; k < 4; ++k) B[(8 + k) * 3 + i] -= AAe(i * 2 + 5, 6 + k);
  }
#undef AAe
#define FIVEPT_B(k) B[((k) % 13) * 3 + (k) / 13]
#define FIVEPT_C(i) coeffs[i]
#define FIVEPT_FENCE(n)
#define FIVEPT_EMIT_C
#include "fivept_poly_gen.inc"
#undef FIVEPT_EMIT_C
#undef FIVEPT_FENCE
#undef FIVEPT_C
#undef FIVEPT_B
  LSEC_END3(10);
}

// One target of the packed term table (the cooperative solver's form of the two functions above): the running sum
// of products over terms[off[t] .. off[t+1]), factors read from `v` (FivePtVals: [e | e2 | e3] resp. b).
template <typename V>
DSM_DEV double five_point_eval_terms(const uint32_t* terms, uint32_t begin, uint32_t end, V v) {
  double acc = 0.0;
  for (uint32_t q = begin; q < end; ++q) {
    const uint32_t w = terms[q];
    double pr = v[w & 0xff];
    const uint32_t lit = (w >> 24) & 3;
    if (lit) pr = (lit == 1 ? 0.5 : (lit == 2 ? 1.5 : 3.0)) * pr;
    const uint32_t f1 = (w >> 8) & 0xff, f2 = (w >> 16) & 0xff;
    if (f1 != 0xff) pr = pr * v[f1];
    if (f2 != 0xff) pr = pr * v[f2];
    if (q == begin)
      acc = (w >> 31) ? -pr : pr;
    else
      acc = (w >> 31) ? acc - pr : acc + pr;
  }
  return acc;
}

// Steps 3-4a for one lane with private (or caller-provided) storage.
template <bool WS>
DSM_DEV void five_point_poly_t(const double* Eb, double* B, double* coeffs, double* ws) {
  double A_loc[WS ? 1 : 200], A1_loc[WS ? 1 : 100], AA_loc[WS ? 1 : 100];
  double* A = WS ? ws : A_loc;  // A[r*20 + c]
  double* A1 = WS ? ws + 200 : A1_loc;
  double* AA = WS ? ws + 300 : AA_loc;
  {
    double Er[36];
    for (int k = 0; k < 36; ++k) Er[k] = Eb[k];
    five_point_build_A<1>(Er, A);
  }
  LSEC_BEGIN2();
  for (int r = 0; r < 10; ++r)  // A1, AA: column-major 10 x 10
    for (int c = 0; c < 10; ++c) {
      A1[c * 10 + r] = A[r * 20 + c];
      AA[c * 10 + r] = A[r * 20 + 10 + c];
    }
  pl_lu_solve_10(A1, AA);
  LSEC_END2(9);
  double S[60];
  for (int r = 4; r < 10; ++r)
    for (int c = 0; c < 10; ++c) S[(r - 4) * 10 + c] = AA[c * 10 + r];
  five_point_B_det(S, B, coeffs);
}

// Step 5: one essential matrix per real root of the determinant polynomial (essential_matrix.cc:124-147).
DSM_DEV int five_point_models(const double* Eb, const double* B, const double* rr, const double* ri, int nroots,
                              double* models) {
  LSEC_BEGIN5();
  int nm = 0;
  for (int i = 0; i < nroots; ++i) {
    if (fabs(ri[i]) > 1e-10) continue;
    const double z1 = rr[i];
    const double z2 = z1 * z1;
    const double z3 = z2 * z1;
    const double z4 = z3 * z1;
    double Bz[9];
    for (int j = 0; j < 3; ++j) {
      Bz[j * 3 + 0] = B[0 * 3 + j] * z3 + B[1 * 3 + j] * z2 + B[2 * 3 + j] * z1 + B[3 * 3 + j];
      Bz[j * 3 + 1] = B[4 * 3 + j] * z3 + B[5 * 3 + j] * z2 + B[6 * 3 + j] * z1 + B[7 * 3 + j];
      Bz[j * 3 + 2] = B[8 * 3 + j] * z4 + B[9 * 3 + j] * z3 + B[10 * 3 + j] * z2 + B[11 * 3 + j] * z1 + B[12 * 3 + j];
    }
    double Vz[9], svz[3];
    pl_jacobi_svd_square<3, false>(Bz, nullptr, Vz, svz);
    const double X0 = Vz[2 * 3 + 0], X1 = Vz[2 * 3 + 1], X2 = Vz[2 * 3 + 2];
    if (fabs(X2) < 1e-10) continue;
    const double sx = X0 / X2, sy = X1 / X2;
    double ev[9];
    for (int k = 0; k < 9; ++k) ev[k] = Eb[k * 4 + 0] * sx + Eb[k * 4 + 1] * sy + Eb[k * 4 + 2] * z1 + Eb[k * 4 + 3];
    double nn = 0.0;
    for (int k = 0; k < 9; ++k) nn += ev[k] * ev[k];
    const double norm = sqrt(nn);
    for (int k = 0; k < 9; ++k) models[nm * 9 + k] = ev[k] / norm;
    ++nm;
  }
  LSEC_END5(12);
  return nm;
}

// five_point_models for the batch kernels: Eb, B and the roots in registers (static indices; the root of the trip is a
// select chain, the 3 x 3 SVD is pr_jacobi_svd_square_V), every model written straight to `out` (global memory) -- no
// private arrays.  real_mask: bit i = root i is real (|imag| <= 1e-10).
DSM_DEV int five_point_models_reg(const double (&Eb)[36], const double (&B)[39], const double (&rr)[10], int real_mask, int nroots,
                                  double* out) {
  int nm = 0;
  for (int i = 0; i < nroots; ++i) {
    if (!((real_mask >> i) & 1)) continue;
    double z1 = rr[0];
#pragma unroll
    for (int q = 1; q < 10; ++q) z1 = (i == q) ? rr[q] : z1;
    const double z2 = z1 * z1;
    const double z3 = z2 * z1;
    const double z4 = z3 * z1;
    double Bz[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Bz[j * 3 + 0] = B[0 * 3 + j] * z3 + B[1 * 3 + j] * z2 + B[2 * 3 + j] * z1 + B[3 * 3 + j];
      Bz[j * 3 + 1] = B[4 * 3 + j] * z3 + B[5 * 3 + j] * z2 + B[6 * 3 + j] * z1 + B[7 * 3 + j];
      Bz[j * 3 + 2] = B[8 * 3 + j] * z4 + B[9 * 3 + j] * z3 + B[10 * 3 + j] * z2 + B[11 * 3 + j] * z1 + B[12 * 3 + j];
    }
    double Vz[9], svz[3];
    pr_jacobi_svd_square_V<3>(Bz, Vz, svz);
    const double X0 = Vz[2 * 3 + 0], X1 = Vz[2 * 3 + 1], X2 = Vz[2 * 3 + 2];
    if (fabs(X2) < 1e-10) continue;
    const double sx = X0 / X2, sy = X1 / X2;
    double ev[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) ev[k] = Eb[k * 4 + 0] * sx + Eb[k * 4 + 1] * sy + Eb[k * 4 + 2] * z1 + Eb[k * 4 + 3];
    double nn = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) nn += ev[k] * ev[k];
    const double norm = sqrt(nn);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[nm * 9 + k] = ev[k] / norm;
    ++nm;
  }
  return nm;
}

template <bool WS>
DSM_DEVN int five_point_finish_t(const double* Eb, double* models, double* ws) {
  double B[39], coeffs[11];
  five_point_poly_t<WS>(Eb, B, coeffs, ws);
  double rr[11], ri[11];
  LSEC_BEGIN4();
  const int nroots = pl_poly_roots<11>(coeffs, 11, rr, ri, WS ? ws + 400 : nullptr);
  LSEC_END4(11);
  if (nroots < 0) return 0;
  return five_point_models(Eb, B, rr, ri, nroots, models);
}

DSM_DEV int five_point_finish(const double* Eb, double* models) { return five_point_finish_t<false>(Eb, models, nullptr); }

// EssentialMatrixFivePointEstimator::Estimate for the minimal sample, essential_matrix.cc:46-150
// Steps 1-2 for the minimal sample: the basis of the null space of Q (essential_matrix.cc:52-74)
DSM_DEV void five_point_basis(const double* xs, double* Eb) {
  double At[45];  // Q^T, 9 x 5
  for (int i = 0; i < 5; ++i) {
    const double x1_0 = xs[i * 4 + 0], x1_1 = xs[i * 4 + 1], x2_0 = xs[i * 4 + 2], x2_1 = xs[i * 4 + 3];
    double* r = At + i * 9;
    r[0] = x1_0 * x2_0; r[1] = x1_1 * x2_0; r[2] = x2_0;
    r[3] = x1_0 * x2_1; r[4] = x1_1 * x2_1; r[5] = x2_1;
    r[6] = x1_0; r[7] = x1_1; r[8] = 1;
  }
  double nv[36];  // columns 5..8 of V, nv[c*9 + r]
  pl_nullspace_9xm(At, 5, 5, nv);
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 4; ++c) Eb[r * 4 + c] = nv[c * 9 + r];
}
// five_point_basis with the 9 x 5 matrix in registers (batch kernels)
DSM_DEV void five_point_basis_reg(const double (&xs)[20], double (&Eb)[36]) {
  double At[45];  // Q^T, 9 x 5
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double x1_0 = xs[i * 4 + 0], x1_1 = xs[i * 4 + 1], x2_0 = xs[i * 4 + 2], x2_1 = xs[i * 4 + 3];
    At[i * 9 + 0] = x1_0 * x2_0; At[i * 9 + 1] = x1_1 * x2_0; At[i * 9 + 2] = x2_0;
    At[i * 9 + 3] = x1_0 * x2_1; At[i * 9 + 4] = x1_1 * x2_1; At[i * 9 + 5] = x2_1;
    At[i * 9 + 6] = x1_0; At[i * 9 + 7] = x1_1; At[i * 9 + 8] = 1;
  }
  double nv[36];  // columns 5..8 of V, nv[c*9 + r]
  pr_nullspace_9xm<5>(At, nv);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) Eb[r * 4 + c] = nv[c * 9 + r];
  }
}
DSM_DEVN int five_point_minimal(const double* xs, double* models) {
  double Eb[36];
  five_point_basis(xs, Eb);
  return five_point_finish(Eb, models);
}

#endif  // DAGSFM_AMD_CSRC_VERIFY_ESTIMATORS_H_
