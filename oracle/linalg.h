// oracle/linalg.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Eigen-free restatement of the Eigen routines the reference's two-view path calls
// (Eigen itself is neither vendored in /root/reference nor
// installed here, and is unpinned upstream: `find_package(Eigen3 REQUIRED)`,
// /root/reference/CMakeLists.txt:97).  Restated from the published Eigen 3.3 algorithms:
//   JacobiSVD<.., ColPivHouseholderQRPreconditioner>   (two-sided Jacobi on the QR-reduced square)
//   ColPivHouseholderQR (LAPACK-style norm downdating)  HouseholderSequence::evalTo
//   EigenSolver/RealSchur (Hessenberg + Francis double-shift QR, eigenvalues only)
//   PartialPivLU::solve, Matrix3d::inverse (cofactors), Quaterniond(Matrix3d)
// Reductions (dot products, norms) inside these Eigen routines have no order the reference fixes:
// Eigen's own association order depends on its packet width and unrolling (SSE2 here, AVX
// elsewhere), so oracle == Eigen only to rounding (checked against numpy and the reference's
// known-answer tests in tests/).  The oracle therefore picks one definite order and the device
// code uses exactly the same one, which makes GPU == oracle bit-exact:
//   * short vectors (every routine on matrices with <= 9 rows): sequential, left to right;
//   * the pivoted QR of a tall matrix (rows > 9: the local-optimisation / final least-squares
//     systems with one or two rows per correspondence): wide_sum() below -- 64 interleaved partial
//     sums combined by a fixed binary tree, the natural order of a 64-lane wavefront.
// Sums the reference itself writes as a sequential loop (centroids, residual_sum, ...) are NOT
// Eigen reductions and stay sequential everywhere (two_view.cc).
#ifndef ORACLE_LINALG_H_
#define ORACLE_LINALG_H_

#include <cfloat>
#include <cmath>
#include <cstddef>
#include <vector>

namespace oracle {

// Dense column-major matrix (Eigen's default storage order).
struct Mat {
  int rows = 0, cols = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r, int c) : rows(r), cols(c), a(static_cast<size_t>(r) * c, 0.0) {}
  double& operator()(int r, int c) { return a[static_cast<size_t>(c) * rows + r]; }
  double operator()(int r, int c) const { return a[static_cast<size_t>(c) * rows + r]; }
};

// sum_{i < n} f(i) in the "wide" order: partial[i % 64] accumulates f(i) in increasing i, then
// partial[l] += partial[l + o] for o = 32, 16, 8, 4, 2, 1.
template <typename F>
inline double wide_sum(int n, F f) {
  double part[64];
  for (int l = 0; l < 64; ++l) part[l] = 0.0;
  for (int i = 0; i < n; ++i) part[i & 63] += f(i);
  for (int o = 32; o > 0; o >>= 1)
    for (int l = 0; l < o; ++l) part[l] += part[l + o];
  return part[0];
}

// ---- Householder (Eigen/src/Householder/Householder.h) -------------------------------------
// makeHouseholder on x[0..n): returns tau, beta; essential part = x[1..n) / (x0 - beta).
inline void make_householder(double* x, int n, double* tau, double* beta, bool wide = false) {
  double tail_sq = 0.0;
  if (wide) {
    tail_sq = wide_sum(n - 1, [x](int i) { return x[i + 1] * x[i + 1]; });
  } else {
    for (int i = 1; i < n; ++i) tail_sq += x[i] * x[i];
  }
  const double c0 = x[0];
  const double tol = DBL_MIN;
  if (tail_sq <= tol) {
    *tau = 0.0;
    *beta = c0;
    for (int i = 1; i < n; ++i) x[i] = 0.0;
  } else {
    double b = std::sqrt(c0 * c0 + tail_sq);
    if (c0 >= 0.0) b = -b;
    for (int i = 1; i < n; ++i) x[i] = x[i] / (c0 - b);
    *tau = (b - c0) / b;
    *beta = b;
  }
}

// M.block(r0,c0,nr,nc).applyHouseholderOnTheLeft(essential, tau)
inline void apply_householder_left(Mat& M, int r0, int c0, int nr, int nc, const double* ess, double tau,
                                   bool wide = false) {
  if (nr == 1) {
    for (int j = 0; j < nc; ++j) M(r0, c0 + j) *= (1.0 - tau);
  } else if (tau != 0.0) {
    for (int j = 0; j < nc; ++j) {
      double tmp = 0.0;
      if (wide) {
        const Mat& Mc = M;
        tmp = wide_sum(nr - 1, [&](int i) { return ess[i] * Mc(r0 + i + 1, c0 + j); });
      } else {
        for (int i = 1; i < nr; ++i) tmp += ess[i - 1] * M(r0 + i, c0 + j);
      }
      tmp += M(r0, c0 + j);
      M(r0, c0 + j) -= tau * tmp;
      for (int i = 1; i < nr; ++i) M(r0 + i, c0 + j) -= tau * ess[i - 1] * tmp;
    }
  }
}

// M.block(r0,c0,nr,nc).applyHouseholderOnTheRight(essential, tau)
inline void apply_householder_right(Mat& M, int r0, int c0, int nr, int nc, const double* ess, double tau) {
  if (nc == 1) {
    for (int i = 0; i < nr; ++i) M(r0 + i, c0) *= (1.0 - tau);
  } else if (tau != 0.0) {
    for (int i = 0; i < nr; ++i) {
      double tmp = 0.0;
      for (int j = 1; j < nc; ++j) tmp += M(r0 + i, c0 + j) * ess[j - 1];
      tmp += M(r0 + i, c0);
      M(r0 + i, c0) -= tau * tmp;
      for (int j = 1; j < nc; ++j) M(r0 + i, c0 + j) -= tau * tmp * ess[j - 1];
    }
  }
}

// ---- ColPivHouseholderQR (Eigen/src/QR/ColPivHouseholderQR.h, 3.3) ---------------------------
struct ColPivQR {
  Mat qr;                    // R in the upper triangle, essential Householder parts below
  std::vector<double> hcoeffs;
  std::vector<int> perm;     // column permutation indices: A * P = Q * R, P(perm[j], j) = 1
  void compute(const Mat& A) {
    qr = A;
    const int rows = qr.rows, cols = qr.cols;
    const int size = rows < cols ? rows : cols;
    hcoeffs.assign(size, 0.0);
    std::vector<int> transp(cols);
    std::vector<double> norms_updated(cols), norms_direct(cols);
    const bool wide = rows > 9;  // reduction order, see the header of this file
    const Mat& cq = qr;
    for (int k = 0; k < cols; ++k) {
      double s = 0.0;
      if (wide) {
        s = wide_sum(rows, [&](int i) { return cq(i, k) * cq(i, k); });
      } else {
        for (int i = 0; i < rows; ++i) s += qr(i, k) * qr(i, k);
      }
      norms_direct[k] = std::sqrt(s);
      norms_updated[k] = norms_direct[k];
    }
    const double norm_downdate_threshold = std::sqrt(DBL_EPSILON);
    for (int k = 0; k < size; ++k) {
      int biggest = k;
      double mx = norms_updated[k];
      for (int j = k + 1; j < cols; ++j)
        if (norms_updated[j] > mx) {
          mx = norms_updated[j];
          biggest = j;
        }
      transp[k] = biggest;
      if (k != biggest) {
        for (int i = 0; i < rows; ++i) {
          const double t = qr(i, k);
          qr(i, k) = qr(i, biggest);
          qr(i, biggest) = t;
        }
        std::swap(norms_updated[k], norms_updated[biggest]);
        std::swap(norms_direct[k], norms_direct[biggest]);
      }
      double tau, beta;
      make_householder(&qr(k, k), rows - k, &tau, &beta, wide);
      hcoeffs[k] = tau;
      qr(k, k) = beta;
      apply_householder_left(qr, k, k + 1, rows - k, cols - k - 1, &qr.a[static_cast<size_t>(k) * rows + k + 1], tau,
                             wide);
      for (int j = k + 1; j < cols; ++j) {
        if (norms_updated[j] != 0.0) {
          double temp = std::fabs(qr(k, j)) / norms_updated[j];
          temp = (1.0 + temp) * (1.0 - temp);
          temp = temp < 0.0 ? 0.0 : temp;
          const double ratio = norms_updated[j] / norms_direct[j];
          const double temp2 = temp * (ratio * ratio);
          if (temp2 <= norm_downdate_threshold) {
            double s = 0.0;
            if (wide) {
              s = wide_sum(rows - k - 1, [&](int i) { return cq(k + 1 + i, j) * cq(k + 1 + i, j); });
            } else {
              for (int i = k + 1; i < rows; ++i) s += qr(i, j) * qr(i, j);
            }
            norms_direct[j] = std::sqrt(s);
            norms_updated[j] = norms_direct[j];
          } else {
            norms_updated[j] *= std::sqrt(temp);
          }
        }
      }
    }
    perm.resize(cols);
    for (int j = 0; j < cols; ++j) perm[j] = j;
    for (int k = 0; k < size; ++k) std::swap(perm[k], perm[transp[k]]);
  }
  // householderQ().evalTo(Q): full rows x rows orthogonal factor.
  Mat householder_q() const {
    const int rows = qr.rows;
    const int size = static_cast<int>(hcoeffs.size());
    Mat Q(rows, rows);
    for (int i = 0; i < rows; ++i) Q(i, i) = 1.0;
    for (int k = size - 1; k >= 0; --k) {
      const int corner = rows - k;
      apply_householder_left(Q, k, k, corner, corner, &qr.a[static_cast<size_t>(k) * rows + k + 1], hcoeffs[k]);
    }
    return Q;
  }
};

// ---- JacobiSVD (Eigen/src/SVD/JacobiSVD.h, 3.3) ------------------------------------------------
struct JacobiRot {
  double c, s;
};

// JacobiRotation::makeJacobi(x, y, z), Eigen/src/Jacobi/Jacobi.h
inline JacobiRot make_jacobi(double x, double y, double z) {
  JacobiRot r;
  const double deno = 2.0 * std::fabs(y);
  if (deno < DBL_MIN) {
    r.c = 1.0;
    r.s = 0.0;
  } else {
    const double tau = (x - z) / deno;
    const double w = std::sqrt(tau * tau + 1.0);
    double t;
    if (tau > 0.0)
      t = 1.0 / (tau + w);
    else
      t = 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / std::sqrt(t * t + 1.0);
    r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
    r.c = n;
  }
  return r;
}

// apply_rotation_in_the_plane on two strided vectors: x' = c x + s y, y' = -s x + c y
inline void rot_apply(double* x, int incx, double* y, int incy, int n, double c, double s) {
  if (c == 1.0 && s == 0.0) return;
  for (int i = 0; i < n; ++i) {
    const double xi = x[static_cast<size_t>(i) * incx];
    const double yi = y[static_cast<size_t>(i) * incy];
    x[static_cast<size_t>(i) * incx] = c * xi + s * yi;
    y[static_cast<size_t>(i) * incy] = -s * xi + c * yi;
  }
}

struct SVD {
  Mat U, V;                 // full U (rows x rows) when want_u, full V (cols x cols)
  std::vector<double> sv;   // singular values, descending
};

// JacobiSVD<MatrixXd>(A, ComputeFullV [| ComputeFullU]).
inline SVD jacobi_svd(const Mat& A, bool want_u) {
  const int rows = A.rows, cols = A.cols;
  const int diag = rows < cols ? rows : cols;
  SVD out;
  const double precision = 2.0 * DBL_EPSILON;
  const double consider_as_zero = DBL_MIN;
  double scale = 0.0;
  for (size_t i = 0; i < A.a.size(); ++i)
    if (std::fabs(A.a[i]) > scale) scale = std::fabs(A.a[i]);
  if (scale == 0.0) scale = 1.0;
  Mat W(diag, diag);
  if (rows > cols) {
    Mat S = A;
    for (double& v : S.a) v /= scale;
    ColPivQR qr;
    qr.compute(S);
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i <= j; ++i) W(i, j) = qr.qr(i, j);
    if (want_u) out.U = qr.householder_q();
    out.V = Mat(cols, cols);
    for (int j = 0; j < cols; ++j) out.V(qr.perm[j], j) = 1.0;
  } else if (cols > rows) {
    Mat S(cols, rows);  // adjoint
    for (int i = 0; i < rows; ++i)
      for (int j = 0; j < cols; ++j) S(j, i) = A(i, j) / scale;
    ColPivQR qr;
    qr.compute(S);
    for (int j = 0; j < rows; ++j)
      for (int i = 0; i <= j; ++i) W(j, i) = qr.qr(i, j);  // upper-triangular R, adjointed
    out.V = qr.householder_q();
    if (want_u) {
      out.U = Mat(rows, rows);
      for (int j = 0; j < rows; ++j) out.U(qr.perm[j], j) = 1.0;
    }
  } else {
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < rows; ++i) W(i, j) = A(i, j) / scale;
    if (want_u) {
      out.U = Mat(rows, rows);
      for (int i = 0; i < rows; ++i) out.U(i, i) = 1.0;
    }
    out.V = Mat(cols, cols);
    for (int i = 0; i < cols; ++i) out.V(i, i) = 1.0;
  }

  double max_diag = 0.0;
  for (int i = 0; i < diag; ++i)
    if (std::fabs(W(i, i)) > max_diag) max_diag = std::fabs(W(i, i));
  bool finished = false;
  while (!finished) {
    finished = true;
    for (int p = 1; p < diag; ++p) {
      for (int q = 0; q < p; ++q) {
        const double thr = consider_as_zero > precision * max_diag ? consider_as_zero : precision * max_diag;
        if (std::fabs(W(p, q)) > thr || std::fabs(W(q, p)) > thr) {
          finished = false;
          // real_2x2_jacobi_svd
          double m00 = W(p, p), m01 = W(p, q), m10 = W(q, p), m11 = W(q, q);
          JacobiRot rot1;
          const double t = m00 + m11;
          const double d = m10 - m01;
          if (std::fabs(d) < DBL_MIN) {
            rot1.s = 0.0;
            rot1.c = 1.0;
          } else {
            const double u = t / d;
            const double tmp = std::sqrt(1.0 + u * u);
            rot1.s = 1.0 / tmp;
            rot1.c = u / tmp;
          }
          // m.applyOnTheLeft(0,1,rot1)
          {
            const double a0 = rot1.c * m00 + rot1.s * m10, a1 = rot1.c * m01 + rot1.s * m11;
            const double b0 = -rot1.s * m00 + rot1.c * m10, b1 = -rot1.s * m01 + rot1.c * m11;
            m00 = a0;
            m01 = a1;
            m10 = b0;
            m11 = b1;
          }
          const JacobiRot jr = make_jacobi(m00, m01, m11);
          // j_left = rot1 * j_right.transpose()
          JacobiRot jl;
          jl.c = rot1.c * jr.c - rot1.s * (-jr.s);
          jl.s = rot1.c * (-jr.s) + rot1.s * jr.c;
          // workMatrix.applyOnTheLeft(p,q,j_left): rows p,q
          rot_apply(&W.a[p], diag, &W.a[q], diag, diag, jl.c, jl.s);
          // matrixU.applyOnTheRight(p,q,j_left.transpose()): columns p,q with the transposed-of-transposed
          if (want_u) rot_apply(&out.U.a[static_cast<size_t>(p) * out.U.rows], 1, &out.U.a[static_cast<size_t>(q) * out.U.rows], 1,
                                out.U.rows, jl.c, jl.s);
          // workMatrix.applyOnTheRight(p,q,j_right): columns p,q with j_right.transpose()
          rot_apply(&W.a[static_cast<size_t>(p) * diag], 1, &W.a[static_cast<size_t>(q) * diag], 1, diag, jr.c, -jr.s);
          rot_apply(&out.V.a[static_cast<size_t>(p) * out.V.rows], 1, &out.V.a[static_cast<size_t>(q) * out.V.rows], 1,
                    out.V.rows, jr.c, -jr.s);
          const double app = std::fabs(W(p, p)), aqq = std::fabs(W(q, q));
          const double mm = app > aqq ? app : aqq;
          if (mm > max_diag) max_diag = mm;
        }
      }
    }
  }
  out.sv.resize(diag);
  for (int i = 0; i < diag; ++i) {
    const double a = W(i, i);
    out.sv[i] = std::fabs(a);
    if (want_u && a < 0.0)
      for (int r = 0; r < out.U.rows; ++r) out.U(r, i) = -out.U(r, i);
  }
  for (int i = 0; i < diag; ++i) out.sv[i] *= scale;
  for (int i = 0; i < diag; ++i) {
    int pos = i;
    double mx = out.sv[i];
    for (int j = i + 1; j < diag; ++j)
      if (out.sv[j] > mx) {
        mx = out.sv[j];
        pos = j;
      }
    if (mx == 0.0) break;
    if (pos != i) {
      std::swap(out.sv[i], out.sv[pos]);
      if (want_u)
        for (int r = 0; r < out.U.rows; ++r) std::swap(out.U(r, i), out.U(r, pos));
      for (int r = 0; r < out.V.rows; ++r) std::swap(out.V(r, i), out.V(r, pos));
    }
  }
  return out;
}

// (tools/sim_roots_lanes.py builds the oracle with -DORACLE_EIG_TRACE to record, per polynomial, the sequence of deflations
// and Francis steps with their windows -- the input of its model of how 64 lanes of k_roots_e share a wave)
#ifdef ORACLE_EIG_TRACE
extern "C" void oracle_eig_trace(int kind, int il, int imm, int iu);
#define ORACLE_EIG_EVENT(kind, il, imm, iu) oracle_eig_trace(kind, il, imm, iu)
#else
#define ORACLE_EIG_EVENT(kind, il, imm, iu) ((void)0)
#endif
// ---- EigenSolver (eigenvalues only): Hessenberg + RealSchur (Eigen/src/Eigenvalues) ----------
// Returns false when the QR iteration does not converge (info() != Success).
inline bool real_eigenvalues(const Mat& Cin, std::vector<double>* re, std::vector<double>* im) {
  const int n = Cin.rows;
  re->assign(n, 0.0);
  im->assign(n, 0.0);
  if (n == 0) return true;
  double scale = 0.0;
  for (double v : Cin.a)
    if (std::fabs(v) > scale) scale = std::fabs(v);
  Mat T = Cin;
  if (scale < DBL_MIN) {
    return true;  // zero matrix: all eigenvalues zero
  }
  for (double& v : T.a) v /= scale;
  // Hessenberg reduction (HessenbergDecomposition::_compute)
  for (int i = 0; i < n - 1; ++i) {
    const int rem = n - i - 1;
    double tau, beta;
    make_householder(&T(i + 1, i), rem, &tau, &beta);
    T(i + 1, i) = beta;
    std::vector<double> ess(rem > 1 ? rem - 1 : 0);
    for (int k = 0; k + 1 < rem; ++k) ess[k] = T(i + 2 + k, i);
    apply_householder_left(T, i + 1, i + 1, rem, rem, ess.data(), tau);
    apply_householder_right(T, 0, i + 1, n, rem, ess.data(), tau);
  }
  for (int j = 0; j < n; ++j)
    for (int i = j + 2; i < n; ++i) T(i, j) = 0.0;  // matrixH(): below the sub-diagonal is zero

  // RealSchur::computeFromHessenberg
  const int max_iters = 40 * n;
  int iu = n - 1, iter = 0, total_iter = 0;
  double exshift = 0.0;
  double norm = 0.0;
  for (int j = 0; j < n; ++j) {
    const int lim = (j + 2 < n) ? j + 2 : n;
    for (int i = 0; i < lim; ++i) norm += std::fabs(T(i, j));
  }
  if (norm != 0.0) {
    while (iu >= 0) {
      // findSmallSubdiagEntry
      int il = iu;
      while (il > 0) {
        double s = std::fabs(T(il - 1, il - 1)) + std::fabs(T(il, il));
        if (s == 0.0) s = norm;
        if (std::fabs(T(il, il - 1)) < DBL_EPSILON * s) break;
        il--;
      }
      if (il == iu) {
        ORACLE_EIG_EVENT(1, il, il, iu);
        T(iu, iu) = T(iu, iu) + exshift;
        if (iu > 0) T(iu, iu - 1) = 0.0;
        iu--;
        iter = 0;
      } else if (il == iu - 1) {
        ORACLE_EIG_EVENT(2, il, il, iu);
        // splitOffTwoRows
        const double p = 0.5 * (T(iu - 1, iu - 1) - T(iu, iu));
        const double q = p * p + T(iu, iu - 1) * T(iu - 1, iu);
        T(iu, iu) += exshift;
        T(iu - 1, iu - 1) += exshift;
        if (q >= 0.0) {
          const double z = std::sqrt(std::fabs(q));
          // rot.makeGivens(p +/- z, T(iu,iu-1))
          const double gp = (p >= 0.0) ? (p + z) : (p - z);
          const double gq = T(iu, iu - 1);
          double gc, gs;
          if (gq == 0.0) {
            gc = gp < 0.0 ? -1.0 : 1.0;
            gs = 0.0;
          } else if (gp == 0.0) {
            gc = 0.0;
            gs = gq < 0.0 ? 1.0 : -1.0;
          } else if (std::fabs(gp) > std::fabs(gq)) {
            const double t = gq / gp;
            double u = std::sqrt(1.0 + t * t);
            if (gp < 0.0) u = -u;
            gc = 1.0 / u;
            gs = -t * gc;
          } else {
            const double t = gp / gq;
            double u = std::sqrt(1.0 + t * t);
            if (gq < 0.0) u = -u;
            gs = -1.0 / u;
            gc = -t * gs;
          }
          // T.rightCols(n-iu+1).applyOnTheLeft(iu-1, iu, rot.adjoint()); adjoint = (c, -s)
          rot_apply(&T.a[static_cast<size_t>(iu - 1) * n + (iu - 1)], n, &T.a[static_cast<size_t>(iu - 1) * n + iu], n,
                    n - iu + 1, gc, -gs);
          // T.topRows(iu+1).applyOnTheRight(iu-1, iu, rot): uses rot.transpose() = (c, -s)
          rot_apply(&T.a[static_cast<size_t>(iu - 1) * n], 1, &T.a[static_cast<size_t>(iu) * n], 1, iu + 1, gc, -gs);
          T(iu, iu - 1) = 0.0;
        }
        if (iu > 1) T(iu - 1, iu - 2) = 0.0;
        iu -= 2;
        iter = 0;
      } else {
        // computeShift
        double sh0 = T(iu, iu), sh1 = T(iu - 1, iu - 1), sh2 = T(iu, iu - 1) * T(iu - 1, iu);
        if (iter == 10) {
          exshift += sh0;
          for (int i = 0; i <= iu; ++i) T(i, i) -= sh0;
          const double s = std::fabs(T(iu, iu - 1)) + std::fabs(T(iu - 1, iu - 2));
          sh0 = 0.75 * s;
          sh1 = 0.75 * s;
          sh2 = -0.4375 * s * s;
        }
        if (iter == 30) {
          double s = (sh1 - sh0) / 2.0;
          s = s * s + sh2;
          if (s > 0.0) {
            s = std::sqrt(s);
            if (sh1 < sh0) s = -s;
            s = s + (sh1 - sh0) / 2.0;
            s = sh0 - sh2 / s;
            exshift += s;
            for (int i = 0; i <= iu; ++i) T(i, i) -= s;
            sh0 = sh1 = sh2 = 0.964;
          }
        }
        iter = iter + 1;
        total_iter = total_iter + 1;
        if (total_iter > max_iters) break;
        // initFrancisQRStep
        int imm;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        for (imm = iu - 2; imm >= il; --imm) {
          const double Tmm = T(imm, imm);
          const double r = sh0 - Tmm;
          const double s = sh1 - Tmm;
          v0 = (r * s - sh2) / T(imm + 1, imm) + T(imm, imm + 1);
          v1 = T(imm + 1, imm + 1) - Tmm - r - s;
          v2 = T(imm + 2, imm + 1);
          if (imm == il) break;
          const double lhs = T(imm, imm - 1) * (std::fabs(v1) + std::fabs(v2));
          const double rhs = v0 * (std::fabs(T(imm - 1, imm - 1)) + std::fabs(Tmm) + std::fabs(T(imm + 1, imm + 1)));
          if (std::fabs(lhs) < DBL_EPSILON * rhs) break;
        }
        ORACLE_EIG_EVENT(3, il, imm, iu);
        // performFrancisQRStep
        for (int k = imm; k <= iu - 2; ++k) {
          const bool first = (k == imm);
          double v[3];
          if (first) {
            v[0] = v0;
            v[1] = v1;
            v[2] = v2;
          } else {
            v[0] = T(k, k - 1);
            v[1] = T(k + 1, k - 1);
            v[2] = T(k + 2, k - 1);
          }
          double tau, beta;
          make_householder(v, 3, &tau, &beta);
          if (beta != 0.0) {
            if (first && k > il)
              T(k, k - 1) = -T(k, k - 1);
            else if (!first)
              T(k, k - 1) = beta;
            apply_householder_left(T, k, k, 3, n - k, &v[1], tau);
            const int nr = ((iu < k + 3) ? iu : k + 3) + 1;
            apply_householder_right(T, 0, k, nr, 3, &v[1], tau);
          }
        }
        {
          double v[2] = {T(iu - 1, iu - 2), T(iu, iu - 2)};
          double tau, beta;
          make_householder(v, 2, &tau, &beta);
          if (beta != 0.0) {
            T(iu - 1, iu - 2) = beta;
            apply_householder_left(T, iu - 1, iu - 1, 2, n - iu + 1, &v[1], tau);
            apply_householder_right(T, 0, iu - 1, iu + 1, 2, &v[1], tau);
          }
        }
        for (int i = imm + 2; i <= iu; ++i) {
          T(i, i - 2) = 0.0;
          if (i > imm + 2) T(i, i - 3) = 0.0;
        }
      }
    }
  }
  if (total_iter > max_iters) return false;
  for (double& v : T.a) v *= scale;
  // EigenSolver::compute: eigenvalues off the quasi-triangular T
  int i = 0;
  while (i < n) {
    if (i == n - 1 || T(i + 1, i) == 0.0) {
      (*re)[i] = T(i, i);
      (*im)[i] = 0.0;
      if (!std::isfinite((*re)[i])) return false;
      ++i;
    } else {
      const double p = 0.5 * (T(i, i) - T(i + 1, i + 1));
      double z;
      {
        double t0 = T(i + 1, i), t1 = T(i, i + 1);
        double maxval = std::fabs(p);
        if (std::fabs(t0) > maxval) maxval = std::fabs(t0);
        if (std::fabs(t1) > maxval) maxval = std::fabs(t1);
        t0 /= maxval;
        t1 /= maxval;
        const double p0 = p / maxval;
        z = maxval * std::sqrt(std::fabs(p0 * p0 + t0 * t1));
      }
      (*re)[i] = T(i + 1, i + 1) + p;
      (*im)[i] = z;
      (*re)[i + 1] = T(i + 1, i + 1) + p;
      (*im)[i + 1] = -z;
      if (!(std::isfinite((*re)[i]) && std::isfinite(z))) return false;
      i += 2;
    }
  }
  return true;
}

// ---- PartialPivLU::solve for an n x n system with m right-hand sides (column-major) ---------
inline void partial_piv_lu_solve(Mat A, Mat B, Mat* X) {
  const int n = A.rows, m = B.cols;
  std::vector<int> piv(n);
  for (int k = 0; k < n; ++k) {
    int r = k;
    double best = std::fabs(A(k, k));
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A(i, k)) > best) {
        best = std::fabs(A(i, k));
        r = i;
      }
    piv[k] = r;
    if (best != 0.0) {
      if (r != k) {
        for (int j = 0; j < n; ++j) std::swap(A(k, j), A(r, j));
      }
      for (int i = k + 1; i < n; ++i) A(i, k) /= A(k, k);
    }
    for (int j = k + 1; j < n; ++j)
      for (int i = k + 1; i < n; ++i) A(i, j) -= A(i, k) * A(k, j);
  }
  for (int k = 0; k < n; ++k)
    if (piv[k] != k)
      for (int j = 0; j < m; ++j) std::swap(B(k, j), B(piv[k], j));
  for (int j = 0; j < m; ++j) {
    for (int i = 0; i < n; ++i) {  // unit lower
      double s = B(i, j);
      for (int k = 0; k < i; ++k) s -= A(i, k) * B(k, j);
      B(i, j) = s;
    }
    for (int i = n - 1; i >= 0; --i) {  // upper
      double s = B(i, j);
      for (int k = n - 1; k > i; --k) s -= A(i, k) * B(k, j);  // column-oriented back substitution
      B(i, j) = s / A(i, i);
    }
  }
  *X = B;
}

// ---- small fixed-size helpers (row-major 3x3 as double[9]) --------------------------------------
struct Mat3 {
  double m[9];
  double& operator()(int r, int c) { return m[r * 3 + c]; }
  double operator()(int r, int c) const { return m[r * 3 + c]; }
};
struct Vec3 {
  double v[3];
};

inline Mat3 mat3_mul(const Mat3& A, const Mat3& B) {
  Mat3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
  return C;
}
inline Mat3 mat3_transpose(const Mat3& A) {
  Mat3 T;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T(i, j) = A(j, i);
  return T;
}
inline double mat3_det(const Mat3& A) {  // Eigen determinant_impl<.,3>: bruteforce_det3_helper
  return A(0, 0) * (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) - A(0, 1) * (A(1, 0) * A(2, 2) - A(1, 2) * A(2, 0)) +
         A(0, 2) * (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0));
}
// Matrix3d::inverse(): cofactors with a single 1/det (Eigen/src/LU/InverseImpl.h)
inline Mat3 mat3_inverse(const Mat3& M) {
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
  };
  const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  const double det = c00 * M(0, 0) + c10 * M(1, 0) + c20 * M(2, 0);
  const double invdet = 1.0 / det;
  Mat3 R;
  R(0, 0) = c00 * invdet;
  R(0, 1) = c10 * invdet;
  R(0, 2) = c20 * invdet;
  R(1, 0) = cof(0, 1) * invdet;
  R(1, 1) = cof(1, 1) * invdet;
  R(2, 2) = cof(2, 2) * invdet;
  R(1, 2) = cof(2, 1) * invdet;
  R(2, 1) = cof(1, 2) * invdet;
  R(2, 0) = cof(0, 2) * invdet;
  return R;
}
inline Mat to_mat(const Mat3& A) {
  Mat M(3, 3);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M(i, j) = A(i, j);
  return M;
}
inline Mat3 to_mat3(const Mat& M) {
  Mat3 A;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A(i, j) = M(i, j);
  return A;
}

// Quaterniond(Matrix3d) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<.,3,3>),
// returned as (w, x, y, z).
inline void rotation_to_quaternion(const Mat3& R, double q[4]) {
  double t = R(0, 0) + R(1, 1) + R(2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R(2, 1) - R(1, 2)) * t;
    q[2] = (R(0, 2) - R(2, 0)) * t;
    q[3] = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R(k, j) - R(j, k)) * t;
    q[1 + j] = (R(j, i) + R(i, j)) * t;
    q[1 + k] = (R(k, i) + R(i, k)) * t;
  }
}

}  // namespace oracle
#endif  // ORACLE_LINALG_H_
