// verify_linalg.h -- device-side FP64 linear algebra of the two-view verification kernels.
//
// Two flavours, both with the *same floating-point operation order* as the CPU path they are
// checked against (reductions accumulate sequentially in index order; -ffp-contract=off):
//   pl_*  "per lane": each lane owns a small problem in private memory (minimal solvers:
//         the 9x5 / 9x7 / 9x8 pivoted QR behind the 5-, 7- and 4-point solvers, 3x3 / 4x4 Jacobi
//         SVD, companion-matrix eigenvalues, 10x10 LU).
//   wv_*  "per wave": the 64 lanes of a one-wave workgroup cooperate on one problem (the tall
//         n x 9 systems of the local-optimisation estimators): element-wise updates are spread
//         over the lanes, every dot product / norm is summed by ONE lane in index order.
//
// Algorithms: Eigen 3.3's JacobiSVD with ColPivHouseholderQR preconditioner, RealSchur
// (Francis double-shift QR), PartialPivLU -- the routines the reference calls at
//   /root/reference/src/estimators/fundamental_matrix.cc:74,174,181
//   /root/reference/src/estimators/essential_matrix.cc:71,80,130
//   /root/reference/src/estimators/homography_matrix.cc:83
//   /root/reference/src/base/polynomial.cc:250, base/essential_matrix.cc:43, base/triangulation.cc:50
// For a matrix with more columns than rows the null-space columns of V come straight from the
// Householder Q of the transposed matrix and are never touched by the Jacobi sweeps, so the
// minimal solvers run the pivoted QR only.
#ifndef DAGSFM_AMD_CSRC_VERIFY_LINALG_H_
#define DAGSFM_AMD_CSRC_VERIFY_LINALG_H_

#include <hip/hip_runtime.h>
#include <float.h>

#include <type_traits>
#include <stdint.h>

#define DSM_DEV __device__ __forceinline__
#define DSM_DEVN __device__ __noinline__

// ====================================================================== several quotients with one denominator
// x / d for many x and one d (JacobiSVD's m / scale over a whole matrix; a Householder tail over c0 - beta): the correctly
// rounded quotient, computed by the hardware's OWN division sequence with the part that depends on d alone done once.
// hipcc expands an FP64 division into v_div_scale (x2), v_rcp_f64, four refinement FMAs of the reciprocal, one multiply,
// one residual FMA, v_div_fmas and v_div_fixup -- 11 VALU, one of them quarter rate.  When no operand needs scaling
// (v_div_scale returns its input, VCC = 0, so v_div_fmas is a plain FMA) and no special case applies (v_div_fixup passes
// the result through), that is: r = refine(refine(rcp(d))); q = x r; e = fma(-d, q, x); result = fma(e, r, q).  The first
// part is shared_divisor(), the second 3 VALU per quotient in div_shared_fast() -- the SAME instructions on the same
// values, hence the same bits.  The range in which the sequence needs neither scaling nor fix-up (v_div_scale_f64: numerator
// exponent above 53, exponent difference below 768, neither 1/d nor x/d denormal) contains |d| in [2^-300, 2^300] with
// |x| in [2^-600, 2^400]; a group with a numerator outside it (zero, tiny, NaN-free huge) takes the plain divisions.
struct SharedDivisor {
  double d, r;
  bool ok;
};
DSM_DEV SharedDivisor shared_divisor(double d) {
  SharedDivisor s;
  s.d = d;
  const double ad = fabs(d);
  s.ok = (ad >= 0x1p-300) && (ad <= 0x1p300);  // false for NaN
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  s.r = r;
  return s;
}
// the three instructions of one quotient (only inside a group that passed div_shared_group_ok)
DSM_DEV double div_shared_fast(double x, const SharedDivisor& s) {
  const double q = x * s.r;
  const double e = __builtin_fma(-s.d, q, x);
  return __builtin_fma(e, s.r, q);
}
// ONE guard for a whole group of numerators: `min_abs` = the smallest |x| of the group (NaN numerators do not lower it:
// they give NaN on either path).  The callers' numerators are bounded by the denominator's own construction (|x| <= scale;
// a Householder tail is bounded by the norm that made c0 - beta), so a finite in-range d implies finite numerators and
// only the lower end needs looking at.  A per-quotient guard would cost a branch per division and eat the gain.
DSM_DEV bool div_shared_group_ok(const SharedDivisor& s, double min_abs) { return s.ok && (min_abs >= 0x1p-600); }

// ====================================================================== per-lane routines
// Column-major storage with leading dimension ld, like Eigen.

// ES = element stride of x (see pl_apply_householder_left)
template <int ES = 1>
DSM_DEV void pl_make_householder(double* x, int n, double* tau, double* beta) {
  double tail_sq = 0.0;
  for (int i = 1; i < n; ++i) tail_sq += x[i * ES] * x[i * ES];
  const double c0 = x[0];
  if (tail_sq <= DBL_MIN) {
    *tau = 0.0;
    *beta = c0;
    for (int i = 1; i < n; ++i) x[i * ES] = 0.0;
  } else {
    double b = sqrt(c0 * c0 + tail_sq);
    if (c0 >= 0.0) b = -b;
    for (int i = 1; i < n; ++i) x[i * ES] = x[i * ES] / (c0 - b);
    *tau = (b - c0) / b;
    *beta = b;
  }
}

// ES = element stride: 1 for a private matrix, 64 for a matrix whose elements are interleaved over the
// lanes of a wave in LDS (element e of lane l at base[e * 64 + l], M = base + l)
// ESS = element stride of the essential part (it may live inside the strided matrix itself)
template <int ES = 1, int ESS = 1>
DSM_DEV void pl_apply_householder_left(double* M, int ld, int r0, int c0, int nr, int nc, const double* ess, double tau) {
  if (nr == 1) {
    for (int j = 0; j < nc; ++j) M[((c0 + j) * ld + r0) * ES] *= (1.0 - tau);
  } else if (tau != 0.0) {
    for (int j = 0; j < nc; ++j) {
      double* col = M + ((c0 + j) * ld + r0) * ES;
      double tmp = 0.0;
      for (int i = 1; i < nr; ++i) tmp += ess[(i - 1) * ESS] * col[i * ES];
      tmp += col[0];
      col[0] -= tau * tmp;
      for (int i = 1; i < nr; ++i) col[i * ES] -= tau * ess[(i - 1) * ESS] * tmp;
    }
  }
}

template <int ES = 1>
DSM_DEV void pl_apply_householder_right(double* M, int ld, int r0, int c0, int nr, int nc, const double* ess, double tau) {
  if (nc == 1) {
    for (int i = 0; i < nr; ++i) M[(c0 * ld + r0 + i) * ES] *= (1.0 - tau);
  } else if (tau != 0.0) {
    for (int i = 0; i < nr; ++i) {
      double tmp = 0.0;
      for (int j = 1; j < nc; ++j) tmp += M[((c0 + j) * ld + r0 + i) * ES] * ess[j - 1];
      tmp += M[(c0 * ld + r0 + i) * ES];
      M[(c0 * ld + r0 + i) * ES] -= tau * tmp;
      for (int j = 1; j < nc; ++j) M[((c0 + j) * ld + r0 + i) * ES] -= tau * tmp * ess[j - 1];
    }
  }
}

// ColPivHouseholderQR::computeInPlace on qr (rows x cols, rows >= cols here), ld = rows.
template <int ES = 1>
DSM_DEV void pl_colpiv_qr(double* qr, int rows, int cols, double* hcoeffs) {
#define QR(i) qr[(i) * ES]
  const int size = rows < cols ? rows : cols;
  double norms_updated[9], norms_direct[9];
  for (int k = 0; k < cols; ++k) {
    double s = 0.0;
    for (int i = 0; i < rows; ++i) s += QR(k * rows + i) * QR(k * rows + i);
    norms_direct[k] = sqrt(s);
    norms_updated[k] = norms_direct[k];
  }
  const double norm_downdate_threshold = sqrt(DBL_EPSILON);
  for (int k = 0; k < size; ++k) {
    int biggest = k;
    double mx = norms_updated[k];
    for (int j = k + 1; j < cols; ++j)
      if (norms_updated[j] > mx) {
        mx = norms_updated[j];
        biggest = j;
      }
    if (k != biggest) {
      for (int i = 0; i < rows; ++i) {
        const double t = QR(k * rows + i);
        QR(k * rows + i) = QR(biggest * rows + i);
        QR(biggest * rows + i) = t;
      }
      double t = norms_updated[k];
      norms_updated[k] = norms_updated[biggest];
      norms_updated[biggest] = t;
      t = norms_direct[k];
      norms_direct[k] = norms_direct[biggest];
      norms_direct[biggest] = t;
    }
    double tau, beta;
    pl_make_householder<ES>(qr + (k * rows + k) * ES, rows - k, &tau, &beta);
    hcoeffs[k] = tau;
    QR(k * rows + k) = beta;
    pl_apply_householder_left<ES, ES>(qr, rows, k, k + 1, rows - k, cols - k - 1, qr + (k * rows + k + 1) * ES, tau);
    for (int j = k + 1; j < cols; ++j) {
      if (norms_updated[j] != 0.0) {
        double temp = fabs(QR(j * rows + k)) / norms_updated[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double ratio = norms_updated[j] / norms_direct[j];
        const double temp2 = temp * (ratio * ratio);
        if (temp2 <= norm_downdate_threshold) {
          double s = 0.0;
          for (int i = k + 1; i < rows; ++i) s += QR(j * rows + i) * QR(j * rows + i);
          norms_direct[j] = sqrt(s);
          norms_updated[j] = norms_direct[j];
        } else {
          norms_updated[j] *= sqrt(temp);
        }
      }
    }
  }
#undef QR
}

// Column j of householderQ() (rows x rows) of a pivoted QR with `size` reflectors.
template <int ES = 1>
DSM_DEV void pl_householder_q_col(const double* qr, int rows, int size, const double* hcoeffs, int j, double* q) {
  for (int i = 0; i < rows; ++i) q[i] = (i == j) ? 1.0 : 0.0;
  for (int k = size - 1; k >= 0; --k) {
    if (j < k) continue;  // the block starts at column k
    const int nr = rows - k;
    const double tau = hcoeffs[k];
    const double* ess = qr + (k * rows + k + 1) * ES;
    if (nr == 1) {
      q[k] *= (1.0 - tau);
    } else if (tau != 0.0) {
      double tmp = 0.0;
      for (int i = 1; i < nr; ++i) tmp += ess[(i - 1) * ES] * q[k + i];
      tmp += q[k];
      q[k] -= tau * tmp;
      for (int i = 1; i < nr; ++i) q[k + i] -= tau * ess[(i - 1) * ES] * tmp;
    }
  }
}

// Null-space columns of V for a `m x 9` system with m < 9 (JacobiSVD "more columns than rows"
// path): V = householderQ of ColPivQR((A / scale)^T).  At holds A^T (9 x m, column-major, ld 9)
// and is destroyed.  Writes columns first..8 of V into out[(c - first) * 9 + r].
// ES: element stride of At (1 = private array, 64 = lane-interleaved LDS).
template <int ES = 1>
DSM_DEV void pl_nullspace_9xm(double* At, int m, int first, double* out) {
  double scale = 0.0;
  for (int i = 0; i < 9 * m; ++i) {
    const double a = fabs(At[i * ES]);
    if (a > scale) scale = a;
  }
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < 9 * m; ++i) At[i * ES] /= scale;
  double hco[8];
  pl_colpiv_qr<ES>(At, 9, m, hco);
  for (int c = first; c < 9; ++c) pl_householder_q_col<ES>(At, 9, m, hco, c, out + (c - first) * 9);
}

// ---------------------------------------------------------------------- register-resident forms (static indices only)
// pl_colpiv_qr / pl_householder_q_col / pl_nullspace_9xm for a 9 x M matrix held in registers: every loop is unrolled
// over its static range and the one data-dependent index -- the pivot column -- becomes a chain of predicated column
// swaps.  Same operations in the same order as the pl_ routines (which the single-kernel reference schedule keeps).
template <int M>
DSM_DEV void pr_colpiv_qr9(double (&qr)[9 * M], double (&hco)[M]) {
#define QRE(i, k) qr[(k) * 9 + (i)]
  double norms_updated[M], norms_direct[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += QRE(i, k) * QRE(i, k);
    norms_direct[k] = sqrt(s);
    norms_updated[k] = norms_direct[k];
  }
  const double norm_downdate_threshold = sqrt(DBL_EPSILON);
#pragma unroll
  for (int k = 0; k < M; ++k) {
    int biggest = k;
    double mx = norms_updated[k];
#pragma unroll
    for (int j = k + 1; j < M; ++j) {
      if (norms_updated[j] > mx) {
        mx = norms_updated[j];
        biggest = j;
      }
    }
#pragma unroll
    for (int j = k + 1; j < M; ++j) {  // unconditional stores of selected values (see pr_jacobi_sweeps9)
      const bool sw = (j == biggest);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double a = QRE(i, k), b = QRE(i, j);
        QRE(i, k) = sw ? b : a;
        QRE(i, j) = sw ? a : b;
      }
      const double ua = norms_updated[k], ub = norms_updated[j];
      norms_updated[k] = sw ? ub : ua;
      norms_updated[j] = sw ? ua : ub;
      const double da = norms_direct[k], db = norms_direct[j];
      norms_direct[k] = sw ? db : da;
      norms_direct[j] = sw ? da : db;
    }
    // pl_make_householder on rows k..8 of column k
    double tail_sq = 0.0;
#pragma unroll
    for (int i = k + 1; i < 9; ++i) tail_sq += QRE(i, k) * QRE(i, k);
    const double c0 = QRE(k, k);
    double tau, beta;
    if (tail_sq <= DBL_MIN) {
      tau = 0.0;
      beta = c0;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) QRE(i, k) = 0.0;
    } else {
      double b = sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0.0) b = -b;
      const SharedDivisor hd = shared_divisor(c0 - b);
      double mn = 0x1p1000;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) mn = fmin(mn, fabs(QRE(i, k)));
      if (div_shared_group_ok(hd, mn)) {
#pragma unroll
        for (int i = k + 1; i < 9; ++i) QRE(i, k) = div_shared_fast(QRE(i, k), hd);
      } else {
#pragma unroll
        for (int i = k + 1; i < 9; ++i) QRE(i, k) = QRE(i, k) / (c0 - b);
      }
      tau = (b - c0) / b;
      beta = b;
    }
    hco[k] = tau;
    QRE(k, k) = beta;
    if (tau != 0.0) {  // rows k..8 (never a single row: k <= M - 1 <= 7)
#pragma unroll
      for (int j = k + 1; j < M; ++j) {
        double tmp = 0.0;
#pragma unroll
        for (int i = k + 1; i < 9; ++i) tmp += QRE(i, k) * QRE(i, j);
        tmp += QRE(k, j);
        QRE(k, j) -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < 9; ++i) QRE(i, j) -= tau * QRE(i, k) * tmp;
      }
    }
#pragma unroll
    for (int j = k + 1; j < M; ++j) {
      if (norms_updated[j] != 0.0) {
        double temp = fabs(QRE(k, j)) / norms_updated[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double ratio = norms_updated[j] / norms_direct[j];
        const double temp2 = temp * (ratio * ratio);
        if (temp2 <= norm_downdate_threshold) {
          double s = 0.0;
#pragma unroll
          for (int i = k + 1; i < 9; ++i) s += QRE(i, j) * QRE(i, j);
          norms_direct[j] = sqrt(s);
          norms_updated[j] = norms_direct[j];
        } else {
          norms_updated[j] *= sqrt(temp);
        }
      }
    }
  }
#undef QRE
}

// column J of householderQ() (9 x 9) of the pivoted QR above
template <int M, int J>
DSM_DEV void pr_householder_q_col9(const double (&qr)[9 * M], const double (&hco)[M], double (&q)[9]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) q[i] = (i == J) ? 1.0 : 0.0;
#pragma unroll
  for (int k = M - 1; k >= 0; --k) {
    if (J < k) continue;  // the block starts at column k
    const double tau = hco[k];
    if (tau != 0.0) {
      double tmp = 0.0;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) tmp += qr[k * 9 + i] * q[i];
      tmp += q[k];
      q[k] -= tau * tmp;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) q[i] -= tau * qr[k * 9 + i] * tmp;
    }
  }
}

// pl_nullspace_9xm for an M-column A^T in registers: columns M..8 of V into out[(c - M) * 9 + r]
template <int M>
DSM_DEV void pr_nullspace_9xm(double (&At)[9 * M], double (&out)[(9 - M) * 9]) {
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < 9 * M; ++i) {
    const double a = fabs(At[i]);
    if (a > scale) scale = a;
  }
  if (scale == 0.0) scale = 1.0;
  {
    const SharedDivisor sd = shared_divisor(scale);
    double mn = 0x1p1000;
#pragma unroll
    for (int i = 0; i < 9 * M; ++i) mn = fmin(mn, fabs(At[i]));
    if (div_shared_group_ok(sd, mn)) {
#pragma unroll
      for (int i = 0; i < 9 * M; ++i) At[i] = div_shared_fast(At[i], sd);
    } else {
#pragma unroll
      for (int i = 0; i < 9 * M; ++i) At[i] /= scale;
    }
  }
  double hco[M];
  pr_colpiv_qr9<M>(At, hco);
  double q[9];
  if constexpr (M <= 8) {
    pr_householder_q_col9<M, 8>(At, hco, q);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[(8 - M) * 9 + i] = q[i];
  }
  if constexpr (M <= 7) {
    pr_householder_q_col9<M, 7>(At, hco, q);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[(7 - M) * 9 + i] = q[i];
  }
  if constexpr (M <= 6) {
    pr_householder_q_col9<M, 6>(At, hco, q);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[(6 - M) * 9 + i] = q[i];
  }
  if constexpr (M <= 5) {
    pr_householder_q_col9<M, 5>(At, hco, q);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[(5 - M) * 9 + i] = q[i];
  }
}

// makeJacobi(x, y, z)
DSM_DEV void dsm_make_jacobi(double x, double y, double z, double* c, double* s) {
  const double deno = 2.0 * fabs(y);
  if (deno < DBL_MIN) {
    *c = 1.0;
    *s = 0.0;
  } else {
    const double tau = (x - z) / deno;
    const double w = sqrt(tau * tau + 1.0);
    double t;
    if (tau > 0.0)
      t = 1.0 / (tau + w);
    else
      t = 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / sqrt(t * t + 1.0);
    // y / |y| (Eigen's makeJacobi) is exactly +-1 for a finite y (never zero here: deno >= DBL_MIN): the sign, without the
    // division sequence; an infinite or NaN y takes the division and its NaN
    const double y_sign = fabs(y) <= DBL_MAX ? copysign(1.0, y) : y / fabs(y);
    *s = -sign_t * y_sign * fabs(t) * n;
    *c = n;
  }
}

// real_2x2_jacobi_svd: left rotation (lc, ls) and right rotation (rc, rs) for the block
// [[wpp, wpq], [wqp, wqq]].
DSM_DEV void dsm_jacobi_2x2(double m00, double m01, double m10, double m11, double* lc, double* ls, double* rc, double* rs) {
  double r1c, r1s;
  const double t = m00 + m11;
  const double d = m10 - m01;
  if (fabs(d) < DBL_MIN) {
    r1s = 0.0;
    r1c = 1.0;
  } else {
    const double u = t / d;
    const double tmp = sqrt(1.0 + u * u);
    const SharedDivisor sd = shared_divisor(tmp);  // the two quotients by tmp (see the top of this file)
    if (div_shared_group_ok(sd, fabs(u))) {
      r1s = div_shared_fast(1.0, sd);
      r1c = div_shared_fast(u, sd);
    } else {
      r1s = 1.0 / tmp;
      r1c = u / tmp;
    }
  }
  const double a0 = r1c * m00 + r1s * m10, a1 = r1c * m01 + r1s * m11;
  const double b1 = -r1s * m01 + r1c * m11;
  double jc, js;
  dsm_make_jacobi(a0, a1, b1, &jc, &js);
  *rc = jc;
  *rs = js;
  *lc = r1c * jc - r1s * (-js);
  *ls = r1c * (-js) + r1s * jc;
}

// JacobiSVD of a square N x N matrix (no preconditioner).  A is row-major in, W/U/V column-major
// scratch; outputs singular values (descending), U and V column-major.
template <int N, bool WANT_U>
DSM_DEV void pl_jacobi_svd_square(const double* A_rowmajor, double* U, double* V, double* sv) {
  double W[N * N];
  double scale = 0.0;
  for (int i = 0; i < N * N; ++i) {
    const double a = fabs(A_rowmajor[i]);
    if (a > scale) scale = a;
  }
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      W[j * N + i] = A_rowmajor[i * N + j] / scale;
      V[j * N + i] = (i == j) ? 1.0 : 0.0;
      if (WANT_U) U[j * N + i] = (i == j) ? 1.0 : 0.0;
    }
  const double precision = 2.0 * DBL_EPSILON;
  double max_diag = 0.0;
  for (int i = 0; i < N; ++i)
    if (fabs(W[i * N + i]) > max_diag) max_diag = fabs(W[i * N + i]);
  bool finished = false;
  while (!finished) {
    finished = true;
    for (int p = 1; p < N; ++p) {
      for (int q = 0; q < p; ++q) {
        const double thr = DBL_MIN > precision * max_diag ? DBL_MIN : precision * max_diag;
        if (fabs(W[q * N + p]) > thr || fabs(W[p * N + q]) > thr) {
          finished = false;
          double lc, ls, rc, rs;
          dsm_jacobi_2x2(W[p * N + p], W[q * N + p], W[p * N + q], W[q * N + q], &lc, &ls, &rc, &rs);
          if (!(lc == 1.0 && ls == 0.0)) {
            for (int j = 0; j < N; ++j) {  // rows p, q
              const double xi = W[j * N + p], yi = W[j * N + q];
              W[j * N + p] = lc * xi + ls * yi;
              W[j * N + q] = -ls * xi + lc * yi;
            }
            if (WANT_U)
              for (int i = 0; i < N; ++i) {  // columns p, q of U
                const double xi = U[p * N + i], yi = U[q * N + i];
                U[p * N + i] = lc * xi + ls * yi;
                U[q * N + i] = -ls * xi + lc * yi;
              }
          }
          if (!(rc == 1.0 && -rs == 0.0)) {
            for (int i = 0; i < N; ++i) {  // columns p, q with (rc, -rs)
              const double xi = W[p * N + i], yi = W[q * N + i];
              W[p * N + i] = rc * xi + (-rs) * yi;
              W[q * N + i] = rs * xi + rc * yi;
            }
            for (int i = 0; i < N; ++i) {
              const double xi = V[p * N + i], yi = V[q * N + i];
              V[p * N + i] = rc * xi + (-rs) * yi;
              V[q * N + i] = rs * xi + rc * yi;
            }
          }
          const double app = fabs(W[p * N + p]), aqq = fabs(W[q * N + q]);
          const double mm = app > aqq ? app : aqq;
          if (mm > max_diag) max_diag = mm;
        }
      }
    }
  }
  for (int i = 0; i < N; ++i) {
    const double a = W[i * N + i];
    sv[i] = fabs(a);
    if (WANT_U && a < 0.0)
      for (int r = 0; r < N; ++r) U[i * N + r] = -U[i * N + r];
  }
  for (int i = 0; i < N; ++i) sv[i] *= scale;
  for (int i = 0; i < N; ++i) {
    int pos = i;
    double mx = sv[i];
    for (int j = i + 1; j < N; ++j)
      if (sv[j] > mx) {
        mx = sv[j];
        pos = j;
      }
    if (mx == 0.0) break;
    if (pos != i) {
      double t = sv[i];
      sv[i] = sv[pos];
      sv[pos] = t;
      for (int r = 0; r < N; ++r) {
        if (WANT_U) {
          t = U[i * N + r];
          U[i * N + r] = U[pos * N + r];
          U[pos * N + r] = t;
        }
        t = V[i * N + r];
        V[i * N + r] = V[pos * N + r];
        V[pos * N + r] = t;
      }
    }
  }
}

// pl_jacobi_svd_square<N, false> with every array in registers (all loops unrolled, the sort's column swaps as
// selected values): the right factor only.
template <int N>
DSM_DEV void pr_jacobi_svd_square_V(const double (&A_rowmajor)[N * N], double (&V)[N * N], double (&sv)[N]) {
  double W[N * N];
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < N * N; ++i) {
    const double a = fabs(A_rowmajor[i]);
    if (a > scale) scale = a;
  }
  if (scale == 0.0) scale = 1.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      W[j * N + i] = A_rowmajor[i * N + j] / scale;
      V[j * N + i] = (i == j) ? 1.0 : 0.0;
    }
  }
  const double precision = 2.0 * DBL_EPSILON;
  double max_diag = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (fabs(W[i * N + i]) > max_diag) max_diag = fabs(W[i * N + i]);
  bool finished = false;
  while (!finished) {
    finished = true;
#pragma unroll
    for (int p = 1; p < N; ++p) {
#pragma unroll
      for (int q = 0; q < p; ++q) {
        const double thr = DBL_MIN > precision * max_diag ? DBL_MIN : precision * max_diag;
        if (fabs(W[q * N + p]) > thr || fabs(W[p * N + q]) > thr) {
          finished = false;
          double lc, ls, rc, rs;
          dsm_jacobi_2x2(W[p * N + p], W[q * N + p], W[p * N + q], W[q * N + q], &lc, &ls, &rc, &rs);
          if (!(lc == 1.0 && ls == 0.0)) {
#pragma unroll
            for (int j = 0; j < N; ++j) {  // rows p, q
              const double xi = W[j * N + p], yi = W[j * N + q];
              W[j * N + p] = lc * xi + ls * yi;
              W[j * N + q] = -ls * xi + lc * yi;
            }
          }
          if (!(rc == 1.0 && -rs == 0.0)) {
#pragma unroll
            for (int i = 0; i < N; ++i) {  // columns p, q with (rc, -rs)
              const double xi = W[p * N + i], yi = W[q * N + i];
              W[p * N + i] = rc * xi + (-rs) * yi;
              W[q * N + i] = rs * xi + rc * yi;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
              const double xi = V[p * N + i], yi = V[q * N + i];
              V[p * N + i] = rc * xi + (-rs) * yi;
              V[q * N + i] = rs * xi + rc * yi;
            }
          }
          const double app = fabs(W[p * N + p]), aqq = fabs(W[q * N + q]);
          const double mm = app > aqq ? app : aqq;
          if (mm > max_diag) max_diag = mm;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) sv[i] = fabs(W[i * N + i]) * scale;
  bool stop = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (!stop) {
      int pos = i;
      double mx = sv[i];
#pragma unroll
      for (int j = i + 1; j < N; ++j) {
        if (sv[j] > mx) {
          mx = sv[j];
          pos = j;
        }
      }
      if (mx == 0.0) {
        stop = true;
      } else {
#pragma unroll
        for (int j = i + 1; j < N; ++j) {  // unconditional stores of selected values (see pr_jacobi_sweeps9)
          const bool sw = (j == pos);
          const double si = sv[i], sj = sv[j];
          sv[i] = sw ? sj : si;
          sv[j] = sw ? si : sj;
#pragma unroll
          for (int r = 0; r < N; ++r) {
            const double a = V[i * N + r], b = V[j * N + r];
            V[i * N + r] = sw ? b : a;
            V[j * N + r] = sw ? a : b;
          }
        }
      }
    }
  }
}

// Eigenvalues of an upper-Hessenberg n x n matrix T (column-major, ld = LD, destroyed):
// EigenSolver(C, false) for companion matrices (the Hessenberg reduction is the identity on
// them: every sub-sub-diagonal entry is already zero).  Returns false on non-convergence.
// VALUES_ONLY: the reflectors are not applied to the columns right of the active block (c > iu).  Those columns
// are deflated for good (iu only decreases) and their entries above the diagonal blocks never feed back into a
// diagonal block, so the eigenvalues keep their bits (LAPACK's job = 'E'); Eigen itself updates them.
template <int LD, int ES, bool VALUES_ONLY = false>
DSM_DEV bool pl_hessenberg_eigenvalues_impl(double* T, int n, double* re, double* im) {
#define TT(r, c) T[((c) * LD + (r)) * ES]
  for (int i = 0; i < n; ++i) {
    re[i] = 0.0;
    im[i] = 0.0;
  }
  if (n == 0) return true;
  double scale = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      const double a = fabs(TT(i, j));
      if (a > scale) scale = a;
    }
  if (scale < DBL_MIN) return true;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) TT(i, j) /= scale;
  const int max_iters = 40 * n;
  int iu = n - 1, iter = 0, total_iter = 0;
  double exshift = 0.0;
  double norm = 0.0;
  for (int j = 0; j < n; ++j) {
    const int lim = (j + 2 < n) ? j + 2 : n;
    for (int i = 0; i < lim; ++i) norm += fabs(TT(i, j));
  }
  if (norm != 0.0) {
    while (iu >= 0) {
      int il = iu;
      while (il > 0) {
        double s = fabs(TT(il - 1, il - 1)) + fabs(TT(il, il));
        if (s == 0.0) s = norm;
        if (fabs(TT(il, il - 1)) < DBL_EPSILON * s) break;
        il--;
      }
      if (il == iu) {
        TT(iu, iu) = TT(iu, iu) + exshift;
        if (iu > 0) TT(iu, iu - 1) = 0.0;
        iu--;
        iter = 0;
      } else if (il == iu - 1) {
        const double p = 0.5 * (TT(iu - 1, iu - 1) - TT(iu, iu));
        const double q = p * p + TT(iu, iu - 1) * TT(iu - 1, iu);
        TT(iu, iu) += exshift;
        TT(iu - 1, iu - 1) += exshift;
        if (q >= 0.0) {
          const double z = sqrt(fabs(q));
          const double gp = (p >= 0.0) ? (p + z) : (p - z);
          const double gq = TT(iu, iu - 1);
          double gc, gs;
          if (gq == 0.0) {
            gc = gp < 0.0 ? -1.0 : 1.0;
            gs = 0.0;
          } else if (gp == 0.0) {
            gc = 0.0;
            gs = gq < 0.0 ? 1.0 : -1.0;
          } else if (fabs(gp) > fabs(gq)) {
            const double t = gq / gp;
            double u = sqrt(1.0 + t * t);
            if (gp < 0.0) u = -u;
            gc = 1.0 / u;
            gs = -t * gc;
          } else {
            const double t = gp / gq;
            double u = sqrt(1.0 + t * t);
            if (gq < 0.0) u = -u;
            gs = -1.0 / u;
            gc = -t * gs;
          }
          if (!(gc == 1.0 && -gs == 0.0)) {
            for (int c = iu - 1; c < (VALUES_ONLY ? iu + 1 : n); ++c) {  // rows iu-1, iu with (gc, -gs)
              const double xi = TT(iu - 1, c), yi = TT(iu, c);
              TT(iu - 1, c) = gc * xi + (-gs) * yi;
              TT(iu, c) = gs * xi + gc * yi;
            }
            for (int r = 0; r <= iu; ++r) {  // columns iu-1, iu with (gc, -gs)
              const double xi = TT(r, iu - 1), yi = TT(r, iu);
              TT(r, iu - 1) = gc * xi + (-gs) * yi;
              TT(r, iu) = gs * xi + gc * yi;
            }
          }
          TT(iu, iu - 1) = 0.0;
        }
        if (iu > 1) TT(iu - 1, iu - 2) = 0.0;
        iu -= 2;
        iter = 0;
      } else {
        double sh0 = TT(iu, iu), sh1 = TT(iu - 1, iu - 1), sh2 = TT(iu, iu - 1) * TT(iu - 1, iu);
        if (iter == 10) {
          exshift += sh0;
          for (int i = 0; i <= iu; ++i) TT(i, i) -= sh0;
          const double s = fabs(TT(iu, iu - 1)) + fabs(TT(iu - 1, iu - 2));
          sh0 = 0.75 * s;
          sh1 = 0.75 * s;
          sh2 = -0.4375 * s * s;
        }
        if (iter == 30) {
          double s = (sh1 - sh0) / 2.0;
          s = s * s + sh2;
          if (s > 0.0) {
            s = sqrt(s);
            if (sh1 < sh0) s = -s;
            s = s + (sh1 - sh0) / 2.0;
            s = sh0 - sh2 / s;
            exshift += s;
            for (int i = 0; i <= iu; ++i) TT(i, i) -= s;
            sh0 = sh1 = sh2 = 0.964;
          }
        }
        iter = iter + 1;
        total_iter = total_iter + 1;
        if (total_iter > max_iters) break;
        int imm;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        for (imm = iu - 2; imm >= il; --imm) {
          const double Tmm = TT(imm, imm);
          const double r = sh0 - Tmm;
          const double s = sh1 - Tmm;
          v0 = (r * s - sh2) / TT(imm + 1, imm) + TT(imm, imm + 1);
          v1 = TT(imm + 1, imm + 1) - Tmm - r - s;
          v2 = TT(imm + 2, imm + 1);
          if (imm == il) break;
          const double lhs = TT(imm, imm - 1) * (fabs(v1) + fabs(v2));
          const double rhs = v0 * (fabs(TT(imm - 1, imm - 1)) + fabs(Tmm) + fabs(TT(imm + 1, imm + 1)));
          if (fabs(lhs) < DBL_EPSILON * rhs) break;
        }
        for (int k = imm; k <= iu - 2; ++k) {
          const bool first = (k == imm);
          double v[3];
          if (first) {
            v[0] = v0;
            v[1] = v1;
            v[2] = v2;
          } else {
            v[0] = TT(k, k - 1);
            v[1] = TT(k + 1, k - 1);
            v[2] = TT(k + 2, k - 1);
          }
          double tau, beta;
          pl_make_householder(v, 3, &tau, &beta);
          if (beta != 0.0) {
            if (first && k > il)
              TT(k, k - 1) = -TT(k, k - 1);
            else if (!first)
              TT(k, k - 1) = beta;
            pl_apply_householder_left<ES>(T, LD, k, k, 3, (VALUES_ONLY ? iu + 1 : n) - k, &v[1], tau);
            const int nr = ((iu < k + 3) ? iu : k + 3) + 1;
            pl_apply_householder_right<ES>(T, LD, 0, k, nr, 3, &v[1], tau);
          }
        }
        {
          double v[2] = {TT(iu - 1, iu - 2), TT(iu, iu - 2)};
          double tau, beta;
          pl_make_householder(v, 2, &tau, &beta);
          if (beta != 0.0) {
            TT(iu - 1, iu - 2) = beta;
            pl_apply_householder_left<ES>(T, LD, iu - 1, iu - 1, 2, (VALUES_ONLY ? iu + 1 : n) - iu + 1, &v[1], tau);
            pl_apply_householder_right<ES>(T, LD, 0, iu - 1, iu + 1, 2, &v[1], tau);
          }
        }
        for (int i = imm + 2; i <= iu; ++i) {
          TT(i, i - 2) = 0.0;
          if (i > imm + 2) TT(i, i - 3) = 0.0;
        }
      }
    }
  }
  if (total_iter > max_iters) return false;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) TT(i, j) *= scale;
  int i = 0;
  while (i < n) {
    if (i == n - 1 || TT(i + 1, i) == 0.0) {
      re[i] = TT(i, i);
      im[i] = 0.0;
      if (!isfinite(re[i])) return false;
      ++i;
    } else {
      const double p = 0.5 * (TT(i, i) - TT(i + 1, i + 1));
      double t0 = TT(i + 1, i), t1 = TT(i, i + 1);
      double maxval = fabs(p);
      if (fabs(t0) > maxval) maxval = fabs(t0);
      if (fabs(t1) > maxval) maxval = fabs(t1);
      t0 /= maxval;
      t1 /= maxval;
      const double p0 = p / maxval;
      const double z = maxval * sqrt(fabs(p0 * p0 + t0 * t1));
      re[i] = TT(i + 1, i + 1) + p;
      im[i] = z;
      re[i + 1] = TT(i + 1, i + 1) + p;
      im[i + 1] = -z;
      if (!(isfinite(re[i]) && isfinite(z))) return false;
      i += 2;
    }
  }
  return true;
#undef TT
}
template <int LD>
DSM_DEVN bool pl_hessenberg_eigenvalues(double* T, int n, double* re, double* im) {
  return pl_hessenberg_eigenvalues_impl<LD, 1>(T, n, re, im);
}

// pl_hessenberg_eigenvalues_impl<.., VALUES_ONLY = true> with the matrix in REGISTERS: every index into T is a
// compile-time constant (all loops over rows / columns / the bulge position are unrolled over their full static range)
// and the lane's own window (il, imm, iu) only appears in predicates, so T never needs an addressable home (LDS costs
// 51 KB per wave and a round trip per element; scratch costs HBM).  The price is that a wave executes the union of
// what its lanes need.  Same operations on the same values in the same order per lane, with two liberties that
// cannot reach an eigenvalue (both already taken by VALUES_ONLY's argument above): the left reflectors also update
// the deflated columns c > iu, whose entries above the diagonal blocks never feed back; entries below the third
// sub-diagonal are never formed (they are exact zeros in the reference iteration too).
// T: column-major N x N, the caller fills rows <= column + 1 (upper Hessenberg) and zeros elsewhere; n <= N.
template <int N>
DSM_DEV bool pr_hessenberg_eigenvalues(double (&T)[N * N], int n, double (&re)[N], double (&im)[N]) {
#define RT(r, c) T[(c) * N + (r)]
#pragma unroll
  for (int i = 0; i < N; ++i) {
    re[i] = 0.0;
    im[i] = 0.0;
  }
  if (n == 0) return true;
  double scale = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i <= j + 1) {  // the rest is zero: it cannot raise the maximum
        const double a = fabs(RT(i, j));
        if (a > scale) scale = a;
      }
    }
  }
  if (scale < DBL_MIN) return true;
  {
    // (a companion matrix is mostly exact zeros: its group never passes the guard; kept as the plain division)
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i <= j + 1) RT(i, j) /= scale;
    }
  }
  const int max_iters = 40 * n;
  int iu = n - 1, iter = 0, total_iter = 0;
  double exshift = 0.0;
  double norm = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i <= j + 1) norm += fabs(RT(i, j));  // entries outside n x n are zeros: + 0.0 changes nothing
  }
  bool failed = false;
  // 2 x 2 blocks split off with exshift == 0 whose Givens step is still owed (bit I: the block of rows I - 1, I); see the note at
  // the split below
  unsigned pending = 0;
  if (norm != 0.0) {
    while (iu >= 0) {
      // il = the largest L <= iu with a negligible sub-diagonal entry (L, L-1), else 0
      int il = 0;
      {
        bool found = false;
#pragma unroll
        for (int L = N - 1; L >= 1; --L) {
          if (!found && L <= iu) {
            double s = fabs(RT(L - 1, L - 1)) + fabs(RT(L, L));
            if (s == 0.0) s = norm;
            if (fabs(RT(L, L - 1)) < DBL_EPSILON * s) {
              il = L;
              found = true;
            }
          }
        }
      }
      if (il == iu) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          if (i == iu) {
            RT(i, i) = RT(i, i) + exshift;
            if (i > 0) RT(i, i - 1) = 0.0;
          }
        }
        iu--;
        iter = 0;
      } else if (il == iu - 1 && exshift == 0.0) {
        // A 2 x 2 block splits off (RealSchur::splitOffTwoRows).  Its Givens step -- a square root, two divisions, the rotation of two
        // rows and two columns: ~450 instructions -- touches the block itself and entries ABOVE it (rows < iu - 1 of its two columns),
        // and nothing the iteration does afterwards reads or writes the block again (the window ends at iu - 2; the left reflectors'
        // updates of the deflated columns stay in rows <= iu - 2, which never feed back).  In lockstep a wave paid those 450
        // instructions in nearly every iteration, because SOME lane splits a block off (3.7 per polynomial x 64 lanes over ~32
        // iterations: 14 400 of a wave's 66 500 instruction slots, tools/sim_roots_lanes.py).  So the step is OWED here (one bit) and
        // paid after the loop, where all lanes take it together, on exactly the values it would have seen -- provided exshift is 0
        // (it is, unless this lane took an exceptional shift: then the branch below runs the step at once, as before).  Same
        // operations on the same operands: the block's four entries, hence its eigenvalues, are bit for bit the same.
        pending |= 1u << iu;
#pragma unroll
        for (int I = 2; I < N; ++I)
          if (I == iu) RT(I - 1, I - 2) = 0.0;
        iu -= 2;
        iter = 0;
      } else if (il == iu - 1) {
        double a11 = 0.0, a22 = 0.0, a21 = 0.0, a12 = 0.0;  // (iu-1, iu-1), (iu, iu), (iu, iu-1), (iu-1, iu)
#pragma unroll
        for (int I = 1; I < N; ++I) {
          if (I == iu) {
            a11 = RT(I - 1, I - 1);
            a22 = RT(I, I);
            a21 = RT(I, I - 1);
            a12 = RT(I - 1, I);
          }
        }
        const double p = 0.5 * (a11 - a22);
        const double q = p * p + a21 * a12;
#pragma unroll
        for (int I = 1; I < N; ++I) {
          if (I == iu) {
            RT(I, I) += exshift;
            RT(I - 1, I - 1) += exshift;
          }
        }
        if (q >= 0.0) {
          const double z = sqrt(fabs(q));
          const double gp = (p >= 0.0) ? (p + z) : (p - z);
          const double gq = a21;
          double gc, gs;
          if (gq == 0.0) {
            gc = gp < 0.0 ? -1.0 : 1.0;
            gs = 0.0;
          } else if (gp == 0.0) {
            gc = 0.0;
            gs = gq < 0.0 ? 1.0 : -1.0;
          } else if (fabs(gp) > fabs(gq)) {
            const double t = gq / gp;
            double u = sqrt(1.0 + t * t);
            if (gp < 0.0) u = -u;
            gc = 1.0 / u;
            gs = -t * gc;
          } else {
            const double t = gp / gq;
            double u = sqrt(1.0 + t * t);
            if (gq < 0.0) u = -u;
            gs = -1.0 / u;
            gc = -t * gs;
          }
          const bool rotate = !(gc == 1.0 && -gs == 0.0);
#pragma unroll
          for (int I = 1; I < N; ++I) {
            if (I == iu) {
              if (rotate) {
#pragma unroll
                for (int c = I - 1; c <= I; ++c) {  // rows iu-1, iu with (gc, -gs)
                  const double xi = RT(I - 1, c), yi = RT(I, c);
                  RT(I - 1, c) = gc * xi + (-gs) * yi;
                  RT(I, c) = gs * xi + gc * yi;
                }
#pragma unroll
                for (int r = 0; r <= I; ++r) {  // columns iu-1, iu with (gc, -gs)
                  const double xi = RT(r, I - 1), yi = RT(r, I);
                  RT(r, I - 1) = gc * xi + (-gs) * yi;
                  RT(r, I) = gs * xi + gc * yi;
                }
              }
              RT(I, I - 1) = 0.0;
            }
          }
        }
#pragma unroll
        for (int I = 2; I < N; ++I)
          if (I == iu) RT(I - 1, I - 2) = 0.0;
        iu -= 2;
        iter = 0;
      } else {
        // here iu >= il + 2 >= 2
        double sh0 = 0.0, sh1 = 0.0, sh2 = 0.0, sub_abs = 0.0;
#pragma unroll
        for (int I = 2; I < N; ++I) {
          if (I == iu) {
            sh0 = RT(I, I);
            sh1 = RT(I - 1, I - 1);
            sh2 = RT(I, I - 1) * RT(I - 1, I);
            sub_abs = fabs(RT(I, I - 1)) + fabs(RT(I - 1, I - 2));
          }
        }
        if (iter == 10) {
          exshift += sh0;
#pragma unroll
          for (int i = 0; i < N; ++i)
            if (i <= iu) RT(i, i) -= sh0;
          const double s = sub_abs;
          sh0 = 0.75 * s;
          sh1 = 0.75 * s;
          sh2 = -0.4375 * s * s;
        }
        if (iter == 30) {
          double s = (sh1 - sh0) / 2.0;
          s = s * s + sh2;
          if (s > 0.0) {
            s = sqrt(s);
            if (sh1 < sh0) s = -s;
            s = s + (sh1 - sh0) / 2.0;
            s = sh0 - sh2 / s;
            exshift += s;
#pragma unroll
            for (int i = 0; i < N; ++i)
              if (i <= iu) RT(i, i) -= s;
            sh0 = sh1 = sh2 = 0.964;
          }
        }
        iter = iter + 1;
        total_iter = total_iter + 1;
        if (total_iter > max_iters) {
          failed = true;
          break;
        }
        int imm = 0;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        {
          bool found = false;
#pragma unroll
          for (int M = N - 3; M >= 0; --M) {
            if (!found && M <= iu - 2 && M >= il) {
              const double Tmm = RT(M, M);
              const double r = sh0 - Tmm;
              const double s = sh1 - Tmm;
              v0 = (r * s - sh2) / RT(M + 1, M) + RT(M, M + 1);
              v1 = RT(M + 1, M + 1) - Tmm - r - s;
              v2 = RT(M + 2, M + 1);
              imm = M;
              if (M == il) {
                found = true;
              } else if (M >= 1) {  // M > il >= 0
                const double lhs = RT(M, M - 1) * (fabs(v1) + fabs(v2));
                const double rhs = v0 * (fabs(RT(M - 1, M - 1)) + fabs(Tmm) + fabs(RT(M + 1, M + 1)));
                if (fabs(lhs) < DBL_EPSILON * rhs) found = true;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k <= N - 2; ++k) {
          if (k <= N - 3 && k >= imm && k <= iu - 2) {
            {
              const bool first = (k == imm);
              double x0, x1, x2;
              if (first || k == 0) {
                x0 = v0;
                x1 = v1;
                x2 = v2;
              } else {
                x0 = RT(k, k > 0 ? k - 1 : 0);
                x1 = RT(k + 1 <= N - 1 ? k + 1 : N - 1, k > 0 ? k - 1 : 0);
                x2 = RT(k + 2 <= N - 1 ? k + 2 : N - 1, k > 0 ? k - 1 : 0);
              }
              // pl_make_householder(v, 3)
              const double tail_sq = (0.0 + x1 * x1) + x2 * x2;
              double tau, beta;
              if (tail_sq <= DBL_MIN) {
                tau = 0.0;
                beta = x0;
                x1 = 0.0;
                x2 = 0.0;
              } else {
                double b = sqrt(x0 * x0 + tail_sq);
                if (x0 >= 0.0) b = -b;
                x1 = x1 / (x0 - b);
                x2 = x2 / (x0 - b);
                tau = (b - x0) / b;
                beta = b;
              }
              if (beta != 0.0) {
                if (k > 0) {
                  if (first && k > il)
                    RT(k, k > 0 ? k - 1 : 0) = -RT(k, k > 0 ? k - 1 : 0);
                  else if (!first)
                    RT(k, k > 0 ? k - 1 : 0) = beta;
                }
                if (tau != 0.0) {
#pragma unroll
                  for (int c = k; c < N; ++c) {  // rows k..k+2
                    const int r1 = k + 1 <= N - 1 ? k + 1 : N - 1, r2 = k + 2 <= N - 1 ? k + 2 : N - 1;
                    double tmp = 0.0;
                    tmp += x1 * RT(r1, c);
                    tmp += x2 * RT(r2, c);
                    tmp += RT(k, c);
                    RT(k, c) -= tau * tmp;
                    RT(r1, c) -= tau * x1 * tmp;
                    RT(r2, c) -= tau * x2 * tmp;
                  }
#pragma unroll
                  for (int r = 0; r <= k + 3; ++r) {  // columns k..k+2, rows 0..min(iu, k+3)
                    if (r <= N - 1 && (r <= k + 2 || r <= iu)) {
                      const int c1 = k + 1 <= N - 1 ? k + 1 : N - 1, c2 = k + 2 <= N - 1 ? k + 2 : N - 1;
                      const int rr = r <= N - 1 ? r : N - 1;
                      double tmp = 0.0;
                      tmp += RT(rr, c1) * x1;
                      tmp += RT(rr, c2) * x2;
                      tmp += RT(rr, k);
                      RT(rr, k) -= tau * tmp;
                      RT(rr, c1) -= tau * tmp * x1;
                      RT(rr, c2) -= tau * tmp * x2;
                    }
                  }
                }
              }
            }
          } else if (k >= 1 && k == iu - 1) {
            const int kc = k > 0 ? k - 1 : 0, k1 = k + 1 <= N - 1 ? k + 1 : N - 1;
            double x0 = RT(k, kc), x1 = RT(k1, kc);
            // pl_make_householder(v, 2)
            const double tail_sq = 0.0 + x1 * x1;
            double tau, beta;
            if (tail_sq <= DBL_MIN) {
              tau = 0.0;
              beta = x0;
              x1 = 0.0;
            } else {
              double b = sqrt(x0 * x0 + tail_sq);
              if (x0 >= 0.0) b = -b;
              x1 = x1 / (x0 - b);
              tau = (b - x0) / b;
              beta = b;
            }
            if (beta != 0.0) {
              RT(k, kc) = beta;
              if (tau != 0.0) {
#pragma unroll
                for (int c = k; c < N; ++c) {  // rows k, k+1
                  double tmp = 0.0;
                  tmp += x1 * RT(k1, c);
                  tmp += RT(k, c);
                  RT(k, c) -= tau * tmp;
                  RT(k1, c) -= tau * x1 * tmp;
                }
#pragma unroll
                for (int r = 0; r <= k + 1; ++r) {  // columns k, k+1, rows 0..iu (= k+1)
                  const int rr = r <= N - 1 ? r : N - 1;
                  double tmp = 0.0;
                  tmp += RT(rr, k1) * x1;
                  tmp += RT(rr, k);
                  RT(rr, k) -= tau * tmp;
                  RT(rr, k1) -= tau * tmp * x1;
                }
              }
            }
          }
        }
#pragma unroll
        for (int i = 2; i < N; ++i) {
          if (i >= imm + 2 && i <= iu) {
            RT(i, i - 2) = 0.0;
            if (i >= 3 && i > imm + 2) RT(i, i - 3) = 0.0;
          }
        }
      }
    }
  }
  if (failed) return false;
  if (pending) {
    // the owed Givens steps (exshift was 0 when each block split off: `+ zero` is the reference's `+= exshift`, kept for -0.0)
    double zero = 0.0;
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int I = 1; I < N; ++I) {
      if ((pending >> I) & 1u) {
        const double a11 = RT(I - 1, I - 1), a22 = RT(I, I), a21 = RT(I, I - 1), a12 = RT(I - 1, I);
        const double p = 0.5 * (a11 - a22);
        const double q = p * p + a21 * a12;
        RT(I, I) += zero;
        RT(I - 1, I - 1) += zero;
        if (q >= 0.0) {
          const double z = sqrt(fabs(q));
          const double gp = (p >= 0.0) ? (p + z) : (p - z);
          const double gq = a21;
          double gc, gs;
          if (gq == 0.0) {
            gc = gp < 0.0 ? -1.0 : 1.0;
            gs = 0.0;
          } else if (gp == 0.0) {
            gc = 0.0;
            gs = gq < 0.0 ? 1.0 : -1.0;
          } else if (fabs(gp) > fabs(gq)) {
            const double t = gq / gp;
            double u = sqrt(1.0 + t * t);
            if (gp < 0.0) u = -u;
            gc = 1.0 / u;
            gs = -t * gc;
          } else {
            const double t = gp / gq;
            double u = sqrt(1.0 + t * t);
            if (gq < 0.0) u = -u;
            gs = -1.0 / u;
            gc = -t * gs;
          }
          const bool rotate = !(gc == 1.0 && -gs == 0.0);
          if (rotate) {
#pragma unroll
            for (int c = I - 1; c <= I; ++c) {  // rows I-1, I with (gc, -gs)
              const double xi = RT(I - 1, c), yi = RT(I, c);
              RT(I - 1, c) = gc * xi + (-gs) * yi;
              RT(I, c) = gs * xi + gc * yi;
            }
#pragma unroll
            for (int r = I - 1; r <= I; ++r) {  // columns I-1, I with (gc, -gs): the block's own rows (the rows above never feed back)
              const double xi = RT(r, I - 1), yi = RT(r, I);
              RT(r, I - 1) = gc * xi + (-gs) * yi;
              RT(r, I) = gs * xi + gc * yi;
            }
          }
          RT(I, I - 1) = 0.0;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i <= j + 1) RT(i, j) *= scale;
  }
  bool ok = true, second = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (ok && i < n) {
      if (second) {
        second = false;  // imaginary partner of the pair that started at i - 1: already written
      } else {
        const double sub = (i + 1 <= N - 1) ? RT(i + 1 <= N - 1 ? i + 1 : N - 1, i) : 0.0;
        if (i == n - 1 || sub == 0.0) {
          re[i] = RT(i, i);
          im[i] = 0.0;
          if (!isfinite(re[i])) ok = false;
        } else {
          const int i1 = i + 1 <= N - 1 ? i + 1 : N - 1;
          const double p = 0.5 * (RT(i, i) - RT(i1, i1));
          double t0 = RT(i1, i), t1 = RT(i, i1);
          double maxval = fabs(p);
          if (fabs(t0) > maxval) maxval = fabs(t0);
          if (fabs(t1) > maxval) maxval = fabs(t1);
          t0 /= maxval;
          t1 /= maxval;
          const double p0 = p / maxval;
          const double z = maxval * sqrt(fabs(p0 * p0 + t0 * t1));
          re[i] = RT(i1, i1) + p;
          im[i] = z;
          re[i1] = RT(i1, i1) + p;
          im[i1] = -z;
          if (!(isfinite(re[i]) && isfinite(z))) ok = false;
          second = true;
        }
      }
    }
  }
  return ok;
#undef RT
}

// FindPolynomialRootsCompanionMatrix (/root/reference/src/base/polynomial.cc:208-275) for up to
// MAXC coefficients (highest degree first).  Returns the number of roots, or -1 on failure.
// ES > 1: the companion matrix lives in `ws` with element stride ES (lane-interleaved LDS) and the
// eigenvalue iteration is inlined into the caller, so that its accesses compile to LDS instructions.
template <int MAXC, int ES = 1>
DSM_DEV int pl_poly_roots(const double* coeffs_all, int ncoef, double* real, double* imag, double* ws = nullptr) {
  int lead = 0;
  for (; lead < ncoef; ++lead)
    if (coeffs_all[lead] != 0) break;
  const double* coeffs = coeffs_all + lead;
  int nc = ncoef - lead;
  const int degree = nc - 1;
  if (degree <= 0) return -1;
  if (degree == 1) {  // FindLinearPolynomialRoots
    if (coeffs[0] == 0) return -1;
    real[0] = -coeffs[1] / coeffs[0];
    imag[0] = 0.0;
    return 1;
  }
  if (degree == 2) {  // FindQuadraticPolynomialRoots (a != 0 here)
    const double a = coeffs[0], b = coeffs[1], c = coeffs[2];
    if (b == 0 && c == 0) {
      real[0] = 0.0;
      imag[0] = 0.0;
      return 1;
    }
    const double d = b * b - 4 * a * c;
    if (d >= 0) {
      const double sqrt_d = sqrt(d);
      if (b >= 0) {
        real[0] = (-b - sqrt_d) / (2 * a);
        real[1] = (2 * c) / (-b - sqrt_d);
      } else {
        real[0] = (2 * c) / (-b + sqrt_d);
        real[1] = (-b + sqrt_d) / (2 * a);
      }
      imag[0] = 0.0;
      imag[1] = 0.0;
    } else {
      real[0] = real[1] = -b / (2 * a);
      imag[0] = sqrt(-d) / (2 * a);
      imag[1] = -imag[0];
    }
    return 2;
  }
  int trail = 0;
  for (; trail < nc; ++trail)
    if (coeffs[nc - 1 - trail] != 0) break;
  nc -= trail;
  if (nc == 1) {
    real[0] = 0.0;
    imag[0] = 0.0;
    return 1;
  }
  const int n = nc - 1;
  constexpr int LD = MAXC - 1;
  double C_loc[ES == 1 ? LD * LD : 1];
  double* C;  // optional caller-provided (LDS) workspace of LD*LD doubles
  if constexpr (ES == 1) C = ws ? ws : C_loc; else C = ws;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) C[(j * LD + i) * ES] = 0.0;
  for (int i = 1; i < n; ++i) C[((i - 1) * LD + i) * ES] = 1.0;
  for (int j = 0; j < n; ++j) C[(j * LD + 0) * ES] = -coeffs[j + 1] / coeffs[0];
  double re[LD], im[LD];
  if (ES == 1) {
    if (!pl_hessenberg_eigenvalues<LD>(C, n, re, im)) return -1;
  } else {
    if (!pl_hessenberg_eigenvalues_impl<LD, ES, true>(C, n, re, im)) return -1;
  }
  const int effective_degree = n < degree ? n + 1 : n;
  for (int i = 0; i < effective_degree; ++i) {
    real[i] = 0.0;
    imag[i] = 0.0;
  }
  for (int i = 0; i < n; ++i) {
    real[i] = re[i];
    imag[i] = im[i];
  }
  return effective_degree;
}

// pl_poly_roots with the companion matrix in registers (pr_hessenberg_eigenvalues): no workspace at all.
template <int MAXC>
DSM_DEV int pr_poly_roots(const double (&coeffs_all)[MAXC], double (&real)[MAXC], double (&imag)[MAXC]) {
  constexpr int LD = MAXC - 1;
  // leading zeros: the polynomial the reference works on starts at the first non-zero coefficient.  Shift it to the
  // front with static indices (lead is almost always 0).
  int lead = 0;
  {
    bool stop = false;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (!stop) {
        if (coeffs_all[i] != 0)
          stop = true;
        else
          lead = i + 1;
      }
    }
  }
  double coeffs[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) coeffs[i] = coeffs_all[i];
  for (int sft = 0; sft < lead; ++sft) {  // (rare) one position per pass
#pragma unroll
    for (int i = 0; i + 1 < MAXC; ++i) coeffs[i] = coeffs[i + 1];
    coeffs[MAXC - 1] = 0.0;
  }
  int nc = MAXC - lead;
  const int degree = nc - 1;
  if (degree <= 0) return -1;
  if (degree == 1) {  // FindLinearPolynomialRoots
    if (coeffs[0] == 0) return -1;
    real[0] = -coeffs[1] / coeffs[0];
    imag[0] = 0.0;
    return 1;
  }
  if (degree == 2) {  // FindQuadraticPolynomialRoots (a != 0 here)
    const double a = coeffs[0], b = coeffs[1], c = coeffs[2];
    if (b == 0 && c == 0) {
      real[0] = 0.0;
      imag[0] = 0.0;
      return 1;
    }
    const double d = b * b - 4 * a * c;
    if (d >= 0) {
      const double sqrt_d = sqrt(d);
      if (b >= 0) {
        real[0] = (-b - sqrt_d) / (2 * a);
        real[1] = (2 * c) / (-b - sqrt_d);
      } else {
        real[0] = (2 * c) / (-b + sqrt_d);
        real[1] = (-b + sqrt_d) / (2 * a);
      }
      imag[0] = 0.0;
      imag[1] = 0.0;
    } else {
      real[0] = real[1] = -b / (2 * a);
      imag[0] = sqrt(-d) / (2 * a);
      imag[1] = -imag[0];
    }
    return 2;
  }
  // trailing zeros
  int trail = 0;
  {
    bool stop = false;
#pragma unroll
    for (int i = MAXC - 1; i >= 0; --i) {
      if (!stop && i < nc) {
        if (coeffs[i] != 0)
          stop = true;
        else
          trail += 1;
      }
    }
  }
  nc -= trail;
  if (nc == 1) {
    real[0] = 0.0;
    imag[0] = 0.0;
    return 1;
  }
  const int n = nc - 1;
  double C[LD * LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) {
#pragma unroll
    for (int i = 0; i < LD; ++i) C[j * LD + i] = 0.0;
  }
#pragma unroll
  for (int i = 1; i < LD; ++i)
    if (i < n) C[(i - 1) * LD + i] = 1.0;
#pragma unroll
  for (int j = 0; j < LD; ++j)
    if (j < n) C[j * LD + 0] = -coeffs[j + 1] / coeffs[0];
  double re[LD], im[LD];
  if (!pr_hessenberg_eigenvalues<LD>(C, n, re, im)) return -1;
  const int effective_degree = n < degree ? n + 1 : n;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    if (i < effective_degree) {
      real[i] = 0.0;
      imag[i] = 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    if (i < n) {
      real[i] = re[i];
      imag[i] = im[i];
    }
  }
  return effective_degree;
}

// A.partialPivLu().solve(B) for 10 x 10 A and B, column-major in place; the solution overwrites B.
DSM_DEVN void pl_lu_solve_10(double* A, double* B) {
  const int n = 10, m = 10;
  int piv[10];
  for (int k = 0; k < n; ++k) {
    int r = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(A[k * n + i]) > best) {
        best = fabs(A[k * n + i]);
        r = i;
      }
    piv[k] = r;
    if (best != 0.0) {
      if (r != k)
        for (int j = 0; j < n; ++j) {
          const double t = A[j * n + k];
          A[j * n + k] = A[j * n + r];
          A[j * n + r] = t;
        }
      for (int i = k + 1; i < n; ++i) A[k * n + i] /= A[k * n + k];
    }
    for (int j = k + 1; j < n; ++j)
      for (int i = k + 1; i < n; ++i) A[j * n + i] -= A[k * n + i] * A[j * n + k];
  }
  for (int k = 0; k < n; ++k)
    if (piv[k] != k)
      for (int j = 0; j < m; ++j) {
        const double t = B[j * n + k];
        B[j * n + k] = B[j * n + piv[k]];
        B[j * n + piv[k]] = t;
      }
  for (int j = 0; j < m; ++j) {
    for (int i = 0; i < n; ++i) {
      double s = B[j * n + i];
      for (int k = 0; k < i; ++k) s -= A[k * n + i] * B[j * n + k];
      B[j * n + i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = B[j * n + i];
      for (int k = n - 1; k > i; --k) s -= A[k * n + i] * B[j * n + k];
      B[j * n + i] = s / A[i * n + i];
    }
  }
}

// ---- row-major 3x3 helpers ---------------------------------------------------------------------
DSM_DEV void m3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
DSM_DEV void m3_transpose(const double* A, double* T) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[j * 3 + i];
}
DSM_DEV double m3_det(const double* A) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
DSM_DEV double m3_cof(const double* M, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return M[i1 * 3 + j1] * M[i2 * 3 + j2] - M[i1 * 3 + j2] * M[i2 * 3 + j1];
}
DSM_DEV void m3_inverse(const double* M, double* R) {
  const double c00 = m3_cof(M, 0, 0), c10 = m3_cof(M, 1, 0), c20 = m3_cof(M, 2, 0);
  const double det = c00 * M[0] + c10 * M[3] + c20 * M[6];
  const double invdet = 1.0 / det;
  R[0] = c00 * invdet;
  R[1] = c10 * invdet;
  R[2] = c20 * invdet;
  R[3] = m3_cof(M, 0, 1) * invdet;
  R[4] = m3_cof(M, 1, 1) * invdet;
  R[8] = m3_cof(M, 2, 2) * invdet;
  R[5] = m3_cof(M, 2, 1) * invdet;
  R[7] = m3_cof(M, 1, 2) * invdet;
  R[6] = m3_cof(M, 0, 2) * invdet;
}

// ====================================================================== per-wave routines
// One-wave workgroups (64 threads): __syncthreads() orders LDS/global traffic between phases.


// ---- exact in-order reductions by the whole wave ------------------------------------------------
// The sum must be accumulated in index order (one rounding per addition, like the sequential CPU
// loop).  The 64 lanes fetch 64 consecutive operands with ONE coalesced load, then every lane walks
// them in order through v_readlane (wave-uniform, a few cycles each) -- no serial chain of memory
// round trips.  All lanes end with the same bit-identical sum.
DSM_DEV double wv_readlane_f64(double v, int i) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), i);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), i);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// s = (((init + f(0)) + f(1)) + ...), f(i) supplied per lane for index base+lane by `load(i)`.
template <typename LoadFn>
DSM_DEV double wv_seq_sum(double init, int n, int lane, LoadFn load) {
  double s = init;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const double v = (i < n) ? load(i) : 0.0;
    const int cnt = (n - base) < 64 ? (n - base) : 64;
    for (int k = 0; k < cnt; ++k) s += wv_readlane_f64(v, k);
  }
  return s;
}

// The reductions INSIDE the pivoted QR of a tall matrix (rows > 9) have no order fixed by the reference
// (Eigen's depends on its packet width): oracle/linalg.h wide_sum() defines them as 64 interleaved partial
// sums -- lane l accumulates f(l), f(l + 64), ... -- combined by the xor butterfly 32, 16, ..., 1
// (s_l + s_(l^o) is commutative, so every lane ends with the same bits as the oracle's tree).
template <typename LoadFn>
DSM_DEV double wv_tree_sum(int n, int lane, LoadFn load) {
  double s = 0.0;
  for (int i = lane; i < n; i += 64) s += load(i);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  return s;
}
// up to NC such sums at once (independent butterflies overlap their latency): out[c] = sum_i load(c, i)
template <int NC, typename LoadFn>
DSM_DEV void wv_tree_sums(int nc, int n, int lane, double* out, LoadFn load) {
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  for (int i = lane; i < n; i += 64) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c < nc) acc[c] += load(c, i);
  }
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += __shfl_xor(acc[c], o);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) out[c] = acc[c];
}

struct WvSvdShared {
  double W[81];            // working square matrix, column-major ld = dsz
  double V[81];            // 9 x 9, column-major ld 9
  double norms_u[9], norms_d[9];
  double colbuf[9];        // per-column dot products
  double hco[9];
  double scal[4];          // broadcast scalars
  int perm[9];
  int iscal[4];
};

DSM_DEV void wv_sync() { __syncthreads(); }

// Pivoted Householder QR of M (rows x cols, column-major ld = rows, rows >= cols, cols <= 9) in
// global/LDS memory, by the whole wave.  Leaves R + essential parts in M, hco/perm in sh.
DSM_DEV void wv_colpiv_qr(double* M, int rows, int cols, WvSvdShared* sh, int lane) {
  const int size = rows < cols ? rows : cols;
  const bool wide = rows > 9;  // reduction order: wide_sum() tree for tall matrices, sequential otherwise
  if (wide) {
    double s[9];
    wv_tree_sums<9>(cols, rows, lane, s, [M, rows](int c, int i) {
      const double v = M[(size_t)c * rows + i];
      return v * v;
    });
#pragma unroll
    for (int c = 0; c < 9; ++c)
      if (lane == 0 && c < cols) {
        sh->norms_d[c] = sqrt(s[c]);
        sh->norms_u[c] = sh->norms_d[c];
        sh->perm[c] = c;
      }
  } else {
    for (int c = 0; c < cols; ++c) {
      const double* col = M + (size_t)c * rows;
      const double s = wv_seq_sum(0.0, rows, lane, [col](int i) { return col[i] * col[i]; });
      if (lane == 0) {
        sh->norms_d[c] = sqrt(s);
        sh->norms_u[c] = sh->norms_d[c];
        sh->perm[c] = c;
      }
    }
  }
  wv_sync();
  const double norm_downdate_threshold = sqrt(DBL_EPSILON);
  for (int k = 0; k < size; ++k) {
    int biggest = k;
    double mx = sh->norms_u[k];
    for (int j = k + 1; j < cols; ++j) {
      const double v = sh->norms_u[j];
      if (v > mx) {
        mx = v;
        biggest = j;
      }
    }
    wv_sync();
    if (k != biggest) {
      double* ck = M + (size_t)k * rows;
      double* cb = M + (size_t)biggest * rows;
      for (int i = lane; i < rows; i += 64) {
        const double t = ck[i];
        ck[i] = cb[i];
        cb[i] = t;
      }
      if (lane == 0) {
        double t = sh->norms_u[k];
        sh->norms_u[k] = sh->norms_u[biggest];
        sh->norms_u[biggest] = t;
        t = sh->norms_d[k];
        sh->norms_d[k] = sh->norms_d[biggest];
        sh->norms_d[biggest] = t;
        const int ti = sh->perm[k];
        sh->perm[k] = sh->perm[biggest];
        sh->perm[biggest] = ti;
      }
    }
    wv_sync();
    // makeHouseholder on column k, rows k..rows-1
    double* x = M + (size_t)k * rows + k;
    const int n = rows - k;
    const double tail_sq = wide ? wv_tree_sum(n - 1, lane, [x](int i) { return x[i + 1] * x[i + 1]; })
                                : wv_seq_sum(0.0, n - 1, lane, [x](int i) { return x[i + 1] * x[i + 1]; });
    if (lane == 0) {
      const double c0 = x[0];
      if (tail_sq <= DBL_MIN) {
        sh->scal[0] = 0.0;   // tau
        sh->scal[1] = c0;    // beta
        sh->scal[2] = 0.0;   // denominator unused
        sh->iscal[0] = 1;    // zero the tail
      } else {
        double b = sqrt(c0 * c0 + tail_sq);
        if (c0 >= 0.0) b = -b;
        sh->scal[0] = (b - c0) / b;
        sh->scal[1] = b;
        sh->scal[2] = c0 - b;
        sh->iscal[0] = 0;
      }
    }
    wv_sync();
    const double tau = sh->scal[0], beta = sh->scal[1], den = sh->scal[2];
    const int zero_tail = sh->iscal[0];
    for (int i = 1 + lane; i < n; i += 64) x[i] = zero_tail ? 0.0 : x[i] / den;
    if (lane == 0) {
      x[0] = beta;
      sh->hco[k] = tau;
    }
    wv_sync();
    // apply to the remaining columns (nr = n rows, nc = cols-k-1 columns)
    const int nc = cols - k - 1;
    const double* ess = x + 1;
    if (nc > 0) {
      if (n == 1) {
        if (lane < nc) M[(size_t)(k + 1 + lane) * rows + k] *= (1.0 - tau);
      } else if (tau != 0.0) {
        if (wide) {
          double tmp[8];
          const double* col0 = M + (size_t)(k + 1) * rows + k;
          wv_tree_sums<8>(nc, n - 1, lane, tmp, [ess, col0, rows](int c, int i) { return ess[i] * col0[(size_t)c * rows + i + 1]; });
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (lane == 0 && c < nc) sh->colbuf[c] = tmp[c] + col0[(size_t)c * rows];
        } else {
          for (int c = 0; c < nc; ++c) {
            const double* col = M + (size_t)(k + 1 + c) * rows + k;
            double tmp = wv_seq_sum(0.0, n - 1, lane, [ess, col](int i) { return ess[i] * col[i + 1]; });
            tmp += col[0];
            if (lane == 0) sh->colbuf[c] = tmp;
          }
        }
        wv_sync();
        if (n >= 64) {  // tall: one column after the other, lanes over the rows (no index division)
          for (int j = 0; j < nc; ++j) {
            double* col = M + (size_t)(k + 1 + j) * rows + k;
            const double tmp = sh->colbuf[j];
            for (int i = lane; i < n; i += 64) {
              if (i == 0)
                col[0] -= tau * tmp;
              else
                col[i] -= tau * ess[i - 1] * tmp;
            }
          }
        } else {
          for (int e = lane; e < nc * n; e += 64) {
            const int j = e / n, i = e - j * n;
            double* col = M + (size_t)(k + 1 + j) * rows + k;
            const double tmp = sh->colbuf[j];
            if (i == 0)
              col[0] -= tau * tmp;
            else
              col[i] -= tau * ess[i - 1] * tmp;
          }
        }
      }
      wv_sync();
      // norm downdating: lane c owns column k+1+c (independent scalar work); the rare re-computation of a
      // column norm is a wave-wide sum, done column by column for the flagged ones
      {
        bool recompute = false;
        if (lane < nc) {
          const int j = k + 1 + lane;
          const double nu = sh->norms_u[j];
          if (nu != 0.0) {
            double temp = fabs(M[(size_t)j * rows + k]) / nu;
            temp = (1.0 + temp) * (1.0 - temp);
            temp = temp < 0.0 ? 0.0 : temp;
            const double ratio = nu / sh->norms_d[j];
            const double temp2 = temp * (ratio * ratio);
            if (temp2 <= norm_downdate_threshold)
              recompute = true;
            else
              sh->norms_u[j] = nu * sqrt(temp);
          }
        }
        unsigned long long todo = __ballot(recompute);
        while (todo) {
          const int c = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int j = k + 1 + c;
          const double* col = M + (size_t)j * rows + (k + 1);
          const double ss = wide ? wv_tree_sum(rows - k - 1, lane, [col](int i) { return col[i] * col[i]; })
                                 : wv_seq_sum(0.0, rows - k - 1, lane, [col](int i) { return col[i] * col[i]; });
          if (lane == 0) {
            sh->norms_d[j] = sqrt(ss);
            sh->norms_u[j] = sh->norms_d[j];
          }
        }
      }
    }
    wv_sync();
  }
}

// Two-sided Jacobi sweeps on sh->W (dsz x dsz) accumulating right rotations into sh->V (9 x 9,
// first dsz columns), then sign/sort.  sv (LDS, >= 9 doubles) receives the singular values.
DSM_DEV void wv_jacobi_sweeps(WvSvdShared* sh, int dsz, double scale, double* sv, int lane) {
  double* W = sh->W;
  double* V = sh->V;
  const double precision = 2.0 * DBL_EPSILON;
  double max_diag = 0.0;
  for (int i = 0; i < dsz; ++i) {
    const double a = fabs(W[i * dsz + i]);
    if (a > max_diag) max_diag = a;
  }
  bool finished = false;
  while (!finished) {
    finished = true;
    for (int p = 1; p < dsz; ++p) {
      for (int q = 0; q < p; ++q) {
        const double thr = DBL_MIN > precision * max_diag ? DBL_MIN : precision * max_diag;
        const double wpq = W[q * dsz + p], wqp = W[p * dsz + q];
        if (fabs(wpq) > thr || fabs(wqp) > thr) {
          finished = false;
          double lc, ls, rc, rs;
          dsm_jacobi_2x2(W[p * dsz + p], wpq, wqp, W[q * dsz + q], &lc, &ls, &rc, &rs);
          wv_sync();
          if (!(lc == 1.0 && ls == 0.0)) {
            if (lane < dsz) {  // rows p, q: element (p, lane), (q, lane)
              const double xi = W[lane * dsz + p], yi = W[lane * dsz + q];
              W[lane * dsz + p] = lc * xi + ls * yi;
              W[lane * dsz + q] = -ls * xi + lc * yi;
            }
          }
          wv_sync();
          if (!(rc == 1.0 && -rs == 0.0)) {
            if (lane < dsz) {  // columns p, q of W
              const double xi = W[p * dsz + lane], yi = W[q * dsz + lane];
              W[p * dsz + lane] = rc * xi + (-rs) * yi;
              W[q * dsz + lane] = rs * xi + rc * yi;
            } else if (lane >= 16 && lane < 25) {  // columns p, q of V (9 rows)
              const int r = lane - 16;
              const double xi = V[p * 9 + r], yi = V[q * 9 + r];
              V[p * 9 + r] = rc * xi + (-rs) * yi;
              V[q * 9 + r] = rs * xi + rc * yi;
            }
          }
          wv_sync();
          const double app = fabs(W[p * dsz + p]), aqq = fabs(W[q * dsz + q]);
          const double mm = app > aqq ? app : aqq;
          if (mm > max_diag) max_diag = mm;
        }
      }
    }
  }
  wv_sync();
  if (lane == 0) {
    for (int i = 0; i < dsz; ++i) sv[i] = fabs(W[i * dsz + i]) * scale;
    for (int i = 0; i < dsz; ++i) {
      int pos = i;
      double mx = sv[i];
      for (int j = i + 1; j < dsz; ++j)
        if (sv[j] > mx) {
          mx = sv[j];
          pos = j;
        }
      if (mx == 0.0) break;
      if (pos != i) {
        double t = sv[i];
        sv[i] = sv[pos];
        sv[pos] = t;
        for (int r = 0; r < 9; ++r) {
          t = V[i * 9 + r];
          V[i * 9 + r] = V[pos * 9 + r];
          V[pos * 9 + r] = t;
        }
      }
    }
  }
  wv_sync();
}

// JacobiSVD<Matrix<double, Dynamic, 9>>(A, ComputeFullV).matrixV() for an m x 9 system, any m >= 1.
// A (column-major m x 9, ld = m) lives in `A`; `At` is scratch of >= 9*m doubles used when m < 9.
// Result: sh->V (9 x 9 column-major).  A is destroyed.
#ifdef DSM_PROFILE_SECTIONS
__device__ unsigned long long g_dsm_prof[16];
#define LSEC_BEGIN() const long long lt__0 = clock64()
#define LSEC_END(sec) do { if (threadIdx.x == 0) atomicAdd(&g_dsm_prof[sec], (unsigned long long)(clock64() - lt__0)); } while (0)
#define LSEC_BEGIN2() const long long lt__2 = clock64()
#define LSEC_END2(sec) do { if (threadIdx.x == 0) atomicAdd(&g_dsm_prof[sec], (unsigned long long)(clock64() - lt__2)); } while (0)
#define LSEC_BEGIN3() const long long lt__3 = clock64()
#define LSEC_END3(sec) do { if (threadIdx.x == 0) atomicAdd(&g_dsm_prof[sec], (unsigned long long)(clock64() - lt__3)); } while (0)
#define LSEC_BEGIN4() const long long lt__4 = clock64()
#define LSEC_END4(sec) do { if (threadIdx.x == 0) atomicAdd(&g_dsm_prof[sec], (unsigned long long)(clock64() - lt__4)); } while (0)
#define LSEC_BEGIN5() const long long lt__5 = clock64()
#define LSEC_END5(sec) do { if (threadIdx.x == 0) atomicAdd(&g_dsm_prof[sec], (unsigned long long)(clock64() - lt__5)); } while (0)
#else
#define LSEC_BEGIN() do {} while (0)
#define LSEC_END(sec) do {} while (0)
#define LSEC_BEGIN2() do {} while (0)
#define LSEC_END2(sec) do {} while (0)
#define LSEC_BEGIN3() do {} while (0)
#define LSEC_END3(sec) do {} while (0)
#define LSEC_BEGIN4() do {} while (0)
#define LSEC_END4(sec) do {} while (0)
#define LSEC_BEGIN5() do {} while (0)
#define LSEC_END5(sec) do {} while (0)
#endif
// First half: everything up to the Jacobi sweeps -- leaves the square working matrix in sh->W (dsz x dsz, column-major
// ld = dsz) and the accumulated right factor in sh->V; returns dsz, *scale_out = the scaling factor.
DSM_DEV int wv_svd_prepare_mx9(double* A, double* At, int m, WvSvdShared* sh, double* scale_out, int lane) {
  // scale = max |a_ij| (exact, order independent)
  double mxl = 0.0;
  for (int e = lane; e < 9 * m; e += 64) {
    const double a = fabs(A[e]);
    if (a > mxl) mxl = a;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_xor(mxl, o);
    if (other > mxl) mxl = other;
  }
  double scale = mxl;
  if (scale == 0.0) scale = 1.0;
  *scale_out = scale;
  wv_sync();
  if (m > 9) {
    for (int e = lane; e < 9 * m; e += 64) A[e] /= scale;
    wv_sync();
    {
      LSEC_BEGIN();
      wv_colpiv_qr(A, m, 9, sh, lane);
      LSEC_END(4);
    }
    for (int e = lane; e < 81; e += 64) {
      const int i = e % 9, j = e / 9;
      sh->W[j * 9 + i] = (i <= j) ? A[(size_t)j * m + i] : 0.0;
      sh->V[e] = 0.0;
    }
    wv_sync();
    if (lane < 9) sh->V[lane * 9 + sh->perm[lane]] = 1.0;
    wv_sync();
    return 9;
  } else if (m < 9) {
    for (int e = lane; e < 9 * m; e += 64) {
      const int i = e % m, j = e / m;  // A(i, j)
      At[(size_t)i * 9 + j] = A[(size_t)j * m + i] / scale;
    }
    wv_sync();
    wv_colpiv_qr(At, 9, m, sh, lane);
    for (int e = lane; e < m * m; e += 64) {
      const int j = e % m, i = e / m;  // W(j, i) = R(i, j) for i <= j
      sh->W[i * m + j] = (i <= j) ? At[(size_t)j * 9 + i] : 0.0;
    }
    // V = householderQ (9 x 9): lane c builds column c
    if (lane < 9) {
      double q[9];
      for (int i = 0; i < 9; ++i) q[i] = (i == lane) ? 1.0 : 0.0;
      for (int k = m - 1; k >= 0; --k) {
        if (lane < k) continue;
        const int nr = 9 - k;
        const double tau = sh->hco[k];
        const double* ess = At + (size_t)k * 9 + k + 1;
        if (nr == 1) {
          q[k] *= (1.0 - tau);
        } else if (tau != 0.0) {
          double tmp = 0.0;
          for (int i = 1; i < nr; ++i) tmp += ess[i - 1] * q[k + i];
          tmp += q[k];
          q[k] -= tau * tmp;
          for (int i = 1; i < nr; ++i) q[k + i] -= tau * ess[i - 1] * tmp;
        }
      }
      for (int i = 0; i < 9; ++i) sh->V[lane * 9 + i] = q[i];
    }
    wv_sync();
    return m;
  }
  for (int e = lane; e < 81; e += 64) {
    sh->W[e] = A[e] / scale;
    sh->V[e] = (e % 9 == e / 9) ? 1.0 : 0.0;
  }
  wv_sync();
  return 9;
}

DSM_DEV void wv_svd_V_mx9(double* A, double* At, int m, WvSvdShared* sh, double* sv, int lane) {
  double scale;
  const int dsz = wv_svd_prepare_mx9(A, At, m, sh, &scale, lane);
  LSEC_BEGIN();
  wv_jacobi_sweeps(sh, dsz, scale, sv, lane);
  LSEC_END(5);
}

// The same sweeps (same rotations, same order, same arithmetic as wv_jacobi_sweeps) by a GROUP of G lanes that
// owns one problem in LDS -- 64 / G problems per wave.  The 2 x 2 rotation (five divisions and three square roots
// in a dependent chain, ~200 instructions) is what a sweep costs; it is computed once per wave instruction for
// all groups, so the wave's cost per rotation hardly depends on G while its yield is 64 / G problems.  gl = lane
// within the group; lane gl updates elements gl, gl + G, ... of the two rows / columns.  All lanes of a group read the
// same LDS words and take the same branches; different groups of a wave diverge.  W, V, sv are the group's LDS
// areas, accessed through volatile pointers: the instruction order of the wave is the only synchronisation that
// the lanes of a group need (as in verify_fivept_coop.h).
typedef volatile double* grp_vd;
template <int G>
DSM_DEV void grp_jacobi_sweeps(grp_vd W, grp_vd V, int dsz, double scale, grp_vd sv, int gl) {
  const double precision = 2.0 * DBL_EPSILON;
  double max_diag = 0.0;
  for (int i = 0; i < dsz; ++i) {
    const double a = fabs(W[i * dsz + i]);
    if (a > max_diag) max_diag = a;
  }
  bool finished = false;
  while (!finished) {
    finished = true;
    for (int p = 1; p < dsz; ++p) {
      for (int q = 0; q < p; ++q) {
        const double thr = DBL_MIN > precision * max_diag ? DBL_MIN : precision * max_diag;
        const double wpq = W[q * dsz + p], wqp = W[p * dsz + q];
        if (fabs(wpq) > thr || fabs(wqp) > thr) {
          finished = false;
          double lc, ls, rc, rs;
          dsm_jacobi_2x2(W[p * dsz + p], wpq, wqp, W[q * dsz + q], &lc, &ls, &rc, &rs);
          if (!(lc == 1.0 && ls == 0.0)) {
            for (int e = gl; e < dsz; e += G) {  // rows p, q: elements (p, e), (q, e)
              const double xi = W[e * dsz + p], yi = W[e * dsz + q];
              W[e * dsz + p] = lc * xi + ls * yi;
              W[e * dsz + q] = -ls * xi + lc * yi;
            }
          }
          if (!(rc == 1.0 && -rs == 0.0)) {
            for (int e = gl; e < dsz; e += G) {  // columns p, q of W
              const double xi = W[p * dsz + e], yi = W[q * dsz + e];
              W[p * dsz + e] = rc * xi + (-rs) * yi;
              W[q * dsz + e] = rs * xi + rc * yi;
            }
            for (int e = gl; e < 9; e += G) {  // columns p, q of V (9 rows)
              const double xi = V[p * 9 + e], yi = V[q * 9 + e];
              V[p * 9 + e] = rc * xi + (-rs) * yi;
              V[q * 9 + e] = rs * xi + rc * yi;
            }
          }
          const double app = fabs(W[p * dsz + p]), aqq = fabs(W[q * dsz + q]);
          const double mm = app > aqq ? app : aqq;
          if (mm > max_diag) max_diag = mm;
        }
      }
    }
  }
  if (gl == 0) {
    for (int i = 0; i < dsz; ++i) sv[i] = fabs(W[i * dsz + i]) * scale;
    for (int i = 0; i < dsz; ++i) {
      int pos = i;
      double mx = sv[i];
      for (int j = i + 1; j < dsz; ++j)
        if (sv[j] > mx) {
          mx = sv[j];
          pos = j;
        }
      if (mx == 0.0) break;
      if (pos != i) {
        double t = sv[i];
        sv[i] = sv[pos];
        sv[pos] = t;
        for (int r = 0; r < 9; ++r) {
          t = V[i * 9 + r];
          V[i * 9 + r] = V[pos * 9 + r];
          V[pos * 9 + r] = t;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------- tall matrix in registers, rows over the lanes
// wv_colpiv_qr for a tall m x 9 matrix (9 < m <= 64 * RPL) whose rows live in REGISTERS: lane l owns rows l, l + 64,
// ... (a[c][r] = element (l + 64 r, c)).  wv_colpiv_qr keeps the matrix in the workgroup's global scratch, so every
// one of its ~60 dependent steps per problem is a memory round trip; here a step is register arithmetic plus the
// reduction.  The reductions are wide_sum()'s (oracle/linalg.h): 64 interleaved partial sums over the SUB-vector that
// starts at row r0, combined by the xor butterfly.  Partial b of that sum covers rows r0 + b + 64 j -- all owned by
// lane (r0 + b) mod 64, in the owner's ascending order -- so the owner accumulates it and one lane rotation puts it
// where the butterfly expects it.  Every lane ends with the same norms / pivots / reflector scalars (they are
// computed redundantly from butterfly results), so nothing goes through LDS.
DSM_DEV double wr_tree_finish(double acc, int r0, int lane) {
  double s = __shfl(acc, (lane + r0) & 63);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  return s;
}
template <int RPL>
DSM_DEV void wr_colpiv_qr9(double (&a)[9][RPL], int m, int lane, double (&hco)[9], int (&perm)[9]) {
  double nu[9], nd[9];
  {
    double acc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] = 0.0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      if (64 * r < m) {
        if (lane + 64 * r < m) {
#pragma unroll
          for (int c = 0; c < 9; ++c) acc[c] += a[c][r] * a[c][r];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      nd[c] = sqrt(wr_tree_finish(acc[c], 0, lane));
      nu[c] = nd[c];
      perm[c] = c;
    }
  }
  const double norm_downdate_threshold = sqrt(DBL_EPSILON);
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    int biggest = k;
    double mx = nu[k];
#pragma unroll
    for (int j = k + 1; j < 9; ++j) {
      if (nu[j] > mx) {
        mx = nu[j];
        biggest = j;
      }
    }
#pragma unroll
    for (int j = k + 1; j < 9; ++j) {  // unconditional stores of selected values (see pr_jacobi_sweeps9)
      const bool sw = (j == biggest);
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double x = a[k][r], y = a[j][r];
        a[k][r] = sw ? y : x;
        a[j][r] = sw ? x : y;
      }
      const double ua = nu[k], ub = nu[j];
      nu[k] = sw ? ub : ua;
      nu[j] = sw ? ua : ub;
      const double da = nd[k], db = nd[j];
      nd[k] = sw ? db : da;
      nd[j] = sw ? da : db;
      const int pa = perm[k], pb = perm[j];
      perm[k] = sw ? pb : pa;
      perm[j] = sw ? pa : pb;
    }
    // makeHouseholder on column k, rows k..m-1 (row k = lane k, slot 0)
    const int r0 = k + 1;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int R = lane + 64 * r;
      if (64 * r < m && R >= r0 && R < m) acc += a[k][r] * a[k][r];
    }
    const double tail_sq = wr_tree_finish(acc, r0, lane);
    const double c0 = wv_readlane_f64(a[k][0], k);  // row k = lane k, slot 0
    double tau, beta, den = 0.0;
    bool zero_tail;
    if (tail_sq <= DBL_MIN) {
      tau = 0.0;
      beta = c0;
      zero_tail = true;
    } else {
      double b = sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0.0) b = -b;
      tau = (b - c0) / b;
      beta = b;
      den = c0 - b;
      zero_tail = false;
    }
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int R = lane + 64 * r;
      if (64 * r < m && R >= r0 && R < m) a[k][r] = zero_tail ? 0.0 : a[k][r] / den;
    }
    if (lane == k) a[k][0] = beta;
    hco[k] = tau;
    if (k < 8) {
      if (tau != 0.0) {  // m - k >= 2 rows: never the single-row form
        double t[8];
#pragma unroll
        for (int c = 0; c < 8 - k; ++c) {
          double ac = 0.0;
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            const int R = lane + 64 * r;
            if (64 * r < m && R >= r0 && R < m) ac += a[k][r] * a[k + 1 + c][r];
          }
          t[c] = wr_tree_finish(ac, r0, lane) + wv_readlane_f64(a[k + 1 + c][0], k);
        }
#pragma unroll
        for (int c = 0; c < 8 - k; ++c) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            const int R = lane + 64 * r;
            if (64 * r < m && R < m) {
              if (R == k)
                a[k + 1 + c][r] -= tau * t[c];
              else if (R > k)
                a[k + 1 + c][r] -= tau * a[k][r] * t[c];
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 8 - k; ++c) {  // norm downdating (every lane the same values)
        const int j = k + 1 + c;
        if (nu[j] != 0.0) {
          double temp = fabs(wv_readlane_f64(a[j][0], k)) / nu[j];
          temp = (1.0 + temp) * (1.0 - temp);
          temp = temp < 0.0 ? 0.0 : temp;
          const double ratio = nu[j] / nd[j];
          const double temp2 = temp * (ratio * ratio);
          if (temp2 <= norm_downdate_threshold) {
            double ac = 0.0;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
              const int R = lane + 64 * r;
              if (64 * r < m && R >= r0 && R < m) ac += a[j][r] * a[j][r];
            }
            nd[j] = sqrt(wr_tree_finish(ac, r0, lane));
            nu[j] = nd[j];
          } else {
            nu[j] *= sqrt(temp);
          }
        }
      }
    }
  }
}

// The sweeps of grp_jacobi_sweeps for a full 9 x 9 problem by ONE lane with W and V in registers: the (p, q) order is
// static, so all 36 rotations of a sweep are unrolled with compile-time indices; a lane whose off-diagonal pair is
// already below the threshold skips its rotation by predicate, a lane that has converged leaves the sweep loop.
// The 2 x 2 rotation chain (five divisions, three square roots) is issued once per wave instruction for 64 problems
// (the 8-lane groups: for 8).  W, V column-major 9 x 9; on return V's columns are sorted by descending singular value
// exactly like JacobiSVD does (selection sort with column swaps), sv holds the values.
DSM_DEV void pr_jacobi_sweeps9(double (&W)[81], double (&V)[81], double scale, double (&sv)[9]) {
  const double precision = 2.0 * DBL_EPSILON;
  double max_diag = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double a = fabs(W[i * 9 + i]);
    if (a > max_diag) max_diag = a;
  }
  bool finished = false;
  while (!finished) {
    finished = true;
#pragma unroll
    for (int p = 1; p < 9; ++p) {
#pragma unroll
      for (int q = 0; q < p; ++q) {
        const double thr = DBL_MIN > precision * max_diag ? DBL_MIN : precision * max_diag;
        const double wpq = W[q * 9 + p], wqp = W[p * 9 + q];
        if (fabs(wpq) > thr || fabs(wqp) > thr) {
          finished = false;
          double lc, ls, rc, rs;
          dsm_jacobi_2x2(W[p * 9 + p], wpq, wqp, W[q * 9 + q], &lc, &ls, &rc, &rs);
          if (!(lc == 1.0 && ls == 0.0)) {
#pragma unroll
            for (int e = 0; e < 9; ++e) {  // rows p, q: elements (p, e), (q, e)
              const double xi = W[e * 9 + p], yi = W[e * 9 + q];
              W[e * 9 + p] = lc * xi + ls * yi;
              W[e * 9 + q] = -ls * xi + lc * yi;
            }
          }
          if (!(rc == 1.0 && -rs == 0.0)) {
#pragma unroll
            for (int e = 0; e < 9; ++e) {  // columns p, q of W
              const double xi = W[p * 9 + e], yi = W[q * 9 + e];
              W[p * 9 + e] = rc * xi + (-rs) * yi;
              W[q * 9 + e] = rs * xi + rc * yi;
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) {  // columns p, q of V
              const double xi = V[p * 9 + e], yi = V[q * 9 + e];
              V[p * 9 + e] = rc * xi + (-rs) * yi;
              V[q * 9 + e] = rs * xi + rc * yi;
            }
          }
          const double app = fabs(W[p * 9 + p]), aqq = fabs(W[q * 9 + q]);
          const double mm = app > aqq ? app : aqq;
          if (mm > max_diag) max_diag = mm;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) sv[i] = fabs(W[i * 9 + i]) * scale;
  bool stop = false;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    if (!stop) {
      int pos = i;
      double mx = sv[i];
#pragma unroll
      for (int j = i + 1; j < 9; ++j) {
        if (sv[j] > mx) {
          mx = sv[j];
          pos = j;
        }
      }
      if (mx == 0.0) {
        stop = true;
      } else {
        // predicated swaps written as unconditional stores of selected VALUES: a store under `if (j == pos)` gets
        // merged by the optimiser into one store through a selected ADDRESS, which sends the whole array to scratch
#pragma unroll
        for (int j = i + 1; j < 9; ++j) {
          const bool sw = (j == pos);
          const double si = sv[i], sj = sv[j];
          sv[i] = sw ? sj : si;
          sv[j] = sw ? si : sj;
#pragma unroll
          for (int r = 0; r < 9; ++r) {
            const double a = V[i * 9 + r], b = V[j * 9 + r];
            V[i * 9 + r] = sw ? b : a;
            V[j * 9 + r] = sw ? a : b;
          }
        }
      }
    }
  }
}

#endif  // DAGSFM_AMD_CSRC_VERIFY_LINALG_H_
