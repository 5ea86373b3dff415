// spatial_verification.cc -- see spatial_verification.h.  Follows, function by function:
//   FeatureKeypoint::ComputeScale / ComputeOrientation     /root/reference/src/feature/types.cc:84-98
//   FeatureGeometry::TransformFromMatch / GetArea / GetAreaUnderTransform    src/retrieval/geometry.cc:37-86
//   AffineTransformEstimator::Estimate                      src/estimators/affine_transform.cc:40-75
//   TwoWayTransform, VotingBin, ComputeScaleError, ComputeTransferError, ComputeInliers, ComputeEffectiveInlierCount,
//   VoteAndVerify                                            src/retrieval/vote_and_verify.cc:46-418
//   the 1-to-1 assignment and the re-ranking of VisualIndex::Query    src/retrieval/visual_index.h:366-500
// Where the reference's result depends on pointer values or on Eigen's reduction order, the order is the one
// oracle/spatial_verification.h and oracle/retrieval.cc define; where it depends on std::unordered_map's iteration order
// (the voting bins) the same container is fed the same insertions;
// tests/test_retrieval.py compares the two implementations bit for bit.  Everything here is float / double arithmetic in
// the reference's own expression order (-ffp-contract=off).
#include "spatial_verification.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <limits>
#include <map>
#include <set>
#include <unordered_map>
#include <utility>

namespace dagsfm_amd {

FeatureGeometry GeometryOfKeypoint(const FeatureKeypoint& k) {
  FeatureGeometry g;
  g.x = k.x;
  g.y = k.y;
  const float scale_x = std::sqrt(k.a11 * k.a11 + k.a21 * k.a21);
  const float scale_y = std::sqrt(k.a12 * k.a12 + k.a22 * k.a22);
  g.scale = (scale_x + scale_y) / 2.0f;
  g.orientation = std::atan2(k.a21, k.a11);
  return g;
}

namespace {

// ---------------------------------------------------------------------------------------- dense least squares
// x = argmin |C x - b| the way Eigen's `C.jacobiSvd(ComputeThinU | ComputeThinV).solve(b)` computes it for a tall
// rows x 6 matrix: column-pivoted Householder QR of C / max|C|, two-sided Jacobi sweeps on the 6 x 6 triangle, singular
// values sorted, x = V S^-1 U^T b over the numerical rank.  Sums over more than nine terms run in the 64-partial order the
// oracle and the device use for tall systems, shorter ones left to right.
const int kCols = 6;

struct TallMatrix {  // column-major rows x kCols
  int rows;
  std::vector<double> a;
  explicit TallMatrix(int r) : rows(r), a(static_cast<size_t>(r) * kCols, 0.0) {}
  double& at(int r, int c) { return a[static_cast<size_t>(c) * rows + r]; }
  double at(int r, int c) const { return a[static_cast<size_t>(c) * rows + r]; }
};

template <typename F>
double SumOf(int n, bool wide, F term) {
  if (!wide) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += term(i);
    return s;
  }
  double part[64];
  for (double& p : part) p = 0.0;
  for (int i = 0; i < n; ++i) part[i & 63] += term(i);
  for (int o = 32; o > 0; o >>= 1)
    for (int l = 0; l < o; ++l) part[l] += part[l + o];
  return part[0];
}

// Householder reflector of x[0 .. n): essential part left in x[1 .. n)
void MakeReflector(double* x, int n, bool wide, double* tau, double* beta) {
  const double tail = wide ? SumOf(n - 1, true, [x](int i) { return x[i + 1] * x[i + 1]; })
                           : SumOf(n - 1, false, [x](int i) { return x[i + 1] * x[i + 1]; });
  const double head = x[0];
  if (tail <= DBL_MIN) {
    *tau = 0.0;
    *beta = head;
    for (int i = 1; i < n; ++i) x[i] = 0.0;
    return;
  }
  double b = std::sqrt(head * head + tail);
  if (head >= 0.0) b = -b;
  for (int i = 1; i < n; ++i) x[i] = x[i] / (head - b);
  *tau = (b - head) / b;
  *beta = b;
}

// (I - tau v v^T) applied from the left to column `col` of M, rows r0 .. r0 + nr; v = (1, ess)
void ReflectColumn(TallMatrix& M, int r0, int nr, int col, const double* ess, double tau, bool wide) {
  if (nr == 1) {
    M.at(r0, col) *= (1.0 - tau);
    return;
  }
  if (tau == 0.0) return;
  const TallMatrix& C = M;
  double t = SumOf(nr - 1, wide, [&](int i) { return ess[i] * C.at(r0 + i + 1, col); });
  t += M.at(r0, col);
  M.at(r0, col) -= tau * t;
  for (int i = 1; i < nr; ++i) M.at(r0 + i, col) -= tau * ess[i - 1] * t;
}

struct Rotation {
  double c, s;
};
void Rotate(double* x, int incx, double* y, int incy, int n, Rotation r) {  // x' = c x + s y, y' = -s x + c y
  if (r.c == 1.0 && r.s == 0.0) return;
  for (int i = 0; i < n; ++i) {
    const double xi = x[static_cast<size_t>(i) * incx], yi = y[static_cast<size_t>(i) * incy];
    x[static_cast<size_t>(i) * incx] = r.c * xi + r.s * yi;
    y[static_cast<size_t>(i) * incy] = -r.s * xi + r.c * yi;
  }
}

void SolveLeastSquares(const TallMatrix& Cin, const std::vector<double>& b, double x[kCols]) {
  const int rows = Cin.rows;
  const bool wide = rows > 9;
  double scale = 0.0;
  for (double v : Cin.a) scale = std::max(scale, std::fabs(v));
  if (scale == 0.0) scale = 1.0;
  TallMatrix U(rows);
  double W[kCols * kCols], V[kCols * kCols];  // column-major
  for (double& v : W) v = 0.0;
  for (double& v : V) v = 0.0;
  if (rows == kCols) {
    // three points: the system is square and JacobiSVD takes it as it is (no QR preconditioner)
    for (int c = 0; c < kCols; ++c) {
      for (int r = 0; r < kCols; ++r) W[c * kCols + r] = Cin.at(r, c) / scale;
      U.at(c, c) = 1.0;
      V[c * kCols + c] = 1.0;
    }
  } else {
    TallMatrix qr(rows);
    for (size_t i = 0; i < Cin.a.size(); ++i) qr.a[i] = Cin.a[i] / scale;
    // column-pivoted Householder QR (rows >= kCols)
    double tau[kCols], norm_now[kCols], norm_ref[kCols];
    int swapped_with[kCols];
    for (int c = 0; c < kCols; ++c) {
      const TallMatrix& Q = qr;
      norm_ref[c] = std::sqrt(SumOf(rows, wide, [&](int i) { return Q.at(i, c) * Q.at(i, c); }));
      norm_now[c] = norm_ref[c];
    }
    const double downdate_threshold = std::sqrt(DBL_EPSILON);
    for (int k = 0; k < kCols; ++k) {
      int pivot = k;
      for (int c = k + 1; c < kCols; ++c)
        if (norm_now[c] > norm_now[pivot]) pivot = c;
      swapped_with[k] = pivot;
      if (pivot != k) {
        for (int i = 0; i < rows; ++i) std::swap(qr.at(i, k), qr.at(i, pivot));
        std::swap(norm_now[k], norm_now[pivot]);
        std::swap(norm_ref[k], norm_ref[pivot]);
      }
      double beta;
      MakeReflector(&qr.at(k, k), rows - k, wide, &tau[k], &beta);
      qr.at(k, k) = beta;
      const double* ess = &qr.a[static_cast<size_t>(k) * rows + k + 1];
      for (int c = k + 1; c < kCols; ++c) ReflectColumn(qr, k, rows - k, c, ess, tau[k], wide);
      for (int c = k + 1; c < kCols; ++c) {
        if (norm_now[c] == 0.0) continue;
        double t = std::fabs(qr.at(k, c)) / norm_now[c];
        t = (1.0 + t) * (1.0 - t);
        if (t < 0.0) t = 0.0;
        const double ratio = norm_now[c] / norm_ref[c];
        if (t * (ratio * ratio) <= downdate_threshold) {
          const TallMatrix& Q = qr;
          norm_ref[c] = std::sqrt(SumOf(rows - k - 1, wide, [&](int i) { return Q.at(k + 1 + i, c) * Q.at(k + 1 + i, c); }));
          norm_now[c] = norm_ref[c];
        } else {
          norm_now[c] *= std::sqrt(t);
        }
      }
    }
    int perm[kCols];
    for (int c = 0; c < kCols; ++c) perm[c] = c;
    for (int k = 0; k < kCols; ++k) std::swap(perm[k], perm[swapped_with[k]]);
    // thin Q: the reflectors, last first, applied to the first kCols columns of the identity (left to right in a sum of
    // fewer than ten terms, like householderQ().evalTo())
    for (int c = 0; c < kCols; ++c) U.at(c, c) = 1.0;
    for (int k = kCols - 1; k >= 0; --k) {
      const double* ess = &qr.a[static_cast<size_t>(k) * rows + k + 1];
      for (int c = k; c < kCols; ++c) ReflectColumn(U, k, rows - k, c, ess, tau[k], false);
    }
    // W = R (upper triangle), V = the column permutation
    for (int c = 0; c < kCols; ++c) {
      for (int r = 0; r <= c; ++r) W[c * kCols + r] = qr.at(r, c);
      V[c * kCols + perm[c]] = 1.0;
    }
  }
  // two-sided Jacobi sweeps (JacobiSVD::compute)
  const double precision = 2.0 * DBL_EPSILON;
  double max_diag = 0.0;
  for (int i = 0; i < kCols; ++i) max_diag = std::max(max_diag, std::fabs(W[i * kCols + i]));
  for (bool done = false; !done;) {
    done = true;
    for (int p = 1; p < kCols; ++p) {
      for (int q = 0; q < p; ++q) {
        const double threshold = std::max(DBL_MIN, precision * max_diag);
        if (!(std::fabs(W[q * kCols + p]) > threshold || std::fabs(W[p * kCols + q]) > threshold)) continue;
        done = false;
        double m00 = W[p * kCols + p], m01 = W[q * kCols + p], m10 = W[p * kCols + q], m11 = W[q * kCols + q];
        Rotation first;
        const double t = m00 + m11, d = m10 - m01;
        if (std::fabs(d) < DBL_MIN) {
          first.s = 0.0;
          first.c = 1.0;
        } else {
          const double u = t / d;
          const double h = std::sqrt(1.0 + u * u);
          first.s = 1.0 / h;
          first.c = u / h;
        }
        {
          const double a0 = first.c * m00 + first.s * m10, a1 = first.c * m01 + first.s * m11;
          const double b0 = -first.s * m00 + first.c * m10, b1 = -first.s * m01 + first.c * m11;
          m00 = a0;
          m01 = a1;
          m10 = b0;
          m11 = b1;
        }
        Rotation right;  // JacobiRotation::makeJacobi(m00, m01, m11)
        {
          const double deno = 2.0 * std::fabs(m01);
          if (deno < DBL_MIN) {
            right.c = 1.0;
            right.s = 0.0;
          } else {
            const double ta = (m00 - m11) / deno;
            const double w = std::sqrt(ta * ta + 1.0);
            const double tt = ta > 0.0 ? 1.0 / (ta + w) : 1.0 / (ta - w);
            const double sign_t = tt > 0.0 ? 1.0 : -1.0;
            const double nn = 1.0 / std::sqrt(tt * tt + 1.0);
            right.s = -sign_t * (m01 / std::fabs(m01)) * std::fabs(tt) * nn;
            right.c = nn;
          }
        }
        Rotation left;  // first * right^T
        left.c = first.c * right.c - first.s * (-right.s);
        left.s = first.c * (-right.s) + first.s * right.c;
        Rotate(&W[p], kCols, &W[q], kCols, kCols, left);                                    // rows p, q of W
        Rotate(&U.a[static_cast<size_t>(p) * rows], 1, &U.a[static_cast<size_t>(q) * rows], 1, rows, left);  // columns p, q of U
        const Rotation rt = {right.c, -right.s};
        Rotate(&W[p * kCols], 1, &W[q * kCols], 1, kCols, rt);                              // columns p, q of W
        Rotate(&V[p * kCols], 1, &V[q * kCols], 1, kCols, rt);
        max_diag = std::max(max_diag, std::max(std::fabs(W[p * kCols + p]), std::fabs(W[q * kCols + q])));
      }
    }
  }
  double sv[kCols];
  for (int i = 0; i < kCols; ++i) {
    const double a = W[i * kCols + i];
    sv[i] = std::fabs(a);
    if (a < 0.0)
      for (int r = 0; r < rows; ++r) U.at(r, i) = -U.at(r, i);
  }
  for (int i = 0; i < kCols; ++i) sv[i] *= scale;
  for (int i = 0; i < kCols; ++i) {  // descending, by selection
    int pos = i;
    for (int j = i + 1; j < kCols; ++j)
      if (sv[j] > sv[pos]) pos = j;
    if (sv[pos] == 0.0) break;
    if (pos != i) {
      std::swap(sv[i], sv[pos]);
      for (int r = 0; r < rows; ++r) std::swap(U.at(r, i), U.at(r, pos));
      for (int r = 0; r < kCols; ++r) std::swap(V[i * kCols + r], V[pos * kCols + r]);
    }
  }
  // SVDBase::rank() and _solve_impl
  int nonzero = 0;
  for (int i = 0; i < kCols; ++i) nonzero += sv[i] != 0.0;
  const double cut = std::max(sv[0] * (kCols * DBL_EPSILON), DBL_MIN);
  int rank = nonzero;
  while (rank > 0 && sv[rank - 1] < cut) --rank;
  double y[kCols] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < rank; ++k) {
    double s = 0.0;
    for (int r = 0; r < rows; ++r) s += U.at(r, k) * b[r];
    y[k] = (1.0 / sv[k]) * s;
  }
  for (int j = 0; j < kCols; ++j) {
    double s = 0.0;
    for (int k = 0; k < rank; ++k) s += V[k * kCols + j] * y[k];
    x[j] = s;
  }
}

// ---------------------------------------------------------------------------------------- vote and verify
int TruncateToInt(float v) {  // static_cast<int>(float) as x86-64 executes it: INT_MIN for NaN and out-of-range values
  if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT_MIN;
  return static_cast<int>(v);
}

struct Similarity {  // FeatureGeometryTransform
  float scale = 0.0f, angle = 0.0f, tx = 0.0f, ty = 0.0f;
};

Similarity SimilarityOfMatch(const FeatureGeometry& a, const FeatureGeometry& b) {
  Similarity t;
  t.scale = b.scale / a.scale;
  t.angle = b.orientation - a.orientation;
  const float sn = std::sin(t.angle), cs = std::cos(t.angle);
  const float m00 = t.scale * cs, m01 = t.scale * -sn, m10 = t.scale * sn, m11 = t.scale * cs;
  t.tx = b.x - (m00 * a.x + m01 * a.y);
  t.ty = b.y - (m10 * a.x + m11 * a.y);
  return t;
}

struct BothWays {  // TwoWayTransform: forward (query -> database) and backward maps, 2 x 2 row-major + translation
  float fwd[4] = {0, 0, 0, 0}, fwd_t[2] = {0, 0}, bwd[4] = {0, 0, 0, 0}, bwd_t[2] = {0, 0};
};

BothWays BothWaysOf(const Similarity& t) {
  BothWays w;
  const float sn = std::sin(t.angle), cs = std::cos(t.angle);
  w.fwd[0] = t.scale * cs;
  w.fwd[1] = t.scale * -sn;
  w.fwd[2] = t.scale * sn;
  w.fwd[3] = t.scale * cs;
  w.fwd_t[0] = t.tx;
  w.fwd_t[1] = t.ty;
  w.bwd[0] = cs / t.scale;
  w.bwd[1] = sn / t.scale;
  w.bwd[2] = -sn / t.scale;
  w.bwd[3] = cs / t.scale;
  w.bwd_t[0] = (-w.bwd[0]) * w.fwd_t[0] + (-w.bwd[1]) * w.fwd_t[1];
  w.bwd_t[1] = (-w.bwd[2]) * w.fwd_t[0] + (-w.bwd[3]) * w.fwd_t[1];
  return w;
}

float AreaOf(const FeatureGeometry& g) { return 1.0f / std::sqrt(4.0f / (g.scale * g.scale * g.scale * g.scale)); }

// What IsInlier needs of a match that does not depend on the hypothesis, computed once per match: the database feature's
// own area and the two distinct entries of Identity / scale^2 (1 / s^2 and 0 / s^2 -- the latter is 0 unless the scale is
// 0, inf or NaN, and is kept as the expression it is in the reference)
struct MatchConstants {
  float measured, d, z;
};

float AreaUnder(float d, float z, const float A[4]) {  // N = A^T (I / scale^2) A, 1 / sqrt(4 N00 N11 - (N10 + N01)^2)
  const float t00 = A[0] * d + A[2] * z, t01 = A[0] * z + A[2] * d;
  const float t10 = A[1] * d + A[3] * z, t11 = A[1] * z + A[3] * d;
  const float n00 = t00 * A[0] + t01 * A[2], n01 = t00 * A[1] + t01 * A[3];
  const float n10 = t10 * A[0] + t11 * A[2], n11 = t10 * A[1] + t11 * A[3];
  const float bsum = n10 + n01;
  return 1.0f / std::sqrt(4.0f * n00 * n11 - bsum * bsum);
}

bool IsInlier(const GeometryMatch& m, const MatchConstants& k, const BothWays& w, float max_transfer_error, float max_scale_error) {
  const float measured = k.measured;
  const float moved = AreaUnder(k.d, k.z, w.bwd);
  const float scale_error = moved > measured ? moved / measured : measured / moved;
  if (!(scale_error <= max_scale_error)) return false;
  const float ax = (m.database.x - (w.fwd[0] * m.query.x + w.fwd[1] * m.query.y)) - w.fwd_t[0];
  const float ay = (m.database.y - (w.fwd[2] * m.query.x + w.fwd[3] * m.query.y)) - w.fwd_t[1];
  const float bx = (m.query.x - (w.bwd[0] * m.database.x + w.bwd[1] * m.database.y)) - w.bwd_t[0];
  const float by = (m.query.y - (w.bwd[2] * m.database.x + w.bwd[3] * m.database.y)) - w.bwd_t[1];
  const float e1 = ax * ax + ay * ay, e2 = bx * bx + by * by;
  return (e1 + e2) <= max_transfer_error;
}

std::vector<int> InliersOf(const std::vector<GeometryMatch>& matches, const std::vector<MatchConstants>& measured, const BothWays& w,
                           float max_transfer_error, float max_scale_error) {
  std::vector<int> idx;
  for (size_t i = 0; i < matches.size(); ++i)
    if (IsInlier(matches[i], measured[i], w, max_transfer_error, max_scale_error)) idx.push_back(static_cast<int>(i));
  return idx;
}

size_t NumTrialsFor(size_t num_inliers, size_t num_samples, double confidence) {  // RANSAC::ComputeNumTrials, 3-point samples
  const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
  const double nom = 1 - confidence;
  if (nom <= 0) return std::numeric_limits<size_t>::max();
  const double denom = 1 - std::pow(inlier_ratio, 3);
  if (denom <= 0) return 1;
  return static_cast<size_t>(std::ceil(std::log(nom) / std::log(denom)));
}

struct Votes {  // VotingBin
  size_t count = 0;
  Similarity sum;
};

}  // namespace

void EstimateAffineTransform(const double* src, const double* dst, size_t n, double A[6]) {
  TallMatrix C(static_cast<int>(2 * n));
  std::vector<double> b(2 * n);
  for (size_t i = 0; i < n; ++i) {
    const int r = static_cast<int>(2 * i);
    C.at(r, 0) = src[2 * i];
    C.at(r, 1) = src[2 * i + 1];
    C.at(r, 2) = 1.0f;
    b[r] = dst[2 * i];
    C.at(r + 1, 3) = src[2 * i];
    C.at(r + 1, 4) = src[2 * i + 1];
    C.at(r + 1, 5) = 1.0f;
    b[r + 1] = dst[2 * i + 1];
  }
  SolveLeastSquares(C, b, A);
}

int VoteAndVerify(const VoteAndVerifyOptions& o, const std::vector<GeometryMatch>& matches) {
  if (matches.size() < 3) return 0;
  const float max_trans = o.max_image_size;
  const float max_log_scale = std::log2(10.0f);
  const float trans_norm = 1.0f / (2.0f * max_trans);
  const float scale_norm = 1.0f / (2.0f * max_log_scale);
  const float angle_norm = 1.0f / (2.0f * M_PI);
  const int kLevels = 6;
  // n_a + num_angle_bins * (n_s + num_scale_bins * (n_x + num_trans_bins * n_y)) as the reference's int expression, with
  // the two's-complement wrap x86 gives it when a coordinate is INT_MIN (a NaN transformation passes the range tests)
  auto key_of = [&](int na, int ns, int nx, int ny) {
    const uint32_t k = static_cast<uint32_t>(na) + static_cast<uint32_t>(o.num_angle_bins) *
        (static_cast<uint32_t>(ns) + static_cast<uint32_t>(o.num_scale_bins) * (static_cast<uint32_t>(nx) + static_cast<uint32_t>(o.num_trans_bins) * static_cast<uint32_t>(ny)));
    return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(k)));
  };
  // The reference keeps one hash map of bins per level.  Here: the votes in match order, and per level the indices of the
  // votes stably sorted by the level's key -- a run of equal keys is a bin, and inside a run the votes are still in match
  // order, which is the order the reference adds them up in (float sums).
  struct Cell {
    int na, ns, nx, ny;
  };
  struct Vote {
    uint64_t key[kLevels];
    Cell cell;
    Similarity T;
  };
  std::vector<Vote> votes;
  votes.reserve(matches.size());
  for (const GeometryMatch& m : matches) {
    const Similarity T = SimilarityOfMatch(m.query, m.database);
    if (std::abs(T.tx) > max_trans || std::abs(T.ty) > max_trans) continue;
    const float log_scale = std::log2(T.scale);
    if (std::abs(log_scale) > max_log_scale) continue;
    const float x = (T.tx + max_trans) * trans_norm;
    const float y = (T.ty + max_trans) * trans_norm;
    const float s = (log_scale + max_log_scale) * scale_norm;
    const float a = (T.angle + M_PI) * angle_norm;
    Vote v;
    v.cell.nx = std::min(TruncateToInt(x * o.num_trans_bins), o.num_trans_bins - 1);
    v.cell.ny = std::min(TruncateToInt(y * o.num_trans_bins), o.num_trans_bins - 1);
    v.cell.ns = std::min(TruncateToInt(s * o.num_scale_bins), o.num_scale_bins - 1);
    v.cell.na = std::min(TruncateToInt(a * o.num_angle_bins), o.num_angle_bins - 1);
    for (int l = 0; l < kLevels; ++l) v.key[l] = key_of(v.cell.na >> l, v.cell.ns >> l, v.cell.nx >> l, v.cell.ny >> l);
    v.T = T;
    votes.push_back(v);
  }
  std::vector<uint64_t> bin_key[kLevels];
  std::vector<uint32_t> bin_count[kLevels];
  std::vector<Votes> finest_bin;   // level 0: count and sums
  std::vector<Cell> finest_cell;   // the cell of the LAST vote of the bin (coords[index] is overwritten by every vote)
  std::vector<uint32_t> by_key(votes.size());
  for (int l = 0; l < kLevels; ++l) {
    for (uint32_t i = 0; i < by_key.size(); ++i) by_key[i] = i;
    std::stable_sort(by_key.begin(), by_key.end(), [&](uint32_t i, uint32_t j) { return votes[i].key[l] < votes[j].key[l]; });
    for (size_t i = 0; i < by_key.size();) {
      size_t j = i;
      Votes bin;
      while (j < by_key.size() && votes[by_key[j]].key[l] == votes[by_key[i]].key[l]) {
        if (l == 0) {
          const Similarity& T = votes[by_key[j]].T;
          bin.sum.scale += T.scale;
          bin.sum.angle += T.angle;
          bin.sum.tx += T.tx;
          bin.sum.ty += T.ty;
        }
        ++j;
      }
      bin.count = j - i;
      bin_key[l].push_back(votes[by_key[i]].key[l]);
      bin_count[l].push_back(static_cast<uint32_t>(j - i));
      if (l == 0) {
        finest_bin.push_back(bin);
        finest_cell.push_back(votes[by_key[j - 1]].cell);
      }
      i = j;
    }
  }
  auto count_at = [&](int l, uint64_t key) -> size_t {
    const auto it = std::lower_bound(bin_key[l].begin(), bin_key[l].end(), key);
    return it != bin_key[l].end() && *it == key ? bin_count[l][it - bin_key[l].begin()] : 0;
  };
  // multi-resolution score of every occupied finest cell.  The reference walks its std::unordered_map of bins and
  // std::partial_sort's what it finds by score, so which of several equally scored bins are among the 30 candidates, and in
  // which order, is whatever libstdc++'s hash table and heap-select make of the sequence of insertions.  The same
  // containers, fed the same sequence (the finest key of every vote, in match order), give the same order here.
  struct Scored {
    uint32_t bin;  // index into finest_bin
    float score;
  };
  std::unordered_map<size_t, uint32_t> walk_order;
  for (const Vote& v : votes) walk_order[static_cast<size_t>(v.key[0])] += 1;
  std::vector<Scored> scored;
  for (const auto& kv : walk_order) {
    const uint32_t b = static_cast<uint32_t>(std::lower_bound(bin_key[0].begin(), bin_key[0].end(), static_cast<uint64_t>(kv.first)) - bin_key[0].begin());
    if (finest_bin[b].count < static_cast<size_t>(o.min_num_votes)) continue;
    const Cell c = finest_cell[b];
    float score = finest_bin[b].count;
    float weight = 0.5f;
    for (int l = 1; l < kLevels; ++l) {
      score += count_at(l, key_of(c.na >> l, c.ns >> l, c.nx >> l, c.ny >> l)) * weight;
      weight *= 0.5f;
    }
    scored.push_back(Scored{b, score});
  }
  const size_t num_candidates = std::min(static_cast<size_t>(o.num_transformations), scored.size());
  std::partial_sort(scored.begin(), scored.begin() + num_candidates, scored.end(), [](const Scored& a, const Scored& b) { return a.score > b.score; });

  const float max_transfer_error = o.max_transfer_error, max_scale_error = o.max_scale_error;
  std::vector<MatchConstants> measured(matches.size());
  for (size_t i = 0; i < matches.size(); ++i) {
    const float s2 = matches[i].query.scale * matches[i].query.scale;
    measured[i] = MatchConstants{AreaOf(matches[i].database), 1.0f / s2, 0.0f / s2};
  }
  size_t max_num_trials = std::numeric_limits<size_t>::max();
  size_t best_count = 0;
  BothWays best;
  for (size_t i = 0; i < num_candidates && i < max_num_trials; ++i) {
    const Votes& v = finest_bin[scored[i].bin];
    const float inv = 1.0f / static_cast<float>(v.count);
    Similarity mean = v.sum;
    mean.scale *= inv;
    mean.angle *= inv;
    mean.tx *= inv;
    mean.ty *= inv;
    const BothWays w = BothWaysOf(mean);
    std::vector<int> inl = InliersOf(matches, measured, w, max_transfer_error, max_scale_error);
    if (inl.size() < best_count || inl.size() < 3) continue;
    best_count = inl.size();
    best = w;
    if (best_count == matches.size()) break;
    // local optimisation: the least-squares affine map of the inliers and its inverse
    std::vector<double> p1(2 * inl.size()), p2(2 * inl.size());
    for (size_t j = 0; j < inl.size(); ++j) {
      const GeometryMatch& m = matches[inl[j]];
      p1[2 * j] = m.query.x;
      p1[2 * j + 1] = m.query.y;
      p2[2 * j] = m.database.x;
      p2[2 * j + 1] = m.database.y;
    }
    double A[6];
    EstimateAffineTransform(p1.data(), p2.data(), inl.size(), A);
    // inverse of [A; 0 0 1] by cofactors with one reciprocal of the determinant (Eigen's 3 x 3 inverse)
    const double M[3][3] = {{A[0], A[1], A[2]}, {A[3], A[4], A[5]}, {0.0, 0.0, 1.0}};
    auto cof = [&](int r, int c) {
      const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
      return M[r1][c1] * M[r2][c2] - M[r1][c2] * M[r2][c1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double invdet = 1.0 / (c00 * M[0][0] + c10 * M[1][0] + c20 * M[2][0]);
    BothWays local;
    local.fwd[0] = static_cast<float>(A[0]);
    local.fwd[1] = static_cast<float>(A[1]);
    local.fwd[2] = static_cast<float>(A[3]);
    local.fwd[3] = static_cast<float>(A[4]);
    local.fwd_t[0] = static_cast<float>(A[2]);
    local.fwd_t[1] = static_cast<float>(A[5]);
    local.bwd[0] = static_cast<float>(c00 * invdet);
    local.bwd[1] = static_cast<float>(c10 * invdet);
    local.bwd_t[0] = static_cast<float>(c20 * invdet);
    local.bwd[2] = static_cast<float>(cof(0, 1) * invdet);
    local.bwd[3] = static_cast<float>(cof(1, 1) * invdet);
    local.bwd_t[1] = static_cast<float>(cof(2, 1) * invdet);
    inl = InliersOf(matches, measured, local, max_transfer_error, max_scale_error);
    if (inl.size() > best_count) {
      best_count = inl.size();
      best = local;
      if (best_count == matches.size()) break;
    }
    max_num_trials = NumTrialsFor(best_count, matches.size(), o.confidence);
  }
  if (best_count == 0) return 0;
  // ComputeEffectiveInlierCount: occupied cells of a 64 x 64 grid over the bounding box of the inliers' query positions
  const int kGrid = 64;
  std::vector<std::pair<float, float>> at;
  float min_x = std::numeric_limits<float>::max(), min_y = std::numeric_limits<float>::max(), max_x = 0, max_y = 0;
  for (size_t i = 0; i < matches.size(); ++i) {
    const GeometryMatch& m = matches[i];
    if (!IsInlier(m, measured[i], best, max_transfer_error, max_scale_error)) continue;
    at.emplace_back(m.query.x, m.query.y);
    min_x = std::min(min_x, m.query.x);
    min_y = std::min(min_y, m.query.y);
    max_x = std::max(max_x, m.query.x);
    max_y = std::max(max_y, m.query.y);
  }
  if (at.empty()) return 0;
  const float sx = kGrid / (max_x - min_x), sy = kGrid / (max_y - min_y);
  std::vector<unsigned char> seen(kGrid * kGrid, 0);
  int occupied = 0;
  for (const auto& p : at) {
    const int cx = std::max(0, std::min(kGrid - 1, TruncateToInt((p.first - min_x) * sx)));
    const int cy = std::max(0, std::min(kGrid - 1, TruncateToInt((p.second - min_y) * sy)));
    if (!seen[cx * kGrid + cy]) {
      seen[cx * kGrid + cy] = 1;
      ++occupied;
    }
  }
  return occupied;
}

namespace {

// The free features of one side during the 1-to-1 assignment: the reference keeps them in a heap ordered by (-open
// candidates, feature index) and always takes the top -- the feature with the FEWEST open candidates, the LARGER index
// among equals.  Here: one bitmap of local ids per open-candidate count (local ids ascend with the feature index, so the
// highest set bit of the first non-empty bitmap is the top), which makes "one candidate fewer" a bit moved to the next
// bitmap instead of an erase + insert in an ordered set.
class FreeFeatures {
 public:
  FreeFeatures(const std::vector<uint32_t>& open_counts) : open_(open_counts.begin(), open_counts.end()), words_((open_counts.size() + 63) / 64) {
    uint32_t most = 0;
    for (uint32_t c : open_counts) most = std::max(most, c);
    bits_.assign(static_cast<size_t>(most + 1) * words_, 0);
    for (uint32_t id = 0; id < open_counts.size(); ++id) Set(open_counts[id], id);
    free_ = open_counts.size();
  }
  bool Empty() const { return free_ == 0; }
  bool IsFree(uint32_t id) const { return open_[id] >= 0; }
  // open candidates of the top feature; Top() must not be called when Empty()
  uint32_t TopOpen() {
    while (!AnyAt(lowest_)) ++lowest_;
    return lowest_;
  }
  uint32_t Top() {
    const uint32_t c = TopOpen();
    for (size_t w = words_; w-- > 0;) {
      const uint64_t x = bits_[c * words_ + w];
      if (x) return static_cast<uint32_t>(w * 64 + 63 - __builtin_clzll(x));
    }
    return 0;
  }
  void Take(uint32_t id) {  // leaves the free set for good
    Clear(static_cast<uint32_t>(open_[id]), id);
    open_[id] = -1;
    --free_;
  }
  void OneCandidateFewer(uint32_t id) {
    const uint32_t c = static_cast<uint32_t>(open_[id]);
    Clear(c, id);
    Set(c - 1, id);
    open_[id] = static_cast<int>(c - 1);
    if (c - 1 < lowest_) lowest_ = c - 1;
  }

 private:
  bool AnyAt(uint32_t c) const {
    for (size_t w = 0; w < words_; ++w)
      if (bits_[c * words_ + w]) return true;
    return false;
  }
  void Set(uint32_t c, uint32_t id) { bits_[c * words_ + id / 64] |= 1ull << (id % 64); }
  void Clear(uint32_t c, uint32_t id) { bits_[c * words_ + id / 64] &= ~(1ull << (id % 64)); }
  std::vector<int> open_;  // open candidates of a free feature, -1 once taken
  size_t words_;
  std::vector<uint64_t> bits_;
  uint32_t lowest_ = 0;
  size_t free_ = 0;
};

}  // namespace

uint32_t SpatialRerank(const std::vector<FeatureGeometry>& query_geometries, const std::vector<RetrievalCandidate>& candidates,
                       int num_images_after_verification, uint32_t count, uint32_t* image_idx, float* scores) {
  if (num_images_after_verification <= 0) return count;
  // candidates grouped by retrieved image
  std::vector<uint32_t> by_image(candidates.size());
  for (uint32_t i = 0; i < by_image.size(); ++i) by_image[i] = i;
  std::stable_sort(by_image.begin(), by_image.end(), [&](uint32_t a, uint32_t b) { return candidates[a].image < candidates[b].image; });
  auto stronger = [&](uint32_t a, uint32_t b) {  // the order of a feature's candidate list (spatial_verification.h)
    const RetrievalCandidate &x = candidates[a], &y = candidates[b];
    if (x.weight != y.weight) return x.weight > y.weight;
    if (x.query_feature != y.query_feature) return x.query_feature > y.query_feature;
    return x.entry_position > y.entry_position;
  };
  std::vector<uint32_t> qfeat, dfeat, qlist, dlist, qstart, dstart, qid, did;
  for (uint32_t k = 0; k < count; ++k) {
    const auto lo = std::lower_bound(by_image.begin(), by_image.end(), image_idx[k],
                                     [&](uint32_t a, uint32_t image) { return candidates[a].image < image; });
    auto hi = lo;
    while (hi != by_image.end() && candidates[*hi].image == image_idx[k]) ++hi;
    if (lo == hi) continue;
    const uint32_t nc = static_cast<uint32_t>(hi - lo);
    // local ids of the features on either side, ascending with the feature index
    qfeat.clear();
    dfeat.clear();
    for (auto it = lo; it != hi; ++it) {
      qfeat.push_back(candidates[*it].query_feature);
      dfeat.push_back(candidates[*it].database_feature);
    }
    std::sort(qfeat.begin(), qfeat.end());
    qfeat.erase(std::unique(qfeat.begin(), qfeat.end()), qfeat.end());
    std::sort(dfeat.begin(), dfeat.end());
    dfeat.erase(std::unique(dfeat.begin(), dfeat.end()), dfeat.end());
    qid.resize(nc);
    did.resize(nc);
    for (uint32_t c = 0; c < nc; ++c) {
      qid[c] = static_cast<uint32_t>(std::lower_bound(qfeat.begin(), qfeat.end(), candidates[lo[c]].query_feature) - qfeat.begin());
      did[c] = static_cast<uint32_t>(std::lower_bound(dfeat.begin(), dfeat.end(), candidates[lo[c]].database_feature) - dfeat.begin());
    }
    // each feature's candidates (positions c in [0, nc)), strongest first, as CSR lists
    auto build_lists = [&](const std::vector<uint32_t>& id, size_t n_ids, std::vector<uint32_t>* list, std::vector<uint32_t>* start) {
      list->resize(nc);
      for (uint32_t c = 0; c < nc; ++c) (*list)[c] = c;
      std::sort(list->begin(), list->end(), [&](uint32_t a, uint32_t b) {
        if (id[a] != id[b]) return id[a] < id[b];
        return stronger(lo[a], lo[b]);
      });
      start->assign(n_ids + 1, 0);
      for (uint32_t c = 0; c < nc; ++c) (*start)[id[c] + 1] += 1;
      for (size_t i = 0; i < n_ids; ++i) (*start)[i + 1] += (*start)[i];
    };
    build_lists(qid, qfeat.size(), &qlist, &qstart);
    build_lists(did, dfeat.size(), &dlist, &dstart);
    std::vector<uint32_t> qopen(qfeat.size()), dopen(dfeat.size());
    for (size_t i = 0; i < qfeat.size(); ++i) qopen[i] = qstart[i + 1] - qstart[i];
    for (size_t i = 0; i < dfeat.size(); ++i) dopen[i] = dstart[i + 1] - dstart[i];
    FreeFeatures free_query(qopen), free_database(dopen);
    std::vector<GeometryMatch> matches;
    while (!free_query.Empty() && !free_database.Empty()) {
      // (-open, index) of the query top >= that of the database top  <=>  not more open candidates
      const bool from_query = free_query.TopOpen() <= free_database.TopOpen();
      FreeFeatures& mine = from_query ? free_query : free_database;
      FreeFeatures& theirs = from_query ? free_database : free_query;
      const std::vector<uint32_t>& my_list = from_query ? qlist : dlist;
      const std::vector<uint32_t>& my_start = from_query ? qstart : dstart;
      const std::vector<uint32_t>& their_list = from_query ? dlist : qlist;
      const std::vector<uint32_t>& their_start = from_query ? dstart : qstart;
      const std::vector<uint32_t>& their_id = from_query ? did : qid;
      const std::vector<uint32_t>& my_id = from_query ? qid : did;
      const uint32_t me = mine.Top();
      mine.Take(me);
      bool assigned = false;
      for (uint32_t p = my_start[me]; p < my_start[me + 1]; ++p) {
        const uint32_t c = my_list[p];
        const uint32_t other = their_id[c];
        if (!theirs.IsFree(other)) continue;  // already assigned or taken
        if (!assigned) {
          assigned = true;
          const RetrievalCandidate& cand = candidates[lo[c]];
          GeometryMatch m;
          m.query = query_geometries[cand.query_feature];
          m.database = cand.database_geometry;
          matches.push_back(m);
          theirs.Take(other);
          // everybody on my side who also pointed at `other` has one candidate fewer
          for (uint32_t r = their_start[other]; r < their_start[other + 1]; ++r) {
            const uint32_t rival = my_id[their_list[r]];
            if (mine.IsFree(rival)) mine.OneCandidateFewer(rival);
          }
        } else {  // a candidate I no longer need: it loses me
          theirs.OneCandidateFewer(other);
        }
      }
    }
    scores[k] += VoteAndVerify(VoteAndVerifyOptions(), matches);
  }
  // re-rank (visual_index.h:486-499): the reference's own std::sort / std::partial_sort calls on the same sequence, so
  // equal scores fall the same way
  std::vector<uint32_t> order(count);
  for (uint32_t k = 0; k < count; ++k) order[k] = k;
  const uint32_t kept = std::min<uint32_t>(count, static_cast<uint32_t>(num_images_after_verification));
  auto better = [&](uint32_t a, uint32_t b) { return scores[a] > scores[b]; };
  if (kept == count)
    std::sort(order.begin(), order.end(), better);
  else
    std::partial_sort(order.begin(), order.begin() + kept, order.end(), better);
  std::vector<uint32_t> new_idx(kept);
  std::vector<float> new_scores(kept);
  for (uint32_t k = 0; k < kept; ++k) {
    new_idx[k] = image_idx[order[k]];
    new_scores[k] = scores[order[k]];
  }
  for (uint32_t k = 0; k < kept; ++k) {
    image_idx[k] = new_idx[k];
    scores[k] = new_scores[k];
  }
  return kept;
}

}  // namespace dagsfm_amd
