// verify_roots_refill.h -- the register-resident eigenvalue iteration of verify_linalg.h (pr_hessenberg_eigenvalues) cut into
// init / ONE pass of its loop / finish, and pr_poly_roots cut at its call of it, so that the iteration can run in a kernel of its
// own in which a lane that has finished its matrix takes the next one while the other lanes of the wave iterate on
// (k_roots_e_init -> k_roots_e_iter -> k_roots_e_finish, EXPERIMENTAL: DSM_ROOTS_REFILL=1).
//
// GENERATED TEXT: the bodies below are the two functions' own lines (tools/gen_roots_refill.py cuts them out of verify_linalg.h;
// tests/test_roots_refill_in_sync.py fails when the two files drift apart).  Same operations on the same values in the same order
// per lane; only what a wave executes together changes.
#ifndef DAGSFM_AMD_CSRC_VERIFY_ROOTS_REFILL_H_
#define DAGSFM_AMD_CSRC_VERIFY_ROOTS_REFILL_H_

#include "verify_linalg.h"

template <int N>
struct PrEigState {
  double T[N * N];  // column-major, upper Hessenberg + zeros (see pr_hessenberg_eigenvalues)
  double scale, norm, exshift;
  int n, iu, iter, total_iter;
  bool failed;
};

#define RT(r, c) T[(c) * N + (r)]

// After the caller has filled S.T and S.n.  Returns false when there is nothing to iterate and nothing to extract (n == 0 or a
// matrix of zeros: all eigenvalues 0, pr_hessenberg_eigenvalues' two early returns); otherwise S.iu >= 0 says whether pr_eig_step
// has work (norm == 0 leaves the loop out as well) and pr_eig_finish extracts the eigenvalues.
template <int N>
DSM_DEV bool pr_eig_init(PrEigState<N>& S) {
  double (&T)[N * N] = S.T;
  const int n = S.n;
  S.failed = false;
  S.iu = -1;
  S.iter = 0;
  S.total_iter = 0;
  S.exshift = 0.0;
  S.norm = 0.0;
  S.scale = 0.0;
  if (n == 0) return false;
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
  if (scale < DBL_MIN) return false;
  {
    // (a companion matrix is mostly exact zeros: its group never passes the guard; kept as the plain division)
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i <= j + 1) RT(i, j) /= scale;
    }
  }
  int iu = n - 1, iter = 0, total_iter = 0;
  double exshift = 0.0;
  double norm = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i <= j + 1) norm += fabs(RT(i, j));  // entries outside n x n are zeros: + 0.0 changes nothing
  }
  S.scale = scale;
  S.norm = norm;
  S.exshift = exshift;
  S.iter = iter;
  S.total_iter = total_iter;
  S.iu = norm != 0.0 ? iu : -1;
  return true;
}

// One pass of the loop `while (iu >= 0)`; precondition S.iu >= 0 && !S.failed.
template <int N>
DSM_DEV void pr_eig_step(PrEigState<N>& S) {
  double (&T)[N * N] = S.T;
  int& iu = S.iu;
  int& iter = S.iter;
  int& total_iter = S.total_iter;
  double& exshift = S.exshift;
  const double norm = S.norm;
  bool& failed = S.failed;
  const int max_iters = 40 * S.n;
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
        return;
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

// The eigenvalues of the finished iteration (false: the iteration failed or a value is not finite).
template <int N>
DSM_DEV bool pr_eig_finish(PrEigState<N>& S, double (&re)[N], double (&im)[N]) {
  double (&T)[N * N] = S.T;
  const int n = S.n;
  const double scale = S.scale;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    re[i] = 0.0;
    im[i] = 0.0;
  }
  if (S.failed) return false;
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
}

#undef RT

// pr_poly_roots (verify_linalg.h) cut at its call of pr_hessenberg_eigenvalues.  _begin: the polynomial's degenerate forms are
// answered at once (return value = pr_poly_roots'); -2: the companion matrix is in S.T, S.n is set, the iteration has to run.
template <int MAXC>
DSM_DEV int pr_poly_roots_begin(const double (&coeffs_all)[MAXC], double (&real)[MAXC], double (&imag)[MAXC], PrEigState<MAXC - 1>& S,
                                int& degree_out) {
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
  double (&C)[LD * LD] = S.T;
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
  S.n = n;
  degree_out = degree;
  return -2;
}
// _end: ok / re / im = pr_eig_finish's result (or true and zeros where pr_eig_init had nothing to iterate).
template <int MAXC>
DSM_DEV int pr_poly_roots_end(bool ok, const double (&re)[MAXC - 1], const double (&im)[MAXC - 1], int n, int degree, double (&real)[MAXC],
                              double (&imag)[MAXC]) {
  constexpr int LD = MAXC - 1;
  if (!ok) return -1;
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

#endif  // DAGSFM_AMD_CSRC_VERIFY_ROOTS_REFILL_H_
