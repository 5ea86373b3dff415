// verify_fivept_coop.h -- the 5-point essential-matrix solver, solved cooperatively by a 16-lane group.
//
// Steps 3-5 of EssentialMatrixFivePointEstimator::Estimate
// (/root/reference/src/estimators/essential_matrix.cc:76-147) are one small dense problem per
// hypothesis (10x20 elimination, 10x10 companion eigenvalues, <= 10 3x3 SVDs).  One lane per
// hypothesis (five_point_finish_t) is right for the 64 independent minimal samples of a batch, but the
// local optimisation of LO-RANSAC is ONE solve on the critical path of a wave: here a group of 16
// lanes owns that one hypothesis, its matrices live in LDS and the element-wise updates are spread
// over the lanes (rows / columns of the 10-wide matrices).  (Measured: for the batch of minimal
// samples the per-lane form is ~2.8x faster than four such groups per wave, for the local
// optimisation the group form is ~2x faster than a single lane.)  Every floating-point operation
// and its order is the same as in the per-lane solver five_point_finish_t (verify_estimators.h), so
// results are bit-identical.
//
// All lanes of a group execute the same control flow (they read the same LDS words); different
// groups of a wave may diverge.  LDS accesses go through volatile pointers: instruction order of one
// wave is the only synchronisation that is needed between the lanes of a group.
#ifndef DAGSFM_AMD_CSRC_VERIFY_FIVEPT_COOP_H_
#define DAGSFM_AMD_CSRC_VERIFY_FIVEPT_COOP_H_

#include "verify_estimators.h"

struct G5Ws {
  double Eb[36];     // null-space basis, [r*4 + c]
  double A[200];     // [r*20 + c]; after the elimination columns 10..19 hold AA
  double ev[108];    // [e | e2 | e3] of the reference's E.data() (9 x 4 column-major): the factors of the term table
  double B[39];      // [row*3 + col]
  double bv[39];     // B.data() of the reference (13 x 3 column-major): the factors of the determinant's terms
  double coeffs[11];
  double T[100];     // companion matrix, column-major ld 10
  double re[10], im[10];
  double models[90];
  int nroots;
  int nmodels;
};

typedef volatile double* g5v;

// max over the 16 lanes of a group (exact, order independent)
DSM_DEV double g5_group_max(double v) {
  for (int o = 1; o < 16; o <<= 1) {
    const double other = __shfl_xor(v, o, 16);
    if (other > v) v = other;
  }
  return v;
}

// Eigenvalues of the n x n upper-Hessenberg companion matrix in ws->T (EigenSolver(C, false)); the
// group-cooperative twin of pl_hessenberg_eigenvalues<10>.  Returns false on non-convergence.
DSM_DEV bool g5_hessenberg_eigenvalues(G5Ws* ws, int n, int gl) {
  g5v T = ws->T;
#define TT(r, c) T[(c) * 10 + (r)]
  if (gl < n) {
    ws->re[gl] = 0.0;
    ws->im[gl] = 0.0;
  }
  if (n == 0) return true;
  double mx = 0.0;
  if (gl < n)
    for (int i = 0; i < n; ++i) {
      const double a = fabs(TT(i, gl));
      if (a > mx) mx = a;
    }
  const double scale = g5_group_max(mx);
  if (scale < DBL_MIN) return true;
  if (gl < n)
    for (int i = 0; i < n; ++i) TT(i, gl) = TT(i, gl) / scale;
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
        const double d = TT(iu, iu) + exshift;
        if (gl == 0) {
          TT(iu, iu) = d;
          if (iu > 0) TT(iu, iu - 1) = 0.0;
        }
        iu--;
        iter = 0;
      } else if (il == iu - 1) {
        const double a11 = TT(iu - 1, iu - 1), a22 = TT(iu, iu), a21 = TT(iu, iu - 1), a12 = TT(iu - 1, iu);
        const double p = 0.5 * (a11 - a22);
        const double q = p * p + a21 * a12;
        if (gl == 0) {
          TT(iu, iu) = a22 + exshift;
          TT(iu - 1, iu - 1) = a11 + exshift;
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
          if (!(gc == 1.0 && -gs == 0.0)) {
            if (gl >= iu - 1 && gl < n) {  // rows iu-1, iu, one column per lane
              const double xi = TT(iu - 1, gl), yi = TT(iu, gl);
              TT(iu - 1, gl) = gc * xi + (-gs) * yi;
              TT(iu, gl) = gs * xi + gc * yi;
            }
            if (gl <= iu) {  // columns iu-1, iu, one row per lane
              const double xi = TT(gl, iu - 1), yi = TT(gl, iu);
              TT(gl, iu - 1) = gc * xi + (-gs) * yi;
              TT(gl, iu) = gs * xi + gc * yi;
            }
          }
          if (gl == 0) TT(iu, iu - 1) = 0.0;
        }
        if (iu > 1 && gl == 0) TT(iu - 1, iu - 2) = 0.0;
        iu -= 2;
        iter = 0;
      } else {
        double sh0 = TT(iu, iu), sh1 = TT(iu - 1, iu - 1), sh2 = TT(iu, iu - 1) * TT(iu - 1, iu);
        if (iter == 10) {
          exshift += sh0;
          if (gl <= iu) TT(gl, gl) = TT(gl, gl) - sh0;
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
            if (gl <= iu) TT(gl, gl) = TT(gl, gl) - s;
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
          // makeHouseholder on the 3-vector (uniform over the group)
          double tau, beta, e0, e1;
          {
            const double tail_sq = v[1] * v[1] + v[2] * v[2];
            const double c0 = v[0];
            if (tail_sq <= DBL_MIN) {
              tau = 0.0;
              beta = c0;
              e0 = 0.0;
              e1 = 0.0;
            } else {
              double b = sqrt(c0 * c0 + tail_sq);
              if (c0 >= 0.0) b = -b;
              e0 = v[1] / (c0 - b);
              e1 = v[2] / (c0 - b);
              tau = (b - c0) / b;
              beta = b;
            }
          }
          if (beta != 0.0) {
            if (first && k > il) {
              const double t = -TT(k, k - 1);
              if (gl == 0) TT(k, k - 1) = t;
            } else if (!first) {
              if (gl == 0) TT(k, k - 1) = beta;
            }
            if (tau != 0.0) {
              if (gl >= k && gl < n) {  // applyHouseholderOnTheLeft: rows k..k+2, one column per lane
                double tmp = 0.0;
                tmp += e0 * TT(k + 1, gl);
                tmp += e1 * TT(k + 2, gl);
                tmp += TT(k, gl);
                TT(k, gl) = TT(k, gl) - tau * tmp;
                TT(k + 1, gl) = TT(k + 1, gl) - tau * e0 * tmp;
                TT(k + 2, gl) = TT(k + 2, gl) - tau * e1 * tmp;
              }
              const int nr = ((iu < k + 3) ? iu : k + 3) + 1;
              if (gl < nr) {  // applyHouseholderOnTheRight: columns k..k+2, one row per lane
                double tmp = 0.0;
                tmp += TT(gl, k + 1) * e0;
                tmp += TT(gl, k + 2) * e1;
                tmp += TT(gl, k);
                TT(gl, k) = TT(gl, k) - tau * tmp;
                TT(gl, k + 1) = TT(gl, k + 1) - tau * tmp * e0;
                TT(gl, k + 2) = TT(gl, k + 2) - tau * tmp * e1;
              }
            }
          }
        }
        {
          const double c0 = TT(iu - 1, iu - 2), x1 = TT(iu, iu - 2);
          double tau, beta, e0;
          const double tail_sq = x1 * x1;
          if (tail_sq <= DBL_MIN) {
            tau = 0.0;
            beta = c0;
            e0 = 0.0;
          } else {
            double b = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0.0) b = -b;
            e0 = x1 / (c0 - b);
            tau = (b - c0) / b;
            beta = b;
          }
          if (beta != 0.0) {
            if (gl == 0) TT(iu - 1, iu - 2) = beta;
            if (tau != 0.0) {
              if (gl >= iu - 1 && gl < n) {  // rows iu-1, iu
                double tmp = 0.0;
                tmp += e0 * TT(iu, gl);
                tmp += TT(iu - 1, gl);
                TT(iu - 1, gl) = TT(iu - 1, gl) - tau * tmp;
                TT(iu, gl) = TT(iu, gl) - tau * e0 * tmp;
              }
              if (gl <= iu) {  // columns iu-1, iu
                double tmp = 0.0;
                tmp += TT(gl, iu) * e0;
                tmp += TT(gl, iu - 1);
                TT(gl, iu - 1) = TT(gl, iu - 1) - tau * tmp;
                TT(gl, iu) = TT(gl, iu) - tau * tmp * e0;
              }
            }
          }
        }
        if (gl >= imm + 2 && gl <= iu) {
          TT(gl, gl - 2) = 0.0;
          if (gl > imm + 2) TT(gl, gl - 3) = 0.0;
        }
      }
    }
  }
  if (total_iter > max_iters) return false;
  if (gl < n)
    for (int i = 0; i < n; ++i) TT(i, gl) = TT(i, gl) * scale;
  bool ok = true;
  int i = 0;
  while (i < n) {
    if (i == n - 1 || TT(i + 1, i) == 0.0) {
      const double r = TT(i, i);
      if (gl == 0) {
        ws->re[i] = r;
        ws->im[i] = 0.0;
      }
      if (!isfinite(r)) {
        ok = false;
        break;
      }
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
      const double r = TT(i + 1, i + 1) + p;
      if (gl == 0) {
        ws->re[i] = r;
        ws->im[i] = z;
        ws->re[i + 1] = r;
        ws->im[i + 1] = -z;
      }
      if (!(isfinite(r) && isfinite(z))) {
        ok = false;
        break;
      }
      i += 2;
    }
  }
  return ok;
#undef TT
}

// FindPolynomialRootsCompanionMatrix on ws->coeffs (11 coefficients, highest degree first); roots land
// in ws->re / ws->im.  Returns the number of roots or -1.
DSM_DEV int g5_poly_roots(G5Ws* ws, int gl) {
  g5v c = ws->coeffs;
  int lead = 0;
  for (; lead < 11; ++lead)
    if (c[lead] != 0) break;
  int nc = 11 - lead;
  const int degree = nc - 1;
  if (degree <= 0) return -1;
  if (degree == 1) {
    const double c0 = c[lead], c1 = c[lead + 1];
    if (c0 == 0) return -1;
    if (gl == 0) {
      ws->re[0] = -c1 / c0;
      ws->im[0] = 0.0;
    }
    return 1;
  }
  if (degree == 2) {
    const double a = c[lead], b = c[lead + 1], cc = c[lead + 2];
    if (b == 0 && cc == 0) {
      if (gl == 0) {
        ws->re[0] = 0.0;
        ws->im[0] = 0.0;
      }
      return 1;
    }
    const double d = b * b - 4 * a * cc;
    double r0, r1, i0, i1;
    if (d >= 0) {
      const double sqrt_d = sqrt(d);
      if (b >= 0) {
        r0 = (-b - sqrt_d) / (2 * a);
        r1 = (2 * cc) / (-b - sqrt_d);
      } else {
        r0 = (2 * cc) / (-b + sqrt_d);
        r1 = (-b + sqrt_d) / (2 * a);
      }
      i0 = 0.0;
      i1 = 0.0;
    } else {
      r0 = r1 = -b / (2 * a);
      i0 = sqrt(-d) / (2 * a);
      i1 = -i0;
    }
    if (gl == 0) {
      ws->re[0] = r0;
      ws->re[1] = r1;
      ws->im[0] = i0;
      ws->im[1] = i1;
    }
    return 2;
  }
  int trail = 0;
  for (; trail < nc; ++trail)
    if (c[lead + nc - 1 - trail] != 0) break;
  nc -= trail;
  if (nc == 1) {
    if (gl == 0) {
      ws->re[0] = 0.0;
      ws->im[0] = 0.0;
    }
    return 1;
  }
  const int n = nc - 1;
  if (gl < n) {  // companion matrix, one column per lane
    for (int i = 0; i < n; ++i) ws->T[gl * 10 + i] = 0.0;
    if (gl + 1 < n) ws->T[gl * 10 + gl + 1] = 1.0;
    ws->T[gl * 10 + 0] = -c[lead + gl + 1] / c[lead];
  }
  if (!g5_hessenberg_eigenvalues(ws, n, gl)) return -1;
  const int effective_degree = n < degree ? n + 1 : n;
  if (gl >= n && gl < effective_degree) {
    ws->re[gl] = 0.0;
    ws->im[gl] = 0.0;
  }
  return effective_degree;
}

// The whole of five_point_finish_t for the basis in ws->Eb.  Must be called by all 16 lanes of the
// group (gl = lane within the group, group_shift = first lane of the group in the wave).  Returns the
// number of models; they are in ws->models[k*9 ..].
DSM_DEV int g5_five_point_finish(G5Ws* ws, int gl, int group_shift) {
  g5v Eb = ws->Eb;
  g5v A = ws->A;
  LSEC_BEGIN();
  // step 3: the reference's generated code, interpreted from the packed term table (five_point_eval_terms), the 200
  // entries dealt out to the 16 lanes
  for (int k = gl; k < 36; k += 16) {
    const double e = Eb[(k % 9) * 4 + k / 9];
    const double e2 = e * e;
    ws->ev[k] = e;
    ws->ev[36 + k] = e2;
    ws->ev[72 + k] = e2 * e;
  }
  for (int t = gl; t < 200; t += 16)
    A[(t % 10) * 20 + t / 10] = five_point_eval_terms(kFivePtATerms, kFivePtAOff[t], kFivePtAOff[t + 1], (g5v)ws->ev);
  LSEC_END(8);
  LSEC_BEGIN2();
  // ---- A[:, :10].partialPivLu().solve(A[:, 10:]) on the augmented 10 x 20 matrix
#define AE(r, c) A[(r) * 20 + (c)]
  for (int k = 0; k < 10; ++k) {
    int r = k;
    double best = fabs(AE(k, k));
    for (int i = k + 1; i < 10; ++i) {
      const double v = fabs(AE(i, k));
      if (v > best) {
        best = v;
        r = i;
      }
    }
    if (best != 0.0) {
      if (r != k) {
        for (int cc = gl; cc < 20; cc += 16) {
          const double t = AE(k, cc);
          AE(k, cc) = AE(r, cc);
          AE(r, cc) = t;
        }
      }
      const double pivot = AE(k, k);
      if (gl > k && gl < 10) AE(gl, k) = AE(gl, k) / pivot;
    }
    for (int cc = gl; cc < 20; cc += 16) {
      if (cc > k) {
        const double akc = AE(k, cc);
        for (int i = k + 1; i < 10; ++i) AE(i, cc) = AE(i, cc) - AE(i, k) * akc;
      }
    }
  }
  if (gl < 10) {  // back substitution, one right-hand side per lane
    const int cc = 10 + gl;
    for (int i = 9; i >= 0; --i) {
      double s = AE(i, cc);
      for (int k = 9; k > i; --k) s -= AE(i, k) * AE(k, cc);
      AE(i, cc) = s / AE(i, i);
    }
  }
#define AAe(r, c) AE(r, 10 + (c))
  if (gl < 3) {
    const int i = gl;
    g5v B = ws->B;
    B[0 * 3 + i] = 0;
    B[4 * 3 + i] = 0;
    B[8 * 3 + i] = 0;
    for (int k = 0; k < 3; ++k) {
      B[(1 + k) * 3 + i] = AAe(i * 2 + 4, k);
      B[(5 + k) * 3 + i] = AAe(i * 2 + 4, 3 + k);
    }
    for (int k = 0; k < 4; ++k) B[(9 + k) * 3 + i] = AAe(i * 2 + 4, 6 + k);
    for (int k = 0; k < 3; ++k) {
      B[(0 + k) * 3 + i] = B[(0 + k) * 3 + i] - AAe(i * 2 + 5, k);
      B[(4 + k) * 3 + i] = B[(4 + k) * 3 + i] - AAe(i * 2 + 5, 3 + k);
    }
    for (int k = 0; k < 4; ++k) B[(8 + k) * 3 + i] = B[(8 + k) * 3 + i] - AAe(i * 2 + 5, 6 + k);
  }
#undef AAe
#undef AE
  LSEC_END2(9);
  LSEC_BEGIN3();
  // ---- determinant polynomial of B(z): one coefficient per lane (essential_matrix_coeffs.h, term table)
  for (int k = gl; k < 39; k += 16) ws->bv[k] = ws->B[(k % 13) * 3 + k / 13];
  if (gl < 11) ws->coeffs[gl] = five_point_eval_terms(kFivePtCTerms, kFivePtCOff[gl], kFivePtCOff[gl + 1], (g5v)ws->bv);
  LSEC_END3(10);
  LSEC_BEGIN4();
  const int nroots = g5_poly_roots(ws, gl);
  LSEC_END4(11);
  if (nroots < 0) return 0;
  LSEC_BEGIN5();
  // ---- one root per lane
  bool valid = false;
  double model[9];
  if (gl < nroots) {
    const double rim = ws->im[gl];
    if (!(fabs(rim) > 1e-10)) {
      const double z1 = ws->re[gl];
      const double z2 = z1 * z1;
      const double z3 = z2 * z1;
      const double z4 = z3 * z1;
      double Bz[9];
      g5v B = ws->B;
      for (int j = 0; j < 3; ++j) {
        Bz[j * 3 + 0] = B[0 * 3 + j] * z3 + B[1 * 3 + j] * z2 + B[2 * 3 + j] * z1 + B[3 * 3 + j];
        Bz[j * 3 + 1] = B[4 * 3 + j] * z3 + B[5 * 3 + j] * z2 + B[6 * 3 + j] * z1 + B[7 * 3 + j];
        Bz[j * 3 + 2] = B[8 * 3 + j] * z4 + B[9 * 3 + j] * z3 + B[10 * 3 + j] * z2 + B[11 * 3 + j] * z1 + B[12 * 3 + j];
      }
      double Vz[9], svz[3];
      pl_jacobi_svd_square<3, false>(Bz, nullptr, Vz, svz);
      const double X0 = Vz[2 * 3 + 0], X1 = Vz[2 * 3 + 1], X2 = Vz[2 * 3 + 2];
      if (!(fabs(X2) < 1e-10)) {
        const double sx = X0 / X2, sy = X1 / X2;
        double ev[9];
        for (int k = 0; k < 9; ++k) ev[k] = ws->Eb[k * 4 + 0] * sx + ws->Eb[k * 4 + 1] * sy + ws->Eb[k * 4 + 2] * z1 + ws->Eb[k * 4 + 3];
        double nn = 0.0;
        for (int k = 0; k < 9; ++k) nn += ev[k] * ev[k];
        const double norm = sqrt(nn);
        for (int k = 0; k < 9; ++k) model[k] = ev[k] / norm;
        valid = true;
      }
    }
  }
  // models keep the order of the roots
  const unsigned long long bal = __ballot(valid);
  const unsigned gm = (unsigned)((bal >> group_shift) & 0xffffull);
  const int pos = __popc(gm & ((1u << gl) - 1u));
  if (valid)
    for (int k = 0; k < 9; ++k) ws->models[pos * 9 + k] = model[k];
  LSEC_END5(12);
  return __popc(gm);
}

#endif  // DAGSFM_AMD_CSRC_VERIFY_FIVEPT_COOP_H_
