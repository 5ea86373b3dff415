#!/usr/bin/env python3
"""csrc/verify_roots_refill.h = pr_hessenberg_eigenvalues and pr_poly_roots (csrc/verify_linalg.h) cut into init / one pass of the
loop / finish and begin / end.  The bodies are those functions' own lines, found by their landmarks;
tests/test_roots_refill_in_sync.py regenerates the header and compares it with the committed one.
    python tools/gen_roots_refill.py [output path]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEADER = '''// verify_roots_refill.h -- the register-resident eigenvalue iteration of verify_linalg.h (pr_hessenberg_eigenvalues) cut into
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
@@INIT@@
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
@@STEP@@
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
@@FINISH@@
}

#undef RT

// pr_poly_roots (verify_linalg.h) cut at its call of pr_hessenberg_eigenvalues.  _begin: the polynomial's degenerate forms are
// answered at once (return value = pr_poly_roots'); -2: the companion matrix is in S.T, S.n is set, the iteration has to run.
template <int MAXC>
DSM_DEV int pr_poly_roots_begin(const double (&coeffs_all)[MAXC], double (&real)[MAXC], double (&imag)[MAXC], PrEigState<MAXC - 1>& S,
                                int& degree_out) {
  constexpr int LD = MAXC - 1;
@@POLY_BEGIN@@
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
@@POLY_END@@
}

#endif  // DAGSFM_AMD_CSRC_VERIFY_ROOTS_REFILL_H_
'''


def generate():
    src = open(os.path.join(ROOT, "dagsfm_amd", "csrc", "verify_linalg.h")).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("DSM_DEV bool pr_hessenberg_eigenvalues("))
    end = next(i for i in range(start, len(src)) if src[i] == "}")
    fn = src[start:end + 1]

    def find(pred, lo=0):
        return next(i for i in range(lo, len(fn)) if pred(fn[i]))
    i_scale = find(lambda l: l.strip() == "double scale = 0.0;")
    i_failed = find(lambda l: l.strip() == "bool failed = false;")
    i_while = find(lambda l: l.strip() == "while (iu >= 0) {")
    i_after = find(lambda l: l.strip() == "if (failed) return false;")
    # the loop closes with "    }" followed by "  }" (the `if (norm != 0.0)`) right before `if (failed) return false;`
    assert fn[i_after - 1] == "  }" and fn[i_after - 2] == "    }", (fn[i_after - 2], fn[i_after - 1])
    i_ret = find(lambda l: l.strip() == "return ok;", i_after)
    init = [l for l in fn[i_scale:i_failed] if l.strip() != "const int max_iters = 40 * n;"]
    init_txt = "\n".join(init)
    assert "if (scale < DBL_MIN) return true;" in init_txt
    init_txt = init_txt.replace("if (scale < DBL_MIN) return true;", "if (scale < DBL_MIN) return false;")
    body = fn[i_while + 1:i_after - 2]
    body_txt = "\n".join(l[2:] if l.startswith("  ") else l for l in body)  # one level less deep
    assert body_txt.count("break;") == 1 and "continue;" not in body_txt and "return" not in body_txt
    body_txt = body_txt.replace("break;", "return;")
    fin_txt = "\n".join(fn[i_after + 1:i_ret + 1])
    # pr_poly_roots, cut at the call
    ps = next(i for i, l in enumerate(src) if l.startswith("DSM_DEV int pr_poly_roots("))
    pe = next(i for i in range(ps, len(src)) if src[i] == "}")
    pf = src[ps:pe + 1]
    i_ld = next(i for i, l in enumerate(pf) if l.strip() == "constexpr int LD = MAXC - 1;")
    i_re = next(i for i, l in enumerate(pf) if l.strip() == "double re[LD], im[LD];")
    i_call = next(i for i, l in enumerate(pf) if l.strip() == "if (!pr_hessenberg_eigenvalues<LD>(C, n, re, im)) return -1;")
    assert i_call == i_re + 1
    begin_txt = "\n".join(pf[i_ld + 1:i_re])
    assert begin_txt.count("double C[LD * LD];") == 1
    begin_txt = begin_txt.replace("double C[LD * LD];", "double (&C)[LD * LD] = S.T;")
    end_txt = "\n".join(pf[i_call + 1:len(pf) - 1])
    return (HEADER.replace("@@INIT@@", init_txt).replace("@@STEP@@", body_txt).replace("@@FINISH@@", fin_txt)
            .replace("@@POLY_BEGIN@@", begin_txt).replace("@@POLY_END@@", end_txt))


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "dagsfm_amd", "csrc", "verify_roots_refill.h")
    open(out, "w").write(generate())
