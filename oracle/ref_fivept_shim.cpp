// oracle/ref_fivept_shim.cpp -- TEST INFRASTRUCTURE ONLY.  Compiles the reference's own generated 5-point
// polynomial headers (/root/reference/src/estimators/essential_matrix_poly.h, essential_matrix_coeffs.h; they are
// #included at essential_matrix.cc:76-77 and :101-102) where they lie, behind stand-ins for the four Eigen objects
// they touch (A.data(), E.data(), B.data(), coeffs(i)).  Built into oracle/_ref/libfivept_ref.so by `make ref` only
// where /root/reference exists (.cpp so that the oracle's *.cc wildcard does not pick it up); used by
// tests/test_fivept_reference_order.py to pin oracle/two_view.cc's fivept_build_A / fivept_det_poly (and with
// them the term table dagsfm_amd/csrc/fivept_terms.tbl) to the reference bit for bit.
#include <cstddef>

namespace {
struct Flat {
  double* p;
  double* data() { return p; }
  const double* data() const { return p; }
  double& operator()(int i) { return p[i]; }
};
}  // namespace

extern "C" void ref_fivept_build_A(const double* e_colmajor_9x4, double* a_colmajor_10x20) {
  const Flat E{const_cast<double*>(e_colmajor_9x4)};
  Flat A{a_colmajor_10x20};
#include "estimators/essential_matrix_poly.h"
}

extern "C" void ref_fivept_coeffs(const double* b_colmajor_13x3, double* coeffs11) {
  const Flat B{const_cast<double*>(b_colmajor_13x3)};
  Flat coeffs{coeffs11};
#include "estimators/essential_matrix_coeffs.h"
}
