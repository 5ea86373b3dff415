// oracle/ref_misc_shim.cpp -- TEST INFRASTRUCTURE ONLY.  Two more pieces of the reference that are plain standard C++ and
// compile where they lie, with nothing of the reference's build system and no stand-in for anything:
//   src/optim/support_measurement.{h,cc}   InlierSupportMeasurer::Evaluate / Compare -- the comparison every RANSAC trial of
//                                          the hot path goes through (loransac.h:147-150, 177-181)
//   src/retrieval/utils.h                  HammingDistWeightFunctor<64>: the voting weights of the retrieval
// Built into oracle/_ref/libmisc_ref.so by `make ref` only where /root/reference exists (.cpp so that the oracle's *.cc
// wildcard does not pick it up); tests/test_oracle_estimators.py and tests/test_retrieval.py pin oracle/two_view.cc's
// EvaluateSupport / CompareSupport and the Hamming weight table (oracle, host shim, device) to it bit for bit.
#include <cstddef>
#include <cstdint>
#include <vector>

#include "optim/support_measurement.cc"
#include "retrieval/utils.h"

extern "C" {

void ref_inlier_support(const double* residuals, uint64_t n, double max_residual, uint64_t* num_inliers, double* residual_sum) {
  colmap::InlierSupportMeasurer measurer;
  const auto s = measurer.Evaluate(std::vector<double>(residuals, residuals + n), max_residual);
  *num_inliers = s.num_inliers;
  *residual_sum = s.residual_sum;
}

int ref_inlier_support_compare(uint64_t num_inliers1, double residual_sum1, uint64_t num_inliers2, double residual_sum2) {
  colmap::InlierSupportMeasurer measurer;
  colmap::InlierSupportMeasurer::Support a, b;
  a.num_inliers = num_inliers1;
  a.residual_sum = residual_sum1;
  b.num_inliers = num_inliers2;
  b.residual_sum = residual_sum2;
  return measurer.Compare(a, b) ? 1 : 0;
}

float ref_hamming_weight(uint32_t hamming_dist) {
  static const colmap::retrieval::HammingDistWeightFunctor<64> functor;
  return functor(hamming_dist);
}
uint32_t ref_max_hamming_distance() { return static_cast<uint32_t>(colmap::retrieval::HammingDistWeightFunctor<64>::kMaxHammingDistance); }

}  // extern "C"
