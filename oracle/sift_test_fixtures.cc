// oracle/sift_test_fixtures.cc -- TEST INFRASTRUCTURE ONLY.
//
// Restates the descriptor fixture generator of the reference's matcher tests so that their
// known answers can be re-checked here:
//   CreateRandomFeatureDescriptors     /root/reference/src/feature/sift_test.cc:243-253
//   L2NormalizeFeatureDescriptors      /root/reference/src/feature/utils.cc:48-51
//   FeatureDescriptorsToUnsignedByte   /root/reference/src/feature/utils.cc:65-77
// PRNG: SetPRNGSeed(0) -> std::mt19937(0); RandomReal<float>(0,1) ->
// std::uniform_real_distribution<float> (/root/reference/src/util/random.h:100-109).  This uses
// the real libstdc++ classes, like the reference.  The row norm is accumulated sequentially in
// float (Eigen's vectorised summation order is not available here; see DESIGN.md "oracle").
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

static uint8_t truncate_cast_u8(float v) {  // TruncateCast<float,uint8_t>, /root/reference/src/util/math.h
  if (v < 0.0f) return 0;
  if (v > 255.0f) return 255;
  return static_cast<uint8_t>(v);
}

extern "C" void oracle_l2_normalize_to_u8(const float* row, uint8_t* out) {
  float ss = 0.0f;
  for (int j = 0; j < 128; ++j) ss += row[j] * row[j];
  const float norm = std::sqrt(ss);
  for (int j = 0; j < 128; ++j) {
    const float scaled = std::round(512.0f * (row[j] / norm));
    out[j] = truncate_cast_u8(scaled);
  }
}

extern "C" void oracle_create_random_feature_descriptors(int num_features, uint8_t* out) {
  std::mt19937 prng(0);
  std::vector<float> row(128);
  for (int i = 0; i < num_features; ++i) {
    for (int j = 0; j < 128; ++j) {
      std::uniform_real_distribution<float> distribution(0.0f, 1.0f);
      row[j] = static_cast<float>(std::pow(distribution(prng), 2));
    }
    oracle_l2_normalize_to_u8(row.data(), out + static_cast<size_t>(i) * 128);
  }
}
