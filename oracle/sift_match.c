/*
 * oracle/sift_match.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the reference's brute-force SIFT matcher.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (dagsfm_amd/, include/) never does.
 *
 * Follows, line by line:
 *   ComputeSiftDistanceMatrix  /root/reference/src/feature/sift.cc:76-109
 *   FindBestMatchesOneWay      /root/reference/src/feature/sift.cc:111-162
 *   FindBestMatches            /root/reference/src/feature/sift.cc:164-198
 *   MatchSiftFeaturesCPU       /root/reference/src/feature/sift.cc:810-822
 *   MatchGuidedSiftFeaturesCPU /root/reference/src/feature/sift.cc:824-875 (guided_filter of :76-109)
 *
 * Parity pinning: the reference's own known-answer tests for this function
 * (/root/reference/src/feature/sift_test.cc:300-325 and the CPU-vs-GPU counts at :505-571)
 * are restated in tests/test_oracle_match.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* sift.cc:76-109 -- dists(i1,i2) = <d1[i1], d2[i2]> in int (guided_filter == nullptr). */
static void compute_sift_distance_matrix(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int* dists) {
  for (int i1 = 0; i1 < n1; ++i1) {
    const uint8_t* a = d1 + (size_t)i1 * 128;
    for (int i2 = 0; i2 < n2; ++i2) {
      const uint8_t* b = d2 + (size_t)i2 * 128;
      int dot = 0;
      for (int k = 0; k < 128; ++k) dot += (int)a[k] * (int)b[k];
      dists[(size_t)i1 * n2 + i2] = dot;
    }
  }
}

/* The guided filters of MatchGuidedSiftFeaturesCPU (sift.cc:838-866), float arithmetic: true = the pair of
 * keypoints violates the geometry, its distance entry becomes 0.  mode 1: Sampson error against F (configs
 * CALIBRATED / UNCALIBRATED), mode 2: transfer error of H (PLANAR / PANORAMIC / PLANAR_OR_PANORAMIC).  M is
 * F / H cast to float, row-major.  The reference evaluates these expressions through Eigen's fixed-size float
 * kernels, whose association order is not specified by the reference; the order below (products summed left to
 * right) is this oracle's definition and the device code uses the same one.  Parity unpinned: the reference
 * has no test of guided matching. */
int oracle_guided_filter(int mode, const float* M, float x1, float y1, float x2, float y2, float max_residual) {
  if (mode == 1) {
    const float Fx1_0 = (M[0] * x1 + M[1] * y1) + M[2];
    const float Fx1_1 = (M[3] * x1 + M[4] * y1) + M[5];
    const float Fx1_2 = (M[6] * x1 + M[7] * y1) + M[8];
    const float Ftx2_0 = (M[0] * x2 + M[3] * y2) + M[6];
    const float Ftx2_1 = (M[1] * x2 + M[4] * y2) + M[7];
    const float x2tFx1 = (x2 * Fx1_0 + y2 * Fx1_1) + Fx1_2;
    return x2tFx1 * x2tFx1 / (((Fx1_0 * Fx1_0 + Fx1_1 * Fx1_1) + Ftx2_0 * Ftx2_0) + Ftx2_1 * Ftx2_1) > max_residual;
  }
  const float Hp_0 = (M[0] * x1 + M[1] * y1) + M[2];
  const float Hp_1 = (M[3] * x1 + M[4] * y1) + M[5];
  const float Hp_2 = (M[6] * x1 + M[7] * y1) + M[8];
  const float d0 = Hp_0 / Hp_2 - x2, d1 = Hp_1 / Hp_2 - y2;
  return d0 * d0 + d1 * d1 > max_residual;
}

/* sift.cc:111-162.  `dists` is rows x cols with element (i1,i2) at dists[i1*rs + i2*cs]
 * so that the transposed call of sift.cc:175-176 needs no copy. */
static size_t find_best_matches_one_way(const int* dists, int rows, int cols, size_t rs, size_t cs,
                                        float max_ratio, float max_distance, int* matches) {
  const float kDistNorm = 1.0f / (512.0f * 512.0f);
  size_t num_matches = 0;
  for (int i1 = 0; i1 < rows; ++i1) matches[i1] = -1;
  for (int i1 = 0; i1 < rows; ++i1) {
    int best_i2 = -1;
    int best_dist = 0;
    int second_best_dist = 0;
    for (int i2 = 0; i2 < cols; ++i2) {
      const int dist = dists[(size_t)i1 * rs + (size_t)i2 * cs];
      if (dist > best_dist) {
        best_i2 = i2;
        second_best_dist = best_dist;
        best_dist = dist;
      } else if (dist > second_best_dist) {
        second_best_dist = dist;
      }
    }
    if (best_i2 == -1) continue;
    const float best_dist_normed = acosf(fminf(kDistNorm * (float)best_dist, 1.0f));
    if (best_dist_normed > max_distance) continue;
    const float second_best_dist_normed = acosf(fminf(kDistNorm * (float)second_best_dist, 1.0f));
    if (best_dist_normed >= max_ratio * second_best_dist_normed) continue;
    num_matches += 1;
    matches[i1] = best_i2;
  }
  return num_matches;
}

/* FindBestMatches (sift.cc:164-198) on a finished distance matrix. */
static int find_best_matches(int* dists, int n1, int n2, float max_ratio, float max_distance, int cross_check,
                             uint32_t* matches_out);

/* MatchGuidedSiftFeaturesCPU (sift.cc:824-875): mode 0 (any other configuration) returns -1 and leaves the
 * inlier matches alone (:861-863); kp: n x 2 float (x, y); max_error is the double option, squared in double and
 * narrowed to float as at :833. */
int oracle_match_guided_sift_features_cpu(double max_ratio_d, double max_distance_d, int cross_check, double max_error,
                                          const float* kp1, const float* kp2, const uint8_t* d1, int n1, const uint8_t* d2,
                                          int n2, int mode, const double* M_d, uint32_t* matches_out) {
  if (mode != 1 && mode != 2) return -1;
  if (n1 <= 0 || n2 <= 0) return 0;
  const float max_residual = (float)(max_error * max_error);
  float M[9];
  for (int k = 0; k < 9; ++k) M[k] = (float)M_d[k];
  int* dists = (int*)malloc((size_t)n1 * n2 * sizeof(int));
  for (int i1 = 0; i1 < n1; ++i1) {
    const uint8_t* a = d1 + (size_t)i1 * 128;
    for (int i2 = 0; i2 < n2; ++i2) {
      if (oracle_guided_filter(mode, M, kp1[2 * i1], kp1[2 * i1 + 1], kp2[2 * i2], kp2[2 * i2 + 1], max_residual)) {
        dists[(size_t)i1 * n2 + i2] = 0;
      } else {
        const uint8_t* b = d2 + (size_t)i2 * 128;
        int dot = 0;
        for (int k = 0; k < 128; ++k) dot += (int)a[k] * (int)b[k];
        dists[(size_t)i1 * n2 + i2] = dot;
      }
    }
  }
  const int n = find_best_matches(dists, n1, n2, (float)max_ratio_d, (float)max_distance_d, cross_check, matches_out);
  free(dists);
  return n;
}

/* MatchSiftFeaturesCPU (sift.cc:810-822) = distance matrix + FindBestMatches (sift.cc:164-198).
 * matches_out holds n1 x 2 uint32; returns the number of matches. */
int oracle_match_sift_features_cpu(double max_ratio_d, double max_distance_d, int cross_check, const uint8_t* d1,
                                   int n1, const uint8_t* d2, int n2, uint32_t* matches_out) {
  if (n1 <= 0 || n2 <= 0) return 0; /* Eigen 0-row matrices: no matches (sift_test.cc:316-324) */
  int* dists = (int*)malloc((size_t)n1 * n2 * sizeof(int));
  compute_sift_distance_matrix(d1, n1, d2, n2, dists);
  const int n = find_best_matches(dists, n1, n2, (float)max_ratio_d, (float)max_distance_d, cross_check, matches_out);
  free(dists);
  return n;
}

static int find_best_matches(int* dists, int n1, int n2, float max_ratio, float max_distance, int cross_check,
                             uint32_t* matches_out) {
  int* m12 = (int*)malloc((size_t)n1 * sizeof(int));
  int* m21 = (int*)malloc((size_t)n2 * sizeof(int));
  find_best_matches_one_way(dists, n1, n2, (size_t)n2, 1, max_ratio, max_distance, m12);
  int n = 0;
  if (cross_check) {
    find_best_matches_one_way(dists, n2, n1, 1, (size_t)n2, max_ratio, max_distance, m21);
    for (int i1 = 0; i1 < n1; ++i1) {
      if (m12[i1] != -1 && m21[m12[i1]] != -1 && m21[m12[i1]] == i1) {
        matches_out[2 * n] = (uint32_t)i1;
        matches_out[2 * n + 1] = (uint32_t)m12[i1];
        ++n;
      }
    }
  } else {
    for (int i1 = 0; i1 < n1; ++i1) {
      if (m12[i1] != -1) {
        matches_out[2 * n] = (uint32_t)i1;
        matches_out[2 * n + 1] = (uint32_t)m12[i1];
        ++n;
      }
    }
  }
  free(m12);
  free(m21);
  return n;
}
