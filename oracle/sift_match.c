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

/* MatchSiftFeaturesCPU (sift.cc:810-822) = distance matrix + FindBestMatches (sift.cc:164-198).
 * matches_out holds n1 x 2 uint32; returns the number of matches. */
int oracle_match_sift_features_cpu(double max_ratio_d, double max_distance_d, int cross_check, const uint8_t* d1,
                                   int n1, const uint8_t* d2, int n2, uint32_t* matches_out) {
  if (n1 <= 0 || n2 <= 0) return 0; /* Eigen 0-row matrices: no matches (sift_test.cc:316-324) */
  const float max_ratio = (float)max_ratio_d;       /* narrowed at the FindBestMatches call */
  const float max_distance = (float)max_distance_d;
  int* dists = (int*)malloc((size_t)n1 * n2 * sizeof(int));
  int* m12 = (int*)malloc((size_t)n1 * sizeof(int));
  int* m21 = (int*)malloc((size_t)n2 * sizeof(int));
  compute_sift_distance_matrix(d1, n1, d2, n2, dists);
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
  free(dists);
  free(m12);
  free(m21);
  return n;
}
