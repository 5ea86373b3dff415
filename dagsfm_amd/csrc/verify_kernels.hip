// verify_kernels.hip -- two-view geometric verification on MI355X (gfx950).
//
// One 64-lane wavefront (one-wave workgroup) verifies one image pair at a time and walks the
// pair list with a grid stride.  Per pair it reproduces, decision for decision, the reference's
//   TwoViewGeometry::Estimate            /root/reference/src/estimators/two_view_geometry.cc:113-126
//   EstimateCalibrated / Uncalibrated    :292-489      EstimateWithRelativePose :232-290
//   DetectWatermark                      :491-555
//   LORANSAC<E, LE>::Estimate            /root/reference/src/optim/loransac.h:91-233
//   RandomSampler / Shuffle / mt19937    /root/reference/src/optim/random_sampler.cc:43-62, util/random.h:122-129
// with a defined per-pair seed (the reference seeds from the wall clock, random.cc:40-56).
//
// LO-RANSAC is sequential ("first best wins", adaptive stop).  The wave speculates a batch of 64
// trials: lane 0 draws the 64 minimal samples from the pair's MT19937 stream (libstdc++'s
// Lemire uniform_int mapping), every lane solves one minimal problem, the wave then scores every
// model of the batch (wavefront-per-hypothesis: lanes stride over the correspondences,
// __ballot/popc for the inlier count) and finally replays the batch in trial order: support
// comparison, in-order residual_sum for candidates, local optimisation on improvement, dynamic
// stop.  On an early stop the PRNG is rewound to the state after the last consumed sample, so
// the next model family continues on exactly the stream position the sequential code would.
//
// FP64 throughout, compiled with -ffp-contract=off; every reduction is summed in index order.
// RANSAC::ComputeNumTrials (ransac.h:150-167) needs libm pow/log/ceil; it is tabulated on the
// host with the host libm per (num_samples, min_samples, confidence) and looked up here, the
// same construction as the acos LUT of the matcher.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dagsfm_mi355x.h"
#include "kernels.h"
#include "verify_estimators.h"
#include "verify_camera.h"
#include "verify_fivept_coop.h"

#define BATCH 64

// ------------------------------------------------------------------------------------ shared state
struct MtState {
  uint32_t mt[624];
  int mti;
  uint32_t calls;  // raw generator calls since the last reset
};

// LDS of one verification workgroup.  The replay kernels only allocate the prefix up to `gen`, the
// final kernel up to `mt_bak`; the tail is used by the in-kernel LO-RANSAC (lo_ransac) only.
struct VSmem {
  WvSvdShared svd;
  double sv[9];
  double lo_models[90];
  double cur_model[9];
  double bcast[16];
  int ibcast[8];
  G5Ws g5;                 // LDS workspace of the group-cooperative 5-point solver (local optimisation)
  MtState gen;
  uint32_t mt_bak[624];
  int mti_bak;
  int sample[BATCH * 7];
  uint32_t draws_end[BATCH];
  int nmodels[BATCH];
  int counts[BATCH * 10];
};

// ------------------------------------------------------------------------------------ MT19937 (lane 0)
DSM_DEV void mt_seed(MtState* s, uint32_t seed) {
  s->mt[0] = seed;
  for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->mti = 624;
}
DSM_DEV uint32_t mt_next(MtState* s) {
  if (s->mti >= 624) {
    uint32_t* mt = s->mt;
    int kk;
    for (kk = 0; kk < 624 - 397; ++kk) {
      const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < 623; ++kk) {
      const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    s->mti = 0;
  }
  uint32_t y = s->mt[s->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  s->calls++;
  return y;
}
// std::uniform_int_distribution<uint32_t>(a, b)(mt19937) of libstdc++ (GCC 11,
// bits/uniform_int_dist.h:246-317): Lemire's nearly divisionless method on 32-bit draws.
DSM_DEV uint32_t uniform_u32(MtState* s, uint32_t a, uint32_t b) {
  const uint64_t urange = (uint64_t)b - (uint64_t)a;
  if (urange == 0xffffffffull) return mt_next(s) + a;
  const uint32_t range = (uint32_t)(urange + 1);
  uint64_t product = (uint64_t)mt_next(s) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)mt_next(s) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32) + a;
}

// ---- the same generator driven by the whole wave ---------------------------------------------------
DSM_DEV uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
DSM_DEV uint32_t mt_mix(uint32_t hi, uint32_t lo) {
  const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// The state regeneration of mt_next by 64 lanes: element k needs the OLD mt[k], mt[k+1] and mt[k+397] (old
// for k < 227, already regenerated for k >= 227), so chunks of 64 consecutive k in ascending order are
// independent inside a chunk (every lane reads before any lane of the chunk writes).
DSM_DEV void mt_twist_wave(uint32_t* mt, int lane) {
  for (int k0 = 0; k0 < 623; k0 += 64) {
    const int k = k0 + lane;
    uint32_t v = 0;
    if (k < 623) v = mt[k < 227 ? k + 397 : k - 227] ^ mt_mix(mt[k], mt[k + 1]);
    wv_sync();
    if (k < 623) mt[k] = v;
    wv_sync();
  }
  if (lane == 0) mt[623] = mt[396] ^ mt_mix(mt[623], mt[0]);
  wv_sync();
}
// state after `target` further mt_next calls, without producing them (uniform over the wave)
DSM_DEV void mt_skip_wave(MtState* s, uint32_t target, int lane) {
  uint32_t pos = (uint32_t)s->mti + target;
  wv_sync();
  while (pos > 624u) {  // a call that finds mti >= 624 regenerates first
    mt_twist_wave(s->mt, lane);
    pos -= 624u;
  }
  if (lane == 0) s->mti = (int)pos;
  wv_sync();
}

// LDS work area of wv_draw_samples
struct WvSampler {
  uint32_t raw[640];  // tempered outputs not yet consumed (<= 6 left over + one regenerated block); the draws of the block being worked
                      // on are replaced IN PLACE by their swap partners (jb = raw + pos) once the block is known to hold no rejection --
                      // a second array for them was 2.5 of the 8.7 KB that bound k_sample<H>'s occupancy
  uint64_t plain[10];  // per 64 trials of the block: the trials whose swaps are independent of each other
};

// RandomSampler::Sample x nb (src/optim/random_sampler.cc:43-62) for K-element samples out of n, by the wave:
// draw d of a trial picks j = uniform_int_distribution(i, n-1) and swaps sidx[i] <-> sidx[j].  The generator
// outputs of a whole block are tempered and mapped through Lemire's multiply-shift by all lanes; the swaps are a
// sequential chain through the index array only from trial to trial, so lane i makes draw i of a trial (below).  A draw that Lemire's
// method would reject (probability (2^32 mod range) / 2^32, ~6e-8 at 256 matches) shifts every later draw:
// the first block that contains one, and everything after it, is replayed serially with the same
// semantics as uniform_u32 above.  smp: [nb][7] samples; de (optional): generator calls after each trial.
template <int K_>
DSM_DEV void wv_draw_samples(MtState* gen, WvSampler* ws, uint32_t* sidx, uint32_t n, int nb, uint32_t* smp, uint32_t* de,
                             int lane, bool force_serial) {
  const int mti0 = gen->mti;
  int have = 624 - mti0;
  if (have < 0) have = 0;
  for (int e = lane; e < have; e += 64) ws->raw[e] = mt_temper(gen->mt[mti0 + e]);
  int pos = 0;
  uint32_t calls = 0;
  int t = 0;
  bool serial = force_serial;
  wv_sync();
  while (t < nb && !serial) {
    if (have < K_) {  // move the left-over to the front, regenerate, temper the new block behind it
      const uint32_t keep = (lane < have) ? ws->raw[pos + lane] : 0u;
      wv_sync();
      if (lane < have) ws->raw[lane] = keep;
      mt_twist_wave(gen->mt, lane);
      for (int e = lane; e < 624; e += 64) ws->raw[have + e] = mt_temper(gen->mt[e]);
      pos = 0;
      have += 624;
      wv_sync();
    }
    int nt = have / K_;
    if (nt > nb - t) nt = nb - t;
    const int nd = nt * K_;
    bool rej = false;
    for (int e = lane; e < nd; e += 64) {
      const uint32_t i = (uint32_t)(e % K_);  // a block starts at a trial boundary
      const uint32_t range = n - i;
      const uint32_t low = (uint32_t)((uint64_t)ws->raw[pos + e] * (uint64_t)range);
      if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        if (low < threshold) rej = true;
      }
    }
    if (__ballot(rej) != 0ull) {  // (the serial replay below reads the block's raw outputs: nothing has been overwritten yet)
      serial = true;
      break;
    }
    uint32_t* const jb = ws->raw + pos;
    for (int e = lane; e < nd; e += 64) {  // every lane replaces the entries it read itself
      const uint32_t i = (uint32_t)(e % K_);
      jb[e] = (uint32_t)(((uint64_t)jb[e] * (uint64_t)(n - i)) >> 32) + i;
    }
    wv_sync();
    // Which trials are "plain" -- every partner outside the head (j >= K_) and no partner named twice -- depends on the
    // draws alone, not on the index array: decided for the whole block by a lane per trial, as ballot masks.  The
    // generator position after each trial goes out on the same occasion.
    for (int c = 0; c < nt; c += 64) {
      const int tt = c + lane;
      bool plain = false;
      if (tt < nt) {
        uint32_t j[K_];
#pragma unroll
        for (int i = 0; i < K_; ++i) j[i] = jb[tt * K_ + i];
        plain = true;
#pragma unroll
        for (int i = 0; i < K_; ++i) plain = plain && j[i] >= (uint32_t)K_;
#pragma unroll
        for (int i = 1; i < K_; ++i) {
#pragma unroll
          for (int i2 = 0; i2 < i; ++i2) plain = plain && j[i] != j[i2];
        }
        if (de) de[t + tt] = calls + (uint32_t)(tt + 1) * K_;
      }
      const uint64_t m = __ballot(plain);
      if (lane == 0) ws->plain[c >> 6] = m;
    }
    wv_sync();
    // The swaps are a sequential chain through the index array, but only from trial to trial: the K_ swaps of a plain
    // trial are independent, so lane i makes draw i -- one partner read, one write of the head entry it replaces, the
    // head entry itself in a register (h).  A trial then costs one LDS round trip instead of K_ times the single lane's
    // instruction stream (measured, lane 0 alone: ~1 200 cycles per 7-point trial).  The LDS queue of a wave is in
    // order, so the next trial's read sees this trial's write.  A trial that is not plain (a few per cent: ~K_^2 / n)
    // takes the one-swap-at-a-time form on lane 0 with the head back in the array.  Either way the state after the
    // trial is that of K_ sequential swaps.
    // Runs of plain trials go through a loop without a branch in it: the execution mask is set once for the run, an iteration
    // is two LDS reads (the next trial's partner, the entry that becomes the head), one LDS write and the store of the sample
    // -- and the write and the store of trial r use the head read in trial r - 1, so the read of trial r is in flight behind
    // them (the first form of this loop tested the plain bit, saved and restored the mask and waited for its own read in
    // every trial: ~25 scalar instructions and a full LDS round trip per trial, the scalar unit of the CU was its limit).
    {
      const bool mine = lane < K_;
      uint32_t h = mine ? sidx[lane] : 0u;
      uint32_t jn = mine ? jb[lane] : 0u;
      for (int c = 0; c < nt; c += 64) {
        const uint64_t pm = ws->plain[c >> 6];
        // (the builtin returns int: through uint32_t, or the low word's bit 31 would be sign-extended over the high word)
        const uint32_t pm_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pm), pm_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pm >> 32));
        const uint64_t pmu = ((uint64_t)pm_hi << 32) | (uint64_t)pm_lo;
        const int lim = nt - c < 64 ? nt - c : 64;
        int u = 0;
        while (u < lim) {
          const uint64_t not_plain = ~(pmu >> u);  // bit r: trial u + r is not plain (the shifted-in zeros end the run at bit 64 - u)
          int run = not_plain ? __builtin_ctzll(not_plain) : 64;
          if (run > lim - u) run = lim - u;
          const int tt = c + u;
          if (run > 0) {
            if (mine) {
              uint32_t* out = smp + (size_t)(t + tt) * 7 + lane;
              const uint32_t* jp = jb + (tt + 1) * K_ + lane;  // (one trial past the block's last is inside raw[]: read, never used)
              uint32_t j = jn;
              jn = *jp;
              jp += K_;
              uint32_t v = sidx[j];
              sidx[j] = h;
              h = v;
              for (int r = 1; r < run; ++r) {
                j = jn;
                jn = *jp;
                jp += K_;
                v = sidx[j];
                sidx[j] = h;
                *out = h;
                out += 7;
                h = v;
              }
              *out = h;
            }
            u += run;
          } else {
            if (mine) {
              jn = jb[(tt + 1) * K_ + lane];
              sidx[lane] = h;
            }
            wv_sync();
            if (lane == 0) {
              uint32_t pj[K_];  // the partners first, all at once: read inside the loop they cost every swap a second LDS round trip
#pragma unroll
              for (int i = 0; i < K_; ++i) pj[i] = jb[tt * K_ + i];
#pragma unroll
              for (int i = 0; i < K_; ++i) {
                const uint32_t a = sidx[i];
                sidx[i] = sidx[pj[i]];
                sidx[pj[i]] = a;
              }
            }
            wv_sync();
            if (mine) {
              h = sidx[lane];
              smp[(size_t)(t + tt) * 7 + lane] = h;
            }
            u += 1;
          }
        }
      }
      if (mine) sidx[lane] = h;
    }
    pos += nd;
    have -= nd;
    calls += (uint32_t)nd;
    t += nt;
    wv_sync();
  }
  if (lane == 0) {
    if (serial && t < nb) {
      gen->mti = 624;  // every output of the current block is in raw[]; mt_next regenerates when raw[] runs dry
      auto next = [&]() -> uint32_t {
        ++calls;
        if (have > 0) {
          --have;
          return ws->raw[pos++];
        }
        return mt_next(gen);
      };
      for (; t < nb; ++t) {
        for (uint32_t i = 0; i < (uint32_t)K_; ++i) {
          const uint32_t range = n - i;
          uint64_t product = (uint64_t)next() * (uint64_t)range;
          uint32_t low = (uint32_t)product;
          if (low < range) {
            const uint32_t threshold = (0u - range) % range;
            while (low < threshold) {
              product = (uint64_t)next() * (uint64_t)range;
              low = (uint32_t)product;
            }
          }
          const uint32_t j = (uint32_t)(product >> 32) + i;
          const uint32_t a = sidx[i];
          sidx[i] = sidx[j];
          sidx[j] = a;
        }
        for (int i = 0; i < K_; ++i) smp[(size_t)t * 7 + i] = sidx[i];
        if (de) de[t] = calls;
      }
      if (have > 0) gen->mti = 624 - have;
    } else {
      gen->mti = 624 - have;
    }
    gen->calls = calls;
  }
  wv_sync();
}

// Per-pair generator record in global memory: [0..623] state, [624] index at the end of the last sampling;
// [640..1263] + [1264] the snapshot taken before the last sampling round; [1265] draws to skip from the
// snapshot, [1266] != 0: the next consumer must resume from the snapshot + skip (an early stop left the
// "current" state ahead of the sequential stream position).
#define PAIR_STATE_WORDS 1280
#define PS_SNAP 640
#define PS_SKIP 1265
#define PS_USE_SNAP 1266

DSM_DEV void generator_load(MtState* sm, uint32_t* st, int lane) {
  const bool use_snap = st[PS_USE_SNAP] != 0;
  const uint32_t* src = use_snap ? st + PS_SNAP : st;
  for (int i = lane; i < 624; i += 64) sm->mt[i] = src[i];
  wv_sync();
  if (lane == 0) sm->mti = (int)src[624];
  wv_sync();
  if (use_snap) {
    mt_skip_wave(sm, st[PS_SKIP], lane);
    if (lane == 0) st[PS_USE_SNAP] = 0;
  }
  wv_sync();
}
DSM_DEV void generator_store(const MtState* sm, uint32_t* dst, int lane) {
  for (int i = lane; i < 624; i += 64) dst[i] = sm->mt[i];
  if (lane == 0) dst[624] = (uint32_t)sm->mti;
}

// ------------------------------------------------------------------------------------ families
enum { FAM_E = 0, FAM_F = 1, FAM_H = 2, FAM_T = 3 };

template <int FAM> struct Fam;
template <> struct Fam<FAM_E> { static constexpr int K = 5, MAXM = 10, LO_MIN = 5, MSZ = 9; };
template <> struct Fam<FAM_F> { static constexpr int K = 7, MAXM = 3, LO_MIN = 8, MSZ = 9; };
template <> struct Fam<FAM_H> { static constexpr int K = 4, MAXM = 1, LO_MIN = 4, MSZ = 9; };
template <> struct Fam<FAM_T> { static constexpr int K = 1, MAXM = 1, LO_MIN = 1, MSZ = 9; };

template <int FAM>
DSM_DEV double fam_residual(const double* M, const double* p) {
  if (FAM == FAM_H) return homography_residual(M, p[0], p[1], p[2], p[3]);
  if (FAM == FAM_T) return translation_residual(M, p[0], p[1], p[2], p[3]);
  return sampson_residual(M, p[0], p[1], p[2], p[3]);
}

template <int FAM>
DSM_DEV int fam_minimal(const double* xs, double* models) {
  if (FAM == FAM_E) return five_point_minimal(xs, models);
  if (FAM == FAM_F) return seven_point(xs, models);
  if (FAM == FAM_H) return homography_four_point_reg(xs, models);
  // TranslationTransformEstimator<2>::Estimate with one point, translation_transform.h:81-104
  const double sx = (0.0 + xs[0]) / 1, sy = (0.0 + xs[1]) / 1, dx = (0.0 + xs[2]) / 1, dy = (0.0 + xs[3]) / 1;
  models[0] = dx - sx;
  models[1] = dy - sy;
  for (int k = 2; k < 9; ++k) models[k] = 0.0;
  return 1;
}

// ------------------------------------------------------------------------------------ per-pair work area
struct PairWork {
  int n;                 // correspondences
  const double* pts;     // n x 4 (x1 y1 x2 y2) of the family's coordinate frame
  double* resid;         // n
  int* inl;              // n: ordered inlier indices of the last compaction
  double* tall;          // >= 18 n + 81 doubles: LO constraint matrix (+ 81 for the transposed small case)
  double* models;        // BATCH * MAXM * 9
  VSmem* sm;
  const uint32_t* nt_table;  // ComputeNumTrials(k, n) for k = 0..n of this family
  int lane;
};

// residuals of all correspondences for model M -> resid[]; returns inlier count (uniform)
template <int FAM>
DSM_DEV int score_model(const PairWork& w, const double* M, double max_residual, bool store) {
  int count = 0;
  for (int base = 0; base < w.n; base += 64) {
    const int i = base + w.lane;
    bool in = false;
    if (i < w.n) {
      const double r = fam_residual<FAM>(M, w.pts + (size_t)i * 4);
      if (store) w.resid[i] = r;
      in = r <= max_residual;
    }
    count += __popcll(__ballot(in));
  }
  return count;
}

// InlierSupportMeasurer::Evaluate's residual_sum (support_measurement.cc:43-48): in index order.
DSM_DEV double ordered_residual_sum(const PairWork& w, double max_residual) {
  wv_sync();
  double s = 0;
  for (int base = 0; base < w.n; base += 64) {
    const int i = base + w.lane;
    const double v = (i < w.n) ? w.resid[i] : 0.0;
    const int cnt = (w.n - base) < 64 ? (w.n - base) : 64;
    for (int k = 0; k < cnt; ++k) {
      const double r = wv_readlane_f64(v, k);
      if (r <= max_residual) s += r;
    }
  }
  return s;
}

// score_model(store) + ordered_residual_sum in one pass: the residuals stay in registers, and the in-order sum walks
// only the inliers of every 64-chunk (ascending lane order = index order; the same additions in the same order as
// the loop over all elements that skips the outliers).  resid[] is still written for the compaction / final mask.
template <int FAM>
DSM_DEV double score_and_sum(const PairWork& w, const double* M, double max_residual, uint32_t* count_out) {
  double s = 0;
  uint32_t count = 0;
  for (int base = 0; base < w.n; base += 64) {
    const int i = base + w.lane;
    bool in = false;
    double r = 0.0;
    if (i < w.n) {
      r = fam_residual<FAM>(M, w.pts + (size_t)i * 4);
      w.resid[i] = r;
      in = r <= max_residual;
    }
    unsigned long long mask = __ballot(in);
    count += (uint32_t)__popcll(mask);
    while (mask) {
      const int k = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      s += wv_readlane_f64(r, k);
    }
  }
  wv_sync();
  *count_out = count;
  return s;
}

// ordered compaction of the inliers of resid[] into inl[]; returns the count
DSM_DEV int compact_inliers(const PairWork& w, double max_residual) {
  int total = 0;
  for (int base = 0; base < w.n; base += 64) {
    const int i = base + w.lane;
    const bool in = (i < w.n) && (w.resid[i] <= max_residual);
    const unsigned long long bal = __ballot(in);
    if (in) w.inl[total + __popcll(bal & ((1ull << w.lane) - 1ull))] = i;
    total += __popcll(bal);
  }
  wv_sync();
  return total;
}

// ------------------------------------------------------------------------------------ local estimators
// Writes up to MAXM models into sm->lo_models; returns their number (uniform).
template <int FAM>
DSM_DEV int fam_local(const PairWork& w, int ninl) {
  VSmem* sm = w.sm;
  const int lane = w.lane;
  const int* inl = w.inl;
  auto idx = [inl](int i) { return inl[i]; };
  if (FAM == FAM_T) {
    // TranslationTransformEstimator<2>::Estimate, translation_transform.h:81-104
    const double* tpts = w.pts;
    double sx = wv_seq_sum(0.0, ninl, lane, [tpts, inl](int i) { return tpts[(size_t)inl[i] * 4 + 0]; });
    double sy = wv_seq_sum(0.0, ninl, lane, [tpts, inl](int i) { return tpts[(size_t)inl[i] * 4 + 1]; });
    double dx = wv_seq_sum(0.0, ninl, lane, [tpts, inl](int i) { return tpts[(size_t)inl[i] * 4 + 2]; });
    double dy = wv_seq_sum(0.0, ninl, lane, [tpts, inl](int i) { return tpts[(size_t)inl[i] * 4 + 3]; });
    sx /= ninl; sy /= ninl;
    dx /= ninl; dy /= ninl;
    if (lane == 0) {
      sm->lo_models[0] = dx - sx;
      sm->lo_models[1] = dy - sy;
      for (int k = 2; k < 9; ++k) sm->lo_models[k] = 0.0;
    }
    wv_sync();
    return 1;
  }
  if (FAM == FAM_E) {
    // EssentialMatrixFivePointEstimator::Estimate with all inliers, essential_matrix.cc:46-150
    const int m = ninl;
    double* Q = w.tall;
    for (int i = lane; i < m; i += 64) {
      const double* p = w.pts + (size_t)inl[i] * 4;
      const double x1_0 = p[0], x1_1 = p[1], x2_0 = p[2], x2_1 = p[3];
      Q[(size_t)0 * m + i] = x1_0 * x2_0; Q[(size_t)1 * m + i] = x1_1 * x2_0; Q[(size_t)2 * m + i] = x2_0;
      Q[(size_t)3 * m + i] = x1_0 * x2_1; Q[(size_t)4 * m + i] = x1_1 * x2_1; Q[(size_t)5 * m + i] = x2_1;
      Q[(size_t)6 * m + i] = x1_0; Q[(size_t)7 * m + i] = x1_1; Q[(size_t)8 * m + i] = 1;
    }
    wv_sync();
    wv_svd_V_mx9(Q, Q + (size_t)9 * m, m, &sm->svd, sm->sv, lane);
    {
      LSEC_BEGIN();
      if (lane < 36) sm->g5.Eb[lane] = sm->svd.V[(5 + (lane & 3)) * 9 + (lane >> 2)];  // Eb[r*4 + c] = V(r, 5 + c)
      wv_sync();
      if (lane < 16) {
        const int nm = g5_five_point_finish(&sm->g5, lane, 0);
        for (int e = lane; e < nm * 9; e += 16) sm->lo_models[e] = sm->g5.models[e];
        if (lane == 0) sm->ibcast[0] = nm;
      }
      LSEC_END(7);
    }
    wv_sync();
    return sm->ibcast[0];
  }
  // F (8-point) and H share the normalisation prologue.
  double n1[3], n2[3];
  wv_center_and_normalize(w.pts, 0, ninl, idx, lane, &n1[0], &n1[1], &n1[2]);
  wv_center_and_normalize(w.pts, 1, ninl, idx, lane, &n2[0], &n2[1], &n2[2]);
  if (FAM == FAM_F) {
    // FundamentalMatrixEightPointEstimator::Estimate, fundamental_matrix.cc:150-192
    const int m = ninl;
    double* C = w.tall;
    for (int i = lane; i < m; i += 64) {
      const double* p = w.pts + (size_t)inl[i] * 4;
      double a0, a1, b0, b1;
      apply_norm(n1[0], n1[1], n1[2], p[0], p[1], &a0, &a1);
      apply_norm(n2[0], n2[1], n2[2], p[2], p[3], &b0, &b1);
      const double h[3] = {a0, a1, 1.0};
      for (int k = 0; k < 3; ++k) {
        C[(size_t)k * m + i] = h[k] * b0;
        C[(size_t)(3 + k) * m + i] = h[k] * b1;
        C[(size_t)(6 + k) * m + i] = h[k];
      }
    }
    wv_sync();
    wv_svd_V_mx9(C, C + (size_t)9 * m, m, &sm->svd, sm->sv, lane);
    if (lane == 0) {
      LSEC_BEGIN();
      double nv[9];
      for (int k = 0; k < 9; ++k) nv[k] = sm->svd.V[8 * 9 + k];
      eight_point_finish(nv, n1, n2, sm->lo_models);
      LSEC_END(6);
    }
    wv_sync();
    return 1;
  }
  {
    // HomographyMatrixEstimator::Estimate, homography_matrix.cc:44-92
    const int N = ninl, m = 2 * ninl;
    double* A = w.tall;
    for (int e = lane; e < 9 * m; e += 64) A[e] = 0.0;
    wv_sync();
    for (int i = lane; i < N; i += 64) {
      const double* p = w.pts + (size_t)inl[i] * 4;
      double s_0, s_1, d_0, d_1;
      apply_norm(n1[0], n1[1], n1[2], p[0], p[1], &s_0, &s_1);
      apply_norm(n2[0], n2[1], n2[2], p[2], p[3], &d_0, &d_1);
      const int j = N + i;
      A[(size_t)0 * m + i] = -s_0; A[(size_t)1 * m + i] = -s_1; A[(size_t)2 * m + i] = -1;
      A[(size_t)6 * m + i] = s_0 * d_0; A[(size_t)7 * m + i] = s_1 * d_0; A[(size_t)8 * m + i] = d_0;
      A[(size_t)3 * m + j] = -s_0; A[(size_t)4 * m + j] = -s_1; A[(size_t)5 * m + j] = -1;
      A[(size_t)6 * m + j] = s_0 * d_1; A[(size_t)7 * m + j] = s_1 * d_1; A[(size_t)8 * m + j] = d_1;
    }
    wv_sync();
    wv_svd_V_mx9(A, A + (size_t)9 * m, m, &sm->svd, sm->sv, lane);
    if (lane == 0) {
      double nv[9];
      for (int k = 0; k < 9; ++k) nv[k] = sm->svd.V[8 * 9 + k];
      homography_finish(nv, n1, n2, sm->lo_models);
    }
    wv_sync();
    return 1;
  }
}

// ------------------------------------------------------------------------------------ LO-RANSAC
struct RansacOpt {
  double max_error;
  uint32_t min_num_trials;
  uint32_t max_num_trials;  // already min(options.max_num_trials, ctor's dyn_max), ransac.h:135-148
};

// LORANSAC<Estimator, LocalEstimator>::Estimate, loransac.h:91-233.  The final inlier mask stays in
// w.resid (residuals of the returned model) for the caller.  sidx: LDS index array (n entries).
template <int FAM>
DSM_DEV void lo_ransac(const PairWork& w, const RansacOpt& opt, uint32_t* sidx, RansacReport* rep) {
  typedef Fam<FAM> F;
  VSmem* sm = w.sm;
  const int lane = w.lane;
  const int n = w.n;
  rep->success = false;
  rep->num_trials = 0;
  rep->num_models = 0;
  rep->num_inliers = 0;
  rep->residual_sum = DBL_MAX;
  for (int k = 0; k < 9; ++k) rep->model[k] = 0.0;
  if (n < F::K) return;

  uint32_t best_n = 0;
  double best_sum = DBL_MAX;
  double best_model[9];
  for (int k = 0; k < 9; ++k) best_model[k] = 0.0;
  bool abort = false;
  const double max_residual = opt.max_error * opt.max_error;
  for (int i = lane; i < n; i += 64) sidx[i] = (uint32_t)i;  // sampler.Initialize: iota
  wv_sync();
  const uint32_t max_num_trials = opt.max_num_trials;
  uint32_t dyn_max_num_trials = max_num_trials;
  uint32_t trial = 0;  // report.num_trials

  while (trial < max_num_trials && !abort) {
    const int nb = (int)((max_num_trials - trial) < (uint32_t)BATCH ? (max_num_trials - trial) : (uint32_t)BATCH);
    // ---- backup of the generator, then nb samples by lane 0 (RandomSampler::Sample)
    for (int i = lane; i < 624; i += 64) sm->mt_bak[i] = sm->gen.mt[i];
    if (lane == 0) {
      sm->mti_bak = sm->gen.mti;
      sm->gen.calls = 0;
      const uint32_t last_idx = (uint32_t)(n - 1);
      for (int t = 0; t < nb; ++t) {
        for (uint32_t i = 0; i < (uint32_t)F::K; ++i) {
          const uint32_t j = uniform_u32(&sm->gen, i, last_idx);
          const uint32_t a = sidx[i];
          sidx[i] = sidx[j];
          sidx[j] = a;
        }
        for (int i = 0; i < F::K; ++i) sm->sample[t * 7 + i] = (int)sidx[i];
        sm->draws_end[t] = sm->gen.calls;
      }
    }
    wv_sync();
    // ---- one minimal problem per lane
    {
      int nm = 0;
      if (lane < nb) {
        double xs[F::K * 4];
        for (int i = 0; i < F::K; ++i) {
          const double* p = w.pts + (size_t)sm->sample[lane * 7 + i] * 4;
          xs[i * 4 + 0] = p[0]; xs[i * 4 + 1] = p[1]; xs[i * 4 + 2] = p[2]; xs[i * 4 + 3] = p[3];
        }
        double mloc[F::MAXM * 9];
        nm = fam_minimal<FAM>(xs, mloc);
        double* dst = w.models + (size_t)lane * F::MAXM * 9;
        for (int k = 0; k < nm * 9; ++k) dst[k] = mloc[k];
      }
      if (lane < BATCH) sm->nmodels[lane] = nm;
    }
    wv_sync();
    // ---- wavefront-per-hypothesis scoring of every model of the batch
    for (int t = 0; t < nb; ++t) {
      const int nm = sm->nmodels[t];
      for (int m = 0; m < nm; ++m) {
        const int c = score_model<FAM>(w, w.models + ((size_t)t * F::MAXM + m) * 9, max_residual, false);
        if (lane == 0) sm->counts[t * 10 + m] = c;
      }
    }
    wv_sync();
    // ---- in-order replay
    int t_stop = nb - 1;
    for (int t = 0; t < nb && !abort; ++t, ++trial) {
      const int nm = sm->nmodels[t];
      for (int m = 0; m < nm; ++m) {
        rep->num_models += 1;
        const uint32_t cnt = (uint32_t)sm->counts[t * 10 + m];
        const double* M = w.models + ((size_t)t * F::MAXM + m) * 9;
        bool better = false;
        double sum = 0.0;
        if (cnt > best_n || cnt == best_n) {
          // candidate: residuals + in-order residual_sum (Compare, support_measurement.cc:52-60)
          score_model<FAM>(w, M, max_residual, true);
          sum = ordered_residual_sum(w, max_residual);
          better = (cnt > best_n) || (cnt == best_n && sum < best_sum);
        }
        if (better) {
          best_n = cnt;
          best_sum = sum;
          for (int k = 0; k < 9; ++k) best_model[k] = M[k];
          if (cnt > (uint32_t)F::K && cnt >= (uint32_t)F::LO_MIN) {
            const int ninl = compact_inliers(w, max_residual);
            const int nlo = fam_local<FAM>(w, ninl);
            for (int l = 0; l < nlo; ++l) {
              rep->num_models += 1;
              const uint32_t lc = (uint32_t)score_model<FAM>(w, sm->lo_models + l * 9, max_residual, true);
              const double lsum = ordered_residual_sum(w, max_residual);
              if (lc > best_n || (lc == best_n && lsum < best_sum)) {
                best_n = lc;
                best_sum = lsum;
                for (int k = 0; k < 9; ++k) best_model[k] = sm->lo_models[l * 9 + k];
              }
            }
          }
          dyn_max_num_trials = w.nt_table[best_n];
        }
        if (trial >= dyn_max_num_trials && trial >= opt.min_num_trials) {
          abort = true;
          break;
        }
      }
      if (abort) {
        t_stop = t;
        break;  // `trial` stays at the aborting trial
      }
    }
    if (abort) {
      // loransac.h:129-134: one more loop increment, and +1 inside the loop if it is entered again
      trial = (trial + 1 < max_num_trials) ? trial + 2 : trial + 1;
      // rewind the generator to just after the sample of trial t_stop
      if (t_stop != nb - 1) {
        wv_sync();
        for (int i = lane; i < 624; i += 64) sm->gen.mt[i] = sm->mt_bak[i];
        wv_sync();
        if (lane == 0) {
          sm->gen.mti = sm->mti_bak;
          const uint32_t target = sm->draws_end[t_stop];
          sm->gen.calls = 0;
          while (sm->gen.calls < target) (void)mt_next(&sm->gen);
        }
        wv_sync();
      }
    }
  }
  rep->num_trials = trial;
  rep->num_inliers = best_n;
  rep->residual_sum = best_sum;
  for (int k = 0; k < 9; ++k) rep->model[k] = best_model[k];
  if (best_n < (uint32_t)F::K) return;
  rep->success = true;
  score_model<FAM>(w, best_model, max_residual, true);  // residuals of the final model -> mask
  wv_sync();
}

// ------------------------------------------------------------------------------------ cameras
// Camera::ImageToWorld / ImageToWorldThreshold / CalibrationMatrix for all eleven models: verify_camera.h

// ------------------------------------------------------------------------------------ relative pose
// TriangulatePoint (triangulation.cc:39-52) with P1 = [I | 0] and P2 = [R | t]; returns false never.
DSM_DEVN void triangulate_point(const double* R, const double* t, double p1x, double p1y, double p2x, double p2y, double* X) {
  double A[16];
  const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const double P2[12] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2]};
  for (int c = 0; c < 4; ++c) {
    A[0 * 4 + c] = p1x * P1[8 + c] - P1[0 + c];
    A[1 * 4 + c] = p1y * P1[8 + c] - P1[4 + c];
    A[2 * 4 + c] = p2x * P2[8 + c] - P2[0 + c];
    A[3 * 4 + c] = p2y * P2[8 + c] - P2[4 + c];
  }
  double V[16], sv[4];
  pr_jacobi_svd_square_V<4>(A, V, sv);  // registers only
  const double w = V[3 * 4 + 3];
  X[0] = V[3 * 4 + 0] / w;
  X[1] = V[3 * 4 + 1] / w;
  X[2] = V[3 * 4 + 2] / w;
}

// CheckCheirality (pose.cc:225-247): lanes stride over the inlier correspondences; points in front
// of both cameras are appended, in index order, to pts3d (3 doubles each).  Returns their number.
DSM_DEV int check_cheirality(const double* R, const double* t, const double* ipts, int n, double* pts3d, int lane) {
  const double kMinDepth = DBL_EPSILON;
  double rt[3];
  for (int i = 0; i < 3; ++i) rt[i] = R[0 * 3 + i] * t[0] + R[1 * 3 + i] * t[1] + R[2 * 3 + i] * t[2];
  const double max_depth = 1000.0f * sqrt(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]);
  const double n1 = sqrt(0.0 * 0.0 + 0.0 * 0.0 + 1.0 * 1.0);
  const double n2 = sqrt(R[2] * R[2] + R[5] * R[5] + R[8] * R[8]);
  int total = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    bool ok = false;
    double X[3] = {0, 0, 0};
    if (i < n) {
      const double* p = ipts + (size_t)i * 4;
      triangulate_point(R, t, p[0], p[1], p[2], p[3], X);
      const double d1 = (0.0 * X[0] + 0.0 * X[1] + 1.0 * X[2] + 0.0 * 1.0) * n1;
      if (d1 > kMinDepth && d1 < max_depth) {
        const double d2 = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2] * 1.0) * n2;
        if (d2 > kMinDepth && d2 < max_depth) ok = true;
      }
    }
    const unsigned long long bal = __ballot(ok);
    if (ok) {
      double* dst = pts3d + (size_t)(total + __popcll(bal & ((1ull << lane) - 1ull))) * 3;
      dst[0] = X[0];
      dst[1] = X[1];
      dst[2] = X[2];
    }
    total += __popcll(bal);
  }
  wv_sync();
  return total;
}

DSM_DEV void normalized3(const double* a, double* o) {
  const double n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (n2 > 0) {
    const double n = sqrt(n2);
    for (int i = 0; i < 3; ++i) o[i] = a[i] / n;
  } else {
    for (int i = 0; i < 3; ++i) o[i] = a[i];
  }
}

// Quaterniond(R) -> (w, x, y, z), Eigen/src/Geometry/Quaternion.h (pose.cc:70-73)
DSM_DEV void rotation_to_quaternion(const double* R, double* q) {
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}

DSM_DEV double opp_minor(const double* M, int row, int col) {
  const int col1 = col == 0 ? 1 : 0;
  const int col2 = col == 2 ? 1 : 2;
  const int row1 = row == 0 ? 1 : 0;
  const int row2 = row == 2 ? 1 : 2;
  return (M[row1 * 3 + col2] * M[row2 * 3 + col1] - M[row1 * 3 + col1] * M[row2 * 3 + col2]);
}
DSM_DEV int sign_of(double v) { return (0.0 < v) - (v < 0.0); }

// DecomposeHomographyMatrix, base/homography_matrix.cc:65-165 (uniform; every lane computes it)
DSM_DEVN int decompose_homography(const double* H, const double* K1, const double* K2, double* Rs, double* ts) {
  double K2inv[9], T[9], Hn[9];
  m3_inverse(K2, K2inv);
  m3_mul(K2inv, H, T);
  m3_mul(T, K1, Hn);
  double V[9], sv[3];
  pl_jacobi_svd_square<3, false>(Hn, nullptr, V, sv);
  const double s1 = sv[1];
  for (int i = 0; i < 9; ++i) Hn[i] /= s1;
  double Hnt[9], S[9];
  m3_transpose(Hn, Hnt);
  m3_mul(Hnt, Hn, S);
  S[0] -= 1; S[4] -= 1; S[8] -= 1;
  double inf_norm = 0;
  for (int i = 0; i < 9; ++i) inf_norm = fabs(S[i]) > inf_norm ? fabs(S[i]) : inf_norm;
  if (inf_norm < 1e-3) {
    for (int i = 0; i < 9; ++i) Rs[i] = Hn[i];
    ts[0] = ts[1] = ts[2] = 0;
    return 1;
  }
  const double M00 = opp_minor(S, 0, 0), M11 = opp_minor(S, 1, 1), M22 = opp_minor(S, 2, 2);
  const double rtM00 = sqrt(M00), rtM11 = sqrt(M11), rtM22 = sqrt(M22);
  const double M01 = opp_minor(S, 0, 1), M12 = opp_minor(S, 1, 2), M02 = opp_minor(S, 0, 2);
  const int e12 = sign_of(M12), e02 = sign_of(M02), e01 = sign_of(M01);
  const double nS[3] = {fabs(S[0]), fabs(S[4]), fabs(S[8])};
  int idx = 0;
  if (nS[1] > nS[idx]) idx = 1;
  if (nS[2] > nS[idx]) idx = 2;
  double np1[3], np2[3];
  if (idx == 0) {
    np1[0] = S[0]; np2[0] = S[0];
    np1[1] = S[1] + rtM22; np2[1] = S[1] - rtM22;
    np1[2] = S[2] + e12 * rtM11; np2[2] = S[2] - e12 * rtM11;
  } else if (idx == 1) {
    np1[0] = S[1] + rtM22; np2[0] = S[1] - rtM22;
    np1[1] = S[4]; np2[1] = S[4];
    np1[2] = S[5] - e02 * rtM00; np2[2] = S[5] + e02 * rtM00;
  } else {
    np1[0] = S[2] + e01 * rtM11; np2[0] = S[2] - e01 * rtM11;
    np1[1] = S[5] + rtM00; np2[1] = S[5] - rtM00;
    np1[2] = S[8]; np2[2] = S[8];
  }
  const double traceS = S[0] + S[4] + S[8];
  const double v = 2.0 * sqrt(1.0 + traceS - M00 - M11 - M22);
  const double ESii = sign_of(S[idx * 3 + idx]);
  const double r_2 = 2 + traceS + v;
  const double nt_2 = 2 + traceS - v;
  const double r = sqrt(r_2);
  const double n_t = sqrt(nt_2);
  double n1[3], n2[3];
  normalized3(np1, n1);
  normalized3(np2, n2);
  const double half_nt = 0.5 * n_t;
  const double esii_t_r = ESii * r;
  double t1_star[3], t2_star[3];
  for (int i = 0; i < 3; ++i) {
    t1_star[i] = half_nt * (esii_t_r * n2[i] - n_t * n1[i]);
    t2_star[i] = half_nt * (esii_t_r * n1[i] - n_t * n2[i]);
  }
  double Mx[9], R1[9], R2[9];
  const double s = 2.0 / v;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Mx[i * 3 + j] = (i == j ? 1.0 : 0.0) - (s * t1_star[i]) * n1[j];
  m3_mul(Hn, Mx, R1);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Mx[i * 3 + j] = (i == j ? 1.0 : 0.0) - (s * t2_star[i]) * n2[j];
  m3_mul(Hn, Mx, R2);
  double t1[3], t2[3];
  for (int i = 0; i < 3; ++i) {
    t1[i] = R1[i * 3 + 0] * t1_star[0] + R1[i * 3 + 1] * t1_star[1] + R1[i * 3 + 2] * t1_star[2];
    t2[i] = R2[i * 3 + 0] * t2_star[0] + R2[i * 3 + 1] * t2_star[1] + R2[i * 3 + 2] * t2_star[2];
  }
  for (int i = 0; i < 9; ++i) {
    Rs[0 * 9 + i] = R1[i];
    Rs[1 * 9 + i] = R1[i];
    Rs[2 * 9 + i] = R2[i];
    Rs[3 * 9 + i] = R2[i];
  }
  for (int i = 0; i < 3; ++i) {
    ts[0 * 3 + i] = t1[i];
    ts[1 * 3 + i] = -t1[i];
    ts[2 * 3 + i] = t2[i];
    ts[3 * 3 + i] = -t2[i];
  }
  return 4;
}

// ------------------------------------------------------------------------------------ kernels
// The verification of a pair list runs as five launches over the same grid-stride pair loop so that
// each phase gets its own register allocation (the 5-point solver would otherwise pin the whole
// pipeline at one wave per SIMD):
//   k_verify_prep      gather matched points (+ ImageToWorld), seed the pair's MT19937
//   k_ransac<E|F|H>    one LO-RANSAC family each, in the stream order E -> F -> H; the generator
//                      state travels between the launches through pair_state
//   k_verify_final     decision tree, inlier extraction, watermark, relative pose, output
// Per resident workgroup scratch (doubles), n = n_max of the launch:
//   resid [n]  tall [18n + 96]  models [64*10*9]  pts3d_a [3n]  pts3d_b [3n]  ipts [4n]  inl (int) [n]
__host__ __device__ inline size_t verify_scratch_doubles(size_t n) {
  return n + (18 * n + 96) + (size_t)BATCH * 10 * 9 + 3 * n + 3 * n + 4 * n + (n + 1) / 2 + 8;
}

// Dynamic work hand-out of the persistent wave-per-pair kernels.  One atomicAdd per item on ONE counter serialises at the
// L2 (~30 ns each): with 10^5 light items per launch that WAS the launch (k_replay_lo: 54 ns per pair).  A wave takes
// `gr` consecutive items per atomic instead -- 4 when the grid has at least 8 items per wave to balance with, else 1.
struct WorkGrab {
  uint32_t next = 0, left = 0;
};
DSM_DEV uint32_t work_grain(uint32_t n_items) { return n_items >= 8u * gridDim.x ? 4u : 1u; }
DSM_DEV uint32_t grab_item(WorkGrab& g, uint32_t* counter, uint32_t* s_slot, int lane, uint32_t gr) {
  if (g.left == 0) {
    if (lane == 0) *s_slot = atomicAdd(counter, gr);
    __syncthreads();
    g.next = *s_slot;
    g.left = gr;
  }
  g.left--;
  return g.next++;
}

// The hand-out counter split 64 ways (round 6).  A device-scope atomicAdd is served at ~11.4 ns PER ADDRESS whoever issues it and
// whether or not the value is used (tools/exp/atomic_rate.hip on MI355X: 131 072 of them on one counter from 4 096 waves 1.49 ms,
// on 64 counters 128 bytes apart 0.035 ms; the eight XCDs' L2s are not coherent with each other, so a device-scope atomic is
// resolved behind them), and the replay scans issued 1.3 - 1.5 of them per visited pair on ONE 128-byte line (hand-out, queue append,
// "still active"): 124 750 pairs = 2.3 ms, which WAS the launch (profiles/r06_replay_dispatches_before.txt).  Here segment s of the
// work list -- items [s * seg_len, (s + 1) * seg_len) -- has its own counter on its own line; a wave starts on segment
// blockIdx.x % 64 and, when that is used up, moves on to the next segment that still has items (found with one plain load per lane,
// not with atomics).  Returns an item index, possibly >= n_items inside the last grain (the caller skips it), or GRAB_DONE.
#define GRAB_SEGS 64
#define GRAB_STRIDE 32  // words: one counter per 128-byte line
#define GRAB_AREA_WORDS (GRAB_SEGS * GRAB_STRIDE)
#define GRAB_DONE 0xffffffffu
#define HYP_T_BITS 11  // an entry of the hypothesis list (hyp_of_lane): pl << 11 | t (t < batch <= 2048; the planner keeps a chunk below 2^21 pairs)
// the four areas of a lane's counter block (capi.hip lays them out so that every phase still costs ONE fill), relative to
// VerifyParams::active_count (the classic 32 words): k_sample's in front of it, the replay's (grab_ctr), k_lo_prepare's and
// k_verify_final's behind it
#define GRAB_SAMPLE(p) ((p).active_count - GRAB_AREA_WORDS)
#define GRAB_REPLAY(p) ((p).active_count + 32)
#define GRAB_PREPARE(p) ((p).active_count + 32 + GRAB_AREA_WORDS)
#define GRAB_FINAL(p) ((p).active_count + 32 + 2 * GRAB_AREA_WORDS)
struct SegGrab {
  uint32_t next = 0, left = 0, seg = 0xffffffffu;
};
DSM_DEV uint32_t grab_seg(SegGrab& g, uint32_t* ctrs, uint32_t n_items, int lane, uint32_t gr) {
  if (g.left == 0) {
    const uint32_t seg_len = ((n_items + GRAB_SEGS * gr - 1) / (GRAB_SEGS * gr)) * gr;  // a multiple of gr
    uint32_t s = g.seg == 0xffffffffu ? blockIdx.x % GRAB_SEGS : g.seg;
    for (;;) {
      uint32_t v = 0;
      if (lane == 0) v = atomicAdd(ctrs + s * GRAB_STRIDE, gr);
      v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
      const uint32_t start = s * seg_len + v;
      if (v < seg_len && start < n_items) {
        g.next = start;
        g.left = gr;
        g.seg = s;
        break;
      }
      const uint32_t c = __hip_atomic_load(ctrs + lane * GRAB_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool has = c < seg_len && (uint32_t)lane * seg_len + c < n_items;
      const unsigned long long m = __ballot(has);
      if (!m) {
        g.seg = s;
        return GRAB_DONE;
      }
      const unsigned long long above = s < 63u ? ((m >> (s + 1u)) << (s + 1u)) : 0ull;  // the next one after s, cyclically
      s = (uint32_t)(__ffsll((long long)(above ? above : m)) - 1);
    }
  }
  g.left--;
  return g.next++;
}

struct WgScratch {
  double *resid, *tall, *models, *pts3d_a, *pts3d_b, *ipts;
  int* inl;
};
DSM_DEV WgScratch wg_scratch(const VerifyParams& p) {
  const size_t n = p.n_max;
  WgScratch w;
  double* base = p.scratch + (size_t)blockIdx.x * verify_scratch_doubles(n);
  w.resid = base;
  w.tall = w.resid + n;
  w.models = w.tall + 18 * n + 96;
  w.pts3d_a = w.models + (size_t)BATCH * 10 * 9;
  w.pts3d_b = w.pts3d_a + 3 * n;
  w.ipts = w.pts3d_b + 3 * n;
  w.inl = reinterpret_cast<int*>(w.ipts + 4 * n);
  return w;
}

__global__ __launch_bounds__(64) void k_verify_prep(const VerifyParams p) {
  __shared__ MtState sm;
  const int lane = threadIdx.x;
  for (uint32_t pl = blockIdx.x; pl < p.n_chunk; pl += gridDim.x) {
    const uint32_t pi = p.pair0 + pl;
    const uint32_t im1 = p.pairs[2 * pi], im2 = p.pairs[2 * pi + 1];
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    if ((uint64_t)n < p.opt.min_num_inliers) {
      // DEGENERATE at once (two_view_geometry.cc:298-301): no family runs.  Its states must say so -- the buffer may
      // have just grown over memory that holds anything (found by tools/fuzz_stage.py: a pair list longer than any
      // before it on the context, k_sample taking garbage for an active pair)
      if (p.fam_state != nullptr && lane < 3) {
        FamState fs;
        memset(&fs, 0, sizeof(fs));
        fs.rep.residual_sum = DBL_MAX;
        p.fam_state[(size_t)pi * 3 + lane] = fs;
        p.reports[(size_t)pi * 3 + lane] = fs.rep;
      }
      continue;
    }
    const uint32_t* matches = p.matches + 2 * moff;
    const dsm_camera cam1 = p.cams[im1], cam2 = p.cams[im2];
    const bool calibrated = cam1.has_prior_focal_length && cam2.has_prior_focal_length;
    const double* kp1 = p.kp + (size_t)p.img_row0[im1] * 2;
    const double* kp2 = p.kp + (size_t)p.img_row0[im2] * 2;
    double* pts_px = p.pts_px + 4 * moff;
    double* pts_norm = p.pts_norm + 4 * moff;
    for (int i = lane; i < n; i += 64) {
      const uint32_t i1 = matches[2 * i], i2 = matches[2 * i + 1];
      const double x1 = kp1[2 * (size_t)i1], y1 = kp1[2 * (size_t)i1 + 1];
      const double x2 = kp2[2 * (size_t)i2], y2 = kp2[2 * (size_t)i2 + 1];
      pts_px[4 * i + 0] = x1; pts_px[4 * i + 1] = y1; pts_px[4 * i + 2] = x2; pts_px[4 * i + 3] = y2;
      if (calibrated) {
        double u1, v1, u2, v2;
        image_to_world(cam1, x1, y1, &u1, &v1);
        image_to_world(cam2, x2, y2, &u2, &v2);
        pts_norm[4 * i + 0] = u1; pts_norm[4 * i + 1] = v1; pts_norm[4 * i + 2] = u2; pts_norm[4 * i + 3] = v2;
      }
    }
    wv_sync();
    uint32_t* st = p.pair_state + (size_t)pi * PAIR_STATE_WORDS;
    if (p.reseed) {  // a later pass of EstimateMultiple continues the pair's stream instead
      if (lane == 0) mt_seed(&sm, p.seeds[pi]);
      wv_sync();
      generator_store(&sm, st, lane);
      if (lane == 0) {
        st[PS_USE_SNAP] = 0;
        st[PS_SKIP] = 0;
      }
    }
    if (p.fam_state != nullptr && lane < 3) {  // phase-split pipeline: initial family states
      const int K[3] = {5, 7, 4};
      FamState fs;
      fs.rep.success = false;
      fs.rep.num_trials = 0;
      fs.rep.num_models = 0;
      fs.rep.num_inliers = 0;
      fs.rep.residual_sum = DBL_MAX;
      for (int k = 0; k < 9; ++k) fs.rep.model[k] = 0.0;
      fs.dyn_max = p.max_trials[lane];
      fs.active = (n >= K[lane] && (lane != FAM_E || calibrated) && p.max_trials[lane] > 0) ? 1u : 0u;
      fs.rounds = 0;
      fs.nb = 0;
      fs.t_pos = fs.m_pos = fs.lo_wait = fs.lo_ninl = fs.lo_nm = fs.pad = 0;
      p.fam_state[(size_t)pi * 3 + lane] = fs;
      // a family that never runs (fewer matches than its minimal sample, E without prior focal lengths) must still
      // hand k_verify_final the report LORANSAC::Estimate returns at once (loransac.h:97-103), not what an earlier call
      // left in this pair's slot
      p.reports[(size_t)pi * 3 + lane] = fs.rep;
    }
    wv_sync();
  }
}

#ifdef DSM_CHECK_BUILD  // cross-check schedule: libdagsfm_mi355x_check.so only
template <int FAM>
__global__ __launch_bounds__(64, (FAM == FAM_H ? 2 : (FAM == FAM_F ? 3 : 4))) void k_ransac(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  VSmem* sm = reinterpret_cast<VSmem*>(smem_raw);
  uint32_t* sidx = reinterpret_cast<uint32_t*>(smem_raw + ((sizeof(VSmem) + 15) / 16) * 16);
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  for (uint32_t pl = blockIdx.x; pl < p.n_chunk; pl += gridDim.x) {
    const uint32_t pi = p.pair0 + pl;
    wv_sync();
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    if ((uint64_t)n < p.opt.min_num_inliers) continue;
    const uint32_t im1 = p.pairs[2 * pi], im2 = p.pairs[2 * pi + 1];
    const dsm_camera& cam1 = p.cams[im1];
    const dsm_camera& cam2 = p.cams[im2];
    const bool calibrated = cam1.has_prior_focal_length && cam2.has_prior_focal_length;
    if (FAM == FAM_E && !calibrated) continue;
    uint32_t* st = p.pair_state + (size_t)pi * PAIR_STATE_WORDS;
    generator_load(&sm->gen, st, lane);
    PairWork w;
    w.n = n;
    w.resid = ws.resid;
    w.inl = ws.inl;
    w.tall = ws.tall;
    w.models = ws.models;
    w.sm = sm;
    w.lane = lane;
    w.pts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
    w.nt_table = p.nt_table + p.nt_off[n] + (size_t)FAM * (size_t)(n + 1);
    RansacOpt ro;
    ro.max_error = p.opt.max_error;
    if (FAM == FAM_E)  // two_view_geometry.cc:319-323
      ro.max_error = (image_to_world_threshold(cam1, p.opt.max_error) + image_to_world_threshold(cam2, p.opt.max_error)) / 2;
    ro.min_num_trials = (uint32_t)p.opt.min_num_trials;
    ro.max_num_trials = p.max_trials[FAM];
    RansacReport rep;
    lo_ransac<FAM>(w, ro, sidx, &rep);
    const double mr = ro.max_error * ro.max_error;
    unsigned char* mask = p.masks + (size_t)FAM * p.mask_stride + moff;
    if (rep.success)
      for (int i = lane; i < n; i += 64) mask[i] = ws.resid[i] <= mr;
    if (lane == 0) p.reports[(size_t)pi * 3 + FAM] = rep;
    wv_sync();
    generator_store(&sm->gen, st, lane);
  }
}

#endif  // DSM_CHECK_BUILD
// WAVES = waves per SIMD the register allocation aims at.  Since the candidate poses moved out (k_final_pose), both
// instances need 254 VGPRs; <2> spills 12 of them and runs two waves per SIMD, <1> spills none and runs one.  <2> is the
// product path: verification of config 2 349 -> 341 ms (profiles/r03_check_schedules.txt); DSM_FINAL_WAVES=1 selects the
// other, and tools/check_schedules.py compares the two on all 124 750 pairs of config 2 (byte-identical).
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_verify_final(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  VSmem* sm = reinterpret_cast<VSmem*>(smem_raw);
  uint32_t* sidx = reinterpret_cast<uint32_t*>(smem_raw + ((sizeof(VSmem) + 15) / 16) * 16);
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  double* resid = ws.resid;
  int* inl = ws.inl;
  double* pts3d_a = ws.pts3d_a;

  SegGrab wgrab;
  const uint32_t grain = work_grain(p.n_chunk);
  uint32_t pl_static = blockIdx.x;
  for (;;) {
    wv_sync();
    uint32_t pl;
    if (p.final_list != nullptr) {  // second visit of the pairs that waited for a translation table
      if (pl_static >= p.n_final) break;
      pl = p.final_list[pl_static] - p.pair0;
      pl_static += gridDim.x;
    } else if (p.active_count != nullptr) {  // pipeline: pairs handed out dynamically (segmented work counters)
      pl = grab_seg(wgrab, GRAB_FINAL(p), p.n_chunk, threadIdx.x, grain);
      if (pl == GRAB_DONE) break;
      if (pl >= p.n_chunk) continue;
    } else {
      pl = pl_static;
      pl_static += gridDim.x;
    }
    if (p.final_list == nullptr && pl >= p.n_chunk) break;
    const uint32_t pi = p.pair0 + pl;
    const uint32_t im1 = p.pairs[2 * pi], im2 = p.pairs[2 * pi + 1];
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    const uint32_t* matches = p.matches + 2 * moff;
    dsm_two_view_geometry* out = p.tvg + pi;
    uint32_t* out_inl = p.inlier_matches + 2 * moff;
    const dsm_camera cam1 = p.cams[im1], cam2 = p.cams[im2];
    const dsm_two_view_options& o = p.opt;
    const double* pts_px = p.pts_px + 4 * moff;
    const double* pts_norm = p.pts_norm + 4 * moff;

    // result defaults: TwoViewGeometry(), two_view_geometry.h:159-166
    int config = DSM_CONFIG_UNDEFINED;
    double Em[9], Fm[9], Hm[9], qvec[4] = {0, 0, 0, 0}, tvec[3] = {0, 0, 0}, tri_angle = 0;
    for (int k = 0; k < 9; ++k) Em[k] = Fm[k] = Hm[k] = 0.0;
    uint32_t ntr[4] = {0, 0, 0, 0}, nmo[4] = {0, 0, 0, 0};
    uint32_t num_inliers = 0;
    bool have_mask = false;
    bool gen_loaded = false;
    bool pose_pending = false;  // the pair has a pose job: k_final_pose / k_final_finish complete the record

    const bool calibrated = cam1.has_prior_focal_length && cam2.has_prior_focal_length;
    if ((uint64_t)n < o.min_num_inliers) {
      config = DSM_CONFIG_DEGENERATE;  // two_view_geometry.cc:298-301, 433-436
    } else {
      RansacReport E_rep, F_rep, H_rep;
      E_rep.success = false;
      E_rep.num_inliers = 0;
      E_rep.num_trials = E_rep.num_models = 0;
      if (calibrated) E_rep = p.reports[(size_t)pi * 3 + FAM_E];
      F_rep = p.reports[(size_t)pi * 3 + FAM_F];
      H_rep = p.reports[(size_t)pi * 3 + FAM_H];
      const unsigned char* maskE = p.masks + (size_t)FAM_E * p.mask_stride + moff;
      unsigned char* maskF = p.masks + (size_t)FAM_F * p.mask_stride + moff;
      const unsigned char* maskH = p.masks + (size_t)FAM_H * p.mask_stride + moff;
      if (calibrated) {
        for (int k = 0; k < 9; ++k) Em[k] = E_rep.model[k];
        ntr[0] = E_rep.num_trials;
        nmo[0] = E_rep.num_models;
      }
      for (int k = 0; k < 9; ++k) {
        Fm[k] = F_rep.model[k];
        Hm[k] = H_rep.model[k];
      }
      ntr[1] = F_rep.num_trials;
      nmo[1] = F_rep.num_models;
      ntr[2] = H_rep.num_trials;
      nmo[2] = H_rep.num_models;
      // the generator continues where the H family stopped (watermark RANSAC, :547-549)
      // (generator_load consumes the "resume from the snapshot" mark an early stop of the H family leaves; a pair that is
      // parked below for its translation table comes through here a second time and must find the mark again)
      const uint32_t snap_mark = p.pair_state[(size_t)pi * PAIR_STATE_WORDS + PS_USE_SNAP];
      generator_load(&sm->gen, p.pair_state + (size_t)pi * PAIR_STATE_WORDS, lane);
      gen_loaded = true;

      PairWork w;
      w.n = n;
      w.resid = resid;
      w.inl = inl;
      w.tall = ws.tall;
      w.models = ws.models;
      w.sm = sm;
      w.lane = lane;
      w.pts = pts_px;
      w.nt_table = nullptr;
      const unsigned char* best_mask = nullptr;
      const uint64_t mni = o.min_num_inliers;
      if (calibrated) {
        // EstimateCalibrated decision tree, two_view_geometry.cc:344-413
        if ((!E_rep.success && !F_rep.success && !H_rep.success) ||
            (E_rep.num_inliers < mni && F_rep.num_inliers < mni && H_rep.num_inliers < mni)) {
          config = DSM_CONFIG_DEGENERATE;
        } else {
          const double E_F = (double)E_rep.num_inliers / (double)F_rep.num_inliers;
          const double H_F = (double)H_rep.num_inliers / (double)F_rep.num_inliers;
          const double H_E = (double)H_rep.num_inliers / (double)E_rep.num_inliers;
          if (E_rep.success && E_F > o.min_E_F_inlier_ratio && E_rep.num_inliers >= mni) {
            if (E_rep.num_inliers >= F_rep.num_inliers) {
              num_inliers = E_rep.num_inliers;
              best_mask = maskE;
            } else {
              num_inliers = F_rep.num_inliers;
              best_mask = maskF;
            }
            if (H_E > o.max_H_inlier_ratio) {
              config = DSM_CONFIG_PLANAR_OR_PANORAMIC;
              if (H_rep.num_inliers > num_inliers) {
                num_inliers = H_rep.num_inliers;
                best_mask = maskH;
              }
            } else {
              config = DSM_CONFIG_CALIBRATED;
            }
          } else if (F_rep.success && F_rep.num_inliers >= mni) {
            num_inliers = F_rep.num_inliers;
            best_mask = maskF;
            if (H_F > o.max_H_inlier_ratio) {
              config = DSM_CONFIG_PLANAR_OR_PANORAMIC;
              if (H_rep.num_inliers > num_inliers) {
                num_inliers = H_rep.num_inliers;
                best_mask = maskH;
              }
            } else {
              config = DSM_CONFIG_UNCALIBRATED;
            }
          } else if (H_rep.success && H_rep.num_inliers >= mni) {
            num_inliers = H_rep.num_inliers;
            best_mask = maskH;
            config = DSM_CONFIG_PLANAR_OR_PANORAMIC;
          } else {
            config = DSM_CONFIG_DEGENERATE;
          }
        }
      } else {
        // EstimateUncalibrated, two_view_geometry.cc:461-488
        if ((!F_rep.success && !H_rep.success) || (F_rep.num_inliers < mni && H_rep.num_inliers < mni)) {
          config = DSM_CONFIG_DEGENERATE;
        } else {
          const double H_F = (double)H_rep.num_inliers / (double)F_rep.num_inliers;
          config = (H_F > o.max_H_inlier_ratio) ? DSM_CONFIG_PLANAR_OR_PANORAMIC : DSM_CONFIG_UNCALIBRATED;
          if (F_rep.success) {
            num_inliers = F_rep.num_inliers;
            best_mask = maskF;
          } else {
            num_inliers = 0;  // all-false mask
            for (int i = lane; i < n; i += 64) maskF[i] = 0;
            best_mask = maskF;
            wv_sync();
          }
        }
      }

      if (best_mask != nullptr) {
        have_mask = true;
        // ExtractInlierMatches (two_view_geometry.cc:53-65): ordered compaction; also inl[] for later
        int total = 0;
        for (int b0 = 0; b0 < n; b0 += 64) {
          const int i = b0 + lane;
          const bool in = (i < n) && best_mask[i];
          const unsigned long long bal = __ballot(in);
          if (in) {
            const int pos = total + __popcll(bal & ((1ull << lane) - 1ull));
            out_inl[2 * pos] = matches[2 * i];
            out_inl[2 * pos + 1] = matches[2 * i + 1];
            inl[pos] = i;
          }
          total += __popcll(bal);
        }
        wv_sync();
        num_inliers = (uint32_t)total;

        // DetectWatermark, two_view_geometry.cc:491-555
        if (o.detect_watermark && num_inliers > 0) {
          const double diagonal1 = sqrt((double)(cam1.width * cam1.width + cam1.height * cam1.height));
          const double diagonal2 = sqrt((double)(cam2.width * cam2.width + cam2.height * cam2.height));
          const double minx1 = o.watermark_border_size * diagonal1, miny1 = minx1;
          const double maxx1 = cam1.width - minx1, maxy1 = cam1.height - miny1;
          const double minx2 = o.watermark_border_size * diagonal2, miny2 = minx2;
          const double maxx2 = cam2.width - minx2, maxy2 = cam2.height - miny2;
          int in_border = 0;
          for (int b0 = 0; b0 < total; b0 += 64) {
            const int j = b0 + lane;
            bool hit = false;
            if (j < total) {
              const double* q = pts_px + 4 * (size_t)inl[j];
              const bool in1 = q[0] >= minx1 && q[0] <= maxx1 && q[1] >= miny1 && q[1] <= maxy1;
              const bool in2 = q[2] >= minx2 && q[2] <= maxx2 && q[3] >= miny2 && q[3] <= maxy2;
              hit = !in1 && !in2;
            }
            in_border += __popcll(__ballot(hit));
          }
          const double border_ratio = (double)in_border / (double)num_inliers;
          if (!(border_ratio < o.watermark_min_inlier_ratio)) {
            // RANSAC::ComputeNumTrials of the translation estimator is tabulated per sample count like the others,
            // but only for the inlier counts that ever get here (a watermark suspect is rare).  No table yet: note
            // the pair and leave it untouched -- nothing has been drawn from its generator, everything written so
            // far is rewritten identically -- the host builds the table and sends the pair through this kernel again.
            const uint64_t toff = p.nt_off_t[total];
            if (toff == 0) {
              if (lane == 0) {
                const uint32_t slot = atomicAdd(p.wm_count, 1u);
                p.wm_redo[slot] = pi;
                p.wm_total[slot] = (uint32_t)total;
                p.pose_jobs[pi].ncmb = 0;  // nothing for k_final_pose / k_final_finish until the pair comes back
                p.pair_state[(size_t)pi * PAIR_STATE_WORDS + PS_USE_SNAP] = snap_mark;
              }
              continue;
            }
            // translation LO-RANSAC over the inlier points (gathered into pts_norm's unused half or ipts)
            double* tp = pts3d_a;  // 4 * total doubles fit: pts3d_a + pts3d_b are contiguous (6n)
            for (int j = lane; j < total; j += 64) {
              const double* q = pts_px + 4 * (size_t)inl[j];
              tp[4 * j + 0] = q[0]; tp[4 * j + 1] = q[1]; tp[4 * j + 2] = q[2]; tp[4 * j + 3] = q[3];
            }
            wv_sync();
            PairWork wt = w;
            wt.n = total;
            wt.pts = tp;
            wt.nt_table = p.nt_table_t + (toff - 1);
            RansacOpt rt;
            rt.max_error = o.max_error;
            rt.min_num_trials = (uint32_t)o.min_num_trials;
            rt.max_num_trials = p.max_trials[FAM_T];
            RansacReport T_rep;
            lo_ransac<FAM_T>(wt, rt, sidx, &T_rep);
            ntr[3] = T_rep.num_trials;
            nmo[3] = T_rep.num_models;
            const double inlier_ratio = (double)T_rep.num_inliers / (double)num_inliers;
            if (inlier_ratio >= o.watermark_min_inlier_ratio) config = DSM_CONFIG_WATERMARK;
            // inl[] was reused by the translation RANSAC: rebuild it from the mask
            int tot2 = 0;
            for (int b0 = 0; b0 < n; b0 += 64) {
              const int i = b0 + lane;
              const bool in = (i < n) && best_mask[i];
              const unsigned long long bal = __ballot(in);
              if (in) inl[tot2 + __popcll(bal & ((1ull << lane) - 1ull))] = i;
              tot2 += __popcll(bal);
            }
            wv_sync();
          }
        }
      }

      // EstimateWithRelativePose, two_view_geometry.cc:232-290 (skipped for DEGENERATE: SURVEY.md H8, and for a pair the
      // stage's post-filter is about to discard).  Here only the candidate poses; checking them (triangulating every
      // inlier per candidate) is k_final_pose's work, at an occupancy this kernel cannot have.
      if (calibrated && have_mask && config != DSM_CONFIG_DEGENERATE && config != DSM_CONFIG_UNDEFINED &&
          !(p.stage_filter && (uint64_t)num_inliers < o.min_num_inliers)) {
        const int ni = (int)num_inliers;
        // inlier_points{1,2}_N, in inlier order: compacted to the front of the pair's own pts_norm rows (inl[j] >= j,
        // a chunk is read completely before it is written; nothing reads pts_norm after this kernel)
        double* ipts_g = p.pts_norm + 4 * moff;
        for (int b0 = 0; b0 < ni; b0 += 64) {
          const int j = b0 + lane;
          double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
          if (j < ni) {
            const double* q = pts_norm + 4 * (size_t)inl[j];
            q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
          }
          wv_sync();
          if (j < ni) {
            ipts_g[4 * j + 0] = q0; ipts_g[4 * j + 1] = q1; ipts_g[4 * j + 2] = q2; ipts_g[4 * j + 3] = q3;
          }
          wv_sync();
        }
        double Rc[4 * 9], tc[4 * 3];
        int ncmb;
        if (config == DSM_CONFIG_CALIBRATED || config == DSM_CONFIG_UNCALIBRATED) {
          // DecomposeEssentialMatrix, base/essential_matrix.cc:41-62
          double U[9], V[9], sv[3];
          pl_jacobi_svd_square<3, true>(Em, U, V, sv);
          double Ur[9], Vt[9];  // row-major U, and V^T (row-major)
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
              Ur[i * 3 + j] = U[j * 3 + i];
              Vt[i * 3 + j] = V[i * 3 + j];
            }
          if (m3_det(Ur) < 0)
            for (int i = 0; i < 9; ++i) Ur[i] *= -1;
          if (m3_det(Vt) < 0)
            for (int i = 0; i < 9; ++i) Vt[i] *= -1;
          const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
          double Wt[9], T1[9], R1[9], R2[9];
          m3_transpose(W, Wt);
          m3_mul(Ur, W, T1);
          m3_mul(T1, Vt, R1);
          m3_mul(Ur, Wt, T1);
          m3_mul(T1, Vt, R2);
          const double u2[3] = {Ur[2], Ur[5], Ur[8]};
          double t0[3];
          normalized3(u2, t0);
          for (int k = 0; k < 9; ++k) {
            Rc[0 * 9 + k] = R1[k];
            Rc[1 * 9 + k] = R2[k];
            Rc[2 * 9 + k] = R1[k];
            Rc[3 * 9 + k] = R2[k];
          }
          for (int k = 0; k < 3; ++k) {
            tc[0 * 3 + k] = t0[k];
            tc[1 * 3 + k] = t0[k];
            tc[2 * 3 + k] = -t0[k];
            tc[3 * 3 + k] = -t0[k];
          }
          ncmb = 4;
        } else {
          // PoseFromHomographyMatrix, base/homography_matrix.cc:167-192 (K from Camera::CalibrationMatrix)
          double K1[9], K2[9];
          calibration_matrix(cam1, K1);
          calibration_matrix(cam2, K2);
          ncmb = decompose_homography(Hm, K1, K2, Rc, tc);
        }
        if (lane == 0) {
          PoseJob* job = p.pose_jobs + pi;
          job->ncmb = ncmb;
          job->ni = ni;
          for (int k = 0; k < 36; ++k) job->Rc[k] = Rc[k];
          for (int k = 0; k < 12; ++k) job->tc[k] = tc[k];
        }
        pose_pending = true;
      }
    }
    if (!pose_pending && lane == 0) p.pose_jobs[pi].ncmb = 0;

    if (p.keep_generator && gen_loaded) {  // EstimateMultiple: the next pass continues this stream
      wv_sync();
      generator_store(&sm->gen, p.pair_state + (size_t)pi * PAIR_STATE_WORDS, lane);
    }
    // SiftFeatureMatcher::Match post-filter (matching.cc:824-831) when requested
    if (p.stage_filter && (uint64_t)num_inliers < o.min_num_inliers) {
      config = DSM_CONFIG_UNDEFINED;
      for (int k = 0; k < 9; ++k) Em[k] = Fm[k] = Hm[k] = 0.0;
      for (int k = 0; k < 4; ++k) qvec[k] = 0.0;
      for (int k = 0; k < 3; ++k) tvec[k] = 0.0;
      tri_angle = 0;
      num_inliers = 0;
    }
    if (lane == 0) {
      out->config = config;
      out->num_inliers = num_inliers;
      out->num_matches = (uint32_t)n;
      out->reserved = 0;
      for (int k = 0; k < 9; ++k) {
        out->F[k] = Fm[k];
        out->E[k] = Em[k];
        out->H[k] = Hm[k];
      }
      for (int k = 0; k < 4; ++k) out->qvec[k] = qvec[k];
      for (int k = 0; k < 3; ++k) out->tvec[k] = tvec[k];
      out->tri_angle = tri_angle;
      p.inl_counts[pi] = num_inliers;
      for (int k = 0; k < 4; ++k) {
        out->num_trials[k] = ntr[k];
        out->num_models[k] = nmo[k];
      }
    }
  }
}

#ifndef K_FINAL_POSE_WAVES
#define K_FINAL_POSE_WAVES 3
#endif
// EstimateWithRelativePose, the part that costs: for every candidate pose CheckCheirality (pose.cc:225-247) --
// every inlier triangulated, a 4 x 4 SVD per point -- and, of the points in front of both cameras, the median
// triangulation angle (triangulation.cc:183-218, math.h:211-229).  A wave per (pair, candidate) at three waves per SIMD;
// inside k_verify_final (one wave per SIMD, 512 VGPRs of decision tree around it) the same loops ran latency-bound.
// The median is computed for every candidate although only the winner's is used: it is cheap next to the SVDs and
// saves a second pass over the winner's points.
__global__ __launch_bounds__(64, K_FINAL_POSE_WAVES) void k_final_pose(const VerifyParams p) {
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  double* pts3d = ws.pts3d_a;
  // the angles of a candidate's points: read cnt times each by the rank counting below -- from LDS when they fit
  // (a broadcast ds_read instead of a global load per comparison), from the work area otherwise
  constexpr int kAngLds = 1024;
  __shared__ double s_ang[kAngLds];
  const uint32_t n_items = (p.final_list ? p.n_final : p.n_chunk) * 4u;
  for (uint32_t idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    const uint32_t pi = p.final_list ? p.final_list[idx >> 2] : p.pair0 + (idx >> 2);
    const int c = (int)(idx & 3u);
    PoseJob* job = p.pose_jobs + pi;
    if (c >= job->ncmb) continue;
    wv_sync();
    const uint64_t moff = p.match_off[pi];
    const double* ipts = p.pts_norm + 4 * moff;
    double R[9], t[3];
    for (int k = 0; k < 9; ++k) R[k] = job->Rc[c * 9 + k];
    for (int k = 0; k < 3; ++k) t[k] = job->tc[c * 3 + k];
    const int cnt = check_cheirality(R, t, ipts, job->ni, pts3d, lane);
    double med = 0.0;
    if (cnt > 0) {
      // Median(CalculateTriangulationAnglesWithPM), triangulation.cc:183-218, math.h:211-229
      double c2[3];
      for (int i = 0; i < 3; ++i) c2[i] = -(R[0 * 3 + i] * t[0] + R[1 * 3 + i] * t[1] + R[2 * 3 + i] * t[2]);
      const double c1[3] = {-(1.0 * 0.0 + 0.0 * 0.0 + 0.0 * 0.0), -(0.0 * 0.0 + 1.0 * 0.0 + 0.0 * 0.0), -(0.0 * 0.0 + 0.0 * 0.0 + 1.0 * 0.0)};
      double baseline2 = 0;
      for (int i = 0; i < 3; ++i) baseline2 += (c1[i] - c2[i]) * (c1[i] - c2[i]);
      // (two instantiations, not a selected pointer: that would compile to flat loads)
      auto median_of_angles = [&](double* ang) {
        for (int j = lane; j < cnt; j += 64) {
          const double* X = pts3d + 3 * (size_t)j;
          double r1 = 0, r2 = 0;
          for (int k = 0; k < 3; ++k) {
            r1 += (X[k] - c1[k]) * (X[k] - c1[k]);
            r2 += (X[k] - c2[k]) * (X[k] - c2[k]);
          }
          const double ray1 = sqrt(r1), ray2 = sqrt(r2);
          const double angle = fabs(acos((ray1 * ray1 + ray2 * ray2 - baseline2) / (2 * ray1 * ray2)));
          ang[j] = isnan(angle) ? 0.0 : (angle < M_PI - angle ? angle : M_PI - angle);
        }
        wv_sync();
        // median by rank counting: element of rank mid (and mid-1 for even sizes)
        const int mid = cnt / 2;
        double lo_v = 0.0, hi_v = 0.0;
        for (int b0 = 0; b0 < cnt; b0 += 64) {
          const int j = b0 + lane;
          bool is_mid = false, is_lo = false;
          double a = 0.0;
          if (j < cnt) {
            a = ang[j];
            int rank = 0;
            for (int k = 0; k < cnt; ++k) {
              const double b = ang[k];
              rank += (b < a) || (b == a && k < j);
            }
            is_mid = rank == mid;
            is_lo = rank == mid - 1;
          }
          const unsigned long long bm = __ballot(is_mid), bl = __ballot(is_lo);
          if (bm) hi_v = __shfl(a, __ffsll((long long)bm) - 1);
          if (bl) lo_v = __shfl(a, __ffsll((long long)bl) - 1);
        }
        return (cnt % 2 == 0) ? (hi_v + lo_v) / 2.0 : hi_v;
      };
      med = cnt <= kAngLds ? median_of_angles(s_ang) : median_of_angles(ws.resid);
    }
    if (lane == 0) {
      job->cnt[c] = cnt;
      job->med[c] = med;
    }
  }
}

// The winner among the candidates (pose.cc:80-105: the last one with the most points in front of both cameras),
// its quaternion, the median angle, PLANAR vs PANORAMIC (two_view_geometry.cc:266-279).  A lane per pair.
__global__ __launch_bounds__(64) void k_final_finish(const VerifyParams p) {
  const uint32_t n_items = p.final_list ? p.n_final : p.n_chunk;
  const uint32_t idx = blockIdx.x * 64u + threadIdx.x;
  if (idx >= n_items) return;
  const uint32_t pi = p.final_list ? p.final_list[idx] : p.pair0 + idx;
  const PoseJob* job = p.pose_jobs + pi;
  if (job->ncmb <= 0) return;
  dsm_two_view_geometry* out = p.tvg + pi;
  double Rbest[9], tvec[3] = {0, 0, 0};
  for (int k = 0; k < 9; ++k) Rbest[k] = 0.0;
  int nbest = 0, cbest = -1;
  for (int c = 0; c < job->ncmb; ++c) {
    if (job->cnt[c] >= nbest) {
      for (int k = 0; k < 9; ++k) Rbest[k] = job->Rc[c * 9 + k];
      for (int k = 0; k < 3; ++k) tvec[k] = job->tc[c * 3 + k];
      nbest = job->cnt[c];
      cbest = c;
    }
  }
  double qvec[4];
  rotation_to_quaternion(Rbest, qvec);
  double tri_angle = (nbest == 0 || cbest < 0) ? 0.0 : job->med[cbest];
  int config = out->config;
  if (config == DSM_CONFIG_PLANAR_OR_PANORAMIC) {
    const double tn = sqrt(tvec[0] * tvec[0] + tvec[1] * tvec[1] + tvec[2] * tvec[2]);
    if (tn == 0) {
      config = DSM_CONFIG_PANORAMIC;
      tri_angle = 0;
    } else {
      config = DSM_CONFIG_PLANAR;
    }
  }
  out->config = config;
  for (int k = 0; k < 4; ++k) out->qvec[k] = qvec[k];
  for (int k = 0; k < 3; ++k) out->tvec[k] = tvec[k];
  out->tri_angle = tri_angle;
}

size_t verify_scratch_bytes_per_block(uint32_t n_max) { return verify_scratch_doubles(n_max) * sizeof(double); }
size_t verify_smem_bytes(uint32_t n_max) { return ((sizeof(VSmem) + 15) / 16) * 16 + (size_t)(n_max > 0 ? n_max : 1) * 4; }

static void launch_final_pose_finish(const VerifyParams& p, uint32_t n_blocks, hipStream_t st);
#ifdef DSM_CHECK_BUILD  // cross-check schedule: libdagsfm_mi355x_check.so only
void launch_verify(const VerifyParams& p, uint32_t n_blocks, hipStream_t st) {
  if (p.n_pairs == 0 || n_blocks == 0) return;
  const size_t smem = verify_smem_bytes(p.n_max);
  hipLaunchKernelGGL(k_verify_prep, dim3(n_blocks), dim3(64), 0, st, p);
  hipLaunchKernelGGL(k_ransac<FAM_E>, dim3(n_blocks), dim3(64), smem, st, p);
  hipLaunchKernelGGL(k_ransac<FAM_F>, dim3(n_blocks), dim3(64), smem, st, p);
  hipLaunchKernelGGL(k_ransac<FAM_H>, dim3(n_blocks), dim3(64), smem, st, p);
  hipLaunchKernelGGL(k_verify_final<2>, dim3(n_blocks), dim3(64), smem, st, p);
  launch_final_pose_finish(p, n_blocks, st);
}
#endif  // DSM_CHECK_BUILD

// ------------------------------------------------------------------------------------ phase-split pipeline
// Same LO-RANSAC, different schedule: the sequential part of loransac.h:91-233 is only (a) drawing the
// samples and (b) comparing supports in trial order; solving and scoring the speculated trials is flat
// data-parallel work.  Per family and round:
//   k_sample        wave per pair: `batch` minimal samples, a lane per draw of a trial (light kernel, high occupancy)
//   k_solve/k_score lane per hypothesis (per model) over ALL pairs x trials: minimal solve, inlier counts
//   k_replay        wave per pair: scans the counts in trial order (ballot-skipping the trials that can change
//                   nothing), re-scores candidates with the in-order residual_sum, runs the local
//                   optimisation, applies the dynamic stop, rewinds the generator on an early stop
// The host repeats the round while any pair is still active (active_count).
uint32_t vp_batch(int fam, uint32_t max_trials, uint32_t min_trials) {
  // E / F normally stop after ~60-250 trials (64 / 128 speculated per round); a pair can never stop before
  // min_num_trials, so with a large minimum (fixed-trial schedules: min = max = 4 096) whole
  // rounds of that size are certain to be consumed -- fewer, larger rounds
  uint32_t b;
  if (fam == FAM_E) b = 64;
  else if (fam == FAM_F) b = 128;
  else b = ((max_trials + 63u) / 64u) * 64u;  // H runs to its cap on non-planar scenes
  const uint32_t cap = fam == FAM_E ? 512u : (fam == FAM_F ? 1024u : 2048u);
  const uint32_t want = ((min_trials < max_trials ? min_trials : max_trials) + 63u) / 64u * 64u;
  if (want > b) b = want;
  return b < 64u ? 64u : (b > cap ? cap : b);
}
uint32_t vp_maxm(int fam) { return fam == FAM_E ? 10u : (fam == FAM_F ? 3u : 1u); }

#define SAMPLER_PREFIX (((sizeof(MtState) + sizeof(WvSampler) + 15) / 16) * 16)

template <int FAM>
__global__ __launch_bounds__(64) void k_sample(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  MtState* gen = reinterpret_cast<MtState*>(smem_raw);
  WvSampler* ws = reinterpret_cast<WvSampler*>(smem_raw + sizeof(MtState));
  uint32_t* sidx = reinterpret_cast<uint32_t*>(smem_raw + SAMPLER_PREFIX);
  typedef Fam<FAM> F;
  const int lane = threadIdx.x;
  SegGrab wgrab;
  const uint32_t grain = work_grain(p.n_chunk);
  uint32_t pl_static = blockIdx.x;
  for (;;) {
    wv_sync();
    uint32_t pl;
    if (FAM == FAM_H) {  // thousands of draws per pair: dynamic hand-out as in k_replay; E / F rounds are too short for it
      pl = grab_seg(wgrab, GRAB_SAMPLE(p), p.n_chunk, lane, grain);
      if (pl == GRAB_DONE) break;
      if (pl >= p.n_chunk) continue;
    } else {
      pl = pl_static;
      pl_static += gridDim.x;
    }
    if (pl >= p.n_chunk) break;
    const uint32_t pi = p.pair0 + pl;
    FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
    if (!fs->active) continue;
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    uint32_t* st = p.pair_state + (size_t)pi * PAIR_STATE_WORDS;
    generator_load(gen, st, lane);
    uint32_t* sg = p.sidx_g + moff;
    if (fs->rounds == 0) {
      for (int i = lane; i < n; i += 64) sidx[i] = (uint32_t)i;  // sampler.Initialize
    } else {
      for (int i = lane; i < n; i += 64) sidx[i] = sg[i];
    }
    const uint32_t remaining = p.max_trials[FAM] - fs->rep.num_trials;
    // Speculation is only useful up to the dynamic stop: the loop aborts at the first trial >= max(dyn_max_num_trials,
    // min_num_trials) that yields a model (loransac.h:190-194), and dyn_max_num_trials never grows (the best inlier
    // count never shrinks).  After the first round, draw just that many trials plus a margin for samples without a
    // model; if the margin was too small the pair simply stays active for another round.
    uint32_t want = p.first_batch[FAM] < p.batch ? p.first_batch[FAM] : p.batch;
    if (fs->rounds > 0) {
      want = p.batch;
      const uint32_t mt = (uint32_t)p.opt.min_num_trials;
      const uint32_t thr = fs->dyn_max > mt ? fs->dyn_max : mt;
      const uint32_t T0 = fs->rep.num_trials;
      const uint32_t margin = p.spec_margin[FAM];  // (E 8, F / H 4: capi.hip)
      const uint32_t need = (thr > T0 ? thr - T0 : 0u) + margin;
      if (need < want) want = need;
    }
    const int nb = (int)(remaining < want ? remaining : want);
    // the round's list of hypotheses for the compact solver grids (hyp_of_lane): this pair's nb entries in segment pl % 64
    uint32_t hyp_base = 0;
    if (FAM != FAM_H && p.hyp_map && lane == 0) hyp_base = atomicAdd(GRAB_SAMPLE(p) + (pl % GRAB_SEGS) * GRAB_STRIDE, (uint32_t)nb);
    generator_store(gen, st + PS_SNAP, lane);  // snapshot before this round's draws
    wv_sync();
    wv_draw_samples<F::K>(gen, ws, sidx, (uint32_t)n, nb, p.samples + ((size_t)pl * p.batch) * 7,
                          p.draws_end + (size_t)pl * p.batch, lane, p.sampler_serial != 0);
    if (FAM != FAM_H && p.hyp_map) {
      hyp_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)hyp_base);
      uint32_t* hm = p.hyp_map + (size_t)(pl % GRAB_SEGS) * p.hyp_seg_cap + hyp_base;
      for (int i = lane; i < nb; i += 64) hm[i] = (pl << HYP_T_BITS) | (uint32_t)i;
    }
    if (lane == 0) {
      fs->nb = (uint32_t)nb;
      fs->t_pos = 0;
      fs->m_pos = 0;
      fs->lo_wait = 0;
    }
    wv_sync();
    generator_store(gen, st, lane);
    for (int i = lane; i < n; i += 64) sg[i] = sidx[i];
  }
}

// the pair's correspondences are staged in LDS when they fit
#define VP_LDS_PTS 1536
// A pair's correspondences into LDS (and / or the largest |coordinate| per column): EIGHT loads of a lane in flight at a time.
// As a plain loop the compiler emitted load, s_waitcnt vmcnt(0), ds_write per element -- sixteen memory latencies in a row in front
// of every 64-slot workgroup of a 256-match pair, as long as the scoring loop that follows them.  (max is exact and order-free.)
template <bool STORE, bool MAXIMA>
DSM_DEV double stage_points_batched(const double* gpts, int n4, double* spts, int lane) {
  constexpr int U = 8;
  double m = 0.0;  // this lane only ever sees column (lane & 3): e = lane + 64 k
  int e = lane;
  for (; e + (U - 1) * 64 < n4; e += U * 64) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = gpts[e + u * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (STORE) spts[e + u * 64] = v[u];
      if (MAXIMA) m = fmax(m, fabs(v[u]));
    }
  }
  if (n4 > 0) {  // the rest: up to U - 1 further elements of this lane, still issued together (addresses clamped, not branched around)
    double v[U - 1];
#pragma unroll
    for (int u = 0; u < U - 1; ++u) v[u] = gpts[e + u * 64 < n4 ? e + u * 64 : n4 - 1];
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
      if (e + u * 64 < n4) {
        if (STORE) spts[e + u * 64] = v[u];
        if (MAXIMA) m = fmax(m, fabs(v[u]));
      }
    }
  }
  return m;
}
// Which (pair, trial) does this lane solve?  Two grids for the lane-per-hypothesis solvers (k_solve<F>, k_solve_e_build / _lu /
// k_roots_e):
//   identity  grid (n_chunk, batch / 64): workgroup (pl, b) takes the trials 64 b .. 64 b + 63 of pair pl.  A pair's FIRST round
//             speculates whole waves (64 / 128 trials); every later round only what its dynamic stop still asks for (k_sample), a
//             partial wave per pair: 18 % of the E solvers' lanes and 30 % of k_solve<F>'s had no trial (round 6 counters).
//   compact   from the second round on k_sample lists the round's hypotheses of ALL pairs (p.hyp_map) and the solvers take 64
//             consecutive entries of that list per wave, whatever pairs they belong to.  The list has GRAB_SEGS segments (pair pl
//             appends to segment pl % 64, one atomic per pair and round on the segment's own cache line -- a single counter would
//             serialise 10^5 same-address atomics at 11.4 ns each); segment s holds hyp_seg_cap entries from s * hyp_seg_cap, its
//             length is k_sample's hand-out counter s (zeroed at round start, unused by the E / F samplers otherwise).
// Which lane computes a hypothesis does not enter its result: both grids write the same bytes to the same slots.
// COMPACT is a template parameter of the kernels, not a run-time branch: on the pair grid pl is the workgroup's (a scalar) and the
// kernels address the pair's records from a wave-uniform base + lane; behind a run-time branch pl became a vector value on BOTH grids
// and k_solve_e_lu_reg, which streams 1.6 KB per hypothesis, ran 6.7 instead of 5.6 ms.
template <bool COMPACT>
DSM_DEV bool hyp_of_lane(const VerifyParams& p, int fam, uint32_t& pl, int& t) {
  if (COMPACT) {
    const uint32_t s = blockIdx.x % GRAB_SEGS, w = blockIdx.x / GRAB_SEGS;
    const uint32_t cnt = GRAB_SAMPLE(p)[s * GRAB_STRIDE];
    const uint32_t i = w * 64 + threadIdx.x;
    if (i >= cnt) return false;
    const uint32_t e = p.hyp_map[(size_t)s * p.hyp_seg_cap + i];
    pl = e >> HYP_T_BITS;
    t = (int)(e & ((1u << HYP_T_BITS) - 1u));
    return true;
  }
  pl = blockIdx.x;
  const FamState* fs = p.fam_state + (size_t)(p.pair0 + pl) * 3 + fam;
  if (!fs->active) return false;
  t = blockIdx.y * 64 + threadIdx.x;
  return t < (int)fs->nb;
}
// F and H: solver and inlier counting as two kernels.  k_solve keeps the solver's working set (F: the 9 x 7
// matrix in lane-interleaved LDS; H: ~220 VGPRs) away from the counting loop, which is pure FP64 VALU work
// with a 9-double model per lane and wants many resident waves; k_score gives every (trial, model) slot its
// own lane, so a 7-point sample with three roots costs three lanes instead of three passes of its lane.
template <int FAM, bool COMPACT = false>
__global__ __launch_bounds__(64, 2) void k_solve(const VerifyParams p) {
  typedef Fam<FAM> F;
  uint32_t pl;
  int t;
  if (!hyp_of_lane<COMPACT>(p, FAM, pl, t)) return;
  const uint32_t pi = p.pair0 + pl;
  const double* pts = p.pts_px + 4 * p.match_off[pi];
  const uint32_t* smp = p.samples + ((size_t)pl * p.batch + t) * 7;
  double xs[F::K * 4];
#pragma unroll
  for (int i = 0; i < F::K; ++i) {
    const double* q = pts + (size_t)smp[i] * 4;
    xs[i * 4 + 0] = q[0]; xs[i * 4 + 1] = q[1]; xs[i * 4 + 2] = q[2]; xs[i * 4 + 3] = q[3];
  }
  double mloc[F::MAXM * 9];
  int nm;
  if constexpr (FAM == FAM_F)
    nm = seven_point_reg(xs, mloc);  // 9 x 7 pivoted QR, cubic roots: all in registers
  else
    nm = fam_minimal<FAM>(xs, mloc);
  p.nmodels[(size_t)pl * p.batch + t] = nm;
  double* gm = p.models + ((size_t)pl * p.batch + t) * F::MAXM * 9;
#pragma unroll
  for (int k = 0; k < F::MAXM * 9; ++k)
    if (k < nm * 9) gm[k] = mloc[k];
}

// if (r <= max_residual) { cnt += 1; sum += r; } -- InlierSupportMeasurer::Evaluate's loop body (support_measurement.cc:43-48)
// -- under the execution mask instead of as selected values: the compiler's form is a compare, an add and three or four
// v_cndmask per point (5.5 VALU of the 35 a homography residual costs); masked, the same compare and the same two adds are
// 3 VALU and two scalar instructions.  Same IEEE operations on the same values.  s_and_saveexec_b64 writes SCC: it is in
// the clobber list (without it the compiler keeps the loop's s_cmp result across the block).
DSM_DEV void count_inlier(double r, double max_residual, double& sum, int& cnt) {
  unsigned long long saved_exec;
  asm volatile(
      "v_cmp_le_f64 vcc, %[r], %[mx]\n\t"
      "s_and_saveexec_b64 %[sv], vcc\n\t"
      "v_add_f64 %[sum], %[sum], %[r]\n\t"
      "v_add_u32 %[cnt], 1, %[cnt]\n\t"
      "s_mov_b64 exec, %[sv]"
      : [sum] "+v"(sum), [cnt] "+v"(cnt), [sv] "=&s"(saved_exec)
      : [r] "v"(r), [mx] "v"(max_residual)
      : "vcc", "scc");
}

// InlierSupportMeasurer::Evaluate (support_measurement.cc:43-48) of ONE model by one lane: inlier count and the in-order
// residual_sum over the pair's correspondences (staged in LDS when they fit: spts; otherwise gpts)
template <int FAM>
DSM_DEV void exact_support(const double* M, int n, bool in_lds, const double* spts, const double* gpts, double max_residual, int& cnt,
                           double& sum) {
  // InlierSupportMeasurer::Evaluate (support_measurement.cc:43-48): the lane walks the correspondences in index
  // order, so its running sum IS the in-order residual_sum that decides ties between equal inlier counts
  // (two loops, not one over a selected pointer: a pointer that may be LDS or global compiles to flat loads with a
  // full wait per point; apart, the staged points are ds_read_b128 broadcasts and the loop is unrolled over four points)
  cnt = 0;
  sum = 0;
  if (in_lds) {
    int i = 0;
    for (; i + 4 <= n; i += 4) {  // four points per trip by hand: a loop with inline assembly is not unrolled for us
      const double r0 = fam_residual<FAM>(M, spts + (size_t)i * 4), r1 = fam_residual<FAM>(M, spts + (size_t)(i + 1) * 4);
      const double r2 = fam_residual<FAM>(M, spts + (size_t)(i + 2) * 4), r3 = fam_residual<FAM>(M, spts + (size_t)(i + 3) * 4);
      count_inlier(r0, max_residual, sum, cnt);
      count_inlier(r1, max_residual, sum, cnt);
      count_inlier(r2, max_residual, sum, cnt);
      count_inlier(r3, max_residual, sum, cnt);
    }
    for (; i < n; ++i) count_inlier(fam_residual<FAM>(M, spts + (size_t)i * 4), max_residual, sum, cnt);
  } else {
    int i = 0;
    for (; i + 4 <= n; i += 4) {  // four points per trip by hand: a loop with inline assembly is not unrolled for us
      const double r0 = fam_residual<FAM>(M, gpts + (size_t)i * 4), r1 = fam_residual<FAM>(M, gpts + (size_t)(i + 1) * 4);
      const double r2 = fam_residual<FAM>(M, gpts + (size_t)(i + 2) * 4), r3 = fam_residual<FAM>(M, gpts + (size_t)(i + 3) * 4);
      count_inlier(r0, max_residual, sum, cnt);
      count_inlier(r1, max_residual, sum, cnt);
      count_inlier(r2, max_residual, sum, cnt);
      count_inlier(r3, max_residual, sum, cnt);
    }
    for (; i < n; ++i) count_inlier(fam_residual<FAM>(M, gpts + (size_t)i * 4), max_residual, sum, cnt);
  }
}

// exact_support of ONE model by a QUAD of lanes (round 6, k_score_needed): lane ql of the quad computes the residuals of the points
// i = 4 j + ql, and all four lanes walk the in-order sum together -- the quad's residuals are read in index order through DPP
// quad_perm broadcasts, a point that is no inlier contributes + 0.0 (exact: the sum is a sum of squares, never - 0.0 before an inlier
// made it positive) -- so count and residual_sum are bit for bit those of the one-lane walk (support_measurement.cc:43-48), at a
// quarter of its residual arithmetic per lane.  Every lane of the quad returns the same (cnt, sum).
template <int K>
DSM_DEV double quad_bcast_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);  // quad_perm:[K, K, K, K]
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(u & 0xffffffffull), ctrl, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(u >> 32), ctrl, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int FAM>
DSM_DEV void exact_support_quad(const double* M, int n, bool in_lds, const double* spts, const double* gpts, double max_residual, int lane, int& cnt,
                                double& sum) {
  const int ql = lane & 3;
  cnt = 0;
  sum = 0.0;
  for (int base = 0; base < n; base += 4) {
    const int i = base + ql;
    double r = 0.0;
    bool in = false;
    if (i < n) {
      double q[4];
      if (in_lds) {  // (two branches, not one selected pointer: that would compile to flat loads)
        for (int k = 0; k < 4; ++k) q[k] = spts[(size_t)i * 4 + k];
      } else {
        for (int k = 0; k < 4; ++k) q[k] = gpts[(size_t)i * 4 + k];
      }
      r = fam_residual<FAM>(M, q);
      in = r <= max_residual;
    }
    const double a = in ? r : 0.0;
    sum += quad_bcast_f64<0>(a);
    sum += quad_bcast_f64<1>(a);
    sum += quad_bcast_f64<2>(a);
    sum += quad_bcast_f64<3>(a);
    const unsigned long long bal = __ballot(in);
    cnt += __popcll((bal >> (lane & ~3)) & 0xfull);
  }
}

template <int FAM>
__global__ __launch_bounds__(64, 8) void k_score(const VerifyParams p) {
  typedef Fam<FAM> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* spts = reinterpret_cast<double*>(smem_raw);  // min(n_max, VP_LDS_PTS) x 4 doubles
  const uint32_t pl = blockIdx.x;
  const uint32_t pi = p.pair0 + pl;
  const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
  if (!fs->active) return;
  const int lane = threadIdx.x;
  const int nb = (int)fs->nb;
  const int slot = blockIdx.y * 64 + lane;  // (trial, model) = (slot / MAXM, slot % MAXM)
  if ((int)(blockIdx.y * 64) / F::MAXM >= nb) return;
  const int t = slot / F::MAXM, m = slot - t * F::MAXM;
  const uint64_t moff = p.match_off[pi];
  const int n = (int)(p.match_off[pi + 1] - moff);
  const double* gpts = p.pts_px + 4 * moff;
  const bool in_lds = n <= VP_LDS_PTS;
  if (in_lds) {
    (void)stage_points_batched<true, false>(gpts, 4 * n, spts, lane);
    __syncthreads();
  }
  const double max_residual = p.opt.max_error * p.opt.max_error;
  if (t >= nb || m >= p.nmodels[(size_t)pl * p.batch + t]) return;
  const double* gm = p.models + ((size_t)pl * p.batch + t) * F::MAXM * 9 + m * 9;
  double M[9];
  for (int k = 0; k < 9; ++k) M[k] = gm[k];
  int cnt = 0;
  double sum = 0;
  exact_support<FAM>(M, n, in_lds, spts, gpts, max_residual, cnt, sum);
  p.counts[((size_t)pl * p.batch + t) * F::MAXM + m] = cnt;
  p.sums[((size_t)pl * p.batch + t) * F::MAXM + m] = sum;
}

// ------------------------------------------------------------------------------------ k_score as bound + exact (round 4)
// LO-RANSAC only ever LOOKS at the support of a trial that can still beat the best support so far (loransac.h:150-158:
// `if (support_measurer.Compare(support, best_support))`): a model whose inlier count is below a count some EARLIER model
// (or an earlier round's best) has reached cannot change anything, whatever its exact count and sum are -- and on a
// non-planar pair that is all but a few dozen of the 1 765 homographies.  So the scoring runs in two steps:
//   k_prescore    every (trial, model) slot: a LOWER and an UPPER bound of its inlier count from a division-free test in
//                 fused arithmetic (18 VALU per homography residual instead of 36 + a quarter-rate v_rcp_f64), each point
//                 classified "surely inlier" / "surely outlier" / "uncertain" with margins derived below (H: a lane per slot;
//                 E and F, whose samples have 0..10 / 1 or 3 models: k_prescore_compact, a lane per model in compacted order);
//   k_score_needed  per pair: the running maximum of the LOWER bounds in trial order (starting from the best count of the
//                 earlier rounds); a slot whose UPPER bound reaches it is scored EXACTLY (exact_support: the reference's own
//                 operations, count and in-order sum), every other slot is written as support (0, 0) -- fewer inliers than
//                 the best support at that point of the scan, which is all the replay ever asks of it.
// The bounds are rigorous, not heuristic (u = 2^-53; all comparisons are false on NaN, which makes a point uncertain):
//   H (homography_matrix.cc:94-131): with A_k = |H_k0| max|s_0| + |H_k1| max|s_1| + |H_k2|, the reference's pd_k (3 roundings)
//     and the fused pd_k here (2) both lie within E_k = 4u A_k of the exact value.  A point is classified only if
//     |pd_2| >= P_min = max(2^34 E_2, 2^26 max(E_0, E_1), 2^-400): then pd_2 is known to 2^-34 relative and pd_0 / pd_2 to
//     2^-26 px + 2^-34 |pd_0 / pd_2|, so with coordinates below 2^14 px the reference's dd_k and e_k / pd_2 here
//     (e_k = d_k pd_2 - pd_k) both lie within a = 2^-20 (1 + 2^-5) px + 2^-33 |dd_k| of the exact difference, and
//     sqrt(r) is pinned to 2.8e-6 px + 3e-10 relative on either side: L = e_0^2 + e_1^2 <= T pd_2^2 (1 - 2^-12) implies
//     r_ref <= T and L >= T pd_2^2 (1 + 2^-12) implies r_ref > T for every T >= 2^-6 (margin >= 2.6x).
//   F (utils.cc:87-131): g = F x1, h = F^T x2 (first two rows), C = x2^T F x1.  With B_k / Bt_j the sums of absolute terms
//     over the pair's coordinate maxima, every g_k, h_j is within 4u B of exact, C within E_C = 8u (max|x2_0| B_0 +
//     max|x2_1| B_1 + B_2), D = g_0^2 + g_1^2 + h_0^2 + h_1^2 within 2 sqrt(D) E_D + 4u D, E_D = 4u (B_0 + B_1 + Bt_0 + Bt_1).
//     A point is classified only if D >= D_min = max((2^26 E_D)^2, (2^16 E_C)^2 / T): D is then known to 2^-25 relative and
//     |C| to 2^-16 sqrt(T D), so C^2 <= T D (1 - 2^-12) implies r_ref <= T and C^2 >= T D (1 + 2^-12) implies r_ref > T
//     (needed: delta >= 2^-14 (1 + 2^-10); margin 4x).
//   Magnitudes outside [2^-400, 2^300] (or coordinates beyond 2^14, or T outside [2^-6, 2^40]) switch the classification
//   off for the model: every point uncertain, upper bound n, lower bound 0 -- the slot is then simply scored exactly.
// DSM_SCORE_PREFILTER=0 (dsm_set_debug_option) runs the plain k_score / k_models_score_e instead, =3 the slot-per-lane k_prescore
// for F and the fused k_models_score_e for E: independent schedules of the same results (tools/check_schedules.py, tests).
struct PreBounds {
  double c0, c1;   // H: P_min, unused;  F: D_min, unused   (NaN = classification off)
  double t_lo, t_hi;
};
#define PRESCORE_DELTA 0x1p-12
#define PRESCORE_CU 0x1p-51 /* 4u */

template <int FAM>
DSM_DEV PreBounds prescore_bounds(const double* M, const double mx[4], double T) {
  PreBounds b;
  b.t_lo = T * (1.0 - PRESCORE_DELTA);
  b.t_hi = T * (1.0 + PRESCORE_DELTA);
  b.c1 = 0.0;
  const double nan = __builtin_nan("");
  // (H's margins are absolute, in pixels: T must not be tiny; the Sampson test is scale-free -- E's T is ~1e-5)
  const bool t_ok = (T >= (FAM == FAM_H ? 0x1p-6 : 0x1p-200)) && (T <= 0x1p40);
  const bool x_ok = (mx[0] <= 0x1p14) && (mx[1] <= 0x1p14) && (mx[2] <= 0x1p14) && (mx[3] <= 0x1p14);
  if (FAM == FAM_H) {
    const double A0 = fabs(M[0]) * mx[0] + fabs(M[1]) * mx[1] + fabs(M[2]);
    const double A1 = fabs(M[3]) * mx[0] + fabs(M[4]) * mx[1] + fabs(M[5]);
    const double A2 = fabs(M[6]) * mx[0] + fabs(M[7]) * mx[1] + fabs(M[8]);
    const double E01 = PRESCORE_CU * fmax(A0, A1), E2 = PRESCORE_CU * A2;
    const double pmin = fmax(fmax(0x1p34 * E2, 0x1p26 * E01), 0x1p-400);
    b.c0 = (t_ok && x_ok && pmin <= 0x1p300) ? pmin : nan;
  } else {
    const double B0 = fabs(M[0]) * mx[0] + fabs(M[1]) * mx[1] + fabs(M[2]);
    const double B1 = fabs(M[3]) * mx[0] + fabs(M[4]) * mx[1] + fabs(M[5]);
    const double B2 = fabs(M[6]) * mx[0] + fabs(M[7]) * mx[1] + fabs(M[8]);
    const double Bt0 = fabs(M[0]) * mx[2] + fabs(M[3]) * mx[3] + fabs(M[6]);
    const double Bt1 = fabs(M[1]) * mx[2] + fabs(M[4]) * mx[3] + fabs(M[7]);
    const double ED = PRESCORE_CU * (B0 + B1 + Bt0 + Bt1);
    const double EC = 2.0 * PRESCORE_CU * (mx[2] * B0 + mx[3] * B1 + B2);
    const double d1 = 0x1p26 * ED, d2 = 0x1p16 * EC;
    const double dmin = fmax(fmax(d1 * d1, d2 * d2 / T) * (1.0 + 0x1p-10), 0x1p-600);
    b.c0 = (t_ok && x_ok && ED <= 0x1p170 && EC <= 0x1p170 && dmin <= 0x1p400) ? dmin : nan;
  }
  return b;
}

// one correspondence: surely an inlier of the reference's test / surely an outlier (neither: uncertain)
template <int FAM>
DSM_DEV void prescore_flags(const double* M, const PreBounds& b, const double* q, bool& sure_in, bool& sure_out);
template <int FAM>
DSM_DEV void prescore_point(const double* M, const PreBounds& b, const double* q, int& lb, int& sure_out) {
  bool in, out;
  prescore_flags<FAM>(M, b, q, in, out);
  lb += in ? 1 : 0;
  sure_out += out ? 1 : 0;
}
template <int FAM>
DSM_DEV void prescore_flags(const double* M, const PreBounds& b, const double* q, bool& sure_in, bool& sure_out) {
  if (FAM == FAM_H) {
    const double s0 = q[0], s1 = q[1], d0 = q[2], d1 = q[3];
    const double pd0 = __builtin_fma(M[0], s0, __builtin_fma(M[1], s1, M[2]));
    const double pd1 = __builtin_fma(M[3], s0, __builtin_fma(M[4], s1, M[5]));
    const double pd2 = __builtin_fma(M[6], s0, __builtin_fma(M[7], s1, M[8]));
    const double e0 = __builtin_fma(d0, pd2, -pd0);
    const double e1 = __builtin_fma(d1, pd2, -pd1);
    const double L = __builtin_fma(e0, e0, e1 * e1);
    const double q2 = pd2 * pd2;
    const bool ok = fabs(pd2) >= b.c0;
    sure_in = ok && (L <= b.t_lo * q2);
    sure_out = ok && (L >= b.t_hi * q2);
  } else {
    const double x10 = q[0], x11 = q[1], x20 = q[2], x21 = q[3];
    const double g0 = __builtin_fma(M[0], x10, __builtin_fma(M[1], x11, M[2]));
    const double g1 = __builtin_fma(M[3], x10, __builtin_fma(M[4], x11, M[5]));
    const double g2 = __builtin_fma(M[6], x10, __builtin_fma(M[7], x11, M[8]));
    const double h0 = __builtin_fma(M[0], x20, __builtin_fma(M[3], x21, M[6]));
    const double h1 = __builtin_fma(M[1], x20, __builtin_fma(M[4], x21, M[7]));
    const double C = __builtin_fma(x20, g0, __builtin_fma(x21, g1, g2));
    const double num = C * C;
    const double D = __builtin_fma(g0, g0, __builtin_fma(g1, g1, __builtin_fma(h0, h0, h1 * h1)));
    const bool ok = D >= b.c0;
    sure_in = ok && (num <= b.t_lo * D);
    sure_out = ok && (num >= b.t_hi * D);
  }
}

// stages the pair's points in LDS (when they fit) and returns max |coordinate| per column (x1, y1, x2, y2) to every lane
DSM_DEV void stage_points_with_maxima(const double* gpts, int n, bool in_lds, double* spts, int lane, double mx[4]) {
  double m = in_lds ? stage_points_batched<true, true>(gpts, 4 * n, spts, lane) : stage_points_batched<false, true>(gpts, 4 * n, spts, lane);
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) m = fmax(m, __shfl_xor(m, o));
#pragma unroll
  for (int c = 0; c < 4; ++c) mx[c] = __shfl(m, c);
  if (in_lds) __syncthreads();
}

template <int FAM>
__global__ __launch_bounds__(64, 8) void k_prescore(const VerifyParams p) {
  typedef Fam<FAM> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* spts = reinterpret_cast<double*>(smem_raw);  // min(n_max, VP_LDS_PTS) x 4 doubles
  const uint32_t pl = blockIdx.x;
  const uint32_t pi = p.pair0 + pl;
  const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
  if (!fs->active) return;
  const int lane = threadIdx.x;
  const int nb = (int)fs->nb;
  const int slot = blockIdx.y * 64 + lane;  // (trial, model) = (slot / MAXM, slot % MAXM)
  if ((int)(blockIdx.y * 64) / F::MAXM >= nb) return;
  const int t = slot / F::MAXM, m = slot - t * F::MAXM;
  const uint64_t moff = p.match_off[pi];
  const int n = (int)(p.match_off[pi + 1] - moff);
  const double* gpts = p.pts_px + 4 * moff;
  // (measured, round 4: the points of a pair that fits the LDS by SCALAR loads instead -- wave-uniform addresses, no LDS
  // traffic -- are slower: k_prescore<H> 48.4 vs 37.2 ms; the scalar cache does not hold the 8 KB of every resident pair)
  const bool in_lds = n <= VP_LDS_PTS;
  const double T = p.opt.max_error * p.opt.max_error;
  // bounds of the slot: counts[] = upper bound (-1: no model in this slot), the low word of sums[] = lower bound
  const bool has_slot = t < nb;
  const bool has_model = has_slot && m < p.nmodels[(size_t)pl * p.batch + (has_slot ? t : 0)];
  const double* gm = p.models + ((size_t)pl * p.batch + (has_slot ? t : 0)) * F::MAXM * 9 + m * 9;
  double M[9];
  for (int k = 0; k < 9; ++k) M[k] = has_model ? gm[k] : 0.0;
  int lb = 0, sure_out = n + 1;
  // two separate paths (not one loop over a selected pointer, and no barrier on the way to the global-memory loop: its
  // loads have wave-uniform addresses and nothing may have been stored before them, so they can be scalar loads)
  if (in_lds) {
    double mx[4];
    stage_points_with_maxima(gpts, n, true, spts, lane, mx);
    if (has_model) {
      const PreBounds b = prescore_bounds<FAM>(M, mx, T);
      sure_out = 0;
#pragma unroll 4
      for (int i = 0; i < n; ++i) prescore_point<FAM>(M, b, spts + (size_t)i * 4, lb, sure_out);
    }
  } else {
    double mx[4];
    stage_points_with_maxima(gpts, n, false, nullptr, lane, mx);
    if (has_model) {
      const PreBounds b = prescore_bounds<FAM>(M, mx, T);
      sure_out = 0;
#pragma unroll 4
      for (int i = 0; i < n; ++i) prescore_point<FAM>(M, b, gpts + (size_t)i * 4, lb, sure_out);
    }
  }
  if (has_slot) {
    p.counts[((size_t)pl * p.batch + t) * F::MAXM + m] = n - sure_out;  // -1: no model in this slot
    reinterpret_cast<int32_t*>(p.sums + ((size_t)pl * p.batch + t) * F::MAXM + m)[0] = lb;
  }
}

// ------------------------------------------------------------------------------------ the H bound step with an f32 first stage (round 5)
// k_prescore<H> spends 18 FP64 VALU instructions on every (model, point) to learn what a much coarser test decides for almost all of
// them: is the transfer error below, or above, the threshold?  k_prescore_h2 classifies every point first in PACKED f32 (v_pk_fma_f32:
// two points per instruction) against a BAND of +- 12.5 % around the threshold radius,
//   L~ < 0.875^2 (1 - 2^-8) T Q~  -> surely in        L~ > 1.125^2 (1 + 2^-8) T Q~  -> surely out        (both only if |p~_2| >= g)
//   (L~ = e~_0^2 + e~_1^2, e~_k = d_k p~_2 - p~_k, Q~ = p~_2^2; all f32),
// and only the points in between -- a residual within an eighth of the threshold radius: a few per model -- go, per lane, onto a short
// list that the FP64 test of prescore_flags works off afterwards.  (A first form sent a point to the FP64 test whenever ANY lane of the
// wave failed to reject it: with 64 models per wave that is nearly every point -- 72 ms against k_prescore<H>'s 36;
// profiles/r05_prescore_f32_stage.txt.)  The band is what makes f32 safe without fine margins (u = 2^-24; H scaled by a power of two so
// that its largest entry is in [1, 2), which the homogeneous test does not see; A_k as in prescore_bounds, on the scaled model):
//   inputs rounded to f32 and two fused roundings per row: |p~_k - P_k| <= 5u A_k;  |e~_k - e*_k| <= u (G_k + 1.01 |e*_k|) with
//   G_k = 6.01 max|d_k| A_2 + 5 A_k;  sqrt(L~) within 2.03u ||e*|| + 1.02u G of ||e*|| (G = G_0 + G_1), sqrt(Q~) within 5.6u A_2 of |P_2|.
//   out:  L~ > fl(K_out Q~), K_out >= 1.1272^2 T  =>  ||e*|| >= [1.1272 sqrt(T) (1 - u) (|P_2| - 5.6u A_2) - 1.02u G] / (1 + 2.03u)
//         >= 1.004 sqrt(T) |P_2|   once   |P_2| >= u (51.3 A_2 + 8.29 G / sqrt(T));
//   in:   L~ < fl(K_in Q~),  K_in <= 0.8733^2 T   =>  ||e*|| <= [0.8733 sqrt(T) (1 + u) (|P_2| + 5.6u A_2) + 1.02u G] / (1 - 2.03u)
//         <= 0.996 sqrt(T) |P_2|   once   |P_2| >= u (40 A_2 + 8.4 G / sqrt(T)).
//   The guard on the COMPUTED value, |p~_2| >= g = max(u (58 A_2 + 8.6 G / sqrt(T)), 2^-16 A_2, 2^-24 max(A_0, A_1), 2^-60), covers both
//   (|P_2| >= |p~_2| - 5u A_2), covers the entries that flush to zero in f32 after the scaling (below 2^-126: they move p~_k by at most
//   2^-110), and implies the FP64 test's own precondition |P_2| >= P_min (2^34 E_2 = 2^-17 A_2, 2^26 E_01 = 2^-25 A_01), under which the
//   reference's evaluation is within 2^-20 px (1 + 2^-5) + 2^-33 |dd| of the exact residual (analysis above k_prescore): an exact
//   residual <= 0.996 sqrt(T) is an inlier of the reference, one >= 1.004 sqrt(T) an outlier, for T >= 2^-6 and coordinates below 2^14
//   (t_ok, x_ok: otherwise the f32 stage is off, like the FP64 one).  Inf / NaN (a model beyond f32's range) make every compare
//   false: the point goes to the FP64 test.  tools/check_score_bounds.py holds every slot's exact count against [lower, upper].
// Cost per (model, point): 12 packed instructions per TWO points, three compares and two counts per point: ~12 instead of 18, plus the
// FP64 test for the listed points (the wave loops to its longest list).
typedef float dsm_f32x2 __attribute__((ext_vector_type(2)));
DSM_DEV dsm_f32x2 pk_fma(dsm_f32x2 a, dsm_f32x2 b, dsm_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
DSM_DEV dsm_f32x2 pk2(float x) {
  dsm_f32x2 v = {x, x};
  return v;
}
#define PRESCORE_LIST_CAP 24
__global__ __launch_bounds__(64, 6) void k_prescore_h2(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s32 = reinterpret_cast<float*>(smem_raw);  // the points as f32 pairs (16 bytes per point), then the lanes' lists; the FP64 points stay in global memory / L2
  // (with the FP64 copy in the LDS as well -- 15 KB per 256 points -- the kernel runs at 2.5 waves per SIMD: 57 ms instead of 33)
  const uint32_t pl = blockIdx.x;
  const uint32_t pi = p.pair0 + pl;
  const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM_H;
  if (!fs->active) return;
  const int lane = threadIdx.x;
  const int nb = (int)fs->nb;
  const int t = blockIdx.y * 64 + lane;
  if ((int)(blockIdx.y * 64) >= nb) return;
  const uint64_t moff = p.match_off[pi];
  const int n = (int)(p.match_off[pi + 1] - moff);
  const double* gpts = p.pts_px + 4 * moff;
  const bool in_lds = n <= VP_LDS_PTS;
  const double T = p.opt.max_error * p.opt.max_error;
  const bool has_slot = t < nb;
  const bool has_model = has_slot && 0 < p.nmodels[(size_t)pl * p.batch + (has_slot ? t : 0)];
  const double* gm = p.models + ((size_t)pl * p.batch + (has_slot ? t : 0)) * 9;
  double M[9];
  for (int k = 0; k < 9; ++k) M[k] = has_model ? gm[k] : 0.0;
  int lb = 0, sure_out = n + 1;
  double mx[4];
  stage_points_with_maxima(gpts, n, false, nullptr, lane, mx);
  if (!in_lds) {  // the points do not fit the LDS: the FP64 loop over wave-uniform global addresses, as k_prescore<H>
    if (has_model) {
      const PreBounds b = prescore_bounds<FAM_H>(M, mx, T);
      sure_out = 0;
#pragma unroll 4
      for (int i = 0; i < n; ++i) prescore_point<FAM_H>(M, b, gpts + (size_t)i * 4, lb, sure_out);
    }
  } else {
    // the points once more as f32, two points per 32-byte record: (s0a, s0b, s1a, s1b, d0a, d0b, d1a, d1b); an odd last point twice
    const int npair = (n + 1) / 2;
    uint16_t* lst = reinterpret_cast<uint16_t*>(s32 + (size_t)8 * npair);  // entry k of lane l at [k * 64 + l]
    for (int j = lane; j < npair; j += 64) {
      const int ia = 2 * j, ib = 2 * j + 1 < n ? 2 * j + 1 : 2 * j;
      float4 lo, hi;
      lo.x = (float)gpts[ia * 4 + 0];
      lo.y = (float)gpts[ib * 4 + 0];
      lo.z = (float)gpts[ia * 4 + 1];
      lo.w = (float)gpts[ib * 4 + 1];
      hi.x = (float)gpts[ia * 4 + 2];
      hi.y = (float)gpts[ib * 4 + 2];
      hi.z = (float)gpts[ia * 4 + 3];
      hi.w = (float)gpts[ib * 4 + 3];
      reinterpret_cast<float4*>(s32)[2 * j] = lo;
      reinterpret_cast<float4*>(s32)[2 * j + 1] = hi;
    }
    __syncthreads();
    const PreBounds b = prescore_bounds<FAM_H>(M, mx, T);
    // ---- the model in f32 (largest entry scaled into [1, 2)), the guard g, the band's constants
    float H[9], g = __builtin_inff(), k_out = __builtin_inff(), k_in = -1.0f;
    {
      double big = 0.0;
      for (int k = 0; k < 9; ++k) big = fmax(big, fabs(M[k]));
      const bool usable = has_model && (b.c0 == b.c0) && big >= 0x1p-900 && big <= 0x1p900;  // (c0 is NaN when t_ok / x_ok fail)
      const double sc = usable ? ldexp(1.0, -ilogb(big)) : 0.0;
      double Ms[9];
      for (int k = 0; k < 9; ++k) {
        Ms[k] = M[k] * sc;
        H[k] = (float)Ms[k];
      }
      const double A0 = fabs(Ms[0]) * mx[0] + fabs(Ms[1]) * mx[1] + fabs(Ms[2]);
      const double A1 = fabs(Ms[3]) * mx[0] + fabs(Ms[4]) * mx[1] + fabs(Ms[5]);
      const double A2 = fabs(Ms[6]) * mx[0] + fabs(Ms[7]) * mx[1] + fabs(Ms[8]);
      const double G = 6.01 * (mx[2] + mx[3]) * A2 + 5.0 * (A0 + A1 + 2.0 * A2);
      const double gd = fmax(fmax(0x1p-24 * (58.0 * A2 + 8.6 * G / sqrt(T)), 0x1p-16 * A2), fmax(0x1p-24 * fmax(A0, A1), 0x1p-60)) * 1.001;
      if (usable && gd <= 0x1p100) {
        g = (float)gd * 1.0001f;                                     // rounded up
        k_out = (float)(1.265625 * T * (1.0 + 0x1p-8)) * 1.0001f;      // rounded up
        k_in = (float)(0.765625 * T * (1.0 - 0x1p-8)) * 0.9999f;       // rounded down
      }
    }
    int cnt = 0;
    if (has_model) {
      sure_out = 0;
      const dsm_f32x2 h0 = pk2(H[0]), h1 = pk2(H[1]), h2 = pk2(H[2]), h3 = pk2(H[3]), h4 = pk2(H[4]), h5 = pk2(H[5]), h6 = pk2(H[6]), h7 = pk2(H[7]),
                      h8 = pk2(H[8]), ko = pk2(k_out), ki = pk2(k_in);
      for (int j = 0; j < npair; ++j) {
        const float4 lo = reinterpret_cast<const float4*>(s32)[2 * j], hi = reinterpret_cast<const float4*>(s32)[2 * j + 1];
        const dsm_f32x2 s0 = {lo.x, lo.y}, s1 = {lo.z, lo.w}, d0 = {hi.x, hi.y}, d1 = {hi.z, hi.w};
        const dsm_f32x2 p0 = pk_fma(h0, s0, pk_fma(h1, s1, h2));
        const dsm_f32x2 p1 = pk_fma(h3, s0, pk_fma(h4, s1, h5));
        const dsm_f32x2 p2 = pk_fma(h6, s0, pk_fma(h7, s1, h8));
        const dsm_f32x2 e0 = pk_fma(d0, p2, -p0);
        const dsm_f32x2 e1 = pk_fma(d1, p2, -p1);
        const dsm_f32x2 L = pk_fma(e0, e0, e1 * e1);
        const dsm_f32x2 Q = p2 * p2;
        const dsm_f32x2 QO = ko * Q, QI = ki * Q;
        const bool two = 2 * j + 1 < n;
        const bool ok_a = fabsf(p2.x) >= g, ok_b = two && fabsf(p2.y) >= g;
        const bool out_a = ok_a && (L.x > QO.x), out_b = ok_b && (L.y > QO.y);
        const bool in_a = ok_a && (L.x < QI.x), in_b = ok_b && (L.y < QI.y);
        sure_out += (out_a ? 1 : 0) + (out_b ? 1 : 0);
        lb += (in_a ? 1 : 0) + (in_b ? 1 : 0);
        if (!(in_a || out_a)) {  // inside the band (or unguarded): the FP64 test decides, later
          if (cnt < PRESCORE_LIST_CAP) lst[cnt * 64 + lane] = (uint16_t)(2 * j);
          ++cnt;
        }
        if (two && !(in_b || out_b)) {
          if (cnt < PRESCORE_LIST_CAP) lst[cnt * 64 + lane] = (uint16_t)(2 * j + 1);
          ++cnt;
        }
      }
    }
    // ---- the listed points in FP64; a lane whose list overflowed (a model the f32 stage cannot judge) redoes all of its points
    const bool overflow = cnt > PRESCORE_LIST_CAP;
    if (overflow) {
      lb = 0;
      sure_out = 0;
      for (int i = 0; i < n; ++i) prescore_point<FAM_H>(M, b, gpts + (size_t)i * 4, lb, sure_out);
      cnt = 0;
    }
    int maxc = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
    // four listed points per trip, their loads issued together (a trip pays one memory round trip, not four)
    for (int k0 = 0; k0 < maxc; k0 += 4) {
      double q[4][4];
      bool on[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        on[u] = k0 + u < cnt;
        const double* src = gpts + (size_t)(on[u] ? lst[(k0 + u) * 64 + lane] : 0) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) q[u][c] = src[c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bool in, out;
        prescore_flags<FAM_H>(M, b, q[u], in, out);
        lb += (on[u] && in) ? 1 : 0;
        sure_out += (on[u] && out) ? 1 : 0;
      }
    }
  }
  if (has_slot) {
    p.counts[(size_t)pl * p.batch + t] = n - sure_out;  // -1: no model in this slot
    reinterpret_cast<int32_t*>(p.sums + (size_t)pl * p.batch + t)[0] = lb;
  }
}

#ifdef DSM_CHECK_BUILD
// ------------------------------------------------------------------------------------ the H bound step on the matrix pipe (round 5)
// MEASURED AND NOT ADOPTED -- check build only (DSM_SCORE_PREFILTER=9), kept as the A/B the round-4 verdict asked for:
// config 2, one lane: k_prescore_h_mfma 44.3 ms against k_prescore<H> 35.6 ms (profiles/r05_prescore_mfma_ab.txt); whole step
// 525.1 vs 514.2 ms, alternated twice on one box.  Same parity (113 GPU tests), 0 bound violations on ~9 x 10^8 slots.
// Why it loses: per 16 models x 16 points the VALU form spends 18 x 16 = 288 wave-instructions, this form 14 x 16 = 224 plus
// twelve v_mfma_f64_16x16x4_f64.  On MI355X the FP64 matrix peak EQUALS the FP64 vector peak (78.6 TF: a 16x16x4 f64 MFMA
// holds the SIMD's FP64 datapath for 64 cycles), and the measured time is the SUM of the two streams, not their maximum
// (1 770 cycles per wave and step = 12 x 64 + 224 x ~4.5): the f64 MFMA is another way to issue the same FMAs -- with a
// quarter of them spent on the zero column of K = (s_0, s_1, 1, 0) -- not a second pipe next to the VALU.  The f32 matrix
// pipe is separate, but its results would have to be widened again (3 v_cvt per evaluation) or the whole test redone in f32
// with re-derived margins.
// k_prescore<H> is VALU-issue-bound (18 FP64 VALU per model x point, profiles/r04_verify_pmc.json) and a third of that is a
// K = 3 contraction: pd_k = H_k . (s_0, s_1, 1) for the three rows of 64 models against every point -- a 16 x 16 x 4 matrix
// product per 16 models x 16 points (v_mfma_f64_16x16x4_f64, K = (s_0, s_1, 1, 0)).  Here the matrix pipe computes the three
// pd rows and the VALU keeps e_k = d_k pd_2 - pd_k, L = e_0^2 + e_1^2, the three compares and the counts: 13 instead of 18.
// Layout (guide: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D: col = l & 15, row = (l >> 4) + 4 reg):
//   a wave = 64 consecutive trials as 4 groups of 16 models; lane l feeds group g's A with coefficient (l >> 4) of row r of model
//   16 g + (l & 15), feeds B with coordinate (l >> 4) of point base + (l & 15), and receives pd_r of models 16 g + (l >> 4) + 4 i
//   (i = 0..3) at point base + (l & 15): sixteen (model, point) evaluations per lane and 16-point step, their counts summed over
//   the sixteen lanes that share (l >> 4) at the end (lane (l & 15) = 4 g + i writes model 16 g + (l >> 4) + 4 i).
// Bounds: the analysis above k_prescore holds with two changes.  (1) The matrix pipe's evaluation order of the K = 4 sum is
//   not documented; whatever it is -- fused or not -- at most 3 products and 3 additions round (the k = 3 term is exactly 0),
//   so pd_k lies within 6u A_k (1 + 3u) of the exact value: E_k = PRESCORE_CU_MFMA A_k with PRESCORE_CU_MFMA = 8u in place of 4u (the
//   thresholds P_min scale with it, every inequality of the derivation is unchanged).  (2) Instead of a per-model P_min in
//   registers, the model is scaled by the power of two s = 2^-ilogb(P_min) (exact; the test L <= T pd_2^2 is homogeneous in H)
//   and |s pd_2| is compared with the constant 2 >= s P_min: a stricter test, i.e. at most more uncertain points.  Entries that
//   underflow under s < 1 are below 2^-1022 where |s pd_2| >= 2 is required of a classified point: a perturbation of 2^-1008
//   relative, inside the slack between 6u and 8u.  NaN / Inf from an overflowing scale make every compare false (uncertain).
//   A model whose classification is off (P_min = NaN), a slot without a model: A = 0, nothing is ever classified.
// Pairs whose points do not fit the LDS take the VALU loop of k_prescore<H> (same results; rigorous bounds either way).
#define PRESCORE_CU_MFMA 0x1p-50 /* 8u */
typedef double dsm_f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64, 4) void k_prescore_h_mfma(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* spts = reinterpret_cast<double*>(smem_raw);  // min(n_max, VP_LDS_PTS) x 4 doubles
  const uint32_t pl = blockIdx.x;
  const uint32_t pi = p.pair0 + pl;
  const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM_H;
  if (!fs->active) return;
  const int lane = threadIdx.x;
  const int nb = (int)fs->nb;
  const int t0 = (int)blockIdx.y * 64;
  if (t0 >= nb) return;
  const uint64_t moff = p.match_off[pi];
  const int n = (int)(p.match_off[pi + 1] - moff);
  const double* gpts = p.pts_px + 4 * moff;
  const bool in_lds = n <= VP_LDS_PTS;
  const double T = p.opt.max_error * p.opt.max_error;
  const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
  const double* gmod = p.models + (size_t)pl * p.batch * 9;
  int32_t* counts = p.counts + (size_t)pl * p.batch;
  double* sums = p.sums + (size_t)pl * p.batch;
  // ---- lane = model: its P_min and the scale that brings it into [1, 2)
  const int t = t0 + lane;
  const bool has_slot = t < nb;
  const bool has_model = has_slot && nmod[has_slot ? t : 0] > 0;
  double M[9];
  for (int k = 0; k < 9; ++k) M[k] = has_model ? gmod[(size_t)t * 9 + k] : 0.0;
  double mx[4];
  stage_points_with_maxima(gpts, n, in_lds, spts, lane, mx);
  if (!in_lds) {  // the VALU form (k_prescore<H>'s loop over wave-uniform global addresses)
    int lb = 0, sure_out = n + 1;
    if (has_model) {
      const PreBounds b = prescore_bounds<FAM_H>(M, mx, T);
      sure_out = 0;
#pragma unroll 4
      for (int i = 0; i < n; ++i) prescore_point<FAM_H>(M, b, gpts + (size_t)i * 4, lb, sure_out);
    }
    if (has_slot) {
      counts[t] = n - sure_out;
      reinterpret_cast<int32_t*>(sums + t)[0] = lb;
    }
    return;
  }
  double scale = 0.0;
  {
    const double t_ok = (T >= 0x1p-6) && (T <= 0x1p40);
    const bool x_ok = (mx[0] <= 0x1p14) && (mx[1] <= 0x1p14) && (mx[2] <= 0x1p14) && (mx[3] <= 0x1p14);
    const double A0 = fabs(M[0]) * mx[0] + fabs(M[1]) * mx[1] + fabs(M[2]);
    const double A1 = fabs(M[3]) * mx[0] + fabs(M[4]) * mx[1] + fabs(M[5]);
    const double A2 = fabs(M[6]) * mx[0] + fabs(M[7]) * mx[1] + fabs(M[8]);
    const double E01 = PRESCORE_CU_MFMA * fmax(A0, A1), E2 = PRESCORE_CU_MFMA * A2;
    const double pmin = fmax(fmax(0x1p34 * E2, 0x1p26 * E01), 0x1p-400);
    if (has_model && t_ok != 0.0 && x_ok && pmin <= 0x1p300) scale = ldexp(1.0, -ilogb(pmin));  // (NaN fails pmin <= ...: scale 0)
  }
  // ---- A operands: coefficient kq = lane >> 4 of row r of model 16 g + (lane & 15), scaled (kq = 3: the zero column)
  const int kq = lane >> 4, jq = lane & 15;
  double a[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ta = t0 + 16 * g + jq;
    const double sg = __shfl(scale, 16 * g + jq);
    const bool live = kq < 3 && sg != 0.0;  // (sg = 0: no slot / no model / classification off -- nothing is read)
#pragma unroll
    for (int r = 0; r < 3; ++r) a[g][r] = live ? gmod[(size_t)ta * 9 + 3 * r + kq] * sg : 0.0;
  }
  const double t_lo = T * (1.0 - PRESCORE_DELTA), t_hi = T * (1.0 + PRESCORE_DELTA);
  const dsm_f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  uint32_t cnt[4][4];  // low half: surely inliers, high half: surely outliers (n <= VP_LDS_PTS < 65 536)
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) cnt[g][i] = 0u;
  for (int base = 0; base < n; base += 16) {
    const int pt = base + jq;
    const bool valid = pt < n;
    const double* q = spts + (size_t)(valid ? pt : n - 1) * 4;
    const double sv = q[kq & 1];  // (lanes 32..63 read a coordinate they do not use: no divergence around the LDS read)
    const double bq = kq < 2 ? sv : (kq == 2 ? 1.0 : 0.0);
    const double d0 = q[2], d1 = q[3];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const dsm_f64x4 pd0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][0], bq, zero4, 0, 0, 0);
      const dsm_f64x4 pd1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][1], bq, zero4, 0, 0, 0);
      const dsm_f64x4 pd2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][2], bq, zero4, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double e0 = __builtin_fma(d0, pd2[i], -pd0[i]);
        const double e1 = __builtin_fma(d1, pd2[i], -pd1[i]);
        const double L = __builtin_fma(e0, e0, e1 * e1);
        const double q2 = pd2[i] * pd2[i];
        const bool ok = valid && fabs(pd2[i]) >= 2.0;
        const bool sure_in = ok && (L <= t_lo * q2);
        const bool sure_out = ok && (L >= t_hi * q2);
        cnt[g][i] += (sure_in ? 1u : 0u) + (sure_out ? 0x10000u : 0u);
      }
    }
  }
  // ---- totals over the sixteen lanes of a row group; lane (lane & 15) = 4 g + i writes model 16 g + (lane >> 4) + 4 i
  uint32_t mine = 0u;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t v = cnt[g][i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
      if (jq == 4 * g + i) mine = v;
    }
  const int mw = 16 * (jq >> 2) + kq + 4 * (jq & 3);
  const int tw = t0 + mw;
  if (tw < nb) {
    const bool model_w = nmod[tw] > 0;
    counts[tw] = model_w ? n - (int)(mine >> 16) : -1;  // -1: no model in this slot
    reinterpret_cast<int32_t*>(sums + tw)[0] = model_w ? (int)(mine & 0xffffu) : 0;
  }
}

#endif  // DSM_CHECK_BUILD

// dynamic LDS: the points (as k_score) + the list of the slots to score exactly (uint16 each, batch * MAXM of them)
#ifndef K_SCORE_NEEDED_WAVES
#define K_SCORE_NEEDED_WAVES 4
#endif
template <int FAM>
__global__ __launch_bounds__(64, K_SCORE_NEEDED_WAVES) void k_score_needed(const VerifyParams p) {
  typedef Fam<FAM> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* spts = reinterpret_cast<double*>(smem_raw);
  const int lane = threadIdx.x;
  for (uint32_t pl = blockIdx.x; pl < p.n_chunk; pl += gridDim.x) {
    const uint32_t pi = p.pair0 + pl;
    const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
    if (!fs->active) continue;
    const int nb = (int)fs->nb;
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    const double* gpts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
    const bool in_lds = n <= VP_LDS_PTS;
    uint16_t* list = reinterpret_cast<uint16_t*>(spts + (size_t)(in_lds ? n : 0) * 4);
    __syncthreads();  // the previous pair's readers are done with the LDS
    if (in_lds) (void)stage_points_batched<true, false>(gpts, 4 * n, spts, lane);
    int32_t* counts = p.counts + (size_t)pl * p.batch * F::MAXM;
    double* sums = p.sums + (size_t)pl * p.batch * F::MAXM;
    const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
    const int n_slots = nb * F::MAXM;
    int carry = (int)fs->rep.num_inliers;  // the best count of the earlier rounds (0 in the first)
    int n_list = 0;
    for (int base = 0; base < n_slots; base += 64) {
      const int slot = base + lane;
      const bool in_range = slot < n_slots;
      int ub = in_range ? counts[slot] : -1;
      // (k_prescore_compact writes the slots that hold a model and nothing else)
      if (F::MAXM > 1 && in_range && (slot % F::MAXM) >= nmod[slot / F::MAXM]) ub = -1;
      const int lb = (in_range && ub >= 0) ? reinterpret_cast<const int32_t*>(sums + slot)[0] : 0;
      int incl = lb;  // inclusive running maximum of the lower bounds along the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl = max(incl, v);
      }
      int excl = __shfl_up(incl, 1);
      if (lane == 0) excl = 0;
      const int before = max(carry, excl);  // what some earlier model has surely reached
      const bool skippable = ub >= 0 && ub < before;
      // bit 2 of score_prefilter (DSM_SCORE_PREFILTER=check): every slot is scored exactly and its bounds are CHECKED
      // against the exact count below; [15] counts the slots the filter would have skipped
      const bool check = (p.score_prefilter & 4) != 0;
      if (check && skippable) atomicAdd(p.active_count + 15, 1u);
      const bool needed = ub >= 0 && (ub >= before || check);
      const unsigned long long mask = __ballot(needed);
      if (needed) list[n_list + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)slot;
      if (in_range && ub >= 0 && !needed) {  // cannot change anything: fewer inliers than the best support at that point
        counts[slot] = 0;
        sums[slot] = 0.0;
      }
      n_list += __popcll(mask);
      carry = max(carry, __shfl(incl, 63));
    }
    __syncthreads();  // points and list visible
    double max_residual = p.opt.max_error * p.opt.max_error;
    if (FAM == FAM_E) {
      const double max_error = (image_to_world_threshold(p.cams[p.pairs[2 * pi]], p.opt.max_error) +
                                image_to_world_threshold(p.cams[p.pairs[2 * pi + 1]], p.opt.max_error)) / 2;
      max_residual = max_error * max_error;
    }
    // the listed slots: a lane per slot while at least 48 are left (a full wave's walk: ~46 000 cycles for up to 64 slots), then a QUAD
    // of lanes per slot, sixteen slots per trip (~14 000 cycles each).  A pair of config 2 lists 15 - 20 of its slots: a lane per slot
    // left three quarters of the wave idle through a 256-point walk (exact_support_quad's note); at a 0.25 inlier ratio the lists
    // are long and the full-wave walk is the cheaper one.
    int done = 0;
    for (; n_list - done >= 48; done += 64) {
      if (done + lane < n_list) {
        const int slot = list[done + lane];
        const double* gm = p.models + (size_t)pl * p.batch * F::MAXM * 9 + (size_t)slot * 9;
        double M[9];
        for (int k = 0; k < 9; ++k) M[k] = gm[k];
        int cnt;
        double sum;
        exact_support<FAM>(M, n, in_lds, spts, gpts, max_residual, cnt, sum);
        if (p.score_prefilter & 4) {  // lower bound <= exact count <= upper bound, or the margins of k_prescore are wrong
          const int ub = counts[slot], lb = reinterpret_cast<const int32_t*>(sums + slot)[0];
          if (cnt < lb || cnt > ub) atomicAdd(p.active_count + 14, 1u);
        }
        counts[slot] = cnt;
        sums[slot] = sum;
      }
    }
    for (int base = done; base < n_list; base += 16) {
      const int e = base + (lane >> 2);
      const bool on = e < n_list;
      const int slot = list[on ? e : base];
      const double* gm = p.models + (size_t)pl * p.batch * F::MAXM * 9 + (size_t)slot * 9;
      double M[9];
      for (int k = 0; k < 9; ++k) M[k] = gm[k];
      int cnt;
      double sum;
      exact_support_quad<FAM>(M, n, in_lds, spts, gpts, max_residual, lane, cnt, sum);
      if (on && (lane & 3) == 0) {
        if (p.score_prefilter & 4) {  // lower bound <= exact count <= upper bound, or the margins of k_prescore are wrong
          const int ub = counts[slot], lb = reinterpret_cast<const int32_t*>(sums + slot)[0];
          if (cnt < lb || cnt > ub) atomicAdd(p.active_count + 14, 1u);
        }
        counts[slot] = cnt;
        sums[slot] = sum;
      }
    }
  }
}

// E family: the minimal solve is split in two kernels.  One lane per hypothesis keeps ~5 KB of matrices in
// scratch memory, and with thousands of resident waves that scratch lives in HBM; two thirds of the time
// went into the companion-matrix eigenvalue iteration alone.  k_solve_e_build (null space + constraint matrix,
// to global memory) and k_solve_e_lu (pivoted elimination in LDS, determinant polynomial) park (Eb, B,
// coefficients) in the hypothesis' model slot; k_roots_e runs the eigenvalue iteration with the 10 x 10 companion
// matrix in registers (pr_hessenberg_eigenvalues: static indices, the lane's deflation window as predicates; the
// round-2 LDS form -- element e of lane l at T[e * 64 + l], 51 KB per wave -- is k_roots_e_lds); k_models_score_e
// builds and scores the models at full occupancy.  Same operations, same order.
#define EPOLY_EB 0
// the 10 x 20 constraint matrix of trial t of pair pl: blocks of 64 trials interleaved, element e of the trial at block + e * 64 +
// t % 64 (batch is a multiple of 64)
template <bool COMPACT>
DSM_DEV double* e_work_of(const VerifyParams& p, uint32_t pl, int t) {
  if (COMPACT) return p.e_work + ((size_t)pl * p.batch + (size_t)(t & ~63)) * 200 + (t & 63);
  return p.e_work + ((size_t)pl * p.batch + (size_t)blockIdx.y * 64) * 200 + threadIdx.x;  // (pair grid: a wave-uniform base + lane)
}
#define EPOLY_B 36
#define EPOLY_COEFFS 75
template <bool COMPACT>
__global__ __launch_bounds__(64, 2) void k_solve_e_build(const VerifyParams p) {
  uint32_t pl;
  int t;
  if (!hyp_of_lane<COMPACT>(p, FAM_E, pl, t)) return;
  const uint32_t pi = p.pair0 + pl;
  const double* pts = p.pts_norm + 4 * p.match_off[pi];
  const uint32_t* smp = p.samples + ((size_t)pl * p.batch + t) * 7;
  double xs[20];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double* q = pts + (size_t)smp[i] * 4;
    xs[i * 4 + 0] = q[0]; xs[i * 4 + 1] = q[1]; xs[i * 4 + 2] = q[2]; xs[i * 4 + 3] = q[3];
  }
  double Eb[36];
  five_point_basis_reg(xs, Eb);
  double* slot = p.models + ((size_t)pl * p.batch + t) * 90;
  for (int k = 0; k < 36; ++k) slot[EPOLY_EB + k] = Eb[k];
  // the wave's 64 hypotheses interleaved: element e of lane l at block + e * 64 + l (batch is a multiple of 64)
  five_point_build_A<64>(Eb, e_work_of<COMPACT>(p, pl, t));
}

// A[:, :10].partialPivLu().solve(A[:, 10:]) (essential_matrix.cc:80) with the 10 x 10 factor in lane-interleaved
// LDS (element (i, k) of lane l at Al[(k*10 + i)*64 + l]; the pivoted row swaps index it dynamically, which
// would otherwise force it into scratch memory), the right-hand sides one column at a time in registers.
// Only rows 4..9 of the solution enter B(z), so the back substitution stops there.  Same operations in the
// same order as pl_lu_solve_10.
#define ELU_SMEM (100 * 64 * 8 + 10 * 64)
// Ag: the hypothesis' 10 x 20 constraint matrix A[r*20 + c]; slot: its 90-double record (B(z) and the determinant
// polynomial are written to it); Al / idx: this lane's column of the lane-interleaved LDS work area.
// ES: element stride of Ag (see five_point_build_A)
template <int ES = 1>
DSM_DEV void e_lu_body(double* Ag_rw, double* slot, double* Al, unsigned char* idx) {
  const double* Ag = Ag_rw;
  double* Sg = Ag_rw;  // the solution's way out of the registers, see below
#define LA(i, k) Al[((k) * 10 + (i)) * 64]
  LSEC_BEGIN2();
  for (int r = 0; r < 10; ++r)
    for (int c = 0; c < 10; ++c) LA(r, c) = Ag[(r * 20 + c) * ES];
  for (int i = 0; i < 10; ++i) idx[i * 64] = (unsigned char)i;
  for (int k = 0; k < 10; ++k) {
    int r = k;
    double best = fabs(LA(k, k));
    for (int i = k + 1; i < 10; ++i) {
      const double a = fabs(LA(i, k));
      if (a > best) {
        best = a;
        r = i;
      }
    }
    if (best != 0.0) {
      if (r != k) {
        for (int j = 0; j < 10; ++j) {
          const double tmp = LA(k, j);
          LA(k, j) = LA(r, j);
          LA(r, j) = tmp;
        }
        const unsigned char ti = idx[k * 64];
        idx[k * 64] = idx[r * 64];
        idx[r * 64] = ti;
      }
      const double piv = LA(k, k);
      for (int i = k + 1; i < 10; ++i) LA(i, k) /= piv;
    }
    for (int j = k + 1; j < 10; ++j) {
      const double akj = LA(k, j);
      for (int i = k + 1; i < 10; ++i) LA(i, j) -= LA(i, k) * akj;
    }
  }
  // The rows of a right-hand side follow the pivoting, so its loads can only be issued once the factor is final.  As ONE unrolled
  // block -- the factor hoisted out of LDS into registers, the 60 solution entries accumulating in registers -- the compiler had no
  // room to keep loads in flight: the ISA showed each of the hundred loads followed by its own s_waitcnt vmcnt(0), a hundred memory
  // latencies in a row.  A ROLLED loop over the columns: the next column's ten loads are issued at the top of an iteration and
  // waited for at the top of the next, and a solved column leaves the registers at once -- into the part of the lane's A that the
  // elimination no longer needs (rows 0..5 of the left block; the right-hand sides are columns 10..19), from where the solution
  // comes back in one batch for B(z).
  int prow[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) prow[i] = ((int)idx[i * 64] * 20 + 10) * ES;
  double bn[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) bn[i] = Ag[prow[i]];
#pragma unroll 1
  for (int j = 0; j < 10; ++j) {
    double b[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = bn[i];
    const int jn = j < 9 ? j + 1 : 9;  // (the last iteration re-reads its own column: no branch around the loads)
#pragma unroll
    for (int i = 0; i < 10; ++i) bn[i] = Ag[prow[i] + jn * ES];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      double s = b[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= LA(i, k) * b[k];
      b[i] = s;
    }
#pragma unroll
    for (int i = 9; i >= 4; --i) {
      double s = b[i];
#pragma unroll
      for (int k = 9; k > i; --k) s -= LA(i, k) * b[k];
      b[i] = s / LA(i, i);
    }
#pragma unroll
    for (int i = 4; i < 10; ++i) Sg[((i - 4) * 20 + j) * ES] = b[i];
  }
  double S[60];  // S[(r-4)*10 + c] = solution(r, c), rows 4..9
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c < 10; ++c) S[r * 10 + c] = Sg[(r * 20 + c) * ES];
  }
  LSEC_END2(9);
#undef LA
  double B[39], coeffs[11];
  five_point_B_det(S, B, coeffs);
  for (int k = 0; k < 39; ++k) slot[EPOLY_B + k] = B[k];
  for (int k = 0; k < 11; ++k) slot[EPOLY_COEFFS + k] = coeffs[k];
}
// The same elimination with the 10 x 10 matrix in REGISTERS (round 6).  e_lu_body keeps the lane's matrix in lane-interleaved LDS
// because the pivot row is the lane's own (a dynamic index): 51 KB per wave, three waves per CU, every element access an LDS round trip
// -- 8.1 ms per step at a VALU issue rate of 0.17, and a third of what the 5-point pipeline costs at a 0.25 inlier ratio.  Here every
// index is a compile-time constant and the lane's pivot only appears in PREDICATES (pr_colpiv_qr9's construction): the row swap is a
// chain of predicated exchanges over the candidate rows, the factor stays in 200 VGPRs (one wave per SIMD, no LDS at all), the ten
// right-hand sides follow the pivoting as ten loads by permuted row index -- the next column's issued while this one is solved, as
// before.  Same operations on the same values in the same order per element: PartialPivLU's unblocked elimination (first largest
// |a| of the column, whole rows exchanged, the column divided by the pivot, one rank-1 update per k), unit-lower forward
// substitution, backward substitution for the six rows B(z) needs.
template <int ES = 1>
DSM_DEV void e_lu_body_reg(double* Ag_rw, double* slot) {
  const double* Ag = Ag_rw;
  double* Sg = Ag_rw;  // the solution's way out of the registers (rows 0..5 of the left block, dead by then), as e_lu_body
  double a[10][10];
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#pragma unroll
    for (int c = 0; c < 10; ++c) a[r][c] = Ag[(r * 20 + c) * ES];
  }
  int perm[10];  // perm[i]: the original row now in row i
#pragma unroll
  for (int i = 0; i < 10; ++i) perm[i] = i;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    int r = k;
    double best = fabs(a[k][k]);
#pragma unroll
    for (int i = k + 1; i < 10; ++i) {
      const double v = fabs(a[i][k]);
      const bool up = v > best;
      best = up ? v : best;
      r = up ? i : r;
    }
    if (best != 0.0) {
      // exchange rows k and r (r == k: nothing moves): unconditional stores of selected VALUES
#pragma unroll
      for (int i = k + 1; i < 10; ++i) {
        const bool sw = r == i;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const double x = a[k][j], y = a[i][j];
          a[k][j] = sw ? y : x;
          a[i][j] = sw ? x : y;
        }
        const int pk = perm[k], pi = perm[i];
        perm[k] = sw ? pi : pk;
        perm[i] = sw ? pk : pi;
      }
      const double piv = a[k][k];
#pragma unroll
      for (int i = k + 1; i < 10; ++i) a[i][k] /= piv;
    }
#pragma unroll
    for (int j = k + 1; j < 10; ++j) {
      const double akj = a[k][j];
#pragma unroll
      for (int i = k + 1; i < 10; ++i) a[i][j] -= a[i][k] * akj;
    }
  }
  int prow[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) prow[i] = (perm[i] * 20 + 10) * ES;
#pragma unroll 1
  for (int j = 0; j < 10; ++j) {
    double b[10];  // (no prefetch of the next column: two waves per SIMD hide the loads, and the twenty registers are what lets two fit)
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = Ag[prow[i] + j * ES];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      double sacc = b[i];
#pragma unroll
      for (int k = 0; k < i; ++k) sacc -= a[i][k] * b[k];
      b[i] = sacc;
    }
#pragma unroll
    for (int i = 9; i >= 4; --i) {
      double sacc = b[i];
#pragma unroll
      for (int k = 9; k > i; --k) sacc -= a[i][k] * b[k];
      b[i] = sacc / a[i][i];
    }
#pragma unroll
    for (int i = 4; i < 10; ++i) Sg[((i - 4) * 20 + j) * ES] = b[i];
  }
  double S[60];  // S[(r-4)*10 + c] = solution(r, c), rows 4..9
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c < 10; ++c) S[r * 10 + c] = Sg[(r * 20 + c) * ES];
  }
  double B[39], coeffs[11];
  five_point_B_det(S, B, coeffs);
  for (int k = 0; k < 39; ++k) slot[EPOLY_B + k] = B[k];
  for (int k = 0; k < 11; ++k) slot[EPOLY_COEFFS + k] = coeffs[k];
}
__global__ __launch_bounds__(64, 2) void k_solve_e_lu_reg(const VerifyParams p) {
  uint32_t pl;
  int t;
  if (!hyp_of_lane<false>(p, FAM_E, pl, t)) return;  // (always the pair grid: launch_vp_solve_score)
  e_lu_body_reg<64>(e_work_of<false>(p, pl, t), p.models + ((size_t)pl * p.batch + t) * 90);
}
__global__ __launch_bounds__(64) void k_solve_e_lu(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Al = reinterpret_cast<double*>(smem_raw) + threadIdx.x;
  unsigned char* idx = smem_raw + 100 * 64 * 8 + threadIdx.x;  // idx[i*64]: original row now in row i
  uint32_t pl;
  int t;
  if (!hyp_of_lane<false>(p, FAM_E, pl, t)) return;
  e_lu_body<64>(e_work_of<false>(p, pl, t), p.models + ((size_t)pl * p.batch + t) * 90, Al, idx);
}

// slot coefficients -> slot roots (real parts); returns the code for nmodels: bits 0..9 root i is real
// (essential_matrix.cc:126), bits 16..: number of roots
DSM_DEV int e_roots_body(double* slot) {
  double coeffs[11], rr[11], ri[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) coeffs[k] = slot[EPOLY_COEFFS + k];
  LSEC_BEGIN4();
  const int nroots = pr_poly_roots<11>(coeffs, rr, ri);  // companion matrix in registers
  LSEC_END4(11);
  int code = 0;
  if (nroots > 0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      if (i < nroots) {
        if (!(fabs(ri[i]) > 1e-10)) code |= 1 << i;
        slot[EPOLY_COEFFS + i] = rr[i];
      }
    }
    code |= nroots << 16;
  }
  return code;
}
// slot (Eb, B, roots) + code -> models written to `out` (which may be the slot itself: its inputs are in registers by
// then); returns their number
DSM_DEV int e_models_body(const double* slot, int code, double* out) {
  const int nroots = code >> 16;
  if (nroots <= 0) return 0;
  double Eb[36], B[39], rr[10];
#pragma unroll
  for (int k = 0; k < 36; ++k) Eb[k] = slot[EPOLY_EB + k];
#pragma unroll
  for (int k = 0; k < 39; ++k) B[k] = slot[EPOLY_B + k];
#pragma unroll
  for (int i = 0; i < 10; ++i) rr[i] = slot[EPOLY_COEFFS + i];  // entries beyond the root count are never selected
  return five_point_models_reg(Eb, B, rr, code & 0x3ff, nroots, out);
}

// roots of the determinant polynomial and the essential matrices of its real roots: slot (Eb, B, coefficients) ->
// slot models + nmodels for k_models_score_e
template <bool COMPACT>
__global__ __launch_bounds__(64, 2) void k_roots_e(const VerifyParams p) {
  uint32_t pl;
  int t;
  if (!hyp_of_lane<COMPACT>(p, FAM_E, pl, t)) return;
  // roots, then the models of the real roots in place of the hypothesis' record (its inputs are in registers by then)
  double* slot = p.models + ((size_t)pl * p.batch + t) * 90;
  const int code = e_roots_body(slot);
  p.nmodels[(size_t)pl * p.batch + t] = e_models_body(slot, code, slot);
}
#ifdef DSM_CHECK_BUILD  // cross-check schedule: libdagsfm_mi355x_check.so only
// The round-2 form of the same kernel, kept for comparison (DSM_ROOTS_LDS=1): the companion matrix of every lane in
// lane-interleaved LDS (51 KB per wave), dynamically indexed.
template <bool COMPACT>
__global__ __launch_bounds__(64) void k_roots_e_lds(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Tl = reinterpret_cast<double*>(smem_raw) + threadIdx.x;  // this lane's companion matrix, stride 64
  uint32_t pl;
  int t;
  if (!hyp_of_lane<COMPACT>(p, FAM_E, pl, t)) return;
  double* slot = p.models + ((size_t)pl * p.batch + t) * 90;
  double coeffs[11], rr[11], ri[11];
  for (int k = 0; k < 11; ++k) coeffs[k] = slot[EPOLY_COEFFS + k];
  const int nroots = pl_poly_roots<11, 64>(coeffs, 11, rr, ri, Tl);
  int code = 0;
  if (nroots > 0) {
    for (int i = 0; i < nroots; ++i) {
      if (!(fabs(ri[i]) > 1e-10)) code |= 1 << i;
      slot[EPOLY_COEFFS + i] = rr[i];
    }
    code |= nroots << 16;
  }
  p.nmodels[(size_t)pl * p.batch + t] = e_models_body(slot, code, slot);
}

#endif  // DSM_CHECK_BUILD

// The bound step with a lane per MODEL in compacted order.  A five-point hypothesis has 0..10 models (4.6 on average), a seven-point one
// 1 or 3 (2.46): a lane per (trial, model) SLOT leaves 54 % / 18 % of the lanes without one.  Here a workgroup takes 64 consecutive
// entries of the pair's models in COMPACTED order -- the prefix sums of nmodels[] over the round's trials, recomputed by every
// workgroup of the pair (at most 1 024 trials: sixteen shuffle scans) -- and a lane scores its model against all correspondences
// exactly like k_prescore (points broadcast from LDS).  For E this replaces k_models_score_e, a wave-wide pass with its reductions per
// model (kept as DSM_SCORE_PREFILTER=3 / =0, where F takes the slot-per-lane k_prescore).  Writes the slot's upper bound into counts[]
// and its lower bound into the low word of sums[]; k_score_needed follows and reads which slots hold a model from nmodels[].
template <int FAM>
__global__ __launch_bounds__(64, 8) void k_prescore_compact(const VerifyParams p) {
  typedef Fam<FAM> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ uint16_t s_map[64];
  double* spts = reinterpret_cast<double*>(smem_raw);  // min(n_max, VP_LDS_PTS) x 4 doubles
  const uint32_t pl = blockIdx.x;
  const uint32_t pi = p.pair0 + pl;
  const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
  if (!fs->active) return;
  const int lane = threadIdx.x;
  const int nb = (int)fs->nb;
  const int c0 = (int)blockIdx.y * 64;  // first compacted entry of this workgroup
  if (c0 >= nb * F::MAXM) return;
  const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
  int run = 0;
  for (int base = 0; base < nb; base += 64) {
    const int t = base + lane;
    const int nm = t < nb ? nmod[t] : 0;
    int incl = nm;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    const int off = run + incl - nm;  // compacted index of this hypothesis' first model
    if (off < c0 + 64 && off + nm > c0) {
      for (int m = 0; m < nm; ++m) {
        const int c = off + m - c0;
        if (c >= 0 && c < 64) s_map[c] = (uint16_t)(t * F::MAXM + m);
      }
    }
    run += __shfl(incl, 63);
    if (run >= c0 + 64) break;  // (wave-uniform) the later hypotheses' models belong to later workgroups
  }
  // (the loop may stop early: `run` is then only known to be >= c0 + 64, which is all that is needed below)
  if (c0 >= run) return;  // wave-uniform: no model of the pair is left for this workgroup
  const uint64_t moff = p.match_off[pi];
  const int n = (int)(p.match_off[pi + 1] - moff);
  const double* gpts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
  const bool in_lds = n <= VP_LDS_PTS;
  double T = p.opt.max_error * p.opt.max_error;
  if (FAM == FAM_E) {
    const double max_error = (image_to_world_threshold(p.cams[p.pairs[2 * pi]], p.opt.max_error) +
                              image_to_world_threshold(p.cams[p.pairs[2 * pi + 1]], p.opt.max_error)) / 2;
    T = max_error * max_error;
  }
  double mx[4];
  stage_points_with_maxima(gpts, n, in_lds, spts, lane, mx);  // (ends with the barrier that also publishes s_map when in_lds)
  if (!in_lds) __syncthreads();
  const bool has_model = c0 + lane < run;
  const int slot = has_model ? (int)s_map[lane] : 0;
  const double* gm = p.models + ((size_t)pl * p.batch * F::MAXM + (size_t)slot) * 9;
  double M[9];
  for (int k = 0; k < 9; ++k) M[k] = has_model ? gm[k] : 0.0;
  if (!has_model) return;
  const PreBounds b = prescore_bounds<FAM>(M, mx, T);
  int lb = 0, sure_out = 0;
  if (in_lds) {
#pragma unroll 4
    for (int i = 0; i < n; ++i) prescore_point<FAM>(M, b, spts + (size_t)i * 4, lb, sure_out);
  } else {
#pragma unroll 4
    for (int i = 0; i < n; ++i) prescore_point<FAM>(M, b, gpts + (size_t)i * 4, lb, sure_out);
  }
  p.counts[(size_t)pl * p.batch * F::MAXM + slot] = n - sure_out;
  reinterpret_cast<int32_t*>(p.sums + (size_t)pl * p.batch * F::MAXM + slot)[0] = lb;
}

// ------------------------------------------------------------------------------------ the E / F bound step with an f32 first stage (round 6)
// k_prescore_compact spends 24 FP64 VALU instructions per (model, point) on the Sampson test; like k_prescore_h2 for the homographies,
// k_prescore_compact2 classifies every point first in PACKED f32 (two points per instruction: 19 v_pk_* per two points) against a band
// of +- 12.5 % around the threshold,
//   C~^2 < 0.875^2 (1 - 2^-8) T D~ -> surely in        C~^2 > 1.125^2 (1 + 2^-8) T D~ -> surely out        (both only if D~ >= g_D^2),
// and sends the few points inside the band, per lane, to the FP64 test of prescore_flags.  Error analysis (u = 2^-24; the model scaled by
// a power of two so that its largest entry is in [1, 2): the test is homogeneous in F; B_k, Bt_j, S_C = max|x2_0| B_0 + max|x2_1| B_1 + B_2
// as in prescore_bounds, on the scaled model; Bn = B_0 + B_1 + Bt_0 + Bt_1 >= the 2-norm of the four bounds):
//   every input rounded to f32 and two fused roundings per row: |g~_k - g*_k| <= 4.03u B_k <= 5u B_k, |h~_j - h*_j| <= 5u Bt_j;
//   C~ = fma(x2_0, g~_0, fma(x2_1, g~_1, g~_2)): |C~ - C*| <= 6.01u (n_0 B_0 + n_1 B_1) + 5u B_2 + 2.02u S_C <= E_C = 8.1u S_C;
//   sqrt(D~) within e = 5u Bn of sqrt(D*) before, and 2.03u relative after, the four roundings of the sum of squares.
//   out:  C~^2 > fl(K_out D~), K_out >= 1.1272^2 T  =>  |C*| >= 1.1272 sqrt(T) (1 - 2u) (sqrt(D*) (1 - 2.03u) - e) - E_C
//         >= 1.004 sqrt(T D*)   once   sqrt(D*) >= 9.16 e + 8.13 E_C / sqrt(T);
//   in:   C~^2 < fl(K_in D~),  K_in <= 0.8733^2 T   =>  |C*| <= 0.8733 sqrt(T) (1 + 2u) (sqrt(D*) (1 + 2.03u) + e) + E_C
//         <= 0.996 sqrt(T D*)   once   sqrt(D*) >= 7.12 e + 8.15 E_C / sqrt(T).
//   The guard on the COMPUTED value, D~ >= g_D^2 with g_D = max(1.01 (10.2 e + 8.2 E_C / sqrt(T)), 2^-40), covers both (sqrt(D*) >= (sqrt(D~) -
//   e) / (1 + 2.03u)); it is > 2^6 x the FP64 test's own precondition D >= D_min (sqrt: 2^-25 Bn, 2^-34 S_C / sqrt(T)), under which the
//   reference's evaluation is within 2^-14 relative of the exact residual (analysis above k_prescore): an exact |C*| <= 0.996 sqrt(T D*)
//   is an inlier of the reference, one >= 1.004 sqrt(T D*) an outlier.  f32 range: coordinates below 2^14 and T in [2^-40, 2^40] (else the
//   f32 stage is off) keep every quantity of a guarded point normal -- D~ >= 2^-80, K D~ >= 2^-121 -- except C~^2, which may flush to zero
//   for |C~| < 2^-63: "in" then, and rightly (0.996 sqrt(T D*) >= 4 E_C + 0.49 x 2^-60).  Entries and results that flush to zero move g, h,
//   C by at most 2^-111: relative 2^-70 of the guard.  Inf / NaN make every compare false: the point goes to the FP64 test.
// tools/check_score_bounds.py (DSM_SCORE_PREFILTER=check) holds every slot's exact count against [lower, upper]; DSM_SCORE_PREFILTER=33
// (check build) runs the pure FP64 k_prescore_compact instead.
#ifndef K_PRESCORE_C2_WAVES
#define K_PRESCORE_C2_WAVES 5  // 96 VGPRs + 24 bytes of scratch per lane: 12.5 + 10.5 ms against 13.1 + 10.8 at four waves per SIMD (100 VGPRs); six spill the stretch loop
#endif
template <int FAM>
__global__ __launch_bounds__(64, K_PRESCORE_C2_WAVES) void k_prescore_compact2(const VerifyParams p) {
  typedef Fam<FAM> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ uint16_t s_map[64];
  float* s32 = reinterpret_cast<float*>(smem_raw);  // the points as f32 pairs (16 bytes per point), then the lanes' lists
  const uint32_t pl = blockIdx.x;
  const uint32_t pi = p.pair0 + pl;
  const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
  if (!fs->active) return;
  const int lane = threadIdx.x;
  const int nb = (int)fs->nb;
  const int c0 = (int)blockIdx.y * 64;  // first compacted entry of this workgroup
  if (c0 >= nb * F::MAXM) return;
  const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
  int run = 0;
  for (int base = 0; base < nb; base += 64) {
    const int t = base + lane;
    const int nm = t < nb ? nmod[t] : 0;
    int incl = nm;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    const int off = run + incl - nm;  // compacted index of this hypothesis' first model
    if (off < c0 + 64 && off + nm > c0) {
      for (int m = 0; m < nm; ++m) {
        const int c = off + m - c0;
        if (c >= 0 && c < 64) s_map[c] = (uint16_t)(t * F::MAXM + m);
      }
    }
    run += __shfl(incl, 63);
    if (run >= c0 + 64) break;  // (wave-uniform) the later hypotheses' models belong to later workgroups
  }
  if (c0 >= run) return;  // wave-uniform: no model of the pair is left for this workgroup
  const uint64_t moff = p.match_off[pi];
  const int n = (int)(p.match_off[pi + 1] - moff);
  const double* gpts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
  const bool in_lds = n <= VP_LDS_PTS;
  double T = p.opt.max_error * p.opt.max_error;
  if (FAM == FAM_E) {
    const double max_error = (image_to_world_threshold(p.cams[p.pairs[2 * pi]], p.opt.max_error) +
                              image_to_world_threshold(p.cams[p.pairs[2 * pi + 1]], p.opt.max_error)) / 2;
    T = max_error * max_error;
  }
  double mx[4];
  stage_points_with_maxima(gpts, n, false, nullptr, lane, mx);
  const int npair = (n + 1) / 2;
  uint16_t* lst = reinterpret_cast<uint16_t*>(s32 + (size_t)8 * (in_lds ? npair : 0));  // entry k of lane l at [k * 64 + l]
  if (in_lds) {
    // the points as f32, two points per 32-byte record: (x1_0a, x1_0b, x1_1a, x1_1b, x2_0a, x2_0b, x2_1a, x2_1b); an odd last point twice
    for (int j = lane; j < npair; j += 64) {
      const int ia = 2 * j, ib = 2 * j + 1 < n ? 2 * j + 1 : 2 * j;
      float4 lo, hi;
      lo.x = (float)gpts[ia * 4 + 0];
      lo.y = (float)gpts[ib * 4 + 0];
      lo.z = (float)gpts[ia * 4 + 1];
      lo.w = (float)gpts[ib * 4 + 1];
      hi.x = (float)gpts[ia * 4 + 2];
      hi.y = (float)gpts[ib * 4 + 2];
      hi.z = (float)gpts[ia * 4 + 3];
      hi.w = (float)gpts[ib * 4 + 3];
      reinterpret_cast<float4*>(s32)[2 * j] = lo;
      reinterpret_cast<float4*>(s32)[2 * j + 1] = hi;
    }
  }
  __syncthreads();  // (also publishes s_map)
  const bool has_model = c0 + lane < run;
  const int slot = has_model ? (int)s_map[lane] : 0;
  const double* gm = p.models + ((size_t)pl * p.batch * F::MAXM + (size_t)slot) * 9;
  double M[9];
  for (int k = 0; k < 9; ++k) M[k] = has_model ? gm[k] : 0.0;
  const PreBounds b = prescore_bounds<FAM>(M, mx, T);
  int lb = 0, sure_out = 0;
  if (!in_lds) {  // the points do not fit the LDS: the FP64 loop over wave-uniform global addresses, as k_prescore_compact
    if (has_model) {
#pragma unroll 4
      for (int i = 0; i < n; ++i) prescore_point<FAM>(M, b, gpts + (size_t)i * 4, lb, sure_out);
    }
  } else {
    // ---- the model in f32 (largest entry scaled into [1, 2)), the guard, the band's constants
    float Ff[9], gd2 = __builtin_inff(), k_out = __builtin_inff(), k_in = -1.0f;
    {
      double big = 0.0;
      for (int k = 0; k < 9; ++k) big = fmax(big, fabs(M[k]));
      const bool usable = has_model && (b.c0 == b.c0) && big >= 0x1p-900 && big <= 0x1p900 && T >= 0x1p-40 && T <= 0x1p40;  // (c0 is NaN when t_ok / x_ok fail)
      const double sc = usable ? ldexp(1.0, -ilogb(big)) : 0.0;
      double Ms[9];
      for (int k = 0; k < 9; ++k) {
        Ms[k] = M[k] * sc;
        Ff[k] = (float)Ms[k];
      }
      const double B0 = fabs(Ms[0]) * mx[0] + fabs(Ms[1]) * mx[1] + fabs(Ms[2]);
      const double B1 = fabs(Ms[3]) * mx[0] + fabs(Ms[4]) * mx[1] + fabs(Ms[5]);
      const double B2 = fabs(Ms[6]) * mx[0] + fabs(Ms[7]) * mx[1] + fabs(Ms[8]);
      const double Bt0 = fabs(Ms[0]) * mx[2] + fabs(Ms[3]) * mx[3] + fabs(Ms[6]);
      const double Bt1 = fabs(Ms[1]) * mx[2] + fabs(Ms[4]) * mx[3] + fabs(Ms[7]);
      const double e = 5.0 * 0x1p-24 * (B0 + B1 + Bt0 + Bt1);
      const double EC = 8.1 * 0x1p-24 * (mx[2] * B0 + mx[3] * B1 + B2);
      const double gD = fmax(1.01 * (10.2 * e + 8.2 * EC / sqrt(T)), 0x1p-40);
      if (usable && gD <= 0x1p40) {
        gd2 = (float)(gD * gD * 1.001) * 1.0001f;                      // rounded up
        k_out = (float)(1.265625 * T * (1.0 + 0x1p-8)) * 1.0001f;      // rounded up
        k_in = (float)(0.765625 * T * (1.0 - 0x1p-8)) * 0.9999f;       // rounded down
      }
    }
    // The WAVE walks the points in stretches: the packed-f32 loop runs until some lane's list of band points is about to fill up (or to
    // the end), then all lanes work their lists off in FP64 together and the loop goes on where it stopped.  (The first form redid ALL
    // points of a lane in FP64 once its list overflowed: with the ~500-match pairs of configs[4] -- 8 192 features -- most lanes
    // overflow, and the step was 25 % slower than pure FP64: profiles/r06_prescore_ef_f32_stage.txt.  At config 2's 256 matches a
    // list almost never fills: one stretch, one drain.)
    const dsm_f32x2 f0 = pk2(Ff[0]), f1 = pk2(Ff[1]), f2 = pk2(Ff[2]), f3 = pk2(Ff[3]), f4 = pk2(Ff[4]), f5 = pk2(Ff[5]), f6 = pk2(Ff[6]), f7 = pk2(Ff[7]),
                    f8 = pk2(Ff[8]), ko = pk2(k_out), ki = pk2(k_in);
    int j = 0;
    while (j < npair) {
      int cnt = 0;
      for (; j < npair; ++j) {
        const float4 lo = reinterpret_cast<const float4*>(s32)[2 * j], hi = reinterpret_cast<const float4*>(s32)[2 * j + 1];
        const dsm_f32x2 x10 = {lo.x, lo.y}, x11 = {lo.z, lo.w}, x20 = {hi.x, hi.y}, x21 = {hi.z, hi.w};
        const dsm_f32x2 g0 = pk_fma(f0, x10, pk_fma(f1, x11, f2));
        const dsm_f32x2 g1 = pk_fma(f3, x10, pk_fma(f4, x11, f5));
        const dsm_f32x2 g2 = pk_fma(f6, x10, pk_fma(f7, x11, f8));
        const dsm_f32x2 h0 = pk_fma(f0, x20, pk_fma(f3, x21, f6));
        const dsm_f32x2 h1 = pk_fma(f1, x20, pk_fma(f4, x21, f7));
        const dsm_f32x2 C = pk_fma(x20, g0, pk_fma(x21, g1, g2));
        const dsm_f32x2 num = C * C;
        const dsm_f32x2 D = pk_fma(g0, g0, pk_fma(g1, g1, pk_fma(h0, h0, h1 * h1)));
        const dsm_f32x2 DO = ko * D, DI = ki * D;
        const bool two = 2 * j + 1 < n;
        const bool ok_a = D.x >= gd2, ok_b = two && D.y >= gd2;
        const bool out_a = ok_a && (num.x > DO.x), out_b = ok_b && (num.y > DO.y);
        const bool in_a = ok_a && (num.x < DI.x), in_b = ok_b && (num.y < DI.y);
        sure_out += (out_a ? 1 : 0) + (out_b ? 1 : 0);
        lb += (in_a ? 1 : 0) + (in_b ? 1 : 0);
        if (has_model && !(in_a || out_a)) {  // inside the band (or unguarded): the FP64 test decides, at the end of the stretch
          lst[cnt * 64 + lane] = (uint16_t)(2 * j);
          ++cnt;
        }
        if (has_model && two && !(in_b || out_b)) {
          lst[cnt * 64 + lane] = (uint16_t)(2 * j + 1);
          ++cnt;
        }
        if (__any(cnt > PRESCORE_LIST_CAP - 2)) {  // (a trip adds at most two entries: no list passes PRESCORE_LIST_CAP)
          ++j;
          break;
        }
      }
      // ---- the listed points in FP64, four per trip with their loads issued together
      int maxc = cnt;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
      for (int k0 = 0; k0 < maxc; k0 += 4) {
        double q[4][4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          on[u] = k0 + u < cnt;
          const double* src = gpts + (size_t)(on[u] ? lst[(k0 + u) * 64 + lane] : 0) * 4;
#pragma unroll
          for (int c = 0; c < 4; ++c) q[u][c] = src[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bool in, out;
          prescore_flags<FAM>(M, b, q[u], in, out);
          lb += (on[u] && in) ? 1 : 0;
          sure_out += (on[u] && out) ? 1 : 0;
        }
      }
    }
  }
  if (has_model) {
    p.counts[(size_t)pl * p.batch * F::MAXM + slot] = n - sure_out;
    reinterpret_cast<int32_t*>(p.sums + (size_t)pl * p.batch * F::MAXM + slot)[0] = lb;
  }
}

// the inlier counts of ALL models of the block's 64 hypotheses (built by k_roots_e) with the lanes spread over the
// correspondences (a hypothesis has 0..10 models: scoring them
// lane-per-hypothesis would run every lane as long as the one with the most models).
template <bool CHECK>  // CHECK: DSM_SCORE_PREFILTER=check -- exact counts for every model, held against the bound step's bounds
__global__ __launch_bounds__(64, 8) void k_models_score_e(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* spts = reinterpret_cast<double*>(smem_raw);  // min(n_max, VP_LDS_PTS) x 4 doubles
  const uint32_t pl = blockIdx.x;
  const uint32_t pi = p.pair0 + pl;
  const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM_E;
  if (!fs->active) return;
  const int lane = threadIdx.x;
  const int t0 = blockIdx.y * 64;
  const int t = t0 + lane;
  const int nb = (int)fs->nb;
  if (t0 >= nb) return;
  const uint64_t moff = p.match_off[pi];
  const int n = (int)(p.match_off[pi + 1] - moff);
  const double* gpts = p.pts_norm + 4 * moff;
  const bool in_lds = n <= VP_LDS_PTS;
  double mx[4];  // max |coordinate| of the pair's points (the bound step's margins, see k_prescore)
  {
    double mm = in_lds ? stage_points_batched<true, true>(gpts, 4 * n, spts, lane) : stage_points_batched<false, true>(gpts, 4 * n, spts, lane);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) mm = fmax(mm, __shfl_xor(mm, o));
#pragma unroll
    for (int c = 0; c < 4; ++c) mx[c] = __shfl(mm, c);
  }
  const dsm_camera& cam1 = p.cams[p.pairs[2 * pi]];
  const dsm_camera& cam2 = p.cams[p.pairs[2 * pi + 1]];
  const double max_error =
      (image_to_world_threshold(cam1, p.opt.max_error) + image_to_world_threshold(cam2, p.opt.max_error)) / 2;
  const double max_residual = max_error * max_error;
  const double* slots = p.models + ((size_t)pl * p.batch + t0) * 90;
  const int nm = t < nb ? p.nmodels[(size_t)pl * p.batch + t] : 0;  // models per hypothesis, written by k_roots_e
  __syncthreads();  // points (LDS) visible to the whole wave
  LSEC_BEGIN();
  int32_t* counts = p.counts + ((size_t)pl * p.batch + t0) * 10;
  const int ntr = (nb - t0) < 64 ? (nb - t0) : 64;
  // bound + exact as for F and H (k_prescore), fused here because the decision is wave-uniform: a model whose UPPER bound
  // is below a count that an earlier model of this block (or an earlier round's best) has reached cannot change
  // anything in the replay; it is written as 0 inliers and its exact count is never taken.  n < 65 536 for the packing.
  const bool prefilter = p.score_prefilter != 0 && n < 65536;
  int run_max = (int)fs->rep.num_inliers;
  for (int tt = 0; tt < ntr; ++tt) {
    const int nmt = __builtin_amdgcn_readlane(nm, tt);
    for (int m = 0; m < nmt; ++m) {
      const double* Mg = slots + (size_t)tt * 90 + m * 9;
      double M[9];
      for (int k = 0; k < 9; ++k) M[k] = Mg[k];
      int ub_chk = n, lb_chk = 0;
      if (prefilter) {
        const PreBounds b = prescore_bounds<FAM_E>(M, mx, max_residual);
        int lb = 0, so = 0;  // wave totals: a ballot + scalar population count per 64 points instead of per-lane counters + a reduction
        if (in_lds) {  // (two loops: a pointer that may be LDS or global compiles to flat loads)
          for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bool fin = false, fout = false;
            if (i < n) prescore_flags<FAM_E>(M, b, spts + (size_t)i * 4, fin, fout);
            so += (int)__popcll(__ballot(fout));
            if constexpr (CHECK) lb += (int)__popcll(__ballot(fin));
          }
        } else {
          for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bool fin = false, fout = false;
            if (i < n) prescore_flags<FAM_E>(M, b, gpts + (size_t)i * 4, fin, fout);
            so += (int)__popcll(__ballot(fout));
            if constexpr (CHECK) lb += (int)__popcll(__ballot(fin));
          }
        }
        if constexpr (CHECK) {
          ub_chk = n - so;
          lb_chk = lb;
        }
        if (n - so < run_max) {  // wave-uniform
          if constexpr (!CHECK) {
            if (lane == 0) counts[tt * 10 + m] = 0;
            continue;
          } else {
            if (lane == 0) atomicAdd(p.active_count + 15, 1u);  // would have been skipped
          }
        }
      }
      int cnt = 0;
      if (in_lds) {  // (apart: a pointer that may be LDS or global compiles to flat loads)
        for (int base = 0; base < n; base += 64) {
          const int i = base + lane;
          cnt += (int)__popcll(__ballot(i < n && fam_residual<FAM_E>(M, spts + (size_t)(i < n ? i : 0) * 4) <= max_residual));
        }
      } else {
        for (int base = 0; base < n; base += 64) {
          const int i = base + lane;
          cnt += (int)__popcll(__ballot(i < n && fam_residual<FAM_E>(M, gpts + (size_t)(i < n ? i : 0) * 4) <= max_residual));
        }
      }
      if (lane == 0) counts[tt * 10 + m] = cnt;
      if constexpr (CHECK) {
        if (lane == 0 && (cnt < lb_chk || cnt > ub_chk)) atomicAdd(p.active_count + 14, 1u);
      }
      run_max = max(run_max, cnt);
    }
  }
  LSEC_END(13);
}

#ifdef DSM_PROFILE_SECTIONS
#define TSEC_BEGIN() const long long t__0 = clock64()
#define TSEC_END(sec) do { if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(p_dbg) + (sec), (unsigned long long)(clock64() - t__0)); } while (0)
#else
#define TSEC_BEGIN() do {} while (0)
#define TSEC_END(sec) do {} while (0)
#endif


// Skip-ahead of the replay: from trial t on, the first trial that can change anything -- one with a model that would
// replace the best one (more inliers, or as many and a smaller residual sum: Compare, support_measurement.cc:52-60),
// or any trial with a model at or past the stopping threshold.  Returns its index (>= nb: none) and adds the models
// of the trials skipped over to *num_models (they are "scored" models for the report's bookkeeping).  Four 64-trial
// windows are loaded at once; count slots at and beyond a trial's model count hold stale values and are masked.
// The essential family has no precomputed sums (its scoring is wave-per-model): every tie is an event there.
template <int FAM>
DSM_DEV int replay_next_event(int t, int nb, const int32_t* nmod, const int32_t* cnts, const double* sums, uint32_t best_n,
                              double best_sum, uint32_t thr, uint32_t T0, uint32_t* num_models, int lane) {
  constexpr int MAXM = Fam<FAM>::MAXM;
  constexpr int PF = 4;
  while (t < nb) {
    int nm_k[PF];
    bool hit_k[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int tt = t + 64 * k + lane;
      nm_k[k] = 0;
      hit_k[k] = false;
      if (tt < nb) {
        const int nm_l = nmod[tt];
        int c[MAXM];
#pragma unroll
        for (int m = 0; m < MAXM; ++m) c[m] = cnts[(size_t)tt * MAXM + m];
        bool hit = false;
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
          if (m < nm_l) {
            if ((uint32_t)c[m] > best_n) hit = true;
            if ((uint32_t)c[m] == best_n) hit = hit || (FAM == FAM_E ? true : sums[(size_t)tt * MAXM + m] < best_sum);
          }
        }
        nm_k[k] = nm_l;
        hit_k[k] = hit;
      }
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int tt = t + 64 * k + lane;
      const bool ev = nm_k[k] > 0 && (hit_k[k] || (T0 + (uint32_t)tt) >= thr);
      const unsigned long long bal = __ballot(ev);
      const int f = bal ? (__ffsll((long long)bal) - 1) : 64;
      int skipped = (lane < f) ? nm_k[k] : 0;
      for (int o = 32; o > 0; o >>= 1) skipped += __shfl_xor(skipped, o);
      *num_models += (uint32_t)skipped;
      if (bal) return t + 64 * k + f;
    }
    t += 64 * PF;
  }
  return t;
}

template <int FAM>
__global__ __launch_bounds__(64, (FAM == FAM_E ? 1 : 2)) void k_replay(const VerifyParams p) {  // no VGPR spills: see k_replay_lo<TAIL>
  uint32_t* p_dbg = p.active_count + 8;
  (void)p_dbg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  VSmem* sm = reinterpret_cast<VSmem*>(smem_raw);
  typedef Fam<FAM> F;
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  // pairs are handed out dynamically (their cost varies with the number of local optimisations, and from the
  // second round on most of them are inactive): p.active_count[16] is the next pair of this launch
  __shared__ uint32_t s_next;
  WorkGrab wgrab;
  const uint32_t grain = work_grain(p.n_chunk);
  for (;;) {
    wv_sync();
    const uint32_t pl = grab_item(wgrab, p.active_count + 16, &s_next, lane, grain);
    if (pl >= p.n_chunk) break;
    const uint32_t pi = p.pair0 + pl;
    FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
    if (!fs->active) continue;
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    const dsm_camera& cam1 = p.cams[p.pairs[2 * pi]];
    const dsm_camera& cam2 = p.cams[p.pairs[2 * pi + 1]];
    PairWork w;
    w.n = n;
    w.resid = ws.resid;
    w.inl = ws.inl;
    w.tall = ws.tall;
    w.models = nullptr;
    w.sm = sm;
    w.lane = lane;
    w.pts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
    w.nt_table = p.nt_table + p.nt_off[n] + (size_t)FAM * (size_t)(n + 1);
    double max_error = p.opt.max_error;
    if (FAM == FAM_E)
      max_error = (image_to_world_threshold(cam1, p.opt.max_error) + image_to_world_threshold(cam2, p.opt.max_error)) / 2;
    const double max_residual = max_error * max_error;
    const uint32_t min_trials = (uint32_t)p.opt.min_num_trials;
    const uint32_t max_num_trials = p.max_trials[FAM];

    uint32_t best_n = fs->rep.num_inliers;
    double best_sum = fs->rep.residual_sum;
    double best_model[9];
    for (int k = 0; k < 9; ++k) best_model[k] = fs->rep.model[k];
    uint32_t dyn_max = fs->dyn_max;
    uint32_t num_models = fs->rep.num_models;
    const uint32_t T0 = fs->rep.num_trials;
    const int nb = (int)fs->nb;
    const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
    const int32_t* cnts = p.counts + (size_t)pl * p.batch * F::MAXM;
    const double* sums = p.sums + (size_t)pl * p.batch * F::MAXM;  // F and H only
    const double* mods = p.models + (size_t)pl * p.batch * F::MAXM * 9;

    bool abort = false;
    int t = 0;
    int t_stop = nb - 1;
    uint32_t trial_abs = T0;
    const long long t_pair0 = clock64();
    (void)t_pair0;
    while (t < nb) {
      // ---- skip ahead to the next trial that can change anything
      t = replay_next_event<FAM>(t, nb, nmod, cnts, sums, best_n, best_sum, dyn_max > min_trials ? dyn_max : min_trials, T0, &num_models, lane);
      if (t >= nb) break;
      trial_abs = T0 + (uint32_t)t;
      // ---- exact sequential processing of trial t (loransac.h:142-198)
      const int nm = nmod[t];
      for (int m = 0; m < nm; ++m) {
        num_models += 1;
        const uint32_t cnt = (uint32_t)cnts[(size_t)t * F::MAXM + m];
        const double* M = mods + ((size_t)t * F::MAXM + m) * 9;
        if (cnt >= best_n) {
          if (p.stats && lane == 0) atomicAdd(p.active_count + 1 + FAM * 2, 1u);
          double sum;
          if constexpr (FAM == FAM_E) {
            TSEC_BEGIN();
            score_model<FAM>(w, M, max_residual, true);
            sum = ordered_residual_sum(w, max_residual);
            TSEC_END(0);
          } else {
            sum = sums[(size_t)t * F::MAXM + m];  // k_score's in-order sum
          }
          if (cnt > best_n || (cnt == best_n && sum < best_sum)) {
            best_n = cnt;
            best_sum = sum;
            for (int k = 0; k < 9; ++k) best_model[k] = M[k];
            if (cnt > (uint32_t)F::K && cnt >= (uint32_t)F::LO_MIN) {
              if (p.stats && lane == 0) atomicAdd(p.active_count + 2 + FAM * 2, 1u);
              int ninl, nlo;
              {
                TSEC_BEGIN();
                if constexpr (FAM != FAM_E) {  // the residuals of the new best model, for the compaction
                  score_model<FAM>(w, M, max_residual, true);
                  wv_sync();
                }
                ninl = compact_inliers(w, max_residual);
                nlo = fam_local<FAM>(w, ninl);
                TSEC_END(1);
              }
              for (int l = 0; l < nlo; ++l) {
                num_models += 1;
                const uint32_t lc = (uint32_t)score_model<FAM>(w, sm->lo_models + l * 9, max_residual, true);
                const double lsum = ordered_residual_sum(w, max_residual);
                if (lc > best_n || (lc == best_n && lsum < best_sum)) {
                  best_n = lc;
                  best_sum = lsum;
                  for (int k = 0; k < 9; ++k) best_model[k] = sm->lo_models[l * 9 + k];
                }
              }
            }
            dyn_max = w.nt_table[best_n];
          }
        }
        if (trial_abs >= dyn_max && trial_abs >= min_trials) {
          abort = true;
          break;
        }
      }
      if (abort) {
        t_stop = t;
        break;
      }
      t += 1;
    }

#ifdef DSM_PROFILE_SECTIONS
    if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(p_dbg) + 2, (unsigned long long)(clock64() - t_pair0));
#endif
    uint32_t trials_done;
    bool finished;
    uint32_t* st = p.pair_state + (size_t)pi * PAIR_STATE_WORDS;
    if (abort) {
      trials_done = (trial_abs + 1 < max_num_trials) ? trial_abs + 2 : trial_abs + 1;  // loransac.h:129-134
      finished = true;
      if (lane == 0) {
        if (t_stop != nb - 1) {  // later samples of this round were speculation: resume from the snapshot
          st[PS_SKIP] = p.draws_end[(size_t)pl * p.batch + t_stop];
          st[PS_USE_SNAP] = 1;
        }
      }
    } else {
      trials_done = T0 + (uint32_t)nb;
      finished = trials_done >= max_num_trials;
    }
    if (lane == 0) {
      fs->rep.num_trials = trials_done;
      fs->rep.num_models = num_models;
      fs->rep.num_inliers = best_n;
      fs->rep.residual_sum = best_sum;
      for (int k = 0; k < 9; ++k) fs->rep.model[k] = best_model[k];
      fs->dyn_max = dyn_max;
      fs->rounds += 1;
      fs->active = finished ? 0u : 1u;
      if (!finished) atomicAdd(p.active_count, 1u);
    }
    if (finished) {
      const bool success = best_n >= (uint32_t)F::K;
      if (success) {
        score_model<FAM>(w, best_model, max_residual, true);
        wv_sync();
        unsigned char* mask = p.masks + (size_t)FAM * p.mask_stride + moff;
        for (int i = lane; i < n; i += 64) mask[i] = ws.resid[i] <= max_residual;
      }
      if (lane == 0) {
        RansacReport rep = fs->rep;
        rep.success = success;
        rep.num_trials = trials_done;
        rep.num_models = num_models;
        rep.num_inliers = best_n;
        rep.residual_sum = best_sum;
        for (int k = 0; k < 9; ++k) rep.model[k] = best_model[k];
        p.reports[(size_t)pi * 3 + FAM] = rep;
      }
    }
  }
}

// Which problems k_lo_prepare_reg takes: a tall matrix (more than nine constraint rows) whose rows fit the lanes'
// registers -- up to LOP_PPL inliers per lane.
#define LOP_PPL 6  // E / F: 384 inliers; beyond that the general kernel
// H: ONE inlier, two rows per lane -- up to 64 inliers, which is every local optimisation of a non-planar scene (its best homographies
// hold 7 - 30 of 256 matches).  (Round 3 measured the register form for H with LOP_PPL = 6 -- twelve rows per lane, 288 shuffles to
// deal the rows out -- slower than k_lo_prepare and left H to the general kernel; with one slot the same code is 9 x 2 doubles per lane.)
#define LOP_PPL_H 1
// ... and three inliers = six rows per lane (the register footprint of E / F at LOP_PPL) for the homographies of PLANAR scenes: up to 192
// inliers; beyond that the general kernel
#define LOP_PPL_H_BIG 3
template <int FAM>
__host__ __device__ inline bool lo_prepare_in_registers(int ninl) {
  const int m = FAM == FAM_H ? 2 * ninl : ninl;
  return m > 9 && ninl <= 64 * (FAM == FAM_H ? LOP_PPL_H_BIG : LOP_PPL);
}
// The register kernel exists in two sizes per family -- E / F: LOP_PPL_SMALL inliers per lane (192: 54 instead of 108 registers of matrix, three waves
// per SIMD instead of two, 15 % faster per problem) and LOP_PPL for the problems beyond that, which the replay counts ([25]) so that the
// second launch only happens when it has work
#define LOP_PPL_SMALL 3
#ifndef LOP_SMALL_WAVES
#define LOP_SMALL_WAVES 3
#endif
template <int FAM>
__host__ __device__ inline bool lo_prepare_big(int ninl) {
  return lo_prepare_in_registers<FAM>(ninl) && ninl > 64 * (FAM == FAM_H ? LOP_PPL_H : LOP_PPL_SMALL);
}

// ------------------------------------------------------------------------------------ replay with batched local optimisation
// k_replay runs a pair's local optimisations inline: ONE problem on a 64-lane wave, most of it scalar chains
// (2 x 2 rotations of the 9 x 9 Jacobi, the 5-point finish) that every lane executes redundantly -- 72 % of its
// wave cycles.  Here the replay SUSPENDS at every local optimisation: k_replay_lo records the pair's state, hands
// the ordered inlier list to lo_inl and puts the pair on a queue; the optimisation then runs for all queued pairs
// at once (k_lo_prepare_reg / k_lo_prepare: wave per pair, matrix + pivoted QR; k_lo_jacobi_reg: a lane per pair for
// the 9 x 9 sweeps, k_lo_jacobi: an 8-lane group for the few smaller problems; k_lo_finish: the 8-point / DLT finish;
// the essential family re-uses the flat 5-point kernels, lane per problem);
// the next k_replay_lo launch (work list = that queue) scores the returned models and scans on.  Same operations
// in the same order as k_replay, which stays as the reference schedule (DSM_VERIFY_INLINE_LO=1).
// TAIL: the same replay for the last few queued pairs of a round, with every local optimisation -- the pending one the
// pair was suspended at, and all later ones -- run inline by the wave (fam_local, as in k_replay): once the queue is
// short, a batched iteration costs its full chain of launches for a handful of problems, and the wave-wide solve's
// latency is the smaller price.  Never suspends.
// (TAIL at one wave per SIMD: at four, 128 VGPRs, the inlined local optimisation spills 127 - 717 VGPRs and 66 - 102 SGPRs,
// and hipcc 7.2 then corrupts an SGPR tuple spilled through VGPR lanes -- the uniform best_model[4..7] of the H family
// came back wrong on 34 of 124 750 pairs; with 512 VGPRs nothing is spilled to memory.  The tail is latency-bound anyway.)
// MODE 2 (LOOKUP): the tail of a round, parallel form: k_tail_enum listed the local optimisations the pair can still reach
// and k_tail_lo computed their outcomes (TailItem); this scan takes the outcome where k_replay would run the
// optimisation.  A step that is not among the pair's items (more than TAIL_KMAX candidates) suspends the pair as in mode 0.
template <int FAM, int MODE>
__global__ __launch_bounds__(64, (MODE == 1 ? 1 : DSM_REPLAY_WAVES)) void k_replay_lo(const VerifyParams p) {
  constexpr bool TAIL = MODE == 1;
  constexpr bool LOOKUP = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  VSmem* sm = reinterpret_cast<VSmem*>(smem_raw);
  typedef Fam<FAM> F;
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  __shared__ uint32_t s_next;
  WorkGrab wgrab;
  const uint32_t grain = work_grain(p.n_work);
  for (;;) {
    wv_sync();
    const uint32_t widx = grab_item(wgrab, p.active_count + 16, &s_next, lane, grain);
    if (widx >= p.n_work) break;
    const uint32_t pl = p.worklist ? p.worklist[widx] : widx;
    const uint32_t pi = p.pair0 + pl;
    FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
    if (!fs->active) continue;
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    const dsm_camera& cam1 = p.cams[p.pairs[2 * pi]];
    const dsm_camera& cam2 = p.cams[p.pairs[2 * pi + 1]];
    PairWork w;
    w.n = n;
    w.resid = ws.resid;
    w.inl = reinterpret_cast<int*>(p.lo_inl + moff);  // the compaction's output IS the hand-over to the LO kernels
    w.tall = ws.tall;
    w.models = nullptr;
    w.sm = sm;
    w.lane = lane;
    w.pts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
    w.nt_table = p.nt_table + p.nt_off[n] + (size_t)FAM * (size_t)(n + 1);
    double max_error = p.opt.max_error;
    if (FAM == FAM_E)
      max_error = (image_to_world_threshold(cam1, p.opt.max_error) + image_to_world_threshold(cam2, p.opt.max_error)) / 2;
    const double max_residual = max_error * max_error;
    const uint32_t min_trials = (uint32_t)p.opt.min_num_trials;
    const uint32_t max_num_trials = p.max_trials[FAM];

    uint32_t best_n = fs->rep.num_inliers;
    double best_sum = fs->rep.residual_sum;
    double best_model[9];
    for (int k = 0; k < 9; ++k) best_model[k] = fs->rep.model[k];
    uint32_t dyn_max = fs->dyn_max;
    uint32_t num_models = fs->rep.num_models;
    const uint32_t T0 = fs->rep.num_trials;
    const int nb = (int)fs->nb;
    const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
    const int32_t* cnts = p.counts + (size_t)pl * p.batch * F::MAXM;
    const double* sums = p.sums + (size_t)pl * p.batch * F::MAXM;  // F and H only
    const double* mods = p.models + (size_t)pl * p.batch * F::MAXM * 9;

    bool abort = false, suspended = false;
    int t = (int)fs->t_pos;
    int m_start = (int)fs->m_pos;
    int t_stop = nb - 1;
    uint32_t trial_abs = T0 + (uint32_t)t;
    bool in_trial = false;  // resume inside trial t at model m_start
    if (fs->lo_wait) {
      // the local optimisation of (trial t, model m_start - 1) has returned: loransac.h:160-178
      int nlo;
      if constexpr (TAIL)
        nlo = fam_local<FAM>(w, (int)fs->lo_ninl);  // the inlier list is still in lo_inl; models -> sm->lo_models
      else
        nlo = (int)fs->lo_nm;
      if constexpr (LOOKUP) {  // item 0 of the pair is the pending step
        const TailItem& it = p.tail_items[(size_t)widx * TAIL_KMAX];
        num_models += it.nlo;
        best_n = it.out_n;
        best_sum = it.out_sum;
        for (int k = 0; k < 9; ++k) best_model[k] = it.out_model[k];
        nlo = 0;
      }
      const double* glom = p.lo_models + (size_t)pl * 90;
      for (int l = 0; l < nlo; ++l) {
        num_models += 1;
        double M[9];
        for (int k = 0; k < 9; ++k) {
          if constexpr (TAIL)
            M[k] = sm->lo_models[l * 9 + k];
          else
            M[k] = glom[l * 9 + k];
        }
        uint32_t lc;
        const double lsum = score_and_sum<FAM>(w, M, max_residual, &lc);
        if (lc > best_n || (lc == best_n && lsum < best_sum)) {
          best_n = lc;
          best_sum = lsum;
          for (int k = 0; k < 9; ++k) best_model[k] = M[k];
        }
      }
      dyn_max = w.nt_table[best_n];
      if (trial_abs >= dyn_max && trial_abs >= min_trials) {
        abort = true;
        t_stop = t;
      }
      in_trial = true;
    }
    while (!abort && t < nb) {
      if (!in_trial) {
        // ---- skip ahead to the next trial that can change anything
        t = replay_next_event<FAM>(t, nb, nmod, cnts, sums, best_n, best_sum, dyn_max > min_trials ? dyn_max : min_trials, T0, &num_models, lane);
        if (t >= nb) break;
        m_start = 0;
      }
      in_trial = false;
      trial_abs = T0 + (uint32_t)t;
      // ---- exact sequential processing of trial t (loransac.h:142-198)
      const int nm = nmod[t];
      for (int m = m_start; m < nm; ++m) {
        num_models += 1;
        const uint32_t cnt = (uint32_t)cnts[(size_t)t * F::MAXM + m];
        const double* M = mods + ((size_t)t * F::MAXM + m) * 9;
        if (cnt >= best_n) {
          if (p.stats && lane == 0) atomicAdd(p.active_count + 1 + FAM * 2, 1u);
          double sum;
          if constexpr (FAM == FAM_E) {
            uint32_t cnt_again;
            sum = score_and_sum<FAM>(w, M, max_residual, &cnt_again);
          } else {
            sum = sums[(size_t)t * F::MAXM + m];  // k_score's in-order sum
          }
          if (cnt > best_n || (cnt == best_n && sum < best_sum)) {
            best_n = cnt;
            best_sum = sum;
            for (int k = 0; k < 9; ++k) best_model[k] = M[k];
            if (cnt > (uint32_t)F::K && cnt >= (uint32_t)F::LO_MIN) {
              if (p.stats && lane == 0) atomicAdd(p.active_count + 2 + FAM * 2, 1u);
              bool looked_up = false;
              if constexpr (LOOKUP) {
                const uint32_t ni = p.tail_n[widx];
                for (uint32_t k = 0; k < ni; ++k) {  // (a pending item 0 is the model the pair was suspended at: it cannot come again)
                  const TailItem& it = p.tail_items[(size_t)widx * TAIL_KMAX + k];
                  if (it.t == (uint32_t)t && it.m == (uint32_t)m) {
                    num_models += it.nlo;
                    best_n = it.out_n;
                    best_sum = it.out_sum;
                    for (int q = 0; q < 9; ++q) best_model[q] = it.out_model[q];
                    looked_up = true;
                    break;
                  }
                }
              }
              if (looked_up) {
                dyn_max = w.nt_table[best_n];
                if (trial_abs >= dyn_max && trial_abs >= min_trials) {
                  abort = true;
                  break;
                }
                continue;
              }
              if constexpr (FAM != FAM_E) {  // the residuals of the new best model, for the compaction
                score_model<FAM>(w, M, max_residual, true);
                wv_sync();
              }
              const int ninl = compact_inliers(w, max_residual);  // -> lo_inl
              if constexpr (TAIL) {
                const int nlo = fam_local<FAM>(w, ninl);
                for (int l = 0; l < nlo; ++l) {
                  num_models += 1;
                  double ML[9];
                  for (int k = 0; k < 9; ++k) ML[k] = sm->lo_models[l * 9 + k];
                  uint32_t lc;
                  const double lsum = score_and_sum<FAM>(w, ML, max_residual, &lc);
                  if (lc > best_n || (lc == best_n && lsum < best_sum)) {
                    best_n = lc;
                    best_sum = lsum;
                    for (int k = 0; k < 9; ++k) best_model[k] = ML[k];
                  }
                }
              } else {
                if (lane == 0) {
                  fs->t_pos = (uint32_t)t;
                  fs->m_pos = (uint32_t)(m + 1);
                  fs->lo_wait = 1;
                  fs->lo_ninl = (uint32_t)ninl;
                  p.lo_queue[atomicAdd(p.lo_count, 1u)] = pl;
                  if (!(p.lo_reg_prepare && lo_prepare_in_registers<FAM>(ninl))) p.lo_queue_g[atomicAdd(p.active_count + 22, 1u)] = pl;
                  if ((FAM == FAM_H ? 2 * ninl : ninl) < 9) atomicAdd(p.active_count + 23, 1u);
                  if (p.lo_reg_prepare && lo_prepare_big<FAM>(ninl)) atomicAdd(p.active_count + 25, 1u);
                }
                suspended = true;
                break;
              }
            }
            dyn_max = w.nt_table[best_n];
          }
        }
        if (trial_abs >= dyn_max && trial_abs >= min_trials) {
          abort = true;
          break;
        }
      }
      if (suspended) break;
      if (abort) {
        t_stop = t;
        break;
      }
      t += 1;
    }
    if (suspended) {
      if (lane == 0) {
        fs->rep.num_models = num_models;
        fs->rep.num_inliers = best_n;
        fs->rep.residual_sum = best_sum;
        for (int k = 0; k < 9; ++k) fs->rep.model[k] = best_model[k];
        fs->dyn_max = dyn_max;
      }
      continue;
    }

    uint32_t trials_done;
    bool finished;
    uint32_t* st = p.pair_state + (size_t)pi * PAIR_STATE_WORDS;
    if (abort) {
      trials_done = (trial_abs + 1 < max_num_trials) ? trial_abs + 2 : trial_abs + 1;  // loransac.h:129-134
      finished = true;
      if (lane == 0) {
        if (t_stop != nb - 1) {  // later samples of this round were speculation: resume from the snapshot
          st[PS_SKIP] = p.draws_end[(size_t)pl * p.batch + t_stop];
          st[PS_USE_SNAP] = 1;
        }
      }
    } else {
      trials_done = T0 + (uint32_t)nb;
      finished = trials_done >= max_num_trials;
    }
    if (lane == 0) {
      fs->rep.num_trials = trials_done;
      fs->rep.num_models = num_models;
      fs->rep.num_inliers = best_n;
      fs->rep.residual_sum = best_sum;
      for (int k = 0; k < 9; ++k) fs->rep.model[k] = best_model[k];
      fs->dyn_max = dyn_max;
      fs->rounds += 1;
      fs->lo_wait = 0;
      fs->active = finished ? 0u : 1u;
      if (!finished) atomicAdd(p.active_count, 1u);
    }
    if (finished) {
      const bool success = best_n >= (uint32_t)F::K;
      if (success) {
        score_model<FAM>(w, best_model, max_residual, true);
        wv_sync();
        unsigned char* mask = p.masks + (size_t)FAM * p.mask_stride + moff;
        for (int i = lane; i < n; i += 64) mask[i] = ws.resid[i] <= max_residual;
      }
      if (lane == 0) {
        RansacReport rep;
        rep.success = success;
        rep.num_trials = trials_done;
        rep.num_models = num_models;
        rep.num_inliers = best_n;
        rep.residual_sum = best_sum;
        for (int k = 0; k < 9; ++k) rep.model[k] = best_model[k];
        p.reports[(size_t)pi * 3 + FAM] = rep;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ replay with the pair resident in LDS (round 6)
// k_replay_lo<fam, 0 / 2> is latency: a visit of a pair (one per local-optimisation step: 4 - 5 per pair, family and round) scored
// every model it looked at straight from global memory -- four dependent round trips through the pair's points per model (one per
// 64 correspondences, each behind the in-order walk of the chunk before it), the residuals written to memory and read back for the
// compaction, every model and every count fetched where it was needed: ~60 dependent round trips, 65 - 77 us of a wave per visit
// at four waves per SIMD (profiles/r06_replay_dispatches_before.txt: 2.36 ms for the first launch of 124 750 pairs on 4 096 waves),
// VALU issue 0.05 - 0.18.  k_replay_rp is the same scan with what a visit needs fetched ONCE, in batches that are in flight together:
//   * the pair's correspondences go to LDS at the start of the visit (n <= rp_cap = min(n_max, RP_CAP); a longer pair reads them
//     from global memory as before), together with the models the local optimisation returned;
//   * a trial's models, counts and sums are fetched in one batch when the scan stops at it;
//   * the inlier set of the model scored last stays in LDS as a bit mask (one 64-bit word per 64 correspondences): the compaction
//     and the final mask read the bits, no residual goes through memory.
// Same operations on the same values in the same order as k_replay_lo (which stays, modes 0 and 2, as a cross-check schedule of the
// check build: DSM_REPLAY_LEGACY; mode 1, the inline tail, is still k_replay_lo's).
#define RP_CAP 256
#ifndef ITEMS_GROUP
#define ITEMS_GROUP 8
#endif
// ITEMS_GROUP: waves (= pairs) per workgroup of k_items_enum (x TAIL_KMAX items = the 64 lanes of the wave that lists the jobs)
#define RP_QL 32  // pairs a wave suspends before it appends them to the queue with ONE atomic (grab_seg's note)
__host__ __device__ inline size_t rp_lds_bytes(uint32_t cap, uint32_t n_max) {
  return (size_t)cap * 32 + 90 * 8 + 10 * 8 + 12 * 4 + (size_t)((n_max + 63) / 64) * 8 + RP_QL * 4;
}
struct RpLds {
  double* pts;               // cap x 4
  double* mdl;               // 90: the models being looked at (a local optimisation's, then a trial's)
  double* sums;              // 10: the trial's in-order sums (F, H)
  int* cnts;                 // 10 counts + the trial's model count
  unsigned long long* bits;  // inlier mask of the model scored last with KEEP
  uint32_t* ql;              // RP_QL: the pairs this wave has suspended and not yet appended to the queue
};
DSM_DEV RpLds rp_lds(unsigned char* raw, uint32_t cap, uint32_t n_max) {
  RpLds l;
  l.pts = reinterpret_cast<double*>(raw);
  l.mdl = l.pts + (size_t)cap * 4;
  l.sums = l.mdl + 90;
  l.cnts = reinterpret_cast<int*>(l.sums + 10);
  l.bits = reinterpret_cast<unsigned long long*>(l.cnts + 12);
  l.ql = reinterpret_cast<uint32_t*>(l.bits + (n_max + 63) / 64);
  return l;
}
// Residuals of model M over the pair: the inlier count; SUM: InlierSupportMeasurer::Evaluate's residual_sum in index order
// (support_measurement.cc:43-48: the walk over the inliers of every 64-chunk in ascending lane order, as score_and_sum);
// KEEP: the inlier mask -> bits[].
template <int FAM, bool SUM, bool KEEP>
DSM_DEV double rp_score(const double* M, int n, bool in_lds, const double* spts, const double* gpts, double max_residual,
                        uint32_t* count_out, unsigned long long* bits, int lane) {
  double s = 0;
  uint32_t count = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    bool in = false;
    double r = 0.0;
    if (i < n) {
      double q[4];
      if (in_lds) {  // (two branches, not one selected pointer: that would compile to flat loads)
        for (int k = 0; k < 4; ++k) q[k] = spts[(size_t)i * 4 + k];
      } else {
        for (int k = 0; k < 4; ++k) q[k] = gpts[(size_t)i * 4 + k];
      }
      r = fam_residual<FAM>(M, q);
      in = r <= max_residual;
    }
    unsigned long long mask = __ballot(in);
    count += (uint32_t)__popcll(mask);
    if (KEEP && lane == 0) bits[base >> 6] = mask;
    if (SUM) {
      while (mask) {
        const int k = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        s += wv_readlane_f64(r, k);
      }
    }
  }
  if (KEEP) wv_sync();
  *count_out = count;
  return s;
}
// ordered compaction of the inliers of bits[] into inl[]; returns the count
DSM_DEV int rp_compact(const unsigned long long* bits, int n, int* inl, int lane) {
  int total = 0;
  for (int base = 0; base < n; base += 64) {
    const unsigned long long bal = bits[base >> 6];
    if ((bal >> lane) & 1ull) inl[total + __popcll(bal & ((1ull << lane) - 1ull))] = base + lane;
    total += __popcll(bal);
  }
  return total;
}

template <int FAM, bool LOOKUP>
__global__ __launch_bounds__(64, DSM_REPLAY_WAVES) void k_replay_rp(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef Fam<FAM> F;
  constexpr int MAXM = F::MAXM;
  const int lane = threadIdx.x;
  const RpLds L = rp_lds(smem_raw, p.rp_cap, p.n_max);
  SegGrab wgrab;
  const uint32_t grain = work_grain(p.n_work);
  // what this wave owes the lane's counters: appended / added with one atomic each when the list is full and at the end
  uint32_t q_n = 0, q_gmask = 0, q_small = 0, q_big = 0, n_active = 0;
  auto flush_queue = [&]() {
    wv_sync();
    uint32_t base = 0, base_g = 0;
    if (lane == 0) {
      base = atomicAdd(p.lo_count, q_n);
      if (q_gmask) base_g = atomicAdd(p.active_count + 22, (uint32_t)__popc(q_gmask));
      if (q_small) atomicAdd(p.active_count + 23, q_small);
      if (q_big) atomicAdd(p.active_count + 25, q_big);
    }
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    base_g = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_g);
    if ((uint32_t)lane < q_n) {
      const uint32_t e = L.ql[lane];
      p.lo_queue[base + (uint32_t)lane] = e;
      if ((q_gmask >> lane) & 1u) p.lo_queue_g[base_g + (uint32_t)__popc(q_gmask & ((1u << lane) - 1u))] = e;
    }
    q_n = 0;
    q_gmask = 0;
    q_small = 0;
    q_big = 0;
    wv_sync();
  };
  for (;;) {
    wv_sync();
    const uint32_t widx = grab_seg(wgrab, GRAB_REPLAY(p), p.n_work, lane, grain);
    if (widx == GRAB_DONE) break;
    if (widx >= p.n_work) continue;
    const uint32_t pl = p.worklist ? p.worklist[widx] : widx;
    const uint32_t pi = p.pair0 + pl;
    FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
    if (!fs->active) continue;
    const uint64_t moff = p.match_off[pi];
    const int n = (int)(p.match_off[pi + 1] - moff);
    const double* gpts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
    const bool in_lds = n <= (int)p.rp_cap;
    // ---- batch 1: the state, the models of a returned local optimisation and the points, all in flight together
    const bool lo_wait = fs->lo_wait != 0;
    int nlo = lo_wait ? (int)fs->lo_nm : 0;
    if (LOOKUP) nlo = 0;
    double lm0 = 0.0, lm1 = 0.0;
    {
      const double* glom = p.lo_models + (size_t)pl * 90;
      if (lane < nlo * 9) lm0 = glom[lane];
      if (lane + 64 < nlo * 9) lm1 = glom[lane + 64];
    }
    uint32_t best_n = fs->rep.num_inliers;
    double best_sum = fs->rep.residual_sum;
    double best_model[9];
    for (int k = 0; k < 9; ++k) best_model[k] = fs->rep.model[k];
    uint32_t dyn_max = fs->dyn_max;
    uint32_t num_models = fs->rep.num_models;
    const uint32_t T0 = fs->rep.num_trials;
    const int nb = (int)fs->nb;
    int t = (int)fs->t_pos;
    int m_start = (int)fs->m_pos;
    const uint32_t* nt_table = p.nt_table + p.nt_off[n] + (size_t)FAM * (size_t)(n + 1);
    double max_error = p.opt.max_error;
    if (FAM == FAM_E) {
      const dsm_camera& cam1 = p.cams[p.pairs[2 * pi]];
      const dsm_camera& cam2 = p.cams[p.pairs[2 * pi + 1]];
      max_error = (image_to_world_threshold(cam1, p.opt.max_error) + image_to_world_threshold(cam2, p.opt.max_error)) / 2;
    }
    if (in_lds) (void)stage_points_batched<true, false>(gpts, 4 * n, L.pts, lane);
    if (lane < nlo * 9) L.mdl[lane] = lm0;
    if (lane + 64 < nlo * 9) L.mdl[lane + 64] = lm1;
    wv_sync();
    const double max_residual = max_error * max_error;
    const uint32_t min_trials = (uint32_t)p.opt.min_num_trials;
    const uint32_t max_num_trials = p.max_trials[FAM];
    const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
    const int32_t* cnts = p.counts + (size_t)pl * p.batch * MAXM;
    const double* sums = p.sums + (size_t)pl * p.batch * MAXM;  // F and H only
    const double* mods = p.models + (size_t)pl * p.batch * MAXM * 9;
    int* inl = reinterpret_cast<int*>(p.lo_inl + moff);  // the compaction's output IS the hand-over to the LO kernels

    bool abort = false, suspended = false;
    int t_stop = nb - 1;
    uint32_t trial_abs = T0 + (uint32_t)t;
    bool in_trial = false;  // resume inside trial t at model m_start
    if (lo_wait) {
      // the local optimisation of (trial t, model m_start - 1) has returned: loransac.h:160-178
      if constexpr (LOOKUP) {  // item 0 of the pair is the pending step
        const TailItem& it = p.tail_items[(size_t)widx * TAIL_KMAX];
        num_models += it.nlo;
        best_n = it.out_n;
        best_sum = it.out_sum;
        for (int k = 0; k < 9; ++k) best_model[k] = it.out_model[k];
      }
      for (int l = 0; l < nlo; ++l) {
        num_models += 1;
        double M[9];
        for (int k = 0; k < 9; ++k) M[k] = L.mdl[l * 9 + k];
        uint32_t lc;
        const double lsum = rp_score<FAM, true, false>(M, n, in_lds, L.pts, gpts, max_residual, &lc, L.bits, lane);
        if (lc > best_n || (lc == best_n && lsum < best_sum)) {
          best_n = lc;
          best_sum = lsum;
          for (int k = 0; k < 9; ++k) best_model[k] = M[k];
        }
      }
      dyn_max = nt_table[best_n];
      if (trial_abs >= dyn_max && trial_abs >= min_trials) {
        abort = true;
        t_stop = t;
      }
      in_trial = true;
    }
    while (!abort && t < nb) {
      if (!in_trial) {
        // ---- skip ahead to the next trial that can change anything
        t = replay_next_event<FAM>(t, nb, nmod, cnts, sums, best_n, best_sum, dyn_max > min_trials ? dyn_max : min_trials, T0, &num_models, lane);
        if (t >= nb) break;
        m_start = 0;
      }
      in_trial = false;
      trial_abs = T0 + (uint32_t)t;
      // ---- batch 2: the trial's model count, counts, sums and models, in flight together
      wv_sync();  // (the models of the step before are no longer read)
      {
        const double a0 = mods[(size_t)t * MAXM * 9 + (lane < MAXM * 9 ? lane : 0)];
        double a1 = 0.0;
        if (MAXM * 9 > 64) a1 = mods[(size_t)t * MAXM * 9 + (lane + 64 < MAXM * 9 ? lane + 64 : 0)];
        int c = 0;
        double sm_ = 0.0;
        if (lane < MAXM) c = cnts[(size_t)t * MAXM + lane];
        if (FAM != FAM_E && lane < MAXM) sm_ = sums[(size_t)t * MAXM + lane];
        if (lane == MAXM) c = nmod[t];
        if (lane < MAXM * 9) L.mdl[lane] = a0;
        if (MAXM * 9 > 64 && lane + 64 < MAXM * 9) L.mdl[lane + 64] = a1;
        if (lane <= MAXM) L.cnts[lane] = c;
        if (FAM != FAM_E && lane < MAXM) L.sums[lane] = sm_;
      }
      wv_sync();
      // ---- exact sequential processing of trial t (loransac.h:142-198)
      const int nm = L.cnts[MAXM];
      for (int m = m_start; m < nm; ++m) {
        num_models += 1;
        const uint32_t cnt = (uint32_t)L.cnts[m];
        if (cnt >= best_n) {
          if (p.stats && lane == 0) atomicAdd(p.active_count + 1 + FAM * 2, 1u);
          double M[9];
          for (int k = 0; k < 9; ++k) M[k] = L.mdl[m * 9 + k];
          double sum;
          bool have_bits = false;
          if constexpr (FAM == FAM_E) {
            uint32_t cnt_again;
            sum = rp_score<FAM, true, true>(M, n, in_lds, L.pts, gpts, max_residual, &cnt_again, L.bits, lane);
            have_bits = true;
          } else {
            sum = L.sums[m];  // k_score's in-order sum
          }
          if (cnt > best_n || (cnt == best_n && sum < best_sum)) {
            best_n = cnt;
            best_sum = sum;
            for (int k = 0; k < 9; ++k) best_model[k] = M[k];
            if (cnt > (uint32_t)F::K && cnt >= (uint32_t)F::LO_MIN) {
              if (p.stats && lane == 0) atomicAdd(p.active_count + 2 + FAM * 2, 1u);
              bool looked_up = false;
              if constexpr (LOOKUP) {
                const uint32_t ni = p.tail_n[widx];
                for (uint32_t k = 0; k < ni; ++k) {  // (a pending item 0 is the model the pair was suspended at: it cannot come again)
                  const TailItem& it = p.tail_items[(size_t)widx * TAIL_KMAX + k];
                  if (it.t == (uint32_t)t && it.m == (uint32_t)m) {
                    num_models += it.nlo;
                    best_n = it.out_n;
                    best_sum = it.out_sum;
                    for (int q = 0; q < 9; ++q) best_model[q] = it.out_model[q];
                    looked_up = true;
                    break;
                  }
                }
              }
              if (looked_up) {
                dyn_max = nt_table[best_n];
                if (trial_abs >= dyn_max && trial_abs >= min_trials) {
                  abort = true;
                  break;
                }
                continue;
              }
              if (!have_bits) {  // the inliers of the new best model, for the compaction
                uint32_t c2;
                (void)rp_score<FAM, false, true>(M, n, in_lds, L.pts, gpts, max_residual, &c2, L.bits, lane);
              }
              const int ninl = rp_compact(L.bits, n, inl, lane);  // -> lo_inl
              if (lane == 0) {
                fs->t_pos = (uint32_t)t;
                fs->m_pos = (uint32_t)(m + 1);
                fs->lo_wait = 1;
                fs->lo_ninl = (uint32_t)ninl;
                L.ql[q_n] = pl;
              }
              if (!(p.lo_reg_prepare && lo_prepare_in_registers<FAM>(ninl))) q_gmask |= 1u << q_n;
              if ((FAM == FAM_H ? 2 * ninl : ninl) < 9) q_small += 1;
              if (p.lo_reg_prepare && lo_prepare_big<FAM>(ninl)) q_big += 1;
              q_n += 1;
              suspended = true;
              break;
            }
            dyn_max = nt_table[best_n];
          }
        }
        if (trial_abs >= dyn_max && trial_abs >= min_trials) {
          abort = true;
          break;
        }
      }
      if (suspended) break;
      if (abort) {
        t_stop = t;
        break;
      }
      t += 1;
    }
    if (suspended) {
      if (lane == 0) {
        fs->rep.num_models = num_models;
        fs->rep.num_inliers = best_n;
        fs->rep.residual_sum = best_sum;
        for (int k = 0; k < 9; ++k) fs->rep.model[k] = best_model[k];
        fs->dyn_max = dyn_max;
      }
      if (q_n == RP_QL) flush_queue();
      continue;
    }

    uint32_t trials_done;
    bool finished;
    uint32_t* st = p.pair_state + (size_t)pi * PAIR_STATE_WORDS;
    if (abort) {
      trials_done = (trial_abs + 1 < max_num_trials) ? trial_abs + 2 : trial_abs + 1;  // loransac.h:129-134
      finished = true;
      if (lane == 0) {
        if (t_stop != nb - 1) {  // later samples of this round were speculation: resume from the snapshot
          st[PS_SKIP] = p.draws_end[(size_t)pl * p.batch + t_stop];
          st[PS_USE_SNAP] = 1;
        }
      }
    } else {
      trials_done = T0 + (uint32_t)nb;
      finished = trials_done >= max_num_trials;
    }
    if (lane == 0) {
      fs->rep.num_trials = trials_done;
      fs->rep.num_models = num_models;
      fs->rep.num_inliers = best_n;
      fs->rep.residual_sum = best_sum;
      for (int k = 0; k < 9; ++k) fs->rep.model[k] = best_model[k];
      fs->dyn_max = dyn_max;
      fs->rounds += 1;
      fs->lo_wait = 0;
      fs->active = finished ? 0u : 1u;
    }
    if (!finished) n_active += 1;
    if (finished) {
      const bool success = best_n >= (uint32_t)F::K;
      if (success) {
        uint32_t c2;
        (void)rp_score<FAM, false, true>(best_model, n, in_lds, L.pts, gpts, max_residual, &c2, L.bits, lane);
        unsigned char* mask = p.masks + (size_t)FAM * p.mask_stride + moff;
        for (int i = lane; i < n; i += 64) mask[i] = (unsigned char)((L.bits[i >> 6] >> lane) & 1ull);
      }
      if (lane == 0) {
        RansacReport rep;
        rep.success = success;
        rep.num_trials = trials_done;
        rep.num_models = num_models;
        rep.num_inliers = best_n;
        rep.residual_sum = best_sum;
        for (int k = 0; k < 9; ++k) rep.model[k] = best_model[k];
        p.reports[(size_t)pi * 3 + FAM] = rep;
      }
    }
  }
  if (q_n) flush_queue();
  if (n_active && lane == 0) atomicAdd(p.active_count, n_active);
}

// ------------------------------------------------------------------------------------ item passes
// The chain of k_replay_lo costs one iteration of launches per local-optimisation step of the SLOWEST pair (14 - 16 per
// round at the benchmark shape), and every iteration its full serial latency (~1 ms for the E family: tall QR, 9 x 9
// Jacobi, 5-point solver, each a serial computation per problem) however few problems it carries.  That is the price
// of a short pair list -- one GPU's share of an 8-GPU job spends a third of its verification time there -- and of the
// last iterations of any list.  But the outcome of a local-optimisation step is a function of the candidate model
// alone (TailItem), so all steps a pair can still reach in the round are independent work:
//   k_items_enum     wave per pair: (item 0 = the step the pair is suspended at, if it is;) then every later model of
//                    the batch that could still become the best one -- a model with at least as many inliers as every
//                    minimal-sample model before it.  The true best support is never smaller than that running
//                    maximum, so the steps the sequential scan will really take are among the items; the scan cannot
//                    go past the first trial at or beyond the stopping threshold of that maximum (ComputeNumTrials
//                    is non-increasing in the inlier count), so the list ends there.  Every item becomes a job.
//   k_items_inliers  wave per job: the candidate's residuals and ordered inlier list (the local estimator's input).
//   k_lo_*           the batched local-optimisation kernels, over the jobs (lo_ref).
//   k_items_outcome  wave per job: the local models scored in order against the candidate -> the item's outcome.
//   k_replay_lo<FAM, 2>  the sequential scan; takes the outcome where it would otherwise suspend.
// One pass finishes the round of every pair with at most TAIL_KMAX candidate steps; the others are queued once more.
template <int FAM>
__global__ __launch_bounds__(64 * ITEMS_GROUP) void k_items_enum(const VerifyParams p) {
  typedef Fam<FAM> F;
  constexpr int MAXM = F::MAXM;
  // a workgroup = ITEMS_GROUP waves, a wave per pair as before; the group's job slots are taken with ONE atomicAdd (a same-address
  // device atomic costs 11.4 ns whoever issues it: one per pair was 0.18 ms of every pass over a 15 600-pair shard; grab_seg's note)
  const int lane = threadIdx.x & 63;
  const uint32_t q = threadIdx.x >> 6;
  __shared__ uint32_t g_n[ITEMS_GROUP], g_pend[ITEMS_GROUP], g_ninl[ITEMS_GROUP], g_pl[ITEMS_GROUP];
  for (uint32_t w0 = blockIdx.x * ITEMS_GROUP; w0 < p.n_work; w0 += gridDim.x * ITEMS_GROUP) {
   __syncthreads();
   if (threadIdx.x < ITEMS_GROUP) g_n[threadIdx.x] = 0;
   __syncthreads();
   for (uint32_t once = 0; once < 1u && w0 + q < p.n_work; ++once) {
    const uint32_t widx = w0 + q;
    const uint32_t pl = p.worklist ? p.worklist[widx] : widx;
    const uint32_t pi = p.pair0 + pl;
    const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
    if (!fs->active) {
      if (lane == 0) p.tail_n[widx] = 0;
      continue;
    }
    TailItem* items = p.tail_items + (size_t)widx * TAIL_KMAX;
    const int n = (int)(p.match_off[pi + 1] - p.match_off[pi]);
    const uint32_t* nt = p.nt_table + p.nt_off[n] + (size_t)FAM * (size_t)(n + 1);
    const uint32_t min_trials = (uint32_t)p.opt.min_num_trials;
    const uint32_t T0 = fs->rep.num_trials;
    const int nb = (int)fs->nb;
    const bool pending = fs->lo_wait != 0;
    // where the scan stands: a suspended pair right behind the model it is suspended at, a fresh one at the start
    const int t0 = (int)fs->t_pos;
    const int m_next = (int)fs->m_pos;
    const int32_t* nmod = p.nmodels + (size_t)pl * p.batch;
    const int32_t* cnts = p.counts + (size_t)pl * p.batch * MAXM;
    uint32_t n_items = 0;
    if (pending) {
      if (lane == 0) {
        items[0].t = (uint32_t)t0;
        items[0].m = (uint32_t)(m_next - 1);
      }
      n_items = 1;
    }
    int rm = (int)fs->rep.num_inliers;  // the best support so far (for a suspended pair: of the model it is suspended at)
    bool stop = false;
    for (int tb = t0; tb < nb && !stop && n_items < TAIL_KMAX; tb += 64) {
      const int tt = tb + lane;
      int nm = 0, c[MAXM], mfirst = 0;
      if (tt < nb) {
        nm = nmod[tt];
        mfirst = tt == t0 ? m_next : 0;
      }
      int tmax = -1;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        c[m] = (m < nm && m >= mfirst) ? cnts[(size_t)tt * MAXM + m] : -1;
        tmax = max(tmax, c[m]);
      }
      // running maximum before this lane's trial (exclusive prefix maximum over the lanes, seeded with rm)
      int incl = tmax;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl = max(incl, v);
      }
      int ex = __shfl_up(incl, 1);
      ex = lane == 0 ? rm : max(ex, rm);
      // the scan stops at the first trial with a model at or beyond the threshold of the maximum so far
      const uint32_t thr = max(nt[ex], min_trials);
      const bool stops_here = nm > 0 && (T0 + (uint32_t)tt) >= thr;
      const unsigned long long sb = __ballot(stops_here);
      const int last = sb ? (__ffsll((long long)sb) - 1) : 63;  // lanes beyond are never reached
      if (sb) stop = true;
      int r = ex, ncand = 0;
      unsigned cmask = 0;
      if (lane <= last) {
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
          if (c[m] >= 0) {
            if (c[m] >= r && c[m] > F::K && c[m] >= F::LO_MIN) {
              cmask |= 1u << m;
              ncand += 1;
            }
            r = max(r, c[m]);
          }
        }
      }
      int pos = ncand;  // inclusive prefix sum
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(pos, o);
        if (lane >= o) pos += v;
      }
      const int total = __shfl(pos, 63);
      int at = (int)n_items + pos - ncand;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if ((cmask >> m) & 1u) {
          if (at < TAIL_KMAX) {
            items[at].t = (uint32_t)tt;
            items[at].m = (uint32_t)m;
          }
          at += 1;
        }
      }
      n_items = min((uint32_t)TAIL_KMAX, n_items + (uint32_t)total);
      rm = max(rm, __shfl(incl, last));
    }
    if (lane == 0) {
      p.tail_n[widx] = n_items;
      g_n[q] = n_items;
      g_pend[q] = pending ? 1u : 0u;
      g_ninl[q] = pending ? fs->lo_ninl : 0u;
      g_pl[q] = pl;
    }
   }
   // the jobs of the group's pairs: job index = item slot (widx * TAIL_KMAX + k), listed compactly for the kernels that follow
   __syncthreads();
   static_assert(ITEMS_GROUP * TAIL_KMAX <= 64, "one lane of the first wave per (pair of the group, item)");
   if (q == 0) {
     const uint32_t qq = ((uint32_t)lane / TAIL_KMAX) % ITEMS_GROUP, k = (uint32_t)lane % TAIL_KMAX;
     const bool lister = (uint32_t)lane < ITEMS_GROUP * TAIL_KMAX;
     uint32_t before = 0, total = 0;
     for (uint32_t r = 0; r < ITEMS_GROUP; ++r) {
       const uint32_t c = g_n[r];
       if (r < qq) before += c;
       total += c;
     }
     uint32_t base = 0;
     if (lane == 0 && total) base = atomicAdd(p.active_count + 24, total);
     base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
     if (lister && k < g_n[qq]) {
       const uint32_t slot = (w0 + qq) * TAIL_KMAX + k;
       LoJob* j = p.lo_jobs + slot;
       j->pl = g_pl[qq];
       j->ninl = k == 0 ? g_ninl[qq] : 0u;
       j->pending = k == 0 ? g_pend[qq] : 0u;
       j->nm = 0;
       p.job_list[base + before + k] = slot;
     }
   }
  }
}

// per-wave set-up shared by the two item kernels below
template <int FAM>
DSM_DEV PairWork item_pair_work(const VerifyParams& p, const WgScratch& ws, VSmem* sm, uint32_t pi, int lane, double* max_residual) {
  const uint64_t moff = p.match_off[pi];
  const dsm_camera& cam1 = p.cams[p.pairs[2 * pi]];
  const dsm_camera& cam2 = p.cams[p.pairs[2 * pi + 1]];
  PairWork w;
  w.n = (int)(p.match_off[pi + 1] - moff);
  w.resid = ws.resid;
  w.inl = ws.inl;
  w.tall = ws.tall;
  w.models = nullptr;
  w.sm = sm;
  w.lane = lane;
  w.pts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
  w.nt_table = nullptr;
  double max_error = p.opt.max_error;
  if (FAM == FAM_E)
    max_error = (image_to_world_threshold(cam1, p.opt.max_error) + image_to_world_threshold(cam2, p.opt.max_error)) / 2;
  *max_residual = max_error * max_error;
  return w;
}

// wave per job: the ordered inlier list of the candidate model (loransac.h:160-163) into the job's slice of the pool,
// its size into the job; the few problems the register kernels do not take are listed for the general ones, as
// k_replay_lo does when it suspends a pair.
template <int FAM>
__global__ __launch_bounds__(64, 4) void k_items_inliers(const VerifyParams p) {
  typedef Fam<FAM> F;
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  for (uint32_t ji = blockIdx.x; ji < p.n_work; ji += gridDim.x) {
    wv_sync();
    const uint32_t slot = p.job_list[ji];
    LoJob* j = p.lo_jobs + slot;
    const uint32_t pi = p.pair0 + j->pl;
    int ninl;
    if (j->pending) {
      ninl = (int)j->ninl;
    } else {
      double max_residual;
      PairWork w = item_pair_work<FAM>(p, ws, nullptr, pi, lane, &max_residual);
      const TailItem& it = p.tail_items[slot];
      const double* M = p.models + (((size_t)j->pl * p.batch + it.t) * F::MAXM + it.m) * 9;
      double Mr[9];
      for (int q = 0; q < 9; ++q) Mr[q] = M[q];
      score_model<FAM>(w, Mr, max_residual, true);
      wv_sync();
      const uint64_t moff = p.match_off[pi];
      w.inl = reinterpret_cast<int*>(p.lo_inl_pool + moff * TAIL_KMAX + (uint64_t)(slot % TAIL_KMAX) * (uint64_t)w.n);
      ninl = compact_inliers(w, max_residual);
      if (lane == 0) j->ninl = (uint32_t)ninl;
    }
    if (lane == 0) {
      if (!(p.lo_reg_prepare && lo_prepare_in_registers<FAM>(ninl))) p.lo_queue_g[atomicAdd(p.active_count + 22, 1u)] = slot;
      if ((FAM == FAM_H ? 2 * ninl : ninl) < 9) atomicAdd(p.active_count + 23, 1u);
      if (p.lo_reg_prepare && lo_prepare_big<FAM>(ninl)) atomicAdd(p.active_count + 25, 1u);
    }
  }
}

// wave per job: the candidate's support, then its local models in order (loransac.h:166-178) -> the item's outcome
template <int FAM>
__global__ __launch_bounds__(64, 4) void k_items_outcome(const VerifyParams p) {
  typedef Fam<FAM> F;
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  for (uint32_t ji = blockIdx.x; ji < p.n_work; ji += gridDim.x) {
    wv_sync();
    const uint32_t slot = p.job_list[ji];
    const LoJob* j = p.lo_jobs + slot;
    const uint32_t pi = p.pair0 + j->pl;
    const FamState* fs = p.fam_state + (size_t)pi * 3 + FAM;
    TailItem* it = p.tail_items + slot;
    double max_residual;
    PairWork w = item_pair_work<FAM>(p, ws, nullptr, pi, lane, &max_residual);
    uint32_t best_n;
    double best_sum, best_model[9];
    if (j->pending) {  // the candidate is the pair's current best
      best_n = fs->rep.num_inliers;
      best_sum = fs->rep.residual_sum;
      for (int q = 0; q < 9; ++q) best_model[q] = fs->rep.model[q];
    } else {
      const uint32_t t = it->t, m = it->m;
      const double* M = p.models + (((size_t)j->pl * p.batch + t) * F::MAXM + m) * 9;
      for (int q = 0; q < 9; ++q) best_model[q] = M[q];
      if constexpr (FAM == FAM_E) {
        best_sum = score_and_sum<FAM>(w, best_model, max_residual, &best_n);
      } else {
        best_n = (uint32_t)p.counts[((size_t)j->pl * p.batch + t) * F::MAXM + m];
        best_sum = p.sums[((size_t)j->pl * p.batch + t) * F::MAXM + m];  // k_score's in-order sum
      }
    }
    const int nlo = (int)j->nm;
    const double* glom = p.lo_models + (size_t)slot * 90;
    for (int l = 0; l < nlo; ++l) {
      double ML[9];
      for (int q = 0; q < 9; ++q) ML[q] = glom[l * 9 + q];
      uint32_t lc;
      const double lsum = score_and_sum<FAM>(w, ML, max_residual, &lc);
      if (lc > best_n || (lc == best_n && lsum < best_sum)) {
        best_n = lc;
        best_sum = lsum;
        for (int q = 0; q < 9; ++q) best_model[q] = ML[q];
      }
    }
    if (lane == 0) {
      it->nlo = (uint32_t)nlo;
      it->out_n = best_n;
      it->out_sum = best_sum;
      for (int q = 0; q < 9; ++q) it->out_model[q] = best_model[q];
    }
  }
}

// What a batched local-optimisation kernel works on: a queued pair (chain of k_replay_lo: per-pair buffers, slot = the
// pair) or a job of an item pass (slot = the job; LoJob).
struct LoRef {
  uint32_t pl, slot;
  int ninl;
  const int* inl;
  uint32_t* nm;
};
template <int FAM>
DSM_DEV LoRef lo_ref(const VerifyParams& p, uint32_t widx) {
  LoRef r;
  const uint32_t w = p.worklist ? p.worklist[widx] : widx;
  if (p.lo_jobs != nullptr) {
    LoJob* j = p.lo_jobs + w;
    const uint32_t pi = p.pair0 + j->pl;
    const uint64_t moff = p.match_off[pi];
    const uint64_t n = p.match_off[pi + 1] - moff;
    r.pl = j->pl;
    r.slot = w;
    r.ninl = (int)j->ninl;
    r.inl = reinterpret_cast<const int*>(j->pending ? p.lo_inl + moff : p.lo_inl_pool + moff * TAIL_KMAX + (uint64_t)(w % TAIL_KMAX) * n);
    r.nm = &j->nm;
  } else {
    FamState* fs = p.fam_state + (size_t)(p.pair0 + w) * 3 + FAM;
    r.pl = w;
    r.slot = w;
    r.ninl = (int)fs->lo_ninl;
    r.inl = reinterpret_cast<const int*>(p.lo_inl + p.match_off[p.pair0 + w]);
    r.nm = &fs->lo_nm;
  }
  return r;
}

// LO step 1 with the constraint matrix in registers (wr_colpiv_qr9): a wave per queued pair, lane l owns rows l,
// l + 64, ...; the pair's inlier points are loaded once and serve the in-order normalisation sums (the four / two
// independent chains of CenterAndNormalizeImagePoints advance together) and the rows.  Same operations as k_lo_prepare.
// ((0 + v[0]) + v[stride]) + ... over n operands in LDS, by this lane alone (n = 0: the lane has no chain): eight loads in flight
// ahead of the dependent adds
DSM_DEV double lds_chain_sum(const double* v, int stride, int n) {
  double s = 0.0;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = v[(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u];
  }
  for (; k < n; ++k) s += v[k * stride];
  return s;
}
template <int FAM, int PPL>
__global__ __launch_bounds__(64, (FAM == FAM_H ? 2 * PPL : PPL) <= 3 ? LOP_SMALL_WAVES : 2) void k_lo_prepare_reg(const VerifyParams p) {  // (three waves per SIMD for the small sizes: 173 / 185 VGPRs left to itself)
  constexpr int RPL = FAM == FAM_H ? 2 * PPL : PPL;
  __shared__ double lo_prep_lds[FAM == FAM_E ? 1 : 64 * PPL * 4];  // the operands of the normalisation's in-order sums (F, H)
  const int lane = threadIdx.x;
  const uint32_t widx = blockIdx.x;
  if (widx >= p.n_work) return;
  const LoRef ref = lo_ref<FAM>(p, widx);
  const uint32_t pi = p.pair0 + ref.pl;
  const int ninl = ref.ninl;
  if (!lo_prepare_in_registers<FAM>(ninl)) return;
  if (lo_prepare_big<FAM>(ninl) != (PPL == (FAM == FAM_H ? LOP_PPL_H_BIG : LOP_PPL))) return;  // the other size's
  const uint64_t moff = p.match_off[pi];
  const double* pts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
  const int* inl = ref.inl;
  double* out = p.lo_work + (size_t)ref.slot * LO_WORK_DOUBLES;
  const int m = FAM == FAM_H ? 2 * ninl : ninl;
  double px[PPL][4];
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const int i = lane + 64 * r;
#pragma unroll
    for (int k = 0; k < 4; ++k) px[r][k] = 0.0;
    if (i < ninl) {
      const double* q = pts + (size_t)inl[i] * 4;
      px[r][0] = q[0]; px[r][1] = q[1]; px[r][2] = q[2]; px[r][3] = q[3];
    }
  }
  double n1[3] = {0, 0, 0}, n2[3] = {0, 0, 0};
  if (FAM != FAM_E) {
    // CenterAndNormalizeImagePoints (utils.cc:40-64) for both images: sums in index order.  The four coordinate sums and then the
    // two sums of squared distances are independent sequential chains over the inliers: the operands go through LDS and lane c
    // walks chain c -- one add per inlier and chain member for the wave.  (Rounds 3 - 5 walked every chain on all 64 lanes through
    // v_readlane: two readlanes and an add per operand, 3 000 of the 8 200 instructions an F problem cost.)
    double* sp = lo_prep_lds;
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const int i = lane + 64 * r;
      if (i < ninl) {
#pragma unroll
        for (int c = 0; c < 4; ++c) sp[i * 4 + c] = px[r][c];
      }
    }
    wv_sync();
    double sc = lds_chain_sum(sp + lane, 4, lane < 4 ? ninl : 0);
    wv_sync();
    const double cx1 = wv_readlane_f64(sc, 0) / ninl, cy1 = wv_readlane_f64(sc, 1) / ninl, cx2 = wv_readlane_f64(sc, 2) / ninl,
                 cy2 = wv_readlane_f64(sc, 3) / ninl;
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const int i = lane + 64 * r;
      const double dx1 = px[r][0] - cx1, dy1 = px[r][1] - cy1, dx2 = px[r][2] - cx2, dy2 = px[r][3] - cy2;
      if (i < ninl) {
        sp[i * 2 + 0] = dx1 * dx1 + dy1 * dy1;
        sp[i * 2 + 1] = dx2 * dx2 + dy2 * dy2;
      }
    }
    wv_sync();
    sc = lds_chain_sum(sp + (lane & 1), 2, lane < 2 ? ninl : 0);
    const double q1 = wv_readlane_f64(sc, 0), q2 = wv_readlane_f64(sc, 1);
    const double rms1 = sqrt(q1 / ninl), rms2 = sqrt(q2 / ninl);
    n1[0] = sqrt(2.0) / rms1; n1[1] = -n1[0] * cx1; n1[2] = -n1[0] * cy1;
    n2[0] = sqrt(2.0) / rms2; n2[1] = -n2[0] * cx2; n2[2] = -n2[0] * cy2;
  }
  double a[9][RPL];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) a[c][r] = 0.0;
  }
  if (FAM == FAM_E) {
    // EssentialMatrixFivePointEstimator::Estimate with all inliers, essential_matrix.cc:52-66
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const double x1_0 = px[r][0], x1_1 = px[r][1], x2_0 = px[r][2], x2_1 = px[r][3];
      a[0][r] = x1_0 * x2_0; a[1][r] = x1_1 * x2_0; a[2][r] = x2_0;
      a[3][r] = x1_0 * x2_1; a[4][r] = x1_1 * x2_1; a[5][r] = x2_1;
      a[6][r] = x1_0; a[7][r] = x1_1; a[8][r] = 1;
    }
  } else if (FAM == FAM_F) {
    // FundamentalMatrixEightPointEstimator::Estimate, fundamental_matrix.cc:150-171
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      double a0, a1, b0, b1;
      apply_norm(n1[0], n1[1], n1[2], px[r][0], px[r][1], &a0, &a1);
      apply_norm(n2[0], n2[1], n2[2], px[r][2], px[r][3], &b0, &b1);
      const double h[3] = {a0, a1, 1.0};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a[k][r] = h[k] * b0;
        a[3 + k][r] = h[k] * b1;
        a[6 + k][r] = h[k];
      }
    }
  } else {
    // HomographyMatrixEstimator::Estimate, homography_matrix.cc:44-82: row R < N from inlier R (x block), row
    // R >= N from inlier R - N (y block).  The owner of a row of the second block is not the lane that holds the
    // inlier's point, so the normalised points go through one rotation per slot pair.
    double sn[PPL][4];  // normalised (s_0, s_1, d_0, d_1) of this lane's inliers
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      apply_norm(n1[0], n1[1], n1[2], px[r][0], px[r][1], &sn[r][0], &sn[r][1]);
      apply_norm(n2[0], n2[1], n2[2], px[r][2], px[r][3], &sn[r][2], &sn[r][3]);
    }
    const int N = ninl;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      if (64 * r < m) {  // uniform: the shuffles below need every lane
        const int R = lane + 64 * r;
        const bool valid = R < m;
        const bool second = R >= N;
        const int src = valid ? (second ? R - N : R) : 0;  // inlier index: lane src & 63, slot src >> 6
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double got = 0.0;
#pragma unroll
          for (int sl = 0; sl < PPL; ++sl) {
            const double w = __shfl(sn[sl][k], src & 63);
            got = ((src >> 6) == sl) ? w : got;
          }
          v[k] = got;
        }
        const double s_0 = v[0], s_1 = v[1], d_0 = v[2], d_1 = v[3];
        if (valid && !second) {
          a[0][r] = -s_0; a[1][r] = -s_1; a[2][r] = -1;
          a[6][r] = s_0 * d_0; a[7][r] = s_1 * d_0; a[8][r] = d_0;
        } else if (valid) {
          a[3][r] = -s_0; a[4][r] = -s_1; a[5][r] = -1;
          a[6][r] = s_0 * d_1; a[7][r] = s_1 * d_1; a[8][r] = d_1;
        }
      }
    }
  }
  // wv_svd_prepare_mx9, m > 9: scale = max |a_ij| (exact, order independent), pivoted QR, W = R, V = permutation
  double mxl = 0.0;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    if (lane + 64 * r < m) {
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const double v = fabs(a[c][r]);
        if (v > mxl) mxl = v;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_xor(mxl, o);
    if (other > mxl) mxl = other;
  }
  double scale = mxl;
  if (scale == 0.0) scale = 1.0;
  // (the rows of an E / F design matrix end in an exact 1 and an H row carries three exact zeros: no group guard passes;
  // the plain divisions stay)
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    if (lane + 64 * r < m) {
#pragma unroll
      for (int c = 0; c < 9; ++c) a[c][r] /= scale;
    }
  }
  double hco[9];
  int perm[9];
  wr_colpiv_qr9<RPL>(a, m, lane, hco, perm);
  (void)hco;
  if (lane < 9) {
    int pc = 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      out[c * 9 + lane] = (lane <= c) ? a[c][0] : 0.0;  // W(i, j) = R(i, j) for i <= j
      pc = (lane == c) ? perm[c] : pc;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) out[81 + lane * 9 + i] = (i == pc) ? 1.0 : 0.0;  // V = the column permutation
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      out[162 + k] = n1[k];
      out[165 + k] = n2[k];
    }
    out[168] = scale;
    out[169] = 9.0;
  }
}

// LO step 1, wave per queued pair: the local estimator's constraint matrix over the pair's inlier list and its
// reduction to the square problem -- fam_local up to (not including) the Jacobi sweeps.
template <int FAM>
__global__ __launch_bounds__(64, 2) void k_lo_prepare(const VerifyParams p) {  // two waves per SIMD: no register spills (see k_replay_lo<TAIL>)
  __shared__ WvSvdShared svd;
  const int lane = threadIdx.x;
  const WgScratch ws = wg_scratch(p);
  SegGrab wgrab;
  const uint32_t grain = work_grain(p.n_work);
  for (;;) {
    wv_sync();
    const uint32_t widx = grab_seg(wgrab, GRAB_PREPARE(p), p.n_work, lane, grain);
    if (widx == GRAB_DONE) break;
    if (widx >= p.n_work) continue;
    const LoRef ref = lo_ref<FAM>(p, widx);
    const uint32_t pi = p.pair0 + ref.pl;
    const uint64_t moff = p.match_off[pi];
    const double* pts = (FAM == FAM_E ? p.pts_norm : p.pts_px) + 4 * moff;
    const int* inl = ref.inl;
    const int ninl = ref.ninl;
    if (p.lo_reg_prepare && lo_prepare_in_registers<FAM>(ninl)) continue;  // k_lo_prepare_reg's
    double* out = p.lo_work + (size_t)ref.slot * LO_WORK_DOUBLES;
    double n1[3] = {0, 0, 0}, n2[3] = {0, 0, 0};
    auto idx = [inl](int i) { return inl[i]; };
    int m;
    double* A = ws.tall;
    if (FAM == FAM_E) {
      // EssentialMatrixFivePointEstimator::Estimate with all inliers, essential_matrix.cc:52-66
      m = ninl;
      for (int i = lane; i < m; i += 64) {
        const double* q = pts + (size_t)inl[i] * 4;
        const double x1_0 = q[0], x1_1 = q[1], x2_0 = q[2], x2_1 = q[3];
        A[(size_t)0 * m + i] = x1_0 * x2_0; A[(size_t)1 * m + i] = x1_1 * x2_0; A[(size_t)2 * m + i] = x2_0;
        A[(size_t)3 * m + i] = x1_0 * x2_1; A[(size_t)4 * m + i] = x1_1 * x2_1; A[(size_t)5 * m + i] = x2_1;
        A[(size_t)6 * m + i] = x1_0; A[(size_t)7 * m + i] = x1_1; A[(size_t)8 * m + i] = 1;
      }
    } else {
      wv_center_and_normalize(pts, 0, ninl, idx, lane, &n1[0], &n1[1], &n1[2]);
      wv_center_and_normalize(pts, 1, ninl, idx, lane, &n2[0], &n2[1], &n2[2]);
      if (FAM == FAM_F) {
        // FundamentalMatrixEightPointEstimator::Estimate, fundamental_matrix.cc:150-171
        m = ninl;
        for (int i = lane; i < m; i += 64) {
          const double* q = pts + (size_t)inl[i] * 4;
          double a0, a1, b0, b1;
          apply_norm(n1[0], n1[1], n1[2], q[0], q[1], &a0, &a1);
          apply_norm(n2[0], n2[1], n2[2], q[2], q[3], &b0, &b1);
          const double h[3] = {a0, a1, 1.0};
          for (int k = 0; k < 3; ++k) {
            A[(size_t)k * m + i] = h[k] * b0;
            A[(size_t)(3 + k) * m + i] = h[k] * b1;
            A[(size_t)(6 + k) * m + i] = h[k];
          }
        }
      } else {
        // HomographyMatrixEstimator::Estimate, homography_matrix.cc:44-82
        const int N = ninl;
        m = 2 * ninl;
        for (int e = lane; e < 9 * m; e += 64) A[e] = 0.0;
        wv_sync();
        for (int i = lane; i < N; i += 64) {
          const double* q = pts + (size_t)inl[i] * 4;
          double s_0, s_1, d_0, d_1;
          apply_norm(n1[0], n1[1], n1[2], q[0], q[1], &s_0, &s_1);
          apply_norm(n2[0], n2[1], n2[2], q[2], q[3], &d_0, &d_1);
          const int j = N + i;
          A[(size_t)0 * m + i] = -s_0; A[(size_t)1 * m + i] = -s_1; A[(size_t)2 * m + i] = -1;
          A[(size_t)6 * m + i] = s_0 * d_0; A[(size_t)7 * m + i] = s_1 * d_0; A[(size_t)8 * m + i] = d_0;
          A[(size_t)3 * m + j] = -s_0; A[(size_t)4 * m + j] = -s_1; A[(size_t)5 * m + j] = -1;
          A[(size_t)6 * m + j] = s_0 * d_1; A[(size_t)7 * m + j] = s_1 * d_1; A[(size_t)8 * m + j] = d_1;
        }
      }
    }
    wv_sync();
    double scale;
    const int dsz = wv_svd_prepare_mx9(A, A + (size_t)9 * m, m, &svd, &scale, lane);
    for (int e = lane; e < 81; e += 64) {
      out[e] = svd.W[e];
      out[81 + e] = svd.V[e];
    }
    if (lane == 0) {
      for (int k = 0; k < 3; ++k) {
        out[162 + k] = n1[k];
        out[165 + k] = n2[k];
      }
      out[168] = scale;
      out[169] = (double)dsz;
    }
  }
}

// LO step 2, a group of LOJ_G lanes per queued pair (64 / LOJ_G pairs per wave): the Jacobi sweeps of JacobiSVD on the 9 x 9 (or
// smaller) problem; the sorted right factor V goes back to the pair's record.
#define LOJ_GROUP_DOUBLES (81 + 81 + 9)
#define LOJ_G 8  // lanes per problem
template <int FAM, bool SMALL_ONLY>
__global__ __launch_bounds__(64) void k_lo_jacobi(const VerifyParams p) {
  __shared__ double lds[(64 / LOJ_G) * LOJ_GROUP_DOUBLES];
  const int lane = threadIdx.x;
  const int g = lane / LOJ_G, gl = lane % LOJ_G;
  const uint32_t widx = blockIdx.x * (uint32_t)(64 / LOJ_G) + (uint32_t)g;
  if (widx >= p.n_work) return;
  const uint32_t slot = lo_ref<FAM>(p, widx).slot;
  const double* in = p.lo_work + (size_t)slot * LO_WORK_DOUBLES;
  grp_vd W = lds + g * LOJ_GROUP_DOUBLES;
  grp_vd V = W + 81;
  grp_vd sv = V + 81;
  const double scale = in[168];
  const int dsz = (int)in[169];
  if (SMALL_ONLY && dsz == 9) return;  // a full 9 x 9 problem belongs to k_lo_jacobi_reg
  for (int e = gl; e < 81; e += LOJ_G) {
    W[e] = in[e];
    V[e] = in[81 + e];
  }
  grp_jacobi_sweeps<LOJ_G>(W, V, dsz, scale, sv, gl);
  double* outV = p.lo_work + (size_t)slot * LO_WORK_DOUBLES + 81;  // sorted right factor back to the record
  for (int e = gl; e < 81; e += LOJ_G) outV[e] = V[e];
}

// The same step with a LANE per queued pair and the 9 x 9 problem in registers (pr_jacobi_sweeps9); problems that
// the preconditioner left smaller than 9 x 9 (fewer than nine constraint rows: at most 8 inliers) go to k_lo_jacobi.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_lo_jacobi_reg(const VerifyParams p) {
  const uint32_t widx = blockIdx.x * 64u + threadIdx.x;
  if (widx >= p.n_work) return;
  const uint32_t slot = p.worklist ? p.worklist[widx] : widx;  // the pair (chain) or the job (item pass): lo_ref's slot
  double* rec = p.lo_work + (size_t)slot * LO_WORK_DOUBLES;
  if ((int)rec[169] != 9) return;
  double W[81], V[81], sv[9];
#pragma unroll
  for (int e = 0; e < 81; ++e) {
    W[e] = rec[e];
    V[e] = rec[81 + e];
  }
  pr_jacobi_sweeps9(W, V, rec[168], sv);
#pragma unroll
  for (int e = 0; e < 81; ++e) rec[81 + e] = V[e];  // sorted right factor back to the pair's record
}

// LO step 2b (F, H), lane per queued pair: the family's finish on the null vector V(:, 8) -- rank-2 projection +
// de-normalisation (8-point F, fundamental_matrix.cc:172-191) or de-normalisation (H, homography_matrix.cc:86-91).
template <int FAM>
__global__ __launch_bounds__(64) void k_lo_finish(const VerifyParams p) {
  const uint32_t widx = blockIdx.x * 64u + threadIdx.x;
  if (widx >= p.n_work) return;
  const LoRef ref = lo_ref<FAM>(p, widx);
  const double* in = p.lo_work + (size_t)ref.slot * LO_WORK_DOUBLES;
  double nv[9], n1[3], n2[3], model[9];
  for (int k = 0; k < 9; ++k) nv[k] = in[81 + 8 * 9 + k];
  for (int k = 0; k < 3; ++k) {
    n1[k] = in[162 + k];
    n2[k] = in[165 + k];
  }
  if (FAM == FAM_F)
    eight_point_finish(nv, n1, n2, model);
  else
    homography_finish(nv, n1, n2, model);
  double* om = p.lo_models + (size_t)ref.slot * 90;
  for (int k = 0; k < 9; ++k) om[k] = model[k];
  *ref.nm = 1;
}

// LO step 3 (E only), lane per queued pair: the 5-point solver from the null-space basis on, the same device
// functions as the minimal-sample kernels (k_solve_e_build / _lu / k_roots_e / k_models_score_e).
__global__ __launch_bounds__(64, 2) void k_lo_e_build(const VerifyParams p) {
  const uint32_t widx = blockIdx.x * 64u + threadIdx.x;
  if (widx >= p.n_work) return;
  const uint32_t sl = p.worklist ? p.worklist[widx] : widx;  // lo_ref's slot
  double Eb[36];
  const double* V = p.lo_work + (size_t)sl * LO_WORK_DOUBLES + 81;
  double* slot = p.lo_slots + (size_t)sl * 90;
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 4; ++c) Eb[r * 4 + c] = V[(5 + c) * 9 + r];  // Eb[r*4 + c] = V(r, 5 + c), essential_matrix.cc:72-74
  for (int k = 0; k < 36; ++k) slot[EPOLY_EB + k] = Eb[k];
  five_point_build_A<1>(Eb, p.lo_ework + (size_t)sl * 200);
}
__global__ __launch_bounds__(64) void k_lo_e_lu(const VerifyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Al = reinterpret_cast<double*>(smem_raw) + threadIdx.x;
  unsigned char* idx = smem_raw + 100 * 64 * 8 + threadIdx.x;
  const uint32_t widx = blockIdx.x * 64u + threadIdx.x;
  if (widx >= p.n_work) return;
  const uint32_t sl = p.worklist ? p.worklist[widx] : widx;  // lo_ref's slot
  e_lu_body(p.lo_ework + (size_t)sl * 200, p.lo_slots + (size_t)sl * 90, Al, idx);
}
__global__ __launch_bounds__(64, 2) void k_lo_e_lu_reg(const VerifyParams p) {
  const uint32_t widx = blockIdx.x * 64u + threadIdx.x;
  if (widx >= p.n_work) return;
  const uint32_t sl = p.worklist ? p.worklist[widx] : widx;  // lo_ref's slot
  e_lu_body_reg(p.lo_ework + (size_t)sl * 200, p.lo_slots + (size_t)sl * 90);
}
__global__ __launch_bounds__(64) void k_lo_e_roots_models(const VerifyParams p) {
  const uint32_t widx = blockIdx.x * 64u + threadIdx.x;
  if (widx >= p.n_work) return;
  const LoRef ref = lo_ref<FAM_E>(p, widx);
  double* slot = p.lo_slots + (size_t)ref.slot * 90;
  const int code = e_roots_body(slot);
  const int nm = e_models_body(slot, code, p.lo_models + (size_t)ref.slot * 90);
  *ref.nm = (uint32_t)nm;
}

void launch_vp_replay_lo(const VerifyParams& p, int fam, uint32_t n_blocks, int mode, hipStream_t st) {
  if (!p.n_work || !n_blocks) return;
  const size_t smem = ((offsetof(VSmem, gen) + 15) / 16) * 16;
  if (mode != 1 && !p.replay_legacy) {  // the scans with the pair resident in LDS
    const size_t rs = rp_lds_bytes(p.rp_cap, p.n_max);
    if (mode == 2) {
      if (fam == FAM_E) hipLaunchKernelGGL((k_replay_rp<FAM_E, true>), dim3(n_blocks), dim3(64), rs, st, p);
      if (fam == FAM_F) hipLaunchKernelGGL((k_replay_rp<FAM_F, true>), dim3(n_blocks), dim3(64), rs, st, p);
      if (fam == FAM_H) hipLaunchKernelGGL((k_replay_rp<FAM_H, true>), dim3(n_blocks), dim3(64), rs, st, p);
    } else {
      if (fam == FAM_E) hipLaunchKernelGGL((k_replay_rp<FAM_E, false>), dim3(n_blocks), dim3(64), rs, st, p);
      if (fam == FAM_F) hipLaunchKernelGGL((k_replay_rp<FAM_F, false>), dim3(n_blocks), dim3(64), rs, st, p);
      if (fam == FAM_H) hipLaunchKernelGGL((k_replay_rp<FAM_H, false>), dim3(n_blocks), dim3(64), rs, st, p);
    }
    return;
  }
  if (mode == 1) {  // F and H only: the E instance would still spill 134 VGPRs at 512 (its 16-lane 5-point finish is inlined)
    if (fam == FAM_F) hipLaunchKernelGGL((k_replay_lo<FAM_F, 1>), dim3(n_blocks), dim3(64), smem, st, p);
    if (fam == FAM_H) hipLaunchKernelGGL((k_replay_lo<FAM_H, 1>), dim3(n_blocks), dim3(64), smem, st, p);
  }
#ifdef DSM_CHECK_BUILD
  else if (mode == 2) {
    if (fam == FAM_E) hipLaunchKernelGGL((k_replay_lo<FAM_E, 2>), dim3(n_blocks), dim3(64), smem, st, p);
    if (fam == FAM_F) hipLaunchKernelGGL((k_replay_lo<FAM_F, 2>), dim3(n_blocks), dim3(64), smem, st, p);
    if (fam == FAM_H) hipLaunchKernelGGL((k_replay_lo<FAM_H, 2>), dim3(n_blocks), dim3(64), smem, st, p);
  } else {
    if (fam == FAM_E) hipLaunchKernelGGL((k_replay_lo<FAM_E, 0>), dim3(n_blocks), dim3(64), smem, st, p);
    if (fam == FAM_F) hipLaunchKernelGGL((k_replay_lo<FAM_F, 0>), dim3(n_blocks), dim3(64), smem, st, p);
    if (fam == FAM_H) hipLaunchKernelGGL((k_replay_lo<FAM_H, 0>), dim3(n_blocks), dim3(64), smem, st, p);
  }
#endif
}
// item pass, step by step (the host reads the job count between enum and the rest)
void launch_vp_items_enum(const VerifyParams& p, int fam, hipStream_t st) {
  if (!p.n_work) return;
  const uint32_t ng = (p.n_work + ITEMS_GROUP - 1) / ITEMS_GROUP;
  const uint32_t ne = ng < 8192u ? ng : 8192u;
  if (fam == FAM_E) hipLaunchKernelGGL(k_items_enum<FAM_E>, dim3(ne), dim3(64 * ITEMS_GROUP), 0, st, p);
  if (fam == FAM_F) hipLaunchKernelGGL(k_items_enum<FAM_F>, dim3(ne), dim3(64 * ITEMS_GROUP), 0, st, p);
  if (fam == FAM_H) hipLaunchKernelGGL(k_items_enum<FAM_H>, dim3(ne), dim3(64 * ITEMS_GROUP), 0, st, p);
}
// p.n_work = number of jobs, p.job_list their slots
void launch_vp_items_inliers(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st) {
  if (!p.n_work || !n_blocks) return;
  const uint32_t nb = p.n_work < n_blocks ? p.n_work : n_blocks;
  if (fam == FAM_E) hipLaunchKernelGGL(k_items_inliers<FAM_E>, dim3(nb), dim3(64), 0, st, p);
  if (fam == FAM_F) hipLaunchKernelGGL(k_items_inliers<FAM_F>, dim3(nb), dim3(64), 0, st, p);
  if (fam == FAM_H) hipLaunchKernelGGL(k_items_inliers<FAM_H>, dim3(nb), dim3(64), 0, st, p);
}
void launch_vp_items_outcome(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st) {
  if (!p.n_work || !n_blocks) return;
  const uint32_t nb = p.n_work < n_blocks ? p.n_work : n_blocks;
  if (fam == FAM_E) hipLaunchKernelGGL(k_items_outcome<FAM_E>, dim3(nb), dim3(64), 0, st, p);
  if (fam == FAM_F) hipLaunchKernelGGL(k_items_outcome<FAM_F>, dim3(nb), dim3(64), 0, st, p);
  if (fam == FAM_H) hipLaunchKernelGGL(k_items_outcome<FAM_H>, dim3(nb), dim3(64), 0, st, p);
}
void launch_vp_local_opt(const VerifyParams& p, int fam, uint32_t n_blocks, uint32_t n_wave_prepare, uint32_t n_small_jacobi,
                         uint32_t n_big_prepare, hipStream_t st) {
  if (!p.n_work || !n_blocks) return;
  const dim3 g4((p.n_work + 64 / LOJ_G - 1) / (64 / LOJ_G)), g64((p.n_work + 63) / 64);
#ifdef DSM_CHECK_BUILD
  const bool reg_jacobi = !p.dbg_jacobi_groups;  // =1: the 8-lane-group kernel for every problem (round-2 form)
#else
  const bool reg_jacobi = true;
  (void)g4;
#endif
  const bool reg_prepare = p.lo_reg_prepare && n_wave_prepare < p.n_work;
  // the general kernels work through k_replay_lo's list of the problems that need them (a handful per iteration: the
  // first local optimisations of a pair have 6 - 9 inliers), not through the whole queue
  VerifyParams pg = p;
  pg.worklist = p.lo_queue_g;
  pg.n_work = n_wave_prepare;
  const uint32_t nb_prep = pg.n_work < n_blocks ? pg.n_work : n_blocks;  // one scratch area per workgroup (wg_scratch)
  const dim3 g4g((pg.n_work + 64 / LOJ_G - 1) / (64 / LOJ_G));
  if (fam == FAM_E) {
    if (reg_prepare && n_big_prepare < p.n_work) hipLaunchKernelGGL((k_lo_prepare_reg<FAM_E, LOP_PPL_SMALL>), dim3(p.n_work), dim3(64), 0, st, p);
    if (reg_prepare && n_big_prepare) hipLaunchKernelGGL((k_lo_prepare_reg<FAM_E, LOP_PPL>), dim3(p.n_work), dim3(64), 0, st, p);
    if (n_wave_prepare) hipLaunchKernelGGL(k_lo_prepare<FAM_E>, dim3(nb_prep), dim3(64), 0, st, pg);
    if (reg_jacobi) {
      hipLaunchKernelGGL(k_lo_jacobi_reg, g64, dim3(64), 0, st, p);
      if (n_small_jacobi) hipLaunchKernelGGL((k_lo_jacobi<FAM_E, true>), g4g, dim3(64), 0, st, pg);
    }
#ifdef DSM_CHECK_BUILD
    else
      hipLaunchKernelGGL((k_lo_jacobi<FAM_E, false>), g4, dim3(64), 0, st, p);
#endif
    hipLaunchKernelGGL(k_lo_e_build, g64, dim3(64), 0, st, p);
    if (p.dbg_elu_lds)
      hipLaunchKernelGGL(k_lo_e_lu, g64, dim3(64), ELU_SMEM, st, p);
    else
      hipLaunchKernelGGL(k_lo_e_lu_reg, g64, dim3(64), 0, st, p);
    hipLaunchKernelGGL(k_lo_e_roots_models, g64, dim3(64), 0, st, p);
  }
  if (fam == FAM_F) {
    if (reg_prepare && n_big_prepare < p.n_work) hipLaunchKernelGGL((k_lo_prepare_reg<FAM_F, LOP_PPL_SMALL>), dim3(p.n_work), dim3(64), 0, st, p);
    if (reg_prepare && n_big_prepare) hipLaunchKernelGGL((k_lo_prepare_reg<FAM_F, LOP_PPL>), dim3(p.n_work), dim3(64), 0, st, p);
    if (n_wave_prepare) hipLaunchKernelGGL(k_lo_prepare<FAM_F>, dim3(nb_prep), dim3(64), 0, st, pg);
    if (reg_jacobi) {
      hipLaunchKernelGGL(k_lo_jacobi_reg, g64, dim3(64), 0, st, p);
      if (n_small_jacobi) hipLaunchKernelGGL((k_lo_jacobi<FAM_F, true>), g4g, dim3(64), 0, st, pg);
    }
#ifdef DSM_CHECK_BUILD
    else
      hipLaunchKernelGGL((k_lo_jacobi<FAM_F, false>), g4, dim3(64), 0, st, p);
#endif
    hipLaunchKernelGGL(k_lo_finish<FAM_F>, g64, dim3(64), 0, st, p);
  }
  if (fam == FAM_H) {
    if (reg_prepare && n_big_prepare < p.n_work) hipLaunchKernelGGL((k_lo_prepare_reg<FAM_H, LOP_PPL_H>), dim3(p.n_work), dim3(64), 0, st, p);
    if (reg_prepare && n_big_prepare) hipLaunchKernelGGL((k_lo_prepare_reg<FAM_H, LOP_PPL_H_BIG>), dim3(p.n_work), dim3(64), 0, st, p);
    if (n_wave_prepare) hipLaunchKernelGGL(k_lo_prepare<FAM_H>, dim3(nb_prep), dim3(64), 0, st, pg);
    if (reg_jacobi) {
      hipLaunchKernelGGL(k_lo_jacobi_reg, g64, dim3(64), 0, st, p);
      if (n_small_jacobi) hipLaunchKernelGGL((k_lo_jacobi<FAM_H, true>), g4g, dim3(64), 0, st, pg);
    }
#ifdef DSM_CHECK_BUILD
    else
      hipLaunchKernelGGL((k_lo_jacobi<FAM_H, false>), g4, dim3(64), 0, st, p);
#endif
    hipLaunchKernelGGL(k_lo_finish<FAM_H>, g64, dim3(64), 0, st, p);
  }
}

// The lanes' counter traffic as ONE-WAVE kernels.  hipMemcpyAsync (device -> pinned host) and hipMemsetAsync are blit kernels of the
// runtime with 512- / 256-thread workgroups; a lane issues one after every replay launch, and while the OTHER lane has a kernel in
// flight that fills every SIMD's register file (k_solve<H>, k_roots_e, k_final_pose, k_prescore_h2: tens of ms each) an eight-wave
// workgroup never finds its eight slots on one CU free at the same time -- the trace of a config-2 step shows two such 128-byte copies
// taking 15 - 20 ms and 7 - 16 ms, the lane's chain of short launches parked behind them (profiles/r06_lane_counter_kernels.txt).  A
// single wave is handed the first slot that frees up (k_lo_jacobi_reg, 512 VGPRs, starts within microseconds in the same situation).
__global__ __launch_bounds__(64) void k_lane_counters(const uint32_t* src, uint32_t* dst_host, uint32_t n_copy, uint4* zero, uint32_t n_zero16) {
  const uint32_t lane = threadIdx.x;
  // read-back first: the copied words may lie inside the zeroed range
  for (uint32_t i = lane; i < n_copy; i += 64) __hip_atomic_store(dst_host + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (uint32_t i = lane; i < n_zero16; i += 64) zero[i] = make_uint4(0u, 0u, 0u, 0u);
  __threadfence_system();
}
void launch_lane_counters(const void* src, uint32_t* dst_host, uint32_t copy_bytes, void* zero, size_t zero_bytes, hipStream_t st) {
  // zero_bytes is rounded up to 16 (every caller's range ends inside the lane's counter block, which is padded for it)
  hipLaunchKernelGGL(k_lane_counters, dim3(1), dim3(64), 0, st, static_cast<const uint32_t*>(src), dst_host, copy_bytes / 4u,
                     static_cast<uint4*>(zero), (uint32_t)((zero_bytes + 15) / 16));
}

void launch_vp_prep(const VerifyParams& p, uint32_t n_blocks, hipStream_t st) {
  if (!p.n_pairs || !n_blocks) return;
  hipLaunchKernelGGL(k_verify_prep, dim3(n_blocks), dim3(64), 0, st, p);
}
void launch_vp_sample(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st) {
  if (!p.n_chunk || !n_blocks) return;
  const size_t smem = SAMPLER_PREFIX + (size_t)(p.n_max > 0 ? p.n_max : 1) * 4;
  if (fam == FAM_E) hipLaunchKernelGGL(k_sample<FAM_E>, dim3(n_blocks), dim3(64), smem, st, p);
  if (fam == FAM_F) hipLaunchKernelGGL(k_sample<FAM_F>, dim3(n_blocks), dim3(64), smem, st, p);
  if (fam == FAM_H) hipLaunchKernelGGL(k_sample<FAM_H>, dim3(n_blocks), dim3(64), smem, st, p);
}
void launch_vp_solve_score(const VerifyParams& p, int fam, hipStream_t st) {
  if (!p.n_chunk) return;
  const dim3 grid(p.n_chunk, (p.batch + 63) / 64);
  // the lane-per-hypothesis solvers: the compact grid (hyp_of_lane) when the round has a list; its non-empty workgroups come first
  const dim3 grid_hyp = p.hyp_map ? dim3((uint32_t)GRAB_SEGS * (p.hyp_seg_cap / 64u)) : grid;
  const size_t smem = (size_t)(p.n_max < VP_LDS_PTS ? (p.n_max > 0 ? p.n_max : 1) : VP_LDS_PTS) * 32;
  // the bound steps with an f32 first stage: the points as f32 (16 bytes each, padded to a pair) + the lanes' lists of band points
  const size_t smem_c2 = (size_t)(p.n_max < VP_LDS_PTS ? (p.n_max > 0 ? p.n_max : 1) : VP_LDS_PTS) * 16 + 32 + (size_t)PRESCORE_LIST_CAP * 64 * 2;
  if (fam == FAM_E) {
    if (p.hyp_map)
      hipLaunchKernelGGL(k_solve_e_build<true>, grid_hyp, dim3(64), 0, st, p);
    else
      hipLaunchKernelGGL(k_solve_e_build<false>, grid, dim3(64), 0, st, p);
    // (the elimination stays on the pair grid: it streams the 1.6 KB of e_work per hypothesis from a wave-uniform base + lane, and with a
    // base per lane it was 14 % slower -- 6.4 vs 5.6 ms per step at config 2 -- for the sixth of its lanes the compact grid would fill)
    if (p.dbg_elu_lds)  // check build, DSM_ELU_LDS: the elimination in lane-interleaved LDS (rounds 2 - 5)
      hipLaunchKernelGGL(k_solve_e_lu, grid, dim3(64), ELU_SMEM, st, p);
    else
      hipLaunchKernelGGL(k_solve_e_lu_reg, grid, dim3(64), 0, st, p);
#ifdef DSM_CHECK_BUILD
    if (p.dbg_roots_lds != 0) {
      if (p.hyp_map)
        hipLaunchKernelGGL(k_roots_e_lds<true>, grid_hyp, dim3(64), 100 * 64 * sizeof(double), st, p);
      else
        hipLaunchKernelGGL(k_roots_e_lds<false>, grid, dim3(64), 100 * 64 * sizeof(double), st, p);
    } else
#endif
    if (p.hyp_map)
      hipLaunchKernelGGL(k_roots_e<true>, grid_hyp, dim3(64), 0, st, p);
    else
      hipLaunchKernelGGL(k_roots_e<false>, grid, dim3(64), 0, st, p);
    // scoring: a lane per model (k_prescore_compact<E> -> k_score_needed<E>), or the wave-per-hypothesis kernel with the bound step fused in
    // (DSM_SCORE_PREFILTER=3; also what =0 runs, without its bound step)
    const size_t smem2e = smem + (size_t)p.batch * 10 * 2;
    const uint32_t nb_needed_e = p.n_chunk < 256u * 32u ? p.n_chunk : 256u * 32u;
    if ((p.score_prefilter & 1) && !(p.score_prefilter & 2) && p.batch * 10 <= 65535 && smem2e <= 64 * 1024) {  // 1, and 5 = check
      // the bound step with its packed-f32 first stage (k_prescore_compact2); check build, DSM_SCORE_PREFILTER=33: the pure FP64 form
#ifdef DSM_CHECK_BUILD
      if (p.score_prefilter & 32)
        hipLaunchKernelGGL(k_prescore_compact<FAM_E>, dim3(p.n_chunk, (p.batch * 10 + 63) / 64), dim3(64), smem, st, p);
      else
#endif
        hipLaunchKernelGGL(k_prescore_compact2<FAM_E>, dim3(p.n_chunk, (p.batch * 10 + 63) / 64), dim3(64), smem_c2, st, p);
      hipLaunchKernelGGL(k_score_needed<FAM_E>, dim3(nb_needed_e), dim3(64), smem2e, st, p);
#ifdef DSM_CHECK_BUILD
    } else if (p.score_prefilter & 4) {
      hipLaunchKernelGGL(k_models_score_e<true>, grid, dim3(64), smem, st, p);
#endif
    } else {
      hipLaunchKernelGGL(k_models_score_e<false>, grid, dim3(64), smem, st, p);
    }
  }
  // scoring: bound + exact (k_prescore, k_score_needed) unless DSM_SCORE_PREFILTER=0; the slot list of k_score_needed must
  // fit 16-bit indices and, with the points, the LDS
  const uint32_t nb_needed = p.n_chunk < 256u * 32u ? p.n_chunk : 256u * 32u;
  if (fam == FAM_F) {
    if (p.hyp_map)
      hipLaunchKernelGGL((k_solve<FAM_F, true>), grid_hyp, dim3(64), 0, st, p);
    else
      hipLaunchKernelGGL((k_solve<FAM_F, false>), grid, dim3(64), 0, st, p);
    const size_t smem2 = smem + (size_t)p.batch * 3 * 2;
    if (p.score_prefilter && p.batch * 3 <= 65535 && smem2 <= 64 * 1024) {
      if (p.score_prefilter & 2)  // DSM_SCORE_PREFILTER=3: a lane per slot
        hipLaunchKernelGGL(k_prescore<FAM_F>, dim3(p.n_chunk, (p.batch * 3 + 63) / 64), dim3(64), smem, st, p);
#ifdef DSM_CHECK_BUILD
      else if (p.score_prefilter & 32)
        hipLaunchKernelGGL(k_prescore_compact<FAM_F>, dim3(p.n_chunk, (p.batch * 3 + 63) / 64), dim3(64), smem, st, p);
#endif
      else
        hipLaunchKernelGGL(k_prescore_compact2<FAM_F>, dim3(p.n_chunk, (p.batch * 3 + 63) / 64), dim3(64), smem_c2, st, p);
      hipLaunchKernelGGL(k_score_needed<FAM_F>, dim3(nb_needed), dim3(64), smem2, st, p);
    } else {
      hipLaunchKernelGGL(k_score<FAM_F>, dim3(p.n_chunk, (p.batch * 3 + 63) / 64), dim3(64), smem, st, p);
    }
  }
  if (fam == FAM_H) {
    hipLaunchKernelGGL(k_solve<FAM_H>, grid, dim3(64), 0, st, p);
    const size_t smem2 = smem + (size_t)p.batch * 2;
    if (p.score_prefilter && p.batch <= 65535 && smem2 <= 64 * 1024) {
      // the bound step with its packed-f32 first stage (k_prescore_h2: the points once more as f32 behind the FP64 copy in the LDS).
      // Check build: DSM_SCORE_PREFILTER=17 the pure FP64 k_prescore<H> (round 4's form), =9 its K = 3 products on the FP64 matrix
      // pipe (k_prescore_h_mfma: measured slower)
      const size_t smem_h2 = (size_t)(p.n_max < VP_LDS_PTS ? (p.n_max > 0 ? p.n_max : 1) : VP_LDS_PTS) * 16 + 32 + (size_t)PRESCORE_LIST_CAP * 64 * 2;
#ifdef DSM_CHECK_BUILD
      if (p.score_prefilter & 8)
        hipLaunchKernelGGL(k_prescore_h_mfma, grid, dim3(64), smem, st, p);
      else if (p.score_prefilter & 16)
        hipLaunchKernelGGL(k_prescore<FAM_H>, grid, dim3(64), smem, st, p);
      else
#endif
        hipLaunchKernelGGL(k_prescore_h2, grid, dim3(64), smem_h2, st, p);
      hipLaunchKernelGGL(k_score_needed<FAM_H>, dim3(nb_needed), dim3(64), smem2, st, p);
    } else {
      hipLaunchKernelGGL(k_score<FAM_H>, grid, dim3(64), smem, st, p);
    }
  }
}
void launch_vp_replay(const VerifyParams& p, int fam, uint32_t n_blocks, hipStream_t st) {
  if (!p.n_chunk || !n_blocks) return;
  const size_t smem = ((offsetof(VSmem, gen) + 15) / 16) * 16;  // the replay needs no generator / sampler state
  if (fam == FAM_E) hipLaunchKernelGGL(k_replay<FAM_E>, dim3(n_blocks), dim3(64), smem, st, p);
  if (fam == FAM_F) hipLaunchKernelGGL(k_replay<FAM_F>, dim3(n_blocks), dim3(64), smem, st, p);
  if (fam == FAM_H) hipLaunchKernelGGL(k_replay<FAM_H>, dim3(n_blocks), dim3(64), smem, st, p);
}
static void launch_final_pose_finish(const VerifyParams& p, uint32_t n_blocks, hipStream_t st) {
  const uint32_t n = p.final_list ? p.n_final : p.n_chunk;
  if (!n) return;
  const uint32_t items = n * 4u;
  hipLaunchKernelGGL(k_final_pose, dim3(items < n_blocks ? items : n_blocks), dim3(64), 0, st, p);
  hipLaunchKernelGGL(k_final_finish, dim3((n + 63) / 64), dim3(64), 0, st, p);
}
void launch_vp_final(const VerifyParams& p, uint32_t n_blocks, hipStream_t st) {
  if (!p.n_pairs || !n_blocks) return;
#ifdef DSM_CHECK_BUILD
  if (p.dbg_final_waves == 1)
    hipLaunchKernelGGL(k_verify_final<1>, dim3(n_blocks), dim3(64), verify_smem_bytes(p.n_max), st, p);
  else
#endif
    hipLaunchKernelGGL(k_verify_final<2>, dim3(n_blocks), dim3(64), verify_smem_bytes(p.n_max), st, p);
  launch_final_pose_finish(p, n_blocks, st);
}

void debug_read_prof(unsigned long long* out16) {
#ifdef DSM_PROFILE_SECTIONS
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dsm_prof), 16 * sizeof(unsigned long long));
  unsigned long long zero[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dsm_prof), zero, sizeof(zero));
#else
  for (int i = 0; i < 16; ++i) out16[i] = 0;
#endif
}

// Compaction of the per-pair inlier matches (stored at the pair's match offset) into list order.
__global__ __launch_bounds__(64) void k_compact_inliers(const uint64_t* match_off, const uint64_t* inl_off,
                                                        const uint32_t* inl_counts, const uint32_t* src, uint32_t* dst,
                                                        uint32_t n_pairs) {
  for (uint32_t pi = blockIdx.x; pi < n_pairs; pi += gridDim.x) {
    const uint2* s = reinterpret_cast<const uint2*>(src) + match_off[pi];
    uint2* d = reinterpret_cast<uint2*>(dst) + inl_off[pi];
    for (uint32_t i = threadIdx.x; i < inl_counts[pi]; i += 64) d[i] = s[i];
  }
}
void launch_compact_inliers(const uint64_t* match_off, const uint64_t* inl_off, const uint32_t* inl_counts,
                            const uint32_t* src, uint32_t* dst, uint32_t n_pairs, hipStream_t st) {
  if (!n_pairs) return;
  const uint32_t blocks = n_pairs < 8192 ? n_pairs : 8192;
  hipLaunchKernelGGL(k_compact_inliers, dim3(blocks), dim3(64), 0, st, match_off, inl_off, inl_counts, src, dst, n_pairs);
}

// ------------------------------------------------------------------------------------ EstimateMultiple
// TwoViewGeometry::EstimateMultiple (two_view_geometry.cc:128-167) is a loop of Estimate() over the matches
// that are not inliers of the geometries found so far; the host repeats the whole verification pipeline over
// all pairs (finished pairs carry zero matches) and these kernels do the per-pair bookkeeping between passes.
__global__ __launch_bounds__(64) void k_multi_accumulate(const MultiParams p) {
  const int lane = threadIdx.x;
  for (uint32_t pi = blockIdx.x; pi < p.n_pairs; pi += gridDim.x) {
    MultiState* ms = p.state + pi;
    const uint64_t off = p.cur_off[pi];
    const int n = (int)(p.cur_off[pi + 1] - off);
    __syncthreads();
    if (ms->done) {
      if (lane == 0) p.next_count[pi] = 0;
      continue;
    }
    const dsm_two_view_geometry* t = p.tvg + pi;
    const int config = t->config;
    if (lane < 4) {
      ms->trials[lane] += t->num_trials[lane];
      ms->models[lane] += t->num_models[lane];
    }
    if (config == DSM_CONFIG_DEGENERATE) {  // :136-138
      if (lane == 0) {
        ms->done = 1;
        p.next_count[pi] = 0;
      }
      continue;
    }
    const uint32_t ninl = p.inl_counts[pi];
    const uint2* inl = reinterpret_cast<const uint2*>(p.inl) + off;
    const bool accept = !(p.ignore_watermark && config == DSM_CONFIG_WATERMARK);  // :140-146
    const uint32_t acc0 = ms->acc_inl, ngeo0 = ms->ngeo;
    __syncthreads();
    if (accept) {
      uint2* acc = reinterpret_cast<uint2*>(p.acc) + p.orig_off[pi] + acc0;
      for (uint32_t i = lane; i < ninl; i += 64) acc[i] = inl[i];
      if (ngeo0 == 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(t);
        uint32_t* dst = reinterpret_cast<uint32_t*>(p.first + pi);
        for (uint32_t i = lane; i < sizeof(dsm_two_view_geometry) / 4; i += 64) dst[i] = src[i];
      }
      if (lane == 0) {
        ms->acc_inl = acc0 + ninl;
        ms->ngeo = ngeo0 + 1;
      }
    }
    // ExtractOutlierMatches (:67-88): a match stays unless the same (idx1, idx2) pair is an inlier match.  The
    // inlier list is an ordered subsequence of the matches; when idx1 is strictly ascending (every list the
    // matcher produces) a binary search finds it, otherwise every inlier is compared.
    const uint2* m = reinterpret_cast<const uint2*>(p.cur_matches) + off;
    bool asc = true;
    for (int i = lane; i + 1 < n; i += 64) asc = asc && (m[i].x < m[i + 1].x);
    const bool sorted = __ballot(!asc) == 0ull;
    uint32_t cnt = 0;
    for (int b0 = 0; b0 < n; b0 += 64) {
      const int i = b0 + lane;
      bool keep = false;
      if (i < n) {
        const uint2 q = m[i];
        bool found = false;
        if (sorted) {
          uint32_t lo = 0, hi = ninl;
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (inl[mid].x < q.x) lo = mid + 1; else hi = mid;
          }
          found = lo < ninl && inl[lo].x == q.x && inl[lo].y == q.y;
        } else {
          for (uint32_t k = 0; k < ninl && !found; ++k) found = inl[k].x == q.x && inl[k].y == q.y;
        }
        keep = !found;
        p.keep[off + i] = keep ? 1 : 0;
      }
      cnt += (uint32_t)__popcll(__ballot(keep));
    }
    if (lane == 0) {
      if (cnt == (uint32_t)n) {
        // not DEGENERATE, yet no match removed (min_num_inliers = 0, F failed, H succeeded, inliers from F's empty
        // mask): the reference would repeat the pass on the same matches indefinitely; the pair ends here, like in the
        // oracle (oracle/two_view.cc, EstimateMultiple)
        ms->done = 1;
        p.next_count[pi] = 0;
      } else {
        p.next_count[pi] = cnt;
        atomicAdd(p.active, 1u);
      }
    }
  }
}

__global__ __launch_bounds__(64) void k_multi_scatter(const MultiParams p) {
  const int lane = threadIdx.x;
  for (uint32_t pi = blockIdx.x; pi < p.n_pairs; pi += gridDim.x) {
    const uint32_t cnt = p.next_count[pi];
    if (cnt == 0) continue;
    const uint64_t off = p.cur_off[pi];
    const int n = (int)(p.cur_off[pi + 1] - off);
    const uint2* m = reinterpret_cast<const uint2*>(p.cur_matches) + off;
    uint2* dst = reinterpret_cast<uint2*>(p.next_matches) + p.next_off[pi];
    uint32_t total = 0;
    for (int b0 = 0; b0 < n; b0 += 64) {
      const int i = b0 + lane;
      const bool keep = i < n && p.keep[off + i] != 0;
      const unsigned long long bal = __ballot(keep);
      if (keep) dst[total + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = m[i];
      total += (uint32_t)__popcll(bal);
    }
  }
}

// :154-166 -- no geometry: DEGENERATE; one: that geometry; several: config MULTIPLE with the inlier matches of
// all of them, everything else as in a fresh TwoViewGeometry().  Then SiftFeatureMatcher::Match's post-filter.
__global__ __launch_bounds__(64) void k_multi_finalize(const MultiParams p) {
  const int lane = threadIdx.x;
  for (uint32_t pi = blockIdx.x; pi < p.n_pairs; pi += gridDim.x) {
    const MultiState ms = p.state[pi];
    dsm_two_view_geometry* out = p.out_tvg + pi;
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
    const uint32_t n_orig = (uint32_t)(p.orig_off[pi + 1] - p.orig_off[pi]);
    if (ms.ngeo == 1) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(p.first + pi);
      for (uint32_t i = lane; i < sizeof(dsm_two_view_geometry) / 4; i += 64) o32[i] = src[i];
    } else {
      for (uint32_t i = lane; i < sizeof(dsm_two_view_geometry) / 4; i += 64) o32[i] = 0;
    }
    __syncthreads();
    uint32_t ninl = ms.acc_inl;
    if (lane == 0) {
      if (ms.ngeo == 0) out->config = DSM_CONFIG_DEGENERATE;
      if (ms.ngeo > 1) out->config = DSM_CONFIG_MULTIPLE;
      out->num_matches = n_orig;
      out->num_inliers = ninl;
      for (int k = 0; k < 4; ++k) {
        out->num_trials[k] = ms.trials[k];
        out->num_models[k] = ms.models[k];
      }
      if (p.stage_filter && (uint64_t)ninl < p.min_num_inliers) {  // matching.cc:824-831
        const dsm_two_view_geometry keepc = *out;
        uint32_t* z = reinterpret_cast<uint32_t*>(out);
        for (uint32_t i = 0; i < sizeof(dsm_two_view_geometry) / 4; ++i) z[i] = 0;
        out->num_matches = keepc.num_matches;
        for (int k = 0; k < 4; ++k) {
          out->num_trials[k] = keepc.num_trials[k];
          out->num_models[k] = keepc.num_models[k];
        }
        ninl = 0;
      }
      p.out_inl_counts[pi] = ninl;
    }
    __syncthreads();
  }
}
static uint32_t multi_blocks(uint32_t n) { return n < 16384u ? (n ? n : 1u) : 16384u; }
void launch_multi_accumulate(const MultiParams& p, hipStream_t st) {
  hipLaunchKernelGGL(k_multi_accumulate, dim3(multi_blocks(p.n_pairs)), dim3(64), 0, st, p);
}
void launch_multi_scatter(const MultiParams& p, hipStream_t st) {
  hipLaunchKernelGGL(k_multi_scatter, dim3(multi_blocks(p.n_pairs)), dim3(64), 0, st, p);
}
void launch_multi_finalize(const MultiParams& p, hipStream_t st) {
  hipLaunchKernelGGL(k_multi_finalize, dim3(multi_blocks(p.n_pairs)), dim3(64), 0, st, p);
}

// ------------------------------------------------------------------------------------ debug hooks
// Sample sequence of the device sampler (MT19937 + Lemire + partial Fisher-Yates) for parity tests.
__global__ void k_debug_samples(uint32_t seed, uint32_t k, uint32_t total, uint32_t n_draws, uint32_t* out, uint32_t* idx,
                                uint32_t* tmp7, int mode) {
  __shared__ MtState sm;
  __shared__ WvSampler ws;
  const int lane = threadIdx.x;
  if (lane == 0) {
    mt_seed(&sm, seed);
    sm.calls = 0;
    for (uint32_t i = 0; i < total; ++i) idx[i] = i;
  }
  __syncthreads();
  if (mode == 0) {  // lane 0 alone, uniform_u32
    if (lane == 0) {
      for (uint32_t d = 0; d < n_draws; ++d) {
        for (uint32_t i = 0; i < k; ++i) {
          const uint32_t j = uniform_u32(&sm, i, total - 1);
          const uint32_t a = idx[i];
          idx[i] = idx[j];
          idx[j] = a;
        }
        for (uint32_t i = 0; i < k; ++i) out[d * k + i] = idx[i];
      }
    }
    return;
  }
  // wave sampler (mode 1) / its serial replay path from the first block on (mode 2); two calls back to back so
  // that the hand-over of the generator position between rounds is covered too
  const int n1 = (int)(n_draws / 2), n2 = (int)n_draws - n1;
  for (int part = 0; part < 2; ++part) {
    const int nb = part ? n2 : n1;
    if (k == 1) wv_draw_samples<1>(&sm, &ws, idx, total, nb, tmp7, nullptr, lane, mode == 2);
    if (k == 4) wv_draw_samples<4>(&sm, &ws, idx, total, nb, tmp7, nullptr, lane, mode == 2);
    if (k == 5) wv_draw_samples<5>(&sm, &ws, idx, total, nb, tmp7, nullptr, lane, mode == 2);
    if (k == 7) wv_draw_samples<7>(&sm, &ws, idx, total, nb, tmp7, nullptr, lane, mode == 2);
    __syncthreads();
    for (int e = lane; e < nb * (int)k; e += 64) out[(size_t)(part ? n1 : 0) * k + e] = tmp7[(e / k) * 7 + (e % k)];
    __syncthreads();
  }
}
__global__ void k_debug_image_to_world(const dsm_camera cam, uint32_t n, const double* xy, double* uv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double u, v;
  image_to_world(cam, xy[2 * i], xy[2 * i + 1], &u, &v);
  uv[2 * i] = u;
  uv[2 * i + 1] = v;
}
void launch_debug_image_to_world(const dsm_camera& cam, uint32_t n, const double* xy, double* uv, hipStream_t st) {
  hipLaunchKernelGGL(k_debug_image_to_world, dim3((n + 63) / 64), dim3(64), 0, st, cam, n, xy, uv);
}

void launch_debug_samples(uint32_t seed, uint32_t k, uint32_t total, uint32_t n_draws, uint32_t* out, uint32_t* idx, uint32_t* tmp7,
                          int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_debug_samples, dim3(1), dim3(64), 0, st, seed, k, total, n_draws, out, idx, tmp7, mode);
}
