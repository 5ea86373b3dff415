#!/usr/bin/env python3
"""At-scale consistency check of the verification schedules (GPU).  Three independently written device schedules must
produce byte-identical TwoViewGeometry records and inlier matches on the whole workload:
  batched   phase-split pipeline, local optimisation as batched kernels (k_replay_lo + k_lo_*; what bench.py runs)
  inline    phase-split pipeline, local optimisation inline in the wave-per-pair replay (DSM_VERIFY_INLINE_LO=1)
  no_tail   the batched schedule without a tail: every local optimisation through the batched kernels (DSM_LO_TAIL=0)
  tail_inline  the tail of a round as round 2 ran it: inline in k_replay_lo<1>, F and H only (DSM_LO_TAIL_MODE=inline;
            the default since round 3 computes the tail's local optimisations as parallel items, k_tail_enum / k_tail_lo)
  final_1wave   k_verify_final compiled for one wave per SIMD (no register spill; DSM_FINAL_WAVES=1)
  no_prefilter  F / H scoring by the plain k_score instead of bound + exact (DSM_SCORE_PREFILTER=0; round 4)
  e_fused       the essential family's scoring by the wave-per-hypothesis kernel with the bound step fused in (k_models_score_e,
                DSM_SCORE_PREFILTER=3) instead of a lane per model (k_prescore_compact + k_score_needed; with it F takes the slot-per-lane k_prescore)
  one_lane  the batched schedule on a single lane (DSM_VERIFY_LANES=1; the default deals the list out to two lanes)
  legacy    one k_ransac kernel per family, lane-0 sampler, per-lane scratch solvers (DSM_VERIFY_LEGACY=1; --legacy)
The schedules marked * exist in the CHECK build only (libdagsfm_mi355x_check.so, csrc/ctx.h): `batched` and the scheduling knobs
run on the PRODUCT library, the * schedules on a second context of the check library over the same matches, and every
one is compared with the product's `batched` records -- so the comparison also shows that the two builds agree.
  * no_prefilter, e_fused, final_1wave, legacy, h_mfma (DSM_SCORE_PREFILTER=9: the H bound step's products on the FP64 matrix pipe),
    h_f64 (=17: the pure FP64 H bound step of round 4; the product's has a packed-f32 first stage), ef_f64 (=33: the pure FP64 E / F
    bound step k_prescore_compact; the product's k_prescore_compact2 has a packed-f32 first stage since round 6), replay_legacy (DSM_REPLAY_LEGACY:
    the replay scans of rounds 2 - 5, k_replay_lo<fam, 0 / 2>, points and residuals through global memory; the product's k_replay_rp
    keeps the pair in LDS), elu_lds (DSM_ELU_LDS: the 5-point solver's 10 x 10 elimination in lane-interleaved LDS, rounds 2 - 5; the
    product's k_solve_e_lu_reg keeps the matrix in registers), hyp_pair_grid (DSM_HYP_GRID=pair: the lane-per-hypothesis solvers of E / F on the
    (pair, 64 trials) grid in every round; the product's take the hypotheses of later rounds 64 per wave across the pairs), lo_prepare_wave
    (DSM_LO_PREPARE_WAVE: every local optimisation's design matrix + pivoted QR by the general kernel k_lo_prepare, matrix in memory; the
    product's k_lo_prepare_reg keeps it in registers for E / F up to 384 inliers and H up to 64)
This exercises the paths too rare for the oracle-sized tests (a Lemire rejection in the sampler happens for a few
dozen pairs of config 2; pairs with > 15 local optimisations in one round).

    python tools/check_schedules.py [--images 500] [--feats 4096] [--legacy]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic  # noqa: E402


def run(ctx, opts, schedule):
    os.environ.pop("DSM_VERIFY_LEGACY", None)
    os.environ.pop("DSM_VERIFY_LANES", None)
    os.environ.pop("DSM_LO_TAIL", None)
    os.environ.pop("DSM_LO_TAIL_MODE", None)
    if schedule == "tail_inline":
        os.environ["DSM_LO_TAIL_MODE"] = "inline"
    os.environ.pop("DSM_SCORE_PREFILTER", None)
    if schedule == "no_prefilter":
        os.environ["DSM_SCORE_PREFILTER"] = "0"
    if schedule == "e_fused":
        os.environ["DSM_SCORE_PREFILTER"] = "3"
    if schedule == "h_mfma":
        os.environ["DSM_SCORE_PREFILTER"] = "9"
    if schedule == "h_f64":
        os.environ["DSM_SCORE_PREFILTER"] = "17"
    if schedule == "ef_f64":
        os.environ["DSM_SCORE_PREFILTER"] = "33"
    os.environ.pop("DSM_FINAL_WAVES", None)
    if schedule == "final_1wave":
        os.environ["DSM_FINAL_WAVES"] = "1"
    if schedule == "no_tail":
        os.environ["DSM_LO_TAIL"] = "0"
    if schedule == "one_lane":
        os.environ["DSM_VERIFY_LANES"] = "1"
    os.environ["DSM_VERIFY_INLINE_LO"] = "1" if schedule == "inline" else "0"
    if schedule == "legacy":
        os.environ["DSM_VERIFY_LEGACY"] = "1"
    os.environ.pop("DSM_REPLAY_LEGACY", None)
    os.environ.pop("DSM_ELU_LDS", None)
    if schedule == "elu_lds":
        os.environ["DSM_ELU_LDS"] = "1"
    os.environ.pop("DSM_LO_PREPARE_WAVE", None)
    if schedule == "lo_prepare_wave":
        os.environ["DSM_LO_PREPARE_WAVE"] = "1"
    os.environ.pop("DSM_HYP_GRID", None)
    if schedule == "hyp_pair_grid":
        os.environ["DSM_HYP_GRID"] = "pair"
    if schedule == "replay_legacy":
        os.environ["DSM_REPLAY_LEGACY"] = "1"
    ctx.verify_pairs(opts, user_seed=0, stage_filter=True)
    recs = np.zeros((ctx.n_pairs, ctypes.sizeof(capi.TwoViewGeometry)), dtype=np.uint8)
    rc = ctx._L.dsm_get_two_view_geometries(ctx._h, recs.ctypes.data)
    assert rc == 0
    ioffs, inl = ctx.inlier_matches()
    return recs, np.array(ioffs).copy(), np.array(inl).copy(), ctx.verify_kernel_time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--legacy", action="store_true", help="also run the (slow) single-kernel-per-family schedule")
    ap.add_argument("--outlier-frac", type=float, default=0.2, help="0.5: the 0.25-inlier-ratio regime (thousands of trials per pair)")
    ap.add_argument("--planar", action="store_true", help="a planar scene: H is the model, its local optimisations have hundreds of inliers")
    ap.add_argument("--uncalibrated", action="store_true", help="cameras without a focal-length prior: the F + H path of the decision tree")
    a = ap.parse_args()
    scene = synthetic.Scene(a.images, a.feats, seed=0, outlier_frac=a.outlier_frac, planar=a.planar)
    ims = [scene.image(i) for i in range(a.images)]
    pairs = synthetic.exhaustive_pairs(a.images)
    for k in capi.CHECK_OPTION_KEYS:
        os.environ.pop(k, None)
    cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, 0 if a.uncalibrated else 1) for _ in range(a.images)]
    ctxs = {}
    for check in (False, True):  # the same images and the same matches in both builds
        c = capi.Context(0, check=check)
        c.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
        c.match_pairs(pairs)
        ctxs[check] = c
    mo = [np.array(x).copy() for x in ctxs[False].matches()], [np.array(x).copy() for x in ctxs[True].matches()]
    assert all((u == v).all() for u, v in zip(*mo)), "the two builds must produce the same matches"
    opts = capi.default_two_view_options()
    r0 = run(ctxs[False], opts, "batched")
    ok = True
    CHECK_ONLY = ("no_prefilter", "e_fused", "final_1wave", "legacy", "h_mfma", "h_f64", "ef_f64", "replay_legacy", "elu_lds", "hyp_pair_grid", "lo_prepare_wave")
    for name in ["batched_check_build", "replay_legacy", "elu_lds", "hyp_pair_grid", "lo_prepare_wave", "no_prefilter", "e_fused", "h_mfma", "h_f64", "ef_f64", "one_lane", "no_tail", "tail_inline", "final_1wave", "inline"] + (["legacy"] if a.legacy else []):
        use_check = name in CHECK_ONLY or name == "batched_check_build"
        r1 = run(ctxs[use_check], opts, "batched" if name == "batched_check_build" else name)
        for k in capi.CHECK_OPTION_KEYS:  # the product context must not see a check-only switch
            os.environ.pop(k, None)
        same = (r0[0] == r1[0]).all() and (r0[1] == r1[1]).all() and (r0[2] == r1[2]).all()
        # num_trials / num_models are the last 32 bytes of the record
        print("pairs %d  inlier matches %d  batched (product) %.0f ms  %s (%s build) %.0f ms  identical: %s" % (
            len(pairs), len(r0[2]), r0[3], name, "check" if use_check else "product", r1[3], same))
        if not same:
            bad = np.nonzero((r0[0] != r1[0]).any(axis=1))[0]
            print("first differing pairs:", bad[:10])
            ok = False
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
