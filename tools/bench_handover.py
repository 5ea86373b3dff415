#!/usr/bin/env python
"""Hand-over of host buffers through dsm_set_images: pageable numpy arrays (one allocation per image, like the reference's
FeatureDescriptors) against pinned ones; first call (allocations included) and repeated calls.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--scene", action="store_true", help="the bench's own synthetic images (dagsfm_amd.synthetic.Scene) instead of random bytes")
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    desc = [rng.integers(0, 256, size=(args.feats, 128), dtype=np.uint8) for _ in range(args.images)]
    kps = [rng.random((args.feats, 6), dtype=np.float32) * 1000 for _ in range(args.images)]
    if args.scene:
        from dagsfm_amd import synthetic
        scene = synthetic.Scene(args.images, args.feats, seed=0, outlier_frac=0.2)
        ims = [scene.image(i) for i in range(args.images)]
        desc = [im[0] for im in ims]
        kps = [im[1] for im in ims]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True) for _ in range(args.images)]
    nbytes = sum(d.nbytes + k.nbytes for d, k in zip(desc, kps))
    out = {"images": args.images, "feats": args.feats, "bytes": nbytes, "desc": [str(desc[0].dtype), list(desc[0].shape), bool(desc[0].flags.c_contiguous)],
           "kp": [str(kps[0].dtype), list(kps[0].shape), bool(kps[0].flags.c_contiguous)]}
    ctx = capi.Context(0)
    times = []
    for _ in range(args.repeat + 1):
        t = time.perf_counter()
        ctx.set_images(desc, kps, cams)
        times.append(1e3 * (time.perf_counter() - t))
    out["pageable_ms_first"] = times[0]
    out["pageable_ms_repeat"] = min(times[1:])
    out["pageable_gb_per_s_repeat"] = nbytes / min(times[1:]) / 1e6
    # the same bytes in pinned memory (what the shim's FeatureMatcherCache slabs hand over)
    pd = [torch.from_numpy(d).pin_memory() for d in desc]
    pk = [torch.from_numpy(k).pin_memory() for k in kps]
    ctx2 = capi.Context(0)
    times = []
    for _ in range(args.repeat + 1):
        t = time.perf_counter()
        ctx2.set_images([x.numpy() for x in pd], [x.numpy() for x in pk], cams)
        times.append(1e3 * (time.perf_counter() - t))
    out["pinned_ms_first"] = times[0]
    out["pinned_ms_repeat"] = min(times[1:])
    # python-side share of a call: the same argument marshalling with no images behind it is not separable here; an upper bound
    t = time.perf_counter()
    [np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, 128) for d in desc]
    [np.ascontiguousarray(k, dtype=np.float32) for k in kps]
    out["python_marshalling_ms"] = 1e3 * (time.perf_counter() - t)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
