#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI (dsm_exhaustive_matcher = colmap exhaustive_matcher on the MI355X path) over a
synthetic database.db, SQLite included: blocking write-back vs SiftMatchingOptions::async_write_back
(DSM_ASYNC_WRITE_BACK=1).    python tools/bench_cli.py [--images 400] [--feats 1024]"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import synthetic  # noqa: E402
from tests import dbutil  # noqa: E402

CLI = os.path.join(ROOT, "dagsfm_amd", "dsm_exhaustive_matcher")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=400)
    ap.add_argument("--feats", type=int, default=1024)
    ap.add_argument("--block_size", type=int, default=50, help="ExhaustiveMatching.block_size (reference default 50)")
    ap.add_argument("--modes", default="blocking,async,blocking+bulk_journal,async+bulk_journal,async+bulk_journal unsliced,async+bulk_journal serial_setup",
                    help="comma-separated; `default` = no option at all; otherwise words: async, bulk_journal, unsliced (match_slice_pairs 0), "
                         "serial_setup (overlap_setup 0), on_device (assemble_on_device: RCCL gather library)")
    a = ap.parse_args()
    scene = synthetic.Scene(a.images, a.feats, seed=1)
    ims = [scene.image(i) for i in range(a.images)]
    d = tempfile.mkdtemp()
    base = os.path.join(d, "base.db")
    dbutil.create(base, [(im[0], im[1]) for im in ims], prior=True)
    n_pairs = a.images * (a.images - 1) // 2
    for mode in a.modes.split(","):
        path = os.path.join(d, mode.replace("+", "_").replace(" ", "_") + ".db")
        shutil.copy(base, path)
        env = dict(os.environ)
        env.pop("DSM_ASYNC_WRITE_BACK", None)
        t0 = time.perf_counter()
        flags = [CLI, "--database_path", path, "--random_seed", "1", "--ExhaustiveMatching.block_size", str(a.block_size), "--timing", "1"]
        if mode.strip() != "default":  # "default": nothing but the path -- what a user who changes no option gets
            flags += ["--SiftMatching.async_write_back", "1" if "async" in mode else "0",
                      "--SiftMatching.bulk_load_journal", "1" if "bulk" in mode else "0",
                      # (async: a Match() over more than 1.5 x 32 768 pairs runs in slices, slice k written while k + 1 computes)
                      "--SiftMatching.match_slice_pairs", "0" if "unsliced" in mode else "-1",
                      "--ExhaustiveMatching.overlap_setup", "0" if "serial_setup" in mode else "1",
                      "--SiftMatching.assemble_on_device", "1" if "on_device" in mode else "0"]
        subprocess.check_call(flags, env=env, stderr=sys.stdout)
        dt = time.perf_counter() - t0
        print("block_size %d  %-35s %d pairs in %.2f s  (%.0f pairs/s incl. process start, image upload and SQLite)" % (a.block_size, mode, n_pairs, dt, n_pairs / dt), flush=True)
    shutil.rmtree(d)


if __name__ == "__main__":
    main()
