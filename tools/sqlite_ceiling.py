#!/usr/bin/env python3
"""SQLite's own insert ceiling for the rows of BASELINE configs[1]: 256 matches + 165 inlier matches per
pair (2 048 + 1 320 bytes of blobs, the bench scene's averages) into `matches` + `two_view_geometries`, one transaction,
through python's sqlite3 module (the same libsqlite3 the host shim links) -- no device, no shim.  What bounds the drop-in's
write-back is then visible apart from the shim: the byte volume (the same rows with empty blobs go ~10x faster), the
journal mode (WAL = every page written twice), and the file system under the database.  The schema, the page size (4 096,
fixed when the reference's feature extractor created the file) and the pragmas of the reference's Database::Open
(/root/reference/src/base/database.cc:267-276) are the first line; the others are settings a caller COULD choose.

    python tools/sqlite_ceiling.py [--pairs 40000]"""
import argparse
import os
import shutil
import sqlite3
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import dbutil  # noqa: E402

REFERENCE_PRAGMAS = ["PRAGMA synchronous=OFF", "PRAGMA journal_mode=WAL", "PRAGMA temp_store=MEMORY", "PRAGMA foreign_keys=ON"]


def run(name, n, pragmas, where, n_matches=256, n_inliers=165):
    m = np.arange(2 * n_matches, dtype=np.uint32).tobytes()
    inl = np.arange(2 * n_inliers, dtype=np.uint32).tobytes()
    q, t = b"q" * 32, b"t" * 24
    d = tempfile.mkdtemp(dir=where)
    p = os.path.join(d, "database.db")
    dbutil.create(p, [(np.zeros((1, 128), np.uint8), np.zeros((1, 2), np.float32))])
    con = sqlite3.connect(p, isolation_level=None)
    for s in REFERENCE_PRAGMAS + pragmas:
        con.execute(s)
    t0 = time.time()
    con.execute("BEGIN")
    con.executemany("INSERT INTO matches(pair_id, rows, cols, data) VALUES(?, ?, ?, ?)", ((i + 10, n_matches, 2, m) for i in range(n)))
    con.executemany("INSERT INTO two_view_geometries(pair_id, rows, cols, data, config, F, E, H) VALUES(?, ?, ?, ?, ?, ?, ?, ?)",
                    ((i + 10, n_inliers, 2, inl, 2, q, t, None) for i in range(n)))
    con.execute("COMMIT")
    con.close()
    dt = time.time() - t0
    size = os.path.getsize(p)
    shutil.rmtree(d)
    print("%-64s %8.0f pairs/s  %6.0f MB/s of blobs  (database %4.0f MB)" % (name, n / dt, n * (len(m) + len(inl)) / dt / 1e6, size / 1e6), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=40000)
    a = ap.parse_args()
    print("sqlite %s, page size 4096, %d pairs per transaction" % (sqlite3.sqlite_version, a.pairs))
    tmp = tempfile.gettempdir()
    run("reference pragmas (WAL, synchronous OFF), %s" % tmp, a.pairs, [], tmp)
    run("  the same rows with EMPTY blobs (per-row cost only)", a.pairs, [], tmp, 0, 0)
    run("  + locking_mode EXCLUSIVE", a.pairs, ["PRAGMA locking_mode=EXCLUSIVE"], tmp)
    run("  + journal_mode MEMORY (rollback journal in RAM)", a.pairs, ["PRAGMA journal_mode=MEMORY"], tmp)
    run("  + journal_mode OFF (no rollback: NOT usable, Match() rolls back)", a.pairs, ["PRAGMA journal_mode=OFF"], tmp)
    if os.path.isdir("/dev/shm"):
        run("reference pragmas, database on tmpfs (/dev/shm)", a.pairs, [], "/dev/shm")
        run("  + journal_mode OFF, tmpfs", a.pairs, ["PRAGMA journal_mode=OFF"], "/dev/shm")


if __name__ == "__main__":
    main()
