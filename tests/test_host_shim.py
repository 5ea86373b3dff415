"""Host-side C++ shim (dagsfm_amd/host): Database blob/pair-id semantics on CPU, and the full
SiftFeatureMatcher::Match / ExhaustiveFeatureMatcher::Run drop-in on the GPU against the oracle."""
import ctypes
import os
import sqlite3
import subprocess

import numpy as np
import pytest

from tests import dbutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "dagsfm_amd", "libdagsfm_host.so")
CLI = os.path.join(ROOT, "dagsfm_amd", "dsm_exhaustive_matcher")
u32p = ctypes.POINTER(ctypes.c_uint32)
f64p = ctypes.POINTER(ctypes.c_double)


def host():
    assert os.path.exists(HOST_LIB), "run __graft_entry__.build() first"
    L = ctypes.CDLL(HOST_LIB)
    L.dsm_host_image_pair_to_pair_id.restype = ctypes.c_uint64
    L.dsm_host_image_pair_to_pair_id.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    L.dsm_host_db_write_pair.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, u32p, ctypes.c_uint32, ctypes.c_int,
                                         f64p, f64p, u32p, ctypes.c_uint32]
    L.dsm_host_db_read_pair.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p,
                                        ctypes.POINTER(ctypes.c_int), f64p, f64p, u32p, u32p, ctypes.c_uint32]
    L.dsm_host_exhaustive_matcher.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_int, ctypes.c_int]
    return L


def test_pair_id(tmp_path):
    # Database::ImagePairToPairId, /root/reference/src/base/database.h:336-347
    L = host()
    assert L.dsm_host_image_pair_to_pair_id(1, 2) == 2147483647 * 1 + 2
    assert L.dsm_host_image_pair_to_pair_id(2, 1) == 2147483647 * 1 + 2
    assert L.dsm_host_image_pair_to_pair_id(7, 7) == 2147483647 * 7 + 7
    assert L.dsm_host_image_pair_to_pair_id(100, 3) == dbutil.pair_id(3, 100)


def _write(L, path, a, b, m, config, q, t, inl):
    m = np.ascontiguousarray(m, np.uint32).reshape(-1, 2)
    inl = np.ascontiguousarray(inl, np.uint32).reshape(-1, 2)
    q, t = np.asarray(q, np.float64), np.asarray(t, np.float64)
    rc = L.dsm_host_db_write_pair(path.encode(), a, b, m.ctypes.data_as(u32p), len(m), config, q.ctypes.data_as(f64p),
                                  t.ctypes.data_as(f64p), inl.ctypes.data_as(u32p), len(inl))
    assert rc == 0


def _read(L, path, a, b):
    m, inl = np.zeros((256, 2), np.uint32), np.zeros((256, 2), np.uint32)
    nm, ni, cfg = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_int(0)
    q, t = np.zeros(4), np.zeros(3)
    rc = L.dsm_host_db_read_pair(path.encode(), a, b, m.ctypes.data_as(u32p), ctypes.byref(nm), ctypes.byref(cfg),
                                 q.ctypes.data_as(f64p), t.ctypes.data_as(f64p), inl.ctypes.data_as(u32p), ctypes.byref(ni), 256)
    assert rc == 0
    return m[:nm.value].copy(), cfg.value, q, t, inl[:ni.value].copy()


def test_database_round_trip_and_dagsfm_columns(tmp_path):
    """base/database_test.cc:283-360 patterns + DAGSfM's F:=qvec / E:=tvec / H:=NULL columns (database.cc:733-747)."""
    L = host()
    path = str(tmp_path / "db.db")
    m = [[0, 1], [2, 3], [5, 4]]
    inl = [[0, 1], [5, 4]]
    q = [0.5, 0.5, -0.5, 0.5]
    t = [1.0, 2.0, 3.0]
    _write(L, path, 1, 2, m, 2, q, t, inl)
    rm, cfg, rq, rt, ri = _read(L, path, 1, 2)
    assert rm.tolist() == m and ri.tolist() == inl and cfg == 2 and rq.tolist() == q and rt.tolist() == t
    # reading in swapped order swaps the match columns and inverts the pose (database.cc:527-529, pose.cc:192-196)
    rm2, cfg2, rq2, rt2, ri2 = _read(L, path, 2, 1)
    assert rm2.tolist() == [[1, 0], [3, 2], [4, 5]] and ri2.tolist() == [[1, 0], [4, 5]]
    assert rq2.tolist() == [0.5, -0.5, 0.5, -0.5]
    w, x, y, z = 0.5, -0.5, 0.5, -0.5
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    assert np.allclose(rt2, -R @ np.array(t), atol=1e-12)
    # raw rows
    con = sqlite3.connect(path)
    row = con.execute("SELECT pair_id, rows, cols, data, config, F, E, H FROM two_view_geometries").fetchone()
    assert row[0] == dbutil.pair_id(1, 2) and row[1] == 2 and row[2] == 2 and row[4] == 2
    assert np.frombuffer(row[5], np.float64).tolist() == q and np.frombuffer(row[6], np.float64).tolist() == t
    assert row[7] is None
    mrow = con.execute("SELECT rows, cols, data FROM matches").fetchone()
    assert mrow[0] == 3 and mrow[1] == 2 and np.frombuffer(mrow[2], np.uint32).reshape(3, 2).tolist() == m
    con.close()
    # writing with id1 > id2 stores the swapped matches under the same pair id convention
    _write(L, path, 9, 4, [[7, 8]], 3, q, t, [[7, 8]])
    con = sqlite3.connect(path)
    data = con.execute("SELECT data FROM matches WHERE pair_id = ?", (dbutil.pair_id(4, 9),)).fetchone()[0]
    assert np.frombuffer(data, np.uint32).tolist() == [8, 7]
    con.close()
    # no inliers -> zero-length F/E blobs (database.cc:739-747)
    _write(L, path, 5, 6, [], 0, [0, 0, 0, 0], [0, 0, 0], [])
    con = sqlite3.connect(path)
    row = con.execute("SELECT rows, data, config, F, E FROM two_view_geometries WHERE pair_id = ?", (dbutil.pair_id(5, 6),)).fetchone()
    assert row[0] == 0 and row[2] == 0 and len(row[3] or b"") == 0 and len(row[4] or b"") == 0
    assert con.execute("PRAGMA user_version").fetchone()[0] == 3600
    con.close()


# "sliced": a Match() call cut into slices of a few pairs (match_slice_pairs; 32 768 by default, which no test list reaches), here
# with the asynchronous write-back -- slice k is written while slice k + 1 is on the device; test_bulk_load_journal_... slices a
# blocking run
_SLICED = ["--SiftMatching.async_write_back", "1", "--SiftMatching.match_slice_pairs", "4"]
_BLOCKING = ["--SiftMatching.async_write_back", "0"]  # (the asynchronous write-back is the default since round 5)


@pytest.mark.gpu
@pytest.mark.parametrize("async_write", [False, True, "sliced"])
def test_exhaustive_matcher_drop_in(tmp_path, oracle, async_write, monkeypatch):
    """colmap exhaustive_matcher equivalent over a synthetic database.db == oracle, incl. resume semantics; also with
    the write-back on a background thread (SiftMatchingOptions::async_write_back, DSM_ASYNC_WRITE_BACK)."""
    extra = _SLICED if async_write == "sliced" else (_BLOCKING if async_write is False else [])
    if async_write is True:
        monkeypatch.setenv("DSM_ASYNC_WRITE_BACK", "1")
    from dagsfm_amd import capi, synthetic
    n_img = 6
    scene = synthetic.Scene(n_img, 640, seed=33, n_pool=1800)
    ims = [scene.image(i) for i in range(n_img)]
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
    assert os.path.exists(CLI)
    subprocess.check_call([CLI, "--database_path", path, "--random_seed", "5"] + extra)  # one block: pairs visited as (i, j), i < j
    matches, tvgs = dbutil.read_results(path)
    assert len(matches) == n_img * (n_img - 1) // 2 == len(tvgs)
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True)
    opts = capi.default_two_view_options()
    n_geo = 0
    for i in range(n_img):
        for j in range(i + 1, n_img):
            pid = dbutil.pair_id(i + 1, j + 1)
            ref_m = oracle.match_sift_features_cpu(ims[i][0], ims[j][0])
            ref, ref_inl = oracle.estimate_two_view_geometry(cam, ims[i][1].astype(np.float64), cam, ims[j][1].astype(np.float64),
                                                             ref_m, opts, capi.pair_seed(i + 1, j + 1, 5))
            exp_m = ref_m if len(ref_m) >= 15 else np.zeros((0, 2), np.uint32)
            assert (matches[pid] == exp_m).all()
            t = tvgs[pid]
            if ref.num_inliers >= 15:
                n_geo += 1
                assert t["config"] == ref.config and (t["inliers"] == ref_inl).all()
                assert np.allclose(np.frombuffer(t["F"], np.float64), list(ref.qvec), rtol=1e-6, atol=1e-12)
                assert np.allclose(np.frombuffer(t["E"], np.float64), list(ref.tvec), rtol=1e-6, atol=1e-12)
                assert t["H"] is None
            else:
                assert t["config"] == 0 and len(t["inliers"]) == 0
    assert n_geo >= 8
    # resume: (1) nothing to do; (2) a deleted two_view_geometries row is re-verified from the stored matches
    before = dbutil.read_results(path)
    subprocess.check_call([CLI, "--database_path", path, "--random_seed", "5"] + extra)
    after = dbutil.read_results(path)
    assert all((before[0][k] == after[0][k]).all() for k in before[0])
    con = sqlite3.connect(path)
    pid = dbutil.pair_id(1, 2)
    con.execute("DELETE FROM two_view_geometries WHERE pair_id = ?", (pid,))
    con.commit()
    con.close()
    subprocess.check_call([CLI, "--database_path", path, "--random_seed", "5"] + extra)
    again = dbutil.read_results(path)
    assert (again[1][pid]["inliers"] == before[1][pid]["inliers"]).all() and again[1][pid]["config"] == before[1][pid]["config"]
    assert again[1][pid]["F"] == before[1][pid]["F"]


def _visit_order(n, block_size):
    """ExhaustiveFeatureMatcher::Run's block loop, /root/reference/src/feature/matching.cc:870-905 (0-based indices)."""
    out = []
    for s1 in range(0, n, block_size):
        e1 = min(n, s1 + block_size) - 1
        for s2 in range(0, n, block_size):
            e2 = min(n, s2 + block_size) - 1
            for i1 in range(s1, e1 + 1):
                for i2 in range(s2, e2 + 1):
                    b1, b2 = i1 % block_size, i2 % block_size
                    if (i1 > i2 and b1 <= b2) or (i1 < i2 and b1 < b2):
                        out.append((i1, i2))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("async_write", [False, True, "sliced"])
def test_exhaustive_matcher_blocks_and_swapped_pairs(tmp_path, oracle, async_write, monkeypatch):
    """block_size < #images: some pairs are visited as (larger id, smaller id); the rows are stored swapped /
    inverted exactly as Database::WriteMatches / WriteTwoViewGeometry do (database.cc:681-751)."""
    extra = [_SLICED[0], _SLICED[1], _SLICED[2], "2"] if async_write == "sliced" else (_BLOCKING if async_write is False else [])
    if async_write is True:  # several Match() calls: the write-back of one block overlaps the device work of the next
        monkeypatch.setenv("DSM_ASYNC_WRITE_BACK", "1")
    from dagsfm_amd import capi, synthetic
    n_img = 6
    scene = synthetic.Scene(n_img, 512, seed=34, n_pool=1400)
    ims = [scene.image(i) for i in range(n_img)]
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
    subprocess.check_call([CLI, "--database_path", path, "--ExhaustiveMatching.block_size", "4", "--random_seed", "9"] + extra)
    matches, tvgs = dbutil.read_results(path)
    order = _visit_order(n_img, 4)
    assert len({frozenset(p) for p in order}) == len(order) == n_img * (n_img - 1) // 2 == len(matches)
    assert any(a > b for a, b in order)
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True)
    opts = capi.default_two_view_options()
    n_swapped_geo = 0
    for a, b in order:
        pid = dbutil.pair_id(a + 1, b + 1)
        ref_m = oracle.match_sift_features_cpu(ims[a][0], ims[b][0])
        ref, ref_inl = oracle.estimate_two_view_geometry(cam, ims[a][1].astype(np.float64), cam, ims[b][1].astype(np.float64),
                                                         ref_m, opts, capi.pair_seed(a + 1, b + 1, 9))
        swap = a > b
        exp_m = ref_m[:, ::-1] if swap else ref_m
        assert (matches[pid] == exp_m).all()
        t = tvgs[pid]
        if ref.num_inliers >= 15:
            assert t["config"] == ref.config
            assert (t["inliers"] == (ref_inl[:, ::-1] if swap else ref_inl)).all()
            q = np.frombuffer(t["F"], np.float64)
            tv = np.frombuffer(t["E"], np.float64)
            rq, rt = np.array(list(ref.qvec)), np.array(list(ref.tvec))
            if swap:
                n_swapped_geo += 1
                w, x, y, z = rq[0], -rq[1], -rq[2], -rq[3]
                R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                assert np.allclose(q, [w, x, y, z], rtol=1e-6, atol=1e-12) and np.allclose(tv, -R @ rt, rtol=1e-6, atol=1e-9)
            else:
                assert np.allclose(q, rq, rtol=1e-6, atol=1e-12) and np.allclose(tv, rt, rtol=1e-6, atol=1e-12)
    assert n_swapped_geo >= 1


def test_database_bulk_write_rate(tmp_path):
    """Write-back of SiftFeatureMatcher::Match (Exists x2 + WriteMatches + WriteTwoViewGeometry per pair, one
    transaction, statements prepared once): must stay far above the GPU's ~1e5 pairs/s only in the sense of not being
    orders of magnitude below it -- the rate is printed and recorded in DESIGN.md."""
    L = host()
    L.dsm_host_db_bulk_write_bench.restype = ctypes.c_double
    L.dsm_host_db_bulk_write_bench.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    rate = L.dsm_host_db_bulk_write_bench(str(tmp_path / "bulk.db").encode(), 20000, 256, 160)
    print("db write-back: %.0f pairs/s" % rate)
    assert rate > 500  # a correctness test, not a benchmark: the machine may be busy; ~5e4 pairs/s when idle
    con = sqlite3.connect(str(tmp_path / "bulk.db"))
    assert con.execute("SELECT COUNT(*) FROM matches").fetchone()[0] == 20000
    assert con.execute("SELECT COUNT(*) FROM two_view_geometries WHERE rows = 160").fetchone()[0] == 20000
    con.close()


@pytest.mark.gpu
def test_exhaustive_matcher_guided_matching(tmp_path, oracle):
    """--SiftMatching.guided_matching 1: the two_view_geometries rows hold the guided matches (matching.cc:647-667)."""
    from dagsfm_amd import capi, synthetic
    n_img = 4
    scene = synthetic.Scene(n_img, 512, seed=35, n_pool=1200)
    ims = [scene.image(i) for i in range(n_img)]
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
    subprocess.check_call([CLI, "--database_path", path, "--random_seed", "5", "--SiftMatching.guided_matching", "1"])
    matches, tvgs = dbutil.read_results(path)
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True)
    opts = capi.default_two_view_options()
    n_guided = 0
    for i in range(n_img):
        for j in range(i + 1, n_img):
            pid = dbutil.pair_id(i + 1, j + 1)
            ref_m = oracle.match_sift_features_cpu(ims[i][0], ims[j][0])
            ref, ref_inl = oracle.estimate_two_view_geometry(cam, ims[i][1].astype(np.float64), cam, ims[j][1].astype(np.float64),
                                                             ref_m, opts, capi.pair_seed(i + 1, j + 1, 5))
            exp = ref_inl
            if ref.num_inliers >= 15:
                g = oracle.match_guided_sift_features_cpu(ims[i][1], ims[j][1], ims[i][0], ims[j][0], ref)
                if g is not None:
                    exp, n_guided = g, n_guided + 1
            t = tvgs[pid]
            if len(exp) >= 15:
                assert t["config"] == ref.config and (t["inliers"] == exp).all()
            else:
                assert t["config"] == 0 and len(t["inliers"]) == 0
    assert n_guided >= 3


def test_feature_matcher_cache_is_a_bounded_lru(tmp_path):
    """FeatureMatcherCache keeps at most cache_size images of keypoints / descriptors (matching.h:203-206), except
    that what one Match() call has requested stays pinned until it is done (no whole-database load)."""
    L = host()
    L.dsm_host_cache_lru_probe.argtypes = [ctypes.c_char_p, ctypes.c_uint32, u32p, ctypes.c_uint32, ctypes.c_uint32]
    rng = np.random.default_rng(0)
    ims = [(rng.integers(0, 255, (8, 128)).astype(np.uint8), rng.uniform(0, 100, (8, 2)).astype(np.float32)) for _ in range(20)]
    path = str(tmp_path / "db.db")
    dbutil.create(path, ims)
    ids = np.array(list(range(1, 21)) * 2, dtype=np.uint32)
    assert L.dsm_host_cache_lru_probe(path.encode(), 5, ids.ctypes.data_as(u32p), len(ids), 1) == 5
    assert L.dsm_host_cache_lru_probe(path.encode(), 5, ids.ctypes.data_as(u32p), len(ids), 8) == 8  # a call may pin more
    assert L.dsm_host_cache_lru_probe(path.encode(), 100, ids.ctypes.data_as(u32p), len(ids), 1) == 20


@pytest.mark.gpu
def test_exhaustive_matcher_several_device_contexts(tmp_path):
    """SiftMatchingOptions::gpu_index with a device list: one context (and one host thread) per entry, the pair list
    cut into one contiguous share per context, results merged in list order before the rows are written.  "0,0" puts
    two contexts on the one GPU of the test box; the database must equal the single-context run's row for row.  With
    a block size below the image count the resident image set is also replaced between Match() calls."""
    from dagsfm_amd import synthetic
    n_img = 7
    scene = synthetic.Scene(n_img, 512, seed=35, n_pool=1400)
    ims = [scene.image(i) for i in range(n_img)]
    res = []
    for k, gpu_index in enumerate(["0", "0,0", "-1"]):
        path = str(tmp_path / ("database%d.db" % k))
        dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
        subprocess.check_call([CLI, "--database_path", path, "--ExhaustiveMatching.block_size", "3", "--random_seed", "4",
                               "--SiftMatching.gpu_index", gpu_index])
        res.append(dbutil.read_results(path))
    base_m, base_t = res[0]
    assert len(base_m) == n_img * (n_img - 1) // 2 and sum(len(v) for v in base_m.values()) > 300
    for m, t in res[1:]:
        assert m.keys() == base_m.keys() and t.keys() == base_t.keys()
        for pid in base_m:
            assert (m[pid] == base_m[pid]).all()
            assert t[pid]["config"] == base_t[pid]["config"] and (t[pid]["inliers"] == base_t[pid]["inliers"]).all()
            assert t[pid]["F"] == base_t[pid]["F"] and t[pid]["E"] == base_t[pid]["E"]


@pytest.mark.gpu
@pytest.mark.parametrize("async_write", [False, True, "sliced"])
def test_failed_device_call_leaves_existing_rows_alone(tmp_path, async_write, monkeypatch):
    """Match() used to delete the stale rows of resume-path pairs BEFORE any device work; a device failure
    (reported as an exception, where the reference CHECK-aborts) then committed the deletes and the putative matches
    were gone.  Now the deletes travel with the new rows.  Set-up: a finished database, one two_view_geometries row
    removed (so that pair is on the resume path: its `matches` row would be deleted and rewritten), and a camera
    model id the device rejects -- the run must fail and every row must still be there."""
    extra = [_SLICED[0], _SLICED[1], _SLICED[2], "2"] if async_write == "sliced" else (_BLOCKING if async_write is False else [])
    if async_write is True:
        monkeypatch.setenv("DSM_ASYNC_WRITE_BACK", "1")
    from dagsfm_amd import synthetic
    n_img = 4
    scene = synthetic.Scene(n_img, 512, seed=44, n_pool=1400)
    ims = [scene.image(i) for i in range(n_img)]
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
    subprocess.check_call([CLI, "--database_path", path, "--random_seed", "5"] + extra)
    before = dbutil.read_results(path)
    pid = dbutil.pair_id(1, 2)
    assert len(before[0][pid]) > 15
    con = sqlite3.connect(path)
    con.execute("DELETE FROM two_view_geometries WHERE pair_id = ?", (pid,))
    con.execute("UPDATE cameras SET model = 99")
    con.commit()
    con.close()
    r = subprocess.run([CLI, "--database_path", path, "--random_seed", "5"] + extra, capture_output=True, text=True)
    assert r.returncode != 0 and "camera model" in (r.stderr + r.stdout)
    after = dbutil.read_results(path)
    assert set(after[0]) == set(before[0]) and all((after[0][k] == before[0][k]).all() for k in before[0])
    assert set(after[1]) == set(before[1]) - {pid}
    # and with the camera repaired the pair is verified from its stored matches, as if nothing had happened
    con = sqlite3.connect(path)
    con.execute("UPDATE cameras SET model = 0")
    con.commit()
    con.close()
    subprocess.check_call([CLI, "--database_path", path, "--random_seed", "5"] + extra)
    again = dbutil.read_results(path)
    assert (again[1][pid]["inliers"] == before[1][pid]["inliers"]).all() and again[1][pid]["F"] == before[1][pid]["F"]


def test_malformed_blobs_are_errors_not_overreads(tmp_path):
    """keypoints / descriptors / matches rows whose `rows` x `cols` do not match the blob (the reference CHECK-aborts in
    BlobToMatrix / FeatureMatchesFromBlob, database.cc:60-122, 103-122): the shim reports an error; found reading past the
    blob by an AddressSanitizer run over mutated databases."""
    L = host()
    L.dsm_host_cache_lru_probe.restype = ctypes.c_int
    L.dsm_host_cache_lru_probe.argtypes = [ctypes.c_char_p, ctypes.c_uint32, u32p, ctypes.c_uint32, ctypes.c_uint32]
    rng = np.random.default_rng(0)
    ims = [(rng.integers(0, 256, (n, 128)).astype(np.uint8), rng.uniform(0, 100, (n, 2)).astype(np.float32)) for n in (5, 0, 17)]
    ids = (ctypes.c_uint32 * 3)(1, 2, 3)
    good = str(tmp_path / "good.db")
    dbutil.create(good, ims)
    assert L.dsm_host_cache_lru_probe(good.encode(), 2, ids, 3, 2) >= 0
    # an image without features stored as a 0 x 0 matrix with an empty (or NULL) blob is NOT malformed: the reference's
    # ReadDynamicMatrixBlob accepts it (database.cc:60-77 only CHECKs rows * cols * size == num_bytes); ADVICE r03
    for k, sql in enumerate(["UPDATE descriptors SET rows = 0, cols = 0, data = x'' WHERE image_id = 2",
                             "UPDATE descriptors SET rows = 0, cols = 0, data = NULL WHERE image_id = 2"]):
        path = str(tmp_path / ("empty%d.db" % k))
        dbutil.create(path, ims)
        con = sqlite3.connect(path)
        con.execute(sql)
        con.commit()
        con.close()
        assert L.dsm_host_cache_lru_probe(path.encode(), 2, ids, 3, 2) >= 0, sql
    for k, sql in enumerate(["UPDATE keypoints SET rows = 1000 WHERE image_id = 1", "UPDATE keypoints SET cols = 3 WHERE image_id = 3",
                             "UPDATE keypoints SET rows = -5 WHERE image_id = 1", "UPDATE descriptors SET rows = 999 WHERE image_id = 3",
                             "UPDATE descriptors SET cols = 64 WHERE image_id = 3", "UPDATE descriptors SET data = NULL WHERE image_id = 1",
                             "UPDATE cameras SET params = x'00112233445566778899aabbcc'",   # 13 bytes: not whole doubles
                             "UPDATE cameras SET model = 4"]):                                # OPENCV wants 8 parameters, the row has 3
        path = str(tmp_path / ("bad%d.db" % k))
        dbutil.create(path, ims)
        con = sqlite3.connect(path)
        con.execute(sql)
        con.commit()
        con.close()
        assert L.dsm_host_cache_lru_probe(path.encode(), 2, ids, 3, 2) < 0, sql
    # a matches row that claims 50 rows over 24 bytes
    path = str(tmp_path / "badm.db")
    dbutil.create(path, ims)
    con = sqlite3.connect(path)
    con.execute("INSERT INTO matches VALUES (?, ?, ?, ?)", (dbutil.pair_id(1, 3), 50, 2, b"x" * 24))
    con.commit()
    con.close()
    m, inl = np.zeros((256, 2), np.uint32), np.zeros((256, 2), np.uint32)
    nm, ni, cfg = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_int(0)
    q, t = np.zeros(4), np.zeros(3)
    assert L.dsm_host_db_read_pair(path.encode(), 1, 3, m.ctypes.data_as(u32p), ctypes.byref(nm), ctypes.byref(cfg), q.ctypes.data_as(f64p),
                                   t.ctypes.data_as(f64p), inl.ctypes.data_as(u32p), ctypes.byref(ni), 256) != 0


@pytest.mark.gpu
def test_bulk_load_journal_writes_the_same_rows_and_restores_wal(tmp_path):
    """SiftMatchingOptions::bulk_load_journal (extension, tools/sqlite_ceiling.py): the run appends its rows under an in-memory
    rollback journal instead of the WAL; the rows are the blocking run's, and the file is back in WAL mode afterwards -- the
    mode the reference's Database::Open sets (database.cc:267-276)."""
    from dagsfm_amd import synthetic
    n_img = 6
    scene = synthetic.Scene(n_img, 512, seed=36, n_pool=1400)
    ims = [scene.image(i) for i in range(n_img)]
    res = []
    for k, flags in enumerate([[], ["--SiftMatching.bulk_load_journal", "1"], ["--SiftMatching.bulk_load_journal", "1", "--SiftMatching.async_write_back", "1"],
                               ["--SiftMatching.match_slice_pairs", "2"]]):  # (the last: a blocking run whose Match() calls go to the device in slices)
        path = str(tmp_path / ("database%d.db" % k))
        dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
        r = subprocess.run([CLI, "--database_path", path, "--ExhaustiveMatching.block_size", "3", "--random_seed", "4", "--timing", "1"] + flags,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "SQLite write-back" in r.stderr
        con = sqlite3.connect(path)
        assert con.execute("PRAGMA journal_mode").fetchone()[0] == "wal"
        con.close()
        res.append(dbutil.read_results(path))
    base_m, base_t = res[0]
    assert len(base_m) == n_img * (n_img - 1) // 2
    for m, t in res[1:]:
        assert m.keys() == base_m.keys() and t.keys() == base_t.keys()
        for pid in base_m:
            assert (m[pid] == base_m[pid]).all() and t[pid]["config"] == base_t[pid]["config"] and (t[pid]["inliers"] == base_t[pid]["inliers"]).all()
            assert t[pid]["F"] == base_t[pid]["F"] and t[pid]["E"] == base_t[pid]["E"]


@pytest.mark.gpu
@pytest.mark.parametrize("commit", [1, 0])
def test_match_returns_with_every_row_written_into_the_callers_transaction(tmp_path, commit):
    """Round 5: the asynchronous write-back is the DEFAULT (SiftMatchingOptions::async_write_back).  The contract a reference
    caller relies on must hold with it: ExhaustiveFeatureMatcher::Run wraps Match() in one DatabaseTransaction
    (/root/reference/src/feature/matching.cc:903) and Match() writes before it returns (:819-836).  Here Match() runs over all
    pairs in slices of 4 (slice k's rows on the writer thread while slice k + 1 is on the device) inside the caller's
    transaction: when it returns the connection already sees every row, the transaction is still the caller's, a rollback
    removes them all and a commit keeps them -- identical to the blocking CLI run."""
    from dagsfm_amd import synthetic
    L = host()
    u64 = ctypes.c_uint64
    L.dsm_host_probe_match_in_callers_transaction.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                                              ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_int)]
    n_img = 6
    scene = synthetic.Scene(n_img, 640, seed=33, n_pool=1800)
    ims = [scene.image(i) for i in range(n_img)]
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
    nm, ng, open_ = u64(0), u64(0), ctypes.c_int(0)
    rc = L.dsm_host_probe_match_in_callers_transaction(path.encode(), 4, commit, 5, ctypes.byref(nm), ctypes.byref(ng), ctypes.byref(open_))
    assert rc == 0
    n_pairs = n_img * (n_img - 1) // 2
    assert open_.value == 1, "Match() must leave the caller's transaction open"
    assert nm.value == n_pairs and ng.value == n_pairs, "every row is written when Match() returns"
    matches, tvgs = dbutil.read_results(path)
    if not commit:
        assert len(matches) == 0 and len(tvgs) == 0, "the rows were the caller's transaction's: its rollback removes them"
        return
    assert len(matches) == n_pairs == len(tvgs)
    ref = str(tmp_path / "blocking.db")
    dbutil.create(ref, [(im[0], im[1]) for im in ims], prior=True)
    subprocess.check_call([CLI, "--database_path", ref, "--random_seed", "5"] + _BLOCKING)
    rm, rt = dbutil.read_results(ref)
    assert matches.keys() == rm.keys() and tvgs.keys() == rt.keys()
    for pid in rm:
        assert (matches[pid] == rm[pid]).all()
        a, b = tvgs[pid], rt[pid]
        assert a["config"] == b["config"] and (a["inliers"] == b["inliers"]).all() and a["F"] == b["F"] and a["E"] == b["E"] and a["H"] == b["H"]


@pytest.mark.gpu
def test_shares_assembled_on_the_device_by_rccl_write_the_same_rows(tmp_path):
    """SiftMatchingOptions::assemble_on_device (round 5): the C++ drop-in hands the devices' shares to libdagsfm_gather.so -- one
    communicator over the gpu_index devices (ncclCommInitAll), grouped all-gather of the fixed-size records, broadcasts of the
    match lists -- and fetches the assembled graph from device 0 in one copy per array.  A one-GPU box runs it with a one-rank
    communicator; the rows must be those of the direct fetch, also with several slices and blocks per run.  (Reference analogue:
    one matcher per gpu_index device, outputs collected by the caller, /root/reference/src/feature/matching.cc:631-645, 814-836.)"""
    from dagsfm_amd import synthetic
    n_img = 7
    scene = synthetic.Scene(n_img, 512, seed=37, n_pool=1400)
    ims = [scene.image(i) for i in range(n_img)]
    blocks = ["--SiftMatching.match_slice_pairs", "3", "--ExhaustiveMatching.block_size", "4"]  # several Match() calls, several slices each
    n_compared = 0
    for k, common in enumerate([[], blocks]):  # (a block visit order changes which image of a pair is "first": compare like with like)
        res = []
        for flags in ([], ["--SiftMatching.assemble_on_device", "1"]):
            path = str(tmp_path / ("database%d_%d.db" % (k, len(flags))))
            dbutil.create(path, [(im[0], im[1]) for im in ims], prior=True)
            r = subprocess.run([CLI, "--database_path", path, "--random_seed", "4", "--timing", "1"] + common + flags, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            res.append(dbutil.read_results(path))
        (base_m, base_t), (m, t) = res
        assert len(base_m) == n_img * (n_img - 1) // 2 and sum(len(v) for v in base_m.values()) > 500
        assert m.keys() == base_m.keys() and t.keys() == base_t.keys()
        for pid in base_m:
            assert m[pid].shape == base_m[pid].shape and (m[pid] == base_m[pid]).all()
            a, b = t[pid], base_t[pid]
            assert a["config"] == b["config"] and (a["inliers"] == b["inliers"]).all() and a["F"] == b["F"] and a["E"] == b["E"] and a["H"] == b["H"]
            n_compared += 1
    assert n_compared == 2 * n_img * (n_img - 1) // 2
    # the CLI maps RCCL only when asked to
    needed = subprocess.run(["ldd", CLI], capture_output=True, text=True).stdout
    assert "rccl" not in needed
