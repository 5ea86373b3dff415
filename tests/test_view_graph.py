"""View-graph ingest + rotation-cycle filter: DistributedMapperController::LoadTwoviewGeometries +
ViewGraph::FilterViewGraphCyclesByRotation (src/graph/view_graph.cpp:115-165).

CPU: the oracle (oracle/view_graph.cc) on graphs whose answer is known by construction -- consistent rotations keep
every edge that lies on a triangle, an edge with a corrupted rotation poisons exactly the triangles through it, an edge
on no triangle is dropped.  GPU: identical keep flags and triplet counts for seeded random graphs, including the
verified geometries of a real pair list."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib


def _oracle_filter(pairs, qvecs, thr=5.0):
    L = oracle_lib.load().lib
    L.oracle_view_graph_filter_cycles.restype = ctypes.c_uint64
    L.oracle_view_graph_filter_cycles.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                                  ctypes.c_void_p]
    p = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    q = np.ascontiguousarray(qvecs, np.float64).reshape(-1, 4)
    keep = np.zeros(max(len(p), 1), np.uint8)
    err = np.zeros(max(len(p), 1), np.float64)
    n = L.oracle_view_graph_filter_cycles(len(p), p.ctypes.data, q.ctypes.data, thr, keep.ctypes.data, err.ctypes.data)
    return keep[:len(p)].astype(bool), int(n), err[:len(p)]


def _rand_rot(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def _qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _qconj(a):
    return np.array([a[0], -a[1], -a[2], -a[3]])


def _graph(rng, n_img, pairs, corrupt=(), noise=0.0):
    """Relative rotations q_ij = q_j * conj(q_i) of random absolute orientations (loop error 0), optionally a few
    edges replaced by random rotations, optionally small noise on all."""
    absq = _rand_rot(rng, n_img)
    q = np.array([_qmul(absq[j], _qconj(absq[i])) for i, j in pairs])
    if noise:
        q = q + rng.normal(scale=noise, size=q.shape)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    for e in corrupt:
        q[e] = _rand_rot(rng, 1)[0]
    return q


def test_oracle_known_answers():
    rng = np.random.default_rng(3)
    # a 5-clique plus a pendant edge (5, 6) that lies on no triangle
    pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)] + [(5, 6)]
    q = _graph(rng, 7, pairs)
    keep, nt, err = _oracle_filter(pairs, q)
    assert nt == 10 and keep[:10].all() and not keep[10] and err[:10].max() < 1e-6
    # corrupt edge (0, 1): the three triangles through it fail, yet its end points keep their other edges
    q2 = _graph(np.random.default_rng(3), 7, pairs, corrupt=[0])
    keep2, nt2, err2 = _oracle_filter(pairs, q2)
    assert nt2 == 10 and not keep2[0] and keep2[1:10].all() and err2[0] > 5.0
    # a single triangle with 3 degrees of loop error passes at 5, fails at 2
    tri = [(10, 20), (10, 30), (20, 30)]
    ang = np.deg2rad(3.0)
    qt = np.array([[1, 0, 0, 0], [1, 0, 0, 0], [np.cos(ang / 2), np.sin(ang / 2), 0, 0]], dtype=np.float64)
    k5, n5, e5 = _oracle_filter(tri, qt, 5.0)
    k2, _, _ = _oracle_filter(tri, qt, 2.0)
    assert n5 == 1 and k5.all() and not k2.any() and abs(e5[0] - 3.0) < 1e-9
    # duplicates and swapped order: the first occurrence counts (ViewGraph::AddTwoViewGeometry)
    k, n, _ = _oracle_filter(tri + [(20, 10)], np.vstack([qt, _rand_rot(rng, 1)]), 5.0)
    assert n == 1 and list(k) == [True, True, True, False]


@pytest.mark.gpu
@pytest.mark.parametrize("n_img,deg,n_corrupt,noise", [(40, 8, 10, 0.0), (300, 30, 400, 0.01), (1000, 12, 0, 0.03), (5, 4, 1, 0.0)])
def test_device_filter_equals_oracle(dsm, n_img, deg, n_corrupt, noise):
    rng = np.random.default_rng(n_img)
    pairs = set()
    for i in range(n_img):
        for j in rng.choice(n_img, min(deg, n_img - 1), replace=False):
            if i != j:
                pairs.add((min(i, int(j)), max(i, int(j))))
    pairs = sorted(pairs)
    order = rng.permutation(len(pairs))
    pairs = [pairs[k] for k in order]  # arbitrary list order, arbitrary ids
    ids = rng.permutation(10 * n_img)[:n_img] + 1
    q = _graph(rng, n_img, pairs, corrupt=rng.choice(len(pairs), min(n_corrupt, len(pairs)), replace=False), noise=noise)
    pid = [(int(ids[a]), int(ids[b])) for a, b in pairs]
    # the stored rotation is for image_id1 < image_id2: conjugate where the renumbering flipped the order
    qs = np.array([qq if x < y else _qconj(qq) for (x, y), qq in zip(pid, q)])
    pid = [(min(x, y), max(x, y)) for x, y in pid]
    ref_keep, ref_n, ref_err = _oracle_filter(pid, qs)
    keep, n = dsm.view_graph_filter_cycles(pid, qs, 5.0)
    assert n == ref_n
    # a decision can only differ where a loop error sits within rounding of the threshold: none does here
    assert np.abs(ref_err[np.isfinite(ref_err)] - 5.0).min() > 1e-6
    assert (keep == ref_keep).all()
    assert 0 < keep.sum() < len(keep) or n_corrupt == 0


@pytest.mark.gpu
def test_filter_over_verified_pairs(dsm, oracle):
    """The geometries the stage itself produces (qvec of dsm_verify_pairs) through the filter, device == oracle."""
    from dagsfm_amd import capi, synthetic
    n_img = 9
    scene = synthetic.Scene(n_img, 640, seed=4, n_pool=1800)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True) for _ in range(n_img)]
    dsm.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    pairs = synthetic.exhaustive_pairs(n_img)
    dsm.match_pairs(pairs)
    dsm.verify_pairs(capi.default_two_view_options(), user_seed=2, stage_filter=True)
    tv = dsm.two_view_geometries()
    sel = [k for k in range(len(pairs)) if tv[k].config in (2, 3, 4, 5, 6)]
    p = pairs[sel]
    q = np.array([list(tv[k].qvec) for k in sel])
    ref_keep, ref_n, ref_err = _oracle_filter(p, q)
    keep, n = dsm.view_graph_filter_cycles(p, q, 5.0)
    assert n == ref_n and (keep == ref_keep).all() and keep.sum() >= 10
