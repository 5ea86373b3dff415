"""dsm_ctx_set_memory_budget (round 6): the host application says how much transient chunk scratch the stages may hold; smaller
chunks cost time, never results.  The reference documents its matcher's footprint and leaves the rest of the GPU to the
application (/root/reference/doc/faq.rst:353-356)."""
import ctypes

import numpy as np
import pytest

from dagsfm_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _run(ctx, pairs, opts):
    ctx.match_pairs(pairs)
    ctx.verify_pairs(opts, user_seed=5, stage_filter=True)
    recs = np.zeros((ctx.n_pairs, ctypes.sizeof(capi.TwoViewGeometry)), dtype=np.uint8)
    assert ctx._L.dsm_get_two_view_geometries(ctx._h, recs.ctypes.data) == 0
    offs, m = ctx.matches()
    ioffs, inl = ctx.inlier_matches()
    return recs, np.array(offs).copy(), np.array(m).copy(), np.array(ioffs).copy(), np.array(inl).copy()


def test_a_budget_changes_the_chunks_not_the_records():
    """A config-2-shaped list (4 096 features, ~256 matches per pair) under budgets from generous to tiny: the scratch the context
    holds stays inside the budget, several chunks are really taken, and matches, TwoViewGeometry records and inlier matches are
    byte-identical to the unbudgeted run."""
    n_img = 40
    scene = synthetic.Scene(n_img, 4096, seed=0)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, True) for _ in range(n_img)]
    pairs = synthetic.exhaustive_pairs(n_img)  # 780 pairs
    opts = capi.default_two_view_options()
    ctx = capi.Context(0)
    ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    ref = _run(ctx, pairs, opts)
    res0, scr0 = ctx.memory_footprint()
    assert res0 > 0 and scr0 > 0
    assert sum(1 for r in ref[0] if r[0] > 1) > 700  # (config byte: most pairs have a geometry)
    for budget in (2 << 30, 512 << 20, 128 << 20):
        ctx.set_memory_budget(budget)
        _, held = ctx.memory_footprint()
        assert held <= budget                       # scratch of the larger run before was given back at once
        got = _run(ctx, pairs, opts)
        res, scr = ctx.memory_footprint()
        assert scr <= budget, (budget, scr)
        for a, b in zip(got, ref):
            assert a.shape == b.shape and (a == b).all(), budget
    assert scr < scr0 / 2                           # the 128 MiB run really ran in several chunks
    ctx.set_memory_budget(0)                        # back to the defaults
    got = _run(ctx, pairs, opts)
    for a, b in zip(got, ref):
        assert (a == b).all()


def test_budget_is_per_context_and_argument_checked():
    a, b = capi.Context(0), capi.Context(0)
    a.set_memory_budget(1 << 30)
    assert a.memory_footprint()[1] <= 1 << 30 and b.memory_footprint() == (b.memory_footprint()[0], b.memory_footprint()[1])
    L = capi.lib()
    L.dsm_ctx_set_memory_budget.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    assert L.dsm_ctx_set_memory_budget(None, 1) != 0
    assert L.dsm_ctx_memory_footprint(None, None, None) != 0
