"""CPU-only: the ONE stdout line of bench.py fits the driver's 8 KB tail and carries the contract keys.

Round 4's line was 43 KB (a whole counter collection inlined) and the driver recorded `parsed: null`; the line is now
built by bench.format_line, which drops the prose, rounds the floats and refuses to print anything >= 4 096 bytes."""
import copy
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_result():
    """A full-size result: the long form of a real round-6 run of the driver's command (every key the bench emits, real
    magnitudes), with `from_profiles` as a FRESH pair of counter collections would fill it."""
    out = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_default_long.json")))
    out["from_profiles"] = {
        "k1": {"file": "profiles/r06_k1_pmc.json", "commit": "0123abc", "source_hash": "0123456789ab", "hbm_bytes_per_launch": 91181200000.0,
               "mfma_i8_insts_per_launch": 8809640000.0, "executed_frac_at_this_runs_time": 0.568622},
        "verify": {"file": "profiles/r06_verify_pmc.json", "commit": "0123abc", "source_hash": "0123456789ab", "all_kernels_ms_per_step": 314.859,
                   "executed_fp64_tflops_over_all": 14.8349, "executed_frac_over_all": 0.188636, "scoring_kernels_frac": 0.412345,
                   "top5_ms_execfrac_laneutil": {"k%d" % i: [39.1141, 0.292239, 0.960292] for i in range(5)}},
        "stale": 5, "stale_files": ["profiles/r0%d_k1_pmc.json" % i for i in range(3, 6)]}
    out["roofline"]["traffic"] = 91181200000.0
    out["roofline"]["traffic_file"] = "profiles/r06_k1_pmc.json"
    return out


def test_line_fits_and_has_the_contract_keys(tmp_path):
    import bench
    out = _stub_result()
    dump = tmp_path / "long.json"
    line = bench.format_line(copy.deepcopy(out), str(dump))
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096
    d = json.loads(line)
    for k in ("metric", "value", "ms_per_step", "steps", "dtype", "unit", "n_gpus", "warmup", "scaling", "vs_baseline", "data"):
        assert k in d, k
    assert d["config"]["workload"].startswith("500 images x 4096 feats")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] == pytest.approx(out["roofline"]["achieved"] / out["roofline"]["peak"], rel=1e-5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["value"] == pytest.approx(out["value"], rel=1e-5)
    # the prose stays in the side file
    assert "note" not in d["roofline"] and "note" in json.load(open(dump))["roofline"]
    # figures read from committed files are labelled as such, with the file they came from, its commit and the hash of its sources
    for k in ("k1", "verify"):
        assert d["from_profiles"][k]["file"].startswith("profiles/") and d["from_profiles"][k]["commit"]
        assert json.load(open(dump))["from_profiles"][k]["source_hash"]  # (long form: the hash bench.py compared with this tree's sources)
    assert "top5_ms_execfrac_laneutil" not in d["from_profiles"]["verify"]          # long form only
    assert len(json.load(open(dump))["from_profiles"]["verify"]["top5_ms_execfrac_laneutil"]) == 5
    # the timed graph held against the oracle, in the line, and the side measurements with theirs
    ps = d["parity_sample"]
    assert ps["pairs"] >= 600 and ps["match_mismatches"] == 0 and ps["geometry_mismatches"] == 0 and ps["pose_max_rel"] <= 1e-6
    for k in ("low_inlier_regime", "uncalibrated", "config1", "config3_match_only"):
        assert d["extra"][k]["parity_sample"]["match_mismatches"] == 0, k
    assert d["extra"]["config1"]["parity_sample"]["pairs"] == 1225 and d["extra"]["config1"]["parity_sample"]["geometry_mismatches"] == 0


def test_an_oversize_line_is_refused():
    import bench
    out = _stub_result()
    out["roofline_verify"]["executed"] = {"k%d" % i: list(range(40)) for i in range(60)}
    with pytest.raises(SystemExit) as e:
        bench.format_line(out)
    assert "limit" in str(e.value)


def test_a_missing_contract_key_is_refused():
    import bench
    for k in ("value", "roofline", "config"):
        out = _stub_result()
        out.pop(k)
        with pytest.raises(SystemExit):
            bench.format_line(out)
    out = _stub_result()
    out["roofline"].pop("traffic")
    with pytest.raises(SystemExit):
        bench.format_line(out)


def test_contexts_are_created_before_the_process_group():
    """profiles/r05_lanes_hw_queues.txt: the HIP runtime hands a stream its hardware queue at creation; a context created BEHIND an
    RCCL communicator found the pool exhausted, its two verification lanes shared a queue and every rank of an N > 1 run verified
    20 % slower.  bench.py must create its contexts (dsm_ctx_create creates both lane streams) before init_process_group, and print
    the result line after flushing the C streams (RCCL's banner must not land behind it)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    first_ctx = main.index("capi.Context(dev_index)")
    assert first_ctx < main.index('dist.init_process_group("nccl"'), "contexts first, then the RCCL process group"
    assert main.index("fflush(None)") < main.index("print(result_line"), "C streams flushed before the JSON line"
    capi_src = open(os.path.join(ROOT, "dagsfm_amd", "csrc", "capi.hip")).read()
    create = capi_src[capi_src.index("dsm_ctx_create(int device"):]
    create = create[:create.index("\n}\n")]
    assert "lanes[1].stream" in create and "lanes[0].stream = c->stream" in create, "both lane streams exist when dsm_ctx_create returns"


def test_a_profile_of_other_sources_is_refused(tmp_path):
    """bench.profile_figures only speaks for counter collections whose stamped source hash is that of the kernel sources in this
    tree: a stale file is counted under `stale` and contributes no figure."""
    import shutil
    import bench
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    shutil.copytree(os.path.join(ROOT, "dagsfm_amd", "csrc"), root / "dagsfm_amd" / "csrc",
                    ignore=shutil.ignore_patterns("*.o", "check", "*.inc"))
    good = {"images": 500, "feats": 4096, "pairs": 124750, "k1_traffic_bytes_per_launch": 9.1e10, "commit": "abc1234",
            "SQ_INSTS_VALU_MFMA_I8": {"pass1": {"mean_per_dispatch": 8.0e9}},
            "source_hash": {k: bench.source_hash(str(root), k) for k in bench.PROFILE_SOURCES}}
    json.dump(good, open(root / "profiles" / "r06_k1_pmc.json", "w"))
    fp = bench.profile_figures(str(root), 500, 4096, 124750, True)
    assert fp["k1"]["hbm_bytes_per_launch"] == 9.1e10 and fp["k1"]["commit"] == "abc1234" and "stale" not in fp
    with open(root / "dagsfm_amd" / "csrc" / "match_kernels.hip", "a") as f:
        f.write("// changed\n")
    fp = bench.profile_figures(str(root), 500, 4096, 124750, True)
    assert "k1" not in fp and fp["stale"] == 1 and fp["stale_files"] == ["profiles/r06_k1_pmc.json"]


def test_parity_sample_counts_what_differs():
    """bench.parity_sample on the oracle's own results: zero mismatches against themselves, one per perturbed pair."""
    import numpy as np
    import bench
    from dagsfm_amd import capi, synthetic
    from tests import oracle_lib
    orc = oracle_lib.load()
    scene = synthetic.Scene(4, 512, seed=3)
    ims = [scene.image(i) for i in range(4)]
    cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, True) for _ in range(4)]
    pairs = synthetic.exhaustive_pairs(4)
    opts = capi.default_two_view_options()
    keep = {}
    bench.cpu_baseline(orc, "", "-O3", ims, pairs, 30.0, True, cams, opts, 0, 2, keep=keep, every_pair=True)
    assert sorted(keep) == list(range(len(pairs)))

    def view(mutate=None):
        ms = [np.array(keep[k][0], dtype=np.uint32).reshape(-1, 2).copy() for k in range(len(pairs))]
        tv = [capi.TwoViewGeometry.from_buffer_copy(bytes(keep[k][1])) for k in range(len(pairs))]
        il = [np.array(keep[k][2], dtype=np.uint32).reshape(-1, 2).copy() for k in range(len(pairs))]
        if mutate:
            mutate(ms, tv, il)
        return bench.GraphView(np.array([len(m) for m in ms]), np.concatenate(ms), tv, np.array([len(i) for i in il]), np.concatenate(il))
    ps = bench.parity_sample(keep, view(), True, int(opts.min_num_inliers))
    assert ps == {"pairs": len(pairs), "match_mismatches": 0, "geometry_mismatches": 0, "pose_max_rel": 0.0}

    def flip_match(ms, tv, il):
        ms[1][0, 1] ^= 1

    def flip_bit_of_F(ms, tv, il):
        tv[2].F[4] = np.nextafter(tv[2].F[4], 1e9)

    def drop_inlier(ms, tv, il):
        il[3] = il[3][:-1]
        il[0] = np.concatenate([il[0], il[0][-1:]])
    assert bench.parity_sample(keep, view(flip_match), True, 15)["match_mismatches"] == 1
    r = bench.parity_sample(keep, view(flip_bit_of_F), True, 15)
    assert r["geometry_mismatches"] == 1 and r["first_bad_pair"] == 2 and bench.parity_failed(r)
    assert bench.parity_sample(keep, view(drop_inlier), True, 15)["geometry_mismatches"] == 2
