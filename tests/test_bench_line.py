"""CPU-only: the ONE stdout line of bench.py fits the driver's 8 KB tail and carries the contract keys.

Round 4's line was 43 KB (a whole counter collection inlined) and the driver recorded `parsed: null`; the line is now
built by bench.format_line, which drops the prose, rounds the floats and refuses to print anything >= 4 096 bytes."""
import copy
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_result():
    """A full-size result: round 4's own kept line (every key the bench emits, real magnitudes) without the inlined dump,
    plus what round 5 added."""
    import bench
    line = open(os.path.join(ROOT, "profiles", "r04_bench_default.json")).read().strip().splitlines()[-1]
    out = json.loads(line)
    out["roofline_verify"].pop("executed", None)
    out["kernel_ms_per_step"]["pass-2 k1_resolve_index + compaction"] = 9.123456789
    out["kernel_ms_per_step"]["exchange"] = 0.3816789
    out["from_profiles"] = bench.profile_figures(ROOT, 500, 4096, 124750, True)
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
    # figures read from committed files are labelled as such, with the file they came from
    assert d["from_profiles"]["k1"]["file"].startswith("profiles/")
    assert len(d["from_profiles"]["verify"]["top5_ms_execfrac_laneutil"]) == 5


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
