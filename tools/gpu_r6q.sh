#!/bin/bash
# round 6, session q: the kd-tree word search as a 16-lane group per query (k_flann_search_kd_grp): parity against the reference's own
# FLANN, searches/s, the retrieval row end to end
out=gpurun_out/${1:-r6q}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_retrieval_flann.py tests/test_retrieval.py -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 900 python tools/bench_flann_search.py > $out/flann_search.json 2> $out/flann_search.err; tail -2 $out/flann_search.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6q/flann_search.json").read().strip().splitlines()[-1])
for name, r in d["indices"].items():
    for k in ("k1", "k5"):
        x = r[k]
        print(name, k, "device %.2f M/s (%.0f ms)" % (x["device_searches_per_s_kernel"] / 1e6, x["device_kernel_ms"]), "host all threads %.0f k/s" % (x["host_all_threads_searches_per_s"] / 1e3), "identical", x["ids_and_distances_identical_on_the_host_sample"])
PY
timeout 900 python tools/bench_retrieval.py --flann kdtree > $out/bench_retrieval_kdtree.json 2> $out/bench_retrieval.err; tail -1 $out/bench_retrieval_kdtree.json | cut -c1-600
