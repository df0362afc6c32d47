#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02b; mkdir -p $O
summ() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "%.3e rays/s ms %.1f kern %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]), d["roofline"].get("split_launch_ms"))
PY
}
scripts/microbench/host_xfer > $O/host_xfer.txt 2>&1; cat $O/host_xfer.txt
for s in 1 4 8; do
timeout -k 10 300 python bench.py --workload return_radiation --streams $s --steps 3 --no-cpu-baseline --no-fast-math-extra > $O/bench_rr_s$s.json 2> $O/err_rr$s.txt || tail -3 $O/err_rr$s.txt; summ $O/bench_rr_s$s.json "return_radiation streams=$s"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -15 $O/gputest.log
