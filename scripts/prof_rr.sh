#!/bin/bash
# GPU box: per-kernel time of the returning-radiation pass (rocprofv3 --kernel-trace --stats).  -> gpurun_out/prof_rr/
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_rr; mkdir -p $OUT; cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload return_radiation --steps 3 --warmup 1 --no-cpu-baseline --no-fast-math-extra > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"]; i = n.find("kr::")
    print(r["Calls"], "%.3f ms avg  %.1f ms total" % (float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6), n[i:i + 100] if i >= 0 else n[:60])
PY
head -c 400 $OUT/bench.json
