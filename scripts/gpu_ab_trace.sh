#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) of bench.py for several builds of the library.
# usage: scripts/gpu_ab_trace.sh "<bench args>" tag [tag ...]     (tag "base" = libkrtrace.so, else libkrtrace_<tag>.so)
set -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
ARGS="$1"; shift
for tag in "$@"; do
  OUT=$ROOT/gpurun_out/ab_$tag
  mkdir -p $OUT
  if [ "$tag" = base ]; then unset KRTRACE_LIB; else export KRTRACE_LIB=$ROOT/raytrace_cpu_amd/csrc/libkrtrace_$tag.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-math-extra $ARGS > $OUT/bench.json 2> $OUT/err.txt || { echo "$tag failed"; tail -5 $OUT/err.txt; exit 1; }
  echo "== $tag"
  python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: b[k] for k in ("value", "ms_per_step", "rk_steps_per_sec")}), "frac", b["roofline"]["frac"], "kernel_ms", b["roofline"]["avg_kernel_ms"])
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if float(row["Percentage"]) > 0.05:
            n = row["Name"]
            n = n[n.find("trace_kernel"):n.find(">") + 1] if "trace_kernel" in n else n[:60]
            print("   %-60s calls %3s avg %10.3f ms" % (n, row["Calls"], float(row["AverageNs"]) / 1e6))
PY
done
