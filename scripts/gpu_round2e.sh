#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02e; mkdir -p $O
summ() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "%.3e rays/s %.3e steps/s ms %.1f kern %.1f frac %.3f" % (d["value"], d["rk_steps_per_sec"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]), d["roofline"].get("split_launch_ms"))
PY
}
B="--steps 3 --no-cpu-baseline --no-fast-math-extra"
for tag in base opt2; do
  if [ $tag = base ]; then unset KRTRACE_LIB; else export KRTRACE_LIB=$PWD/raytrace_cpu_amd/csrc/libkrtrace_$tag.so; fi
  timeout -k 10 300 python bench.py $B > $O/emis_$tag.json 2> $O/err.txt || tail -3 $O/err.txt; summ $O/emis_$tag.json "$tag emissivity rk4"
  timeout -k 10 300 python bench.py $B --workload return_radiation --streams 8 > $O/rr8_$tag.json 2> $O/err.txt || tail -3 $O/err.txt; summ $O/rr8_$tag.json "$tag return_radiation streams=8"
  timeout -k 10 300 python bench.py $B --workload imageplane > $O/ip_$tag.json 2> $O/err.txt || tail -3 $O/err.txt; summ $O/ip_$tag.json "$tag imageplane"
done
export KRTRACE_LIB=$PWD/raytrace_cpu_amd/csrc/libkrtrace_opt2.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -15 $O/gputest.log
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
