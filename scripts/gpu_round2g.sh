#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02g; mkdir -p $O
summ() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "%.3e rays/s ms %.1f kern %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]), d["roofline"].get("split_launch_ms"))
PY
}
B="--steps 3 --no-cpu-baseline --no-fast-math-extra"
for tag in inl base; do
  if [ $tag = base ]; then unset KRTRACE_LIB; else export KRTRACE_LIB=$PWD/raytrace_cpu_amd/csrc/libkrtrace_$tag.so; fi
  timeout -k 10 300 python bench.py $B > $O/emis_$tag.json 2> $O/err.txt || tail -3 $O/err.txt; summ $O/emis_$tag.json "$tag emissivity rk4"
  timeout -k 10 300 python bench.py $B --integrator rk45 > $O/rk45_$tag.json 2> $O/err.txt || tail -3 $O/err.txt; summ $O/rk45_$tag.json "$tag emissivity rk45"
  timeout -k 10 300 python bench.py $B --workload imageplane > $O/ip_$tag.json 2> $O/err.txt || tail -3 $O/err.txt; summ $O/ip_$tag.json "$tag imageplane"
  timeout -k 10 300 python bench.py $B --workload return_radiation --streams 8 > $O/rr_$tag.json 2> $O/err.txt || tail -3 $O/err.txt; summ $O/rr_$tag.json "$tag return_radiation s8"
  timeout -k 10 300 python scripts/rk45_tol_sweep.py hybrid > $O/sweep_$tag.json 2> $O/err.txt || tail -3 $O/err.txt
  python - <<PY
import json
d = json.load(open("$O/sweep_$tag.json"))["concurrent"]
print("$tag sweep hybrid", {k: (round(v) if isinstance(v, float) and v > 10 else v) for k, v in d.items() if k in ("wall_ms", "slowest_single_point_ms", "wall_over_slowest_point")})
PY
done
