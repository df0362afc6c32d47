#!/bin/bash
# GPU box, round 2: return-radiation with overlapping launches, the concurrent tolerance sweep, headline + image plane.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02a; mkdir -p $O
summ() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "%.3e rays/s ms %.1f kern %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]), d["roofline"].get("split_launch_ms"), d.get("rk45"),
              {k: round(v["avg_kernel_ms"], 1) for k, v in d.get("other_arithmetic_modes", {}).items()})
PY
}
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_emis.json 2> $O/err1.txt; summ $O/bench_emis.json "emissivity rk4"
for s in 1 2 4 8; do
timeout -k 10 300 python bench.py --workload return_radiation --streams $s --steps 3 --no-cpu-baseline --no-fast-math-extra > $O/bench_rr_s$s.json 2> $O/err_rr$s.txt; summ $O/bench_rr_s$s.json "return_radiation streams=$s"
done
timeout -k 10 300 python bench.py --workload imageplane --steps 3 --no-cpu-baseline --no-fast-math-extra > $O/bench_ip.json 2> $O/err_ip.txt; summ $O/bench_ip.json "imageplane"
timeout -k 10 400 python scripts/rk45_tol_sweep.py strict > $O/rk45_tol_sweep_strict.json 2> $O/err_sweep.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02a/rk45_tol_sweep_strict.json"))
print("sweep strict", [(r["h"], r["tol"], round(r["kernel_ms"])) for r in d["runs"] if r["integrator"] == "rk45"][::4], {k: v for k, v in d["concurrent"].items() if k != "per_point_span_ms"})
PY
timeout -k 10 300 python bench.py --integrator rk45 --steps 3 --no-cpu-baseline > $O/bench_rk45.json 2> $O/err_rk45.txt; summ $O/bench_rk45.json "emissivity rk45"
