#!/bin/bash
# GPU box: the RK45 measurements of a round (image plane, tolerance sweeps, 1e7-ray bench, lane occupancy).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
summ() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "%.3e rays/s ms %.1f kern %.1f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"]), d.get("rk45"),
              {k: round(v["avg_kernel_ms"], 1) for k, v in d.get("other_arithmetic_modes", {}).items()})
PY
}
timeout -k 10 300 python bench.py --workload imageplane --integrator rk45 --no-cpu-baseline > $O/bench_ip_rk45.json 2>/dev/null; summ $O/bench_ip_rk45.json "imageplane rk45"
timeout -k 10 300 python scripts/rk45_tol_sweep.py hybrid > $O/rk45_tol_sweep_hybrid.json 2> $O/rk45_sweep.err
timeout -k 10 300 python scripts/rk45_tol_sweep.py fast > $O/rk45_tol_sweep_fast.json 2>> $O/rk45_sweep.err
python - <<'PY'
import json
for f in ("hybrid", "fast"):
    d = json.load(open("gpurun_out/rk45_tol_sweep_%s.json" % f))
    print(f, [(r["h"], r["tol"], round(r["kernel_ms"])) for r in d["runs"] if r["integrator"] == "rk45"][::4])
PY
timeout -k 10 300 python bench.py --integrator rk45 --no-cpu-baseline > $O/bench_r01h_rk45.json 2>/dev/null; summ $O/bench_r01h_rk45.json "emissivity rk45"
timeout -k 10 400 bash scripts/pmc_lanes.sh rk45 --integrator rk45 > $O/pmc_lanes_rk45.log 2>&1; tail -22 $O/pmc_lanes_rk45.log
