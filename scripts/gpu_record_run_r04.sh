#!/bin/bash
# GPU box: round-4 record run of the final tree -- every bench workload (the first two with their cpu_baseline leg: pipeline_bins_check /
# pipeline_planes_check of the TIMED pipeline), the RK45 sweep, the RCCL path on one rank, the app wall times, then scripts/profile_all.sh (kernel
# trace + PMC passes of all five workloads: the counter record of the round).  usage: KR_TREE_COMMIT=<sha> scripts/gpu_record_run_r04.sh [bench|profile <workload ...>]
# (one gpurun call is limited to 20 minutes: "bench" is the first half, "profile rk4 rk45 euler" and "profile imageplane return_radiation" the second)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/record_r04; mkdir -p $O
if [ "$1" = "profile" ]; then shift; scripts/profile_all.sh r04 "$@" > $O/profile_$1.log 2>&1; tail -3 $O/profile_$1.log; exit 0; fi
timeout -k 10 900 python bench.py > $O/bench_n1_emissivity.json 2> $O/err_emis.txt || tail -3 $O/err_emis.txt
timeout -k 10 400 python bench.py --workload imageplane > $O/bench_n1_imageplane.json 2> $O/err_ip.txt || tail -3 $O/err_ip.txt
timeout -k 10 300 python bench.py --workload return_radiation --no-cpu-baseline > $O/bench_n1_return_radiation.json 2> $O/err_rr.txt || tail -3 $O/err_rr.txt
timeout -k 10 300 python bench.py --integrator rk45 --no-cpu-baseline > $O/bench_n1_emissivity_rk45.json 2> $O/err_rk45.txt || tail -3 $O/err_rk45.txt
timeout -k 10 300 python bench.py --integrator euler --no-cpu-baseline > $O/bench_n1_emissivity_euler.json 2> $O/err_eu.txt || tail -3 $O/err_eu.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --no-cpu-baseline --no-fast-math-extra --workload imageplane > $O/bench_torchrun1_imageplane.json 2> $O/err_tr.txt || tail -5 $O/err_tr.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/record_r04/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "%.3e rays/s %.3e steps/s ms %.1f kern %.1f frac %.3f" % (d["value"], d["rk_steps_per_sec"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]), d["scaling"], d["roofline"].get("split_launch_ms"), d.get("rk45"))
            if "cpu_baseline" in d:
                c = d["cpu_baseline"]; print("   cpu_baseline", {k: c[k] for k in ("value", "cores", "kind", "wall_s") if k in c}, c.get("sample", "")[:80])
                for k in ("pipeline_bins_check", "pipeline_planes_check", "bins_check", "planes_check", "rays_check"):
                    if k in c: print("     ", k, {a: b for a, b in c[k].items() if a != "what"})
PY
timeout -k 10 600 scripts/app_wall.sh > /dev/null 2>&1; cp gpurun_out/app_wall.txt $O/app_wall.txt; grep -E "^==|^wall" $O/app_wall.txt
timeout -k 10 600 python scripts/rk45_tol_sweep.py strict > $O/rk45_tol_sweep_strict.json 2> $O/err_sweep.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/record_r04/rk45_tol_sweep_strict.json"))
print("sweep strict", [(r["h"], r["tol"], round(r["kernel_ms"])) for r in d["runs"] if r["integrator"] == "rk45"][::4], {k: v for k, v in d["concurrent"].items() if k != "per_point_span_ms"})
PY
if [ "$1" != "bench" ]; then scripts/profile_all.sh r04 > $O/profile.log 2>&1; tail -3 $O/profile.log; fi
