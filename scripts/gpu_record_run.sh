#!/bin/bash
# GPU box: the round's record run -- full GPU suite, every bench workload, rocprofv3 kernel trace + PMC passes of the headline.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/record; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -6 $O/gputest.log
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
timeout -k 10 600 python bench.py > $O/bench_n1_emissivity.json 2> $O/err_emis.txt || tail -3 $O/err_emis.txt
timeout -k 10 300 python bench.py --workload imageplane > $O/bench_n1_imageplane.json 2> $O/err_ip.txt || tail -3 $O/err_ip.txt
timeout -k 10 300 python bench.py --workload return_radiation --no-cpu-baseline > $O/bench_n1_return_radiation.json 2> $O/err_rr.txt || tail -3 $O/err_rr.txt
timeout -k 10 300 python bench.py --integrator rk45 --no-cpu-baseline > $O/bench_n1_emissivity_rk45.json 2> $O/err_rk45.txt || tail -3 $O/err_rk45.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --no-cpu-baseline --no-fast-math-extra --workload imageplane > $O/bench_torchrun1_imageplane.json 2> $O/err_tr.txt || tail -5 $O/err_tr.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 3 --no-cpu-baseline --no-fast-math-extra --scaling strong > $O/bench_torchrun1_emissivity_strong.json 2> $O/err_tr2.txt || tail -5 $O/err_tr2.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/record/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "%.3e rays/s %.3e steps/s ms %.1f kern %.1f frac %.3f" % (d["value"], d["rk_steps_per_sec"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]), d["scaling"], d["roofline"].get("split_launch_ms"), d.get("rk45"))
PY
timeout -k 10 600 python scripts/rk45_tol_sweep.py strict > $O/rk45_tol_sweep_strict.json 2> $O/err_sweep.txt
timeout -k 10 600 python scripts/rk45_tol_sweep.py hybrid > $O/rk45_tol_sweep_hybrid.json 2>> $O/err_sweep.txt
python - <<'PY'
import json
for m in ("strict", "hybrid"):
    d = json.load(open("gpurun_out/record/rk45_tol_sweep_%s.json" % m))
    print("sweep", m, [(r["h"], r["tol"], round(r["kernel_ms"])) for r in d["runs"] if r["integrator"] == "rk45"][::4], {k: v for k, v in d["concurrent"].items() if k != "per_point_span_ms"})
PY
KR_TIMING=1 scripts/app_wall.sh > /dev/null 2>&1; cp gpurun_out/app_wall.txt $O/app_wall.txt; grep -E "^==|wall" $O/app_wall.txt
scripts/profile_round.sh r02 > $O/profile.log 2>&1; tail -3 $O/profile.log
