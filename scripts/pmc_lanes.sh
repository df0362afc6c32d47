#!/bin/bash
# GPU box: lane occupancy and VALU busy time of the trace kernels of one bench.py configuration.
# usage: scripts/pmc_lanes.sh <tag> [bench args...]   -> gpurun_out/pmc_lanes_<tag>.json
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lanes_$TAG
mkdir -p $OUT; cd $ROOT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-fast-math-extra $@"
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err || { tail -3 $OUT/sq.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -3 $OUT/trace.err; exit 1; }
python3 - "$OUT" "$TAG" "$ARGS" <<'PY'
import collections, csv, glob, json, sys
out, tag, args = sys.argv[1:4]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(glob.glob(f"{out}/sq/*/*_counter_collection.csv")[0])):
    n = row["Kernel_Name"]
    if "trace_kernel" in n:
        per[n[n.index("trace_kernel"):n.index(">") + 1]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = {}
for row in csv.DictReader(open(glob.glob(f"{out}/trace/*/*_kernel_stats.csv")[0])):
    n = row["Name"]
    if "trace_kernel" in n:
        dur[n[n.index("trace_kernel"):n.index(">") + 1]] = float(row["AverageNs"]) / 1e6
res = {"tag": tag, "command": f"rocprofv3 --pmc ... / --kernel-trace --stats -- python3 bench.py {args}", "kernels": {}}
for k, d in per.items():
    c = {a: sum(v) / len(v) for a, v in d.items()}
    ms = dur.get(k)
    res["kernels"][k] = {"avg_ms_unprofiled_pass": ms, "valu_wave_instructions": c["SQ_INSTS_VALU"],
                         "lane_occupancy_pct(SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU))": 100 * c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
                         "valu_busy_ms_per_simd(SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz)": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.4e6}
b = json.loads([l for l in open(f"{out}/bench_trace.json") if l.startswith("{")][-1])
res["bench"] = {k: b[k] for k in ("value", "ms_per_step", "rk_steps_per_sec")}
res["bench"]["rk45"] = b.get("rk45")
json.dump(res, open(f"{out}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
