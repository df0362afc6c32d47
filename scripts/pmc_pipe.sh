#!/bin/bash
# GPU box: instruction mix of the in-kernel pipeline's trace kernels against the record kernels' (two --pmc passes each).
# usage: scripts/pmc_pipe.sh  -> gpurun_out/pmc_pipe/<passes|kernel>_<a|b>.txt
set -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pipe
mkdir -p $OUT; cd $ROOT
for pl in passes kernel; do
  ARGS="--pipeline $pl --steps 1 --warmup 1 --no-cpu-baseline --no-fast-math-extra"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/${pl}_a -- python3 bench.py $ARGS > /dev/null 2> $OUT/${pl}_a.err || { tail -3 $OUT/${pl}_a.err; exit 1; }
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $OUT/${pl}_b -- python3 bench.py $ARGS > /dev/null 2> $OUT/${pl}_b.err || { tail -3 $OUT/${pl}_b.err; exit 1; }
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for pl in ("passes", "kernel"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for ab in "ab":
        for f in glob.glob(f"{out}/{pl}_{ab}/*/*_counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                n = row["Kernel_Name"]
                if "trace_" in n and "kernel" in n:
                    per[n[n.index("trace_"):n.index(">") + 1]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in per.items():
        print(pl, k, {a: "%.3g" % (sum(v) / len(v)) for a, v in sorted(d.items())})
PY
