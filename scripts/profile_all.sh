#!/bin/bash
# GPU box (via gpurun): the round's counter record, ONE script for every bench workload, so that no committed counter file can describe an
# older kernel than the one beside it.  Per workload: a kernel-trace pass (per-kernel durations) and separate --pmc passes (SQ instruction
# mix / busy / lane occupancy, GRBM, FETCH_SIZE, WRITE_SIZE -- never combined with a trace domain), condensed by scripts/profile_all_summary.py
# into gpurun_out/prof_<tag>/<workload>_counters.json; copy those (and trace_kernel_hbm_traffic.json) into profiles/.
# usage: KR_TREE_COMMIT=$(git rev-parse --short HEAD) scripts/profile_all.sh <tag> [workload ...]     workloads: rk4 rk45 euler imageplane return_radiation (default: all)
# (the GPU box has no .git: the commit the tree was built from is handed in)
set -o pipefail
TAG=${1:-r04}; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
WL=${@:-rk4 rk45 euler imageplane return_radiation}
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --no-fast-math-extra"
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU"
G3="SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"
for w in $WL; do
    case $w in
        rk4) ARGS="";;
        rk45) ARGS="--integrator rk45";;
        euler) ARGS="--integrator euler";;
        imageplane) ARGS="--workload imageplane";;
        return_radiation) ARGS="--workload return_radiation";;
        *) echo "unknown workload $w"; exit 2;;
    esac
    OUT=$ROOT/gpurun_out/prof_$TAG/$w
    rm -rf $OUT; mkdir -p $OUT
    echo "== $w: python3 bench.py $COMMON $ARGS"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $COMMON $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { echo "trace pass failed"; tail -5 $OUT/trace.err; exit 1; }
    i=0
    for G in "$G1" "$G2" "$G3" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_$i -- python3 bench.py $COMMON $ARGS > $OUT/bench_pmc_$i.json 2> $OUT/pmc_$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc_$i.err; }
    done
    python3 scripts/profile_all_summary.py $TAG $w "python3 bench.py $COMMON $ARGS" || exit 1
    # the raw per-dispatch CSVs are large; the summary and the kernel-stats table are what is kept
    find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
done
ls -la $ROOT/gpurun_out/prof_$TAG/
