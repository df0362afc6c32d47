#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# usage: scripts/profile_round.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r02}; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $ROOT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-fast-math-extra $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { echo trace failed; tail -5 $OUT/trace.err; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || { echo pmc_sq failed; tail -5 $OUT/pmc_sq.err; }
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 bench.py $ARGS > $OUT/bench_pmc_grbm.json 2> $OUT/pmc_grbm.err || { echo pmc_grbm failed; tail -5 $OUT/pmc_grbm.err; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || { echo pmc_fetch failed; tail -5 $OUT/pmc_fetch.err; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || { echo pmc_write failed; tail -5 $OUT/pmc_write.err; }
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_f64 -- python3 bench.py $ARGS > $OUT/bench_pmc_f64.json 2> $OUT/pmc_f64.err || { echo pmc_f64 failed; tail -3 $OUT/pmc_f64.err; }
find $OUT -name "*.csv" | head -50
du -sh $OUT
