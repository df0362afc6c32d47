#!/bin/bash
# GPU box: lane occupancy of the trace kernels' step loop, from the diagnostic build (libkrtrace_occ.so: -DKR_OCC_STATS=1,
# `python scripts/ab_kernels.py --build occ:-DKR_OCC_STATS=1` in the build container).  usage: scripts/gpu_occ_stats.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export KRTRACE_LIB=$ROOT/raytrace_cpu_amd/csrc/libkrtrace_occ.so
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-math-extra "$@" 2>&1 | grep -v "^kr_occ" | tail -1 | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print({k:b[k] for k in ('value','ms_per_step')}, b['config'])"
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fast-math-extra "$@" 2>&1 | grep "^kr_occ" | sort | uniq -c
