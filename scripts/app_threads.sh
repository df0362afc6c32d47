#!/bin/bash
# GPU box: the reference's emissivity main() on the class API under different OMP_NUM_THREADS (KR_TIMING marks).  usage: scripts/app_threads.sh [threads ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAR=$ROOT/raytrace_cpu_amd/apps/par
DROPIN=$ROOT/dropin/_build
export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
export KR_TIMING=1
for nt in "${@:-default}"; do
  W=$(mktemp -d); mkdir -p $W/par $W/run; cp $PAR/emissivity_c2.par $W/par/emissivity.par
  if [ "$nt" = default ]; then unset OMP_NUM_THREADS; else export OMP_NUM_THREADS=$nt; fi
  echo "== OMP_NUM_THREADS=$nt"
  t0=$(date +%s%N)
  ( cd $W/run && $DROPIN/emissivity --outfile=$W/out.dat 2>&1 | grep -E "kr_timing: t" | grep -E "ctor|redshift_start: begin|run_raytrace|redshift: end|dtor" | cut -c1-110 )
  t1=$(date +%s%N); echo "wall $(( (t1 - t0) / 1000000 )) ms"
  rm -rf $W
done
