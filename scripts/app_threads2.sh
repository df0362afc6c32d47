#!/bin/bash
# GPU box: the reference's emissivity main() on the class API under different KRTRACE_HOST_THREADS (team size of the mirror's own loops).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 KR_TIMING=1
for nt in "$@"; do
  W=$(mktemp -d); mkdir -p $W/par $W/run; cp $ROOT/raytrace_cpu_amd/apps/par/emissivity_c2.par $W/par/emissivity.par
  export KRTRACE_HOST_THREADS=$nt
  echo "== KRTRACE_HOST_THREADS=$nt"
  t0=$(date +%s%N)
  ( cd $W/run && $ROOT/dropin/_build/emissivity --outfile=$W/out.dat 2>&1 | grep -E "kr_timing" | grep -E "first touched|redshift_start|staged|run_raytrace: end|dtor: end" | cut -c1-140 | head -8 )
  t1=$(date +%s%N); echo "wall $(( (t1 - t0) / 1000000 )) ms"
  rm -rf $W
done
