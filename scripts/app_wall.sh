#!/bin/bash
# End-to-end wall time of the applications at BASELINE sizes (GPU box): device-resident programs vs the reference's own
# main() on the host-array class API.  usage: scripts/app_wall.sh  (writes gpurun_out/app_wall.txt)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
PAR=$ROOT/raytrace_cpu_amd/apps/par
NATIVE=$ROOT/raytrace_cpu_amd/apps/_build
DROPIN=$ROOT/dropin/_build
export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
W=$(mktemp -d)
timed() { local t0=$(date +%s%N); "$@" 2>&1 | grep -E "timing|rror" | cut -c1-200 ; local t1=$(date +%s%N); echo "wall $(( (t1 - t0) / 1000000 )) ms"; }
{
for rep in 1 2; do
  echo "== kr_emissivity (device-resident), emissivity_c2.par, run $rep"
  timed $NATIVE/kr_emissivity --parfile=$PAR/emissivity_c2.par --outfile=$W/c2_native.dat --timing
done
echo "== reference emissivity main() on the class API (host ray array, H2D + D2H), same grid, RK45 as hard-coded there"
( export KR_TIMING=1; mkdir -p $W/par $W/run && cp $PAR/emissivity_c2.par $W/par/emissivity.par && cd $W/run && timed $DROPIN/emissivity --outfile=$W/c2_dropin.dat )
for rep in 1 2; do
  echo "== kr_imageplane_disc_image (device-resident), imageplane_c4.par, run $rep"
  timed $NATIVE/kr_imageplane_disc_image --parfile=$PAR/imageplane_c4.par --outfile=$W/c4_native.fits --timing
done
echo "== reference imageplane_disc_image main() on the class API, same par"
( export KR_TIMING=1; timed $DROPIN/imageplane_disc_image --parfile=$PAR/imageplane_c4.par --outfile=$W/c4_dropin.fits )
ls -la $W
cmp $W/c4_native.fits $W/c4_dropin.fits && echo "c4: native and drop-in FITS files are byte-identical" || echo "c4: files differ (sum order of atomics)"
} > $OUT/app_wall.txt 2>&1
rm -rf $W
cat $OUT/app_wall.txt
