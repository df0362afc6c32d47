#!/bin/bash
# Regenerates tests/golden/apps/*.{dat,fits,txt}: outputs of the REFERENCE's own applications (CPU build,
# oracle/_ref/apps/, produced by dropin/build_apps.sh from the sources under /root/reference) on the par
# files next to this script.  Build container only.  Fixtures are the apps' output files (data), nothing else.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
APPS=$HERE/../../oracle/_ref/apps
G=$HERE/apps
export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 LD_LIBRARY_PATH=/opt/conda/lib
# emissivity*.cpp keep a pointer into a std::string that has gone out of scope when --parfile is given
# (emissivity.cpp:22-27), so they are run from a scratch dir where the built-in default ../par/<app>.par resolves.
W=$(mktemp -d); mkdir -p $W/par $W/run
cp $G/emissivity.par $G/emissivity_rd.par $W/par/
( cd $W/run && $APPS/emissivity --outfile=$G/emissivity.dat > /dev/null && $APPS/emissivity_rd --outfile=$G/emissivity_rd.dat > /dev/null )
rm -rf $W
# imageplane_disc_image_rd.cpp has the same dangling --parfile pointer (:42-50): default path from a scratch dir
W=$(mktemp -d); mkdir -p $W/par $W/run
cp $G/imageplane_rd.par $W/par/imageplane_disc_image_rd.par
rm -f $G/imageplane_rd.fits $G/imageplane_isco.fits
( cd $W/run && $APPS/imageplane_disc_image_rd --outfile=$G/imageplane_rd.fits > /dev/null )
rm -rf $W
$APPS/imageplane_disc_image_isco --parfile=$G/imageplane_isco.par --outfile=$G/imageplane_isco.fits > /dev/null
rm -f $G/imageplane_rk4.fits $G/imageplane_rk45.fits
$APPS/imageplane_disc_image --parfile=$G/imageplane_rk4.par  --outfile=$G/imageplane_rk4.fits  > /dev/null
$APPS/imageplane_disc_image --parfile=$G/imageplane_rk45.par --outfile=$G/imageplane_rk45.fits > /dev/null
for c in caustic_discplane caustic_discplane_rk45 caustic_sourceplane caustic_plane; do
    rm -f $G/$c.fits
    app=${c%_rk45}
    $APPS/$app --parfile=$G/$c.par --outfile=$G/$c.fits > /dev/null
done
# BASELINE configs[2]: one point (tol = 1e-8) of the RK45 tolerance sweep, src/tests/emissivity_rk45_plot.cpp
$APPS/emissivity_rk45_plot $G/emissivity_rk45_plot.csv 1e-8 > /dev/null
# ... and the two ends of that sweep (src/tests/emissivity_rk45_tol_sweep.py:38)
$APPS/emissivity_rk45_plot $G/emissivity_rk45_plot_tol1e-6.csv 1e-6 > /dev/null
$APPS/emissivity_rk45_plot $G/emissivity_rk45_plot_tol1e-10.csv 1e-10 > /dev/null
# src/tests/integrator_perf_test.cpp: the step statistics of its report (timing lines left out)
$APPS/integrator_perf_test | grep -E "^(Total rays|Valid rays|Invalid rays|Steps per ray|Total steps|Total func)" > $G/integrator_perf_test.txt
$APPS/raytrace_rk4_test    | tail -16 > $G/raytrace_rk4_test.txt
$APPS/emissivity_rk45_test | tail -40 > $G/emissivity_rk45_test.txt
ls -la $G
