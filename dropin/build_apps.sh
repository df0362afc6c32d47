#!/bin/bash
# dropin/build_apps.sh -- drop-in demonstration (build container only; needs /root/reference).
#
# Compiles the REFERENCE's own application sources, unmodified and read where they lie, against
#   (a) the reference's own library sources            -> oracle/_ref/apps/<app>     (CPU, the baseline)
#   (b) this repo's host-side API mirror + libkrtrace  -> dropin/_build/<app>        (MI355X path)
# Both output dirs are git-ignored; the binaries travel to the GPU box with the snapshot, the sources do not.
#
# Include resolution: the apps use `#include "../raytracer/imageplane.h"`-style paths, which a compiler
# resolves relative to the including file first.  For (b) the source is therefore piped to the compiler on
# stdin from inside raytrace_cpu_amd/host/raytracer/, so that `../raytracer/*.h`, `raytracer/*.h`, `../include/*.h` and
# `include/*.h` all find this repo's headers: the class API mirror AND the utility headers (kerr, par_file, par_args,
# text_output, fits_output, array, disc, gramschmidt_basis under raytrace_cpu_amd/host/include/).  The MI355X builds
# therefore use nothing of the reference but each program's own main(), and do not link cfitsio (this repo's
# fits_output.h writes the files itself); the reference tree stays on the include path only as a fallback for headers
# this repo does not provide.  Nothing is copied.
#
# src/emissivity/emissivity.cpp does not compile as shipped (`disc_r + dr` is double* + double, line 79);
# the same one-token fix that src/emissivity/emissivity_rd.cpp:88 carries is applied in the pipe for it.
set -euo pipefail
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
HOST=$ROOT/raytrace_cpu_amd/host
CPUOUT=$ROOT/oracle/_ref/apps
GPUOUT=$ROOT/dropin/_build
FLAGS="-O2 -std=c++14 -ffp-contract=off -fopenmp -w"
CFITS_INC=${CFITS_INC:-/opt/conda/include}
CFITS_LIB=${CFITS_LIB:-/opt/conda/lib}
[ -f $REF/src/raytracer/raytracer.cpp ] || { echo "reference tree not found at $REF"; exit 1; }
mkdir -p $CPUOUT $GPUOUT
make -s -C $HOST

REFLIB="$REF/src/raytracer/raytracer.cpp $REF/src/raytracer/pointsource.cpp $REF/src/raytracer/imageplane.cpp"

build() {   # name  source  extra-flags  [sed-expression]
    local name=$1 src=$2 extra=$3 fix=${4:-}
    local feed="cat $src"; [ -n "$fix" ] && feed="sed -e $fix $src"
    # (a) CPU reference
    ( cd $(dirname $src) && $feed | g++ $FLAGS $extra -I$REF/src -I$REF/src/raytracer -x c++ - -x none $REFLIB -o $CPUOUT/$name ${LINK:-} )
    # (b) this repo: headers from $HOST first, reference utilities second
    ( cd $HOST/raytracer && $feed | g++ $FLAGS $extra -I$HOST -I$REF/src -I$(dirname $src) -x c++ - -x none -o $GPUOUT/$name \
        -L$HOST -lkr_host -L$ROOT/raytrace_cpu_amd/csrc -lkrtrace -Wl,-rpath,'$ORIGIN/../../raytrace_cpu_amd/host' -Wl,-rpath,'$ORIGIN/../../raytrace_cpu_amd/csrc' )
    echo "built $name"
}

build emissivity_rd         $REF/src/emissivity/emissivity_rd.cpp ""
build emissivity            $REF/src/emissivity/emissivity.cpp    "" 's/disc_r[[:space:]]*+[[:space:]]*dr/dr/'
build raytrace_rk4_test     $REF/src/tests/raytrace_rk4_test.cpp  ""
build emissivity_rk45_test  $REF/src/tests/emissivity_rk45_test.cpp ""
build integrator_perf_test  $REF/src/tests/integrator_perf_test.cpp ""
build emissivity_rk45_plot  $REF/src/tests/emissivity_rk45_plot.cpp ""
if [ -f $CFITS_INC/fitsio.h ]; then
    # link cfitsio by path: -L$CFITS_LIB would also pull conda's (older) libstdc++ ahead of the system one that
    # libamdhip64 needs.  Run the binaries with LD_PRELOAD=<system libstdc++.so.6> LD_LIBRARY_PATH=$CFITS_LIB.
    for app in imageplane_disc_image imageplane_disc_image_isco imageplane_disc_image_rd; do
        LINK="$CFITS_LIB/libcfitsio.so" build $app $REF/src/imageplane/$app.cpp "-fpermissive -I$CFITS_INC"
    done
    for app in caustic_discplane caustic_sourceplane caustic_plane; do
        LINK="$CFITS_LIB/libcfitsio.so" build $app $REF/src/caustic/$app.cpp "-fpermissive -I$CFITS_INC"
    done
else
    echo "cfitsio not found: imageplane_disc_image skipped"
fi
ls -la $CPUOUT $GPUOUT
