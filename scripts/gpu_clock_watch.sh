#!/bin/bash
# GPU box: samples the GPU clock / power (rocm-smi, read-only) every 0.2 s while a command runs.  usage: gpu_clock_watch.sh <out> -- cmd...
OUT=$1; shift; shift
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')"; sleep 0.2; done ) > $OUT &
W=$!
"$@"
kill $W 2>/dev/null
