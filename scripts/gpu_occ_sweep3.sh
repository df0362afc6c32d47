#!/bin/bash
run() { python bench.py --workload imageplane --steps 2 --warmup 1 --no-cpu-baseline $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 kernel_ms %.1f steps/s %.3e' % (d['roofline']['avg_kernel_ms'], d['roofline']['kernel_steps_per_sec']))"; }
KR_BLOCKS_PER_CU=3 run "base b3"
KRTRACE_LIB=raytrace_cpu_amd/csrc/libkrtrace_w4.so KR_BLOCKS_PER_CU=4 run "w4 b4"
KRTRACE_LIB=raytrace_cpu_amd/csrc/libkrtrace_w4.so KR_BLOCKS_PER_CU=3 run "w4 b3"
KRTRACE_LIB=raytrace_cpu_amd/csrc/libkrtrace_w5.so KR_BLOCKS_PER_CU=5 run "w5 b5"
KR_BLOCKS_PER_CU=3 run "base b3 fast" --fast-math
KRTRACE_LIB=raytrace_cpu_amd/csrc/libkrtrace_w4.so KR_BLOCKS_PER_CU=4 run "w4 b4 fast" --fast-math
