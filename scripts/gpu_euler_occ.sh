#!/bin/bash
# GPU box: Euler all-fast 1e7 rays at 2..5 resident waves per SIMD (KR_BLOCKS_PER_CU), and the returning-radiation batch at 3 / 4
cd ${GRAFT_REPO_ROOT:-.}
for b in 2 3 4 5; do echo "== KR_BLOCKS_PER_CU=$b"; KR_BLOCKS_PER_CU=$b timeout -k 10 120 python scripts/ab_kernels.py --integrator euler --rays 1e7 --rounds 3 base@1 2>&1 | cut -c1-200; done
for b in 3 4; do echo "== return radiation, flags blocks_per_cu=$b"; KR_RR_BLOCKS=$b KR_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --workload return_radiation --steps 3 --warmup 1 --no-cpu-baseline --no-fast-math-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['split_launch_ms'], d['roofline']['frac'])"; done
