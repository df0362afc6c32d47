#!/bin/bash
# GPU box: does the hybrid launch still overlap its two kernels once torch.distributed / RCCL is initialised in the process?
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; env "$@" KR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-math-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  rays/s %.3e  ms/step %.1f  trace ms %.1f' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms']))"; }
run KR_SIDE_STREAM_PRIORITY=low
run KR_SIDE_STREAM_PRIORITY=default
run KR_SIDE_STREAM_PRIORITY=high
run KR_SIDE_STREAM_PRIORITY=default GPU_MAX_HW_QUEUES=8
echo "== no dist, default prio"; KR_SIDE_STREAM_PRIORITY=default timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-math-extra 2>/dev/null | cut -c1-120
echo "== no dist, low prio"; KR_SIDE_STREAM_PRIORITY=low timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-math-extra 2>/dev/null | cut -c1-120
