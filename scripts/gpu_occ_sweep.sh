#!/bin/bash
# GPU box: trace-kernel time vs resident workgroups per CU (KR_BLOCKS_PER_CU), emissivity RK4
for rays in 3e6 1e7; do for b in 1 2 3; do
  echo -n "rays=$rays blocks_per_cu=$b  "
  KR_BLOCKS_PER_CU=$b python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rays $rays 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms %.1f steps/s %.3e' % (d['roofline']['avg_kernel_ms'], d['roofline']['kernel_steps_per_sec']))"
done; done
