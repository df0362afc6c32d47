#!/bin/bash
for wl in imageplane; do for b in 1 2 3; do
  echo -n "$wl blocks_per_cu=$b  "
  KR_BLOCKS_PER_CU=$b python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms %.1f steps/s %.3e' % (d['roofline']['avg_kernel_ms'], d['roofline']['kernel_steps_per_sec']))"
done; done
for b in 1 2; do echo -n "emissivity rk45 blocks_per_cu=$b  "; KR_BLOCKS_PER_CU=$b python bench.py --integrator rk45 --rays 1e6 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms %.1f steps/s %.3e' % (d['roofline']['avg_kernel_ms'], d['roofline']['kernel_steps_per_sec']), d['rk45'])"; done
