#!/bin/bash
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 kernel_ms %.1f steps/s %.3e ms_per_step %.1f' % (d['roofline']['avg_kernel_ms'], d['roofline']['kernel_steps_per_sec'], d['ms_per_step']))"; }
for b in 1 2 3; do KR_BLOCKS_PER_CU=$b run "strict b$b" "--arithmetic strict"; done
for b in 1 2 3; do KR_BLOCKS_PER_CU=$b run "fast b$b" --arithmetic fast; done
for b in 2 3; do KR_BLOCKS_PER_CU=$b run "strict 3e7 b$b" "--rays 3e7"; done
