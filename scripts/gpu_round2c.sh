#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02c; mkdir -p $O
python scripts/ab_kernels.py --rays 1e7 --rounds 3 nok@2 base@2 kfast@2 > $O/ab_k.txt 2>&1; cat $O/ab_k.txt | cut -c1-400
scripts/gpu_ab_trace.sh "" base kfast nok > $O/ab_trace.log 2>&1; cat $O/ab_trace.log
scripts/app_wall.sh > /dev/null 2>&1; cp gpurun_out/app_wall.txt $O/; cat $O/app_wall.txt | head -40
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -15 $O/gputest.log
