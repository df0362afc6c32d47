#!/bin/bash
# prints VGPR / SGPR / scratch / occupancy of every trace kernel instance (hipcc cross-compile, no GPU needed)
cd "$(dirname "$0")/../raytrace_cpu_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics "$@" -c kr_trace.hip -o /tmp/kr_trace_res.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - \
 | sed -E 's/Function Name: _ZN2kr12_GLOBAL__N_112trace_kernelI([fd])Li([0-9])ELb([01])ELi([0-9]+)E[^\t]*/\1 method=\2 dest=\3 refill=\4/'
