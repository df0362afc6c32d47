#!/bin/bash
# Disassembles the gfx950 code of the three objects of libkrtrace.so into <outdir>/<object>.s (addresses and encodings stripped),
# so that two builds can be compared kernel by kernel:   scripts/isa_dump.sh /tmp/isa_a ; <edit> ; build ; scripts/isa_dump.sh /tmp/isa_b ;
# diff -r /tmp/isa_a /tmp/isa_b.   An edit that only removes dead preprocessor branches must leave every file identical.
# With a second argument (a kernel-name substring) it also prints that kernel's instruction mix.
set -e
OUT=${1:-/tmp/isa}
CSRC="$(dirname "$0")/../raytrace_cpu_amd/csrc"
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT"
for f in kr_trace kr_post kr_capi; do
    objcopy -O binary --only-section=.hip_fatbin "$CSRC/$f.o" "$OUT/$f.fat"
    $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$OUT/$f.fat" --output="$OUT/$f.hsaco" --unbundle
    $LLVM/llvm-objdump -d --no-show-raw-insn --no-leading-addr "$OUT/$f.hsaco" | sed -E 's/[[:space:]]*\/\/ [0-9A-Fa-f]+:.*$//' > "$OUT/$f.s"
    rm -f "$OUT/$f.fat"
done
wc -l "$OUT"/*.s
