#!/bin/bash
# usage: scripts/disasm.sh lib.so out.s   -- gfx950 ISA of the class library's device code
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/k.fatbin
$L/clang-offload-bundler --unbundle --type=o --input=$T/k.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
$L/llvm-objdump -d --no-show-raw-insn $T/k.co > "$2"
rm -rf $T
