#!/bin/bash
# tools/disasm_kernel.sh OBJ SYMBOL_SUBSTRING : gfx950 disassembly of the kernels of a fat object whose mangled name contains the substring
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1" /dev/null
$B/clang-offload-bundler --type=o --unbundle --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
for s in $($B/llvm-readelf -s $T/dev.co | awk '{print $8}' | grep "$2" | grep -v "\.kd$" | sort -u); do
  $B/llvm-objdump -d $T/dev.co --disassemble-symbols=$s
done
rm -rf $T
