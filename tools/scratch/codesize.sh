#!/bin/bash
# code bytes per kernel of one .hip file: tools/codesize.sh file.hip [flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only "$@" -S "$f" -o /tmp/_cs.s 2>/dev/null
grep -E "codeLenInByte|^_Z[A-Za-z0-9_]*:" /tmp/_cs.s | paste - - | c++filt | sed 's/(anonymous namespace):://g; s/(PlaneArgs)//' | awk '{print $NF, $1, $2, $3, $4, $5}'
