#!/bin/bash
# builds libgq_hip.so variants with parts of the QTIP band engine removed (QT_ABL bits, csrc/qtip.hip) -- run on the build host:
#   tools/qtip_ablation.sh build     -> guidedquant_amd/abl/libgq_abl<N>.so
# and times them on the GPU box:  tools/qtip_ablation.sh run
cd "$(dirname "$0")/.."
C=guidedquant_amd/csrc; O=guidedquant_amd/abl${OSUF}; mkdir -p $O
FL="-fno-slp-vectorize -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function"
VARS="${VARS:-1 2 4 8 16 31 5 13}"
if [ "$1" = build ]; then
  for v in $VARS; do
    ( /opt/rocm/bin/hipcc $FL $XDEF -DQT_ABL=$v -c $C/qtip.hip -o $O/qtip_abl$v.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/qtip_abl$v.o $(ls $C/*.o | grep -v qtip.o) -fopenmp -L/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib -o $O/libgq_abl$v.so ) &
  done; wait; ls -la $O/*.so
else
  echo "== full"; MV_ONLY=$MV_ONLY python tools/bench_qtip_mv.py 2>/dev/null | grep "qtip_matvec\|linear"
  for v in $VARS; do echo "== QT_ABL=$v"; GQ_LIB_PATH=$PWD/$O/libgq_abl$v.so MV_ONLY=$MV_ONLY python tools/bench_qtip_mv.py 2>/dev/null | grep "qtip_matvec\|linear"; done
fi
