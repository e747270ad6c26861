#!/bin/bash
# per-kernel register / scratch summary of one .hip file: tools/kres.sh guidedquant_amd/csrc/ap_plane.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep "remark:" | sed 's/.*remark: *//; s/ *\[-Rpass.*//' \
 | awk -F': ' '/^Function Name/{if(n)print n, v, a, s, sc, sp, oc; n=$2} /^VGPRs:/{v="v="$2} /^AGPRs/{a="a="$2} /^TotalSGPRs/{s="s="$2} /^ScratchSize/{sc="scratch="$2} /^VGPRs Spill/{sp="spill="$2} /^Occupancy/{oc="occ="$2} END{print n, v, a, s, sc, sp, oc}' | c++filt | sed 's/(anonymous namespace):://g; s/(PlaneArgs)//'
