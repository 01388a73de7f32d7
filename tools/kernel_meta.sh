#!/bin/bash
# kernel_meta.sh [FILE.hip] [extra flags] -- registers, spills and scratch of every kernel of one source file as the repo's flags
# compile it for gfx950 (the metadata block of the device assembly).
F=${1:-nvorbis_amd/csrc/kernels_synth.hip}
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DNVH_SRC_HASH='"x"' --cuda-device-only -S "$F" -o $T/k.s $2 2>/dev/null || exit 1
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|name|vgpr_spill_count):" $T/k.s | paste - - - - - | sed 's/\s\+/ /g'
rm -rf $T
