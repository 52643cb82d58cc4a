#!/bin/bash
# Register / scratch / LDS usage of every kernel of a .hip source, from the gfx950 code object's notes.
# Usage: bash scripts/kernel_resources.sh neptune_amd/csrc/qp_kernels.hip [extra hipcc flags]
SRC=$1; shift
OUT=/tmp/$(basename "$SRC" .hip).gfx950.co
FP=""; case "$SRC" in *geom_kernels*) FP="-ffp-contract=off";; esac
QF=""; case "$SRC" in *geom_kernels*) QF="-mllvm -disable-machine-licm";; *qp_reg_kernel*) QF="-mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $FP $QF "$@" --cuda-device-only --no-gpu-bundle-output -c "$SRC" -o "$OUT" || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT" | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|group_segment_fixed" \
  | sed -e 's/^ *//' | awk '/^\.name:/{if(line)print line; line=$2; next}{line=line"  "$1$2} END{print line}' | c++filt
