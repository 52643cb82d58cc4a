#!/bin/bash
# PMC passes of the entangle front end on the bench's config-5 inputs (256 agents + 100 obstacles, tethers of 2-4 bend points, 32 scenes):
# one counter group per pass, kernel trace only (never with sys / hip traces).   Usage: bash scripts/profile_fe_ent.sh <tag>
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export NEP_SCRIPT_BENDS=1
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmcfe/$name" -o pmc --output-format csv -- python scripts/fe_ent_time.py 32 3 > "$OUT/pmcfe_$name.log" 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python scripts/pmc_summary.py "$OUT/pmcfe" > "$OUT/pmc_summary_fe_ent.txt"
rm -rf "$OUT/pmcfe"
grep -A20 'frontend_kernel<true, 3' "$OUT/pmc_summary_fe_ent.txt" | head -22
