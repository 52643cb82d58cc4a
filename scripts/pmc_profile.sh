#!/bin/bash
# PMC passes for the bench command (one counter group per pass; never combined with sys/hip traces).
# Usage (on the GPU box, from the repo root): bash scripts/pmc_profile.sh <outdir> [bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o pmc --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-chain --no-full-rows ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
find "$OUT" -name "*.csv" | head -20
