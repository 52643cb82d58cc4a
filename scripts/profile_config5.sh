#!/bin/bash
# One call on the GPU box: BASELINE configs[4] (256 agents + 100 obstacles, entangle check on) through `bench.py --config5-only`:
# its JSON leg, the rocprofv3 kernel trace of the same command and the PMC passes (one counter group per pass).
# Usage: bash scripts/profile_config5.sh <tag> [extra bench args]   ->  gpurun_out/<tag>/{config5_line.json,config5_kernel_stats.txt,pmc_summary_config5.txt,config5_chain_kernel_stats.txt}
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export NEP_BENCH_SCENE_CACHE=/tmp/nep_c5_scenes.pkl
ARGS="--config5-only --no-cpu-baseline --no-graph $*"
python bench.py --config5-only --steps 100 --warmup 5 --detail "$OUT/config5_line.json" "$@" > "$OUT/config5_short.json" 2> "$OUT/config5_line.err"      # config5_line.json: the leg's full record
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format rocpd -- python bench.py $ARGS --steps 20 --warmup 3 > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name "*.db" | head -1)
python scripts/rocpd_summary.py "$DB" > "$OUT/config5_kernel_stats.txt" 2>> "$OUT/kt.log"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc/$name" -o pmc --output-format csv -- python bench.py $ARGS --steps 6 --warmup 2 > "$OUT/pmc_$name.log" 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python scripts/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary_config5.txt"
# the config-5 chain (frontend_kernel<true>, ent_check_kernel, ...): kernel trace of the default command with short legs
rocprofv3 --kernel-trace --stats -d "$OUT/ktf" -o kt --output-format rocpd -- python bench.py --steps 5 --warmup 2 --aux-steps 10 --no-full-rows --no-cpu-baseline --no-graph > "$OUT/ktf.log" 2>&1
python scripts/rocpd_summary.py "$(find "$OUT/ktf" -name "*.db" | head -1)" > "$OUT/config5_chain_kernel_stats.txt" 2>> "$OUT/ktf.log"
rm -rf "$OUT/kt" "$OUT/pmc" "$OUT/ktf"
head -12 "$OUT/config5_kernel_stats.txt"; python -c "
import json,sys; d=json.load(open('$OUT/config5_line.json'))['config5']; print({k:d[k] for k in ('value','ms_per_step','kernel_ms','qp_kernel','line_cull_radius_m','presolve_redo_last_step')})"
