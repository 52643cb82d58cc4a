#!/bin/bash
# One call on the GPU box: the default bench line, the rocprofv3 kernel traces of the headline leg and of the chain / moving legs,
# and the PMC passes of both (one counter group per pass; never combined with sys/hip traces).
# Usage: bash scripts/profile_bench.sh <tag>  ->  gpurun_out/<tag>/{bench_line.json,bench_detail.json,bench_kernel_stats.txt,bench_chain_kernel_stats.txt,pmc_summary.txt,pmc_summary_chain.txt}
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail.json" > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"      # the driver's command; bench_line.json = the short line it parses
HEAD="--steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-graph"
CHAIN="--steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-graph --no-full-rows --aux-steps 10"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format rocpd -- python bench.py $HEAD > "$OUT/kt.log" 2>&1
python scripts/rocpd_summary.py "$(find "$OUT/kt" -name "*.db" | head -1)" > "$OUT/bench_kernel_stats.txt" 2>> "$OUT/kt.log"
rocprofv3 --kernel-trace --stats -d "$OUT/ktc" -o kt --output-format rocpd -- python bench.py $CHAIN > "$OUT/ktc.log" 2>&1
python scripts/rocpd_summary.py "$(find "$OUT/ktc" -name "*.db" | head -1)" > "$OUT/bench_chain_kernel_stats.txt" 2>> "$OUT/ktc.log"
pmc() { dir=$1; shift; args=$1; shift; name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$dir/$name" -o pmc --output-format csv -- python bench.py $args > "$OUT/${dir}_$name.log" 2>&1; }
for leg in "pmc|--steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --no-graph" "pmcc|--steps 4 --warmup 2 --no-cpu-baseline --no-config5 --no-graph --no-full-rows --aux-steps 4"; do
  dir=${leg%%|*}; args=${leg#*|}
  pmc $dir "$args" fetch FETCH_SIZE
  pmc $dir "$args" write WRITE_SIZE
  pmc $dir "$args" sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
  pmc $dir "$args" f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA
  pmc $dir "$args" sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
done
python scripts/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt"
python scripts/pmc_summary.py "$OUT/pmcc" > "$OUT/pmc_summary_chain.txt"
rm -rf "$OUT/kt" "$OUT/ktc" "$OUT/pmc" "$OUT/pmcc"
python scripts/bench_brief.py "$OUT/bench_detail.json"; head -8 "$OUT/bench_kernel_stats.txt"; head -14 "$OUT/bench_chain_kernel_stats.txt"
