#!/bin/bash
# One call on the GPU box: the headline bench line, the rocprofv3 kernel trace of the same command and the PMC passes.
# Usage: bash scripts/profile_bench.sh <tag>     ->  gpurun_out/<tag>/{bench_line.json,bench_kernel_stats.txt,pmc_summary.txt}
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format rocpd -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-chain --presolve-radius 0 > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name "*.db" | head -1)
python scripts/rocpd_summary.py "$DB" > "$OUT/bench_kernel_stats.txt" 2>> "$OUT/kt.log"
rocprofv3 --kernel-trace --stats -d "$OUT/ktc" -o kt --output-format rocpd -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --presolve-radius 0 > "$OUT/ktc.log" 2>&1
DBC=$(find "$OUT/ktc" -name "*.db" | head -1)
python scripts/rocpd_summary.py "$DBC" > "$OUT/bench_chain_kernel_stats.txt" 2>> "$OUT/ktc.log"
bash scripts/pmc_profile.sh "$OUT/pmc" > /dev/null 2>&1
python scripts/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt"
rm -rf "$OUT/kt" "$OUT/ktc" "$OUT/pmc"
python scripts/bench_brief.py "$OUT/bench_line.json"; head -8 "$OUT/bench_kernel_stats.txt"
