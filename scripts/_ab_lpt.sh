cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_kd.txt 2>&1; grep -E 'passed|failed' gpurun_out/pytest_kd.txt | tail -1
for v in 0 2 0 2 0 2; do NEP_QP_KEY_DECAY=$v python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs > gpurun_out/bench_lpt$v.json 2> gpurun_out/bench_lpt$v.err; python - <<EOF
import json
d=json.loads(open("gpurun_out/bench_lpt$v.json").read().strip().splitlines()[-1])
print("QP key decay=$v headline", round(d["value"]), round(d["ms_per_step"],4), round(d["kernel_ms"]["qp"],4))
EOF
done
