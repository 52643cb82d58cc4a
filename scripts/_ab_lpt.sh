cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do NEP_FE_LPT=$v python bench.py --steps 10 --warmup 3 --aux-steps 100 --no-cpu-baseline --no-config5 > gpurun_out/bench_lpt$v.json 2> gpurun_out/bench_lpt$v.err; python - <<EOF
import json
d=json.loads(open("gpurun_out/bench_lpt$v.json").read().strip().splitlines()[-1])
print("FE_LPT=$v", {k: (round(d[k]["value"]), round(d[k]["kernel_ms"]["frontend_with_hulls"],3)) for k in ("chain","moving","crossing") if d.get(k)})
EOF
done
