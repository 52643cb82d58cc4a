cd $GRAFT_REPO_ROOT
for v in 0 1 2; do echo "ENT decay $v: $(NEP_FE_KEY_DECAY=$v python scripts/fe_ent_time.py 32 7 2>&1 | grep -E '^round [3-6]' | cut -c1-50 | tr '\n' ' ')"; done
for v in 0 1 4 0 1 4; do NEP_FE_KEY_DECAY=$v python bench.py --steps 10 --warmup 3 --aux-steps 100 --no-cpu-baseline --no-config5 > gpurun_out/bench_lpt$v.json 2> gpurun_out/bench_lpt$v.err; python - <<EOF
import json
d=json.loads(open("gpurun_out/bench_lpt$v.json").read().strip().splitlines()[-1])
print("plain FE decay=$v", {k: (round(d[k]["value"]), round(d[k]["kernel_ms"]["frontend_with_hulls"],3)) for k in ("chain","moving","crossing") if d.get(k)})
EOF
done
