cd $GRAFT_REPO_ROOT
O=gpurun_out/${NEP_SWEEP_OUT:-parity_sweep.txt}
echo "# scripts/parity_sweep.py with the kernels at HEAD: EVERY replan of the scenes against the CPU oracle" > $O
for cfg in "5 0 40" "8 20 40" "64 20 16" "16 8 24"; do set -- $cfg
  echo "== agents $1 statics $2 scenes $3: front-end guesses, then the scenes own guesses" >> $O
  NEP_AGENTS=$1 NEP_STATICS=$2 python scripts/parity_sweep.py $3 2>/dev/null >> $O
  NEP_NO_FRONTEND=1 NEP_AGENTS=$1 NEP_STATICS=$2 python scripts/parity_sweep.py $3 2>/dev/null >> $O
done
echo "== agents 256 statics 100 scenes 2 (config-5 size, the handle default: presolve, packed separator, qp_reg_kernel): the scenes own guesses" >> $O
NEP_NO_FRONTEND=1 NEP_AGENTS=256 NEP_STATICS=100 python scripts/parity_sweep.py 2 2>/dev/null >> $O
echo "== stress: v_max 2.0 a_max 1.0 (many active rows, relaxed and failed solves), 64 agents, 8 scenes, own guesses then front-end guesses" >> $O
NEP_AMAX=1.0 NEP_NO_FRONTEND=1 python scripts/parity_sweep.py 8 2>/dev/null >> $O
NEP_AMAX=1.0 python scripts/parity_sweep.py 4 2>/dev/null >> $O
cat $O
