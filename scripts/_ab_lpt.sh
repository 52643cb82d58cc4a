cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frontend_entangle.py tests/test_gpu_parity.py -x -q -k "entangle or safety or config5" 2>&1 | tail -1
NEP_SCRIPT_BENDS=1 python scripts/fe_ent_time.py 32 4 2>&1 | grep -E '^round|safety pass' | cut -c1-110
