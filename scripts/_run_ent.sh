cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frontend_entangle.py -x -q > gpurun_out/pytest_ent.txt 2>&1; tail -2 gpurun_out/pytest_ent.txt
python scripts/fe_ent_time.py 32 4 2>&1 | grep round
python scripts/fe_ent_phases.py 4 > gpurun_out/fe_ent_phases_new.txt 2>&1
