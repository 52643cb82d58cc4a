cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frontend_entangle.py -x -q 2>&1 | tail -1
for v in 1 1; do echo "ENT bends: $(NEP_SCRIPT_BENDS=1 python scripts/fe_ent_time.py 32 5 2>&1 | grep -E '^round [1-4]' | sed 's/.*searches \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')"; done
for v in 1; do echo "ENT plain: $(python scripts/fe_ent_time.py 32 5 2>&1 | grep -E '^round [1-4]' | sed 's/.*searches \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')"; done
