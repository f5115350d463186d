cd $GRAFT_REPO_ROOT
python scripts/ab.py --docs 1250000 --steps 100 --envs "SA_GROUP_WARM=16;SA_GROUP_WARM=8;SA_GROUP_WARM=4;SA_GROUP_WARM=2;SA_GROUP_WARM=16,SA_GROUP_SIDE=0" --ks 10 --qsets baseline 2>&1 | grep "^{"
python scripts/host_cost.py --docs 1250000 2>&1 | tail -5
