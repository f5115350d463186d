cd $GRAFT_REPO_ROOT
python scripts/ab.py --corpus-cache /tmp/corpus --envs "SA_HG=0;SA_HG=1;SA_HG=1,SA_GROUP_WARM=4" --ks 10,100,1000 --qsets baseline 2>&1 | grep "^{"
python scripts/ab.py --docs 1250000 --steps 50 --envs "SA_HG=0;SA_HG=1;SA_HG=1,SA_GROUP_WARM=4" --ks 10 --qsets baseline 2>&1 | grep "^{"
