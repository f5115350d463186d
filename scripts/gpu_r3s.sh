#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for d in 4 8 16 2; do
echo "SA_DENSE_DIV=$d"
SA_DENSE_DIV=$d timeout 300 python scripts/ab.py --corpus-cache /tmp/corpus --ks 10 --qsets baseline --steps 30 2>/dev/null | grep "^{" | cut -c1-300
done
