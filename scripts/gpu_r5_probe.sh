#!/bin/bash
# round 5: issue-slot / LDS microbenchmark (scripts/issue_probe.hip)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 120 build/issue_probe 400 ) > $O/issue_probe.jsonl 2> $O/issue_probe.err
exit 0
