#!/bin/bash
# round 6: the staged kernel's 1024-thread instance (one workgroup per CU, 2048-doc tiles: option stage_docs = 2048) against the default
# (two 512-thread workgroups per CU, 1024-doc tiles), same box, same library.  The instance was measured and NOT kept: this script
# needs profiles/stage_kernel_1024_thread_instance_r06.patch applied to csrc/sa_stage.hip (without it stage_docs is clamped to 1024)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( timeout 1200 python scripts/ab.py --corpus-cache $C --ks 1,10,32,100 --qsets baseline,distinct --envs "stage=1,trace=1;stage=1,stage_docs=2048,trace=1;stage=1;stage=1,stage_docs=2048;stage=1,stage_docs=1536" ) 2>&1 | grep -E "^\{|sa_launch_stage: rows 0" | sort -u | cut -c1-420 > $O/stage_big.jsonl
cat $O/stage_big.jsonl
( timeout 600 python -m pytest tests/test_stage.py -m gpu -q -x 2>&1 | tail -2 )
exit 0
