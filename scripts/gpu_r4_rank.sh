#!/bin/bash
# round 4: the rank-sized shard (1.25 M docs), with and without the one-rank RCCL exchange, + kernel stats of the exchange run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
A="--no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 200 --pipeline ${PIPE:-8} --corpus-cache /tmp/corpus"
( timeout 300 python bench.py $A ) > $O/rank_nocomm.log 2>&1
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py $A ) > $O/dist1_rccl.log 2>&1
python - <<'PY'
import json
for f in ("rank_nocomm", "dist1_rccl"):
    l = [x for x in open(f"/root/repo/gpurun_out/{f}.log") if x.startswith("{")]
    if not l:
        print(f, "no line"); continue
    d = json.loads(l[-1])
    print(f, d["value"], d["ms_per_step"], "latency", d.get("fresh_batch_latency_ms"), "replay", d.get("replay", {}).get("ms_per_step"), d.get("replay", {}).get("kernel_ms"),
          "wide", (d.get("wide_batch") or {}).get("value"), (d.get("wide_batch") or {}).get("ms_per_step"))
PY
cd /tmp
rm -rf $O/prof_rank
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29534 SA_BENCH_FORCE_COMM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rank -- python $R/bench.py $A ) > $O/prof_rank.log 2>&1
find $O/prof_rank -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | head -24
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
exit 0
