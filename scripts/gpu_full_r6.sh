#!/bin/bash
# round 6, full GPU pass on the final library: smoke, all gpu tests, the bench line and its variants, the staged-tile route against round 5's
# library and against the other routes by k and query set (route rule), its phase cycles (-DSA_PROBE build) and SQ counters, the
# no-impact-stream batches (where dynamic pruning is the default), the one-launch dense call against the rounds 1-5 route, host cost of a
# step, rocprofv3 kernel stats per leg, phrase / slop benches.  Copied into profiles/ by scripts/collect_profiles.py r06.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for d in prof_main prof_distinct prof_bench prof_sq1 prof_sq2 prof_slop prof_phrase prof_slopb prof_rank prof_k1000 prof_dense; do rm -rf $O/$d; done
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q -rxX ) > $O/pytest_gpu.log 2>&1
( time timeout 900 python bench.py --corpus-cache $C ) > $O/bench.log 2> $O/bench.err
( time timeout 300 python bench.py --corpus-cache $C --k 100 --no-cpu-baseline --no-phrase-legs ) > $O/bench_k100.log 2>&1
( time timeout 400 python bench.py --corpus-cache $C --k 1000 --no-cpu-baseline --no-phrase-legs ) > $O/bench_k1000.log 2>&1
( time SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --corpus-cache $C --no-cpu-baseline --no-pmc ) > $O/bench_comm1.log 2>&1
A="--no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 200 --pipeline 8 --corpus-cache $C"
( time RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py $A ) > $O/dist1_rccl.log 2>&1
( time timeout 300 python bench.py $A ) > $O/rank_nocomm.log 2>&1
# round 5's library (built from commit e89a768) and this round's on the same box, same resident batch
( time timeout 600 python scripts/ab.py --corpus-cache $C --ks 1,10,32,100,1000 --qsets baseline,distinct --libs build/libsearcharray_hip_r05.so --envs "SA_SPARSE=0" ) > $O/kernel_ab.log 2>&1
( time timeout 600 python scripts/ab.py --corpus-cache $C --ks 1,10,32,100,1000 --qsets baseline,distinct --libs searcharray_amd/libsearcharray_hip.so --envs "default=1" ) >> $O/kernel_ab.log 2>&1
# the route rule: every route forced, and the library's own choice, by k and query set (10 M docs; then the rank-sized shard)
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 1,10,32,100,1000 --qsets baseline,distinct,hot --envs "sparse=0,stage=0;sparse=1,stage=0;stage=1,trace=1;default=1" ) 2> $O/route_rule.err | grep "^{" > $O/route_rule.jsonl
( time timeout 600 python scripts/ab.py --corpus-cache $C --docs 1250000 --ks 10,100,1000 --qsets baseline,distinct --envs "sparse=0,stage=0;sparse=1,stage=0;stage=1;default=1" ) 2>> $O/route_rule.err | grep "^{" > $O/route_rule_1250k.jsonl
# batches without an impact stream (option impact = 0: HBM for the 8-byte-per-posting stream not spent): the TF kernels against dynamic pruning
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10,100,1000 --qsets baseline,distinct --envs "impact=0,sparse=0;impact=0,sparse=1;impact=0,default=1" ) 2>> $O/route_rule.err | grep "^{" > $O/route_rule_no_impact.jsonl
# the staged-tile kernel: phase cycles per tile pass (-DSA_PROBE), other tile sizes / workgroups per CU / everything streamed
( timeout 300 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline,hot --libs build/libsearcharray_hip_probe.so --envs "stage=1" ) 2>&1 | grep "^{" > $O/stage_probe.jsonl
( timeout 600 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --envs "stage=1;stage=1,stage_docs=768;stage=1,stage_docs=512;stage=1,stage_wgs=1;stage=1,stage_wgs=3;stage=1,stage_probe=0;stage=1,probe_div=32;stage=1,probe_div=512;stage=1,stage_cw=1;stage=1,stage_cw=4;stage=1,stage_cw=16;stage=1,stage_cw=64" ) 2>&1 | grep "^{" > $O/stage_sweep.jsonl
bash scripts/gpu_r6_shards.sh > /dev/null 2>&1
bash scripts/gpu_r6_pipe.sh > /dev/null 2>&1
( timeout 600 python scripts/first_batch_ms.py $C ) > $O/first_batch_ms.jsonl 2>/dev/null
( time timeout 300 python scripts/dense_ab.py --corpus-cache $C ) > $O/dense_ab.jsonl 2> $O/dense_ab.err
( time timeout 300 python scripts/host_cost.py --docs 1250000 ) > $O/host_cost.log 2>&1
( time timeout 300 python scripts/host_cost.py --docs 1250000 --comm ) >> $O/host_cost.log 2>&1
( time timeout 300 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
( time timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
( time timeout 120 python scripts/msmarco.py ) > $O/msmarco.log 2>&1
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_main -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --steps 12 --envs "default=1" ) > $O/prof_main.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_distinct -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets distinct --steps 12 --envs "default=1" ) > $O/prof_distinct.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1000 -- python $R/scripts/ab.py --corpus-cache $C --ks 1000 --qsets baseline --steps 12 --envs "default=1" ) > $O/prof_k1000.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dense -- python $R/scripts/dense_ab.py --corpus-cache $C --calls 12 ) > $O/prof_dense.log 2>&1
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-pmc --corpus-cache $C ) > $O/prof_bench.log 2>&1
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29534 SA_BENCH_FORCE_COMM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rank -- python $R/bench.py $A ) > $O/prof_rank.log 2>&1
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
( timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/prof_sq1 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --envs "default=1" ) > $O/prof_sq1.log 2>&1
( timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/prof_sq2 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --envs "default=1" ) > $O/prof_sq2.log 2>&1
python $R/scripts/sq_summary.py $O/prof_sq1 $O/prof_sq2 > $O/sq_summary.json
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_phrase -- python $R/scripts/phrase_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_phrase.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop -- python $R/scripts/slop_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_slop.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slopb -- python $R/scripts/slop_batch_prof.py ) > $O/prof_slopb.log 2>&1
cd /tmp
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*hip_api_trace.csv" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
grep -E "passed|failed" $O/pytest_gpu.log
exit 0
