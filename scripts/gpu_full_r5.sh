#!/bin/bash
# round 5, full GPU pass on the final library: smoke, all gpu tests, the bench line and its variants, host cost of a step, the
# rank-sized shard with and without the one-rank exchange, rocprofv3 kernel stats PER LEG, SQ counters of the grouped kernel,
# the route-rule sweep, threaded dense calls, the issue-slot microbenchmark, slop / phrase profiles.  Everything is copied into
# profiles/ by scripts/collect_profiles.py r05.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for d in prof_main prof_distinct prof_bench prof_sq1 prof_sq2 prof_slop prof_phrase prof_slopb prof_slop2 prof_slop3 pmc_phrase_f pmc_phrase_w prof_rank prof_k1000; do rm -rf $O/$d; done
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q -rxX ) > $O/pytest_gpu.log 2>&1
( time timeout 120 build/issue_probe 400 ) > $O/issue_probe.jsonl 2> $O/issue_probe.err
( time timeout 900 python bench.py --corpus-cache $C ) > $O/bench.log 2> $O/bench.err
( time timeout 300 python bench.py --corpus-cache $C --k 100 --no-cpu-baseline --no-phrase-legs ) > $O/bench_k100.log 2>&1
( time timeout 400 python bench.py --corpus-cache $C --k 1000 --no-cpu-baseline --no-phrase-legs ) > $O/bench_k1000.log 2>&1
( time SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --corpus-cache $C --no-cpu-baseline --no-pmc ) > $O/bench_comm1.log 2>&1
A="--no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 200 --pipeline 8 --corpus-cache $C"
( time RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py $A ) > $O/dist1_rccl.log 2>&1
( time timeout 300 python bench.py $A ) > $O/rank_nocomm.log 2>&1
( time timeout 600 python scripts/ab.py --corpus-cache $C --ks 10,100,1000 --qsets baseline,distinct --libs build/libsearcharray_hip_r04.so,searcharray_amd/libsearcharray_hip.so --envs "SA_SPARSE=0" ) > $O/kernel_ab.log 2>&1
( timeout 400 python scripts/ab.py --corpus-cache $C --ks 10,100,1000 --qsets baseline,distinct --libs searcharray_amd/libsearcharray_hip.so --envs "SA_SPARSE=0,group_item=16;SA_SPARSE=0;SA_SPARSE=0,group_item=64" ) 2>&1 | grep "^{" > $O/item_ab.jsonl
bash scripts/gpu_r5_item.sh > /dev/null 2>&1
( time timeout 300 python scripts/host_cost.py --docs 1250000 ) > $O/host_cost.log 2>&1
( time timeout 300 python scripts/host_cost.py --docs 1250000 --comm ) >> $O/host_cost.log 2>&1
( time timeout 900 python scripts/route_rule.py --steps 30 ) > $O/route_rule.jsonl 2> $O/route_rule.err
( time timeout 600 python scripts/route_rule.py --steps 30 --docs 1250000 ) > $O/route_rule_1250k.jsonl 2>> $O/route_rule.err
( time timeout 300 python scripts/dense_threads.py ) > $O/dense_threads.jsonl 2> $O/dense_threads.err
( time timeout 300 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
( time timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
( time timeout 300 python scripts/slop_routes.py ) > $O/slop_routes.log 2>&1
( time timeout 120 python scripts/msmarco.py ) > $O/msmarco.log 2>&1
# where a wave of the grouped kernel spends its cycles (-DSA_PROBE builds: scripts/build_probe.sh), resident waves per SIMD 4 / 3 / 2 / 1
( timeout 300 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline,distinct,hot --libs build/libsearcharray_hip_probe.so --envs "SA_SPARSE=0,group_item=16;SA_SPARSE=0" ) 2>&1 | grep "^{" > $O/probe_sections.jsonl
( timeout 300 python scripts/ab.py --corpus-cache $C --ks 10 --docs 1250000 --qsets baseline --libs build/libsearcharray_hip_probe.so --envs "SA_SPARSE=0" ) 2>&1 | grep "^{" >> $O/probe_sections.jsonl
( timeout 300 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --libs build/libsearcharray_hip_probe_fine.so --envs "SA_SPARSE=0,group_item=16" ) 2>&1 | grep "^{" > $O/probe_sections_fine.jsonl
rm -f $O/occupancy.jsonl
for pad in 0 3200 10240 30720; do
  ( SA_PROBE_LDS_PAD=$pad timeout 300 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --libs build/libsearcharray_hip_probe.so --envs "SA_SPARSE=0,group_item=16" ) 2>&1 | grep "^{" | sed "s/^{/{\"lds_pad\": $pad, /" >> $O/occupancy.jsonl
done
( timeout 120 build/lds_fadd_probe ) > $O/lds_fadd_probe.json 2>&1
bash $R/scripts/gpu_r5_latency.sh > $O/latency.log 2>&1
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_main -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --steps 12 ) > $O/prof_main.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_distinct -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets distinct --steps 12 ) > $O/prof_distinct.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1000 -- python $R/scripts/ab.py --corpus-cache $C --ks 1000 --qsets baseline --steps 12 ) > $O/prof_k1000.log 2>&1
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-pmc --corpus-cache $C ) > $O/prof_bench.log 2>&1
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29534 SA_BENCH_FORCE_COMM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rank -- python $R/bench.py $A ) > $O/prof_rank.log 2>&1
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
( timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/prof_sq1 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 ) > $O/prof_sq1.log 2>&1
( timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/prof_sq2 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 ) > $O/prof_sq2.log 2>&1
python $R/scripts/sq_summary.py $O/prof_sq1 $O/prof_sq2 > $O/sq_summary.json
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_phrase -- python $R/scripts/phrase_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_phrase.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop -- python $R/scripts/slop_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_slop.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slopb -- python $R/scripts/slop_batch_prof.py ) > $O/prof_slopb.log 2>&1
( timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_phrase_f -- python $R/scripts/phrase_bench.py --phrases 16 --cpu-phrases 1 ) > $O/pmc_phrase_f.log 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_phrase_w -- python $R/scripts/phrase_bench.py --phrases 16 --cpu-phrases 1 ) > $O/pmc_phrase_w.log 2>&1
bash $R/scripts/gpu_slop_prof.sh > $O/slop_heavy.log 2>&1
bash $R/scripts/gpu_slop_pmc.sh > $O/slop_pmc.log 2>&1
cd /tmp
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*hip_api_trace.csv" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
grep -E "passed|failed" $O/pytest_gpu.log
exit 0
