#!/bin/bash
# round 3, full GPU pass on the final library: smoke, all gpu tests, the bench line and its variants, same-box A/B against
# round 2's library, side benches, rocprofv3 kernel stats PER LEG, SQ counters, the HIP-API trace of the fresh-batch loop.
# Everything is copied into profiles/ by scripts/collect_profiles.py r03.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_main $O/prof_distinct $O/prof_bench $O/prof_sq1 $O/prof_sq2 $O/prof_hip_a $O/prof_hip_b $O/prof_slop $O/prof_phrase $O/prof_slopb $O/prof_slop2 $O/prof_slop3 $O/pmc_phrase_f $O/pmc_phrase_w
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( time timeout 900 python bench.py --corpus-cache $C ) > $O/bench.log 2>&1
( time timeout 300 python bench.py --corpus-cache $C --k 100 --no-cpu-baseline --no-phrase-legs ) > $O/bench_k100.log 2>&1
( time timeout 300 python bench.py --corpus-cache $C --k 1000 --no-cpu-baseline --no-phrase-legs ) > $O/bench_k1000.log 2>&1
( time SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --corpus-cache $C --no-cpu-baseline --no-pmc ) > $O/bench_comm1.log 2>&1
( time RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --docs 1250000 --steps 100 ) > $O/dist1_rccl.log 2>&1
( time timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 100 --pipeline 8 ) > $O/rank_nocomm.log 2>&1
if [ -f searcharray_amd/libsearcharray_hip_r2.so ]; then     # (round 2's library, rebuilt from commit dfc26d0 into the tree for the A/B)
( time timeout 600 python scripts/ab.py --corpus-cache $C --libs searcharray_amd/libsearcharray_hip_r2.so,searcharray_amd/libsearcharray_hip.so --ks 10,100,1000 --qsets baseline,distinct ) > $O/kernel_ab.log 2>&1
( time timeout 300 python scripts/ab.py --docs 1250000 --steps 50 --libs searcharray_amd/libsearcharray_hip_r2.so,searcharray_amd/libsearcharray_hip.so --ks 10 --qsets baseline ) >> $O/kernel_ab.log 2>&1
fi
( time timeout 300 python scripts/host_cost.py --docs 1250000 ) > $O/host_cost.log 2>&1
( time timeout 300 python scripts/host_cost.py --docs 1250000 --comm ) >> $O/host_cost.log 2>&1
( time timeout 300 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
( time timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
( time timeout 300 python scripts/slop_routes.py ) > $O/slop_routes.log 2>&1
cd /tmp
# kernel stats per leg: the main leg alone (one resident batch replayed), the distinct-terms leg alone, then the whole bench
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_main -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --steps 12 ) > $O/prof_main.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_distinct -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets distinct --steps 12 ) > $O/prof_distinct.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --corpus-cache $C ) > $O/prof_bench.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/prof_sq1 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 ) > $O/prof_sq1.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/prof_sq2 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 ) > $O/prof_sq2.log 2>&1
python $R/scripts/sq_summary.py $O/prof_sq1 $O/prof_sq2 > $O/sq_summary.json
# HIP API trace of the fresh-batch loop, two lengths: what the steady state calls is the difference
( timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $O/prof_hip_a -- python $R/scripts/fresh_trace.py --corpus-cache $C --steps 100 ) > $O/prof_hip_a.log 2>&1
( timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $O/prof_hip_b -- python $R/scripts/fresh_trace.py --corpus-cache $C --steps 600 ) > $O/prof_hip_b.log 2>&1
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
find $O -type f -size +8M -delete 2>/dev/null
grep -E "passed|failed" $O/pytest_gpu.log
exit 0
