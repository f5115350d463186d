#!/usr/bin/env python
"""Copy the outputs of scripts/gpu_full_r6.sh (r05: gpu_full_r5.sh, r04: gpu_full_r4.sh, r03: gpu_full_r3.sh) from gpurun_out/ (scratch) into profiles/ (tracked), named per round:
bench JSON lines (which carry their own PMC traffic / L2 hit rates: bench.py profiles itself under rocprofv3),
rocprofv3 kernel stats per leg, SQ counters of the scoring kernels, the HIP-API summary of the fresh-batch loop's
steady state, PMC traffic of the phrase and slop kernels."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"


def json_lines(path):
    if not os.path.exists(path):
        return []
    out = []
    for line in open(path).read().splitlines():
        if line.startswith("{"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def hip_api_counts(sub):
    """{api: (calls, total ns)} from a rocprofv3 --hip-trace --stats run"""
    f = newest(os.path.join(OUT, sub, "**", "*hip_api_stats.csv")) or newest(os.path.join(OUT, sub, "**", "*hip_stats.csv"))
    if not f:
        return None
    res = {}
    for r in csv.DictReader(open(f)):
        res[r["Name"]] = (int(r["Calls"]), int(float(r["TotalDurationNs"])))
    return res


def pmc_per_kernel(fetch_dir, write_dir, prefix="sa_k_"):
    """mean per dispatch: FETCH_SIZE (KiB, raw), WRITE_SIZE (KiB), L2 hit rate, duration"""
    acc = defaultdict(lambda: defaultdict(list))
    for sub in (fetch_dir, write_dir):
        f = newest(os.path.join(OUT, sub, "**", "*counter_collection.csv"))
        if not f:
            continue
        per = defaultdict(lambda: defaultdict(float))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith(prefix):
                continue
            did = int(r["Dispatch_Id"])
            meta[did] = (k, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            per[did][r["Counter_Name"]] += float(r["Counter_Value"])
        for did, cs in per.items():
            k, dur = meta[did]
            for c, v in cs.items():
                acc[k][c].append(v)
            acc[k]["_dur_" + sub].append(dur)
    out = {}
    for k, cs in acc.items():
        e = {"dispatches": max(len(v) for v in cs.values())}
        if "FETCH_SIZE" in cs:
            e["fetch_KiB_raw"] = round(sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), 1)
            e["fetch_bytes_x2_wide_read_correction"] = int(e["fetch_KiB_raw"] * 1024 * 2)
        if "WRITE_SIZE" in cs:
            e["write_KiB"] = round(sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]), 1)
        h, m = sum(cs.get("TCC_HIT_sum", [])), sum(cs.get("TCC_MISS_sum", []))
        if h + m > 0:
            e["l2_hit_rate"] = round(h / (h + m), 4)
        durs = [x for kk, v in cs.items() if kk.startswith("_dur_") for x in v]
        if durs:
            e["mean_us_under_pmc"] = round(sum(durs) / len(durs) / 1e3, 2)
        out[k] = e
    return out


def main():
    os.makedirs(PROF, exist_ok=True)
    for src, dst in [("bench.log", f"bench_{RND}.json"), ("bench_k100.log", f"bench_{RND}_k100.json"),
                     ("bench_k1000.log", f"bench_{RND}_k1000.json"), ("bench_comm1.log", f"bench_{RND}_comm_1rank.json"),
                     ("dist1_rccl.log", f"bench_{RND}_dist1rank_1250k.json"), ("rank_nocomm.log", f"bench_{RND}_rank_sized_1250k_nocomm.json"),
                     ("phrase_bench.log", f"phrase_bench_{RND}.json"), ("slop_bench.log", f"slop_bench_{RND}.json"),
                     ("msmarco.log", f"msmarco_{RND}.json")]:
        js = json_lines(os.path.join(OUT, src))
        if js:
            json.dump(js[-1], open(os.path.join(PROF, dst), "w"), indent=1)
            print("wrote", dst)
    for src, dst in [("kernel_ab.log", f"kernel_ab_{RND}.jsonl"), ("host_cost.log", f"host_cost_{RND}.jsonl"),
                     ("route_rule.jsonl", f"route_rule_{RND}.jsonl"), ("route_rule_1250k.jsonl", f"route_rule_{RND}_1250k.jsonl"),
                     ("dense_threads.jsonl", f"dense_threads_{RND}.jsonl"), ("issue_probe.jsonl", f"issue_probe_{RND}.jsonl"),
                     ("probe_sections.jsonl", f"group_kernel_cycle_sections_{RND}.jsonl"), ("probe_sections_fine.jsonl", f"group_kernel_cycle_sections_fine_{RND}.jsonl"),
                     ("occupancy.jsonl", f"group_kernel_occupancy_{RND}.jsonl"), ("item_ab.jsonl", f"group_item_passes_ab_{RND}.jsonl"), ("item_sweep.jsonl", f"group_item_passes_by_shard_size_{RND}.jsonl"), ("lds_fadd_probe.json", f"lds_fadd_probe_{RND}.jsonl"),
                     ("slop_heavy.log", f"slop_heaviest_{RND}.jsonl"), ("slop_routes.log", f"slop_routes_{RND}.jsonl"),
                     # round 6: the staged-tile route
                     ("route_rule_no_impact.jsonl", f"route_rule_{RND}_no_impact_stream.jsonl"), ("stage_probe.jsonl", f"stage_kernel_cycle_sections_{RND}.jsonl"),
                     ("stage_sweep.jsonl", f"stage_kernel_option_sweep_{RND}.jsonl"), ("dense_ab.jsonl", f"dense_call_one_launch_ab_{RND}.jsonl"),
                     ("shard_routes.jsonl", f"shard_routes_{RND}.jsonl"), ("pipeline_sweep.jsonl", f"pipeline_sweep_{RND}.jsonl"),
                     ("ab_sweep1.log", f"stage_kernel_streaming_loads_ab_{RND}.jsonl"), ("ab_x2b.log", f"stage_kernel_hoisted_search_ab_{RND}.jsonl"),
                     ("ab_mix.log", f"stage_kernel_register_diet_ab_{RND}.jsonl")]:
        js = json_lines(os.path.join(OUT, src))
        if js:
            open(os.path.join(PROF, dst), "w").write("\n".join(json.dumps(j) for j in js) + "\n")
            print("wrote", dst)
    for sub, dst in [("prof_main", f"{RND}_main_leg_kernel_stats.csv"), ("prof_distinct", f"{RND}_distinct_leg_kernel_stats.csv"),
                     ("prof_bench", f"{RND}_bench_kernel_stats.csv"), ("prof_phrase", f"{RND}_phrase_bench_kernel_stats.csv"),
                     ("prof_slop", f"{RND}_slop_bench_kernel_stats.csv"), ("prof_slopb", f"{RND}_phrase_slop_batch_legs_kernel_stats.csv"),
                     ("prof_k1000", f"{RND}_main_leg_k1000_kernel_stats.csv"), ("prof_dense", f"{RND}_dense_call_ab_kernel_stats.csv"),
                     ("prof_rank", f"{RND}_rank_sized_shard_exchange_kernel_stats.csv"),
                     ("prof_slop2", f"{RND}_slop_heaviest_2term_kernel_stats.csv"),
                     ("prof_slop3", f"{RND}_slop_heaviest_3term_kernel_stats.csv")]:
        f = newest(os.path.join(OUT, sub, "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(PROF, dst))
            print("wrote", dst)
    lat = os.path.join(OUT, "latency_summary.json")
    if os.path.exists(lat) and os.path.getsize(lat) > 10:
        json.dump({"command": "rocprofv3 --pmc VmemLatency | LdsLatency | SmemLatency --kernel-trace -- python scripts/ab.py --ks 10 --steps 2 (one pass per "
                              "counter; derived counters: in-flight level accumulated per cycle / instructions of the class); cycles, mean per dispatch",
                   "kernels": json.load(open(lat))}, open(os.path.join(PROF, f"{RND}_scoring_kernels_latency_counters.json"), "w"), indent=1)
        print("wrote", f"{RND}_scoring_kernels_latency_counters.json")
    sq = os.path.join(OUT, "sq_summary.json")
    if os.path.exists(sq) and os.path.getsize(sq) > 10:
        out = {"command": "rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python scripts/ab.py --ks 10 --steps 2 (two passes); 10M docs, "
                          "256 x 4-term BASELINE queries, top-10, one resident batch (the library's default route: r06 staged, r02-r05 grouped exhaustive); mean per dispatch; "
                          "SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles",
               "kernels": json.load(open(sq))}
        json.dump(out, open(os.path.join(PROF, f"{RND}_scoring_kernels_sq_counters.json"), "w"), indent=1)
        print("wrote", f"{RND}_scoring_kernels_sq_counters.json")
    # the steady state of the fresh-batch loop: HIP API calls of 600 steps minus those of 100 steps, per step
    a, b = hip_api_counts("prof_hip_a"), hip_api_counts("prof_hip_b")
    if a and b:
        la, lb = json_lines(os.path.join(OUT, "prof_hip_a.log")), json_lines(os.path.join(OUT, "prof_hip_b.log"))
        sa, sb = (la[-1]["steps"] if la else 100), (lb[-1]["steps"] if lb else 600)
        per_step = {}
        for name in sorted(set(a) | set(b)):
            d = b.get(name, (0, 0))[0] - a.get(name, (0, 0))[0]
            if d:
                per_step[name] = round(d / (sb - sa), 3)
        alloc = {n: (a.get(n, (0, 0))[0], b.get(n, (0, 0))[0]) for n in sorted(set(a) | set(b))
                 if any(x in n for x in ("Malloc", "Free", "HostRegister", "hipMemcpy ", "hipMemcpy\"")) or n in ("hipMemcpy", "hipStreamSynchronize", "hipDeviceSynchronize")}
        json.dump({"command": f"rocprofv3 --hip-trace --stats -- python scripts/fresh_trace.py --steps {sa} | {sb} (10 M docs, 8 rotating query "
                              "sets, 6 batches in flight: idf gather + sa_batch_reset + sa_batch_run + sa_batch_fetch per step)",
                   "runs": [la[-1] if la else None, lb[-1] if lb else None],
                   "hip_calls_per_steady_state_step": per_step,
                   "allocation_and_blocking_calls_total_in_run_a_vs_run_b": alloc,
                   "note": "identical totals in both runs = none of these is called in the steady state"},
                  open(os.path.join(PROF, f"{RND}_fresh_batches_hip_api.json"), "w"), indent=1)
        print("wrote", f"{RND}_fresh_batches_hip_api.json")
    ph = pmc_per_kernel("pmc_phrase_f", "pmc_phrase_w", prefix="sa_k_phrase")
    if ph:
        json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -- python scripts/phrase_bench.py "
                              "--phrases 16 --cpu-phrases 1 (zipf-1M; single dense calls: sa_k_phrase_fused; 256-phrase batches: sa_k_phrase_tiles); "
                              "mean per dispatch", "kernels": ph},
                  open(os.path.join(PROF, f"{RND}_phrase_pmc_traffic.json"), "w"), indent=1)
        print("wrote", f"{RND}_phrase_pmc_traffic.json")
    sb = pmc_per_kernel("pmc_slopb_f_new", "pmc_slopb_w_new", prefix="sa_k_span")
    if sb:
        json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -- python scripts/slop_batch_prof.py "
                              "(zipf-1M; bench.py's slop_batch leg: 256 two-token slop-2 phrases, half over terms of ranks 1-50); mean per dispatch",
                   "kernels": sb}, open(os.path.join(PROF, f"{RND}_slop_batch_pmc_traffic.json"), "w"), indent=1)
        print("wrote", f"{RND}_slop_batch_pmc_traffic.json")
    if newest(os.path.join(OUT, "pmc_slop_f", "**", "*counter_collection.csv")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "slop_pmc.py"), OUT], capture_output=True, text=True)
        if r.returncode == 0 and r.stdout.strip().startswith("{"):
            open(os.path.join(PROF, f"{RND}_slop_pmc_traffic.json"), "w").write(r.stdout)
            print("wrote", f"{RND}_slop_pmc_traffic.json")


if __name__ == "__main__":
    main()
