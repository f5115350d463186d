#!/usr/bin/env python
"""Copy the outputs of scripts/gpu_full_r2.sh from gpurun_out/ into profiles/ (tracked), named per round:
bench JSON lines (which carry their own PMC traffic / L2 hit rates: bench.py profiles itself under rocprofv3),
rocprofv3 kernel stats, and the SQ counters of the scoring kernels."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r02"


def last_json_line(path):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def main():
    os.makedirs(PROF, exist_ok=True)
    for src, dst in [("bench.log", f"bench_{RND}.json"), ("bench_k100.log", f"bench_{RND}_k100.json"),
                     ("bench_k1000.log", f"bench_{RND}_k1000.json"), ("bench_nogroup.log", f"bench_{RND}_per_query_kernel.json"),
                     ("bench_comm1.log", f"bench_{RND}_comm_1rank.json"), ("dist1_rccl.log", f"bench_{RND}_dist1rank_1250k.json"),
                     ("phrase_bench.log", f"phrase_bench_{RND}.json"), ("slop_bench.log", f"slop_bench_{RND}.json"),
                     ("io_bench.log", f"io_bench_{RND}.json"), ("sim_bench.log", f"sim_bench_{RND}.json")]:
        j = last_json_line(os.path.join(OUT, src))
        if j is not None:
            json.dump(j, open(os.path.join(PROF, dst), "w"), indent=1)
            print("wrote", dst)
    ab = os.path.join(OUT, "group_ab.log")
    if os.path.exists(ab):
        lines = [ln for ln in open(ab).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(PROF, f"group_ab_{RND}.jsonl"), "w").write("\n".join(lines) + "\n")
            print("wrote", f"group_ab_{RND}.jsonl")
    for sub, dst in [("prof_stats", f"{RND}_bench_kernel_stats.csv"), ("prof_phrase", f"{RND}_phrase_bench_kernel_stats.csv"),
                     ("prof_slop", f"{RND}_slop_bench_kernel_stats.csv"),
                     ("prof_slop2", f"{RND}_slop_heaviest_2term_kernel_stats.csv"),      # scripts/slop_heavy.py: 6 runs of ONE query
                     ("prof_slop3", f"{RND}_slop_heaviest_3term_kernel_stats.csv")]:
        f = newest(os.path.join(OUT, sub, "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(PROF, dst))
            print("wrote", dst)
    # HBM traffic of the slop pipeline (scripts/gpu_slop_pmc.sh + scripts/slop_pmc.py)
    if newest(os.path.join(OUT, "pmc_slop_f", "**", "*counter_collection.csv")):
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "slop_pmc.py"), OUT], capture_output=True, text=True)
        if r.returncode == 0 and r.stdout.strip().startswith("{"):
            open(os.path.join(PROF, f"{RND}_slop_pmc_traffic.json"), "w").write(r.stdout)
            print("wrote", f"{RND}_slop_pmc_traffic.json")
    hv = os.path.join(OUT, "slop_heavy.log")
    if os.path.exists(hv):
        lines = [ln for ln in open(hv).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(PROF, f"slop_heaviest_{RND}.jsonl"), "w").write("\n".join(lines) + "\n")
            print("wrote", f"slop_heaviest_{RND}.jsonl")
    # SQ counters of the scoring kernels (two passes of 8 counters), mean per dispatch
    sq = {}
    for sub in ("prof_grp_sq", "prof_grp_sq2"):
        f = newest(os.path.join(OUT, sub, "**", "*counter_collection.csv"))
        if not f:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "bm25" in k or "topk_merge" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            sq.setdefault(k, {}).update({c: round(sum(x) / len(x)) for c, x in v.items()})
            sq[k]["dispatches"] = len(next(iter(v.values())))
    if sq:
        out = {"command": "rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python scripts/group_ab.py --ks 10 --only 1 --qsets baseline --steps 2 "
                          "(two passes); 10M docs, 256 x 4-term BASELINE queries, top-10, grouped exhaustive path; mean per dispatch; "
                          "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
               "kernels": sq}
        json.dump(out, open(os.path.join(PROF, f"{RND}_scoring_kernels_sq_counters.json"), "w"), indent=1)
        print("wrote", f"{RND}_scoring_kernels_sq_counters.json")


if __name__ == "__main__":
    main()
