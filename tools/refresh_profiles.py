"""Condenses the rocprofv3 outputs of a round-end run into the files committed under profiles/ (see profiles/README.md).

usage: python tools/refresh_profiles.py r02      # reads gpurun_out/<tag>_prof_head, gpurun_out/<tag>_prof_final, gpurun_out/<tag>_bench_default.json
On the GPU box the inputs come from:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<tag>_prof_head -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 --no-secondary --no-pruned > gpurun_out/<tag>_prof_head_bench.json
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<tag>_prof_final -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/<tag>_prof_final_bench.json
  python3 bench.py > gpurun_out/<tag>_bench_default.json
"""
import csv, glob, json, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def last_json(path):
    return json.loads([l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1])


def stats_rows(d):
    f = (glob.glob(f"gpurun_out/{tag}_{d}/run_kernel_stats.csv") + glob.glob(f"gpurun_out/{tag}_{d}/*/*kernel_stats.csv"))[0]
    return list(csv.DictReader(open(f)))


rows = stats_rows("prof_head")
with open(f"profiles/{tag}_final_headline_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:6]:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
b = last_json(f"gpurun_out/{tag}_prof_head_bench.json")
json.dump(b, open(f"profiles/{tag}_final_headline_bench_under_rocprof.json", "w"))
r = rows[0]
open(f"profiles/{tag}_final_headline_summary.txt", "w").write(
    f"# {tag} final, headline only: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 --no-secondary --no-pruned\n"
    "# every launch of the kernel below is one step of the benchmark (10 M reads x 150 bp vs 2 kb)\n"
    f"{r['Name']}: calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.3f} min_ms={float(r['MinNs'])/1e6:.3f} max_ms={float(r['MaxNs'])/1e6:.3f} pct={r['Percentage']}\n"
    f"bench line of the same run: kernel_ms (HIP events inside bench.py) = {b['roofline']['kernel_ms']:.3f}, ms_per_step = {b['ms_per_step']:.3f}, value = {b['value']:.4g} reads/s\n")
rows = stats_rows("prof_final")
with open(f"profiles/{tag}_final_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:40]:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
out = [f"# {tag} final: rocprofv3 --kernel-trace --stats on `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (one MI355X).",
       "# Headline launches = score_kernel_v2<4,38,0> (10 M reads x 150 bp vs 2 kb: warm-up + 3 timed; its average also contains the",
       f"# 2,048-read parity-check launch and the 1 M-read launches of the `secondary` section, see {tag}_final_headline_summary.txt for the clean figure);",
       "# the other kernels belong to `secondary` (10 M reads exact align; 1 M reads each: 3-pass align, ranges, filter, mixed lengths on side streams)",
       "# and to the runs with the opt-in column-pruned first pass (prune_strip_kernel / prune_window_kernel + score_kernel_v2 on the rescore lists):",
       "# the headline workload again (`exact_pruning`) and the `with_pruned_first_pass` repeats of the secondary entries.", ""]
for r in rows[:40]:
    out.append(f"{r['Name'][:84]:84s} calls={r['Calls']:>3s} avg_ms={float(r['AverageNs'])/1e6:9.3f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={r['Percentage']}")
b = last_json(f"gpurun_out/{tag}_prof_final_bench.json")
out += ["", "bench line of the same run: value=%.4g reads/s, ms_per_step=%.2f, roofline.kernel_ms=%.2f" % (b["value"], b["ms_per_step"], b["roofline"]["kernel_ms"])]
open(f"profiles/{tag}_final_summary.txt", "w").write("\n".join(out) + "\n")
d = last_json(f"gpurun_out/{tag}_bench_default.json")
json.dump(d, open(f"profiles/{tag}_bench_default.json", "w"))
print("default bench:", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("cpu_baseline", {}).get("value"))
print({k: (v.get("reads_per_s_end_to_end_incl_d2h") or v.get("reads_per_s") or v.get("pairs_per_s")) for k, v in d["secondary"].items()})
