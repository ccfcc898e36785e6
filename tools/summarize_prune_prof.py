"""Condenses a tools/profile_prune.sh output directory into profiles/<tag>.txt: durations (rocprofv3 --kernel-trace --stats)
and PMC passes of the kernels of the column-pruned score pass, per launch, next to the full pass on the same reads."""
import collections, csv, glob, sys

src, tag = sys.argv[1], sys.argv[2]
n_reads = float(sys.argv[3]) if len(sys.argv) > 3 else 2e6
protein = len(sys.argv) > 4 and sys.argv[4] == "protein"  # tools/profile_protein.sh: the WIDE kernels on a 25-letter alphabet
want = ("prune_strip_kernel", "prune_window_kernel", "score_kernel_v2")
out = [f"# {tag}: rocprofv3 on `python3 tools/try_prune.py {int(n_reads)} --score-only` ({int(n_reads)} synthetic 150 bp reads vs 2 kb, sw_score_from_i8 w256:",
       "# the full pass (score_kernel_v2<4,38,0>, 4 launches) and the column-pruned pass (strip + window + score_kernel_v2 on the rescore list, 4 rounds) in one process)",
       "# passes: --kernel-trace --stats | --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE | --pmc FETCH_SIZE | --pmc WRITE_SIZE"]
if protein:
    n_call = n_reads
    out[:2] = [f"# {tag}: rocprofv3 on `python3 tools/bench_protein.py {int(n_reads)} 0.03` ({int(n_reads)} sequences of 150 residues, pieces of a 2,000-residue reference with 3 % of",
               "# the residues substituted + 2 % unrelated sequences, 25-letter BLOSUM-shaped matrix, sw_score_from_i8 w256: the column-pruned pass",
               "# (prune_strip_kernel<24,WIDE> + prune_window_kernel<24,4,32,0,WIDE> + score_kernel_v2<4,38,0,WIDE> on the rescore list, 4 calls of 3 rounds: the",
               "# first chip-full is the bail-out probe) and the full pass (score_kernel_v2<4,38,0,WIDE>, 4 launches) in one process).",
               "# A call of 1 M reads is TWO launches of the strip and window kernels (393,216 reads: the probe; 606,784: the rest), so the per-launch",
               "# averages below cover 500,000 reads and the per-read figures are computed with that."]
    n_call, n_reads = n_reads, n_reads / 2
for f in glob.glob(f"{src}/bench_stats.txt"):
    out += ["# " + l.strip() for l in open(f) if l.startswith(("n=", "pruned", "full pass", "identical", "---"))]
dur = {}
for f in glob.glob(f"{src}/stats/*/*kernel_stats.csv"):
    out.append("## kernel durations (score_kernel_v2: 4 full launches and 4 launches over the rescore list, see the trace split below)")
    for r in list(csv.DictReader(open(f)))[:8]:
        out.append(f"{r['Name'][:86]:86s} calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.3f} min_ms={float(r['MinNs'])/1e6:.3f} max_ms={float(r['MaxNs'])/1e6:.3f}")
        dur[r["Name"]] = float(r["AverageNs"]) / 1e9
for f in glob.glob(f"{src}/stats/*/*kernel_trace.csv"):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in want):
            d[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    out.append("## per-launch durations in ms (kernel trace)")
    for k, v in d.items():
        out.append(f"{k[:86]:86s} " + " ".join(f"{x:.2f}" for x in v))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for dd in ("pmc_a", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{src}/{dd}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(w in k for w in want):
                agg[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), v in agg.items():
            for c, x in v.items():
                vals[k][c].append(x)
out.append("## PMC per launch (score_kernel_v2: the large values are the full pass, the small ones the rescore launches)")
for k in sorted(vals):
    out.append(k[:100])
    for c in sorted(vals[k]):
        xs = vals[k][c]
        if "score_kernel_v2" in k:
            big = [x for x in xs if x > 0.3 * max(xs)]
            small = [x for x in xs if x <= 0.3 * max(xs)]
            out.append(f"    {c:18s} full pass {sum(big)/max(len(big),1):.4g}   rescore launch {sum(small)/max(len(small),1):.4g}")
        else:
            out.append(f"    {c:18s} {sum(xs)/len(xs):.4g}")
    v = {c: sum(x) / len(x) for c, x in vals[k].items()}
    t = next((dur[n] for n in dur if n[:60] == k[:60]), None)
    if "score_kernel_v2" not in k and t and "SQ_INSTS_VALU" in v:
        clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / t if v.get("GRBM_GUI_ACTIVE") else 2.4e9
        out.append(f"    -> {t*1e3:.2f} ms, held clock {clk/1e9:.2f} GHz; cycles per wave64 VALU instruction = 1024 SIMDs x t x clk / SQ_INSTS_VALU = {1024*t*clk/v['SQ_INSTS_VALU']:.2f} (issue roof 4.0)")
        out.append(f"    -> VALU wave-instructions per read = {v['SQ_INSTS_VALU']/n_reads:.0f}")
    if "score_kernel_v2" not in k and "WRITE_SIZE" in v:
        out.append(f"    -> WRITE_SIZE x 1 KiB = {v['WRITE_SIZE']*1024/1e9:.2f} GB per launch = {v['WRITE_SIZE']*1024/n_reads/1e3:.2f} kB per read")
    if "score_kernel_v2" not in k and "FETCH_SIZE" in v:
        out.append(f"    -> FETCH_SIZE x 1 KiB = {v['FETCH_SIZE']*1024/1e9:.2f} GB per launch = {v['FETCH_SIZE']*1024/n_reads/1e3:.2f} kB per read")
open(f"profiles/{tag}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
