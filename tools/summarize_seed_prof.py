"""Condenses a tools/profile_seed.sh output directory into profiles/<tag>.txt: per-kernel durations (rocprofv3 --kernel-trace
--stats) and PMC passes per launch, for every kernel that takes more than 0.2 % of the GPU time of the run."""
import collections, csv, glob, sys

src, tag = sys.argv[1], sys.argv[2]
n_reads = float(sys.argv[3]) if len(sys.argv) > 3 else 1e7
out = [f"# {tag}: rocprofv3 on `python3 tools/try_seed.py {int(n_reads)} ...` (full pass with ZSW_OPTION_EXACT_PRUNING 0, then the default seeded pass,",
       "# 4 timed calls each, score then score_ranges, in one process). Passes: --kernel-trace --stats | --pmc SQ_INSTS_VALU SQ_INSTS_SALU",
       "# SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE | --pmc FETCH_SIZE | --pmc WRITE_SIZE"]
for f in glob.glob(f"{src}/bench_stats.txt"):
    out += ["# " + l.strip() for l in open(f) if ": n=" in l]
dur, total = {}, 0.0
rows = []
for f in glob.glob(f"{src}/stats/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
keep = [r for r in rows if float(r["TotalDurationNs"]) > 0.002 * total]
out.append("## kernel durations")
for r in keep:
    out.append(f"{r['Name'][:96]:96s} calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.3f} min_ms={float(r['MinNs'])/1e6:.3f} max_ms={float(r['MaxNs'])/1e6:.3f} share={float(r['TotalDurationNs'])/total:.3f}")
    dur[r["Name"]] = float(r["AverageNs"]) / 1e9
names = [r["Name"] for r in keep]
trace = collections.defaultdict(list)
for f in glob.glob(f"{src}/stats/*/*kernel_trace.csv"):
    d = trace
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] in names:
            d[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    out.append("## per-launch durations in ms (kernel trace, launch order)")
    for k, v in d.items():
        out.append(f"{k[:96]:96s} " + " ".join(f"{x:.2f}" for x in v[:24]))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for dd in ("pmc_a", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{src}/{dd}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if k in names:
                agg[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), v in agg.items():
            for c, x in v.items():
                vals[k][c].append(x)
out.append("## PMC per launch (a kernel launched both over the whole batch and over the worklist: 'large' = launches above 30 % of the largest, 'small' = the rest)")
for k in sorted(vals):
    out.append(k[:110])
    for c in sorted(vals[k]):
        xs = vals[k][c]
        big = [x for x in xs if x > 0.3 * max(xs)]
        small = [x for x in xs if x <= 0.3 * max(xs)]
        line = f"    {c:18s} large {sum(big)/max(len(big),1):.4g} (x{len(big)})"
        if small:
            line += f"   small {sum(small)/len(small):.4g} (x{len(small)})"
        out.append(line)
    # derived figures per class of launches (durations from the kernel trace, split the same way)
    ds = trace.get(k, [])
    for name, pick in (("large", lambda x, m: x > 0.3 * m), ("small", lambda x, m: x <= 0.3 * m)):
        v = {}
        for c, xs in vals[k].items():
            sel = [x for x in xs if pick(x, max(xs))]
            if sel:
                v[c] = sum(sel) / len(sel)
        tsel = [x for x in ds if pick(x, max(ds))] if ds else []
        if not v or (name == "small" and not tsel):
            continue
        t = sum(tsel) / len(tsel) / 1e3 if tsel else None
        if t and "SQ_INSTS_VALU" in v and "score_kernel_v2" not in k:
            clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / t if v.get("GRBM_GUI_ACTIVE") else 2.4e9
            out.append(f"    -> {name}: avg {t*1e3:.3f} ms; cycles per wave64 VALU instruction = 1024 SIMDs x t x clk({clk/1e9:.2f} GHz) / SQ_INSTS_VALU = {1024*t*clk/v['SQ_INSTS_VALU']:.2f} (issue roof 4.0); VALU wave-instructions per read of the batch = {v['SQ_INSTS_VALU']/n_reads:.0f}")
        if "WRITE_SIZE" in v:
            out.append(f"    -> {name}: WRITE_SIZE x 1 KiB = {v['WRITE_SIZE']*1024/1e9:.3f} GB per launch = {v['WRITE_SIZE']*1024/n_reads:.1f} B per read of the batch")
        if "FETCH_SIZE" in v:
            out.append(f"    -> {name}: FETCH_SIZE x 1 KiB = {v['FETCH_SIZE']*1024/1e9:.3f} GB per launch = {v['FETCH_SIZE']*1024/n_reads:.1f} B per read of the batch (raw counter; x2 for wide streaming reads on gfx950, MI355X_MICROARCH.md)")
open(f"profiles/{tag}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
