"""Condenses rocprofv3 output directories (gpurun_out/prof_r01/...) into the small files committed under profiles/."""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
out = [f"# {tag}: rocprofv3 on `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify 0` (10M x 150 bp reads vs 2 kb reference, one MI355X)"]
ks = glob.glob(f"{src}/stats/*/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:8]:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    out.append("## --kernel-trace --stats (top kernels)")
    for r in rows[:4]:
        out.append(f"{r['Name'][:70]:70s} calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.3f} pct={r['Percentage']}")
vals = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "score_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
out.append("## --pmc (separate passes), mean per launch of the score kernel")
for k, v in sorted(vals.items()):
    out.append(f"{k} = {v:.6g}")
if "FETCH_SIZE" in vals:
    out.append(f"HBM read  = FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-B requests as 64 B) = {vals['FETCH_SIZE']*1024*2/1e9:.3f} GB  (algorithmic: 1.500 GB of read bytes)")
if "WRITE_SIZE" in vals:
    out.append(f"HBM write = WRITE_SIZE KiB x 1024 = {vals['WRITE_SIZE']*1024/1e9:.3f} GB  (algorithmic: 0.060 GB = 10M x (4+1+1) B)")
bj = glob.glob(f"{src}/bench_stats.json")
if bj:
    try:
        d = json.loads(open(bj[0]).read().strip().splitlines()[-1])
        out.append(f"## bench line under the profiler: value={d['value']:.4g} reads/s, kernel_ms={d['roofline']['kernel_ms']:.2f}")
        if "SQ_INSTS_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
            ms = d['roofline']['kernel_ms']
            clk = vals["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3)
            out.append(f"held clock ~ GRBM_GUI_ACTIVE/8/kernel_time = {clk/1e9:.2f} GHz (PMC pass timing differs slightly)")
            out.append(f"cycles per wave64 VALU instruction = 1024 SIMDs x {ms:.1f} ms x clk / SQ_INSTS_VALU = {1024*ms*1e-3*clk/vals['SQ_INSTS_VALU']:.2f}")
    except Exception as e:
        out.append(f"(bench line not parsed: {e})")
open(f"profiles/{tag}_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
