"""Condenses a tools/profile_align.sh output directory into profiles/<tag>.txt: kernel durations (rocprofv3 --kernel-trace
--stats) and the PMC passes of the pass-2 alignment kernels, per launch, with the derived per-read and per-cycle figures."""
import collections, csv, glob, sys

src, tag = sys.argv[1], sys.argv[2]
n_reads = float(sys.argv[3]) if len(sys.argv) > 3 else 1e6
out = [f"# {tag}: rocprofv3 on `python3 tools/bench_align.py {int(n_reads)} 1` ({int(n_reads)} synthetic 150 bp reads vs 2 kb, sw_align_from_i8 w256, one MI355X)",
       "# passes: --kernel-trace --stats | --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE |",
       "#         --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD | --pmc FETCH_SIZE | --pmc WRITE_SIZE"]
dur = {}
for f in glob.glob(f"{src}/stats/*/*kernel_stats.csv"):
    out.append("## kernel durations")
    for r in list(csv.DictReader(open(f)))[:6]:
        out.append(f"{r['Name'][:72]:72s} calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.3f} pct={r['Percentage']}")
        dur[r["Name"]] = float(r["AverageNs"]) / 1e9
vals = collections.defaultdict(dict)
for d in ("pmc_a", "pmc_b", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "align_kernel" in k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            for c, x in v.items():
                vals[k][c] = x / len(launches[k])
out.append("## PMC, mean per launch")
for k in sorted(vals, key=lambda k: -vals[k].get("SQ_INSTS_VALU", 0)):
    v = vals[k]
    out.append(k[:90])
    for c in sorted(v):
        out.append(f"    {c:24s} {v[c]:.4g}")
    t = next((dur[n] for n in dur if n[:40] == k[:40]), None)
    if "SQ_INSTS_VALU" in v and t:
        clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / t if "GRBM_GUI_ACTIVE" in v else 2.4e9
        out.append(f"    -> kernel {t*1e3:.2f} ms, held clock {clk/1e9:.2f} GHz (GRBM_GUI_ACTIVE / 8 / t)")
        out.append(f"    -> cycles per wave64 VALU instruction = 1024 SIMDs x t x clk / SQ_INSTS_VALU = {1024*t*clk/v['SQ_INSTS_VALU']:.2f}  (issue roof: 4.0)")
        out.append(f"    -> (VALU + SALU) x 4 cycles / (1024 SIMDs x t x clk) = {(v['SQ_INSTS_VALU']+v.get('SQ_INSTS_SALU',0))*4/(1024*t*clk):.2f}")
        if v["SQ_INSTS_VALU"] > 1e10:
            out.append(f"    -> VALU wave-instructions per read = {v['SQ_INSTS_VALU']/n_reads:.0f}")
    if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
        wc = v["SQ_WAVE_CYCLES"]
        out.append(f"    -> of the wave cycles: issuing {v['SQ_ACTIVE_INST_ANY']/wc:.0%}, waiting on s_waitcnt {v['SQ_WAIT_ANY']/wc:.0%}, issue-stalled {v['SQ_WAIT_INST_ANY']/wc:.0%}")
    if "FETCH_SIZE" in v:
        out.append(f"    -> FETCH_SIZE x 1 KiB = {v['FETCH_SIZE']*1024/1e9:.2f} GB read per launch (uncorrected: the traceback's byte loads are not the calibrated streaming pattern)")
    if "WRITE_SIZE" in v:
        out.append(f"    -> WRITE_SIZE x 1 KiB = {v['WRITE_SIZE']*1024/1e9:.2f} GB written per launch = {v['WRITE_SIZE']*1024/n_reads/1e3:.1f} kB per read (the flag ring)")
open(f"profiles/{tag}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
