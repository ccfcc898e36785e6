"""Sums rocprofv3 --pmc passes of tools/pmc_entry.py per kernel and per call -> profiles/<tag>.txt
usage: python tools/summarize_pmc_entry.py <tag> <dir>=<entry>:<n_reads> [...]
<dir> holds the SQ_INSTS_VALU pass; <dir>_fetch and <dir>_write, when present, the FETCH_SIZE and WRITE_SIZE passes (separate runs,
as MI355X_MICROARCH.md prescribes; KiB units; the raw counters are reported, without the x2 for wide streaming reads)."""
import collections, csv, glob, sys

CALLS = 3  # tools/pmc_entry.py
tag = sys.argv[1]
out = [f"# {tag}: rocprofv3 --pmc SQ_INSTS_VALU -- python3 tools/pmc_entry.py <entry> <n_reads> ({CALLS} identical calls per run; every kernel of the",
       "# process is counted, the synthetic-read generator excluded). VALU = wave64 VALU instructions issued (SQ_INSTS_VALU summed over the",
       "# dispatches' XCD rows). bench.py's secondary valu_roofline entries divide these per-read counts by the measured kernel time."]
for spec in sys.argv[2:]:
    d, rest = spec.split("=")
    entry, n = rest.split(":")
    n = float(n)
    per = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "SQ_INSTS_VALU" or "synth" in r["Kernel_Name"]:
                continue
            per[r["Kernel_Name"]] += float(r["Counter_Value"])
            launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
    total = sum(per.values())
    hbm = {}
    for cname, suffix in (("FETCH_SIZE", "_fetch"), ("WRITE_SIZE", "_write")):
        tot = 0.0
        found = False
        for f in glob.glob(f"{d}{suffix}/*/*counter_collection.csv"):
            found = True
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == cname and "synth" not in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"])
        if found:
            hbm[cname] = tot * 1024 / CALLS
    extra = "".join(f"; {c} {v / 1e9:.3f} GB per call = {v / n:.0f} B per read" for c, v in hbm.items())
    out.append(f"## {entry}: {int(n)} reads per call: {total / CALLS:.4g} VALU per call = {total / CALLS / n:.0f} per read{extra}")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
        if v > 0.005 * total:
            out.append(f"    {k[:120]:120s} launches/call={len(launches[k]) / CALLS:.1f} VALU/read={v / CALLS / n:.0f} share={v / total:.3f}")
open(f"profiles/{tag}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
