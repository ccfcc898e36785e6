"""Rebuilds profiles/constants.json (the per-read VALU and HBM counts bench.py prices its rooflines with) and the summaries it cites from
a directory written by tools/refresh_constants.sh on the GPU box, and stamps every section with the hash of the library they were
measured on (zoe_amd/build.py::fatbin_sha256): bench.py marks a figure derived from another library's counts "stale".

usage: python tools/refresh_constants.py gpurun_out/r04_const r04
"""
import collections, csv, glob, json, subprocess, sys

src, tag = sys.argv[1], sys.argv[2]
lib_hash = open(f"{src}/fatbin_sha256.txt").read().strip()
KIB = 1024.0


def counters(d, name):
    """{kernel: [value per dispatch, XCD rows summed]} of one --pmc pass, the synthetic-read generator left out."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "synth" not in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in agg.items()}


def total(d, name, pick=lambda k: True, small_only=()):
    """Sum over the dispatches of the kernels `pick` keeps; for kernels named in small_only, the launches over a worklist only
    (below 30 % of the kernel's largest launch: the same kernel also runs the full pass over the whole batch in that process)."""
    t = 0.0
    for k, xs in counters(d, name).items():
        if not pick(k):
            continue
        if any(s in k for s in small_only):
            xs = [x for x in xs if x <= 0.3 * max(xs)]
        t += sum(xs)
    return t


# --- the banded kernel of the headline: tools/try_seed.py 10 M = 4 seeded score calls (mode 0) + 4 seeded ranges calls (other modes)
N_BAND, CALLS_BAND = 1e7, 4
band0 = lambda k: "seed_band_kernel<" in k and k.split("seed_band_kernel<")[1].split(">")[0].replace(" ", "").endswith(",0")
band_valu = total(f"{src}/band/pmc_a", "SQ_INSTS_VALU", band0) / CALLS_BAND / N_BAND
band_fetch = total(f"{src}/band/pmc_fetch", "FETCH_SIZE", band0) * KIB / CALLS_BAND / N_BAND
band_write = total(f"{src}/band/pmc_write", "WRITE_SIZE", band0) * KIB / CALLS_BAND / N_BAND
names = sorted("seed_band_kernel<" + k.split("seed_band_kernel<")[1].split(">")[0].replace(" ", "") + ">" for k in counters(f"{src}/band/pmc_a", "SQ_INSTS_VALU") if band0(k))
subprocess.check_call([sys.executable, "tools/summarize_seed_prof.py", f"{src}/band", f"{tag}_band_summary", str(int(N_BAND))])
for f in glob.glob(f"{src}/band/stats/*/*kernel_stats.csv"):  # the rocprofv3 --kernel-trace --stats table itself, kernels above 0.1 %
    rows = list(csv.DictReader(open(f)))
    with open(f"profiles/{tag}_band_kernel_stats.csv", "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if float(r["Percentage"]) >= 0.1:
                w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

# --- secondary entry points: tools/pmc_entry.py, 3 identical calls of 1 M reads
N_SEC, CALLS_SEC = 1e6, 3
sec = {}
for e, key in (("ranges", "ranges"), ("3pass", "threepass"), ("mixed", "mixed"), ("align", "align")):
    d = f"{src}/pmc_{e}"
    hbm = (total(d + "_fetch", "FETCH_SIZE") + total(d + "_write", "WRITE_SIZE")) * KIB / CALLS_SEC / N_SEC
    sec[f"{key}_hbm_per_read"] = round(hbm)
    if e == "align":  # pass 2 alone: what bench.py's pass2_kernel_ms times
        sec["align_pass2_valu_per_read"] = round(total(d, "SQ_INSTS_VALU", lambda k: "align_kernel_pk" in k) / CALLS_SEC / N_SEC)
    else:
        sec[f"{key}_valu_per_read"] = round(total(d, "SQ_INSTS_VALU") / CALLS_SEC / N_SEC)
subprocess.check_call([sys.executable, "tools/summarize_pmc_entry.py", f"{tag}_secondary_valu"]
                      + [f"{src}/pmc_{e}={e}:{int(N_SEC)}" for e in ("ranges", "3pass", "mixed", "align")])

# --- 25-letter alphabet: tools/bench_protein.py 1 M, 4 pruned calls then 4 full-pass calls in one process
N_PROT, CALLS_PROT = 1e6, 4
pruned = lambda k: "prune_strip_kernel" in k or "prune_window_kernel" in k or "score_kernel_v2" in k
prot_valu = total(f"{src}/protein/pmc_a", "SQ_INSTS_VALU", pruned, small_only=("score_kernel_v2",)) / CALLS_PROT / N_PROT
prot_hbm = (total(f"{src}/protein/pmc_fetch", "FETCH_SIZE", pruned, small_only=("score_kernel_v2",))
            + total(f"{src}/protein/pmc_write", "WRITE_SIZE", pruned, small_only=("score_kernel_v2",))) * KIB / CALLS_PROT / N_PROT
subprocess.check_call([sys.executable, "tools/summarize_prune_prof.py", f"{src}/protein", f"{tag}_protein_prune_summary", str(int(N_PROT)), "protein"])

out = {
    "band": {"source": f"profiles/{tag}_band_summary.txt", "fatbin_sha256": lib_hash, "valu_per_read": round(band_valu),
             "hbm_bytes_per_read": round(band_fetch + band_write, 1), "kernels": names,
             "note": f"10 M reads of 150 bp vs 2 kb, every launch of the score-only banded kernels of one call (first tier over every read, second "
                     f"tier over the reads that fail it): {band_valu:.0f} VALU wave-instructions, {band_fetch:.0f} B fetched + {band_write:.0f} B written per read of the batch"},
    "secondary": dict({"source": f"profiles/{tag}_secondary_valu.txt", "fatbin_sha256": lib_hash}, **sec),
    "protein": {"source": f"profiles/{tag}_protein_prune_summary.txt", "fatbin_sha256": lib_hash, "valu_per_read": round(prot_valu), "hbm_per_read": round(prot_hbm)},
}
json.dump(out, open("profiles/constants.json", "w"), indent=1)
print(json.dumps(out, indent=1))
