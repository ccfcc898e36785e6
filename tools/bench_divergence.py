"""The default (seeded, banded) score pass against read divergence: reads/s, the share of reads handed back to the full pass,
and equality of every result with the full pass, for reads that are x % substituted pieces of the reference (+ x / 10 % indels,
+ 2 % unrelated reads).
usage: python tools/bench_divergence.py [n_reads] [rates, per cent, comma-separated] [--ends] [--noband] [--json]
The seeded pass proves a read's score with the k-mers the read shares with the reference: a path elsewhere loses at least
lambda (7 with 2 / -5, -10 / -1) per sampled 8-mer without an occurrence there, i.e. 0.7 per column. A read that is itself further
than that from the reference (one substitution per ten bases) cannot be told from such a path and is scored over all its cells:
the curve falls from the banded kernel's rate to the full pass's between 5 % and 12 %."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import zoe_amd
from zoe_amd import _lib, synth


def sweep(ctx, n, rates, ends=False, read_len=150, ref_len=2000, reps=3):
    ref = synth.reference_host(ref_len)
    dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
    out = []
    for rate in rates:
        rb = synth.diverged_reads_device(ctx, ref, n, read_len, rate / 100.0)
        prof = zoe_amd.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
        fn = (lambda: prof.sw_score_ranges_from_i8(zoe_amd.SeqSrc.Reference(ref))) if ends else (lambda: prof.sw_score_from_i8(ref))

        def timed():
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            return r, (time.perf_counter() - t0) / reps

        got, t = timed()
        back = ctx.prune_rescored()
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
        try:
            full, t_full = timed()
        finally:
            ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
        fields = [f for f in ("score", "status", "tier", "ref_start", "ref_end", "query_start", "query_end") if getattr(got, f, None) is not None]
        same = all(bool(torch.equal(getattr(got, f), getattr(full, f))) for f in fields)
        out.append({"substitutions_pct": rate, "reads_per_s": n / t, "ms": t * 1e3, "handed_back_fraction": back / n, "full_pass_reads_per_s": n / t_full,
                    "identical": same, "mean_score": float(got.score.float().mean())})
        del rb, prof, got, full
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 1_000_000
    rates = [float(x) for x in args[1].split(",")] if len(args) > 1 else [1, 2, 3, 5, 8, 12]
    if "--noband" in sys.argv:  # whole rows around the anchor (seed_window_kernel) instead of the banded kernel
        zoe_amd.SwContext.get(0).debug_set(_lib.DEBUG_SEED_NO_BAND)
    res = sweep(zoe_amd.SwContext.get(0), n, rates, ends="--ends" in sys.argv)
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        for r in res:
            print(f"{r['substitutions_pct']:5.1f} % substitutions: {r['reads_per_s']/1e6:7.1f} M reads/s ({r['ms']:.2f} ms), handed back {r['handed_back_fraction']:.3%}, "
                  f"full pass {r['full_pass_reads_per_s']/1e6:.1f} M reads/s, identical {r['identical']}, mean score {r['mean_score']:.1f}", flush=True)
    assert all(r["identical"] for r in res)
