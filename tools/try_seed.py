"""Seeded exact pass vs the full pass on synthetic reads: identical results, kernel times, handed-back fraction.
usage: python tools/try_seed.py [n_reads] [read_len] [ref_len] [--mixed] [--noband] [--seeded-first]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import zoe_amd
from zoe_amd import _lib, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 1_000_000
L = int(args[1]) if len(args) > 1 else 150
R = int(args[2]) if len(args) > 2 else 2000
mixed = "--mixed" in sys.argv
ctx = zoe_amd.SwContext.get(0)
if "--noband" in sys.argv:  # score-only calls through seed_window_kernel instead of seed_band_kernel
    ctx.debug_set(_lib.DEBUG_SEED_NO_BAND)
ref = synth.reference_host(R)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
rb = synth.reads_ragged_device(ctx, ref, 0, n, 75, 400) if mixed else synth.reads_device(ctx, ref, 0, n, L)
prof = zoe_amd.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)


def run(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ks, launches = ctx.timing_read()
    ws, wl = ctx.timing_read_window()
    ctx.timing_enable(False)
    return r, dt, ks / max(launches, 1), ws / max(reps, 1)


for name, fn in (("score", lambda: prof.sw_score_from_i8(ref)), ("ranges", lambda: prof.sw_score_ranges_from_i8(zoe_amd.SeqSrc.Reference(ref)))):
    if "--seeded-first" in sys.argv:  # the order bench.py runs them in
        got, t_got, k_got, w_got = run(fn)
        resc = ctx.prune_rescored()
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    full, t_full, k_full, _ = run(fn)
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    if "--seeded-first" not in sys.argv:
        got, t_got, k_got, w_got = run(fn)
        resc = ctx.prune_rescored()
    fields = [f for f in ("score", "status", "tier", "ref_start", "ref_end", "query_start", "query_end") if getattr(got, f, None) is not None]
    same = all(bool(torch.equal(getattr(got, f), getattr(full, f))) for f in fields)
    print(f"{name}: n={n} L={'75-400' if mixed else L} R={R}: full {t_full*1e3:.2f} ms (kernels {k_full*1e3:.2f}), seeded {t_got*1e3:.2f} ms "
          f"(kernels {k_got*1e3:.2f}, window/band {w_got*1e3:.2f}) -> {n/t_got/1e6:.1f} M reads/s, handed back {resc} ({resc/n:.3%}), identical {same}", flush=True)
    assert same
