"""Ad-hoc check of the column-pruned score pass against the default pass: parity and time (python tools/try_prune.py N)."""
import sys
import time

import numpy as np
import torch

import zoe_amd
from zoe_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ref = synth.reference_host(2000)
ctx = zoe_amd.SwContext.get(0)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
rb = synth.reads_device(ctx, ref, 0, n, 150)
prof = zoe_amd.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)


def run(flags, reps=3):
    ctx.debug_set(flags)
    out = prof.sw_score_from_i8(ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = prof.sw_score_from_i8(ref)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return out, dt


base, t_base = run(0)
pr, t_pr = run(_lib.DEBUG_SCORE_PRUNE)
rescored = ctx.prune_rescored()
ctx.debug_set(0)
same = bool((base.score == pr.score).all() and (base.status == pr.status).all() and (base.tier == pr.tier).all())
print(f"n={n}: default {t_base*1e3:.2f} ms ({n/t_base/1e6:.1f} M/s), pruned {t_pr*1e3:.2f} ms ({n/t_pr/1e6:.1f} M/s), identical={same}, rescored over all cells: {rescored} reads ({rescored/n:.2%})")
if not same:
    bad = torch.nonzero(base.score != pr.score).flatten()[:10].cpu().numpy()
    print("first differences (id, default, pruned):", [(int(i), int(base.score[i]), int(pr.score[i])) for i in bad], "of", int((base.score != pr.score).sum()))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if n <= 2_000_000 and "--score-only" not in sys.argv:
    seq = zoe_amd.SeqSrc.Reference(ref)
    direct = zoe_amd.StripedProfileBatch(rb, dna, -10, -1, "i16", 16)
    for name, fn in (("sw_score_ranges", lambda: direct.sw_score_ranges(seq)), ("sw_align_from_i8", lambda: prof.sw_align_from_i8(seq)),
                     ("sw_align_from_i8_3pass", lambda: prof.sw_align_from_i8_3pass(seq))):
        ctx.debug_set(0)
        t0 = timed(fn)
        ctx.debug_set(_lib.DEBUG_SCORE_PRUNE)
        t1 = timed(fn)
        ctx.debug_set(0)
        print(f"{name}: default {t0*1e3:.1f} ms ({n/t0/1e6:.1f} M/s), pruned first pass {t1*1e3:.1f} ms ({n/t1/1e6:.1f} M/s)")

if "--mixed" in sys.argv:
    ref30 = synth.reference_host(30000)
    rr = synth.reads_ragged_device(ctx, ref30, 0, 1_000_000, 75, 400)
    pm = zoe_amd.LocalProfilesBatch.new_with_w256(rr, dna, -10, -1)
    cells = float(rr.offsets[-1]) * 30000
    res = {}
    for name, flags in (("default", 0), ("pruned", _lib.DEBUG_SCORE_PRUNE)):
        ctx.debug_set(flags)
        t = timed(lambda: pm.sw_score_from_i8(ref30), reps=2)
        res[name] = pm.sw_score_from_i8(ref30)
        extra = f", rescored {ctx.prune_rescored()}" if flags else ""
        print(f"mixed 1M x 75-400 bp vs 30 kb, {name}: {t*1e3:.1f} ms, {1e6/t/1e6:.2f} M reads/s, {cells/t/1e12:.2f} TCUPS-equivalent{extra}")
    ctx.debug_set(0)
    print("identical:", bool(torch.equal(res["default"].score, res["pruned"].score) and torch.equal(res["default"].status, res["pruned"].status)))
