"""Large-batch equality sweep: the column-pruned first pass against the full pass (scores, and score + ranges), 1 M reads per
case, over read lengths, reference lengths and seeds.  usage: python tools/sweep_prune.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zoe_amd
from zoe_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = zoe_amd.SwContext.get(0)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
bad = 0
cases = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else None
for seed, (L, R) in enumerate([(80, 500), (100, 2000), (125, 5000), (150, 2000), (152, 1999), (200, 3000), (250, 10000), (304, 2000), (400, 4000), (330, 30000)]):
    if cases is not None and seed not in cases:
        continue
    ref = synth.reference_host(R, seed=1000 + seed)
    rb = synth.reads_device(ctx, ref, 7 * seed, n, L, seed=2000 + seed)
    prof = zoe_amd.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    sp = zoe_amd.StripedProfileBatch(rb, dna, -10, -1, "i16", 16)
    seq = zoe_amd.SeqSrc.Reference(ref)
    res = {}
    for name, flags in (("full", 0), ("pruned", _lib.DEBUG_SCORE_PRUNE)):
        ctx.debug_set(flags)
        prof.sw_score_from_i8(ref)  # warm-up: the pruned pass allocates its workspace with the first call for a reference length
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = prof.sw_score_from_i8(ref)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        resc = ctx.prune_rescored()
        r = sp.sw_score_ranges(seq)
        torch.cuda.synchronize()
        res[name] = (s, r, t1 - t0, resc)
    ctx.debug_set(0)
    (s0, r0, t0, _), (s1, r1, t1, resc) = res["full"], res["pruned"]
    same = all(torch.equal(getattr(s0, f), getattr(s1, f)) for f in ("score", "status", "tier")) and \
        all(torch.equal(getattr(r0, f), getattr(r1, f)) for f in ("score", "status", "ref_start", "ref_end", "query_start", "query_end"))
    bad += not same
    print(f"L={L} R={R}: identical={same}  score full {t0*1e3:.1f} ms, pruned {t1*1e3:.1f} ms, rescored {resc/n:.2%}", flush=True)
print("SWEEP", "OK" if not bad else f"FAILED ({bad} cases)")
sys.exit(1 if bad else 0)
