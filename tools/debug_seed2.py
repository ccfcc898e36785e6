import sys, torch, numpy as np
sys.path.insert(0, ".")
import zoe_amd
from zoe_amd import _lib, synth
ctx = zoe_amd.SwContext.get(0)
ref = synth.reference_host(30000)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
n = 300000
for lo, hi in ((75, 400), (140, 152), (300, 304), (353, 400)):
    rb = synth.reads_ragged_device(ctx, ref, 0, n, lo, hi)
    sp = zoe_amd.StripedProfileBatch(rb, dna, -10, -1, T="i16", N=16)
    r0 = sp.sw_score(ref); c0 = ctx.prune_rescored()
    r2 = sp.sw_score_ends(zoe_amd.SeqSrc.Reference(ref)); c2 = ctx.prune_rescored()
    print(f"lens {lo}-{hi}: handed back score {c0} ({c0/n:.2%}), ends {c2} ({c2/n:.2%}), equal scores {bool(torch.equal(r0.score, r2.score))}")
