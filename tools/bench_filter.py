"""Throughput of the sneaky_snake filter: n synthetic 150 bp reads, each against the 150 bp reference window it was drawn near
(diagonal of its score-ranges start), threshold 5 %.  usage: python tools/bench_filter.py [n_reads] [threshold]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import zoe_amd
from zoe_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
R, L = 2000, 150
ref = synth.reference_host(R)
ctx = zoe_amd.SwContext.get(0)
ctx.set_reference(ref)
reads = synth.reads_device(ctx, ref, 0, n, L)
m = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
rg = zoe_amd.StripedProfileBatch(reads, m, -10, -1, T="i16", N=16).sw_score_ranges(zoe_amd.SeqSrc.Reference(ref))
st = (rg.ref_start.to(torch.int64) - rg.query_start.to(torch.int64)).clamp(0, R - L).to(torch.int32)
ln = torch.full((n,), L, dtype=torch.int32, device=st.device)
ctx.timing_enable(True)
for rep in range(3):
    torch.cuda.synchronize()
    ctx.timing_read()
    t0 = time.perf_counter()
    out = zoe_amd.sneaky_snake(ref, reads, st, ln, thr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks, _ = ctx.timing_read()
print(f"{n} pairs, threshold {thr}: {n / dt / 1e6:.1f} M pairs/s wall, kernel {ks * 1e3:.2f} ms = {n / ks / 1e6:.1f} M pairs/s; "
      f"pass {(out == 1).float().mean().item():.3f} reject {(out == 0).float().mean().item():.3f}")
