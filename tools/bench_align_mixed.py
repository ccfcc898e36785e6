"""Full alignment (sw_align_from_i8 / _3pass, w256) of n synthetic reads of 75-400 bp vs a 30 kb reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zoe_amd
from zoe_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ctx = zoe_amd.SwContext.get(0)
ref = synth.reference_host(30000)
rr = synth.reads_ragged_device(ctx, ref, 0, n, 75, 400)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
prof = zoe_amd.into_local_profile(rr, dna, -10, -1)
for name, fn in (("sw_align_from_i8", prof.sw_align_from_i8), ("sw_align_from_i8_3pass", prof.sw_align_from_i8_3pass)):
    for rep in range(3):
        ctx.timing_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = fn(zoe_amd.SeqSrc.Reference(ref))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ks, kl = ctx.timing_read()
        print(f"{name} rep {rep}: {n / dt / 1e6:.3f} M reads/s ({dt * 1e3:.0f} ms), timed kernels {ks * 1e3:.0f} ms, ciglets {len(a.inc)}", flush=True)
