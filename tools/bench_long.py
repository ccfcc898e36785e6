"""Score-only throughput for reads longer than the widest strip configuration: n reads of L bp vs a 30 kb reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zoe_amd
from zoe_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
no_tiles = len(sys.argv) > 3 and sys.argv[3] == "notiles"
ctx = zoe_amd.SwContext.get(0)
if no_tiles:
    from zoe_amd import _lib
    ctx.debug_set(_lib.DEBUG_NO_TILES)
ref = synth.reference_host(30000)
rb = synth.reads_device(ctx, ref, 0, n, L)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
prof = zoe_amd.into_local_profile(rb, dna, -10, -1)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = prof.sw_score_from_i16(ref)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n / dt:.0f} reads/s, {n * L * 30000 / dt / 1e12:.2f} TCUPS ({dt * 1e3:.0f} ms), mean score {s.score.float().mean().item():.0f}, "
          f"tiles {'off' if no_tiles else 'on'}", flush=True)
