"""Config 5 timing: n reads of 75-400 bp vs a 30 kb reference (bucketed launches), score-only."""
import sys, time
sys.path.insert(0, ".")
import torch
import zoe_amd
from zoe_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = zoe_amd.SwContext.get(0)
ref = synth.reference_host(30000)
rb = synth.reads_ragged_device(ctx, ref, 0, n, 75, 400)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
prof = zoe_amd.into_local_profile(rb, dna, -10, -1)
total_bases = int(rb.offsets[-1])
for r in range(3):
    ctx.timing_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = prof.sw_score_from_i8(ref)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks, kl = ctx.timing_read()
    print(f"rep {r}: {n/dt/1e6:.3f} M reads/s ({dt*1e3:.0f} ms; kernels {ks*1e3:.0f} ms), {total_bases*30000/dt/1e12:.2f} TCUPS, "
          f"mean len {total_bases/n:.1f}", flush=True)
