"""Times the full-alignment path (config 3 shape): n synthetic 150 bp reads vs a 2 kb reference, sw_align_from_i8 (w256)."""
import sys, time
sys.path.insert(0, ".")
import torch
import zoe_amd
from zoe_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
three = len(sys.argv) > 3 and sys.argv[3] == "3pass"
ctx = zoe_amd.SwContext.get(0)
ref = synth.reference_host(2000)
rb = synth.reads_device(ctx, ref, 0, n, 150)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
prof = zoe_amd.into_local_profile(rb, dna, -10, -1)
for r in range(reps + 1):
    ctx.timing_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = (prof.sw_align_from_i8_3pass if three else prof.sw_align_from_i8)(zoe_amd.SeqSrc.Reference(ref))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks, kl = ctx.timing_read()
    print(f"rep {r}: {n/dt/1e6:.2f} M reads/s end-to-end ({dt*1e3:.0f} ms), pass-2 kernels {ks*1e3:.0f} ms over {kl} timed regions, "
          f"ciglets {len(a.inc)}, tiers {dict(zip(*__import__('numpy').unique(a.tier, return_counts=True)))}", flush=True)
