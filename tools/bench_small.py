import sys, time
sys.path.insert(0, ".")
import torch, zoe_amd
from zoe_amd import synth
ctx = zoe_amd.SwContext.get(0)
ref = synth.reference_host(2000)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
for n in (1000, 10_000, 100_000, 1_000_000):
    rb = synth.reads_device(ctx, ref, 0, n, 150)
    prof = zoe_amd.into_local_profile(rb, dna, -10, -1)
    for _ in range(3): prof.sw_score_from_i8(ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): prof.sw_score_from_i8(ref)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"n={n}: {dt*1e3:.3f} ms per call, {n/dt/1e6:.2f} M reads/s", flush=True)
