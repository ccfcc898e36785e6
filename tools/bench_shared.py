"""The one-profile-many-sequences role (zsw_*_shared_batch): n synthetic 150 bp reads against the profile of the 2 kb reference.
Times sw_score / sw_score_ends / sw_score_ranges (and sw_align on a tenth of the reads); the scores are compared with the
read-as-profile role (the score of a pair does not depend on the roles for a symmetric matrix).
usage: bench_shared.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zoe_amd
from zoe_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = zoe_amd.SwContext.get(0)
ref = synth.reference_host(2000)
rb = synth.reads_device(ctx, ref, 0, n, 150)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
prof = zoe_amd.SharedProfilesBatch.new_with_w256(ref, dna, -10, -1)
local = zoe_amd.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1).sw_score_from_i8(ref)


def timed(f, reps=3):
    best, out = 1e9, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return out, best


s, t = timed(lambda: prof.sw_score_from_i8(rb))
print(f"sw_score_from_i8 (shared): {n / t / 1e6:.1f} M reads/s ({t * 1e3:.2f} ms); equal to the read-as-profile role: "
      f"{bool(torch.equal(s.score, local.score) and torch.equal(s.status, local.status))}", flush=True)
sp = zoe_amd.SharedStripedProfile(ref, dna, -10, -1, "i16", 16)
e, t = timed(lambda: sp.sw_score_ends(zoe_amd.SeqBatchSrc.Reference(rb)))
print(f"sw_score_ends (shared, i16x16): {n / t / 1e6:.1f} M reads/s ({t * 1e3:.2f} ms); scores equal: {bool(torch.equal(e.score, local.score))}", flush=True)
r, t = timed(lambda: prof.sw_score_ranges_from_i8(zoe_amd.SeqBatchSrc.Query(rb)))
print(f"sw_score_ranges_from_i8 (shared): {n / t / 1e6:.1f} M reads/s ({t * 1e3:.2f} ms)", flush=True)
m = max(n // 10, 1)
rb2 = synth.reads_device(ctx, ref, 0, m, 150)
a, t = timed(lambda: prof.sw_align_from_i8(zoe_amd.SeqBatchSrc.Query(rb2)), reps=2)
print(f"sw_align_from_i8 (shared, {m} reads): {m / t / 1e6:.2f} M reads/s ({t * 1e3:.2f} ms)", flush=True)
a3, t = timed(lambda: prof.sw_align_from_i8_3pass(zoe_amd.SeqBatchSrc.Query(rb)), reps=2)
print(f"sw_align_from_i8_3pass (shared, {n} reads, end to end incl. D2H of records and CIGARs): {n / t / 1e6:.2f} M reads/s ({t * 1e3:.2f} ms)", flush=True)
