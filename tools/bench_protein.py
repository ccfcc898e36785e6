"""Throughput with a 25-letter alphabet (BLOSUM-shaped weights): n 150-residue sequences vs a 2,000-residue reference.
Reads are pieces of the reference with `subs` substituted residues each (default 8 %) plus `junk` unrelated sequences (default
2 %); the default first pass (column-pruned: strip + window, zsw_score_prune.hip WIDE) against the full pass, results compared.
usage: bench_protein.py [n] [subs[,subs...]] [junk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zoe_amd
from zoe_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
subs_list = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.08]
junk = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
keys = b"ACDEFGHIKLMNPQRSTVWYBJZX*"
rng = np.random.default_rng(3)
w = rng.integers(-4, 3, size=(25, 25))
w = np.minimum(w, w.T)
np.fill_diagonal(w, rng.integers(4, 12, size=25))
mp = zoe_amd.ByteIndexMap.new(keys, b"X")
m = zoe_amd.WeightMatrix.new_custom(mp, w.astype(np.int8))
alpha = np.frombuffer(keys[:20], dtype=np.uint8)
refa = rng.choice(alpha, 2000).astype(np.uint8)
ref = refa.tobytes()
ctx = zoe_amd.SwContext.get(0)
ctx.timing_enable(True)
for subs in subs_list:
    print(f"--- {subs * 100:.0f} % substituted residues, {junk * 100:.0f} % unrelated sequences", flush=True)
    start = rng.integers(0, 2000 - 150, size=n)
    reads = refa[start[:, None] + np.arange(150)[None, :]]
    mut = rng.random((n, 150)) < subs
    reads = np.where(mut, rng.choice(alpha, (n, 150)), reads).astype(np.uint8)
    is_junk = rng.random(n) < junk
    reads[is_junk] = rng.choice(alpha, (int(is_junk.sum()), 150))
    rb = zoe_amd.ReadBatch.from_fixed(torch.from_numpy(reads.reshape(-1)).cuda(), 150)
    prof = zoe_amd.into_local_profile(rb, m, -11, -1)
    res = {}
    for name, opt in (("pruned (default)", 1), ("full pass", 0)):
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, opt)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s = prof.sw_score_from_i8(ref)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ks, kn = ctx.timing_read()
            if rep:
                best = min(best, dt)
        res[name] = s
        print(f"{name}: {n / best / 1e6:.2f} M reads/s ({best * 1e3:.2f} ms per {n} reads; kernels {ks / max(kn, 1) * 1e3:.2f} ms), "
              f"rescored over all cells {ctx.prune_rescored()} ({100.0 * ctx.prune_rescored() / n:.2f} %), mean score {s.score.float().mean().item():.1f}", flush=True)
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    a, b = res["pruned (default)"], res["full pass"]
    print("identical:", bool(torch.equal(a.score, b.score) and torch.equal(a.status, b.status) and torch.equal(a.tier, b.tier)))
