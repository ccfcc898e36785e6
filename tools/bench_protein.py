"""Throughput with a 25-letter alphabet (BLOSUM-shaped weights): n random 150-residue sequences vs a 2,000-residue reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zoe_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
keys = b"ACDEFGHIKLMNPQRSTVWYBJZX*"
rng = np.random.default_rng(3)
w = rng.integers(-4, 3, size=(25, 25))
w = np.minimum(w, w.T)
np.fill_diagonal(w, rng.integers(4, 12, size=25))
mp = zoe_amd.ByteIndexMap.new(keys, b"X")
m = zoe_amd.WeightMatrix.new_custom(mp, w.astype(np.int8))
alpha = np.frombuffer(keys[:20], dtype=np.uint8)
ref = rng.choice(alpha, 2000).astype(np.uint8).tobytes()
reads = rng.choice(alpha, (n, 150)).astype(np.uint8)
rb = zoe_amd.ReadBatch.from_fixed(torch.from_numpy(reads.reshape(-1)).cuda(), 150)
prof = zoe_amd.into_local_profile(rb, m, -11, -1)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = prof.sw_score_from_i8(ref)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n / dt / 1e6:.3f} M reads/s, {n * 150 * 2000 / dt / 1e9:.1f} GCUPS, mean score {s.score.float().mean().item():.1f}", flush=True)
