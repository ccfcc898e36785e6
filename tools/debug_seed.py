"""Debug aid: outputs the seeded pass leaves unwritten (poisoned output buffers), tandem / gap_extend 0 case."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import zoe_amd
from zoe_amd import _lib, synth
from test_gpu_prune import _adversarial_reads
from conftest import stable_seed

ctx = zoe_amd.SwContext.get(0)
POISON = -559038737


def run(name, rb, matrix, go, ge, ref, n):
    prof = zoe_amd.LocalProfilesBatch.new_with_w256(rb, matrix, go, ge)
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    full = prof.sw_score_from_i8(ref)
    fs = full.score.clone()
    del full
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE)
    for rep in range(2):
        junk = [torch.full((max(n, 1),), POISON, dtype=torch.int32, device="cuda") for _ in range(8)]
        torch.cuda.synchronize()
        del junk
        got = prof.sw_score_from_i8(ref)
        resc = ctx.prune_rescored()
        g = got.score.clone()
        del got
        unwritten = (g == POISON).nonzero().flatten().cpu().numpy()
        bad = (g != fs).nonzero().flatten().cpu().numpy()
        print(f"{name} rep {rep}: n={n} handed back {resc}, unwritten {len(unwritten)} {unwritten[:20]}, mismatches {len(bad)} {bad[:20]}")
        for i in bad[:8]:
            print(f"    read {i}: full {int(fs[i])} seeded {int(g[i])}")
    ctx.debug_set(0)


scheme, kind = (3, -2, -4, 0), "tandem"
rng = np.random.default_rng(stable_seed(scheme, kind))
base = synth.reference_host(2000)
ref2 = bytes((base[:37] * 60)[:2000])
reads = _adversarial_reads(rng, ref2, 150)
m3 = zoe_amd.WeightMatrix.new_dna_matrix(3, -2, b"N")
for n in (len(reads), 256, 128, 64):
    rbt = zoe_amd.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads[:n]).reshape(-1)).cuda(), 150)
    run("tandem ge=0", rbt, m3, -4, 0, ref2, n)
n = 6000
ref = synth.reference_host(30000)
bases, off = synth.reads_ragged_host(ref, 5, n, 75, 400)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
rb = zoe_amd.ReadBatch(torch.from_numpy(bases).cuda(), n, offsets=torch.from_numpy(off.astype(np.int64)).cuda())
run("ragged 30kb", rb, dna, -10, -1, ref, n)
