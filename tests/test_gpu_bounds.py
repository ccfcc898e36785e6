"""The banded seeded pass's own values against the host model's (tests/models/seed_band.cpp built as a library).

Result parity (every other -m gpu test) cannot see a bound that is too LOW until an input happens to exploit it. Here the kernel
reports, per read, the three numbers its decision rests on — the band's maximum in the doubled domain (odd = a path through a
cell outside the band may hold it), and the most a path that ends above / below the band can score — together with the geometry
it walked (zsw_debug_band_records), and the model, which is checked cell by cell against the full Gotoh matrix in the CPU suite
(tests/test_align_models.py), recomputes them from the read and that geometry. EQUALITY of every value on every read is
required: synthetic reads, diverged reads (3-12 % substitutions + indels), the adversarial sets of test_gpu_prune.py, ragged
pairs that share a lane, both tiers, score-only (tag -1) and ends (tag +1) walks."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, stable_seed

pytestmark = pytest.mark.gpu

SEED_DN, SEED_DM, SEED_TOL = 4, 4, 8  # zsw_score_seed.hpp


@pytest.fixture(scope="module")
def za():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    import zoe_amd

    return zoe_amd


@pytest.fixture(scope="module")
def model():
    d = tempfile.mkdtemp(prefix="zsw_band_model_")
    so = os.path.join(d, "libseed_band_model.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DZSW_MODEL_LIB", "-Wno-unknown-pragmas", "-o", so,
                    os.path.join(ROOT, "tests", "models", "seed_band.cpp")], check=True)
    lib = C.CDLL(so)
    lib.zsw_model_band.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 11 + [C.c_void_p]
    lib.zsw_model_band.restype = C.c_int
    return lib


def diverged_reads(ref: bytes, n: int, length: int, sub_pm: int, seed: int) -> np.ndarray:
    """copies of pieces of the reference with sub_pm per mille substitutions and a tenth of that in single-base indels"""
    rng = np.random.default_rng(seed)
    r = np.frombuffer(ref, dtype=np.uint8)
    out = np.empty((n, length), dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n):
        pos = int(rng.integers(0, len(r) - length - 8))
        q = []
        while len(q) < length:
            b = r[pos] if pos < len(r) else acgt[rng.integers(0, 4)]
            e = int(rng.integers(0, 1000))
            if e < sub_pm:
                b = acgt[(int(np.where(acgt == b)[0][0]) + int(rng.integers(1, 4))) % 4] if b in acgt else acgt[rng.integers(0, 4)]
            elif e < sub_pm + sub_pm // 20:
                pos += 1
                continue
            elif e < sub_pm + sub_pm // 10:
                q.append(acgt[rng.integers(0, 4)])
                continue
            q.append(b)
            pos += 1
        out[i] = q[:length]
    return out


def _check(za, model, matrix, go, ge, ref: bytes, reads, mode: str, min_walked: float):
    """reads: 2-D uint8 array (fixed length) or list of bytes (ragged). Runs the call with the records on, then the model per read."""
    import torch

    ctx = za.SwContext.get(0)
    if isinstance(reads, np.ndarray):
        n = reads.shape[0]
        batch = za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda(), reads.shape[1])
        rows = [reads[i] for i in range(n)]
    else:
        n = len(reads)
        batch = za.ReadBatch.from_sequences(reads)
        rows = [np.frombuffer(x, dtype=np.uint8) for x in reads]
    rec = torch.full((n, 8), -7, dtype=torch.int32, device="cuda")
    prof = za.LocalProfilesBatch.new_with_w256(batch, matrix, go, ge)
    ctx.debug_band_records(rec)
    try:
        if mode == "score":
            prof.sw_score_from_i8(ref)
        else:
            prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
        torch.cuda.synchronize()
    finally:
        ctx.debug_band_records(None)
    rec = rec.cpu().numpy()
    walked = rec[:, 5] != -7
    assert walked.mean() >= min_walked, f"only {walked.mean():.3f} of the reads were walked by the banded kernel"
    w = np.ascontiguousarray(matrix.signed_weights().astype(np.int32))
    S = w.shape[0]
    idx = matrix.mapping.index_map
    ref_idx = np.ascontiguousarray(idx[np.frombuffer(ref, dtype=np.uint8)])
    K = 8
    while K < 12 and (1 << (2 * K)) < 32 * len(ref):
        K += 1
    tag = -1 if mode == "score" else 1
    out = np.zeros(8, dtype=np.int32)
    bad = []
    accepted = 0
    for i in np.nonzero(walked)[0]:
        q = np.ascontiguousarray(idx[rows[i]])
        g = rec[i]
        n_strips, wu, wd = int(g[5] & 0xff), int((g[5] >> 8) & 0xfff), int((g[5] >> 20) & 0xfff)
        rc = model.zsw_model_band(w.ctypes.data, S, -go, -ge, ref_idx.ctypes.data, len(ref_idx), q.ctypes.data, len(q), K, SEED_DN, SEED_DM, SEED_TOL, int(g[7]) >> 8,
                                  n_strips, wu, wd, int(g[3]), int(g[4]), tag, out.ctypes.data)
        assert rc == 0 and out[0] == 1, f"read {i}: the model finds no anchor where the kernel walked a band"
        assert out[1] == g[6], f"read {i}: anchor diagonal {g[6]} (kernel) vs {out[1]} (model)"
        got = (int(g[0]), max(int(g[1]), 0), max(int(g[2]), 0))
        want = (int(out[2]), max(int(out[3]), 0), max(int(out[4]), 0))
        if got != want:
            bad.append((int(i), got, want, (n_strips, wu, wd, int(g[3]), int(g[4]))))
        accepted += int(g[7]) & 1
    assert not bad, f"{len(bad)} reads differ, first: {bad[:5]} (maximum2, oa, ob) kernel vs model"
    return int(walked.sum()), accepted


SCHEMES = [(2, -5, -10, -1), (1, -3, -5, -2), (3, -2, -5, 0), (2, -10, -10, -1)]


@pytest.mark.parametrize("mode", ["score", "ranges"])
@pytest.mark.parametrize("scheme", SCHEMES)
def test_kernel_values_equal_the_model_on_diverged_reads(za, model, mode, scheme):
    """30,000 reads per scheme and mode: the bench's synthetic set and copies with 3 / 5 / 8 / 12 % substitutions (+ indels); both
    tiers run (a read that fails the narrow band reports the wide band's values)."""
    from zoe_amd import synth

    ma, mi, go, ge = scheme
    ref = synth.reference_host(2000)
    parts = [synth.reads_host(ref, 11, 6000, 150)] + [diverged_reads(ref, 6000, 150, r, stable_seed("bounds", scheme, r)) for r in (30, 50, 80, 120)]
    reads = np.concatenate(parts)
    m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
    walked, accepted = _check(za, model, m, go, ge, ref, reads, mode, 0.9)
    assert accepted > 0.2 * walked  # (not vacuous: a good share of the walks end in an accepted read)


@pytest.mark.parametrize("mode", ["score", "ranges"])
def test_kernel_values_equal_the_model_on_adversarial_and_ragged_reads(za, model, mode):
    """Reads built to break the bounds (second copies elsewhere, tandem repeats, chimeras, long gaps, N runs, junk ends, reads over
    the reference's ends) and ragged batches whose lane partners differ in length and anchor."""
    from zoe_amd import synth

    rng = np.random.default_rng(stable_seed("bounds-adv", mode))
    ref = bytearray(synth.reference_host(2500))
    ref[700:760] = ref[300:360]          # a second copy
    for i in range(1200 + 3, 1260):      # a tandem repeat
        ref[i] = ref[i - 3]
    ref[1800:1806] = b"NNNNNN"
    ref = bytes(ref)
    r = np.frombuffer(ref, dtype=np.uint8)
    reads = []
    acgt = b"ACGT"
    for i in range(12000):
        L = int(rng.integers(60, 260))
        kind = i % 8
        pos = int(rng.integers(0, len(r) - L))
        q = bytearray(r[pos : pos + L].tobytes())
        if kind == 1:      # chimera
            p2 = int(rng.integers(0, len(r) - L))
            cut = int(rng.integers(20, L - 20))
            q[cut:] = r[p2 + cut : p2 + L].tobytes()
        elif kind == 2:    # a long deletion from the read
            cut, g = int(rng.integers(20, L - 20)), int(rng.integers(3, 30))
            q = q[:cut] + bytearray(r[min(pos + cut + g, len(r) - 1) : pos + L + g].tobytes())
        elif kind == 3:    # a long insertion
            cut, g = int(rng.integers(20, L - 20)), int(rng.integers(3, 30))
            q = q[:cut] + bytearray(int(acgt[x]) for x in rng.integers(0, 4, g)) + q[cut:]
        elif kind == 4:    # junk ends
            a, b = int(rng.integers(0, 25)), int(rng.integers(0, 25))
            q[:a] = bytes(int(acgt[x]) for x in rng.integers(0, 4, a))
            if b:
                q[-b:] = bytes(int(acgt[x]) for x in rng.integers(0, 4, b))
        elif kind == 5:    # N in the read, in runs between where k-mers are sampled too
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, L - 4))
                q[at : at + int(rng.integers(1, 4))] = b"NNN"[: min(3, L - at)][: int(rng.integers(1, 4))]
        elif kind == 6:    # over an end of the reference
            over = int(rng.integers(5, 40))
            q = bytearray(int(acgt[x]) for x in rng.integers(0, 4, over)) + bytearray(r[: L - over].tobytes()) if i % 16 < 8 else \
                bytearray(r[len(r) - (L - over) :].tobytes()) + bytearray(int(acgt[x]) for x in rng.integers(0, 4, over))
        for j in range(len(q)):  # 2 % substitutions everywhere
            if rng.integers(0, 50) == 0:
                q[j] = acgt[int(rng.integers(0, 4))]
        reads.append(bytes(q[: max(24, len(q))]))
    m = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    walked, accepted = _check(za, model, m, -10, -1, ref, reads, mode, 0.5)
    assert accepted > 0.2 * walked
