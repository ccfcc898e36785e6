"""GPU parity tests of the column-pruned first pass for 8-32 letter alphabets (prune_strip_kernel<., WIDE> +
prune_window_kernel<., WIDE>, zsw_score_prune.hip): the default first pass of amino-acid matrices (the reference's BLOSUM /
PAM tables are WeightMatrix<i8, 25>, src/data/matrices/aa.rs; sw_simd_score is generic in S, striped.rs:65-70).

The bounds add each remaining query column's own potential max(0, max_x w[x][q_c]) (tests/models/prune_bounds.cpp states them
with plain integers against the full Gotoh matrix). Whatever the checks decide, the results must be the full pass's: a read
that fails one is rescored over all its cells. Checker: oracle/ on every case, the full GPU pass on the large batch.
"""
import contextlib

import numpy as np
import pytest

from conftest import stable_seed

pytestmark = pytest.mark.gpu

S_ = 0
KEYS = b"ACDEFGHIKLMNPQRSTVWYBJZX*"


@pytest.fixture(scope="module")
def za():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    import zoe_amd

    return zoe_amd


@pytest.fixture
def any_size(za):
    """the pruned pass for batches of every size (by default batches under 98,304 reads take the full pass)"""
    from zoe_amd import _lib

    ctx = za.SwContext.get(0)
    ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE)
    yield ctx
    ctx.debug_set(0)


@contextlib.contextmanager
def full_pass(ctx):
    from zoe_amd import _lib

    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    try:
        yield
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)


def _matrix(za, seed, S=25, symmetric=True):
    """BLOSUM-shaped: identities 4..11, substitutions -4..2, X and * score -1 / -4 against everything"""
    rng = np.random.default_rng(seed)
    keys = KEYS[:S] if S <= 25 else KEYS + bytes(range(0x61, 0x61 + S - 25))
    w = rng.integers(-4, 3, size=(S, S))
    if symmetric:
        w = np.minimum(w, w.T)
    np.fill_diagonal(w, rng.integers(4, 12, size=S))
    if S == 25:
        w[23, :] = w[:, 23] = -1
        w[24, :] = w[:, 24] = -4
        w[24, 24] = 1
    mp = za.ByteIndexMap.new(keys, keys[min(23, S - 1):min(23, S - 1) + 1])
    return keys, mp, w.astype(np.int8), za.WeightMatrix.new_custom(mp, w.astype(np.int8))


def _reads(rng, ref, n, L, alpha, kinds=8):
    """reads of L residues: mutated pieces of the reference (the fast path), chimeras, long gaps, junk ends, overhangs, repeats of
    one residue, unrelated sequences (every check fails: the full pass)"""
    R = len(ref)
    refa = np.frombuffer(ref, dtype=np.uint8)
    out = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        kind = i % kinds
        r = rng.choice(alpha, L)
        if kind <= 2 and R > L + 4:
            p = int(rng.integers(0, R - L - 4))
            piece = list(refa[p:p + L + 4])
            for _ in range(int(rng.integers(0, L // 6 + 1))):
                piece[int(rng.integers(0, L))] = int(rng.choice(alpha))
            if kind == 2:  # an insertion and a deletion
                piece.insert(int(rng.integers(5, L - 5)), int(rng.choice(alpha)))
                del piece[int(rng.integers(5, L - 5))]
            r = np.array(piece[:L], dtype=np.uint8)
        elif kind == 3 and R > 3 * L:  # two pieces, far apart or a few rows apart
            p = int(rng.integers(0, R - 3 * L))
            cut = int(rng.integers(10, L - 10))
            gap = int(rng.integers(1, 2 * L))
            r = np.concatenate([refa[p:p + cut], refa[p + cut + gap:p + cut + gap + (L - cut)]])
        elif kind == 4 and R > L:  # junk at one end
            p = int(rng.integers(0, R - L))
            j = int(rng.integers(1, L - 1))
            if i & 8:
                r[j:] = refa[p + j:p + L]
            else:
                r[:L - j] = refa[p:p + L - j]
        elif kind == 5:  # hanging over an end of the reference
            k = int(rng.integers(1, min(L - 1, R)))
            if i & 8:
                r[:k] = refa[R - k:]
            else:
                r[L - k:] = refa[:k]
        elif kind == 6:
            r[:] = alpha[int(rng.integers(0, len(alpha)))]
        out[i] = r
    return out


def _batch(za, reads2d):
    import torch

    n, L = reads2d.shape
    return za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads2d).reshape(-1)).cuda(), L)


@pytest.mark.parametrize("L,R,seed", [(150, 2000, 3), (150, 2000, 8), (90, 700, 8), (152, 333, 3), (250, 1500, 8), (380, 2500, 3)])
def test_wide_pruned_pass_equals_oracle(za, oracle, any_size, L, R, seed):
    """score (w256 cascade), score + ends and ranges of 400 protein reads per case through the pruned pass, read by read against the
    oracle; some reads pass the checks and some do not (seed 8: an asymmetric matrix and free gap extension)"""
    keys, mp, w, m = _matrix(za, seed, symmetric=seed != 8)
    rng = np.random.default_rng(stable_seed(L, R, seed))
    alpha = np.frombuffer(keys[:20], dtype=np.uint8)
    ref = bytes(rng.choice(alpha, R))
    reads = _reads(rng, ref, 400, L, alpha)
    go, ge = (-11, -1) if seed == 3 else (-8, 0)
    sc = oracle.Scoring(w, mp.index_map, go, ge)
    rb = _batch(za, reads)
    lp = za.LocalProfilesBatch.new_with_w256(rb, m, go, ge)
    got = lp.sw_score_from_i8(ref)
    rescored = any_size.prune_rescored()
    assert 0 < rescored < 400, rescored  # both outcomes occurred
    rg = lp.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
    p16 = za.StripedProfileBatch(rb, m, go, ge, "i16", 16)
    ends = p16.sw_score_ends(za.SeqSrc.Reference(ref))
    for i in range(400):
        rd = reads[i].tobytes()
        o_st, o_s, o_t = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]) if o_st == S_ else 0, int(got.tier[i])) == (o_st, o_s if o_st == S_ else 0, o_t), i
        e_st, (e_s, e_r, e_q) = oracle.score_ends("i16", 16, sc, rd, ref)
        assert int(ends.status[i]) == e_st, i
        if e_st == S_:
            assert (int(ends.score[i]), int(ends.ref_end[i]), int(ends.query_end[i])) == (e_s, e_r, e_q), i
        if i % 4 == 0:
            r_st, r_s, r_rr, r_qr = oracle.cascade_score_ranges(8, 256, sc, rd, ref)[:4]
            if r_st == S_:
                assert (int(rg.score[i]), (int(rg.ref_start[i]), int(rg.ref_end[i])), (int(rg.query_start[i]), int(rg.query_end[i]))) == (r_s, r_rr, r_qr), i


def test_wide_pruned_alignments_equal_oracle(za, oracle, any_size):
    """sw_simd_align with the pruned first pass (score + the first row holding it): CIGARs against the oracle"""
    keys, mp, w, m = _matrix(za, 5)
    rng = np.random.default_rng(17)
    alpha = np.frombuffer(keys[:20], dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 900))
    reads = _reads(rng, ref, 200, 140, alpha)
    sc = oracle.Scoring(w, mp.index_map, -11, -1)
    al = za.StripedProfileBatch(_batch(za, reads), m, -11, -1, "i16", 16).sw_align(za.SeqSrc.Reference(ref))
    for i in range(200):
        want = oracle.align("i16", 16, sc, reads[i].tobytes(), ref)
        assert al.key(i) == (want.key() if want.status == S_ else (want.status, 0, (0, 0), (0, 0), "", 0, 0)), i


def test_wide_pruned_ragged_batch_equals_oracle(za, oracle, any_size):
    """a ragged batch: the length classes that fall into one pruning class share the strip launch"""
    keys, mp, w, m = _matrix(za, 9)
    rng = np.random.default_rng(23)
    alpha = np.frombuffer(keys[:20], dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 1200))
    reads = []
    for i in range(700):
        L = int(rng.integers(30, 420))
        reads.append(_reads(rng, ref, 8, L, alpha)[i % 8].tobytes())
    sc = oracle.Scoring(w, mp.index_map, -11, -1)
    got = za.LocalProfilesBatch.new_with_w256(reads, m, -11, -1).sw_score_from_i8(ref)
    for i, rd in enumerate(reads):
        o_st, o_s, o_t = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]) if o_st == S_ else 0, int(got.tier[i])) == (o_st, o_s if o_st == S_ else 0, o_t), i


def test_wide_pruned_is_the_default_for_large_batches_and_equals_the_full_pass(za, oracle):
    """300,000 protein reads (above the size threshold, no debug flag): the pruned pass runs, most reads pass its checks, and every
    score, status and tier equals the full pass's; a sample is checked against the oracle"""
    ctx = za.SwContext.get(0)
    keys, mp, w, m = _matrix(za, 3)
    rng = np.random.default_rng(31)
    alpha = np.frombuffer(keys[:20], dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 2000))
    base = _reads(rng, ref, 3000, 150, alpha, kinds=16)  # kinds 8..15: unrelated sequences; 0..2: mutated pieces
    n = 300_000
    reads = np.tile(base, (n // 3000, 1))
    # make the copies differ: a substitution at a random place of every read
    pos = rng.integers(0, 150, size=n)
    reads[np.arange(n), pos] = rng.choice(alpha, n)
    rb = _batch(za, reads)
    lp = za.LocalProfilesBatch.new_with_w256(rb, m, -11, -1)
    got = lp.sw_score_from_i8(ref)
    rescored = ctx.prune_rescored()
    assert 0 < rescored < n, rescored
    with full_pass(ctx):
        want = lp.sw_score_from_i8(ref)
    import torch

    assert torch.equal(got.score, want.score) and torch.equal(got.status, want.status) and torch.equal(got.tier, want.tier)
    sc = oracle.Scoring(w, mp.index_map, -11, -1)
    for i in range(0, n, 997):
        o_st, o_s, o_t = oracle.cascade_score(8, 256, sc, reads[i].tobytes(), ref)
        assert (int(got.status[i]), int(got.score[i]) if o_st == S_ else 0, int(got.tier[i])) == (o_st, o_s if o_st == S_ else 0, o_t), i


def test_diverged_reads_take_the_bail_out(za):
    """2 M unrelated protein sequences: the first chip-full goes through strip + window, nearly all of it is handed back, and the
    rest of the batch takes the full pass at once — same results, and the call costs about the full pass (measured 1.04x)"""
    import time

    import torch

    ctx = za.SwContext.get(0)
    keys, mp, w, m = _matrix(za, 3)
    rng = np.random.default_rng(41)
    alpha = np.frombuffer(keys[:20], dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 2000))
    n = 2_000_000
    reads = rng.choice(alpha, (n, 150)).astype(np.uint8)
    lp = za.LocalProfilesBatch.new_with_w256(_batch(za, reads), m, -11, -1)

    def timed():
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = lp.sw_score_from_i8(ref)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return out, best

    got, t_pruned = timed()
    assert ctx.prune_rescored() == n  # the probe's reads failed their checks, the others were never tried
    with full_pass(ctx):
        want, t_full = timed()
    assert torch.equal(got.score, want.score) and torch.equal(got.status, want.status) and torch.equal(got.tier, want.tier)
    assert t_pruned <= 1.15 * t_full, (t_pruned, t_full)
