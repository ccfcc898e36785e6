"""GPU parity tests of the two exact pruned score passes: the seeded pass (zsw_score_seed.hip, the default) and round 2's
column-pruned pass (zsw_score_prune.hip, ZSW_DEBUG_PRUNE_STRIP).

A pruned pass must return exactly what the full pass returns — for every input, because a read without an anchor or whose
bound checks fail is rescored over all its cells. Checker: oracle/ (CPU restatement) on every case, plus the full GPU pass
(zsw_set_option(ZSW_OPTION_EXACT_PRUNING, 0)) on the large batches. The cases are built to hit both outcomes: reads that
pass the checks (the fast path) and reads that cannot (repeats, long gaps, chimeras, junk ends, low scores), and
`prune_rescored()` shows which happened.
"""
import contextlib

import numpy as np
import pytest

from conftest import stable_seed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def za():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    import zoe_amd

    return zoe_amd


@pytest.fixture(params=["seeded", "wide", "window", "strip"])
def pruned(za, request):
    """The context with a pruned pass switched on for batches of every size: the seeded pass (score-only calls: the banded
    kernel, zsw_score_band.hip — a narrow band first, the full band for the reads that fail in it), the same with the full band
    at once, the seeded pass with whole rows for score-only calls too (seed_window_kernel), or round 2's strip + window pass."""
    from zoe_amd import _lib

    ctx = za.SwContext.get(0)
    ctx.kind = request.param
    extra = {"seeded": 0, "wide": _lib.DEBUG_SEED_WIDE_BAND, "window": _lib.DEBUG_SEED_NO_BAND, "strip": _lib.DEBUG_PRUNE_STRIP}[request.param]
    ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE | extra)
    yield ctx
    ctx.debug_set(0)


@contextlib.contextmanager
def full_pass(ctx):
    """every cell of every read (the option off)"""
    from zoe_amd import _lib

    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    try:
        yield
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)


def _score(za, reads2d, matrix, go, ge, ref):
    import torch

    n, L = reads2d.shape
    rb = za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads2d).reshape(-1)).cuda(), L)
    out = za.LocalProfilesBatch.new_with_w256(rb, matrix, go, ge).sw_score_from_i8(ref)
    return out.score.cpu().numpy().view(np.uint32), out.status.cpu().numpy(), out.tier.cpu().numpy()


def _check(za, oracle, reads2d, matrix, go, ge, ref, ctx):
    sc = oracle.Scoring(matrix.signed_weights(), matrix.mapping.index_map, go, ge)
    want_s, want_st, want_tier = oracle.batch_score_w256(8, sc, reads2d, ref, fixed_len=reads2d.shape[1], threads=8)
    s, st, tier = _score(za, reads2d, matrix, go, ge, ref)
    rescored = ctx.prune_rescored()
    assert np.array_equal(st, want_st)
    assert np.array_equal(s, want_s)
    assert np.array_equal(tier, want_tier)
    return rescored


def _mutate(rng, seq, subs=0.01, indel=0.002):
    out = []
    for b in seq:
        u = rng.random()
        if u < indel:
            continue
        if u < 2 * indel:
            out.append(rng.choice(list(b"ACGT")))
        out.append(rng.choice(list(b"ACGT")) if rng.random() < subs else b)
    return out


def _fit(rng, seq, L):
    seq = list(seq)[:L]
    while len(seq) < L:
        seq.append(rng.choice(list(b"ACGT")))
    return np.array(seq, dtype=np.uint8)


@pytest.mark.parametrize("L", [65, 100, 150, 151, 152])
@pytest.mark.parametrize("n", [1, 2, 3, 513, 4001])
def test_pruned_pass_equals_oracle_on_synthetic_reads(za, oracle, pruned, L, n):
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    reads = synth.reads_host(ref, stable_seed(L, n) % 100000, n, L)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    rescored = _check(za, oracle, reads, dna, -10, -1, ref, pruned)
    if n >= 513:
        assert rescored < n // 2  # most synthetic reads pass the checks


def test_lengths_outside_the_pruned_range_take_the_full_pass(za, oracle, pruned):
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    for L in ((64, 401, 600) if pruned.kind == "strip" else (8, 23)):
        reads = synth.reads_host(ref, 5, 300, L)
        assert _check(za, oracle, reads, dna, -10, -1, ref, pruned) == 0


@pytest.mark.parametrize("R", [1, 10, 31, 32, 33, 150, 500, 2047, 2048])
def test_reference_lengths(za, oracle, pruned, R):
    from zoe_amd import synth

    ref = synth.reference_host(R)
    big = synth.reference_host(2000)
    reads = synth.reads_host(big, 9, 600, 150)  # sampled from another reference: partial and no hits
    if R >= 150:
        reads[:300] = synth.reads_host(ref, 9, 300, 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    _check(za, oracle, reads, dna, -10, -1, ref, pruned)


def _adversarial_reads(rng, ref, L):
    """Reads that stress each bound check: second copies elsewhere, gaps longer than the window margin, chimeras, junk at
    either end, reads hanging over the reference ends, low complexity, N runs, no hit at all."""
    R = len(ref)
    refl = list(ref)
    rows = []

    def sample(p, k):
        return refl[p : p + k]

    for _ in range(60):  # plain
        p = int(rng.integers(0, R - L))
        rows.append(_fit(rng, _mutate(rng, sample(p, L)), L))
    for _ in range(60):  # long deletion (reference bases skipped) / long insertion
        p = int(rng.integers(0, R - 2 * L))
        cut = int(rng.integers(20, L - 20))
        gap = int(rng.integers(5, 80))
        rows.append(_fit(rng, sample(p, cut) + sample(p + cut + gap, L - cut), L))
        ins = [rng.choice(list(b"ACGT")) for _ in range(gap % 40 + 3)]
        rows.append(_fit(rng, sample(p, cut) + ins + sample(p + cut, L), L))
    for _ in range(60):  # chimeras: two far-apart pieces, in either order
        p1, p2 = int(rng.integers(0, R - L)), int(rng.integers(0, R - L))
        cut = int(rng.integers(10, L - 10))
        rows.append(_fit(rng, sample(p1, cut) + sample(p2, L - cut), L))
    for _ in range(60):  # junk prefix / junk suffix of every size
        p = int(rng.integers(0, R - L))
        j = int(rng.integers(1, L))
        junk = [rng.choice(list(b"ACGT")) for _ in range(j)]
        rows.append(_fit(rng, junk + sample(p + j, L - j), L))
        rows.append(_fit(rng, sample(p, L - j) + junk, L))
    for k in range(1, L, 7):  # hanging over the ends of the reference
        rows.append(_fit(rng, sample(R - k, k), L))
        rows.append(_fit(rng, [rng.choice(list(b"ACGT")) for _ in range(L - k)] + sample(0, k), L))
    for unit in (b"A", b"AC", b"ACG", b"AACCGGTT"):  # low complexity
        rows.append(_fit(rng, list(unit * L), L))
    rows.append(np.full(L, ord("N"), dtype=np.uint8))
    rows.append(_fit(rng, list(b"N" * 40) + sample(100, L), L))
    for _ in range(40):  # no hit
        rows.append(_fit(rng, [], L))
    for _ in range(40):  # heavily mutated
        p = int(rng.integers(0, R - L))
        rows.append(_fit(rng, _mutate(rng, sample(p, L), subs=0.15, indel=0.03), L))
    return np.stack(rows)


@pytest.mark.parametrize("scheme", [(2, -5, -10, -1), (1, -1, -2, -1), (5, -4, -12, -2), (3, -2, -4, 0), (1, -3, -5, -2)])
@pytest.mark.parametrize("kind", ["random", "two_copies", "tandem", "low_complexity"])
def test_adversarial_reads_and_references(za, oracle, pruned, scheme, kind):
    from zoe_amd import synth

    rng = np.random.default_rng(stable_seed(scheme, kind))
    base = synth.reference_host(2000)
    if kind == "two_copies":  # every read has two equally good homes (one with a few differences)
        half = bytearray(base[:1000])
        other = bytearray(half)
        for i in range(0, 1000, 97):
            other[i] = ord("A") if other[i] != ord("A") else ord("C")
        ref = bytes(half + other)
    elif kind == "tandem":
        unit = base[:37]
        ref = bytes((unit * 60)[:2000])
    elif kind == "low_complexity":
        ref = bytes((b"A" * 300 + base[:400] + b"AC" * 200 + base[400:800] + b"T" * 100)[:2000])
    else:
        ref = base
    m, x, go, ge = scheme
    matrix = za.WeightMatrix.new_dna_matrix(m, x, b"N")
    reads = _adversarial_reads(rng, ref, 150)
    rescored = _check(za, oracle, reads, matrix, go, ge, ref, pruned)
    if kind == "random":
        assert 0 < rescored < len(reads)  # both outcomes occur


def test_large_batch_equals_the_full_pass_and_is_mostly_pruned(za, pruned):
    """2.5 M reads (more than one round of the two kernels): bit-identical to the full pass."""
    import torch

    from zoe_amd import synth

    n = 2_500_000
    ref = synth.reference_host(2000)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    rb = synth.reads_device(pruned, ref, 0, n, 150)
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    got = prof.sw_score_from_i8(ref)
    rescored = pruned.prune_rescored()
    with full_pass(pruned):
        want = prof.sw_score_from_i8(ref)
        assert pruned.prune_rescored() == 0
    assert torch.equal(got.score, want.score) and torch.equal(got.status, want.status) and torch.equal(got.tier, want.tier)
    assert 0 < rescored < n // 10


def _tensors(x):
    return [getattr(x, f) for f in ("score", "status", "tier", "ref_start", "ref_end", "query_start", "query_end") if getattr(x, f, None) is not None]


@pytest.mark.parametrize("kind", ["synthetic", "adversarial", "two_copies"])
def test_ends_ranges_and_alignments_with_the_pruned_first_pass(za, oracle, pruned, kind):
    """Score + ends (MODE 1, 2 of the window kernel: first row holding the maximum, then first column) behind sw_score_ends,
    sw_score_ranges, the alignment's first pass and the 3-pass alignment: identical to the full pass, and to the oracle."""
    import torch

    from zoe_amd import _lib, synth

    rng = np.random.default_rng(stable_seed("ends", kind))
    base = synth.reference_host(2000)
    if kind == "two_copies":  # two identical homes: the tie rule picks the first row
        ref = bytes(base[:1000] + base[:1000])
    else:
        ref = base
    reads = synth.reads_host(ref, 3, 3000, 150) if kind == "synthetic" else _adversarial_reads(rng, ref, 150)
    n = len(reads)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    rb = za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda(), 150)
    direct = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16)
    casc = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    seq = za.SeqSrc.Reference(ref)

    def run_all():
        return (direct.sw_score_ends(seq), direct.sw_score_ranges(seq), casc.sw_score_ranges_from_i8(seq), casc.sw_align_from_i8(seq),
                casc.sw_align_from_i8_3pass(seq))

    got = run_all()
    rescored = pruned.prune_rescored()
    with full_pass(pruned):
        want = run_all()
    for g, w in zip(got[:3], want[:3]):
        for tg, tw in zip(_tensors(g), _tensors(w)):
            assert torch.equal(tg, tw)
    for g, w in zip(got[3:], want[3:]):
        assert np.array_equal(g.status, w.status)
        for i in range(n):
            assert g.key(i) == w.key(i), i
    sc = oracle.Scoring(dna.signed_weights(), dna.mapping.index_map, -10, -1)
    rg = got[1]
    for i in range(0, n, max(1, n // 150)):
        st, s, rr, qr = oracle.score_ranges("i16", 16, sc, reads[i], ref)
        assert int(rg.status[i]) == st, i
        if st == 0:
            assert (int(rg.score[i]), (int(rg.ref_start[i]), int(rg.ref_end[i])), (int(rg.query_start[i]), int(rg.query_end[i]))) == (s, rr, qr), i
    if kind == "synthetic":
        assert 0 < rescored < n // 2  # the pruned first pass ran (the 2 % random reads always go back), and mostly pruned


@pytest.mark.parametrize("L,R", [(153, 2000), (250, 5000), (304, 3000), (305, 4000), (400, 30000)])
def test_longer_reads_and_references_than_one_row_table(za, oracle, pruned, L, R):
    """The wider classes (48-column strip; 8 x 32 and 16 x 22 window columns) and references longer than one LDS row table
    (each block stages the rows its windows cover)."""
    from zoe_amd import synth

    rng = np.random.default_rng(stable_seed("long", L, R))
    ref = synth.reference_host(R)
    reads = synth.reads_host(ref, 17, 1500, L)
    adv = _adversarial_reads(rng, ref, L)[::3]
    reads = np.concatenate([reads, adv])
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    rescored = _check(za, oracle, reads, dna, -10, -1, ref, pruned)
    assert 0 < rescored < len(reads)


def test_ragged_batch_mixed_lengths_vs_30kb(za, oracle, pruned):
    """BASELINE.json configs[4] shape: reads of 75-400 bp vs a 30 kb reference; the length classes of the ragged batch that
    share a pruning class are pruned together, every other class takes the full pass. Empty reads included."""
    import torch

    from zoe_amd import synth

    n = 6000
    ref = synth.reference_host(30000)
    bases, off = synth.reads_ragged_host(ref, 5, n, 75, 400)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = oracle.Scoring(dna.signed_weights(), dna.mapping.index_map, -10, -1)
    want_s, want_st, want_tier = oracle.batch_score_w256(8, sc, bases, ref, offsets=off, threads=8)
    rb = za.ReadBatch(torch.from_numpy(bases).cuda(), n, offsets=torch.from_numpy(off.astype(np.int64)).cuda())
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    got = prof.sw_score_from_i8(ref)
    rescored = pruned.prune_rescored()
    assert np.array_equal(got.status.cpu().numpy(), want_st)
    assert np.array_equal(got.score.cpu().numpy().view(np.uint32), want_s)
    assert np.array_equal(got.tier.cpu().numpy(), want_tier)
    assert 0 < rescored < n // 2
    rg = prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
    with full_pass(pruned):
        want = prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
    for f in ("score", "status", "tier", "ref_start", "ref_end", "query_start", "query_end"):
        assert torch.equal(getattr(rg, f), getattr(want, f)), f


def test_the_option_switches_the_pruned_pass_off_and_on(za):
    """zsw_set_option(ZSW_OPTION_EXACT_PRUNING): on by default; 0 computes every cell; zsw_debug_set does not touch it."""
    import torch

    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    n = 100000
    rb = synth.reads_device(ctx, ref, 0, n, 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    got = prof.sw_score_from_i8(ref)
    assert 0 < ctx.prune_rescored() < n // 4  # the default path prunes (the 2 % random reads always go back)
    ctx.debug_set(0)  # the debug word and the options are separate
    again = prof.sw_score_from_i8(ref)
    assert 0 < ctx.prune_rescored() < n // 4
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    try:
        full = prof.sw_score_from_i8(ref)
        assert ctx.prune_rescored() == 0
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    for a in (got, again):
        assert torch.equal(a.score, full.score) and torch.equal(a.status, full.status) and torch.equal(a.tier, full.tier)
    with pytest.raises(_lib.ZswError):
        ctx.set_option(99, 1)
    with pytest.raises(_lib.ZswError):
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 2)
