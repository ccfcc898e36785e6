"""GPU tests of the seeded exact pass as the DEFAULT path (zsw_score_seed.hip): what switches it on and off, what it costs
when it cannot help, and the inputs its index has to get right. (Parity on adversarial read sets, both pruned passes:
tests/test_gpu_prune.py; the arithmetic against the full Gotoh matrix, no GPU: tests/test_align_models.py.)"""
import time

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


def _batch(za, reads2d):
    import torch

    n, L = reads2d.shape
    return za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads2d).reshape(-1)).cuda(), L)


def _oracle_check(za, oracle, reads2d, matrix, go, ge, ref):
    sc = oracle.Scoring(matrix.signed_weights(), matrix.mapping.index_map, go, ge)
    ws, wst, wt = oracle.batch_score_w256(8, sc, reads2d, ref, fixed_len=reads2d.shape[1], threads=8)
    got = za.LocalProfilesBatch.new_with_w256(_batch(za, reads2d), matrix, go, ge).sw_score_from_i8(ref)
    assert np.array_equal(got.status.cpu().numpy(), wst)
    assert np.array_equal(got.score.cpu().numpy().view(np.uint32), ws)
    assert np.array_equal(got.tier.cpu().numpy(), wt)
    return za.SwContext.get(0).prune_rescored()


def test_unrelated_reads_cost_the_full_pass_and_nothing_more(za):
    """The bail-out: reads without an anchor go straight from the seed kernel to the full pass. 2 M random reads must cost at
    most 1.05 x the full pass (plus 1 ms), with identical results."""
    import torch

    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    n, L = 2_000_000, 150
    ref = synth.reference_host(2000)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    bases = torch.randint(0, 4, (n * L,), device="cuda", generator=g, dtype=torch.int32)
    bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[bases.long()]
    prof = za.LocalProfilesBatch.new_with_w256(za.ReadBatch.from_fixed(bases, L), za.WeightMatrix.new_dna_matrix(2, -5, b"N"), -10, -1)

    def timed(reps=3):
        prof.sw_score_from_i8(ref)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = prof.sw_score_from_i8(ref)
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) / reps

    seeded, t_seeded = timed()
    handed_back = ctx.prune_rescored()
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    try:
        full, t_full = timed()
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    assert torch.equal(seeded.score, full.score) and torch.equal(seeded.status, full.status) and torch.equal(seeded.tier, full.tier)
    assert handed_back > 0.95 * n  # random 150-mers: a chance pair of agreeing 8-mers is rare
    assert t_seeded <= 1.05 * t_full + 1e-3, (t_seeded, t_full)


def test_the_index_follows_the_reference_and_the_matrix(za, oracle):
    """zsw_set_reference / zsw_set_scoring invalidate the k-mer index: the same reads against reference A, B, A again, then A
    under another matrix (other good residues, other lambda), always equal to the oracle and mostly seeded."""
    from zoe_amd import synth

    ref_a = synth.reference_host(2000)
    ref_b = bytes(reversed(synth.reference_host(3100)))
    reads = np.concatenate([synth.reads_host(ref_a, 0, 1500, 150), synth.reads_host(ref_b, 7, 1500, 150)])
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    other = za.WeightMatrix.new_dna_matrix(3, -4, b"N")
    for ref, m, go, ge in ((ref_a, dna, -10, -1), (ref_b, dna, -10, -1), (ref_a, dna, -10, -1), (ref_a, other, -6, -2), (ref_b, other, -6, -2)):
        back = _oracle_check(za, oracle, reads, m, go, ge, ref)
        assert 1400 < back < 1700  # the half sampled from the other reference has no anchor


@pytest.mark.parametrize("kind", ["n_singles", "n_runs", "lower_case", "iupac_bytes"])
def test_references_with_bytes_that_are_not_good_residues(za, oracle, kind):
    """N (and any byte the map sends to the catch-all) in the reference: windows with up to three of them are indexed under
    every spelling, longer runs cost a path more than a k-mer is worth; reads with N sample fewer usable k-mers."""
    from zoe_amd import synth

    rng = np.random.default_rng(stable_seed("seedref", kind))
    ref = bytearray(synth.reference_host(2500))
    if kind == "n_singles":
        for p in rng.choice(len(ref), 120, replace=False):
            ref[p] = ord("N")
    elif kind == "n_runs":
        for p in rng.choice(len(ref) - 40, 12, replace=False):
            ref[p : p + int(rng.integers(2, 30))] = b"N" * 40
        ref = ref[:2500]
    elif kind == "lower_case":
        ref = bytearray(bytes(ref).lower())
    else:
        for p in rng.choice(len(ref), 200, replace=False):
            ref[p] = int(rng.choice(list(b"RYKMSWBDHVU-")))
    ref = bytes(ref)
    reads = synth.reads_host(ref, 3, 3000, 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    back = _oracle_check(za, oracle, reads, dna, -10, -1, ref)
    assert back < 3000
    # and with N scored like a mismatch (no ignored residue): N is then an ordinary non-good residue with its own loss
    plain = za.WeightMatrix.new_dna_matrix(2, -3, None)
    _oracle_check(za, oracle, reads, plain, -5, -1, ref)


@pytest.mark.parametrize("R", [1, 7, 8, 9, 23, 60, 149, 150, 151])
def test_references_shorter_than_a_kmer_a_read_or_the_window(za, oracle, R):
    from zoe_amd import synth

    big = synth.reference_host(4000)
    ref = big[1000 : 1000 + R]
    reads = synth.reads_host(big, 2, 1200, 150)
    reads[:600] = synth.reads_host(big[900:1300], 4, 600, 150)  # overlap the short reference
    _oracle_check(za, oracle, reads, za.WeightMatrix.new_dna_matrix(2, -5, b"N"), -10, -1, ref)


def test_batches_below_the_size_threshold_take_the_full_pass(za, oracle):
    from zoe_amd import synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    assert _oracle_check(za, oracle, synth.reads_host(ref, 0, 1023, 150), dna, -10, -1, ref) == 0
    assert 0 < _oracle_check(za, oracle, synth.reads_host(ref, 0, 1024, 150), dna, -10, -1, ref) < 200


def test_matrices_the_index_cannot_spell_take_the_full_pass(za, oracle):
    """More than four residues that score the maximum against themselves (a 5-letter identity matrix), a mismatch that scores
    the maximum, free gaps: seed_analyze declines, every cell is computed, the results are the oracle's."""
    from zoe_amd import synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    reads = synth.reads_host(ref, 0, 2000, 150)
    m5 = za.WeightMatrix.new_dna_matrix(2, -5, None)  # N scores +2 against N: five good residues
    assert _oracle_check(za, oracle, reads, m5, -10, -1, ref) == 0
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    assert _oracle_check(za, oracle, reads, dna, 0, 0, ref) == 0  # free gaps
    w = np.array(dna.signed_weights(), dtype=np.int8).reshape(5, 5).copy()
    w[0, 1] = w[1, 0] = 2  # A and C are interchangeable: a mismatch at the maximum
    same = za.WeightMatrix.new_custom(dna.mapping, w)
    assert _oracle_check(za, oracle, reads, same, -10, -1, ref) == 0


def test_long_reads_take_the_wide_window_configurations(za, oracle):
    """Reads of 500-2,400 bases vs a 30 kb reference: 16 x 38 and 64-lane window configurations (a window of len + ~100 rows of
    the 30,000)."""
    from zoe_amd import synth

    ref = synth.reference_host(30000)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    from zoe_amd import _lib

    ctx = za.SwContext.get(0)
    try:
        for flags in (0, _lib.DEBUG_SEED_NO_BAND):  # score-only calls: the banded kernel, then whole rows
            ctx.debug_set(flags)
            for L in (500, 608, 1216, 2400):
                reads = synth.reads_host(ref, L, 1100, L)
                back = _oracle_check(za, oracle, reads, dna, -10, -1, ref)
                if L <= 608:  # sixteen sampled k-mers prove at most 16 x lambda = 112: longer reads lose more than that to their own errors
                    assert back < 600, (L, back, flags)
    finally:
        ctx.debug_set(0)


def test_banded_pass_pairs_lengths_and_edges(za, oracle):
    """The banded kernel walks two reads per lane (16-bit halves) strip by strip: pairs whose anchors lie too far apart to share a
    band (sparse batch on a long reference), pairs of different lengths (ragged batch), an odd read count, reads hanging over both
    ends of the reference (anchor diagonals < 0 and > ref_len - len), free gap extension — each against the oracle, and the two
    seeded kernels must hand back similar numbers of reads."""
    import torch
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    rng = np.random.default_rng(stable_seed("band-edges"))
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    try:
        # sparse: 1,501 reads of 150 over 30 kb (neighbouring anchors ~20 apart, many pairs beyond the 32-diagonal slack)
        ref = synth.reference_host(30000)
        reads = synth.reads_host(ref, 9, 1501, 150)
        ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE)
        back_band = _oracle_check(za, oracle, reads, dna, -10, -1, ref)
        ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE | _lib.DEBUG_SEED_WIDE_BAND)
        _oracle_check(za, oracle, reads, dna, -10, -1, ref)  # the full band at once
        ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE | _lib.DEBUG_SEED_NO_BAND)
        back_window = _oracle_check(za, oracle, reads, dna, -10, -1, ref)
        assert back_window < 150 and back_band < 900, (back_band, back_window)
        # overhanging reads: copies of reference[-40..110) and reference[R-110..R+40), the overhang random
        ref2 = synth.reference_host(2000)
        r2 = np.frombuffer(bytes(ref2), dtype=np.uint8)
        over = np.empty((2049, 150), dtype=np.uint8)
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        for i in range(over.shape[0]):
            h = int(rng.integers(1, 60))
            junk = acgt[rng.integers(0, 4, h)]
            over[i] = np.concatenate([junk, r2[: 150 - h]]) if i % 2 else np.concatenate([r2[len(r2) - (150 - h):], junk])
        for scheme in ((2, -5, -10, -1), (3, -2, -4, 0), (1, -1, -2, -1)):
            m = za.WeightMatrix.new_dna_matrix(scheme[0], scheme[1], b"N")
            ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE)
            _oracle_check(za, oracle, over, m, scheme[2], scheme[3], ref2)
        # ragged: lengths 24-400 side by side in a lane
        n = 5001
        bases, off = synth.reads_ragged_host(ref2, 11, n, 24, 400)
        sc = oracle.Scoring(dna.signed_weights(), dna.mapping.index_map, -10, -1)
        ws, wst, wt = oracle.batch_score_w256(8, sc, bases, ref2, offsets=off, threads=8)
        rb = za.ReadBatch(torch.from_numpy(bases).cuda(), n, offsets=torch.from_numpy(off.astype(np.int64)).cuda())
        got = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1).sw_score_from_i8(ref2)
        assert np.array_equal(got.status.cpu().numpy(), wst)
        assert np.array_equal(got.score.cpu().numpy().view(np.uint32), ws)
        assert np.array_equal(got.tier.cpu().numpy(), wt)
        assert ctx.prune_rescored() < n // 4
    finally:
        ctx.debug_set(0)


def test_every_length_class_of_a_large_ragged_batch_takes_the_seeded_pass(za):
    """6 M reads of 75-400 bases: the length classes share the banded kernel's blocks (and their boundary buffers) in proportion
    to their reads, so the classes' workspace regions fit what the call allocated. A class that did not fit would quietly take the
    full pass: then the 2 % random reads of that class would not show up among the handed-back reads."""
    import torch
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    n = 6_000_000
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    rb = synth.reads_ragged_device(ctx, ref, 0, n, 75, 400)
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    got = prof.sw_score_from_i8(ref)
    back = ctx.prune_rescored()
    assert 0.019 * n < back < 0.04 * n, back  # the 2 % random reads of every class, and little else
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    try:
        full = prof.sw_score_from_i8(ref)
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    assert torch.equal(got.score, full.score) and torch.equal(got.status, full.status) and torch.equal(got.tier, full.tier)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 129, 513])
def test_banded_pass_work_queue_on_tiny_batches(za, oracle, n):
    """The banded kernel hands out pairs 64 at a time and stops at the first chunk past the end: one read, an odd count, counts
    around one and two chunks (two reads per pair), through both tiers; scores and ranges against the full pass (and a few reads
    against the oracle)."""
    import torch
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    reads = synth.reads_host(ref, 1000 + n, n, 150)
    rb = _batch(za, reads)
    casc = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    seq = za.SeqSrc.Reference(ref)
    try:
        ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE)
        got_s = casc.sw_score_from_i8(ref)
        got_r = casc.sw_score_ranges_from_i8(seq)
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
        want_s = casc.sw_score_from_i8(ref)
        want_r = casc.sw_score_ranges_from_i8(seq)
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
        ctx.debug_set(0)
    for f in ("score", "status", "tier"):
        assert torch.equal(getattr(got_s, f), getattr(want_s, f)), f
    for f in ("score", "status", "ref_start", "ref_end", "query_start", "query_end"):
        assert torch.equal(getattr(got_r, f), getattr(want_r, f)), f
    sc = oracle.Scoring(dna.signed_weights(), dna.mapping.index_map, -10, -1)
    for i in range(0, n, max(1, n // 8)):
        st, s, rr, qr = oracle.score_ranges("i16", 16, sc, reads[i], ref)
        assert int(got_r.status[i]) == st, i
        if st == 0:
            assert (int(got_r.score[i]), (int(got_r.ref_start[i]), int(got_r.ref_end[i])), (int(got_r.query_start[i]), int(got_r.query_end[i]))) == (s, rr, qr), i


@pytest.mark.parametrize("rate", [30, 50, 80, 120])
def test_diverged_reads_score_ranges_and_alignments_vs_oracle(za, oracle, rate):
    """Reads 3 / 5 / 8 / 12 % away from the reference (+ a tenth of that in indels): the seeded pass proves fewer and fewer of
    them (the k-mer bound's slope is one substitution per ten bases) and hands the rest to the full pass — every entry point that
    starts with it must equal the oracle on all of them: score cascade, ranges cascade, exact alignment, 3-pass alignment."""
    import torch

    from test_gpu_bounds import diverged_reads
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    n = 1500
    reads = diverged_reads(ref, n, 150, rate, stable_seed("diverged", rate))
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = oracle.Scoring(dna.signed_weights(), dna.mapping.index_map, -10, -1)
    ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE)
    try:
        back = _oracle_check(za, oracle, reads, dna, -10, -1, ref)
        assert (back < 0.2 * n) if rate <= 30 else (back > 0.5 * n if rate >= 120 else True), back
        prof = za.LocalProfilesBatch.new_with_w256(_batch(za, reads), dna, -10, -1)
        rg = prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
        al = prof.sw_align_from_i8(za.SeqSrc.Reference(ref))
        a3 = prof.sw_align_from_i8_3pass(za.SeqSrc.Reference(ref))
        torch.cuda.synchronize()
    finally:
        ctx.debug_set(0)
    for i in range(0, n, 3):
        o_st, o_s, o_rr, o_qr, o_t = oracle.cascade_score_ranges(8, 256, sc, reads[i], ref)
        assert int(rg.status[i]) == o_st, i
        if o_st == 0:
            assert (int(rg.score[i]), (int(rg.ref_start[i]), int(rg.ref_end[i])), (int(rg.query_start[i]), int(rg.query_end[i])), int(rg.tier[i])) == (o_s, o_rr, o_qr, o_t), i
        want, tier = oracle.cascade_align(8, 256, sc, reads[i], ref)
        assert al.key(i) == (want.key() if want.status == 0 else (want.status, 0, (0, 0), (0, 0), "", 0, 0)), i
        want3, tier3, _how = oracle.cascade_align_3pass(8, 256, sc, reads[i], ref)
        assert a3.key(i) == (want3.key() if want3.status == 0 else (want3.status, 0, (0, 0), (0, 0), "", 0, 0)), i


def test_handed_back_reads_against_a_long_reference_run_in_row_chunks(za, oracle):
    """Reads the seeded pass hands back (no anchor, failed proof) are scored over all their cells; against a 30 kb reference that
    pass runs in chunks of rows, each an item of its own, and the largest (score, earliest row, earliest column) over a read's
    chunks is its result (tests/models/chunk_rows.cpp). Unrelated reads, reads from a duplicated stretch of the reference (the
    maximum in several rows: the first must win, whichever chunk holds it) and heavily diverged reads: equal to the whole-row pass
    (ZSW_DEBUG_NO_ROW_CHUNKS) on every read — score, ranges — and to the oracle on a sample."""
    import torch

    from test_gpu_bounds import diverged_reads
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    rng = np.random.default_rng(stable_seed("chunks"))
    ref = bytearray(synth.reference_host(30000))
    ref[21000:21400] = ref[3000:3400]   # a second copy far below the first
    ref[12000:12150] = ref[3100:3250]   # and a third
    ref = bytes(ref)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    r = np.frombuffer(ref, dtype=np.uint8)
    parts = [rng.choice(alpha, (3000, 150)).astype(np.uint8), diverged_reads(ref, 1500, 150, 150, 5)]
    dup = np.stack([r[3000 + int(s):3150 + int(s)] for s in rng.integers(0, 250, 1500)])  # pieces of the duplicated stretch ...
    dup = np.where(rng.random(dup.shape) < 0.12, rng.choice(alpha, dup.shape), dup).astype(np.uint8)  # ... too diverged for the proof
    reads = np.concatenate(parts + [dup])
    n = len(reads)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    prof = za.LocalProfilesBatch.new_with_w256(_batch(za, reads), dna, -10, -1)
    got = prof.sw_score_from_i8(ref)
    assert ctx.prune_rescored() > 0.6 * n
    rg = prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
    ctx.debug_set(_lib.DEBUG_NO_ROW_CHUNKS)
    try:
        want = prof.sw_score_from_i8(ref)
        wrg = prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
    finally:
        ctx.debug_set(0)
    for f in ("score", "status", "tier"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    for f in ("score", "status", "tier", "ref_start", "ref_end", "query_start", "query_end"):
        assert torch.equal(getattr(rg, f), getattr(wrg, f)), f
    sc = oracle.Scoring(dna.signed_weights(), dna.mapping.index_map, -10, -1)
    for i in list(range(0, n, 61)) + list(range(4500, n, 37)):
        o_st, o_s, o_rr, o_qr, o_t = oracle.cascade_score_ranges(8, 256, sc, reads[i], ref)
        assert int(rg.status[i]) == o_st, i
        if o_st == 0:
            assert (int(rg.score[i]), (int(rg.ref_start[i]), int(rg.ref_end[i])), (int(rg.query_start[i]), int(rg.query_end[i])), int(rg.tier[i])) == (o_s, o_rr, o_qr, o_t), i


def test_ranges_reverse_pass_as_a_second_seeded_pass(za, oracle):
    """sw_simd_score_ranges' second pass (striped.rs:355-388) runs, for the reads whose forward maximum sits in one cell, as a seeded
    pass over the reversed reads and the reversed reference; a read is settled if the reversed maximum sits in one cell too, every
    other read takes the exact prefix kernel. 400,000 reads — synthetic, tie-rich (tandem repeats, a duplicated stretch, two halves
    from different places, low complexity) and diverged — must equal the all-exact reverse pass (ZSW_DEBUG_RANGES_EXACT_REVERSE) in
    every field, the 3-pass alignments built on the ranges too, and the oracle on a sample."""
    import torch

    from test_gpu_bounds import diverged_reads
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    rng = np.random.default_rng(stable_seed("rev-seeded"))
    ref = bytearray(synth.reference_host(2500))
    ref[900:1000] = ref[300:400]
    for i in range(1500 + 2, 1580):
        ref[i] = ref[i - 2]
    ref = bytes(ref)
    r = np.frombuffer(ref, dtype=np.uint8)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    L = 150
    syn = synth.reads_host(ref, 17, 300_000, L)
    ties = np.empty((60_000, L), dtype=np.uint8)
    for i in range(len(ties)):
        kind = i % 5
        p = int(rng.integers(0, len(r) - L))
        if kind == 0:
            p = int(rng.integers(250, 320))          # inside the duplicated stretch
        elif kind == 1:
            p = int(rng.integers(1440, 1520))        # across the tandem repeat
        q = r[p:p + L].copy()
        if kind == 2:                                # two halves from different places
            p2 = int(rng.integers(0, len(r) - L))
            q[L // 2:] = r[p2:p2 + L - L // 2]
        elif kind == 3:
            q = rng.choice(alpha[:2], L).astype(np.uint8)
        ties[i] = q
    reads = np.concatenate([syn, ties, diverged_reads(ref, 20_000, L, 50, 3), diverged_reads(ref, 20_000, L, 100, 4)])
    n = len(reads)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    prof = za.LocalProfilesBatch.new_with_w256(_batch(za, reads), dna, -10, -1)
    got = prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
    got3 = prof.sw_align_from_i8_3pass(za.SeqSrc.Reference(ref))
    ctx.debug_set(_lib.DEBUG_RANGES_EXACT_REVERSE)
    try:
        want = prof.sw_score_ranges_from_i8(za.SeqSrc.Reference(ref))
        want3 = prof.sw_align_from_i8_3pass(za.SeqSrc.Reference(ref))
    finally:
        ctx.debug_set(0)
    for f in ("score", "status", "tier", "ref_start", "ref_end", "query_start", "query_end"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    assert np.array_equal(got3.status, want3.status) and np.array_equal(got3.records, want3.records)
    assert np.array_equal(got3.inc, want3.inc) and np.array_equal(got3.op, want3.op)
    sc = oracle.Scoring(dna.signed_weights(), dna.mapping.index_map, -10, -1)
    for i in list(range(0, 300_000, 1511)) + list(range(300_000, n, 257)):
        o_st, o_s, o_rr, o_qr, o_t = oracle.cascade_score_ranges(8, 256, sc, reads[i], ref)
        assert int(got.status[i]) == o_st, i
        if o_st == 0:
            assert (int(got.score[i]), (int(got.ref_start[i]), int(got.ref_end[i])), (int(got.query_start[i]), int(got.query_end[i])), int(got.tier[i])) == (o_s, o_rr, o_qr, o_t), i
