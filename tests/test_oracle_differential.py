"""Differential checks inside the oracle (the reference's test pattern 2, src/alignment/sw/test.rs:7-51, widened):
striped vs scalar restatements, every lane count, every integer type; the AVX2 baseline build vs the plain-array
restatement; and BASELINE.json configs[0] (10k synthetic 150 bp reads vs a 2 kb reference on CPU).

Establishes the facts the GPU design relies on (SURVEY.md §7 hard part 1): score and ends are invariant to the lane
count N and equal the scalar algorithm; CIGARs are not.
"""
import numpy as np
import pytest

from conftest import stable_seed

S_, O_, U_ = 0, 1, 2
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_pairs(rng, n, ref_len=(40, 160), read_len=(10, 50)):
    out = []
    for _ in range(n):
        R = int(rng.integers(*ref_len))
        ref = bytes(rng.choice(ALPHA, R))
        L = int(rng.integers(*read_len))
        if rng.random() < 0.6 and R > L:
            s = int(rng.integers(0, R - L))
            r = bytearray(ref[s : s + L])
            for _ in range(int(rng.integers(0, 4))):
                k = int(rng.integers(0, len(r)))
                t = rng.random()
                if t < 0.4:
                    r[k] = int(rng.choice(ALPHA))
                elif t < 0.7 and len(r) > 1:
                    del r[k]
                else:
                    r.insert(k, int(rng.choice(ALPHA)))
            read = bytes(r)
        else:
            read = bytes(rng.choice(ALPHA[: int(rng.integers(1, 5))], L))
        out.append((read, ref))
    return out


SCHEMES = [(4, -2, -3, -1), (2, -5, -10, -1), (3, -1, -4, -1), (1, -1, 0, 0), (5, -4, -2, 0), (2, -3, -5, -5), (10, -10, -5, -5)]


@pytest.mark.parametrize("scheme", SCHEMES)
def test_score_and_ends_are_layout_invariant(oracle, scheme):
    ma, mi, go, ge = scheme
    sc = oracle.dna_scoring(ma, mi, b"N", go, ge)
    rng = np.random.default_rng(stable_seed(scheme))
    n_cigar_diff = 0
    for read, ref in rand_pairs(rng, 60):
        st0, s0 = oracle.scalar_score(sc, read, ref)
        a0 = oracle.scalar_align(sc, read, ref)
        cigars = set()
        for N in (2, 4, 8, 16, 32, 64):
            assert oracle.score("i16", N, sc, read, ref) == (st0, s0)
            st, (s, re_, qe) = oracle.score_ends("i16", N, sc, read, ref)
            assert st == st0
            a = oracle.align("i16", N, sc, read, ref)
            assert a.status == st0
            if st0 == S_:
                assert (s, re_, qe) == (s0, a0.ref_range[1], a0.query_range[1])
                assert (a.score, a.ref_range[1], a.query_range[1]) == (s0, re_, qe)
                # every emitted CIGAR re-scores to the reported score (sw/mod.rs:399-454)
                assert oracle.score_from_path(sc, read, ref[a.ref_range[0] : a.ref_range[1]], a.cigar) == s0
                cigars.add(a.cigar)
                # i8 and i16 give identical alignments at equal N when nothing saturates
                a8 = oracle.align("i8", N, sc, read, ref)
                if a8.status == S_:
                    assert a8.key() == a.key()
                au = oracle.align("u16", N, sc, read, ref)
                assert au.key() == a.key()
        n_cigar_diff += len(cigars) > 1
    # ties make the traceback layout dependent only when gaps are cheap; never at the reference's test parameters
    if scheme == (2, -5, -10, -1):
        assert n_cigar_diff == 0


def test_overflow_predicates(oracle):
    """score_to_maybe_aligned (striped.rs:610-633): signed Overflowed <=> s >= T::MAX as a true score of 2^bits-1;
    unsigned <=> s >= T::MAX - bias."""
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)  # bias 5
    for L in (120, 124, 125, 126, 127, 128, 130):
        seq = (b"ACGT" * 40)[:L]
        s = 2 * L
        assert oracle.score("i16", 16, sc, seq, seq) == (S_, s)
        assert oracle.score("i8", 32, sc, seq, seq)[0] == (O_ if s >= 255 else S_)
        assert oracle.score("u8", 32, sc, seq, seq)[0] == (O_ if s >= 250 else S_)
        assert oracle.score_ends("i8", 32, sc, seq, seq)[0] == (O_ if s >= 255 else S_)
        assert oracle.align("u8", 16, sc, seq, seq).status == (O_ if s >= 250 else S_)
        st, sv, tier = oracle.cascade_score(8, 256, sc, seq, seq)
        assert (st, sv, tier) == (S_, s, 8 if s < 255 else 16)


def test_avx2_baseline_equals_plain_restatement(oracle):
    from zoe_amd import synth

    ref = synth.reference_host(600)
    reads = synth.reads_host(ref, 0, 300, 150)
    reads[3, 40:] = ord("N")
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    for width in (8, 16):
        s, st, tier = oracle.batch_score_w256(width, sc, reads, ref, fixed_len=150, threads=2)
        for i in range(300):
            assert (int(st[i]), int(s[i]), int(tier[i])) == oracle.cascade_score(width, 256, sc, reads[i], ref), (width, i)
    hb, off = synth.reads_ragged_host(ref, 9, 120, 20, 300)
    s, st, tier = oracle.batch_score_w256(8, sc, hb, ref, offsets=off.astype(np.uint64), threads=2)
    for i in range(120):
        assert (int(st[i]), int(s[i]), int(tier[i])) == oracle.cascade_score(8, 256, sc, hb[off[i] : off[i + 1]], ref), i


def test_config0_10k_reads_cpu(oracle):
    """BASELINE.json configs[0]: sw_simd_score on CPU, 10k synthetic 150 bp reads vs one 2 kb reference, default
    matrix — the plumbing case. i16x16-direct and the i8 -> i16 cascade agree; a slice is cross-checked against the
    plain-array and the scalar restatements."""
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    reads = synth.reads_host(ref, 0, 10000, 150)
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    s8, st8, t8 = oracle.batch_score_w256(8, sc, reads, ref, fixed_len=150, threads=8)
    s16, st16, t16 = oracle.batch_score_w256(16, sc, reads, ref, fixed_len=150, threads=8)
    assert np.array_equal(s8, s16) and np.array_equal(st8, st16)
    assert np.array_equal(t8 == 8, s8 < 255) and (t16 == 16).all()
    assert (st8 == S_).mean() > 0.97 and np.median(s8[st8 == S_]) > 250
    for i in range(0, 10000, 500):
        assert oracle.scalar_score(sc, reads[i], ref) == (int(st8[i]), int(s8[i]))
        assert oracle.score("i16", 16, sc, reads[i], ref) == (int(st8[i]), int(s8[i]))


@pytest.mark.parametrize("scheme", [(2, -5, -10, -1), (4, -2, -3, -1), (3, -1, -4, -1)])
def test_three_pass_is_a_valid_optimal_alignment(oracle, scheme):
    """sw_align_3pass (three_pass.rs:21-104): same score as the striped path, a CIGAR that re-scores to it, all three
    routes exercised (no-gaps shortcut, banded, scalar fallback)."""
    ma, mi, go, ge = scheme
    sc = oracle.dna_scoring(ma, mi, b"N", go, ge)
    rng = np.random.default_rng(stable_seed(scheme))
    hows = set()
    for read, ref in rand_pairs(rng, 150, ref_len=(60, 200), read_len=(12, 70)):
        a, how = oracle.align_3pass("i16", 16, sc, read, ref)
        st, s = oracle.score("i16", 16, sc, read, ref)
        assert a.status == st
        if st == S_:
            hows.add(how)
            assert a.score == s
            assert oracle.score_from_path(sc, read, ref[a.ref_range[0] : a.ref_range[1]], a.cigar) == s
            if how == 0:
                assert "D" not in a.cigar and "I" not in a.cigar
    assert 0 in hows and (1 in hows or 2 in hows)
