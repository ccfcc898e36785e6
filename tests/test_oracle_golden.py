"""Pins the CPU oracle (oracle/) against every known-answer vector the reference holds for the
striped Smith-Waterman path.  Citations are file:line in the reference checkout.

The reference (nightly Rust) cannot be built in this environment, so these literal expected values —
copied as data from the reference's tests/doctests — are what pins the oracle.
"""
import pytest

S_, O_, U_ = 0, 1, 2  # Some / Overflowed / Unmapped


def dna(o, m, x, go, ge, ign=b"N"):
    return o.dna_scoring(m, x, ign, go, ge)


# ---------------------------------------------------------------- scalar known answers
def test_scalar_h1_vs_h5(oracle, h1, h5):
    # src/alignment/sw/test.rs:72-85
    sc = dna(oracle, 2, -5, -10, -1)
    a = oracle.scalar_align(sc, h1, h5)
    assert (a.ref_range[0], a.score) == (336, 37)
    assert oracle.scalar_score(sc, h1, h5) == (S_, 37)
    v = b"A" * 100
    assert oracle.scalar_score(sc, v, v) == (S_, 200)


def test_t_u_check(oracle):
    # src/alignment/sw/test.rs:88-100
    sc = dna(oracle, 2, -5, -10, -1)
    q, r = b"ACGTUNacgtun", b"ACGTTNACGTTN"
    assert oracle.scalar_score(sc, q, r) == (S_, 20)
    assert oracle.score("u16", 16, sc, q, r) == (S_, 20)
    assert oracle.score("i16", 16, sc, q, r) == (S_, 20)


def test_sw_simd_h5_profile(oracle, h1, h5):
    # src/alignment/sw/test.rs:103-114
    sc = dna(oracle, 2, -5, -10, -1)
    assert oracle.score("u8", 16, sc, h5, h1) == (S_, 37)
    assert oracle.score("i16", 16, sc, h5, h1) == (S_, 37)


def test_poly_a(oracle):
    # src/alignment/sw/test.rs:265-271
    sc = dna(oracle, 2, -5, -10, -1)
    v = b"A" * 100
    assert oracle.score("u16", 16, sc, v, v) == (S_, 200)


def test_single_and_profile_set(oracle, cy):
    # src/alignment/sw/test.rs:274-280 and :304-311
    sc = dna(oracle, 2, -5, -10, -1)
    assert oracle.score("u16", 16, sc, cy, cy) == (S_, 3372)
    st, s, tier = oracle.cascade_score(8, 128, sc, cy, cy)
    assert (st, s) == (S_, 3372)
    assert tier == 16  # i8 overflows, i16 answers


def test_lazy_f_regression(oracle):
    # src/alignment/sw/test.rs:283-290
    sc = dna(oracle, 10, -10, -5, -5)
    assert oracle.score("u16", 4, sc, b"AGA", b"AA") == (S_, 15)


def test_overflow_check(oracle):
    # src/alignment/sw/test.rs:293-301
    sc = dna(oracle, 127, 0, -10, -1)
    st, _ = oracle.score("u8", 8, sc, b"AAAA", b"AAAA")
    assert st == O_


# ---------------------------------------------------------------- test_sw_simd_align! ×12
ALIGN_CASES = [
    # (profile_seq, other_seq, int, uint, lanes)   src/alignment/sw/test.rs:116-195
    ("H5", "H1", "i16", "u16", 8),
    (b"TTTAG", b"AAACTA", "i8", "u8", 2),
    (b"AAAAAATAAA", b"AAAAAAAAAA", "i8", "u8", 4),
    (b"CCCCA", b"TAAAA", "i8", "u8", 4),
    (b"CCCCC", b"TCCCC", "i8", "u8", 4),
    (b"CCCCT", b"GCTTTTC", "i8", "u8", 4),
    (b"TTTTTGTTTTCTTTTTTGTTTA", b"TTGTTTTTTTTTGTT", "i8", "u8", 16),
    (b"TTGTTTTGGGGAAAAA", b"TTTTTGTTTGGGAAAAATTCTT", "i8", "u8", 8),
    (b"TTTTTTTCTTGTTTTTG", b"TTTTTGTTTTCTTGGT", "i8", "u8", 16),
    (b"TTTTTTTTACTATTTTTAAATTTATGTTTTGTTA", b"TTTTTTTTTTTTAAAATTTGTAAACGTTTTGTTA", "i8", "u8", 8),
    (b"TTTTTTTTTTTTTTTTTTTCCTTTTTTTTTTTTTTTTTTTTTTTTTTCCCCCCTTTA", b"TTTATTTTTTTTTTTTTTCCCCCCCTTTTTTTTTTTTTTTTTCCCCCCTTT", "i8", "u8", 8),
    (b"TTTTTTTTTTTTTTTCCTTTTTTTTTTTTTTTTTTTCCCCCCCCCTA", b"TTTTTTTTTTTTTTTCCCCCTTTTTTTTTTCCCCCCCCCTT", "i8", "u8", 8),
]


@pytest.mark.parametrize("case", range(len(ALIGN_CASES)))
def test_sw_simd_align_macro(oracle, h1, h5, case):
    # macro body: src/alignment/sw/test.rs:7-51
    p, r, it, ut, lanes = ALIGN_CASES[case]
    if p == "H5":
        p, r = h5, h1
    sc = dna(oracle, 2, -5, -10, -1)
    st, score = oracle.score(it, lanes, sc, p, r)
    a_scalar = oracle.scalar_align(sc, p, r)
    a_simd = oracle.align(it, lanes, sc, p, r)
    assert a_scalar.status == S_
    assert (st, score) == (S_, a_scalar.score)
    assert a_scalar.key() == a_simd.key()
    a_simd_u = oracle.align(ut, lanes, sc, p, r)
    assert a_scalar.key() == a_simd_u.key()
    st, (score, ref_end, query_end) = oracle.score_ends(ut, lanes, sc, p, r)
    assert st == S_ and score == a_scalar.score
    assert ref_end == a_scalar.ref_range[1]
    assert query_end == a_scalar.query_range[1]


def test_sw_simd_locations(oracle):
    # src/alignment/sw/test.rs:198-241
    reference, query = b"TTTTTTCCTTTTTTTTCCCCCTTTTT", b"GGGGGGGCCCCCAAAA"
    sc = dna(oracle, 2, -5, -10, -1)
    a = oracle.scalar_align(sc, query, reference)
    st, (score, ref_end, query_end) = oracle.score_ends("u8", 8, sc, query, reference)
    assert st == S_ and score == a.score
    assert (ref_end, query_end) == (a.ref_range[1], a.query_range[1])
    query_rev = query[:query_end][::-1]
    st, (score2, ref_start, query_start) = oracle.score_ends("u8", 8, sc, query_rev, reference[:ref_end], forward=False)
    assert st == S_ and score2 == score
    assert (ref_start, ref_end) == a.ref_range
    assert (query_start, query_end) == a.query_range
    # reverse_from_forward == profile rebuilt from the reversed prefix (:233-234)
    import numpy as np

    p_rev = oracle.profile_dump("u8", 8, sc, query_rev)
    p_rev2 = oracle.profile_dump("u8", 8, sc, query, rev_end=query_end)
    assert np.array_equal(p_rev, p_rev2)
    st, score3, ref_range, query_range = oracle.score_ranges("u8", 8, sc, query, reference)
    assert st == S_ and score3 == a.score
    assert ref_range == a.ref_range and query_range == a.query_range


def test_sw_simd_ranges(oracle):
    # src/alignment/sw/test.rs:244-262
    sc = dna(oracle, 2, -5, -10, -1)
    a = oracle.scalar_align(sc, b"CCCCA", b"TAAAA")
    st, score, ref_range, query_range = oracle.score_ranges("u8", 8, sc, b"CCCCA", b"TAAAA")
    assert (st, score) == (S_, a.score)
    assert ref_range == a.ref_range and query_range == a.query_range


# ---------------------------------------------------------------- doctests
REF_A, QRY_A = b"ATGCATCGATCGATCGATCGATCGATCGATGC", b"CGTTCGCCATAAAGGGGG"


def test_doc_striped_score(oracle):
    # src/alignment/sw/striped.rs:45-55
    sc = dna(oracle, 4, -2, -3, -1)
    assert oracle.score("u8", 32, sc, QRY_A, REF_A) == (S_, 26)


def test_doc_striped_align(oracle):
    # src/alignment/sw/striped.rs:418-441
    sc = dna(oracle, 4, -2, -3, -1)
    a = oracle.align("u8", 8, sc, QRY_A, REF_A)
    assert (a.status, a.score, a.cigar) == (S_, 26, "6M2D9M3S")


def test_doc_profile_set_ranges(oracle):
    # src/alignment/profile_set.rs:293-310 (new_with_w256, from_i8 → i8x32 answers)
    sc = dna(oracle, 4, -2, -3, -1)
    st, score, ref_range, query_range = oracle.score_ranges("i8", 32, sc, QRY_A, REF_A)
    assert (st, score, query_range, ref_range) == (S_, 26, (0, 15), (14, 31))
    assert oracle.cascade_score(8, 256, sc, QRY_A, REF_A) == (S_, 26, 8)  # profile_set.rs:53-68
    assert oracle.cascade_score_ranges(8, 256, sc, QRY_A, REF_A) == (S_, 26, (14, 31), (0, 15), 8)  # the call of the doc test itself


def test_doc_sw_mod_dna(oracle):
    # src/alignment/sw/mod.rs:164-188, scalar.rs:165-169, mod.rs:224-248 (profile set)
    sc = dna(oracle, 4, -2, -3, -1)
    ref, q = b"GGCCACAGGATTGAG", b"CTCAGATTG"
    a = oracle.align("i8", 32, sc, q, ref)
    assert (a.status, a.score, a.cigar) == (S_, 27, "5M1D4M")
    s = oracle.scalar_align(sc, q, ref)
    assert (s.ref_range[0], s.cigar, s.score) == (3, "5M1D4M", 27)
    assert oracle.scalar_score(sc, q, ref) == (S_, 27)
    c, tier = oracle.cascade_align(8, 256, sc, q, ref)
    assert (c.score, c.cigar, tier) == (27, "5M1D4M", 8)
    assert oracle.cascade_score(8, 256, sc, q, ref)[:2] == (S_, 27)  # nucleotides/mod.rs:255-260


def test_doc_sw_mod_custom_alphabet(oracle):
    # src/alignment/sw/mod.rs:193-218 and :253-276
    m = oracle.byte_index_map(b"ABCD", b"A")
    w = oracle.weight_matrix_new(m, 4, 1, -1, None)
    sc = oracle.Scoring(w, m, -4, -2)
    ref, q = b"BDAACAABDDDB", b"AABDDAB"
    a = oracle.align("i8", 32, sc, q, ref)
    assert (a.status, a.score, a.cigar) == (S_, 5, "5M2S")
    c, _ = oracle.cascade_align(8, 256, sc, q, ref)
    assert (c.score, c.cigar) == (5, "5M2S")


def test_alignment_invert(oracle):
    # src/alignment/types/test.rs:14-38
    sc = dna(oracle, 4, -2, -3, -1)
    ref, q = b"GGCCACAGGATTGAGC", b"TCTCAGATTGCAGTTT"
    a = oracle.scalar_align(sc, q, ref)
    assert (a.ref_range, a.query_range, a.cigar) == ((3, 15), (1, 13), "1S5M1D4M1I2M3S")
    inv = oracle.scalar_align(sc, q, ref, other_is_query=True)
    assert (inv.ref_range, inv.query_range, inv.cigar) == ((1, 13), (3, 15), "3S5M1I4M1D2M1S")


def test_profile_set_get_i8(oracle, cy):
    # src/alignment/profile_set.rs:704-712: LocalProfiles::new_with_w256(..).get_i8() == StripedProfile::<i8,32,5>::new(..)
    # (trivially the same constructor here; also checks the striped layout rule of profile.rs:285)
    import numpy as np

    sc = dna(oracle, 2, -5, -10, -1)
    p = oracle.profile_dump("i8", 32, sc, cy)
    nv = (len(cy) + 31) // 32
    assert p.shape == (5, nv, 32)
    m = sc.index_map
    for ri in (0, 3, 4):
        for v in (0, 1, nv - 1):
            for lane in (0, 7, 31):
                q = v + lane * nv
                want = sc.weights[ri][m[cy[q]]] if q < len(cy) else 0
                assert p[ri, v, lane] == want
    assert np.array_equal(p, oracle.profile_dump("i8", 32, sc, cy))


def test_weight_matrix_new(oracle):
    # src/data/matrices/mod.rs:601-631
    sc = dna(oracle, 2, -5, -10, -1)
    want = [[2, -5, -5, -5, 0], [-5, 2, -5, -5, 0], [-5, -5, 2, -5, 0], [-5, -5, -5, 2, 0], [0, 0, 0, 0, 0]]
    assert sc.weights.tolist() == want
    biased, bias = oracle.to_biased_matrix(sc.weights)
    assert bias == 5 and biased[0][0] == 7 and biased[0][1] == 0 and biased[4][4] == 5


def test_dna_profile_map(oracle):
    # src/data/constants/mappings/dna.rs:177-178
    m = oracle.dna_profile_map()
    for ch, idx in zip(b"ACGTN", range(5)):
        assert m[ch] == idx and m[ch + 32] == idx
    assert m[ord("U")] == 3 and m[ord("u")] == 3
    assert m[ord("R")] == 4 and m[0] == 4 and m[255] == 4 and m[ord("-")] == 4


def test_validate_profile_args(oracle):
    # src/alignment/profile.rs:32-44
    v = oracle.validate_profile_args
    assert v(0, -10, -1) == 1
    assert v(5, 1, 0) == 2 and v(5, -128, -1) == 2
    assert v(5, -10, 1) == 3 and v(5, -10, -128) == 3
    assert v(5, -1, -2) == 4
    assert v(5, -10, -1) == 0 and v(5, 0, 0) == 0 and v(5, -127, -127) == 0
    with pytest.raises(oracle.ProfileError):
        oracle.score("i16", 16, dna(oracle, 2, -5, -10, -1), b"", b"ACGT")


def test_empty_reference(oracle):
    # striped.rs:219-221, :455-457; sw_simd_score on an empty reference leaves best = MIN → Unmapped (:140-141, :627)
    sc = dna(oracle, 2, -5, -10, -1)
    assert oracle.score("i16", 16, sc, b"ACGT", b"")[0] == U_
    assert oracle.score_ends("i16", 16, sc, b"ACGT", b"")[0] == U_
    assert oracle.align("i16", 16, sc, b"ACGT", b"").status == U_
    assert oracle.score("i16", 16, sc, b"ACGT", b"NNNN")[0] == U_


def test_score_from_path(oracle):
    # src/alignment/sw/test.rs:54-69
    sc = dna(oracle, 3, -1, -4, -1)
    reference, q = b"ATTCCTTTTGCCGGG", b"ATTGCGCCCGG"
    a = oracle.scalar_align(sc, q, reference)
    assert oracle.score_from_path(sc, q, reference[a.ref_range[0] : a.ref_range[1]], a.cigar) == a.score


def test_three_pass_and_banded_vectors(oracle):
    # src/alignment/profile_set.rs:183-209 (sw_align_from_i8_3pass doctest): score 26
    sc = dna(oracle, 4, -2, -3, -1)
    a, tier, how = oracle.cascade_align_3pass(8, 256, sc, QRY_A, REF_A)
    assert (a.status, a.score, tier) == (S_, 26, 8)
    assert oracle.score_from_path(sc, QRY_A, REF_A[a.ref_range[0] : a.ref_range[1]], a.cigar) == 26
    # src/alignment/sw/test.rs:364-381 (test_banded_sw_align_simple): score 10; :383-392 empty sequences
    sc2 = dna(oracle, 2, -1, -2, -1)
    b = oracle.banded_align(sc2, b"AACCGG", b"AAACCCGGG", 3)
    assert (b.status, b.score) == (S_, 10) and b.ref_range[1] > b.ref_range[0] and b.query_range[1] > b.query_range[0]
    assert oracle.banded_align(sc2, b"ACGT", b"", 3).status == U_
    with pytest.raises(oracle.ProfileError):
        oracle.banded_align(sc2, b"", b"ACGT", 3)
    # a band as wide as both sequences reproduces the scalar alignment (sw/test.rs:313-345 compares them loosely)
    sc3 = dna(oracle, 4, -2, -3, -1)
    assert oracle.banded_align(sc3, b"CTCAGATTG", b"GGCCACAGGATTGAG", 15).key() == oracle.scalar_align(sc3, b"CTCAGATTG", b"GGCCACAGGATTGAG").key()
