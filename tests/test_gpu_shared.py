"""GPU parity tests of the one-profile-many-sequences role (zsw_*_shared_batch; zsw_shared.hip, align_kernel<N, ., SHARED>): the
profile is built once from a sequence (Nucleotides::into_shared_profile, nucleotides/mod.rs:295-299; SharedProfiles,
profile_set.rs:552-560) and every read is aligned against it (sw/mod.rs:63-67). Checker: oracle/ with the SAME roles — the
oracle's functions take (profile sequence, other sequence) — at the same <T, N> / preset, so the tie rule of the ends and the
striping of the traceback run over the profile sequence exactly as in the reference."""
import numpy as np
import pytest

from conftest import stable_seed

pytestmark = pytest.mark.gpu
S_ = 0


@pytest.fixture(scope="module")
def za():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    import zoe_amd

    return zoe_amd


def osc(oracle, m, go, ge):
    return oracle.Scoring(m.signed_weights(), m.mapping.index_map, go, ge)


def okey(a):
    return a.key() if a.status == S_ else (a.status, 0, (0, 0), (0, 0), "", 0, 0)


def _reads_like_config1(n, seed, ref, L=150):
    from zoe_amd import synth

    return [bytes(r) for r in synth.reads_host(ref, seed, n, L)]


@pytest.mark.parametrize("T,N", [("i16", 16), ("i8", 32), ("i16", 8), ("i32", 8)])
def test_shared_striped_profile_vs_oracle(za, oracle, T, N):
    """score, ends, ranges and alignment (SeqSrc::Query: the usual call) of 3,000 synthetic reads against the profile of the
    2 kb reference at <T, N>; every read's ends and ranges, every fifth read's CIGAR."""
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    reads = _reads_like_config1(3000, 11, ref)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    prof = za.SharedStripedProfile(ref, dna, -10, -1, T, N)
    s = prof.sw_score(reads)
    e = prof.sw_score_ends(za.SeqBatchSrc.Reference(reads))
    r = prof.sw_score_ranges(za.SeqBatchSrc.Reference(reads))
    rq = prof.sw_score_ranges(za.SeqBatchSrc.Query(reads))
    a = prof.sw_align(za.SeqBatchSrc.Query(reads))
    ar = prof.sw_align(za.SeqBatchSrc.Reference(reads[:300]))
    a3 = prof.sw_align_3pass(za.SeqBatchSrc.Query(reads))
    a3r = prof.sw_align_3pass(za.SeqBatchSrc.Reference(reads[:300]))
    for i, rd in enumerate(reads):
        st, sv = oracle.score(T, N, sc, ref, rd)
        assert (int(s.status[i]), int(s.score[i]) if st == S_ else 0) == (st, sv if st == S_ else 0), i
        st, (sv, re_, qe) = oracle.score_ends(T, N, sc, ref, rd)
        assert int(e.status[i]) == st, i
        if st == S_:
            assert (int(e.score[i]), int(e.ref_end[i]), int(e.query_end[i])) == (sv, re_, qe), i
        st, sv, rr, qr = oracle.score_ranges(T, N, sc, ref, rd)
        assert int(r.status[i]) == st, i
        if st == S_:
            assert (int(r.score[i]), (int(r.ref_start[i]), int(r.ref_end[i])), (int(r.query_start[i]), int(r.query_end[i]))) == (sv, rr, qr), i
            assert ((int(rq.query_start[i]), int(rq.query_end[i])), (int(rq.ref_start[i]), int(rq.ref_end[i]))) == (rr, qr), i
        if i % 5 == 0:
            assert a.key(i) == okey(oracle.align(T, N, sc, ref, rd, other_is_query=True)), i
        if i < 300 and i % 3 == 0:
            assert ar.key(i) == okey(oracle.align(T, N, sc, ref, rd, other_is_query=False)), i
            assert a3r.key(i) == okey(oracle.align_3pass(T, N, sc, ref, rd, other_is_query=False)[0]), i
        if i % 4 == 0:  # profile.rs:536-552 with the shared profile: ranges, then banded / scalar alignment in the box
            assert a3.key(i) == okey(oracle.align_3pass(T, N, sc, ref, rd, other_is_query=True)[0]), i


@pytest.mark.parametrize("preset", [128, 256, 512])
def test_shared_profiles_cascade_vs_oracle(za, oracle, preset):
    """sequence.into_shared_profile(..) semantics at the three presets: sw_score_from_i8, sw_score_ranges_from_i8 and
    sw_align_from_i8(SeqSrc::Query(read)): tiers, ranges and CIGARs of the tier that answered."""
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    reads = _reads_like_config1(1500, 5, ref) + _reads_like_config1(300, 9, ref, L=90)  # the short ones answer at i8
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    prof = za.SharedProfilesBatch(ref, dna, -10, -1, preset) if preset != 256 else za.into_shared_profile(ref, dna, -10, -1)
    s = prof.sw_score_from_i8(reads)
    r = prof.sw_score_ranges_from_i8(za.SeqBatchSrc.Reference(reads))
    a = prof.sw_align_from_i8(za.SeqBatchSrc.Query(reads))
    a3 = prof.sw_align_from_i8_3pass(za.SeqBatchSrc.Query(reads))
    tiers = set()
    for i, rd in enumerate(reads):
        st, sv, tier = oracle.cascade_score(8, preset, sc, ref, rd)
        assert (int(s.status[i]), int(s.tier[i])) == (st, tier), i
        if st == S_:
            assert int(s.score[i]) == sv, i
        tiers.add(tier)
        if i % 4 == 0:
            st, sv, rr, qr, tier = oracle.cascade_score_ranges(8, preset, sc, ref, rd)
            assert int(r.status[i]) == st, i
            if st == S_:
                assert (int(r.score[i]), (int(r.ref_start[i]), int(r.ref_end[i])), (int(r.query_start[i]), int(r.query_end[i])), int(r.tier[i])) == (sv, rr, qr, tier), i
            want, wt = oracle.cascade_align(8, preset, sc, ref, rd, other_is_query=True)
            assert a.key(i) == okey(want), i
            assert int(a.tier[i]) == wt, i
            want3, wt3, _how = oracle.cascade_align_3pass(8, preset, sc, ref, rd, other_is_query=True)  # profile_set.rs:212-283 of SharedProfiles
            assert a3.key(i) == okey(want3), i
            if want3.status == S_:
                assert int(a3.tier[i]) == wt3, i
    assert {8, 16} <= tiers


def test_layout_dependent_pairs_with_the_roles_swapped(za, oracle):
    """SURVEY.md §7 hard part 1 from the other side: the long sequence carries the profile, the short one is walked row by row;
    every lane count must give the oracle's CIGAR of that lane count, in both SeqSrc directions."""
    cases = [
        (4, -2, -3, -1, b"GGACTAAGCTAACACAGGTAGGCTTTATAAAAGGTTAAAGTGCGTGAGCTAGGGTGGCTCTCACT", b"TATAAAAGGTTAAAGTGCTGTAGCTTAGGGTTGCTCTC"),
        (3, -1, -4, -1, b"TGGGGCATTTATGCGATGCAAGACAGGTCTAATATTGAAATTTATTCTAGACTATGCGAGGCCGCTCAAAGGAACCATTACCTTTTTCCGTAGGTCTCCCGATCGCGGCTAACTACTGC", b"ATAGCGATCGCAGCGCCAGGTCT"),
    ]
    for ma, mi, go, ge, long_seq, short_seq in cases:
        m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
        sc = osc(oracle, m, go, ge)
        for N in (2, 4, 8, 16, 32, 64):
            prof = za.SharedStripedProfile(long_seq, m, go, ge, "i16", N)
            for is_query in (False, True):
                src = (za.SeqBatchSrc.Query if is_query else za.SeqBatchSrc.Reference)([short_seq, short_seq[::-1], long_seq[5:40]])
                got = prof.sw_align(src)
                for i, other in enumerate((short_seq, short_seq[::-1], long_seq[5:40])):
                    assert got.key(i) == okey(oracle.align("i16", N, sc, long_seq, other, other_is_query=is_query)), (N, is_query, i)


@pytest.fixture
def any_size(za):
    """the seeded pass for batches of every size (by default batches under 1,024 reads skip it)"""
    from zoe_amd import _lib

    ctx = za.SwContext.get(0)
    ctx.debug_set(_lib.DEBUG_SCORE_PRUNE_ANY_SIZE)
    yield ctx
    ctx.debug_set(0)


@pytest.mark.parametrize("seeded", [False, True])
@pytest.mark.parametrize("scheme", [(4, -2, -3, -1), (2, -5, -10, -1), (3, -1, 0, 0), (1, -1, -1, -1), (5, -4, -2, 0)])
def test_random_pairs_ragged_reads_ties_and_low_complexity(za, oracle, scheme, seeded, request):
    """Ragged batches, reads with N and lower case, low-complexity reads against a low-complexity sequence (ties in every row):
    ends, ranges and CIGARs at <i16, 4 / 16>. `seeded`: the same batch through the role-swapped seeded pass (mode 3: a read whose
    maximum sits in one cell keeps that pass's ends, every other read is recomputed under the shared role's own tie rule)."""
    if seeded:
        request.getfixturevalue("any_size")
    ma, mi, go, ge = scheme
    rng = np.random.default_rng(stable_seed("shared", scheme))
    m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
    sc = osc(oracle, m, go, ge)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = bytes(rng.choice(alpha, 300)) + b"ACACACACACACACAC" + bytes(rng.choice(alpha[:2], 60)) + bytes(rng.choice(alpha, 200))
    reads = []
    for _ in range(150):
        L = int(rng.integers(1, 90))
        t = rng.random()
        if t < 0.6:
            s0 = int(rng.integers(0, len(seq) - L))
            r = bytearray(seq[s0 : s0 + L])
            for _ in range(int(rng.integers(0, 4))):
                k = int(rng.integers(0, len(r)))
                u = rng.random()
                if u < 0.4:
                    r[k] = int(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8)))
                elif u < 0.7 and len(r) > 1:
                    del r[k]
                else:
                    r.insert(k, int(rng.choice(alpha)))
            reads.append(bytes(r))
        elif t < 0.85:
            reads.append(bytes(rng.choice(alpha[:2], L)))
        else:
            reads.append(bytes(rng.choice(alpha, L)))
    for N in (4, 16):
        prof = za.SharedStripedProfile(seq, m, go, ge, "i16", N)
        e = prof.sw_score_ends(za.SeqBatchSrc.Reference(reads))
        r = prof.sw_score_ranges(za.SeqBatchSrc.Reference(reads))
        a = prof.sw_align(za.SeqBatchSrc.Query(reads))
        a3 = prof.sw_align_3pass(za.SeqBatchSrc.Query(reads))
        for i, rd in enumerate(reads):
            st, (sv, re_, qe) = oracle.score_ends("i16", N, sc, seq, rd)
            assert int(e.status[i]) == st, (N, i)
            if st == S_:
                assert (int(e.score[i]), int(e.ref_end[i]), int(e.query_end[i])) == (sv, re_, qe), (N, i)
            st, sv, rr, qr = oracle.score_ranges("i16", N, sc, seq, rd)
            assert int(r.status[i]) == st, (N, i)
            if st == S_:
                assert (int(r.score[i]), (int(r.ref_start[i]), int(r.ref_end[i])), (int(r.query_start[i]), int(r.query_end[i]))) == (sv, rr, qr), (N, i)
            assert a.key(i) == okey(oracle.align("i16", N, sc, seq, rd, other_is_query=True)), (N, i)
            assert a3.key(i) == okey(oracle.align_3pass("i16", N, sc, seq, rd, other_is_query=True)[0]), (N, i)


def _tie_rich_reads(rng, seq, n, L):
    """reads of L bases built to hold their maximum in several cells, or in one cell the two tie rules would not both pick first:
    pieces inside tandem repeats (equal scores on neighbouring diagonals), two pieces of equal length from different places in
    either order (first row of the sequence vs first row of the read), a piece followed by two mismatches and five matches (the
    peak is reached twice on one diagonal), plus ordinary reads with a few edits"""
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    sa = np.frombuffer(seq, dtype=np.uint8)
    R = len(seq)
    out = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        kind = i % 6
        p = int(rng.integers(0, R - L))
        r = sa[p:p + L].copy()
        if kind == 1:  # two halves from different places
            q = int(rng.integers(0, R - L))
            h = L // 2
            r = np.concatenate([sa[p:p + h], sa[q:q + L - h]])
        elif kind == 2:  # peak reached twice on the diagonal
            for k in (L - 7, L - 6):
                r[k] = alpha[(int(np.where(alpha == r[k])[0][0]) + 1) % 4]
        elif kind == 3:  # a few edits
            for _ in range(int(rng.integers(1, 5))):
                r[int(rng.integers(0, L))] = rng.choice(alpha)
        elif kind == 4:  # junk start, then a piece
            j = int(rng.integers(5, L // 2))
            r[:j] = rng.choice(alpha, j)
        elif kind == 5:
            r = rng.choice(alpha[:2], L)
        out[i] = r
    return out


@pytest.mark.parametrize("T,N", [("i16", 16), ("i8", 32)])
def test_ends_through_the_seeded_pass_on_tie_rich_reads(za, oracle, T, N):
    """1,500 reads of 150 bases (the default path: batches of 1,024 reads or more take the role-swapped seeded pass) against a
    sequence with tandem repeats and a duplicated stretch: ends, ranges of every read, every eighth CIGAR, against the oracle with
    the same roles"""
    import torch

    rng = np.random.default_rng(stable_seed("tie-rich", T, N))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = [bytes(rng.choice(alpha, 400)), b"ACG" * 30, bytes(rng.choice(alpha, 300)), b"TTGACA" * 20, bytes(rng.choice(alpha, 500)), b"AC" * 50]
    dup = bytes(rng.choice(alpha, 200))
    seq = b"".join(parts) + dup + bytes(rng.choice(alpha, 150)) + dup + bytes(rng.choice(alpha, 100))
    reads2d = _tie_rich_reads(rng, seq, 1500, 150)
    reads = [bytes(r) for r in reads2d]
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    rb = za.ReadBatch.from_fixed(torch.from_numpy(reads2d.reshape(-1)).cuda(), 150)
    prof = za.SharedStripedProfile(seq, dna, -10, -1, T, N)
    e = prof.sw_score_ends(za.SeqBatchSrc.Reference(rb))
    r = prof.sw_score_ranges(za.SeqBatchSrc.Query(rb))
    a = prof.sw_align(za.SeqBatchSrc.Query(rb))
    for i, rd in enumerate(reads):
        st, (sv, re_, qe) = oracle.score_ends(T, N, sc, seq, rd)
        assert int(e.status[i]) == st, i
        if st == S_:
            assert (int(e.score[i]), int(e.ref_end[i]), int(e.query_end[i])) == (sv, re_, qe), i
        st, sv, rr, qr = oracle.score_ranges(T, N, sc, seq, rd)
        assert int(r.status[i]) == st, i
        if st == S_:  # SeqSrc::Query: the ranges change names
            assert (int(r.score[i]), (int(r.ref_start[i]), int(r.ref_end[i])), (int(r.query_start[i]), int(r.query_end[i]))) == (sv, qr, rr), i
        if i % 8 == 0:
            assert a.key(i) == okey(oracle.align(T, N, sc, seq, rd, other_is_query=True)), i


def test_seeded_and_plain_shared_ends_agree_on_a_large_batch(za):
    """300,000 synthetic reads + tie-rich ones: the default path (role-swapped seeded pass + the exact kernel on what it cannot
    settle; for the ranges a second seeded pass over the reversed sequences) against the exact shared-role kernels over every read
    (ZSW_OPTION_EXACT_PRUNING off)"""
    import torch

    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    host = synth.reads_host(ref, 5, 300_000, 150)
    rng = np.random.default_rng(77)
    host[::50] = _tie_rich_reads(rng, bytes(ref), len(host[::50]), 150)
    rb = za.ReadBatch.from_fixed(torch.from_numpy(host.reshape(-1)).cuda(), 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    prof = za.SharedStripedProfile(ref, dna, -10, -1, "i16", 16)
    got = prof.sw_score_ends(za.SeqBatchSrc.Reference(rb))
    got_r = prof.sw_score_ranges(za.SeqBatchSrc.Reference(rb))  # second pass: the seeded pass over the reversed sequences + the exact kernel
    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)
    try:
        want = prof.sw_score_ends(za.SeqBatchSrc.Reference(rb))
        want_r = prof.sw_score_ranges(za.SeqBatchSrc.Reference(rb))
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    for name in ("score", "status", "ref_end", "query_end"):
        assert torch.equal(getattr(got, name), getattr(want, name)), name
    for name in ("score", "status", "ref_start", "ref_end", "query_start", "query_end"):
        assert torch.equal(getattr(got_r, name), getattr(want_r, name)), "ranges " + name


def test_long_profile_sequence_takes_several_tiles_and_rows_in_hbm(za, oracle):
    """A 5 kb profile sequence: three 2,048-column tiles in the ends kernel, nv = 313 at <i16, 16> (DP rows behind the flag ring)."""
    from zoe_amd import synth

    seq = synth.reference_host(5000)
    reads = _reads_like_config1(200, 3, seq, L=120)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    prof = za.SharedStripedProfile(seq, dna, -10, -1, "i16", 16)
    r = prof.sw_score_ranges(za.SeqBatchSrc.Reference(reads))
    a = prof.sw_align(za.SeqBatchSrc.Query(reads))
    for i, rd in enumerate(reads):
        st, sv, rr, qr = oracle.score_ranges("i16", 16, sc, seq, rd)
        assert int(r.status[i]) == st, i
        if st == S_:
            assert (int(r.score[i]), (int(r.ref_start[i]), int(r.ref_end[i])), (int(r.query_start[i]), int(r.query_end[i]))) == (sv, rr, qr), i
        if i % 4 == 0:
            assert a.key(i) == okey(oracle.align("i16", 16, sc, seq, rd, other_is_query=True)), i


def test_empty_inputs_and_errors(za, oracle):
    import torch

    from zoe_amd import _lib

    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    with pytest.raises(za.ProfileError):
        za.SharedStripedProfile(b"", dna, -10, -1)  # StripedProfile::new(empty) -> ProfileError::EmptySequence
    prof = za.SharedStripedProfile(b"ACGTACGTTTGACA", dna, -10, -1, "i16", 16)
    # an empty read is an empty `reference` argument: Unmapped (striped.rs:219-221), in a ragged batch with ordinary reads
    reads = [b"ACGTTTG", b"", b"TTTT"]
    off = torch.tensor([0, 7, 7, 11], dtype=torch.int64, device="cuda")
    rb = za.ReadBatch(torch.frombuffer(bytearray(b"".join(reads)), dtype=torch.uint8).cuda(), 3, offsets=off)
    s = prof.sw_score(rb)
    e = prof.sw_score_ends(za.SeqBatchSrc.Reference(rb))
    a = prof.sw_align(za.SeqBatchSrc.Query(rb))
    sc = osc(oracle, dna, -10, -1)
    want = [2 if not rd else oracle.score("i16", 16, sc, b"ACGTACGTTTGACA", rd)[0] for rd in reads]
    assert want == [0, 2, 0] and [int(x) for x in s.status] == want
    assert int(e.status[1]) == 2 and int(a.status[1]) == 2
    a3 = prof.sw_align_3pass(za.SeqBatchSrc.Query(rb))
    assert [int(x) for x in a3.status] == want
    assert a.key(0) == okey(oracle.align("i16", 16, sc, b"ACGTACGTTTGACA", reads[0], other_is_query=True))
    ctx = za.SwContext.get(0)
    fresh = _lib.load()
    import ctypes as C

    h = C.c_void_p()
    assert fresh.zsw_create(0, C.byref(h)) == 0
    try:  # the shared entry points need a profile sequence
        b = rb.c_batch()
        sc_t = torch.zeros(3, dtype=torch.int32, device="cuda")
        st_t = torch.zeros(3, dtype=torch.uint8, device="cuda")
        assert fresh.zsw_score_shared_batch(h, C.byref(b), 1, 16, sc_t.data_ptr(), st_t.data_ptr(), None) == -5  # ZSW_ERR_NOT_CONFIGURED
    finally:
        fresh.zsw_destroy(h)


@pytest.mark.parametrize("T,N", [("i16", 16), ("i8", 32)])
def test_three_pass_with_the_shared_profile_on_tie_rich_and_diverged_reads(za, oracle, T, N, any_size):
    """SharedProfiles::sw_align_from_i*_3pass / StripedProfile::sw_align_3pass with ONE profile (profile_set.rs:212-283, 552-560;
    profile.rs:536-552 -> three_pass.rs:21-104): the shared ranges (seeded passes, forward and reversed), then the third pass with
    `reference` = the read and a ScalarProfile over the profile sequence's sub-range. Tie-rich reads (several optimal alignments: the
    band / scalar tie-breaking of banded.rs / scalar.rs decides) and reads 5-12 % away from the sequence, both SeqSrc directions."""
    from test_gpu_bounds import diverged_reads
    from zoe_amd import synth

    rng = np.random.default_rng(stable_seed("shared3", T, N))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = bytearray(rng.choice(alpha, 1500).tobytes())
    seq[300:340] = bytes(rng.choice(alpha[:2], 40))
    for i in range(700 + 3, 760):
        seq[i] = seq[i - 3]
    seq = bytes(seq)
    reads = np.concatenate([_tie_rich_reads(rng, seq, 900, 100), diverged_reads(seq, 300, 100, 50, 1), diverged_reads(seq, 300, 100, 120, 2)])
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    prof = za.SharedStripedProfile(seq, dna, -10, -1, T, N)
    import torch

    rb = za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda(), 100)
    aq = prof.sw_align_3pass(za.SeqBatchSrc.Query(rb))
    ar = prof.sw_align_3pass(za.SeqBatchSrc.Reference(rb))
    for i in range(len(reads)):
        assert aq.key(i) == okey(oracle.align_3pass(T, N, sc, seq, reads[i], other_is_query=True)[0]), i
        if i % 3 == 0:
            assert ar.key(i) == okey(oracle.align_3pass(T, N, sc, seq, reads[i], other_is_query=False)[0]), i


@pytest.mark.parametrize("T,N", [("i16", 16), ("i8", 32)])
def test_exact_alignment_with_the_shared_profile_skips_the_literal_pass_where_one_alignment_is_optimal(za, oracle, T, N):
    """run_align_shared in certificate mode (zsw_capi_shared.hip; tests/models/align_gapless_cert.cpp and align_onegap_cert.cpp check
    the swapped roles against the literal oracle): the two seeded passes of the shared ranges tell which reads have both maxima in one
    cell each; those whose only optimal alignment is gapless or has one gap run get it from the classify pass, the others from the
    literal striped recurrence over the shared sequence's profile. 12,000 reads — synthetic (indels in homopolymer runs: tied
    placements), tie-rich, 3-8 % diverged: every record and ciglet must equal the all-literal path (ZSW_DEBUG_ALIGN_NO_CERTIFICATE), both
    SeqSrc directions, and the oracle's literal sw_simd_align on a sample; most reads must have skipped the literal pass."""
    import time

    import torch

    from test_gpu_bounds import diverged_reads
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    rng = np.random.default_rng(stable_seed("shared-cert", T, N))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = bytearray(synth.reference_host(1800))
    seq[300:340] = bytes(rng.choice(alpha[:2], 40))
    for i in range(700 + 3, 760):
        seq[i] = seq[i - 3]
    seq[1200:1260] = seq[100:160]
    seq = bytes(seq)
    L = 120
    reads = np.concatenate([synth.reads_host(seq, 5, 8000, L), _tie_rich_reads(rng, seq, 2000, L), diverged_reads(seq, 1000, L, 30, 3),
                            diverged_reads(seq, 1000, L, 80, 4)])
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    prof = za.SharedStripedProfile(seq, dna, -10, -1, T, N)
    rb = za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda(), L)

    def timed(src):
        prof.sw_align(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = prof.sw_align(src)
        torch.cuda.synchronize()
        return a, time.perf_counter() - t0

    aq, t_cert = timed(za.SeqBatchSrc.Query(rb))
    ar, _ = timed(za.SeqBatchSrc.Reference(rb))
    ctx.debug_set(_lib.DEBUG_ALIGN_NO_CERTIFICATE)
    try:
        lq, t_lit = timed(za.SeqBatchSrc.Query(rb))
        lr, _ = timed(za.SeqBatchSrc.Reference(rb))
    finally:
        ctx.debug_set(0)
    for got, want in ((aq, lq), (ar, lr)):
        assert np.array_equal(got.status, want.status)
        assert np.array_equal(got.records, want.records)
        assert np.array_equal(got.inc, want.inc) and np.array_equal(got.op, want.op)
    assert t_cert < 0.6 * t_lit, (t_cert, t_lit)
    for i in range(0, len(reads), 37):
        assert aq.key(i) == okey(oracle.align(T, N, sc, seq, reads[i], other_is_query=True)), i
