"""GPU parity tests for the score / score+ends kernels, called through the C ABI (via zoe_amd.alignment).

Checker = oracle/ (CPU restatement of the reference, pinned by tests/test_oracle_golden.py).
Bar: bit-exact score, status (and ends) for every read.
"""
import numpy as np
import pytest

from conftest import stable_seed

pytestmark = pytest.mark.gpu

S_, O_, U_, E_ = 0, 1, 2, 3


@pytest.fixture(scope="module")
def za():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    import zoe_amd

    zoe_amd.SwContext.get(0).selftest()
    return zoe_amd


@pytest.fixture(scope="module")
def dna(za):
    return za.WeightMatrix.new_dna_matrix(2, -5, b"N")


def osc(oracle, m, go, ge):
    return oracle.Scoring(m.signed_weights(), m.mapping.index_map, go, ge)


def test_selftest(za):
    za.SwContext.get(0).selftest()


def test_synth_device_matches_host(za):
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    ctx = za.SwContext.get(0)
    rb = synth.reads_device(ctx, ref, 12345, 4096, 150)
    host = synth.reads_host(ref, 12345, 4096, 150)
    assert np.array_equal(rb.bases.cpu().numpy().reshape(4096, 150), host)
    rr = synth.reads_ragged_device(ctx, ref, 77, 2000, 75, 400)
    hb, hoff = synth.reads_ragged_host(ref, 77, 2000, 75, 400)
    assert np.array_equal(rr.offsets.cpu().numpy(), hoff)
    assert np.array_equal(rr.bases.cpu().numpy()[: len(hb)], hb)


def test_config1_10k_reads_i16_and_cascade(za, oracle, dna):
    """BASELINE.json configs[0] inputs: 10k synthetic 150 bp reads vs one 2 kb reference, default matrix."""
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    n = 10000
    host = synth.reads_host(ref, 0, n, 150)
    sc = osc(oracle, dna, -10, -1)
    want_s, want_st, want_tier = oracle.batch_score_w256(8, sc, host, ref, fixed_len=150, threads=8)
    rb = za.ReadBatch.from_fixed(__import__("torch").from_numpy(host.reshape(-1)).cuda(), 150)
    got = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1).sw_score_from_i8(ref)
    assert np.array_equal(got.status.cpu().numpy(), want_st)
    assert np.array_equal(got.score.cpu().numpy().view(np.uint32), want_s)
    assert np.array_equal(got.tier.cpu().numpy(), want_tier)
    got16 = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16).sw_score(ref)
    assert np.array_equal(got16.score.cpu().numpy().view(np.uint32), want_s)
    assert np.array_equal(got16.status.cpu().numpy(), want_st)
    # i8 direct: Overflowed exactly where the true score >= 255 (striped.rs:619)
    got8 = za.StripedProfileBatch(rb, dna, -10, -1, "i8", 32).sw_score(ref)
    st8 = got8.status.cpu().numpy()
    assert np.array_equal(st8 == O_, want_s >= 255)
    # the generic plain-array oracle on a slice (every <T,N>)
    for i in range(0, 64):
        assert (int(want_st[i]), int(want_s[i])) == oracle.score("i16", 16, sc, host[i], ref)


@pytest.mark.parametrize("T,N", [("i8", 32), ("i16", 16), ("i32", 8), ("u8", 32), ("u16", 16), ("u32", 8)])
def test_all_int_types_vs_oracle(za, oracle, dna, T, N):
    from zoe_amd import synth

    ref = synth.reference_host(500)
    host = synth.reads_host(ref, 500, 300, 60)
    # make some reads score low/high: truncate similarity by randomising tails
    rng = np.random.default_rng(5)
    for i in range(0, 300, 3):
        k = int(rng.integers(5, 60))
        host[i, k:] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 60 - k)
    sc = osc(oracle, dna, -10, -1)
    m = dna if T[0] == "i" else dna.to_biased_matrix()
    got = za.StripedProfileBatch([bytes(r) for r in host], m, -10, -1, T, N).sw_score(ref)
    st, s = got.status.cpu().numpy(), got.score.cpu().numpy().view(np.uint32)
    for i in range(300):
        o_st, o_s = oracle.score(T, N, sc, host[i], ref)
        assert (int(st[i]), int(s[i]) if st[i] == S_ else 0) == (o_st, o_s if o_st == S_ else 0), (i, T)


def test_known_answers_through_gpu(za, oracle, h1, h5, cy):
    """The reference's own vectors, run through the HIP path (src/alignment/sw/test.rs)."""
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    # :103-114 profile = H5 (1,760 bp), other = H1 -> Some(37); u8x16 and i16x16
    assert za.StripedProfileBatch([h5], dna.to_biased_matrix(), -10, -1, "u8", 16).sw_score(h1).maybe_aligned(0) == ("Some", 37)
    assert za.StripedProfileBatch([h5], dna, -10, -1, "i16", 16).sw_score(h1).maybe_aligned(0) == ("Some", 37)
    # :88-100
    r = za.StripedProfileBatch([b"ACGTUNacgtun"], dna, -10, -1, "i16", 16).sw_score(b"ACGTTNACGTTN")
    assert r.maybe_aligned(0) == ("Some", 20)
    # :265-271, :274-280, :304-311
    v = b"A" * 100
    assert za.StripedProfileBatch([v], dna.to_biased_matrix(), -10, -1, "u16", 16).sw_score(v).maybe_aligned(0) == ("Some", 200)
    assert za.StripedProfileBatch([cy], dna.to_biased_matrix(), -10, -1, "u16", 16).sw_score(cy).maybe_aligned(0) == ("Some", 3372)
    r = za.LocalProfilesBatch.new_with_w128([cy], dna, -10, -1).sw_score_from_i8(cy)
    assert r.maybe_aligned(0) == ("Some", 3372) and int(r.tier[0]) == 16
    # :283-290 lazy-F regression
    m = za.WeightMatrix.new(za.DNA_PROFILE_MAP, 10, -10, b"N").to_biased_matrix()
    assert za.StripedProfileBatch([b"AGA"], m, -5, -5, "u16", 4).sw_score(b"AA").maybe_aligned(0) == ("Some", 15)
    # :293-301 overflow
    m = za.WeightMatrix.new(za.DNA_PROFILE_MAP, 127, 0, b"N").to_biased_matrix()
    assert za.StripedProfileBatch([b"AAAA"], m, -10, -1, "u8", 8).sw_score(b"AAAA").maybe_aligned(0) == ("Overflowed", None)
    # striped.rs:45-55
    m = za.WeightMatrix.new_biased_dna_matrix(4, -2, b"N")
    r = za.StripedProfileBatch([b"CGTTCGCCATAAAGGGGG"], m, -3, -1, "u8", 32).sw_score(b"ATGCATCGATCGATCGATCGATCGATCGATGC")
    assert r.maybe_aligned(0) == ("Some", 26)
    # sw/mod.rs:193-218 custom alphabet (S = 4, catch-all A)
    mp = za.ByteIndexMap.new(b"ABCD", b"A")
    m = za.WeightMatrix.new(mp, 1, -1, None)
    assert za.StripedProfileBatch([b"AABDDAB"], m, -4, -2, "i8", 32).sw_score(b"BDAACAABDDDB").maybe_aligned(0) == ("Some", 5)


def test_score_ends_vs_oracle(za, oracle, dna):
    from zoe_amd import synth

    ref = synth.reference_host(700)
    host = synth.reads_host(ref, 9000, 400, 90)
    host[5] = np.frombuffer(b"N" * 90, dtype=np.uint8)
    sc = osc(oracle, dna, -10, -1)
    got = za.StripedProfileBatch([bytes(r) for r in host], dna, -10, -1, "i16", 16).sw_score_ends(za.SeqSrc.Reference(ref))
    st, s = got.status.cpu().numpy(), got.score.cpu().numpy()
    re_, qe = got.ref_end.cpu().numpy(), got.query_end.cpu().numpy()
    for i in range(400):
        o_st, (o_s, o_re, o_qe) = oracle.score_ends("i16", 16, sc, host[i], ref)
        assert int(st[i]) == o_st, i
        if o_st == S_:
            assert (int(s[i]), int(re_[i]), int(qe[i])) == (o_s, o_re, o_qe), i
    # low-complexity ties: first row, then first column (striped.rs:312-321 == scalar.rs:207-211)
    reads = [b"TTTTTTTT", b"ACACACAC", b"GGGGG", b"CCCCA", b"TTTAG"]
    refs = b"TTTTTTTTTTTTACACACACACACACGGGGGGGGTAAAACCCC"
    got = za.StripedProfileBatch(reads, dna, -10, -1, "i8", 8).sw_score_ends(za.SeqSrc.Reference(refs))
    for i, rd in enumerate(reads):
        o_st, (o_s, o_re, o_qe) = oracle.score_ends("i8", 8, osc(oracle, dna, -10, -1), rd, refs)
        assert (int(got.status[i]), int(got.score[i]), int(got.ref_end[i]), int(got.query_end[i])) == (o_st, o_s, o_re, o_qe)


def test_ragged_and_edge_cases(za, oracle, dna):
    from zoe_amd import synth

    ref = synth.reference_host(1000)
    hb, hoff = synth.reads_ragged_host(ref, 3, 257, 20, 300)  # odd count: last lane pair half empty
    reads = [hb[hoff[i] : hoff[i + 1]].tobytes() for i in range(257)]
    reads[10] = b"N" * 40  # unmapped
    reads[11] = b"acgu" * 10  # lower case + U
    reads[12] = bytes(range(256))  # arbitrary bytes -> catch-all
    sc = osc(oracle, dna, -10, -1)
    got = za.LocalProfilesBatch.new_with_w256(reads, dna, -10, -1).sw_score_from_i8(ref)
    st, s = got.status.cpu().numpy(), got.score.cpu().numpy()
    for i, rd in enumerate(reads):
        o_st, o_s, o_tier = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(st[i]), int(s[i]) if o_st == S_ else 0, int(got.tier[i])) == (o_st, o_s if o_st == S_ else 0, o_tier), i
    # empty reference -> Unmapped (striped.rs:140-141 with best = MIN)
    got = za.StripedProfileBatch(reads[:8], dna, -10, -1, "i16", 16).sw_score(b"")
    assert (got.status.cpu().numpy() == U_).all()
    # an empty read is ProfileError::EmptySequence (profile.rs:33-34)
    with pytest.raises(za.ProfileError) as ei:
        za.StripedProfileBatch([b"ACGT", b""], dna, -10, -1, "i16", 16)
    assert ei.value.variant == "EmptySequence"
    for go, ge, variant in ((1, 0, "GapOpenOutOfRange"), (-10, 1, "GapExtendOutOfRange"), (-1, -2, "BadGapWeights")):
        with pytest.raises(za.ProfileError) as ei:
            za.StripedProfileBatch([b"ACGT"], dna, go, ge, "i16", 16)
        assert ei.value.variant == variant


def test_generic_matrix_and_i16_saturation(za, oracle):
    """A matrix whose N column is not zero takes the biased-table kernel; weights of 127 push scores past
    i16 so the exact 32-bit kernel answers (cascade tier 32)."""
    rng = np.random.default_rng(11)
    w = rng.integers(-9, 10, size=(5, 5)).astype(np.int8)
    m = za.WeightMatrix.new_custom(za.DNA_PROFILE_MAP, w)
    ref = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 400))
    reads = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(10, 150)))) for _ in range(200)]
    sc = oracle.Scoring(w, za.DNA_PROFILE_MAP.index_map, -4, -2)
    got = za.LocalProfilesBatch.new_with_w256(reads, m, -4, -2).sw_score_from_i8(ref)
    for i, rd in enumerate(reads):
        o_st, o_s, o_tier = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]) if o_st == S_ else 0, int(got.tier[i])) == (o_st, o_s if o_st == S_ else 0, o_tier), i
    big = za.WeightMatrix.new_dna_matrix(127, -127, b"N")
    long_read = b"ACGT" * 150  # 600 x 127 = 76,200 > 65,535
    refl = b"TT" + long_read + b"GG"
    r = za.LocalProfilesBatch.new_with_w256([long_read, b"ACGTACGT"], big, -10, -1).sw_score_from_i8(refl)
    assert r.maybe_aligned(0) == ("Some", 600 * 127) and int(r.tier[0]) == 32
    assert r.maybe_aligned(1) == ("Some", 8 * 127) and int(r.tier[1]) == 16
    r = za.StripedProfileBatch([long_read], big, -10, -1, "i16", 16).sw_score(refl)
    assert r.maybe_aligned(0) == ("Overflowed", None)


def _protein_case(za, seed, S=25, lo=-4, hi=12):
    rng = np.random.default_rng(seed)
    keys = b"ACDEFGHIKLMNPQRSTVWYBJZX*"[:S] if S <= 25 else bytes(range(65, 65 + S))
    mp = za.ByteIndexMap.new(keys, keys[-1:])
    w = rng.integers(lo, 3, size=(S, S))
    w = np.minimum(w, w.T)
    np.fill_diagonal(w, rng.integers(4, hi, size=S))
    w = w.astype(np.int8)
    return rng, keys, mp, w, za.WeightMatrix.new_custom(mp, w)


@pytest.mark.parametrize("wide", [True, False])
def test_protein_alphabet_wide_and_exact_kernels(za, oracle, debug, wide):
    """25-letter alphabets (the reference's BLOSUM matrices are WeightMatrix<i8, 25>, src/data/matrices/aa.rs) run on the
    WIDE packed kernels (zsw_score_wide.hip); ZSW_DEBUG_NO_WIDE forces the exact 32-bit kernel. Both must equal the oracle:
    score, ends and ranges, fixed-length and ragged batches."""
    if not wide:
        debug.set(debug.NO_WIDE)
    rng, keys, mp, w, m = _protein_case(za, 3)
    alpha = np.frombuffer(keys[:20], dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 700))
    sc = oracle.Scoring(w, mp.index_map, -11, -1)
    reads = []
    for i in range(150):
        L = int(rng.integers(5, 200))
        if i % 3 == 0:  # a mutated piece of the reference
            s0 = int(rng.integers(0, 700 - L))
            r = np.frombuffer(ref[s0:s0 + L], dtype=np.uint8).copy()
            for _ in range(L // 8):
                r[int(rng.integers(0, L))] = rng.choice(alpha)
            reads.append(r.tobytes())
        else:
            reads.append(bytes(rng.choice(np.frombuffer(keys, dtype=np.uint8), L)))
    reads.append(b"acdxyz??")  # lower case / unknown bytes go to the catch-all
    p = za.StripedProfileBatch(reads, m, -11, -1, "i16", 16)
    got = p.sw_score(ref)
    ends = p.sw_score_ends(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        o_st, o_s = oracle.score("i16", 16, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]) if o_st == S_ else 0) == (o_st, o_s if o_st == S_ else 0), i
        e_st, (e_s, e_r, e_q) = oracle.score_ends("i16", 16, sc, rd, ref)
        if e_st == S_:
            assert (int(ends.score[i]), int(ends.ref_end[i]), int(ends.query_end[i])) == (e_s, e_r, e_q), i
    fixed = [bytes(rng.choice(alpha, 150)) for _ in range(64)] + [ref[100:250], ref[300:450]]
    lp = za.LocalProfilesBatch.new_with_w256(fixed, m, -11, -1)
    gf = lp.sw_score_from_i8(ref)
    tiers = set()
    for i, rd in enumerate(fixed):
        o_st, o_s, o_t = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(gf.status[i]), int(gf.score[i]) if o_st == S_ else 0, int(gf.tier[i])) == (o_st, o_s if o_st == S_ else 0, o_t), i
        tiers.add(o_t)
    assert tiers == {8, 16}
    if wide:  # ranges and alignment on top of the wide pass 1
        rg = p.sw_score_ranges(za.SeqSrc.Reference(ref))
        al = za.StripedProfileBatch(reads[:40], m, -11, -1, "i16", 16).sw_align(za.SeqSrc.Reference(ref))
        for i, rd in enumerate(reads[:40]):
            o_st, o_s, o_rr, o_qr = oracle.score_ranges("i16", 16, sc, rd, ref)
            if o_st == S_:
                assert (int(rg.score[i]), (int(rg.ref_start[i]), int(rg.ref_end[i])), (int(rg.query_start[i]), int(rg.query_end[i]))) == (o_s, o_rr, o_qr), i
            want = oracle.align("i16", 16, sc, rd, ref)
            assert al.key(i) == (want.key() if want.status == S_ else (want.status, 0, (0, 0), (0, 0), "", 0, 0)), i


def test_wide_kernel_32_letters_and_extreme_weights(za, oracle):
    """The full 32-letter table and scores at the edges of the signed byte (score + gap_extend in [-128, 127])."""
    rng, keys, mp, w, m = _protein_case(za, 11, S=32, lo=-100, hi=100)
    w[0, 1] = w[1, 0] = -127
    w[2, 2] = 120
    m = za.WeightMatrix.new_custom(mp, w)
    alpha = np.frombuffer(keys, dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 400))
    reads = [bytes(rng.choice(alpha, int(rng.integers(3, 120)))) for _ in range(80)] + [ref[50:120], bytes([keys[2]]) * 90]
    sc = oracle.Scoring(w, mp.index_map, -20, -1)
    got = za.LocalProfilesBatch.new_with_w256(reads, m, -20, -1).sw_score_from_i8(ref)
    for i, rd in enumerate(reads):
        o_st, o_s, o_t = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]) if o_st == S_ else 0) == (o_st, o_s if o_st == S_ else 0), i


def test_config5_mixed_lengths_30kb_reference(za, oracle, dna):
    """BASELINE.json configs[4] shape at test size: reads 75-400 bp vs a 30 kb reference, bucketed launches
    (one per strip configuration) + a few reads longer than every configuration (exact kernel)."""
    import torch

    from zoe_amd import synth

    ref = synth.reference_host(30000)
    n = 1200
    hb, hoff = synth.reads_ragged_host(ref, 0, n, 75, 400)
    extra = [ref[1000:3600], ref[20000:22500] + b"ACGT" * 10]  # 2,600 and 2,540 bp: beyond the 2,432-column strips
    bases = np.concatenate([hb, np.frombuffer(b"".join(extra), dtype=np.uint8)])
    off = np.concatenate([hoff, hoff[-1] + np.cumsum([len(e) for e in extra])]).astype(np.int64)
    sc = osc(oracle, dna, -10, -1)
    ws, wst, wt = oracle.batch_score_w256(8, sc, bases, ref, offsets=off.astype(np.uint64), threads=16)
    rb = za.ReadBatch(torch.from_numpy(bases).cuda(), n + 2, offsets=torch.from_numpy(off).cuda(), min_len=75)
    got = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1).sw_score_from_i8(ref)
    assert np.array_equal(got.status.cpu().numpy(), wst)
    assert np.array_equal(got.score.cpu().numpy().view(np.uint32), ws)
    assert np.array_equal(got.tier.cpu().numpy(), wt)
    assert int(got.score[n]) == 5200 and int(got.tier[n]) == 16
    ends = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16).sw_score_ends(za.SeqSrc.Reference(ref))
    for i in list(range(0, n, 97)) + [n, n + 1]:
        st, (s, re_, qe) = oracle.score_ends("i16", 16, sc, bases[off[i] : off[i + 1]], ref)
        assert (int(ends.status[i]), int(ends.score[i]), int(ends.ref_end[i]), int(ends.query_end[i])) == (st, s, re_, qe), i


def test_config5_full_size_1m_mixed_reads_properties(za, oracle, dna, debug):
    """BASELINE.json configs[4] at its full size (1 M reads of 75-400 bp vs a 30 kb reference, one launch per strip
    configuration), through size-independent properties: the two independent kernels (v2 drift-domain / v1 saturating i16)
    agree on every read, every length class is occupied, bounds hold, a second run is identical, and a random sample equals
    the oracle."""
    import torch

    from zoe_amd import synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(30000)
    n = 1_000_000
    rb = synth.reads_ragged_device(ctx, ref, 0, n, 75, 400)
    lens = (rb.offsets[1:] - rb.offsets[:-1])
    assert int(lens.min()) == 75 and int(lens.max()) == 400
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    debug.set(0)
    a = prof.sw_score_from_i8(ref)
    assert 0 < ctx.prune_rescored() < n // 20  # the default path: the seeded exact pass, per length class (r03)
    s2, st2, t2 = a.score.clone(), a.status.clone(), a.tier.clone()
    b = prof.sw_score_from_i8(ref)
    assert torch.equal(b.score, s2) and torch.equal(b.status, st2)
    debug.set(debug.SCORE_V1)
    c = prof.sw_score_from_i8(ref)
    assert ctx.prune_rescored() == 0  # v1: every cell of every read
    debug.set(0)
    assert torch.equal(c.score, s2) and torch.equal(c.status, st2) and torch.equal(c.tier, t2)  # v1 == default on all reads
    assert int(s2.min()) >= 0 and bool((s2.to(torch.int64) <= 2 * lens).all())
    assert torch.equal(st2 == 0, s2 > 0) and torch.equal(t2[st2 == 0] == 8, s2[st2 == 0] < 255)
    sc = osc(oracle, dna, -10, -1)
    rng = np.random.default_rng(9)
    idx = np.sort(rng.choice(n, 400, replace=False))
    off = rb.offsets.cpu().numpy()
    bases = rb.bases.cpu().numpy()
    sample = np.concatenate([bases[off[i] : off[i + 1]] for i in idx])
    soff = np.concatenate([[0], np.cumsum([off[i + 1] - off[i] for i in idx])]).astype(np.uint64)
    ws, wst, wt = oracle.batch_score_w256(8, sc, sample, ref, offsets=soff, threads=16)
    sel = torch.from_numpy(idx).cuda()
    assert np.array_equal(st2[sel].cpu().numpy(), wst) and np.array_equal(s2[sel].cpu().numpy().view(np.uint32), ws)
    assert np.array_equal(t2[sel].cpu().numpy(), wt)


def test_v1_and_v2_score_kernels_agree(za, oracle, dna, debug):
    """score_kernel_v2 (row-drifted domain + v_pk_maximum3_f16) and score_kernel (saturating packed i16) are both
    bit-exact: same 10k-read batch through each, scores / ends compared with the oracle."""
    import torch

    from zoe_amd import synth

    ref = synth.reference_host(2000)
    host = synth.reads_host(ref, 50_000, 6000, 150)
    sc = osc(oracle, dna, -10, -1)
    ws, wst, wt = oracle.batch_score_w256(8, sc, host, ref, fixed_len=150, threads=16)
    rb = za.ReadBatch.from_fixed(torch.from_numpy(host.reshape(-1)).cuda(), 150)
    for force_v1 in (False, True):
        debug.set(debug.SCORE_V1 if force_v1 else 0)
        got = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1).sw_score_from_i8(ref)
        assert np.array_equal(got.score.cpu().numpy().view(np.uint32), ws), force_v1
        assert np.array_equal(got.status.cpu().numpy(), wst) and np.array_equal(got.tier.cpu().numpy(), wt)
        ends = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16).sw_score_ends(za.SeqSrc.Reference(ref))
        for i in range(0, 6000, 211):
            st, (s, re_, qe) = oracle.score_ends("i16", 16, sc, host[i], ref)
            assert (int(ends.status[i]), int(ends.score[i]), int(ends.ref_end[i]), int(ends.query_end[i])) == (st, s, re_, qe), (force_v1, i)


@pytest.mark.parametrize("scheme", [(2, -5, -10, -1), (100, -90, -120, -100), (3, -4, -6, 0), (127, -128, -127, -127), (5, -3, -1, -1), (1, -1, 0, 0)])
def test_v2_ranges_gap_extremes_and_limit(za, oracle, scheme):
    """Large gap_extend shortens v2's re-base period; gap_extend = 0 disables the drift; match = 127 is outside v2's
    signed-byte table (v1 answers); high scores cross v2's representable limit and go to the exact kernel."""
    ma, mi, go, ge = scheme
    m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
    sc = osc(oracle, m, go, ge)
    rng = np.random.default_rng(stable_seed(scheme))
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    ref = bytes(rng.choice(alpha[:4], 2600))
    reads = []
    for _ in range(150):
        L = int(rng.integers(20, 400))
        s0 = int(rng.integers(0, 2600 - L))
        r = bytearray(ref[s0 : s0 + L])
        for _ in range(int(rng.integers(0, 6))):
            k = int(rng.integers(0, len(r)))
            t = rng.random()
            if t < 0.5:
                r[k] = int(rng.choice(alpha))
            elif t < 0.75 and len(r) > 1:
                del r[k]
            else:
                r.insert(k, int(rng.choice(alpha[:4])))
        reads.append(bytes(r))
    reads += [bytes(rng.choice(alpha[:4], int(rng.integers(20, 300)))) for _ in range(30)]
    got = za.LocalProfilesBatch.new_with_w256(reads, m, go, ge).sw_score_from_i8(ref)
    ends = za.StripedProfileBatch(reads, m, go, ge, "i32", 8).sw_score_ends(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        o_st, o_s, o_tier = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]) if o_st == S_ else 0, int(got.tier[i])) == (o_st, o_s if o_st == S_ else 0, o_tier), i
        if i % 7 == 0:
            st, (s, re_, qe) = oracle.score_ends("i32", 8, sc, rd, ref)
            assert (int(ends.status[i]), int(ends.score[i]), int(ends.ref_end[i]), int(ends.query_end[i])) == (st, s, re_, qe), i


def test_score_ranges_vs_oracle(za, oracle, dna):
    """sw_simd_score_ranges (striped.rs:355-388): the reference's vectors (sw/test.rs:198-262, profile_set.rs:293-310)
    and synthetic batches against the oracle's two-pass restatement."""
    import torch

    from zoe_amd import synth

    sc = osc(oracle, dna, -10, -1)

    # sw/test.rs:198-241 and :244-262 (u8x8) — ranges equal the scalar alignment's ranges
    for q, r in ((b"GGGGGGGCCCCCAAAA", b"TTTTTTCCTTTTTTTTCCCCCTTTTT"), (b"CCCCA", b"TAAAA")):
        a = oracle.scalar_align(sc, q, r)
        got = za.StripedProfileBatch([q], dna.to_biased_matrix(), -10, -1, "u8", 8).sw_score_ranges(za.SeqSrc.Reference(r))
        assert (int(got.status[0]), int(got.score[0])) == (S_, a.score)
        assert (int(got.ref_start[0]), int(got.ref_end[0])) == a.ref_range
        assert (int(got.query_start[0]), int(got.query_end[0])) == a.query_range
    # profile_set.rs:293-310
    m = za.WeightMatrix.new_dna_matrix(4, -2, b"N")
    got = za.StripedProfileBatch([b"CGTTCGCCATAAAGGGGG"], m, -3, -1, "i8", 32).sw_score_ranges(za.SeqSrc.Reference(b"ATGCATCGATCGATCGATCGATCGATCGATGC"))
    assert (int(got.score[0]), int(got.query_start[0]), int(got.query_end[0]), int(got.ref_start[0]), int(got.ref_end[0])) == (26, 0, 15, 14, 31)
    lp = za.LocalProfilesBatch.new_with_w256([b"CGTTCGCCATAAAGGGGG"], m, -3, -1).sw_score_ranges_from_i8(za.SeqSrc.Reference(b"ATGCATCGATCGATCGATCGATCGATCGATGC"))
    assert (int(lp.score[0]), int(lp.query_start[0]), int(lp.query_end[0]), int(lp.ref_start[0]), int(lp.ref_end[0]), int(lp.tier[0])) == (26, 0, 15, 14, 31, 8)
    # the cascade (profile_set.rs:313-362) on 150 bp reads: perfect reads overflow i8 and answer at i16
    ref2 = synth.reference_host(2000)
    host2 = synth.reads_host(ref2, 31, 300, 150)
    casc = za.LocalProfilesBatch.new_with_w256([bytes(r) for r in host2], dna, -10, -1).sw_score_ranges_from_i8(za.SeqSrc.Reference(ref2))
    tiers = set()
    for i in range(300):
        o_st, o_s, o_rr, o_qr, o_t = oracle.cascade_score_ranges(8, 256, sc, host2[i], ref2)
        assert int(casc.status[i]) == o_st, i
        if o_st == S_:
            assert (int(casc.score[i]), (int(casc.ref_start[i]), int(casc.ref_end[i])), (int(casc.query_start[i]), int(casc.query_end[i])), int(casc.tier[i])) == (o_s, o_rr, o_qr, o_t), i
            tiers.add(o_t)
    assert tiers == {8, 16}
    # synthetic fixed-length batch
    ref = synth.reference_host(1500)
    host = synth.reads_host(ref, 777, 500, 120)
    host[9] = np.frombuffer(b"N" * 120, dtype=np.uint8)
    rb = za.ReadBatch.from_fixed(torch.from_numpy(host.reshape(-1)).cuda(), 120)
    got = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16).sw_score_ranges(za.SeqSrc.Reference(ref))
    for i in range(500):
        st, s, rr, qr = oracle.score_ranges("i16", 16, sc, host[i], ref)
        assert int(got.status[i]) == st, i
        if st == S_:
            assert (int(got.score[i]), (int(got.ref_start[i]), int(got.ref_end[i])), (int(got.query_start[i]), int(got.query_end[i]))) == (s, rr, qr), i
    # ragged + low-complexity + i8 overflow statuses + SeqSrc::Query swap
    hb, hoff = synth.reads_ragged_host(ref, 5, 200, 20, 300)
    reads = [hb[hoff[i] : hoff[i + 1]].tobytes() for i in range(200)] + [b"ACACACACACAC", b"TTTTTTTTTT", b"GGGGGCCCCC"]
    got = za.StripedProfileBatch(reads, dna, -10, -1, "i8", 32).sw_score_ranges(za.SeqSrc.Query(ref))
    for i, rd in enumerate(reads):
        st, s, rr, qr = oracle.score_ranges("i8", 32, sc, rd, ref)
        assert int(got.status[i]) == st, i
        if st == S_:
            assert (int(got.score[i]), (int(got.query_start[i]), int(got.query_end[i])), (int(got.ref_start[i]), int(got.ref_end[i]))) == (s, rr, qr), i


def test_full_size_10m_reads_properties(za, oracle, dna, debug):
    """BASELINE.json configs[1] at full size (10 M x 150 bp vs 2 kb), checked through size-independent properties:
    the two independent kernels (v1 saturating-i16, v2 drifted/max3) agree on every read, a second run is identical,
    shard-wise generation equals whole-batch generation, bounds hold, and a random sample equals the oracle."""
    import torch

    from zoe_amd import synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    n = 10_000_000
    rb = synth.reads_device(ctx, ref, 0, n, 150)
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    debug.set(0)
    a = prof.sw_score_from_i8(ref)
    assert 0 < ctx.prune_rescored() < n // 20  # the default path: the seeded exact pass (r03)
    s2, st2, t2 = a.score.clone(), a.status.clone(), a.tier.clone()
    b = prof.sw_score_from_i8(ref)
    assert torch.equal(b.score, s2) and torch.equal(b.status, st2)  # deterministic
    debug.set(debug.SCORE_V1)
    c = prof.sw_score_from_i8(ref)
    assert ctx.prune_rescored() == 0  # v1: every cell of every read
    assert torch.equal(c.score, s2) and torch.equal(c.status, st2) and torch.equal(c.tier, t2)  # v1 == default on all 10 M reads
    debug.set(0)
    from zoe_amd import _lib

    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)  # and the v2 full pass
    try:
        f = prof.sw_score_from_i8(ref)
        assert ctx.prune_rescored() == 0
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    assert torch.equal(f.score, s2) and torch.equal(f.status, st2) and torch.equal(f.tier, t2)
    del f
    # bounds: 0 <= score <= 2 * L; status Some <=> score > 0; tier 8 <=> score < 255
    assert int(s2.max()) <= 300 and int(s2.min()) >= 0
    assert torch.equal(st2 == 0, s2 > 0) and torch.equal(t2 == 8, s2 < 255)
    # a shard regenerated on its own (counter-based generator) scores like the same reads inside the big batch
    first, cnt = 7_654_321, 50_000
    shard = synth.reads_device(ctx, ref, first, cnt, 150)
    d = za.LocalProfilesBatch.new_with_w256(shard, dna, -10, -1).sw_score_from_i8(ref)
    assert torch.equal(d.score, s2[first : first + cnt])
    # random sample against the oracle
    rng = np.random.default_rng(1)
    idx = np.sort(rng.choice(n, 3000, replace=False))
    host = rb.bases.view(n, 150)[torch.from_numpy(idx).cuda()].cpu().numpy()
    ws, wst, wt = oracle.batch_score_w256(8, osc(oracle, dna, -10, -1), host, ref, fixed_len=150, threads=16)
    assert np.array_equal(s2.cpu().numpy()[idx].view(np.uint32), ws) and np.array_equal(st2.cpu().numpy()[idx], wst)
    assert np.array_equal(t2.cpu().numpy()[idx], wt)


def test_host_batch_pipeline_matches_device_batch(za, oracle, dna):
    """Host-memory batches above one chunk are scored chunk by chunk with the next chunk's H2D copy in flight
    (zsw_capi.hip run_score): the results must equal the device-resident call and, on a sample, the oracle."""
    import ctypes as C

    import torch

    from zoe_amd import _lib, synth

    n, L = 5_300_000, 150  # three chunks (0.5 M, 4 M, 0.8 M: zsw_capi.hip PIPE_FIRST / PIPE_CHUNK)
    ref = synth.reference_host(2000)
    ctx = za.SwContext.get(0)
    rb = synth.reads_device(ctx, ref, 0, n, L)
    dev = za.into_local_profile(rb, dna, -10, -1).sw_score_from_i8(ref)
    dscore, dstatus, dtier = dev.score.cpu().numpy().view(np.uint32), dev.status.cpu().numpy(), dev.tier.cpu().numpy()
    host = rb.bases.cpu().numpy()
    lib = _lib.load()
    b = _lib.ZswBatch()
    b.bases, b.offsets, b.fixed_len, b.n_reads, b.mem = host.ctypes.data, None, L, n, _lib.MEM_HOST
    score = np.zeros(n, dtype=np.uint32)
    status = np.full(n, 9, dtype=np.uint8)
    tier = np.zeros(n, dtype=np.uint8)
    assert lib.zsw_score_batch_from(ctx.h, C.byref(b), 8, 256, score.ctypes.data, status.ctypes.data, tier.ctypes.data, None) == 0
    assert np.array_equal(score, dscore) and np.array_equal(status, dstatus) and np.array_equal(tier, dtier)
    sc = osc(oracle, dna, -10, -1)
    for i in list(range(0, 200)) + list(range(499_900, 500_100)) + list(range(4_499_900, 4_500_100)) + list(range(n - 100, n)):
        o_st, o_s, o_t = oracle.cascade_score(8, 256, sc, host[i * L:(i + 1) * L], ref)
        assert (int(status[i]), int(score[i]) if o_st == S_ else 0) == (o_st, o_s if o_st == S_ else 0), i
    # the same batch as ZSW_ENCODING_PACKED4 (two residue indices per byte, zsw_pack4_host): half the bytes cross PCIe, the device
    # spells them out again; lower case, U and N among the reads (index_map sends them to the indices the bytes would have)
    host2 = host.copy()
    host2[5:40:7] = np.frombuffer(b"acgtuNnRy"[:5], dtype=np.uint8)[:5]
    packed = np.zeros(n * ((L + 1) // 2), dtype=np.uint8)
    assert lib.zsw_pack4_host(ctx.h, host2.ctypes.data, n, L, packed.ctypes.data) == 0
    bp = _lib.ZswBatch()
    bp.bases, bp.offsets, bp.fixed_len, bp.n_reads, bp.mem, bp.encoding = packed.ctypes.data, None, L, n, _lib.MEM_HOST, 1
    b.bases = host2.ctypes.data
    assert lib.zsw_score_batch_from(ctx.h, C.byref(b), 8, 256, score.ctypes.data, status.ctypes.data, tier.ctypes.data, None) == 0
    score2, status2, tier2 = np.zeros(n, dtype=np.uint32), np.full(n, 9, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    assert lib.zsw_score_batch_from(ctx.h, C.byref(bp), 8, 256, score2.ctypes.data, status2.ctypes.data, tier2.ctypes.data, None) == 0
    assert np.array_equal(score2, score) and np.array_equal(status2, status) and np.array_equal(tier2, tier)
    # a small (single-chunk) packed batch with an odd read length, through the ends call; and the misuse cases
    Lo, no = 151, 3000
    small = synth.reads_host(ref, 3, no, Lo)
    pk = np.zeros(no * ((Lo + 1) // 2), dtype=np.uint8)
    assert lib.zsw_pack4_host(ctx.h, small.ctypes.data, no, Lo, pk.ctypes.data) == 0
    outs = [np.zeros(no, dtype=np.uint32) for _ in range(6)]
    sts = [np.zeros(no, dtype=np.uint8) for _ in range(2)]
    for enc, base, o, st in ((0, small, outs[:3], sts[0]), (1, pk, outs[3:], sts[1])):
        bb = _lib.ZswBatch()
        bb.bases, bb.offsets, bb.fixed_len, bb.n_reads, bb.mem, bb.encoding = base.ctypes.data, None, Lo, no, _lib.MEM_HOST, enc
        assert lib.zsw_score_ends_batch(ctx.h, C.byref(bb), 1, 16, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, st.ctypes.data, None) == 0
    assert all(np.array_equal(outs[k], outs[k + 3]) for k in range(3)) and np.array_equal(sts[0], sts[1])
    bad = _lib.ZswBatch()
    bad.bases, bad.offsets, bad.fixed_len, bad.n_reads, bad.mem, bad.encoding = rb.bases.data_ptr(), None, L, 10, _lib.MEM_DEVICE, 1
    assert lib.zsw_score_batch_from(ctx.h, C.byref(bad), 8, 256, dev.score.data_ptr(), dev.status.data_ptr(), dev.tier.data_ptr(), None) == -1  # device memory: not packed
    bad.mem, bad.encoding = _lib.MEM_HOST, 7
    bad.bases = host.ctypes.data
    assert lib.zsw_score_batch_from(ctx.h, C.byref(bad), 8, 256, score2.ctypes.data, status2.ctypes.data, tier2.ctypes.data, None) == -1
    del dev, rb
    torch.cuda.empty_cache()


def test_long_reads_are_scored_tile_by_tile(za, oracle, dna, debug):
    """Reads longer than the widest strip configuration (2,432 columns) run as several TILED launches of the packed kernel, the
    strip boundary of every reference row passing through HBM: score, ends and ranges must equal the oracle, for DNA and for
    a 25-letter alphabet, in a ragged and in a fixed-length batch, and must equal the exact-kernel path (ZSW_DEBUG_NO_TILES)."""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    R = 3500
    ref = bytes(rng.choice(alpha, R))
    sc = osc(oracle, dna, -10, -1)

    def mutated(L):
        base = bytearray((ref * 3)[int(rng.integers(0, R)):][:L])
        for _ in range(L // 40):
            k = int(rng.integers(0, len(base)))
            u = rng.random()
            if u < 0.5:
                base[k] = int(rng.choice(alpha))
            elif u < 0.75:
                del base[k]
            else:
                base.insert(k, int(rng.choice(alpha)))
        return bytes(base)

    reads = [mutated(L) for L in (2433, 2500, 4864, 4865, 5200, 7400)] + [bytes(rng.choice(alpha, 3000)), ref[500:650], ref[:2432]]
    p = za.StripedProfileBatch(reads, dna, -10, -1, "i32", 8)
    got, ends, rg = p.sw_score(ref), p.sw_score_ends(za.SeqSrc.Reference(ref)), p.sw_score_ranges(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        st, (s, re_, qe) = oracle.score_ends("i32", 8, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i])) == (st, s if st == S_ else 0), i
        assert (int(ends.score[i]), int(ends.ref_end[i]), int(ends.query_end[i])) == (s, re_, qe), i
        st, s, rr, qr = oracle.score_ranges("i32", 8, sc, rd, ref)
        assert (int(rg.score[i]), (int(rg.ref_start[i]), int(rg.ref_end[i])), (int(rg.query_start[i]), int(rg.query_end[i]))) == (s, rr, qr), i
    # 3-pass alignment on top of the tiled forward and reverse passes
    a3 = za.StripedProfileBatch(reads[:4], dna, -10, -1, "i32", 8).sw_align_3pass(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads[:4]):
        want = oracle.align_3pass("i32", 8, sc, rd, ref)[0]
        assert a3.key(i) == (want.key() if want.status == S_ else (want.status, 0, (0, 0), (0, 0), "", 0, 0)), i
    c16 = za.LocalProfilesBatch.new_with_w256(reads, dna, -10, -1).sw_score_from_i16(ref)
    for i, rd in enumerate(reads):
        st, s, tier = oracle.cascade_score(16, 256, sc, rd, ref)
        assert (int(c16.status[i]), int(c16.score[i]) if st == S_ else 0, int(c16.tier[i])) == (st, s if st == S_ else 0, tier), i
    # fixed-length batch of long reads
    fixed = [mutated(5000)[:4900].ljust(4900, b"A") for _ in range(5)]
    gf = za.StripedProfileBatch(fixed, dna, -10, -1, "i32", 8).sw_score_ends(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(fixed):
        st, (s, re_, qe) = oracle.score_ends("i32", 8, sc, rd, ref)
        assert (int(gf.score[i]), int(gf.ref_end[i]), int(gf.query_end[i])) == (s, re_, qe), i
    # 25-letter alphabet
    rngp, keys, mp, w, m = _protein_case(za, 5)
    pa = np.frombuffer(keys[:20], dtype=np.uint8)
    pref = bytes(rngp.choice(pa, 1500))
    preads = [bytes(rngp.choice(pa, 2600)), (pref * 3)[100:3100], bytes(rngp.choice(pa, 5000))]
    psc = oracle.Scoring(w, mp.index_map, -11, -1)
    gp = za.StripedProfileBatch(preads, m, -11, -1, "i32", 8).sw_score_ends(za.SeqSrc.Reference(pref))
    for i, rd in enumerate(preads):
        st, (s, re_, qe) = oracle.score_ends("i32", 8, psc, rd, pref)
        assert (int(gp.score[i]), int(gp.ref_end[i]), int(gp.query_end[i])) == (s, re_, qe), i
    # the exact-kernel path gives the same answers
    debug.set(debug.NO_TILES)
    ex = p.sw_score_ends(za.SeqSrc.Reference(ref))
    for name in ("score", "ref_end", "query_end", "status"):
        assert np.array_equal(getattr(ex, name).cpu().numpy(), getattr(ends, name).cpu().numpy()), name


def test_scores_beyond_the_packed_range_use_the_32bit_tile_kernel(za, oracle, dna, debug):
    """A genome-sized read that matches the reference scores far above what the packed 16-bit kernels can hold (~28,000): the
    packed pass puts it on the worklist and the 32-bit tile kernel (zsw_score_w32.hip) scores it — score, ends and the cascade tier identical to the oracle; ZSW_DEBUG_NO_W32 (exact
    kernel) agrees."""
    rng = np.random.default_rng(123)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    R = 21000
    ref = bytes(rng.choice(alpha, R))
    big = bytearray(ref[300:18300])  # 18 kb, a handful of edits: score ~35,000
    for k in (2000, 7000, 7001, 12000):
        big[k] = ord("A") if big[k] != ord("A") else ord("C")
    del big[9000:9003]
    reads = [bytes(big), ref[1000:16500], ref[50:200], bytes(rng.choice(alpha, 3000))]
    sc = osc(oracle, dna, -10, -1)
    p = za.StripedProfileBatch(reads, dna, -10, -1, "i32", 8)
    got = p.sw_score_ends(za.SeqSrc.Reference(ref))
    plain = p.sw_score(ref)
    for i, rd in enumerate(reads):
        st, (s, re_, qe) = oracle.score_ends("i32", 8, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]), int(got.ref_end[i]), int(got.query_end[i])) == (st, s, re_, qe), i
        assert (int(plain.status[i]), int(plain.score[i])) == (st, s), i
    assert int(got.score[0]) > 32767 and int(got.score[1]) > 30000
    c = za.LocalProfilesBatch.new_with_w256(reads, dna, -10, -1).sw_score_from_i8(ref)
    for i, rd in enumerate(reads):  # an i16 profile holds true scores up to 65,534 (offset i16::MIN): these answer at the i16 tier
        st, s, tier = oracle.cascade_score(8, 256, sc, rd, ref)
        assert (int(c.status[i]), int(c.score[i]), int(c.tier[i])) == (st, s, tier), i
    assert [int(x) for x in c.tier][:2] == [16, 16]
    # a heavier matrix reaches the same range with 3 kb reads: oracle, 32-bit tiles and the exact kernel must agree
    m20 = za.WeightMatrix.new_dna_matrix(20, -30, b"N")
    ref2 = ref[:4000]
    reads2 = [ref2[100:3100], bytes(rng.choice(alpha, 2600)), ref2[2000:2100]]
    sc2 = osc(oracle, m20, -40, -3)
    p2 = za.StripedProfileBatch(reads2, m20, -40, -3, "i32", 8)
    w32 = p2.sw_score_ends(za.SeqSrc.Reference(ref2))
    for i, rd in enumerate(reads2):
        st, (s, re_, qe) = oracle.score_ends("i32", 8, sc2, rd, ref2)
        assert (int(w32.status[i]), int(w32.score[i]), int(w32.ref_end[i]), int(w32.query_end[i])) == (st, s, re_, qe), i
    assert int(w32.score[0]) == 60000
    debug.set(debug.NO_W32)
    ex = p2.sw_score_ends(za.SeqSrc.Reference(ref2))
    for name in ("score", "ref_end", "query_end", "status"):
        assert np.array_equal(getattr(ex, name).cpu().numpy(), getattr(w32, name).cpu().numpy()), name


def test_reconfiguring_the_reference_behind_asynchronous_calls(za, oracle, dna):
    """Score calls on device memory are asynchronous on the caller's stream. Two calls with DIFFERENT references back to back on
    a non-blocking side stream: zsw_set_reference must not overwrite the first reference while the first kernel still reads it."""
    import torch

    from zoe_amd import synth

    ctx = za.SwContext.get(0)
    ref_a, ref_b = synth.reference_host(2000), synth.reference_host(2000)[::-1]
    n = 2_000_000  # ~60 ms of kernel per call: the second configuration call arrives while the first kernel runs
    rb = synth.reads_device(ctx, ref_a, 0, n, 150)
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        a = prof.sw_score_from_i8(ref_a)
        b = prof.sw_score_from_i8(ref_b)
        a2 = prof.sw_score_from_i8(ref_a)
    side.synchronize()
    assert torch.equal(a.score, a2.score) and torch.equal(a.status, a2.status)
    sc = osc(oracle, dna, -10, -1)
    idx = np.arange(0, n, n // 512)
    host = rb.bases.view(n, 150)[torch.from_numpy(idx).cuda()].cpu().numpy()
    for ref, got in ((ref_a, a), (ref_b, b)):
        ws, wst, _ = oracle.batch_score_w256(8, sc, host, ref, fixed_len=150, threads=8)
        assert np.array_equal(got.score[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32), ws)
        assert np.array_equal(got.status[torch.from_numpy(idx).cuda()].cpu().numpy(), wst)


def test_a_megabase_read_against_a_short_reference(za, oracle, dna):
    """No length limit below 2^31 on either sequence: a 1.1 Mb read (453 tiles of 2,432 columns; the exact kernel's scratch is
    sized from a byte budget, not from a fixed slot count) against a 150 bp reference, score and ends equal to the oracle."""
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 150))
    big = bytearray(rng.choice(alpha, 1_100_000))
    big[700_000:700_150] = ref  # the reference itself, with two edits
    big[700_040] = ord("A") if big[700_040] != ord("A") else ord("C")
    del big[700_100]
    reads = [bytes(big), bytes(rng.choice(alpha, 300)), ref[20:120]]
    sc = osc(oracle, dna, -10, -1)
    got = za.StripedProfileBatch(reads, dna, -10, -1, "i32", 8).sw_score_ends(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        st, (s, re_, qe) = oracle.score_ends("i32", 8, sc, rd, ref)
        assert (int(got.status[i]), int(got.score[i]), int(got.ref_end[i]), int(got.query_end[i])) == (st, s, re_, qe), i
    assert int(got.score[0]) > 250
