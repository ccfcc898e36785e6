"""GPU parity tests for the alignment path (sw_simd_align + traceback + CIGAR), through the C ABI.

Checker = oracle/ at the SAME <T, N> instantiation (the striped traceback is layout dependent).
Bar: bit-exact status, score, ref_range, query_range, CIGAR, ref_len, query_len.
"""
import numpy as np
import pytest

from conftest import stable_seed

pytestmark = pytest.mark.gpu

S_, O_, U_ = 0, 1, 2


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


def test_doc_examples(za):
    # striped.rs:418-441
    m = za.WeightMatrix.new_biased_dna_matrix(4, -2, b"N")
    a = za.StripedProfileBatch([b"CGTTCGCCATAAAGGGGG"], m, -3, -1, "u8", 8).sw_align(za.SeqSrc.Reference(b"ATGCATCGATCGATCGATCGATCGATCGATGC"))
    assert a.key(0)[:2] == (S_, 26) and a.cigar(0) == "6M2D9M3S"
    # sw/mod.rs:164-188 and :224-248
    m = za.WeightMatrix.new_dna_matrix(4, -2, b"N")
    a = za.StripedProfileBatch([b"CTCAGATTG"], m, -3, -1, "i8", 32).sw_align(za.SeqSrc.Reference(b"GGCCACAGGATTGAG"))
    assert a.key(0)[:2] == (S_, 27) and a.cigar(0) == "5M1D4M" and a.key(0)[2][0] == 3
    a = za.into_local_profile([b"CTCAGATTG"], m, -3, -1).sw_align_from_i8(za.SeqSrc.Reference(b"GGCCACAGGATTGAG"))
    assert a.key(0)[:2] == (S_, 27) and a.cigar(0) == "5M1D4M" and int(a.tier[0]) == 8
    # sw/mod.rs:193-218 custom alphabet
    mp = za.ByteIndexMap.new(b"ABCD", b"A")
    m4 = za.WeightMatrix.new(mp, 1, -1, None)
    a = za.StripedProfileBatch([b"AABDDAB"], m4, -4, -2, "i8", 32).sw_align(za.SeqSrc.Reference(b"BDAACAABDDDB"))
    assert a.key(0)[:2] == (S_, 5) and a.cigar(0) == "5M2S"


MACRO = [
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


@pytest.mark.parametrize("case", range(len(MACRO)))
def test_sw_simd_align_macro_cases(za, oracle, case):
    """test_sw_simd_align! (src/alignment/sw/test.rs:7-51, 116-195): striped == scalar, signed and unsigned."""
    p, r, it, ut, lanes = MACRO[case]
    m = za.WeightMatrix.new(za.DNA_PROFILE_MAP, 2, -5, b"N")
    want = oracle.scalar_align(osc(oracle, m, -10, -1), p, r)
    a = za.StripedProfileBatch([p], m, -10, -1, it, lanes).sw_align(za.SeqSrc.Reference(r))
    assert a.key(0) == okey(want)
    a = za.StripedProfileBatch([p], m.to_biased_matrix(), -10, -1, ut, lanes).sw_align(za.SeqSrc.Reference(r))
    assert a.key(0) == okey(want)


def test_h5_h1_i16x8(za, oracle, h5, h1):
    # src/alignment/sw/test.rs:116-119 (profile = H5, 1,760 bp: nv = 220 at N = 8)
    m = za.WeightMatrix.new(za.DNA_PROFILE_MAP, 2, -5, b"N")
    want = oracle.scalar_align(osc(oracle, m, -10, -1), h5, h1)
    a = za.StripedProfileBatch([h5], m, -10, -1, "i16", 16).sw_align(za.SeqSrc.Reference(h1))
    assert a.key(0) == okey(oracle.align("i16", 16, osc(oracle, m, -10, -1), h5, h1))
    assert a.key(0) == okey(want)


def test_layout_dependent_cigars(za, oracle):
    """SURVEY.md §7 hard part 1: equal score and ends, different CIGAR per lane count — the GPU must follow N."""
    cases = [
        (4, -2, -3, -1, b"GGACTAAGCTAACACAGGTAGGCTTTATAAAAGGTTAAAGTGCGTGAGCTAGGGTGGCTCTCACT", b"TATAAAAGGTTAAAGTGCTGTAGCTTAGGGTTGCTCTC"),
        (3, -1, -4, -1, b"TGGGGCATTTATGCGATGCAAGACAGGTCTAATATTGAAATTTATTCTAGACTATGCGAGGCCGCTCAAAGGAACCATTACCTTTTTCCGTAGGTCTCCCGATCGCGGCTAACTACTGC", b"ATAGCGATCGCAGCGCCAGGTCT"),
    ]
    for ma, mi, go, ge, ref, read in cases:
        m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
        seen = set()
        for N in (2, 4, 8, 16, 32, 64):
            want = oracle.align("i16", N, osc(oracle, m, go, ge), read, ref)
            got = za.StripedProfileBatch([read], m, go, ge, "i16", N).sw_align(za.SeqSrc.Reference(ref))
            assert got.key(0) == okey(want), N
            seen.add(got.cigar(0))
        assert len(seen) == 2


@pytest.mark.parametrize("scheme", [(4, -2, -3, -1), (2, -5, -10, -1), (3, -1, 0, 0), (1, -1, -1, -1), (5, -4, -2, 0)])
def test_random_pairs_all_lane_counts(za, oracle, scheme):
    ma, mi, go, ge = scheme
    rng = np.random.default_rng(stable_seed(scheme))
    m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
    sc = osc(oracle, m, go, ge)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 160))
    reads = []
    for _ in range(120):
        L = int(rng.integers(8, 60))
        if rng.random() < 0.7:  # sampled from the reference with edits (ties between I and D paths)
            s = int(rng.integers(0, 160 - L))
            r = bytearray(ref[s : s + L])
            for _ in range(int(rng.integers(0, 4))):
                k = int(rng.integers(0, len(r)))
                t = rng.random()
                if t < 0.4:
                    r[k] = int(rng.choice(alpha))
                elif t < 0.7:
                    del r[k]
                else:
                    r.insert(k, int(rng.choice(alpha)))
            reads.append(bytes(r) if r else b"A")
        else:  # low complexity
            reads.append(bytes(rng.choice(alpha[:2], L)))
    for N in (2, 4, 8, 16, 32, 64):
        got = za.StripedProfileBatch(reads, m, go, ge, "i16", N).sw_align(za.SeqSrc.Reference(ref))
        for i, rd in enumerate(reads):
            want = oracle.align("i16", N, sc, rd, ref)
            assert got.key(i) == okey(want), (N, i, rd)


def test_config3_sample_cascade_w256(za, oracle):
    """BASELINE.json configs[2] shape: 150 bp reads vs a 2 kb reference, sw_align_from_i8 at the w256 preset
    (CIGAR from i16x16 when the score >= 255, from i8x32 otherwise)."""
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    n = 3000
    host = synth.reads_host(ref, 0, n, 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    import torch

    rb = za.ReadBatch.from_fixed(torch.from_numpy(host.reshape(-1)).cuda(), 150)
    got = za.into_local_profile(rb, dna, -10, -1).sw_align_from_i8(za.SeqSrc.Reference(ref))
    tiers = set()
    for i in range(n):
        want, tier = oracle.cascade_align(8, 256, sc, host[i], ref)
        assert got.key(i) == okey(want), i
        if want.status == S_:
            assert int(got.tier[i]) == tier
            tiers.add(tier)
            # sw_score_from_path-style self check (sw/mod.rs:399-454)
            rr = got.records[i]
            assert oracle.score_from_path(sc, host[i], ref[int(rr["ref_start"]) : int(rr["ref_end"])], got.cigar(i)) == int(rr["score"])
    assert tiers == {8, 16}


def test_invert_seqsrc_query(za, oracle):
    # src/alignment/types/test.rs:14-38 through the striped path (+ random pairs)
    m = za.WeightMatrix.new_dna_matrix(4, -2, b"N")
    sc = osc(oracle, m, -3, -1)
    ref, q = b"GGCCACAGGATTGAGC", b"TCTCAGATTGCAGTTT"
    a = za.StripedProfileBatch([q], m, -3, -1, "i16", 16).sw_align(za.SeqSrc.Query(ref))
    assert a.key(0) == okey(oracle.align("i16", 16, sc, q, ref, other_is_query=True))
    rng = np.random.default_rng(8)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    other = bytes(rng.choice(alpha, 90))
    reads = [bytes(rng.choice(alpha[:4], int(rng.integers(5, 50)))) for _ in range(80)] + [other[10:60], other[40:85]]
    a = za.StripedProfileBatch(reads, m, -3, -1, "i8", 16).sw_align(za.SeqSrc.Query(other))
    for i, rd in enumerate(reads):
        assert a.key(i) == okey(oracle.align("i8", 16, sc, rd, other, other_is_query=True)), i


def test_long_gap_leaves_window_and_many_ciglets(za, oracle):
    """A free-extension deletion of 700 reference bases makes the traceback span far more rows than the retained
    window (fallback: full window); an alternating M/I/D pattern needs more than the 32 fast-path ciglet slots."""
    rng = np.random.default_rng(21)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 1200))
    read = ref[100:160] + ref[860:920]
    m = za.WeightMatrix.new_dna_matrix(3, -4, b"N")
    sc = osc(oracle, m, -5, 0)
    reads = [read, ref[300:380], ref[500:540] + ref[900:940]]
    got = za.StripedProfileBatch(reads, m, -5, 0, "i16", 16).sw_align(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        want = oracle.align("i16", 16, sc, rd, ref)
        assert got.key(i) == okey(want), i
    assert "700D" in got.cigar(0)
    # many ciglets: read = reference with a base deleted every 6 bases (gaps are cheap)
    src = ref[0:400]
    rd = bytes(b for k, b in enumerate(src) if k % 6 != 5)
    m2 = za.WeightMatrix.new_dna_matrix(5, -9, b"N")
    got = za.StripedProfileBatch([rd], m2, -1, -1, "i16", 16).sw_align(za.SeqSrc.Reference(ref))
    want = oracle.align("i16", 16, osc(oracle, m2, -1, -1), rd, ref)
    assert got.key(0) == okey(want)
    assert want.n_ciglets > 32


def test_overflow_unmapped_and_ragged(za, oracle):
    from zoe_amd import synth

    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    ref = synth.reference_host(800)
    hb, hoff = synth.reads_ragged_host(ref, 11, 301, 20, 260)
    reads = [hb[hoff[i] : hoff[i + 1]].tobytes() for i in range(301)]
    reads[7] = b"N" * 33
    # direct i8: reads scoring >= 255 are Overflowed (striped.rs:557-559), the others align at i8x32
    got = za.StripedProfileBatch(reads, dna, -10, -1, "i8", 32).sw_align(za.SeqSrc.Reference(ref))
    n_over = 0
    for i, rd in enumerate(reads):
        want = oracle.align("i8", 32, sc, rd, ref)
        assert got.key(i) == okey(want), i
        n_over += want.status == O_
    assert n_over > 10 and int(got.status[7]) == U_
    got = za.LocalProfilesBatch.new_with_w512(reads, dna, -10, -1).sw_align_from_i8(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        want, tier = oracle.cascade_align(8, 512, sc, rd, ref)
        assert got.key(i) == okey(want), i


def test_align_degenerate_batches(za, oracle):
    """Nothing to align: empty reference (striped.rs:455-457), all-N reads, a single 1-base read, zero reads."""
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    a = za.StripedProfileBatch([b"ACGT", b"GG"], dna, -10, -1, "i16", 16).sw_align(za.SeqSrc.Reference(b""))
    assert list(a.status) == [U_, U_] and len(a.inc) == 0
    a = za.StripedProfileBatch([b"NNNN", b"nnnnnn"], dna, -10, -1, "i8", 32).sw_align(za.SeqSrc.Reference(b"ACGTACGT"))
    assert list(a.status) == [U_, U_]
    a = za.into_local_profile([b"A"], dna, -10, -1).sw_align_from_i8(za.SeqSrc.Reference(b"CCCACCC"))
    want, tier = oracle.cascade_align(8, 256, osc(oracle, dna, -10, -1), b"A", b"CCCACCC")
    assert a.key(0) == okey(want) and a.cigar(0) == "1M"
    import torch

    rb = za.ReadBatch(torch.zeros(1, dtype=torch.uint8, device="cuda"), 0, fixed_len=10, min_len=10)
    a = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16).sw_align(za.SeqSrc.Reference(b"ACGT"))
    assert len(a.status) == 0 and len(a.inc) == 0
    s = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16).sw_score(b"ACGT")
    assert s.score.numel() == 0


def test_config3_large_batch_properties(za, oracle, debug):
    """BASELINE.json configs[2] at its full size (10 M reads x 150 bp vs 2 kb, full traceback), through size-independent properties: every CIGAR consumes exactly the
    read and its reference range, scores and ends agree with the (independent) score+ends kernel for every read, and a
    random sample is bit-identical to the oracle and re-scores to its own score; the first 2 M reads are also aligned by the
    32-bit step-by-step kernel, which must give the same records and the same ciglet arrays for every one of them."""
    import torch

    from zoe_amd import synth

    ctx = za.SwContext.get(0)
    ref = synth.reference_host(2000)
    n = 10_000_000
    rb = synth.reads_device(ctx, ref, 1_000_000, n, 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    a = za.into_local_profile(rb, dna, -10, -1).sw_align_from_i8(za.SeqSrc.Reference(ref))
    assert 0 < ctx.prune_rescored() < n // 20  # pass 1 took the default path: the seeded exact pass with the end row (r03)
    rec, st = a.records, a.status
    some = st == S_
    from zoe_amd import _lib

    ctx.set_option(_lib.OPTION_EXACT_PRUNING, 0)  # score + ends from the full pass, every cell of every read
    try:
        ends = za.StripedProfileBatch(rb, dna, -10, -1, "i16", 16).sw_score_ends(za.SeqSrc.Reference(ref))
        assert ctx.prune_rescored() == 0
    finally:
        ctx.set_option(_lib.OPTION_EXACT_PRUNING, 1)
    assert np.array_equal(st, ends.status.cpu().numpy())
    assert np.array_equal(rec["score"][some], ends.score.cpu().numpy()[some].view(np.uint32))
    assert np.array_equal(rec["ref_end"][some], ends.ref_end.cpu().numpy()[some].view(np.uint32))
    assert np.array_equal(rec["query_end"][some], ends.query_end.cpu().numpy()[some].view(np.uint32))
    # ciglets are packed back to back in read order
    nc = rec["n_ciglets"].astype(np.int64)
    off = rec["ciglet_offset"].astype(np.int64)
    assert np.array_equal(off[some], (np.cumsum(nc) - nc)[some]) and int(nc.sum()) == len(a.inc)
    # per-read consumption: M+I+S = query_len, M+D = ref span, M+I = query span; no zero increments; no equal neighbours
    inc = a.inc.astype(np.int64)
    op = a.op
    owner = np.repeat(np.arange(n), nc)
    q_cons = np.bincount(owner, weights=inc * np.isin(op, (ord("M"), ord("I"), ord("S"))), minlength=n).astype(np.int64)
    r_cons = np.bincount(owner, weights=inc * np.isin(op, (ord("M"), ord("D"))), minlength=n).astype(np.int64)
    qa_cons = np.bincount(owner, weights=inc * np.isin(op, (ord("M"), ord("I"))), minlength=n).astype(np.int64)
    assert (q_cons[some] == 150).all()
    assert np.array_equal(r_cons[some], (rec["ref_end"].astype(np.int64) - rec["ref_start"])[some])
    assert np.array_equal(qa_cons[some], (rec["query_end"].astype(np.int64) - rec["query_start"])[some])
    assert (inc > 0).all() and set(np.unique(op)) <= set(b"MIDS")
    same_owner = owner[1:] == owner[:-1]
    assert not (same_owner & (op[1:] == op[:-1])).any()
    # the two independent pass-2 kernels (packed, closed-form lazy-F / 32-bit, Zoe's loop step by step) on the first 2 M reads
    m2 = 2_000_000
    sub = za.ReadBatch.from_fixed(rb.bases[: m2 * 150], 150)
    p2 = za.into_local_profile(sub, dna, -10, -1)
    debug.set(0)
    x = p2.sw_align_from_i8(za.SeqSrc.Reference(ref))
    debug.set(debug.ALIGN_NO_PACKED)
    y = p2.sw_align_from_i8(za.SeqSrc.Reference(ref))
    debug.set(0)
    assert np.array_equal(x.status, y.status) and np.array_equal(x.records, y.records)
    assert np.array_equal(x.inc, y.inc) and np.array_equal(x.op, y.op)
    assert np.array_equal(x.records, rec[:m2]) and np.array_equal(x.inc, a.inc[: len(x.inc)])
    del x, y, p2, sub
    # sample vs oracle + sw_score_from_path
    sc = osc(oracle, dna, -10, -1)
    rng = np.random.default_rng(4)
    idx = np.sort(rng.choice(n, 3000, replace=False))
    host = rb.bases.view(n, 150)[torch.from_numpy(idx).cuda()].cpu().numpy()
    for k, i in enumerate(idx):
        want, tier = oracle.cascade_align(8, 256, sc, host[k], ref)
        assert a.key(int(i)) == okey(want), i
        if want.status == S_:
            assert oracle.score_from_path(sc, host[k], ref[want.ref_range[0] : want.ref_range[1]], a.cigar(int(i))) == want.score


@pytest.mark.parametrize("scheme", [(1, -1, -1, -1), (2, -1, -2, -1), (2, -5, -10, -1), (5, -4, -6, -2), (1, -3, -4, -3)])
def test_late_start_on_long_references(za, oracle, scheme):
    """Pass 2 starts L + L*maxw/gap_extend rows before the first retained flag row instead of at row 0 (zsw_align.hip,
    warmup_rows). Long, repetitive references and cheap gaps are where an alignment path can span the most rows: every
    result must still equal the oracle, which always starts at row 0."""
    ma, mi, go, ge = scheme
    rng = np.random.default_rng(stable_seed(scheme))
    m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
    sc = osc(oracle, m, go, ge)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    R = 6000
    unit = bytes(rng.choice(alpha, 37))
    ref = bytearray(rng.choice(alpha[:2], R))           # two-letter background: many near-ties
    for s in range(400, R - 200, 900):                  # repeated unit, so partial alignments exist far upstream
        ref[s : s + 37] = unit
    ref = bytes(ref)
    reads = []
    for _ in range(60):
        L = int(rng.integers(20, 110))
        s = int(rng.integers(0, R - 3 * L))
        t = rng.random()
        if t < 0.4:
            r = bytearray(ref[s : s + L])
        elif t < 0.7:                                   # long deletion relative to the reference
            g = int(rng.integers(5, 2 * L))
            r = bytearray(ref[s : s + L // 2] + ref[s + L // 2 + g : s + L + g])
        else:
            r = bytearray(unit + ref[s : s + L])
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, len(r)))] = int(rng.choice(alpha))
        reads.append(bytes(r))
    for T, N in (("i16", 16), ("i16", 8), ("i8", 32)):
        got = za.StripedProfileBatch(reads, m, go, ge, T, N).sw_align(za.SeqSrc.Reference(ref))
        for i, rd in enumerate(reads):
            want = oracle.align(T, N, sc, rd, ref)
            assert got.key(i) == okey(want), (T, N, i, rd)


def test_profiles_too_long_for_lds_keep_their_rows_in_hbm(za, oracle):
    """nv above ~250 vectors no longer fits one wavefront's LDS: the generic kernel then keeps H, E, the residue codes and the
    flags of a block in HBM behind the flag ring. A 6 kb and a 4.2 kb read at i16x16 (nv = 375 / 263) against a 7 kb reference."""
    rng = np.random.default_rng(31)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 7000))
    m = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, m, -10, -1)

    def mutated(s0, L):
        r = bytearray(ref[s0:s0 + L])
        for _ in range(L // 60):
            k = int(rng.integers(0, len(r)))
            u = rng.random()
            if u < 0.5:
                r[k] = int(rng.choice(alpha))
            elif u < 0.75:
                del r[k]
            else:
                r.insert(k, int(rng.choice(alpha)))
        return bytes(r)

    reads = [mutated(500, 6000), mutated(2000, 4200), ref[100:300]]
    got = za.StripedProfileBatch(reads, m, -10, -1, "i16", 16).sw_align(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        want = oracle.align("i16", 16, sc, rd, ref)
        assert got.key(i) == okey(want), i
    assert int(got.records[0]["score"]) > 10000


@pytest.mark.parametrize("N", [8, 16, 32, 64])
def test_packed_and_32bit_alignment_kernels_agree(za, oracle, debug, N):
    """align_kernel_pk (two reads per lane group in 16-bit halves, lazy-F in closed form; the default for up to 16 vectors)
    and align_kernel_x / align_kernel (one read per lane group, Zoe's loop step by step) give identical records and CIGARs
    for every read of a ragged batch that covers every vector count 1..32 and beyond (33.. go to the 32-bit kernels either
    way), and both equal the oracle at the same <i16, N>."""
    rng = np.random.default_rng(stable_seed("pk", N))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 900))
    m = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    go, ge = -10, -1
    sc = osc(oracle, m, go, ge)
    reads = []
    for k in range(700):
        L = int(rng.integers(1, min(34 * N, 700)))
        s = int(rng.integers(0, 900 - L)) if L < 900 else 0
        r = bytearray(ref[s : s + L])
        for _ in range(int(rng.integers(0, 2 + L // 25))):
            j = int(rng.integers(0, len(r)))
            t = rng.random()
            if t < 0.5:
                r[j] = int(rng.choice(alpha))
            elif t < 0.75 and len(r) > 1:
                del r[j]
            else:
                r.insert(j, int(rng.choice(alpha)))
        if k % 11 == 0:
            r = bytearray(rng.choice(alpha[:2], max(1, L)))  # low complexity
        reads.append(bytes(r))
    prof = za.StripedProfileBatch(reads, m, go, ge, "i16", N)
    debug.set(0)
    a = prof.sw_align(za.SeqSrc.Reference(ref))
    debug.set(debug.ALIGN_NO_PACKED)
    b = prof.sw_align(za.SeqSrc.Reference(ref))
    debug.set(0)
    for i in range(len(reads)):
        assert a.key(i) == b.key(i), (N, i, len(reads[i]))
    for i in range(0, len(reads), 3):
        assert a.key(i) == okey(oracle.align("i16", N, sc, reads[i], ref)), (N, i)


def test_scores_at_the_edge_of_the_packed_kernels_16_bit_lanes(za, oracle):
    """align_kernel_pk keeps true scores in unsigned 16-bit halves and takes reads that scored at most 30,000; larger
    scores form their own group for the 32-bit kernel. Extreme weights (match 127, gap penalties up to 127) put reads on
    both sides of the limit into one batch; every CIGAR equals the oracle's at <i16, 16> (and <i32, 8> for the largest)."""
    rng = np.random.default_rng(stable_seed("edge"))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 600))
    for (ma, mi, go, ge) in ((127, -100, -127, -127), (127, -60, -90, -3), (120, -128, -10, -1)):
        m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
        sc = osc(oracle, m, go, ge)
        reads = []
        for L in (225, 230, 234, 236, 237, 240, 250, 256):  # 236 x 127 = 29,972 <= 30,000 < 237 x 127
            s0 = int(rng.integers(0, 600 - L))
            r = bytearray(ref[s0 : s0 + L])
            reads.append(bytes(r))
            r2 = bytearray(r)
            r2[L // 2] = ord("A") if r2[L // 2] != ord("A") else ord("C")
            del r2[L // 3]
            reads.append(bytes(r2))
        got = za.StripedProfileBatch(reads, m, go, ge, "i16", 16).sw_align(za.SeqSrc.Reference(ref))
        for i, rd in enumerate(reads):
            assert got.key(i) == okey(oracle.align("i16", 16, sc, rd, ref)), (ma, mi, go, ge, i, len(rd))
        assert max(int(got.records[i]["score"]) for i in range(len(reads))) > 30000
        got32 = za.StripedProfileBatch(reads, m, go, ge, "i32", 8).sw_align(za.SeqSrc.Reference(ref))
        for i, rd in enumerate(reads):
            assert got32.key(i) == okey(oracle.align("i32", 8, sc, rd, ref)), (ma, mi, go, ge, i)


def test_reads_with_one_optimal_alignment_skip_the_second_pass(za, oracle):
    """sw_simd_align's CIGAR depends on the striping only where several optimal alignments exist. A read whose maximum sits in one
    cell, whose reversed maximum sits in one cell, whose ranges have equal lengths with the diagonal adding up to the score, and whose
    score lies beyond what a path with an insertion and a deletion between the same corners can reach has exactly one optimal
    alignment — the gapless diagonal — and gets it without the literal striped recurrence (tests/models/align_gapless_cert.cpp).
    500,000 reads (synthetic; tie-rich: tandem repeats, a duplicated stretch, low complexity; diverged; with indels): every record
    and every ciglet must equal the all-literal path (ZSW_DEBUG_ALIGN_NO_CERTIFICATE), both SeqSrc directions, and the oracle's
    cascade on a sample; most synthetic reads must have been certified (the call must be much faster than the literal one)."""
    import time

    import torch

    from test_gpu_bounds import diverged_reads
    from zoe_amd import _lib, synth

    ctx = za.SwContext.get(0)
    rng = np.random.default_rng(stable_seed("gapless-cert"))
    ref = bytearray(synth.reference_host(2500))
    ref[900:1000] = ref[300:400]
    for i in range(1500 + 2, 1580):
        ref[i] = ref[i - 2]
    ref = bytes(ref)
    r = np.frombuffer(ref, dtype=np.uint8)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    L = 150
    syn = synth.reads_host(ref, 29, 400_000, L)
    ties = np.empty((50_000, L), dtype=np.uint8)
    for i in range(len(ties)):
        kind = i % 5
        p = int(rng.integers(0, len(r) - L))
        if kind == 0:
            p = int(rng.integers(250, 320))
        elif kind == 1:
            p = int(rng.integers(1440, 1520))
        q = r[p:p + L].copy()
        if kind == 2:
            p2 = int(rng.integers(0, len(r) - L))
            q[L // 2:] = r[p2:p2 + L - L // 2]
        elif kind == 3:
            q = rng.choice(alpha[:2], L).astype(np.uint8)
        elif kind == 4:  # three or four substitutions: around the threshold of the certificate
            for k in rng.choice(L, int(rng.integers(3, 5)), replace=False):
                q[k] = alpha[(int(np.where(alpha == q[k])[0][0]) + 1) % 4] if q[k] in alpha else q[k]
        ties[i] = q
    reads = np.concatenate([syn, ties, diverged_reads(ref, 25_000, L, 30, 7), diverged_reads(ref, 25_000, L, 80, 8)])
    n = len(reads)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    rb = za.ReadBatch.from_fixed(torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda(), L)
    prof = za.into_local_profile(rb, dna, -10, -1)

    def timed(src):
        prof.sw_align_from_i8(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = prof.sw_align_from_i8(src)
        torch.cuda.synchronize()
        return a, time.perf_counter() - t0

    got, t_cert = timed(za.SeqSrc.Reference(ref))
    gotq = prof.sw_align_from_i8(za.SeqSrc.Query(ref))
    ctx.debug_set(_lib.DEBUG_ALIGN_NO_CERTIFICATE)
    try:
        want, t_lit = timed(za.SeqSrc.Reference(ref))
        wantq = prof.sw_align_from_i8(za.SeqSrc.Query(ref))
    finally:
        ctx.debug_set(0)
    for a, b in ((got, want), (gotq, wantq)):
        assert np.array_equal(a.status, b.status) and np.array_equal(a.tier, b.tier)
        assert np.array_equal(a.records, b.records)
        assert np.array_equal(a.inc, b.inc) and np.array_equal(a.op, b.op)
    assert t_cert < 0.9 * t_lit, (t_cert, t_lit)  # (a tie-rich set: the synthetic reads alone run at about 0.6)
    sc = osc(oracle, dna, -10, -1)
    for i in list(range(0, 400_000, 2003)) + list(range(400_000, n, 499)):
        w, tier = oracle.cascade_align(8, 256, sc, reads[i], ref)
        assert got.key(i) == okey(w), i
