"""GPU parity of the 3-pass alignment path (sw_align_3pass, three_pass.rs:21-104) against the oracle's restatement:
bit-exact status, score, ranges and CIGAR (including the reference's way of adding the outer soft clips)."""
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


def test_doc_vector(za, oracle):
    # src/alignment/profile_set.rs:183-209
    m = za.WeightMatrix.new_dna_matrix(4, -2, b"N")
    a = za.LocalProfilesBatch.new_with_w256([b"CGTTCGCCATAAAGGGGG"], m, -3, -1).sw_align_from_i8_3pass(
        za.SeqSrc.Reference(b"ATGCATCGATCGATCGATCGATCGATCGATGC"))
    assert int(a.status[0]) == S_ and int(a.records[0]["score"]) == 26


@pytest.mark.parametrize("scheme", [(2, -5, -10, -1), (4, -2, -3, -1), (3, -1, -4, -1), (1, -1, 0, 0)])
def test_random_pairs_all_routes(za, oracle, scheme):
    ma, mi, go, ge = scheme
    m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
    sc = osc(oracle, m, go, ge)
    rng = np.random.default_rng(stable_seed(scheme))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 300))
    reads = []
    for _ in range(250):
        L = int(rng.integers(10, 90))
        if rng.random() < 0.75:
            s0 = int(rng.integers(0, 300 - L))
            r = bytearray(ref[s0 : s0 + L])
            for _ in range(int(rng.integers(0, 5))):
                k = int(rng.integers(0, len(r)))
                t = rng.random()
                if t < 0.4:
                    r[k] = int(rng.choice(alpha))
                elif t < 0.7 and len(r) > 1:
                    del r[k]
                else:
                    r.insert(k, int(rng.choice(alpha)))
            reads.append(bytes(r))
        else:
            reads.append(bytes(rng.choice(alpha[:2], L)))
    got = za.StripedProfileBatch(reads, m, go, ge, "i16", 16).sw_align_3pass(za.SeqSrc.Reference(ref))
    hows = set()
    for i, rd in enumerate(reads):
        want, how = oracle.align_3pass("i16", 16, sc, rd, ref)
        assert got.key(i) == okey(want), (i, how, rd)
        if want.status == S_:
            hows.add(how)
    assert {0, 1} <= hows or {0, 2} <= hows
    inv = za.StripedProfileBatch(reads, m, go, ge, "i16", 16).sw_align_3pass(za.SeqSrc.Query(ref))
    for i, rd in enumerate(reads[:80]):
        want, _ = oracle.align_3pass("i16", 16, sc, rd, ref, other_is_query=True)
        assert inv.key(i) == okey(want), i


def test_config3_shape_cascade_and_consistency(za, oracle):
    """150 bp reads vs 2 kb: sw_align_from_i8_3pass equals the oracle; its scores equal sw_align_from_i8's and every CIGAR
    re-scores to the score; most reads take the no-gaps shortcut."""
    import torch

    from zoe_amd import synth

    ref = synth.reference_host(2000)
    n = 4000
    host = synth.reads_host(ref, 5000, n, 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    sc = osc(oracle, dna, -10, -1)
    rb = za.ReadBatch.from_fixed(torch.from_numpy(host.reshape(-1)).cuda(), 150)
    prof = za.into_local_profile(rb, dna, -10, -1)
    got = prof.sw_align_from_i8_3pass(za.SeqSrc.Reference(ref))
    exact = prof.sw_align_from_i8(za.SeqSrc.Reference(ref))
    assert np.array_equal(got.status, exact.status) and np.array_equal(got.records["score"], exact.records["score"])
    n_nogap = 0
    for i in range(n):
        want, tier, how = oracle.cascade_align_3pass(8, 256, sc, host[i], ref)
        assert got.key(i) == okey(want), (i, how)
        if want.status == S_:
            assert int(got.tier[i]) == tier
            n_nogap += how == 0
            rr = got.records[i]
            assert oracle.score_from_path(sc, host[i], ref[int(rr["ref_start"]) : int(rr["ref_end"])], got.cigar(i)) == int(rr["score"])
    assert n_nogap > n // 2


def test_large_box_and_many_ciglets(za, oracle):
    """A 700-base free-extension deletion makes the bounding box far larger than the default slot (rerun with full-size
    resources); cheap gaps give more than 32 ciglets."""
    rng = np.random.default_rng(21)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 1200))
    m = za.WeightMatrix.new_dna_matrix(3, -4, b"N")
    sc = osc(oracle, m, -5, 0)
    reads = [ref[100:160] + ref[860:920], ref[300:380], ref[500:540] + ref[900:940]]
    got = za.StripedProfileBatch(reads, m, -5, 0, "i16", 16).sw_align_3pass(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        want, how = oracle.align_3pass("i16", 16, sc, rd, ref)
        assert got.key(i) == okey(want), (i, how)
    src = ref[0:400]
    rd = bytes(b for k, b in enumerate(src) if k % 6 != 5)
    m2 = za.WeightMatrix.new_dna_matrix(5, -9, b"N")
    got = za.StripedProfileBatch([rd], m2, -1, -1, "i16", 16).sw_align_3pass(za.SeqSrc.Reference(ref))
    want, how = oracle.align_3pass("i16", 16, osc(oracle, m2, -1, -1), rd, ref)
    assert got.key(0) == okey(want) and want.n_ciglets > 32


def test_three_pass_with_a_25_letter_alphabet(za, oracle):
    """Ranges + 3-pass alignment with a protein-sized alphabet: forward and reverse passes on the WIDE packed kernels
    (zsw_score_wide.hip), third pass generic in S."""
    rng = np.random.default_rng(17)
    keys = b"ACDEFGHIKLMNPQRSTVWYBJZX*"
    mp = za.ByteIndexMap.new(keys, b"X")
    w = rng.integers(-4, 3, size=(25, 25))
    w = np.minimum(w, w.T)
    np.fill_diagonal(w, rng.integers(4, 12, size=25))
    w = w.astype(np.int8)
    m = za.WeightMatrix.new_custom(mp, w)
    sc = oracle.Scoring(w, mp.index_map, -11, -1)
    alpha = np.frombuffer(keys[:20], dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 900))
    reads = []
    for i in range(120):
        L = int(rng.integers(20, 180))
        s0 = int(rng.integers(0, 900 - L))
        r = bytearray(ref[s0:s0 + L])
        for _ in range(int(rng.integers(0, 5))):
            k = int(rng.integers(0, len(r)))
            t = rng.random()
            if t < 0.5:
                r[k] = int(rng.choice(alpha))
            elif t < 0.75 and len(r) > 5:
                del r[k]
            else:
                r.insert(k, int(rng.choice(alpha)))
        reads.append(bytes(r) if i % 7 else bytes(rng.choice(alpha, L)))
    got = za.LocalProfilesBatch.new_with_w256(reads, m, -11, -1).sw_align_from_i8_3pass(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        want, tier, _ = oracle.cascade_align_3pass(8, 256, sc, rd, ref)
        assert got.key(i) == okey(want), i


def test_long_reads_take_band_sized_slots(za, oracle):
    """3 kb reads: the bounding box (9 MB of flags) is far beyond a first-launch slot, the banded attempts of reads with few
    indels are not (rlen x (2*band + 1) bytes) and run there; a read with a 300-base deletion needs a wide band and goes to the
    full-size rerun. Routes (how: 1 banded, 2 scalar) and alignments equal the oracle's."""
    rng = np.random.default_rng(stable_seed("3pass-long"))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 9000))
    m = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    go, ge = -10, -1
    sc = osc(oracle, m, go, ge)
    reads = []
    for k in range(10):
        s0 = int(rng.integers(0, 5500))
        r = bytearray(ref[s0 : s0 + 3000])
        for _ in range(6):
            j = int(rng.integers(10, len(r) - 10))
            t = rng.random()
            if t < 0.4:
                r[j] = int(rng.choice(alpha))
            elif t < 0.7:
                del r[j]
            else:
                r.insert(j, int(rng.choice(alpha)))
        reads.append(bytes(r))
    reads.append(ref[1000:2500] + ref[2800:4300])  # one 300-base deletion
    got = za.StripedProfileBatch(reads, m, go, ge, "i32", 8).sw_align_3pass(za.SeqSrc.Reference(ref))
    hows = set()
    for i, rd in enumerate(reads):
        want, how = oracle.align_3pass("i32", 8, sc, rd, ref)
        hows.add(how)
        assert got.key(i) == okey(want), (i, how)
    assert 1 in hows


def test_box_sized_rerun_slot_counts_the_row_padding(za, oracle):
    """Regression (found by tools/fuzz_gpu.py): the rerun's slots are sized from the largest box; the two DP rows are
    16-byte aligned inside a slot, so an odd query range needs 8 bytes more than rlen*qlen + 8*qlen + 32. Boxes are chosen
    so that the unpadded size, rounded to the launch's 64-byte slot granularity, would be too small."""
    rng = np.random.default_rng(stable_seed("3pass-pad"))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = bytes(rng.choice(alpha, 1500))
    m = za.WeightMatrix.new_dna_matrix(3, -4, b"N")
    go, ge = -5, 0  # free extension: one long deletion inside the alignment
    sc = osc(oracle, m, go, ge)
    picks = []
    for gap in range(500, 900):
        for b1 in range(40, 70):
            qlen, rlen = 2 * b1 + 1, 2 * b1 + 1 + gap
            old = rlen * qlen + 8 * qlen + 32
            new = rlen * qlen + ((8 * qlen + 15) & ~15) + 32
            if (old + 63) // 64 * 64 < new and rlen * qlen > 96 * 1024:
                picks.append((b1, gap))
        if len(picks) >= 3:
            break
    assert picks
    reads = [ref[100 : 100 + b1] + ref[100 + b1 + gap : 100 + b1 + gap + b1 + 1] for b1, gap in picks[:3]]
    got = za.StripedProfileBatch(reads, m, go, ge, "i16", 16).sw_align_3pass(za.SeqSrc.Reference(ref))
    for i, rd in enumerate(reads):
        want, how = oracle.align_3pass("i16", 16, sc, rd, ref)
        assert got.key(i) == okey(want), (i, how, picks[i])
