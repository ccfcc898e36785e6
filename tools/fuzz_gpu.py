"""Randomised parity sweep: random scoring schemes, alphabets, lengths, lane counts and batch shapes through every GPU entry
point (score, ends, ranges, cascades, exact and 3-pass alignment, SeqSrc inversion; the shared-profile role's score, ends, ranges
exact and 3-pass alignment with the reference as the profile sequence) against the oracle.
usage: python tools/fuzz_gpu.py [iterations] [seed]     (FUZZ_PRUNE=1: the seeded exact first pass for batches of every size, FUZZ_PRUNE=strip: the column-pruned one; FUZZ_LONG_P: share of long-read cases)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zoe_amd as za
from oracle import oracle

S_ = 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle.build()
if os.environ.get("FUZZ_PRUNE"):  # every first pass that qualifies takes the column-pruned pass (batches of any size)
    from zoe_amd import _lib as _l
    za.SwContext.get(0).debug_set(_l.DEBUG_SCORE_PRUNE_ANY_SIZE | (_l.DEBUG_PRUNE_STRIP if os.environ["FUZZ_PRUNE"] == "strip" else 0))


def okey(a):
    return a.key() if a.status == S_ else (a.status, 0, (0, 0), (0, 0), "", 0, 0)


t_start = time.time()
n_checked = 0
for it in range(iters):
    rng = np.random.default_rng(seed0 * 100003 + it)
    protein = rng.random() < 0.25
    if protein:
        S = int(rng.integers(8, 33))
        keys = bytes(range(65, 65 + S))
        mp = za.ByteIndexMap.new(keys, keys[-1:])
        w = rng.integers(-6, 3, size=(S, S))
        w = np.minimum(w, w.T)
        np.fill_diagonal(w, rng.integers(2, 12, size=S))
        if it % 3 == 0:  # an asymmetric matrix: the shared role scores with the transposed table
            w[rng.integers(0, S, size=S), rng.integers(0, S, size=S)] -= 1
        m = za.WeightMatrix.new_custom(mp, w.astype(np.int8))
        alpha = np.frombuffer(keys, dtype=np.uint8)
    else:
        ma, mi = int(rng.integers(1, 9)), -int(rng.integers(0, 9))
        m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N" if rng.random() < 0.7 else None)
        alpha = np.frombuffer(b"ACGT" if rng.random() < 0.8 else b"ACGTN", dtype=np.uint8)
    go = -int(rng.integers(0, 14))
    ge = -int(rng.integers(0, -go + 1))
    sc = oracle.Scoring(m.signed_weights(), m.mapping.index_map, go, ge)
    R = int(rng.choice([40, 200, 700, 2500]))
    lo_c = rng.random() < 0.3
    ref = bytes(rng.choice(alpha[:2] if lo_c else alpha, R))
    reads = []
    for _ in range(int(rng.integers(20, 70))):
        L = int(rng.integers(1, min(330, R + 40)))
        t = rng.random()
        if t < 0.6 and L < R:
            s0 = int(rng.integers(0, R - L))
            r = bytearray(ref[s0:s0 + L])
            for _ in range(int(rng.integers(0, 2 + L // 10))):
                k = int(rng.integers(0, len(r)))
                u = rng.random()
                if u < 0.5:
                    r[k] = int(rng.choice(alpha))
                elif u < 0.75 and len(r) > 1:
                    del r[k]
                else:
                    r.insert(k, int(rng.choice(alpha)))
            reads.append(bytes(r))
        else:
            reads.append(bytes(rng.choice(alpha, L)))
    long_case = rng.random() < float(os.environ.get("FUZZ_LONG_P", "0.12"))
    if long_case:  # a few reads beyond the widest strip configuration: tiled / 32-bit / HBM-row paths
        R = int(rng.integers(3000, 7000))
        ref = bytes(rng.choice(alpha[:2] if lo_c else alpha, R))
        reads = reads[:6]
        for _ in range(2):
            L = int(rng.integers(2433, 5200))
            r = bytearray((ref * 3)[int(rng.integers(0, R)):][:L])
            for _ in range(L // 50):
                k = int(rng.integers(0, len(r)))
                u = rng.random()
                if u < 0.5:
                    r[k] = int(rng.choice(alpha))
                elif u < 0.75 and len(r) > 1:
                    del r[k]
                else:
                    r.insert(k, int(rng.choice(alpha)))
            reads.append(bytes(r))
        reads = [r for r in reads if len(r) <= R + 40 or len(r) > 2432]
    if not long_case and rng.random() < 0.3:  # fixed-length batch
        L = int(rng.integers(5, 200))
        reads = [(r * (L // len(r) + 1))[:L] for r in reads]
    T, N = [("i16", 16), ("i8", 32), ("i16", 8), ("i32", 8), ("i8", 16), ("i16", 4), ("i16", 32), ("i8", 64)][int(rng.integers(0, 8))]
    if long_case:
        T, N = [("i16", 16), ("i32", 32), ("i16", 64), ("i32", 16)][int(rng.integers(0, 4))]
    preset = int(rng.choice([128, 256, 512]))
    width = int(rng.choice([8, 16, 32]))
    inv = bool(rng.random() < 0.3)
    src = za.SeqSrc.Query(ref) if inv else za.SeqSrc.Reference(ref)
    p = za.StripedProfileBatch(reads, m, go, ge, T, N)
    g_sc, g_en, g_rg = p.sw_score(ref), p.sw_score_ends(za.SeqSrc.Reference(ref)), p.sw_score_ranges(za.SeqSrc.Reference(ref))
    g_al, g_3p = p.sw_align(src), p.sw_align_3pass(src)
    lp = za.LocalProfilesBatch(reads, m, go, ge, preset=preset)
    c_sc = lp._score_from(ref, width)
    c_al = lp._align(src, None, from_width=width, preset=preset)
    c_3p = lp._align(src, None, from_width=width, preset=preset, three_pass=True)
    c_rg = lp._ranges_from(za.SeqSrc.Reference(ref), width)
    # the shared-profile role: the reference carries the profile, the reads are walked row by row (zsw_*_shared_batch)
    shared = not long_case and all(len(r) > 0 for r in reads)
    if shared:
        sp = za.SharedStripedProfile(ref, m, go, ge, T, N)
        s_sc, s_en = sp.sw_score(reads), sp.sw_score_ends(za.SeqBatchSrc.Reference(reads))
        s_rg = sp.sw_score_ranges(za.SeqBatchSrc.Reference(reads))
        s_al = sp.sw_align((za.SeqBatchSrc.Query if inv else za.SeqBatchSrc.Reference)(reads[:16]))
        s_3p = sp.sw_align_3pass((za.SeqBatchSrc.Query if inv else za.SeqBatchSrc.Reference)(reads))  # profile.rs:536-552 with the shared profile
    # the same batch through HOST pointers (staging path of the C ABI) and the sneaky_snake filter on random windows
    import ctypes as C
    from zoe_amd import _lib
    lib = _lib.load()
    cat = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    hb = _lib.ZswBatch()
    hb.bases, hb.offsets, hb.fixed_len, hb.n_reads, hb.mem = cat.ctypes.data, offs.ctypes.data, 0, len(reads), _lib.MEM_HOST
    h_score = np.zeros(len(reads), dtype=np.uint32); h_status = np.zeros(len(reads), dtype=np.uint8); h_tier = np.zeros(len(reads), dtype=np.uint8)
    assert lib.zsw_score_batch_from(lp.ctx.h, C.byref(hb), width, preset, h_score.ctypes.data, h_status.ctypes.data, h_tier.ctypes.data, None) == 0
    assert np.array_equal(h_status, c_sc.status.cpu().numpy()) and np.array_equal(np.where(h_status == 0, h_score, 0), np.where(h_status == 0, c_sc.score.cpu().numpy().view(np.uint32), 0)), ("host score", it)
    thr = float(rng.choice([0.0, 0.05, 0.1, 0.3, 1.0]))
    ws = [int(rng.integers(0, max(1, R - len(r) + 1))) if len(r) <= R else 0 for r in reads]
    wl = [int(np.clip(len(r) + rng.integers(-2, 3), 0, R - s0)) for r, s0 in zip(reads, ws)]
    g_ss = za.sneaky_snake(ref, reads, ws, wl, thr).cpu().numpy()
    for i, rd in enumerate(reads):
        want = oracle.sneaky_snake(ref[ws[i]:ws[i] + wl[i]], rd, thr)
        assert int(g_ss[i]) == {False: 0, True: 1, None: 2}[want], ("sneaky", it, i, thr, ws[i], wl[i])
    for i, rd in enumerate(reads):
        ctxt = (it, i, T, N, preset, width, inv, go, ge, protein)
        st, s = oracle.score(T, N, sc, rd, ref)
        assert (int(g_sc.status[i]), int(g_sc.score[i]) if st == S_ else 0) == (st, s if st == S_ else 0), ("score", ctxt)
        st, (s, re_, qe) = oracle.score_ends(T, N, sc, rd, ref)
        if st == S_:
            assert (int(g_en.score[i]), int(g_en.ref_end[i]), int(g_en.query_end[i])) == (s, re_, qe), ("ends", ctxt)
        st, s, rr, qr = oracle.score_ranges(T, N, sc, rd, ref)
        assert int(g_rg.status[i]) == st, ("ranges status", ctxt)
        if st == S_:
            assert (int(g_rg.score[i]), (int(g_rg.ref_start[i]), int(g_rg.ref_end[i])), (int(g_rg.query_start[i]), int(g_rg.query_end[i]))) == (s, rr, qr), ("ranges", ctxt)
        assert g_al.key(i) == okey(oracle.align(T, N, sc, rd, ref, other_is_query=inv)), ("align", ctxt)
        assert g_3p.key(i) == okey(oracle.align_3pass(T, N, sc, rd, ref, other_is_query=inv)[0]), ("3pass", ctxt)
        st, s, tier = oracle.cascade_score(width, preset, sc, rd, ref)
        assert (int(c_sc.status[i]), int(c_sc.score[i]) if st == S_ else 0) == (st, s if st == S_ else 0), ("cascade score", ctxt)
        want, tier = oracle.cascade_align(width, preset, sc, rd, ref, other_is_query=inv)
        assert c_al.key(i) == okey(want), ("cascade align", ctxt)
        want = oracle.cascade_align_3pass(width, preset, sc, rd, ref, other_is_query=inv)[0]
        assert c_3p.key(i) == okey(want), ("cascade 3pass", ctxt)
        st, s, rr, qr, tier = oracle.cascade_score_ranges(width, preset, sc, rd, ref)
        assert int(c_rg.status[i]) == st, ("cascade ranges status", ctxt)
        if st == S_:
            assert (int(c_rg.score[i]), (int(c_rg.ref_start[i]), int(c_rg.ref_end[i])), (int(c_rg.query_start[i]), int(c_rg.query_end[i])), int(c_rg.tier[i])) == (s, rr, qr, tier), ("cascade ranges", ctxt)
        if shared:
            st, s = oracle.score(T, N, sc, ref, rd)
            assert (int(s_sc.status[i]), int(s_sc.score[i]) if st == S_ else 0) == (st, s if st == S_ else 0), ("shared score", ctxt)
            st, (s, re_, qe) = oracle.score_ends(T, N, sc, ref, rd)
            assert int(s_en.status[i]) == st, ("shared ends status", ctxt)
            if st == S_:
                assert (int(s_en.score[i]), int(s_en.ref_end[i]), int(s_en.query_end[i])) == (s, re_, qe), ("shared ends", ctxt)
            st, s, rr, qr = oracle.score_ranges(T, N, sc, ref, rd)
            assert int(s_rg.status[i]) == st, ("shared ranges status", ctxt)
            if st == S_:
                assert (int(s_rg.score[i]), (int(s_rg.ref_start[i]), int(s_rg.ref_end[i])), (int(s_rg.query_start[i]), int(s_rg.query_end[i]))) == (s, rr, qr), ("shared ranges", ctxt)
            if i < 16:
                assert s_al.key(i) == okey(oracle.align(T, N, sc, ref, rd, other_is_query=inv)), ("shared align", ctxt)
            assert s_3p.key(i) == okey(oracle.align_3pass(T, N, sc, ref, rd, other_is_query=inv)[0]), ("shared 3pass", ctxt)
        n_checked += 1
    print(f"iteration {it}: ok ({len(reads)} reads, R={R}, T={T}x{N}, preset {preset} from i{width}, go={go} ge={ge}, {'protein S=%d' % len(m.mapping) if protein else 'dna'}, invert={inv}) [{time.time() - t_start:.0f} s]", flush=True)
print(f"FUZZ OK: {n_checked} reads x 12 entry points + 5 of the shared-profile role")
