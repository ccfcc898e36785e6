import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/zoe_amd") else ".")
import numpy as np
import zoe_amd as za
from oracle import oracle
oracle.build()
S_=0
def okey(a): return a.key() if a.status == S_ else (a.status, 0, (0, 0), (0, 0), "", 0, 0)
rng = np.random.default_rng(5)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
for it in range(3):
    R = 3200
    ref = bytes(rng.choice(alpha, R))
    m = za.WeightMatrix.new_dna_matrix(2, -5, b"N"); go, ge = -10, -1
    sc = oracle.Scoring(m.signed_weights(), m.mapping.index_map, go, ge)
    reads = []
    for _ in range(6):
        L = int(rng.integers(2440, 3000))
        s0 = int(rng.integers(0, R - L))
        r = bytearray(ref[s0:s0+L])
        for _ in range(40):
            k = int(rng.integers(0, len(r))); u = rng.random()
            if u < 0.5: r[k] = int(rng.choice(alpha))
            elif u < 0.75: del r[k]
            else: r.insert(k, int(rng.choice(alpha)))
        reads.append(bytes(r))
    reads.append(bytes(rng.choice(alpha, 2500)))
    reads.append(ref[100:250])
    for T, N in (("i16", 16), ("i32", 32), ("i16", 64)):
        p = za.StripedProfileBatch(reads, m, go, ge, T, N)
        g_sc, g_en, g_rg = p.sw_score(ref), p.sw_score_ends(za.SeqSrc.Reference(ref)), p.sw_score_ranges(za.SeqSrc.Reference(ref))
        g_al, g_3p = p.sw_align(za.SeqSrc.Reference(ref)), p.sw_align_3pass(za.SeqSrc.Reference(ref))
        for i, rd in enumerate(reads):
            st, s = oracle.score(T, N, sc, rd, ref)
            assert (int(g_sc.status[i]), int(g_sc.score[i]) if st == S_ else 0) == (st, s if st == S_ else 0), ("score", it, i, T, N)
            st, (s, re_, qe) = oracle.score_ends(T, N, sc, rd, ref)
            if st == S_: assert (int(g_en.score[i]), int(g_en.ref_end[i]), int(g_en.query_end[i])) == (s, re_, qe), ("ends", it, i, T, N)
            st, s, rr, qr = oracle.score_ranges(T, N, sc, rd, ref)
            assert int(g_rg.status[i]) == st
            if st == S_: assert (int(g_rg.score[i]), (int(g_rg.ref_start[i]), int(g_rg.ref_end[i])), (int(g_rg.query_start[i]), int(g_rg.query_end[i]))) == (s, rr, qr), ("ranges", it, i, T, N)
            assert g_al.key(i) == okey(oracle.align(T, N, sc, rd, ref)), ("align", it, i, T, N)
            assert g_3p.key(i) == okey(oracle.align_3pass(T, N, sc, rd, ref)[0]), ("3pass", it, i, T, N)
        print("ok", it, T, N, [int(x) for x in g_sc.score[:4]], flush=True)
print("LONG FUZZ OK")
