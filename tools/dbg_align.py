import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle
import zoe_amd as za
scheme=(4,-2,-3,-1)
ma,mi,go,ge=scheme
rng = np.random.default_rng(hash(scheme) % (2**32))
m = za.WeightMatrix.new_dna_matrix(ma, mi, b"N")
sc = oracle.Scoring(m.signed_weights(), m.mapping.index_map, go, ge)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
ref = bytes(rng.choice(alpha, 160))
reads = []
for _ in range(120):
    L = int(rng.integers(8, 60))
    if rng.random() < 0.7:
        s = int(rng.integers(0, 160 - L)); r = bytearray(ref[s : s + L])
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(0, len(r))); t = rng.random()
            if t < 0.4: r[k] = int(rng.choice(alpha))
            elif t < 0.7: del r[k]
            else: r.insert(k, int(rng.choice(alpha)))
        reads.append(bytes(r) if r else b"A")
    else:
        reads.append(bytes(rng.choice(alpha[:2], L)))
N=int(sys.argv[1]) if len(sys.argv)>1 else 2
got = za.StripedProfileBatch(reads, m, go, ge, "i16", N).sw_align(za.SeqSrc.Reference(ref))
bad=0
for i,rd in enumerate(reads):
    w=oracle.align("i16",N,sc,rd,ref)
    wk = w.key() if w.status==0 else (w.status,0,(0,0),(0,0),"",0,0)
    if got.key(i)!=wk:
        bad+=1
        if bad<6: print(i,len(rd),"nv",(len(rd)+N-1)//N,"\n  got ",got.key(i),got.records[i],"\n  want",wk, w.n_ciglets)
print("bad",bad,"of",len(reads))
