"""PCIe-inclusive rate: the headline workload with reads and results in HOST memory (zsw_batch.mem = ZSW_MEM_HOST),
pageable and pinned.  usage: python tools/bench_host.py [n_reads]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zoe_amd
from zoe_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 150
ctx = zoe_amd.SwContext.get(0)
ref = synth.reference_host(2000)
ctx.set_reference(ref)
ctx.set_scoring(zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N"), -10, -1)
dev_reads = synth.reads_device(ctx, ref, 0, n, L)
lib = _lib.load()
for kind in ("pageable", "pinned"):
    if kind == "pageable":
        bases = dev_reads.bases.cpu().numpy().copy()
        score = np.zeros(n, dtype=np.uint32); status = np.zeros(n, dtype=np.uint8); tier = np.zeros(n, dtype=np.uint8)
        pb, ps, pst, pt = bases.ctypes.data, score.ctypes.data, status.ctypes.data, tier.ctypes.data
    else:
        tb = dev_reads.bases.cpu().pin_memory()
        ts = torch.zeros(n, dtype=torch.int32).pin_memory(); tst = torch.zeros(n, dtype=torch.uint8).pin_memory(); tt = torch.zeros(n, dtype=torch.uint8).pin_memory()
        pb, ps, pst, pt = tb.data_ptr(), ts.data_ptr(), tst.data_ptr(), tt.data_ptr()
    b = _lib.ZswBatch()
    b.bases, b.offsets, b.fixed_len, b.n_reads, b.mem = pb, None, L, n, _lib.MEM_HOST
    for rep in range(3):
        t0 = time.perf_counter()
        rc = lib.zsw_score_batch_from(ctx.h, C.byref(b), 8, 256, ps, pst, pt, None)
        dt = time.perf_counter() - t0
        assert rc == 0
        print(f"{kind} rep {rep}: {n / dt / 1e6:.2f} M reads/s ({dt * 1e3:.0f} ms per {n} reads, host in -> host out)", flush=True)

# ZSW_ENCODING_PACKED4: two residue indices per byte (zsw_pack4_host), pinned host memory: half the bytes cross PCIe
tb = dev_reads.bases.cpu()
packed = torch.zeros(n * ((L + 1) // 2), dtype=torch.uint8).pin_memory()
t0 = time.perf_counter()
assert lib.zsw_pack4_host(ctx.h, tb.data_ptr(), n, L, packed.data_ptr()) == 0
print(f"zsw_pack4_host: {time.perf_counter() - t0:.2f} s for {n} reads (one host core)", flush=True)
ts2 = torch.zeros(n, dtype=torch.int32).pin_memory(); tst2 = torch.zeros(n, dtype=torch.uint8).pin_memory(); tt2 = torch.zeros(n, dtype=torch.uint8).pin_memory()
b = _lib.ZswBatch()
b.bases, b.offsets, b.fixed_len, b.n_reads, b.mem, b.encoding = packed.data_ptr(), None, L, n, _lib.MEM_HOST, 1
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.zsw_score_batch_from(ctx.h, C.byref(b), 8, 256, ts2.data_ptr(), tst2.data_ptr(), tt2.data_ptr(), None)
    dt = time.perf_counter() - t0
    assert rc == 0
    print(f"packed4 pinned rep {rep}: {n / dt / 1e6:.2f} M reads/s ({dt * 1e3:.0f} ms per {n} reads, host in -> host out)", flush=True)
assert torch.equal(ts2, ts) and torch.equal(tst2, tst) and torch.equal(tt2, tt), "packed input: results differ from the byte form"
print("packed4 results identical to the byte form", flush=True)
