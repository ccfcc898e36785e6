"""Synthetic benchmark inputs: seed-42 uniform-ACGT reference, seed-1337 counter-based reads
sampled from it (1 % substitutions, 0.1 % insertions, 0.1 % deletions, 0.5 % N, 2 % fully random reads).

The generator itself is C (zoe_amd/csrc/zsw_synth.h), exported by the HIP library both as a device kernel
and as a host twin, so every rank can regenerate exactly its own shard and tests can read the same bytes.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .alignment import ReadBatch, SwContext

REF_SEED = 42
READ_SEED = 1337


def reference_host(length: int, seed: int = REF_SEED) -> bytes:
    out = np.zeros(max(length, 1), dtype=np.uint8)
    _lib.load().zsw_synth_reference_host(seed, length, out.ctypes.data)
    return out[:length].tobytes()


def reads_host(reference: bytes, first: int, n: int, length: int, seed: int = READ_SEED) -> np.ndarray:
    ref = np.frombuffer(reference, dtype=np.uint8)
    out = np.zeros((n, length), dtype=np.uint8)
    _lib.load().zsw_synth_reads_host(seed, first, n, length, ref.ctypes.data, len(reference), out.ctypes.data)
    return out


def ragged_lengths(first: int, n: int, min_len: int, max_len: int, seed: int = READ_SEED) -> np.ndarray:
    lib = _lib.load()
    return np.array([lib.zsw_synth_length(seed, first + i, min_len, max_len) for i in range(n)], dtype=np.int64)


def reads_ragged_host(reference: bytes, first: int, n: int, min_len: int, max_len: int, seed: int = READ_SEED):
    lens = ragged_lengths(first, n, min_len, max_len, seed)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    ref = np.frombuffer(reference, dtype=np.uint8)
    out = np.zeros(int(off[-1]), dtype=np.uint8)
    _lib.load().zsw_synth_reads_ragged_host(seed, first, n, min_len, max_len, off.ctypes.data, ref.ctypes.data, len(reference), out.ctypes.data)
    return out, off.astype(np.int64)


def reads_device(ctx: SwContext, reference: bytes, first: int, n: int, length: int, seed: int = READ_SEED) -> ReadBatch:
    """Generates reads [first, first+n) directly in HBM (the context's reference is set to `reference`)."""
    import torch

    ctx.set_reference(reference)
    bases = torch.empty(max(n * length, 1), dtype=torch.uint8, device=torch.device("cuda", ctx.device))
    ctx.check(ctx.lib.zsw_synth_reads(ctx.h, seed, first, n, length, bases.data_ptr(), ctx.stream()))
    return ReadBatch(bases, n, fixed_len=length, min_len=length)


def reads_ragged_device(ctx: SwContext, reference: bytes, first: int, n: int, min_len: int, max_len: int,
                        seed: int = READ_SEED) -> ReadBatch:
    import torch

    ctx.set_reference(reference)
    dev = torch.device("cuda", ctx.device)
    lens = torch.from_numpy(ragged_lengths(first, n, min_len, max_len, seed))
    off = torch.zeros(n + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    total = int(off[-1])
    off_d = off.to(dev)
    bases = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    ctx.check(ctx.lib.zsw_synth_reads_ragged(ctx.h, seed, first, n, min_len, max_len, off_d.data_ptr(), bases.data_ptr(), ctx.stream()))
    return ReadBatch(bases, n, offsets=off_d, min_len=int(lens.min()) if n else None)


def diverged_reads_device(ctx: SwContext, reference: bytes, n: int, length: int, substitutions: float, indels: float | None = None,
                          unrelated: float = 0.02, seed: int = 20261005) -> ReadBatch:
    """n reads of `length` bases in HBM: pieces of `reference` with the given share of substituted bases, `indels` (default: a tenth
    of that, half insertions, half deletions, single bases) and `unrelated` fully random reads — the divergence sweep of bench.py
    and tools/bench_divergence.py (torch only: no part of the library)."""
    import torch

    ctx.set_reference(reference)
    dev = torch.device("cuda", ctx.device)
    if indels is None:
        indels = substitutions / 10.0
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ref = torch.frombuffer(bytearray(reference), dtype=torch.uint8).to(dev)
    R = ref.numel()
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    code = torch.full((256,), 0, dtype=torch.int64, device=dev)
    code[acgt.long()] = torch.arange(4, device=dev)
    out = torch.empty((n, length), dtype=torch.uint8, device=dev)
    chunk = 1 << 20
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = torch.rand((m, length), device=dev, generator=g)
        is_ins = u < indels / 2
        is_del = (u >= indels / 2) & (u < indels)
        is_sub = (u >= indels) & (u < indels + substitutions)
        step = torch.where(is_ins, 0, torch.where(is_del, 2, 1)).to(torch.int64)
        start = torch.randint(0, max(R - length - 8, 1), (m, 1), device=dev, generator=g)
        src = start + torch.cumsum(step, 1) - 1
        rnd = torch.randint(0, 4, (m, length), device=dev, generator=g)
        base = ref[src.clamp(0, R - 1)]
        sub = acgt[(code[base.long()] + 1 + rnd % 3) % 4]
        b = torch.where(is_ins | (src >= R), acgt[rnd], torch.where(is_sub, sub, base))
        junk = torch.rand((m, 1), device=dev, generator=g) < unrelated
        out[lo:lo + m] = torch.where(junk, acgt[rnd], b)
    return ReadBatch(out.reshape(-1), n, fixed_len=length, min_len=length)
