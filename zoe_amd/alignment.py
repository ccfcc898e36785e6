"""Host-side mirror of the reference's interface for the striped Smith-Waterman path, batched over reads.

Names, argument meaning and error behaviour follow the reference (file:line in the reference checkout):

    ByteIndexMap, DNA_PROFILE_MAP        src/data/constants/mappings/byte_index.rs:231-358, dna.rs:177-178
    WeightMatrix                         src/data/matrices/mod.rs:230-546
    ProfileError / validate_profile_args src/alignment/errors.rs:6-15, profile.rs:32-44
    StripedProfile::{new,sw_score,sw_score_ends,sw_align}   src/alignment/profile.rs:239-519
    LocalProfiles::new_with_w{128,256,512} + ProfileSets    src/alignment/profile_set.rs:71-179,434-483
    MaybeAligned                         src/alignment/types/output.rs:18-25
    SeqSrc                               src/alignment/mod.rs:161-190

The reference builds one profile from one sequence and aligns it against many; here a *batch* of reads
plays the profile role (sw/mod.rs:119-120: the profile is built from the query) and every call runs all
of them against one reference on the GPU through the C ABI (include/zoe_sw.h).  PyTorch is used only
to own device memory and streams.  There is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib

SOME, OVERFLOWED, UNMAPPED, EMPTY = 0, 1, 2, 3


class ProfileError(ValueError):
    """src/alignment/errors.rs:6-15"""

    NAMES = {1: "EmptySequence", 2: "GapOpenOutOfRange", 3: "GapExtendOutOfRange", 4: "BadGapWeights"}

    def __init__(self, code: int):
        self.code = code
        self.variant = self.NAMES[code]
        super().__init__(self.variant)


def validate_profile_args(seq_len: int, gap_open: int, gap_extend: int) -> None:
    """src/alignment/profile.rs:32-44"""
    if seq_len == 0:
        raise ProfileError(1)
    if not -127 <= gap_open <= 0:
        raise ProfileError(2)
    if not -127 <= gap_extend <= 0:
        raise ProfileError(3)
    if gap_extend < gap_open:
        raise ProfileError(4)


class ByteIndexMap:
    """256-entry byte -> residue-index table (byte_index.rs:231-358)."""

    def __init__(self, index_map: np.ndarray, byte_keys: bytes):
        self.index_map = np.ascontiguousarray(index_map, dtype=np.uint8)
        self.byte_keys = bytes(byte_keys)

    @staticmethod
    def new(byte_keys: bytes, catch_all: bytes) -> "ByteIndexMap":
        keys = bytes(byte_keys)
        if len(set(keys)) != len(keys):
            raise ValueError("duplicate byte_keys")
        if catch_all[0] not in keys:
            raise ValueError("The catch_all must be present in the byte_keys.")
        m = np.full(256, keys.index(catch_all[0]), dtype=np.uint8)
        for i, k in enumerate(keys):
            m[k] = i
        return ByteIndexMap(m, keys)

    @staticmethod
    def new_ignoring_case(byte_keys: bytes, catch_all: bytes) -> "ByteIndexMap":
        keys = bytes(byte_keys).upper()
        if len(set(keys)) != len(keys):
            raise ValueError("duplicate byte_keys")
        ca = bytes(catch_all).upper()[0]
        if ca not in keys:
            raise ValueError("The catch_all must be present in the byte_keys.")
        m = np.full(256, keys.index(ca), dtype=np.uint8)
        for i, k in enumerate(keys):
            m[k] = i
            m[bytes([k]).lower()[0]] = i
        return ByteIndexMap(m, keys)

    def add_synonym_ignore_case(self, new_key: bytes, previous_key: bytes) -> "ByteIndexMap":
        m = self.index_map.copy()
        idx = m[previous_key[0]]
        m[bytes(new_key).upper()[0]] = idx
        m[bytes(new_key).lower()[0]] = idx
        return ByteIndexMap(m, self.byte_keys)

    def __len__(self) -> int:
        return len(self.byte_keys)

    def to_index(self, b: int) -> int:
        return int(self.index_map[b])


# dna.rs:177-178
DNA_PROFILE_MAP = ByteIndexMap.new_ignoring_case(b"ACGTN", b"N").add_synonym_ignore_case(b"U", b"T")


class WeightMatrix:
    """weights[ref_residue][query_residue] (matrices/mod.rs:230-235). `bias` is non-zero only for the
    unsigned form produced by to_biased_matrix()."""

    def __init__(self, weights: np.ndarray, mapping: ByteIndexMap, bias: int = 0, signed: bool = True):
        self.weights = np.ascontiguousarray(weights)
        self.mapping = mapping
        self.bias = int(bias)
        self.signed = signed

    @staticmethod
    def new(mapping: ByteIndexMap, matching: int, mismatch: int, ignoring: Optional[bytes]) -> "WeightMatrix":
        S = len(mapping)
        w = np.zeros((S, S), dtype=np.int8)
        skip = None
        if ignoring is not None:
            if ignoring[0] not in mapping.byte_keys:
                raise ValueError("An invalid byte was specified for the ignoring field.")
            skip = mapping.to_index(ignoring[0])
        for i in range(S):
            for j in range(S):
                if skip is not None and (skip == i or skip == j):
                    continue
                w[i, j] = matching if i == j else mismatch
        return WeightMatrix(w, mapping)

    @staticmethod
    def new_custom(mapping: ByteIndexMap, weights) -> "WeightMatrix":
        w = np.asarray(weights, dtype=np.int8)
        S = len(mapping)
        if w.shape != (S, S):
            raise ValueError("weights must be SxS")
        return WeightMatrix(w, mapping)

    @staticmethod
    def new_dna_matrix(matching: int, mismatch: int, ignoring: Optional[bytes]) -> "WeightMatrix":
        return WeightMatrix.new(DNA_PROFILE_MAP, matching, mismatch, ignoring)

    @staticmethod
    def new_biased_dna_matrix(matching: int, mismatch: int, ignoring: Optional[bytes]) -> "WeightMatrix":
        return WeightMatrix.new_dna_matrix(matching, mismatch, ignoring).to_biased_matrix()

    def get_bias(self) -> int:
        return int(min(0, int(self.weights.min())))

    def to_biased_matrix(self) -> "WeightMatrix":
        if not self.signed:
            return self
        b = self.get_bias()
        w = (self.weights.astype(np.int16) - b).astype(np.uint8)
        return WeightMatrix(w, self.mapping, bias=-b, signed=False)

    def signed_weights(self) -> np.ndarray:
        if self.signed:
            return self.weights.astype(np.int8)
        return (self.weights.astype(np.int16) - self.bias).astype(np.int8)

    def get_weight(self, ref_residue: int, query_residue: int) -> int:
        return int(self.weights[self.mapping.to_index(ref_residue)][self.mapping.to_index(query_residue)])


@dataclass(frozen=True)
class SeqSrc:
    """src/alignment/mod.rs:161-166: tells an alignment call which role the non-profile sequence plays."""

    seq: bytes
    is_query: bool

    @staticmethod
    def Reference(seq) -> "SeqSrc":
        return SeqSrc(bytes(seq), False)

    @staticmethod
    def Query(seq) -> "SeqSrc":
        return SeqSrc(bytes(seq), True)


# ------------------------------------------------------------------------------------------------
def _torch():
    import torch

    return torch


class SwContext:
    """One zsw_context per GPU (include/zoe_sw.h). Raises if the HIP library or the device is missing."""

    _cache = {}

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        self.device = device
        h = C.c_void_p()
        rc = self.lib.zsw_create(device, C.byref(h))
        if rc != 0:
            raise _lib.ZswError(rc, "zsw_create: " + self.lib.zsw_last_error_string(None).decode())
        self.h = h
        self._scoring_key = None
        self._ref_key = None

    @classmethod
    def get(cls, device: int = 0) -> "SwContext":
        if device not in cls._cache:
            cls._cache[device] = SwContext(device)
        return cls._cache[device]

    def check(self, rc: int, profile_errors: bool = True):
        if rc == 0:
            return
        if profile_errors and 1 <= rc <= 4:
            raise ProfileError(rc)
        raise _lib.ZswError(rc, self.lib.zsw_last_error_string(self.h).decode())

    def set_scoring(self, matrix: WeightMatrix, gap_open: int, gap_extend: int):
        w = np.ascontiguousarray(matrix.signed_weights(), dtype=np.int8)
        key = (w.tobytes(), matrix.mapping.index_map.tobytes(), gap_open, gap_extend)
        if key == self._scoring_key:
            return
        im = matrix.mapping.index_map
        self.check(self.lib.zsw_set_scoring(self.h, w.ctypes.data, w.shape[0], im.ctypes.data, gap_open, gap_extend))
        self._scoring_key = key

    def set_reference(self, reference) -> None:
        torch = _torch()
        if isinstance(reference, torch.Tensor):
            ref = reference.contiguous()
            key = ("t", ref.data_ptr(), ref.numel())
            if key == self._ref_key:
                return
            mem = _lib.MEM_DEVICE if ref.is_cuda else _lib.MEM_HOST
            self.check(self.lib.zsw_set_reference(self.h, ref.data_ptr(), ref.numel(), mem))
            self._ref_key = None  # tensor contents may change; do not cache
            return
        ref = bytes(reference)
        key = ("b", ref)
        if key == self._ref_key:
            return
        buf = np.frombuffer(ref, dtype=np.uint8) if ref else np.zeros(1, dtype=np.uint8)
        self.check(self.lib.zsw_set_reference(self.h, buf.ctypes.data, len(ref), _lib.MEM_HOST))
        self._ref_key = key

    def stream(self) -> int:
        torch = _torch()
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    def selftest(self):
        self.check(self.lib.zsw_selftest(self.h), profile_errors=False)

    def debug_set(self, flags: int = 0):
        """zsw_debug_set: kernel-selection overrides (_lib.DEBUG_*) for the parity tests; 0 restores the defaults."""
        self.check(self.lib.zsw_debug_set(self.h, int(flags)))

    def debug_band_records(self, records=None):
        """zsw_debug_band_records (tests): the banded seeded pass writes 8 int32 per read — the values its decision rests on —
        into `records` (a CUDA int32 tensor of 8 * n_reads elements that the caller keeps alive); None switches it off."""
        if records is not None:
            assert records.is_cuda and records.dtype == _torch().int32 and records.is_contiguous()
        self._band_records = records
        self.check(self.lib.zsw_debug_band_records(self.h, C.c_void_p(records.data_ptr() if records is not None else None)))

    def set_profile_sequence(self, sequence: bytes):
        """zsw_set_profile_sequence: the sequence the shared profile is built from (the library itself skips the work when it is the
        one already set: no copy of it is kept on this side, which another binding of the same context could leave stale)"""
        sequence = bytes(sequence)
        buf = (C.c_uint8 * max(len(sequence), 1)).from_buffer_copy(sequence if sequence else b"\0")
        self.check(self.lib.zsw_set_profile_sequence(self.h, buf, len(sequence), _lib.MEM_HOST))

    def set_option(self, option: int, value: int):
        """zsw_set_option, e.g. set_option(_lib.OPTION_EXACT_PRUNING, 0): every cell of every read instead of the seeded exact pass."""
        self.check(self.lib.zsw_set_option(self.h, int(option), int(value)), profile_errors=False)

    def prune_rescored(self) -> int:
        """zsw_prune_rescored: reads the last score call's seeded (or column-pruned) pass handed back to the full pass."""
        v = C.c_uint64(0)
        self.check(self.lib.zsw_prune_rescored(self.h, C.byref(v)))
        return int(v.value)

    def timing_enable(self, on: bool):
        self.check(self.lib.zsw_timing_enable(self.h, int(on)))

    def timing_read(self):
        s, n = C.c_double(0), C.c_uint64(0)
        self.check(self.lib.zsw_timing_read(self.h, C.byref(s), C.byref(n)))
        return s.value, n.value

    def timing_read_window(self):
        """zsw_timing_read_window: seconds and launches of the seeded pass's window kernel since the last read."""
        s, n = C.c_double(0), C.c_uint64(0)
        self.check(self.lib.zsw_timing_read_window(self.h, C.byref(s), C.byref(n)))
        return s.value, n.value


class SwGroup:
    """zsw_group: several GPUs (or several contexts on one GPU) behind one handle; reads shard into contiguous index ranges,
    one host thread drives each context (include/zoe_sw.h). Host batches get their results in place; device shards are
    completed on every device by one RCCL all-gather."""

    def __init__(self, device_ids: Sequence[int]):
        self.lib = _lib.load()
        self.devices = [int(d) for d in device_ids]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        rc = self.lib.zsw_group_create(arr, len(self.devices), C.byref(h))
        if rc != 0:
            raise _lib.ZswError(rc, "zsw_group_create: " + self.lib.zsw_last_error_string(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.zsw_group_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def check(self, rc: int):
        if rc == 0:
            return
        if 1 <= rc <= 4:
            raise ProfileError(rc)
        raise _lib.ZswError(rc, self.lib.zsw_group_last_error_string(self.h).decode())

    def __len__(self):
        return int(self.lib.zsw_group_size(self.h))

    def configure(self, matrix: WeightMatrix, gap_open: int, gap_extend: int, reference: bytes):
        w = np.ascontiguousarray(matrix.signed_weights(), dtype=np.int8)
        im = matrix.mapping.index_map
        self.check(self.lib.zsw_group_set_scoring(self.h, w.ctypes.data, w.shape[0], im.ctypes.data, gap_open, gap_extend))
        ref = np.frombuffer(bytes(reference), dtype=np.uint8) if len(reference) else np.zeros(1, dtype=np.uint8)
        self.check(self.lib.zsw_group_set_reference(self.h, ref.ctypes.data, len(reference)))

    def sw_score_from_host(self, bases: np.ndarray, n_reads: int, fixed_len: int = 0, offsets: Optional[np.ndarray] = None,
                           width: int = 8, preset: int = 256):
        """ProfileSets::sw_score_from_i{width} for a batch in host memory; returns (score u32, status u8, tier u8) numpy arrays."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        b = _lib.ZswBatch()
        b.bases = bases.ctypes.data
        off = None
        if offsets is not None:
            off = np.ascontiguousarray(offsets, dtype=np.uint64)
            b.offsets = off.ctypes.data
        b.fixed_len = int(fixed_len)
        b.n_reads = int(n_reads)
        b.mem = _lib.MEM_HOST
        score = np.zeros(max(n_reads, 1), dtype=np.uint32)
        status = np.zeros(max(n_reads, 1), dtype=np.uint8)
        tier = np.zeros(max(n_reads, 1), dtype=np.uint8)
        self.check(self.lib.zsw_group_score_batch_from(self.h, C.byref(b), width, preset, score.ctypes.data, status.ctypes.data, tier.ctypes.data))
        return score[:n_reads], status[:n_reads], tier[:n_reads]

    def sw_align_from_host(self, bases: np.ndarray, n_reads: int, fixed_len: int = 0, offsets: Optional[np.ndarray] = None,
                           width: int = 8, preset: int = 256, invert: bool = False, three_pass: bool = False) -> "AlignmentBatch":
        """ProfileSets::sw_align_from_i{width} (or its _3pass form) for a batch in host memory, sharded over the group."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        b = _lib.ZswBatch()
        b.bases = bases.ctypes.data
        off = None
        if offsets is not None:
            off = np.ascontiguousarray(offsets, dtype=np.uint64)
            b.offsets = off.ctypes.data
        b.fixed_len = int(fixed_len)
        b.n_reads = int(n_reads)
        b.mem = _lib.MEM_HOST
        n = int(n_reads)
        rec = np.zeros(max(n, 1), dtype=ALN_DTYPE)
        status = np.zeros(max(n, 1), dtype=np.uint8)
        tier = np.zeros(max(n, 1), dtype=np.uint8)
        fn = self.lib.zsw_group_align_3pass_batch_from if three_pass else self.lib.zsw_group_align_batch_from
        cap = max(8 * n, 64)
        total = C.c_uint64(0)
        while True:
            inc = np.zeros(cap, dtype=np.uint32)
            op = np.zeros(cap, dtype=np.uint8)
            rc = fn(self.h, C.byref(b), width, preset, int(invert), rec.ctypes.data, status.ctypes.data, tier.ctypes.data,
                    inc.ctypes.data, op.ctypes.data, cap, C.byref(total))
            if rc == -1 and total.value > cap:
                cap = int(total.value)
                continue
            self.check(rc)
            break
        t = int(total.value)
        return AlignmentBatch(status[:n], rec[:n], inc[:t], op[:t], tier[:n])

    def sw_score_from_device(self, shards: Sequence["ReadBatch"], width: int = 8, preset: int = 256):
        """One device-resident ReadBatch per context; returns per device the (score, status) tensors of ALL reads in shard order."""
        torch = _torch()
        if len(shards) != len(self.devices):
            raise ValueError("one shard per context")
        total = sum(s.n_reads for s in shards)
        arr = (_lib.ZswBatch * len(shards))(*[s.c_batch() for s in shards])
        outs = [(torch.zeros(max(total, 1), dtype=torch.int32, device=s.bases.device), torch.zeros(max(total, 1), dtype=torch.uint8, device=s.bases.device))
                for s in shards]
        ps = (C.c_void_p * len(shards))(*[o[0].data_ptr() for o in outs])
        pt = (C.c_void_p * len(shards))(*[o[1].data_ptr() for o in outs])
        torch.cuda.synchronize()
        self.check(self.lib.zsw_group_score_batch_from_device(self.h, arr, width, preset, ps, pt))
        return [(o[0][:total], o[1][:total]) for o in outs]


class ReadBatch:
    """Device-resident reads: `bases` uint8 (concatenated), and either `fixed_len` or `offsets` (int64[n+1])."""

    def __init__(self, bases, n_reads: int, fixed_len: int = 0, offsets=None, min_len: Optional[int] = None):
        self.bases = bases
        self.n_reads = int(n_reads)
        self.fixed_len = int(fixed_len)
        self.offsets = offsets
        self.min_len = min_len  # known on the host when built from host data

    @staticmethod
    def from_sequences(seqs: Sequence[bytes], device: int = 0) -> "ReadBatch":
        torch = _torch()
        lens = [len(s) for s in seqs]
        dev = torch.device("cuda", device)
        cat = b"".join(bytes(s) for s in seqs)
        bases = torch.frombuffer(bytearray(cat if cat else b"\0"), dtype=torch.uint8).to(dev)
        if lens and all(l == lens[0] for l in lens) and lens[0] > 0:
            return ReadBatch(bases, len(seqs), fixed_len=lens[0], min_len=lens[0])
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        return ReadBatch(bases, len(seqs), offsets=torch.from_numpy(off).to(dev), min_len=min(lens) if lens else None)

    @staticmethod
    def from_fixed(bases, fixed_len: int) -> "ReadBatch":
        return ReadBatch(bases, bases.numel() // fixed_len, fixed_len=fixed_len, min_len=fixed_len)

    def c_batch(self) -> _lib.ZswBatch:
        b = _lib.ZswBatch()
        b.bases = self.bases.data_ptr()
        b.offsets = self.offsets.data_ptr() if self.offsets is not None else None
        b.fixed_len = self.fixed_len
        b.n_reads = self.n_reads
        b.mem = _lib.MEM_DEVICE
        b.encoding = 0
        return b

    @property
    def device_index(self) -> int:
        return self.bases.device.index or 0


@dataclass
class ScoreBatch:
    """MaybeAligned<u32> per read: status (SOME/OVERFLOWED/UNMAPPED/EMPTY) + score (valid where SOME)."""

    score: "object"
    status: "object"
    tier: Optional["object"] = None
    ref_end: Optional["object"] = None
    query_end: Optional["object"] = None
    ref_start: Optional["object"] = None
    query_start: Optional["object"] = None

    def maybe_aligned(self, i: int):
        st = int(self.status[i])
        return ("Some", int(self.score[i])) if st == SOME else ({1: "Overflowed", 2: "Unmapped", 3: "Empty"}[st], None)


@dataclass
class AlignmentBatch:
    """MaybeAligned<Alignment<u32>> per read; ciglets flattened (inc, op) with per-read offsets."""

    status: np.ndarray
    records: np.ndarray  # structured: score, ref_start, ref_end, query_start, query_end, ref_len, query_len, n_ciglets, ciglet_offset
    inc: np.ndarray
    op: np.ndarray
    tier: Optional[np.ndarray] = None

    def cigar(self, i: int) -> str:
        r = self.records[i]
        o, n = int(r["ciglet_offset"]), int(r["n_ciglets"])
        return "".join(f"{int(self.inc[o + k])}{chr(int(self.op[o + k]))}" for k in range(n))

    def key(self, i: int):
        st = int(self.status[i])
        if st != SOME:
            return (st, 0, (0, 0), (0, 0), "", 0, 0)
        r = self.records[i]
        return (st, int(r["score"]), (int(r["ref_start"]), int(r["ref_end"])), (int(r["query_start"]), int(r["query_end"])),
                self.cigar(i), int(r["ref_len"]), int(r["query_len"]))


ALN_DTYPE = np.dtype(
    [("score", "<u4"), ("ref_start", "<u4"), ("ref_end", "<u4"), ("query_start", "<u4"), ("query_end", "<u4"),
     ("ref_len", "<u4"), ("query_len", "<u4"), ("n_ciglets", "<u4"), ("ciglet_offset", "<u8")]
)


def _to_host(*tensors):
    """Device tensors -> numpy arrays through page-locked buffers (torch caches them between calls), all copies in flight together:
    a 10 M-read alignment hands back half a gigabyte of records and ciglets, which pageable copies move at a fifth of the PCIe rate."""
    torch = _torch()
    out = []
    for t in tensors:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        out.append(h)
    if tensors:
        torch.cuda.synchronize(tensors[0].device)
    return [h.numpy() for h in out]


def _as_batch(reads, device: int) -> ReadBatch:
    if isinstance(reads, ReadBatch):
        return reads
    return ReadBatch.from_sequences(list(reads), device)


class _ProfileBatchBase:
    def __init__(self, reads, matrix: WeightMatrix, gap_open: int, gap_extend: int, device: int = 0):
        rb = _as_batch(reads, device)
        # validate_profile_args: the sequence check is per read; an empty read known on the host raises here,
        # one only visible on the device comes back with status EMPTY.
        validate_profile_args(1 if rb.min_len is None else rb.min_len, gap_open, gap_extend)
        self.reads = rb
        self.matrix = matrix
        self.gap_open = gap_open
        self.gap_extend = gap_extend
        self.ctx = SwContext.get(rb.device_index)

    def _prep(self, reference):
        self.ctx.set_scoring(self.matrix, self.gap_open, self.gap_extend)
        self.ctx.set_reference(reference)

    def _align(self, seq: SeqSrc, direct, from_width=None, preset=None, three_pass: bool = False) -> AlignmentBatch:
        torch = _torch()
        self._prep(seq.seq)
        n = self.reads.n_reads
        dev = self.reads.bases.device
        aln = torch.zeros(max(n, 1) * ALN_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        status = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        tier = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        cap = max(8 * n, 64)  # typical reads need 1-3 ciglets; the call reports the exact total when this is too small
        b = self.reads.c_batch()
        total = C.c_uint64(0)
        while True:
            inc = torch.empty(cap, dtype=torch.int32, device=dev)
            op = torch.empty(cap, dtype=torch.uint8, device=dev)
            fn_direct = self.ctx.lib.zsw_align_3pass_batch if three_pass else self.ctx.lib.zsw_align_batch
            fn_from = self.ctx.lib.zsw_align_3pass_batch_from if three_pass else self.ctx.lib.zsw_align_batch_from
            if direct is not None:
                rc = fn_direct(self.ctx.h, C.byref(b), direct[0], direct[1], int(seq.is_query), aln.data_ptr(),
                                                  status.data_ptr(), inc.data_ptr(), op.data_ptr(), cap, C.byref(total), self.ctx.stream())
            else:
                rc = fn_from(self.ctx.h, C.byref(b), from_width, preset, int(seq.is_query), aln.data_ptr(),
                                                       status.data_ptr(), tier.data_ptr(), inc.data_ptr(), op.data_ptr(), cap,
                                                       C.byref(total), self.ctx.stream())
            if rc == -1 and total.value > cap:
                cap = int(total.value)
                continue
            self.ctx.check(rc)
            break
        torch.cuda.synchronize(dev)
        t = int(total.value)
        h_aln, h_status, h_inc, h_op, h_tier = _to_host(aln[: n * ALN_DTYPE.itemsize], status[:n], inc[:t], op[:t], tier[:n])
        return AlignmentBatch(h_status, h_aln.view(ALN_DTYPE), h_inc.view(np.uint32), h_op, h_tier if direct is None else None)


class StripedProfileBatch(_ProfileBatchBase):
    """`StripedProfile::<T, N, S>::new(read_i, &matrix, gap_open, gap_extend)` for every read of a batch.

    For unsigned T pass matrix.to_biased_matrix() exactly as the reference requires (profile.rs:215-219)."""

    def __init__(self, reads, matrix: WeightMatrix, gap_open: int, gap_extend: int, T: str = "i16", N: int = 16, device: int = 0):
        if T not in _lib.INT_TYPES:
            raise ValueError(f"T must be one of {list(_lib.INT_TYPES)}")
        if T.startswith("u") and matrix.signed:
            raise ValueError("unsigned T needs matrix.to_biased_matrix() (profile.rs:215-219)")
        if T.startswith("i") and not matrix.signed:
            raise ValueError("signed T needs the signed matrix")
        super().__init__(reads, matrix, gap_open, gap_extend, device)
        self.T, self.N = T, N

    new = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def sw_score(self, reference) -> ScoreBatch:
        """profile.rs:440-446 → sw_simd_score (striped.rs:65-142)"""
        torch = _torch()
        self._prep(reference)
        n = self.reads.n_reads
        dev = self.reads.bases.device
        score = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = self.reads.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_batch(self.ctx.h, C.byref(b), _lib.INT_TYPES[self.T], self.N, score.data_ptr(),
                                                    status.data_ptr(), self.ctx.stream()))
        return ScoreBatch(score[:n], status[:n])

    def sw_score_ends(self, seq: SeqSrc) -> ScoreBatch:
        """profile.rs:456-460 → sw_simd_score_ends (striped.rs:153-162). SeqSrc::Query swaps the two ends."""
        torch = _torch()
        self._prep(seq.seq)
        n = self.reads.n_reads
        dev = self.reads.bases.device
        score = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        rend = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        qend = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = self.reads.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_ends_batch(self.ctx.h, C.byref(b), _lib.INT_TYPES[self.T], self.N, score.data_ptr(),
                                                         rend.data_ptr(), qend.data_ptr(), status.data_ptr(), self.ctx.stream()))
        if seq.is_query:
            rend, qend = qend, rend
        return ScoreBatch(score[:n], status[:n], ref_end=rend[:n], query_end=qend[:n])

    def sw_score_ranges(self, seq: SeqSrc) -> ScoreBatch:
        """profile.rs:529-533 → sw_simd_score_ranges (striped.rs:355-388): ScoreAndRanges per read (0-based half-open);
        SeqSrc::Query swaps the reference and query ranges."""
        torch = _torch()
        self._prep(seq.seq)
        n = self.reads.n_reads
        dev = self.reads.bases.device
        o = [torch.empty(max(n, 1), dtype=torch.int32, device=dev) for _ in range(5)]
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = self.reads.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_ranges_batch(self.ctx.h, C.byref(b), _lib.INT_TYPES[self.T], self.N, o[0].data_ptr(),
                                                           o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(),
                                                           status.data_ptr(), self.ctx.stream()))
        rs, re_, qs, qe = (x[:n] for x in o[1:])
        if seq.is_query:
            rs, re_, qs, qe = qs, qe, rs, re_
        return ScoreBatch(o[0][:n], status[:n], ref_start=rs, ref_end=re_, query_start=qs, query_end=qe)

    def sw_align(self, seq: SeqSrc) -> AlignmentBatch:
        """profile.rs:515-519 → sw_simd_align (striped.rs:449-598)"""
        return self._align(seq, (_lib.INT_TYPES[self.T], self.N))

    def sw_align_3pass(self, seq: SeqSrc) -> AlignmentBatch:
        """profile.rs:546-552 → sw_align_3pass (three_pass.rs:21-104)"""
        return self._align(seq, (_lib.INT_TYPES[self.T], self.N), three_pass=True)


class LocalProfilesBatch(_ProfileBatchBase):
    """`LocalProfiles::new_with_w{128,256,512}(read_i, &matrix, gap_open, gap_extend)` for every read."""

    def __init__(self, reads, matrix: WeightMatrix, gap_open: int, gap_extend: int, preset: int = 256, device: int = 0):
        if not matrix.signed:
            raise ValueError("profile sets take the signed matrix (profile_set.rs:382-390)")
        if preset not in (128, 256, 512):
            raise ValueError("preset must be 128, 256 or 512")
        super().__init__(reads, matrix, gap_open, gap_extend, device)
        self.preset = preset

    @classmethod
    def new_with_w128(cls, reads, matrix, gap_open, gap_extend, device=0):
        return cls(reads, matrix, gap_open, gap_extend, 128, device)

    @classmethod
    def new_with_w256(cls, reads, matrix, gap_open, gap_extend, device=0):
        return cls(reads, matrix, gap_open, gap_extend, 256, device)

    @classmethod
    def new_with_w512(cls, reads, matrix, gap_open, gap_extend, device=0):
        return cls(reads, matrix, gap_open, gap_extend, 512, device)

    def _score_from(self, reference, width: int, out=None) -> ScoreBatch:
        """`out`: an object with preallocated `score` (int32) and `status` (uint8) device tensors of at least n reads, e.g. a
        zoe_amd.dist.ResultSlab, so that the kernel writes straight into the buffer a collective sends."""
        torch = _torch()
        self._prep(reference)
        n = self.reads.n_reads
        dev = self.reads.bases.device
        if out is not None:
            score, status = out.score, out.status
            if score.numel() < n or status.numel() < n or score.device != dev:
                raise ValueError("out.score / out.status too small or on another device")
        else:
            score = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        tier = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = self.reads.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_batch_from(self.ctx.h, C.byref(b), width, self.preset, score.data_ptr(),
                                                         status.data_ptr(), tier.data_ptr(), self.ctx.stream()))
        return ScoreBatch(score[:n], status[:n], tier=tier[:n])

    def sw_score_from_i8(self, reference, out=None) -> ScoreBatch:
        return self._score_from(reference, 8, out)

    def sw_score_from_i16(self, reference) -> ScoreBatch:
        return self._score_from(reference, 16)

    def sw_score_from_i32(self, reference) -> ScoreBatch:
        return self._score_from(reference, 32)

    def _ranges_from(self, seq: SeqSrc, width: int) -> ScoreBatch:
        """profile_set.rs:313-362: sw_score_ranges over the tiers i{width} -> i32 of this preset"""
        torch = _torch()
        self._prep(seq.seq)
        n = self.reads.n_reads
        dev = self.reads.bases.device
        o = [torch.empty(max(n, 1), dtype=torch.int32, device=dev) for _ in range(5)]
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        tier = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = self.reads.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_ranges_batch_from(self.ctx.h, C.byref(b), width, self.preset, o[0].data_ptr(), o[1].data_ptr(),
                                                                o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), status.data_ptr(),
                                                                tier.data_ptr(), self.ctx.stream()))
        rs, re_, qs, qe = (x[:n] for x in o[1:])
        if seq.is_query:
            rs, re_, qs, qe = qs, qe, rs, re_
        return ScoreBatch(o[0][:n], status[:n], tier=tier[:n], ref_start=rs, ref_end=re_, query_start=qs, query_end=qe)

    def sw_score_ranges_from_i8(self, seq: SeqSrc) -> ScoreBatch:
        return self._ranges_from(seq, 8)

    def sw_score_ranges_from_i16(self, seq: SeqSrc) -> ScoreBatch:
        return self._ranges_from(seq, 16)

    def sw_score_ranges_from_i32(self, seq: SeqSrc) -> ScoreBatch:
        return self._ranges_from(seq, 32)

    def sw_align_from_i8(self, seq: SeqSrc) -> AlignmentBatch:
        return self._align(seq, None, 8, self.preset)

    def sw_align_from_i16(self, seq: SeqSrc) -> AlignmentBatch:
        return self._align(seq, None, 16, self.preset)

    def sw_align_from_i8_3pass(self, seq: SeqSrc) -> AlignmentBatch:
        """profile_set.rs:212-229 → sw_align_3pass (three_pass.rs:21-104)"""
        return self._align(seq, None, 8, self.preset, three_pass=True)

    def sw_align_from_i16_3pass(self, seq: SeqSrc) -> AlignmentBatch:
        return self._align(seq, None, 16, self.preset, three_pass=True)

    def sw_align_from_i32_3pass(self, seq: SeqSrc) -> AlignmentBatch:
        return self._align(seq, None, 32, self.preset, three_pass=True)

    def sw_align_from_i32(self, seq: SeqSrc) -> AlignmentBatch:
        return self._align(seq, None, 32, self.preset)


FILTER_REJECT, FILTER_PASS, FILTER_NONE = 0, 1, 2


def sneaky_snake(reference, reads, ref_start, ref_len, threshold: float, device: int = 0):
    """Batched `alignment::sneaky_snake(&reference[ref_start[i]..][..ref_len[i]], read_i, threshold)`
    (sneaky_snake.rs:78-131): a uint8 device tensor of FILTER_REJECT = Some(false) / FILTER_PASS = Some(true) /
    FILTER_NONE = None per read. `ref_start` / `ref_len`: one candidate window of `reference` per read."""
    torch = _torch()
    rb = _as_batch(reads, device)
    ctx = SwContext.get(rb.device_index)
    ctx.set_reference(reference)
    dev = rb.bases.device
    n = rb.n_reads
    rs = torch.as_tensor(np.asarray(ref_start, dtype=np.int64) if not isinstance(ref_start, torch.Tensor) else ref_start).to(dev, torch.int32).contiguous()
    rl = torch.as_tensor(np.asarray(ref_len, dtype=np.int64) if not isinstance(ref_len, torch.Tensor) else ref_len).to(dev, torch.int32).contiguous()
    if rs.numel() != n or rl.numel() != n:
        raise ValueError("one (ref_start, ref_len) window per read")
    out = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    b = rb.c_batch()
    ctx.check(ctx.lib.zsw_sneaky_snake_batch(ctx.h, C.byref(b), rs.data_ptr(), rl.data_ptr(), C.c_float(threshold), out.data_ptr(),
                                             ctx.stream()), profile_errors=False)
    return out[:n]


def into_local_profile(reads, matrix: WeightMatrix, gap_open: int, gap_extend: int, device: int = 0) -> LocalProfilesBatch:
    """Nucleotides::into_local_profile (nucleotides/mod.rs:262-266): the w256 preset."""
    return LocalProfilesBatch.new_with_w256(reads, matrix, gap_open, gap_extend, device)


class _SharedBase:
    """The one-profile-many-sequences role: the profile is built ONCE from `sequence` (sw/mod.rs:63-67 "the profile can be
    aligned against any number of different sequences") and every read of a batch is the sequence it is aligned against,
    walked row by row. Results follow the reference's conventions: `ref_*` is the non-profile sequence (the read) unless the
    reads are passed as `SeqSrc.Query(reads)`, which swaps the roles back (alignment/mod.rs:176-190)."""

    def __init__(self, sequence, matrix: WeightMatrix, gap_open: int, gap_extend: int, device: int = 0):
        if not isinstance(sequence, (bytes, bytearray, memoryview, np.ndarray)):
            # (this class once took the READS here; bytes(ReadBatch) or bytes(list) would be accepted and mean nothing)
            raise TypeError("the shared profile is built from ONE sequence (bytes); the reads are the arguments of its methods")
        sequence = bytes(sequence)
        validate_profile_args(len(sequence), gap_open, gap_extend)
        self.sequence = sequence
        self.matrix = matrix
        self.gap_open = gap_open
        self.gap_extend = gap_extend
        self.device = device
        self.ctx = SwContext.get(device)

    def _prep(self):
        self.ctx.set_scoring(self.matrix, self.gap_open, self.gap_extend)
        self.ctx.set_profile_sequence(self.sequence)

    @staticmethod
    def _reads_of(seq):
        """reads, or SeqSrc.Reference(reads) / SeqSrc.Query(reads) with a ReadBatch or a list of byte strings inside"""
        if isinstance(seq, SeqBatchSrc):
            return seq.reads, seq.is_query
        return seq, False

    def _ends(self, seq, direct) -> ScoreBatch:
        torch = _torch()
        reads, is_query = self._reads_of(seq)
        rb = _as_batch(reads, self.device)
        self._prep()
        n, dev = rb.n_reads, rb.bases.device
        score = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        rend = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        qend = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = rb.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_ends_shared_batch(self.ctx.h, C.byref(b), direct[0], direct[1], score.data_ptr(), rend.data_ptr(),
                                                                qend.data_ptr(), status.data_ptr(), self.ctx.stream()))
        if is_query:
            rend, qend = qend, rend
        return ScoreBatch(score[:n], status[:n], ref_end=rend[:n], query_end=qend[:n])

    def _ranges(self, seq, direct=None, from_width=None, preset=None) -> ScoreBatch:
        torch = _torch()
        reads, is_query = self._reads_of(seq)
        rb = _as_batch(reads, self.device)
        self._prep()
        n, dev = rb.n_reads, rb.bases.device
        o = [torch.empty(max(n, 1), dtype=torch.int32, device=dev) for _ in range(5)]
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        tier = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = rb.c_batch()
        if direct is not None:
            self.ctx.check(self.ctx.lib.zsw_score_ranges_shared_batch(self.ctx.h, C.byref(b), direct[0], direct[1], o[0].data_ptr(), o[1].data_ptr(),
                                                                      o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), status.data_ptr(),
                                                                      self.ctx.stream()))
        else:
            self.ctx.check(self.ctx.lib.zsw_score_ranges_shared_batch_from(self.ctx.h, C.byref(b), from_width, preset, o[0].data_ptr(), o[1].data_ptr(),
                                                                           o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), status.data_ptr(),
                                                                           tier.data_ptr(), self.ctx.stream()))
        rs, re_, qs, qe = (x[:n] for x in o[1:])
        if is_query:
            rs, re_, qs, qe = qs, qe, rs, re_
        return ScoreBatch(o[0][:n], status[:n], tier=tier[:n] if direct is None else None, ref_start=rs, ref_end=re_, query_start=qs, query_end=qe)

    def _align(self, seq, direct=None, from_width=None, preset=None, threepass=False) -> AlignmentBatch:
        torch = _torch()
        lib = self.ctx.lib
        f_direct = lib.zsw_align_3pass_shared_batch if threepass else lib.zsw_align_shared_batch
        f_from = lib.zsw_align_3pass_shared_batch_from if threepass else lib.zsw_align_shared_batch_from
        reads, is_query = self._reads_of(seq)
        rb = _as_batch(reads, self.device)
        self._prep()
        n, dev = rb.n_reads, rb.bases.device
        aln = torch.zeros(max(n, 1) * ALN_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        status = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        tier = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        cap = max(8 * n, 64)
        b = rb.c_batch()
        total = C.c_uint64(0)
        while True:
            inc = torch.empty(cap, dtype=torch.int32, device=dev)
            op = torch.empty(cap, dtype=torch.uint8, device=dev)
            if direct is not None:
                rc = f_direct(self.ctx.h, C.byref(b), direct[0], direct[1], int(is_query), aln.data_ptr(), status.data_ptr(),
                              inc.data_ptr(), op.data_ptr(), cap, C.byref(total), self.ctx.stream())
            else:
                rc = f_from(self.ctx.h, C.byref(b), from_width, preset, int(is_query), aln.data_ptr(), status.data_ptr(),
                            tier.data_ptr(), inc.data_ptr(), op.data_ptr(), cap, C.byref(total), self.ctx.stream())
            if rc == -1 and total.value > cap:
                cap = int(total.value)
                continue
            self.ctx.check(rc)
            break
        torch.cuda.synchronize(dev)
        t = int(total.value)
        h_aln, h_status, h_inc, h_op, h_tier = _to_host(aln[: n * ALN_DTYPE.itemsize], status[:n], inc[:t], op[:t], tier[:n])
        return AlignmentBatch(h_status, h_aln.view(ALN_DTYPE), h_inc.view(np.uint32), h_op, h_tier if direct is None else None)


@dataclass
class SeqBatchSrc:
    """SeqSrc (alignment/mod.rs:161-166) over a batch of reads: which role the non-profile sequences play in the results."""

    reads: "object"
    is_query: bool

    @staticmethod
    def Reference(reads) -> "SeqBatchSrc":
        return SeqBatchSrc(reads, False)

    @staticmethod
    def Query(reads) -> "SeqBatchSrc":
        return SeqBatchSrc(reads, True)


class SharedStripedProfile(_SharedBase):
    """`StripedProfile::<T, N, S>::new(sequence, &matrix, gap_open, gap_extend)` once, used against every read of a batch
    (profile.rs:239-306, 440-552): sw_score(reads), sw_score_ends / sw_score_ranges / sw_align(SeqBatchSrc)."""

    def __init__(self, sequence, matrix: WeightMatrix, gap_open: int, gap_extend: int, T: str = "i16", N: int = 16, device: int = 0):
        if T not in _lib.INT_TYPES:
            raise ValueError(f"T must be one of {list(_lib.INT_TYPES)}")
        if T.startswith("u") and matrix.signed:
            raise ValueError("unsigned T needs matrix.to_biased_matrix() (profile.rs:215-219)")
        if T.startswith("i") and not matrix.signed:
            raise ValueError("signed T needs the signed matrix")
        super().__init__(sequence, matrix, gap_open, gap_extend, device)
        self.T, self.N = T, N

    def sw_score(self, reads) -> ScoreBatch:
        torch = _torch()
        rb = _as_batch(reads, self.device)
        self._prep()
        n, dev = rb.n_reads, rb.bases.device
        score = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = rb.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_shared_batch(self.ctx.h, C.byref(b), _lib.INT_TYPES[self.T], self.N, score.data_ptr(), status.data_ptr(),
                                                           self.ctx.stream()))
        return ScoreBatch(score[:n], status[:n])

    def sw_score_ends(self, seq) -> ScoreBatch:
        return self._ends(seq, (_lib.INT_TYPES[self.T], self.N))

    def sw_score_ranges(self, seq) -> ScoreBatch:
        return self._ranges(seq, direct=(_lib.INT_TYPES[self.T], self.N))

    def sw_align(self, seq) -> AlignmentBatch:
        return self._align(seq, direct=(_lib.INT_TYPES[self.T], self.N))

    def sw_align_3pass(self, seq) -> AlignmentBatch:
        """profile.rs:536-552 (three_pass.rs:21-104) with the shared profile: ranges, then banded / scalar alignment in the box"""
        return self._align(seq, direct=(_lib.INT_TYPES[self.T], self.N), threepass=True)


class SharedProfilesBatch(_SharedBase):
    """`SharedProfiles::new_with_w{128,256,512}(sequence, &matrix, gap_open, gap_extend)` (profile_set.rs:552-700) used against
    every read of a batch: the i8 -> i16 -> i32 cascade of `sw_score_from_i*`, `sw_score_ranges_from_i*`, `sw_align_from_i*`."""

    def __init__(self, sequence, matrix: WeightMatrix, gap_open: int, gap_extend: int, preset: int = 256, device: int = 0):
        if not matrix.signed:
            raise ValueError("profile sets take the signed matrix (profile_set.rs:552-560)")
        if preset not in (128, 256, 512):
            raise ValueError("preset must be 128, 256 or 512")
        super().__init__(sequence, matrix, gap_open, gap_extend, device)
        self.preset = preset

    @classmethod
    def new_with_w128(cls, sequence, matrix, gap_open, gap_extend, device=0):
        return cls(sequence, matrix, gap_open, gap_extend, 128, device)

    @classmethod
    def new_with_w256(cls, sequence, matrix, gap_open, gap_extend, device=0):
        return cls(sequence, matrix, gap_open, gap_extend, 256, device)

    @classmethod
    def new_with_w512(cls, sequence, matrix, gap_open, gap_extend, device=0):
        return cls(sequence, matrix, gap_open, gap_extend, 512, device)

    def _score_from(self, reads, width: int) -> ScoreBatch:
        torch = _torch()
        rb = _as_batch(reads, self.device)
        self._prep()
        n, dev = rb.n_reads, rb.bases.device
        score = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        tier = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        b = rb.c_batch()
        self.ctx.check(self.ctx.lib.zsw_score_shared_batch_from(self.ctx.h, C.byref(b), width, self.preset, score.data_ptr(), status.data_ptr(),
                                                                tier.data_ptr(), self.ctx.stream()))
        return ScoreBatch(score[:n], status[:n], tier=tier[:n])

    def sw_score_from_i8(self, reads) -> ScoreBatch:
        return self._score_from(reads, 8)

    def sw_score_from_i16(self, reads) -> ScoreBatch:
        return self._score_from(reads, 16)

    def sw_score_from_i32(self, reads) -> ScoreBatch:
        return self._score_from(reads, 32)

    def sw_score_ranges_from_i8(self, seq) -> ScoreBatch:
        return self._ranges(seq, from_width=8, preset=self.preset)

    def sw_score_ranges_from_i16(self, seq) -> ScoreBatch:
        return self._ranges(seq, from_width=16, preset=self.preset)

    def sw_score_ranges_from_i32(self, seq) -> ScoreBatch:
        return self._ranges(seq, from_width=32, preset=self.preset)

    def sw_align_from_i8(self, seq) -> AlignmentBatch:
        return self._align(seq, from_width=8, preset=self.preset)

    def sw_align_from_i16(self, seq) -> AlignmentBatch:
        return self._align(seq, from_width=16, preset=self.preset)

    def sw_align_from_i32(self, seq) -> AlignmentBatch:
        return self._align(seq, from_width=32, preset=self.preset)

    def sw_align_from_i8_3pass(self, seq) -> AlignmentBatch:
        """ProfileSets::sw_align_from_i8_3pass (profile_set.rs:212-283) of SharedProfiles (:552-560)"""
        return self._align(seq, from_width=8, preset=self.preset, threepass=True)

    def sw_align_from_i16_3pass(self, seq) -> AlignmentBatch:
        return self._align(seq, from_width=16, preset=self.preset, threepass=True)

    def sw_align_from_i32_3pass(self, seq) -> AlignmentBatch:
        return self._align(seq, from_width=32, preset=self.preset, threepass=True)


def into_shared_profile(sequence, matrix: WeightMatrix, gap_open: int, gap_extend: int, device: int = 0) -> SharedProfilesBatch:
    """Nucleotides::into_shared_profile (nucleotides/mod.rs:295-299): SharedProfiles<32, 16, 8, S> of `sequence` (the w256 preset)"""
    return SharedProfilesBatch(sequence, matrix, gap_open, gap_extend, preset=256, device=device)
