"""ctypes front-end of the parity oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (zoe_amd/) never does.  The oracle is a CPU
restatement of CDCgov/zoe's striped Smith-Waterman path (see zoe_oracle.hpp for
the file:line map); it is pinned by the reference's own known-answer tests
(tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libzoe_oracle.so")

SOME, OVERFLOWED, UNMAPPED = 0, 1, 2
STATUS_NAMES = {0: "Some", 1: "Overflowed", 2: "Unmapped"}
TCODE = {"i8": 0, "i16": 1, "i32": 2, "u8": 3, "u16": 4, "u32": 5}
PROFILE_ERRORS = {
    1: "EmptySequence",
    2: "GapOpenOutOfRange",
    3: "GapExtendOutOfRange",
    4: "BadGapWeights",
}


class ProfileError(ValueError):
    def __init__(self, code: int):
        super().__init__(PROFILE_ERRORS.get(code, f"error {code}"))
        self.code = code


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (g++)."""
    srcs = [os.path.join(_DIR, f) for f in ("zoe_oracle.hpp", "zoe_oracle_capi.cpp", "zoe_cpu_fast.cpp", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if stale:
        subprocess.run(["make", "-C", _DIR, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.zor_profile_dump.restype = C.c_long
        _lib.zor_score_from_path.restype = C.c_longlong
    return _lib


def _u8(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(0, dtype=np.uint8)


def _p(a: np.ndarray, t=C.c_uint8):
    return a.ctypes.data_as(C.POINTER(t))


@dataclass(frozen=True)
class Scoring:
    """WeightMatrix<i8,S> + ByteIndexMap<S> + gap penalties (as passed to StripedProfile::new)."""

    weights: np.ndarray  # int8 [S,S], row = reference residue, col = query residue
    index_map: np.ndarray  # uint8 [256]
    gap_open: int
    gap_extend: int

    @property
    def S(self) -> int:
        return int(self.weights.shape[0])


def dna_profile_map() -> np.ndarray:
    out = np.zeros(256, dtype=np.uint8)
    lib().zor_dna_profile_map(_p(out))
    return out


def byte_index_map(keys: bytes, catch_all: bytes, ignore_case: bool = False) -> np.ndarray:
    out = np.zeros(256, dtype=np.uint8)
    k = _u8(keys)
    lib().zor_byte_index_map(_p(k), len(keys), C.c_uint8(catch_all[0]), int(ignore_case), _p(out))
    return out


def weight_matrix_new(index_map: np.ndarray, S: int, matching: int, mismatch: int, ignoring: Optional[bytes]) -> np.ndarray:
    out = np.zeros((S, S), dtype=np.int8)
    lib().zor_weight_matrix_new(_p(index_map), S, matching, mismatch, -1 if ignoring is None else ignoring[0], _p(out, C.c_int8))
    return out


def dna_scoring(matching=2, mismatch=-5, ignoring: Optional[bytes] = b"N", gap_open=-10, gap_extend=-1) -> Scoring:
    """WeightMatrix::new_dna_matrix + the reference's test gaps (sw/mod.rs:465-471)."""
    m = dna_profile_map()
    return Scoring(weight_matrix_new(m, 5, matching, mismatch, ignoring), m, gap_open, gap_extend)


def to_biased_matrix(weights: np.ndarray):
    S = weights.shape[0]
    out = np.zeros((S, S), dtype=np.uint8)
    w = np.ascontiguousarray(weights, dtype=np.int8)
    bias = lib().zor_to_biased_matrix(_p(w, C.c_int8), S, _p(out))
    return out, bias


@dataclass
class Aln:
    status: int
    score: int = 0
    ref_range: tuple = (0, 0)
    query_range: tuple = (0, 0)
    cigar: str = ""
    ref_len: int = 0
    query_len: int = 0
    n_ciglets: int = 0

    def key(self):
        return (self.status, self.score, self.ref_range, self.query_range, self.cigar, self.ref_len, self.query_len)


def _sc_args(sc: Scoring):
    w = np.ascontiguousarray(sc.weights, dtype=np.int8)
    im = np.ascontiguousarray(sc.index_map, dtype=np.uint8)
    return w, im, (sc.S, _p(w, C.c_int8), _p(im), int(sc.gap_open), int(sc.gap_extend))


def _check(rc: int):
    if rc > 0:
        raise ProfileError(rc)
    if rc < 0:
        raise RuntimeError(f"oracle error {rc}")


def score(T: str, lanes: int, sc: Scoring, prof_seq, other):
    """StripedProfile::<T,lanes,S>::new(prof_seq, ..).sw_score(other) → (status, score)."""
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st, s = C.c_uint32(0), C.c_uint32(0)
    _check(lib().zor_score(TCODE[T], lanes, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), C.byref(st), C.byref(s)))
    return st.value, s.value


def score_ends(T: str, lanes: int, sc: Scoring, prof_seq, other, forward=True):
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st = C.c_uint32(0)
    out = (C.c_uint64 * 3)()
    _check(lib().zor_score_ends(TCODE[T], lanes, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), int(forward), C.byref(st), out))
    return st.value, tuple(int(x) for x in out)


def score_ranges(T: str, lanes: int, sc: Scoring, prof_seq, other):
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st = C.c_uint32(0)
    out = (C.c_uint64 * 5)()
    _check(lib().zor_score_ranges(TCODE[T], lanes, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), C.byref(st), out))
    o = [int(x) for x in out]
    return st.value, o[0], (o[1], o[2]), (o[3], o[4])


def cascade_score_ranges(from_width: int, preset: int, sc: Scoring, prof_seq, other):
    """LocalProfiles::new_with_w{preset}(prof_seq).sw_score_ranges_from_i{from_width}(SeqSrc::Reference(other))
    → (status, score, ref_range, query_range, tier)."""
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st, tier = C.c_uint32(0), C.c_int(0)
    out = (C.c_uint64 * 5)()
    _check(lib().zor_cascade_score_ranges(from_width, preset, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), C.byref(st), out, C.byref(tier)))
    o = [int(x) for x in out]
    return st.value, o[0], (o[1], o[2]), (o[3], o[4]), tier.value


def _aln(st, f, buf) -> Aln:
    if st != SOME:
        return Aln(status=st)
    f = [int(x) for x in f]
    return Aln(st, f[0], (f[1], f[2]), (f[3], f[4]), buf.value.decode(), f[5], f[6], f[7])


def align(T: str, lanes: int, sc: Scoring, prof_seq, other, other_is_query=False, want_flags=False):
    """profile.sw_align(SeqSrc::Reference(other)) (or SeqSrc::Query → inverted)."""
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st = C.c_uint32(0)
    f = (C.c_uint64 * 8)()
    cap = 16 * (len(ps) + len(ot)) + 64
    buf = C.create_string_buffer(cap)
    flags = None
    nv = (len(ps) + lanes - 1) // lanes
    if want_flags:
        flags = np.zeros(max(1, len(ot) * nv * lanes), dtype=np.uint8)
    _check(
        lib().zor_align(
            TCODE[T], lanes, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), int(other_is_query),
            C.byref(st), f, buf, C.c_size_t(cap), _p(flags) if want_flags else None, C.c_size_t(flags.size if want_flags else 0),
        )
    )
    r = _aln(st.value, f, buf)
    if want_flags:
        return r, flags.reshape(len(ot), nv, lanes) if len(ot) else flags
    return r


def scalar_score(sc: Scoring, prof_seq, other):
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st, s = C.c_uint32(0), C.c_uint32(0)
    _check(lib().zor_scalar_score(*a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), C.byref(st), C.byref(s)))
    return st.value, s.value


def scalar_align(sc: Scoring, prof_seq, other, other_is_query=False) -> Aln:
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st = C.c_uint32(0)
    f = (C.c_uint64 * 8)()
    cap = 16 * (len(ps) + len(ot)) + 64
    buf = C.create_string_buffer(cap)
    _check(lib().zor_scalar_align(*a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), int(other_is_query), C.byref(st), f, buf, C.c_size_t(cap)))
    return _aln(st.value, f, buf)


def cascade_score(from_width: int, preset: int, sc: Scoring, prof_seq, other):
    """LocalProfiles::new_with_w{preset}(prof_seq).sw_score_from_i{from_width}(other) → (status, score, tier)."""
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st, s, tier = C.c_uint32(0), C.c_uint32(0), C.c_int(0)
    _check(lib().zor_cascade_score(from_width, preset, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), C.byref(st), C.byref(s), C.byref(tier)))
    return st.value, s.value, tier.value


def cascade_align(from_width: int, preset: int, sc: Scoring, prof_seq, other, other_is_query=False):
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st, tier = C.c_uint32(0), C.c_int(0)
    f = (C.c_uint64 * 8)()
    cap = 16 * (len(ps) + len(ot)) + 64
    buf = C.create_string_buffer(cap)
    _check(lib().zor_cascade_align(from_width, preset, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), int(other_is_query), C.byref(st), f, buf, C.c_size_t(cap), C.byref(tier)))
    return _aln(st.value, f, buf), tier.value


def banded_align(sc: Scoring, prof_seq, other, band_width: int) -> Aln:
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st = C.c_uint32(0)
    f = (C.c_uint64 * 8)()
    cap = 16 * (len(ps) + len(ot)) + 64
    buf = C.create_string_buffer(cap)
    _check(lib().zor_banded_align(*a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), C.c_size_t(band_width), C.byref(st), f, buf, C.c_size_t(cap)))
    return _aln(st.value, f, buf)


def align_3pass(T: str, lanes: int, sc: Scoring, prof_seq, other, other_is_query=False):
    """profile.sw_align_3pass(SeqSrc::Reference(other), ..) → (Aln, how) with how 0 = no-gaps shortcut, 1 = banded, 2 = scalar."""
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st, how = C.c_uint32(0), C.c_int(-1)
    f = (C.c_uint64 * 8)()
    cap = 16 * (len(ps) + len(ot)) + 64
    buf = C.create_string_buffer(cap)
    _check(lib().zor_align_3pass(TCODE[T], lanes, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), int(other_is_query), C.byref(st), f, buf, C.c_size_t(cap), C.byref(how)))
    return _aln(st.value, f, buf), how.value


def cascade_align_3pass(from_width: int, preset: int, sc: Scoring, prof_seq, other, other_is_query=False):
    """LocalProfiles::new_with_w{preset}(prof_seq).sw_align_from_i{from_width}_3pass(..) → (Aln, tier, how)."""
    w, im, a = _sc_args(sc)
    ps, ot = _u8(prof_seq), _u8(other)
    st, tier, how = C.c_uint32(0), C.c_int(0), C.c_int(-1)
    f = (C.c_uint64 * 8)()
    cap = 16 * (len(ps) + len(ot)) + 64
    buf = C.create_string_buffer(cap)
    _check(lib().zor_cascade_align_3pass(from_width, preset, *a, _p(ps), C.c_size_t(len(ps)), _p(ot), C.c_size_t(len(ot)), int(other_is_query), C.byref(st), f, buf, C.c_size_t(cap), C.byref(tier), C.byref(how)))
    return _aln(st.value, f, buf), tier.value, how.value


def sneaky_snake(reference, query, threshold: float):
    """alignment::sneaky_snake(reference, query, threshold) → True / False / None (sneaky_snake.rs:78-131)."""
    r, q = _u8(reference), _u8(query)
    rc = lib().zor_sneaky_snake(_p(r), C.c_size_t(len(r)), _p(q), C.c_size_t(len(q)), C.c_float(threshold))
    return {0: False, 1: True, 2: None}[rc]


def profile_dump(T: str, lanes: int, sc: Scoring, seq, rev_end: int = 0) -> np.ndarray:
    """StripedProfile::new(seq) (or .reverse_from_forward(rev_end)) as int64 [S, nv, lanes]."""
    w, im, a = _sc_args(sc)
    s = _u8(seq)
    cap = sc.S * ((len(s) + lanes - 1) // lanes + 1) * lanes
    out = np.zeros(cap, dtype=np.int64)
    nv = lib().zor_profile_dump(TCODE[T], lanes, *a, _p(s), C.c_size_t(len(s)), C.c_size_t(rev_end), _p(out, C.c_int64), C.c_size_t(cap))
    if nv < 0:
        if -nv in PROFILE_ERRORS:
            raise ProfileError(-nv)
        raise RuntimeError(f"oracle error {nv}")
    return out[: sc.S * nv * lanes].reshape(sc.S, nv, lanes)


def score_from_path(sc: Scoring, query, ref_in_alignment, cigar: str) -> int:
    w, im, a = _sc_args(sc)
    q, r = _u8(query), _u8(ref_in_alignment)
    return int(lib().zor_score_from_path(*a, _p(q), C.c_size_t(len(q)), _p(r), C.c_size_t(len(r)), cigar.encode()))


def validate_profile_args(seq_len: int, gap_open: int, gap_extend: int) -> int:
    return int(lib().zor_validate_profile_args(C.c_size_t(seq_len), gap_open, gap_extend))


def batch_score_w256(from_width: int, sc: Scoring, reads: np.ndarray, reference, offsets: Optional[np.ndarray] = None,
                     fixed_len: int = 0, threads: int = 1):
    """Batched `read.into_local_profile(..).sw_score_from_i{8,16}(reference)` (AVX2 restatement, w256 preset).

    Returns (score u32[n], status u8[n], tier u8[n])."""
    w, im, a = _sc_args(sc)
    reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
    ref = _u8(reference)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        offp = _p(offsets, C.c_uint64)
    else:
        n = reads.size // fixed_len
        offp = None
    score = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint8)
    tier = np.zeros(n, dtype=np.uint8)
    rc = lib().zor_batch_score_w256(from_width, *a, _p(reads), offp, C.c_size_t(fixed_len), C.c_size_t(n), _p(ref), C.c_size_t(len(ref)), threads, _p(score, C.c_uint32), _p(status), _p(tier))
    _check(rc)
    return score, status, tier


def batch_score_shared_w256(from_width: int, sc: Scoring, reads: np.ndarray, profile_seq, offsets: Optional[np.ndarray] = None,
                            fixed_len: int = 0, threads: int = 1):
    """SharedProfiles usage (sw/mod.rs:73-78): one profile built from `profile_seq` (the long sequence), every read
    passed as `reference`. Returns (score u32[n], status u8[n])."""
    w, im, a = _sc_args(sc)
    reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
    ps = _u8(profile_seq)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        offp = _p(offsets, C.c_uint64)
    else:
        n = reads.size // fixed_len
        offp = None
    score = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint8)
    rc = lib().zor_batch_score_shared_w256(from_width, *a, _p(reads), offp, C.c_size_t(fixed_len), C.c_size_t(n), _p(ps), C.c_size_t(len(ps)), threads, _p(score, C.c_uint32), _p(status))
    _check(rc)
    return score, status


def hardware_threads() -> int:
    return int(lib().zor_hardware_threads())
