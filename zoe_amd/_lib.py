"""ctypes binding of libzoe_sw_hip.so (the C ABI declared in include/zoe_sw.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzoe_sw_hip.so")

ZSW_OK = 0
ERROR_NAMES = {
    0: "ZSW_OK",
    1: "ProfileError::EmptySequence",
    2: "ProfileError::GapOpenOutOfRange",
    3: "ProfileError::GapExtendOutOfRange",
    4: "ProfileError::BadGapWeights",
    -1: "ZSW_ERR_INVALID_ARGUMENT",
    -2: "ZSW_ERR_HIP",
    -3: "ZSW_ERR_NO_DEVICE",
    -4: "ZSW_ERR_UNSUPPORTED",
    -5: "ZSW_ERR_NOT_CONFIGURED",
}
MEM_HOST, MEM_DEVICE = 0, 1
# zsw_debug_flag (kernel-selection overrides for the parity tests; results never depend on them)
DEBUG_SCORE_V1, DEBUG_NO_TILES, DEBUG_NO_W32, DEBUG_NO_WIDE, DEBUG_NO_SIDE_STREAMS, DEBUG_NO_PIPELINE, DEBUG_ALIGN_NO_PACKED = 1, 2, 4, 8, 16, 32, 64
DEBUG_SCORE_PRUNE, DEBUG_SCORE_PRUNE_ANY_SIZE, DEBUG_PRUNE_STRIP, DEBUG_ALIGN_LONG_WARMUP, DEBUG_SEED_NO_BAND, DEBUG_SEED_WIDE_BAND = 128, 256, 512, 1024, 2048, 4096
DEBUG_NO_ROW_CHUNKS = 8192
DEBUG_RANGES_EXACT_REVERSE = 16384
DEBUG_ALIGN_NO_CERTIFICATE = 32768
OPTION_EXACT_PRUNING = 1
INT_TYPES = {"i8": 0, "i16": 1, "i32": 2, "u8": 3, "u16": 4, "u32": 5}

# every symbol include/zoe_sw.h declares (tests/test_capi_symbols.py checks the two lists agree)
SYMBOLS = [
    "zsw_pack4_host", "zsw_create", "zsw_destroy", "zsw_last_error_string", "zsw_device_count", "zsw_set_scoring", "zsw_set_reference",
    "zsw_score_batch", "zsw_score_batch_from", "zsw_score_ends_batch", "zsw_score_ranges_batch", "zsw_score_ranges_batch_from", "zsw_align_batch", "zsw_align_batch_from", "zsw_align_3pass_batch", "zsw_align_3pass_batch_from", "zsw_sneaky_snake_batch",
    "zsw_set_profile_sequence", "zsw_score_shared_batch", "zsw_score_shared_batch_from", "zsw_score_ends_shared_batch", "zsw_score_ranges_shared_batch",
    "zsw_score_ranges_shared_batch_from", "zsw_align_shared_batch", "zsw_align_shared_batch_from", "zsw_align_3pass_shared_batch", "zsw_align_3pass_shared_batch_from",
    "zsw_synth_reads", "zsw_synth_reads_ragged", "zsw_synth_length", "zsw_synth_reference_host", "zsw_synth_reads_host",
    "zsw_synth_reads_ragged_host", "zsw_selftest", "zsw_timing_enable", "zsw_timing_read", "zsw_timing_read_window", "zsw_debug_set", "zsw_debug_band_records", "zsw_prune_rescored", "zsw_set_option",
    "zsw_group_create", "zsw_group_destroy", "zsw_group_size", "zsw_group_context", "zsw_group_last_error_string", "zsw_group_set_scoring",
    "zsw_group_set_reference", "zsw_group_score_batch_from", "zsw_group_score_batch_from_device", "zsw_group_align_batch_from",
    "zsw_group_align_3pass_batch_from",
]


class ZswBatch(C.Structure):
    _fields_ = [
        ("bases", C.c_void_p),
        ("offsets", C.c_void_p),
        ("fixed_len", C.c_uint32),
        ("n_reads", C.c_uint64),
        ("mem", C.c_int),
        ("encoding", C.c_int),
    ]


class ZswAlignment(C.Structure):
    _fields_ = [
        ("score", C.c_uint32),
        ("ref_start", C.c_uint32),
        ("ref_end", C.c_uint32),
        ("query_start", C.c_uint32),
        ("query_end", C.c_uint32),
        ("ref_len", C.c_uint32),
        ("query_len", C.c_uint32),
        ("n_ciglets", C.c_uint32),
        ("ciglet_offset", C.c_uint64),
    ]


class ZswError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        name = ERROR_NAMES.get(code, str(code))
        super().__init__(f"{name}{': ' + detail if detail else ''}")


_lib = None


def load() -> C.CDLL:
    """Loads the HIP library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZswError(-3, f"{LIB_PATH} not built: run `python -m zoe_amd.build` (hipcc, gfx950)")
    # PyTorch owns device memory and streams here and bundles its own libamdhip64 (SONAME libamdhip64.so.7).
    # It must be loaded first so that this library binds to the SAME HIP runtime instance; the other order
    # puts two runtimes in one process and torch's device pointers would mean nothing to ours.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    lib.zsw_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.zsw_destroy.argtypes = [vp]
    lib.zsw_destroy.restype = None
    lib.zsw_last_error_string.argtypes = [vp]
    lib.zsw_last_error_string.restype = C.c_char_p
    lib.zsw_set_scoring.argtypes = [vp, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.zsw_set_reference.argtypes = [vp, C.c_void_p, C.c_size_t, C.c_int]
    lib.zsw_score_batch.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, u32p, u8p, vp]
    lib.zsw_score_batch_from.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, u32p, u8p, u8p, vp]
    lib.zsw_score_ends_batch.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, u32p, u32p, u32p, u8p, vp]
    lib.zsw_score_ranges_batch.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, u32p, u32p, u32p, u32p, u32p, u8p, vp]
    lib.zsw_score_ranges_batch_from.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, u32p, u32p, u32p, u32p, u32p, u8p, u8p, vp]
    lib.zsw_align_batch.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, C.c_int, vp, u8p, u32p, u8p, C.c_uint64, C.POINTER(C.c_uint64), vp]
    lib.zsw_align_batch_from.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, C.c_int, vp, u8p, u8p, u32p, u8p, C.c_uint64, C.POINTER(C.c_uint64), vp]
    lib.zsw_align_3pass_batch.argtypes = lib.zsw_align_batch.argtypes
    lib.zsw_align_3pass_batch_from.argtypes = lib.zsw_align_batch_from.argtypes
    lib.zsw_set_profile_sequence.argtypes = [vp, C.c_void_p, C.c_size_t, C.c_int]
    lib.zsw_score_shared_batch.argtypes = lib.zsw_score_batch.argtypes
    lib.zsw_score_shared_batch_from.argtypes = lib.zsw_score_batch_from.argtypes
    lib.zsw_score_ends_shared_batch.argtypes = lib.zsw_score_ends_batch.argtypes
    lib.zsw_score_ranges_shared_batch.argtypes = lib.zsw_score_ranges_batch.argtypes
    lib.zsw_score_ranges_shared_batch_from.argtypes = lib.zsw_score_ranges_batch_from.argtypes
    lib.zsw_align_shared_batch.argtypes = lib.zsw_align_batch.argtypes
    lib.zsw_align_shared_batch_from.argtypes = lib.zsw_align_batch_from.argtypes
    lib.zsw_align_3pass_shared_batch.argtypes = lib.zsw_align_shared_batch.argtypes
    lib.zsw_align_3pass_shared_batch_from.argtypes = lib.zsw_align_batch_from.argtypes
    lib.zsw_sneaky_snake_batch.argtypes = [vp, C.POINTER(ZswBatch), u32p, u32p, C.c_float, u8p, vp]
    lib.zsw_synth_reads.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp]
    lib.zsw_synth_reads_ragged.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, vp, vp]
    lib.zsw_synth_length.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.zsw_synth_length.restype = C.c_uint32
    lib.zsw_synth_reference_host.argtypes = [C.c_uint64, C.c_uint64, vp]
    lib.zsw_synth_reference_host.restype = None
    lib.zsw_synth_reads_host.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp, C.c_uint32, vp]
    lib.zsw_synth_reads_host.restype = None
    lib.zsw_synth_reads_ragged_host.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, vp, C.c_uint32, vp]
    lib.zsw_synth_reads_ragged_host.restype = None
    lib.zsw_selftest.argtypes = [vp]
    lib.zsw_timing_enable.argtypes = [vp, C.c_int]
    lib.zsw_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.zsw_timing_read_window.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.zsw_debug_set.argtypes = [vp, C.c_uint32]
    lib.zsw_debug_band_records.argtypes = [vp, vp]
    lib.zsw_pack4_host.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp]
    lib.zsw_prune_rescored.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.zsw_set_option.argtypes = [vp, C.c_int, C.c_int64]
    lib.zsw_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.zsw_group_destroy.argtypes = [vp]
    lib.zsw_group_destroy.restype = None
    lib.zsw_group_size.argtypes = [vp]
    lib.zsw_group_context.argtypes = [vp, C.c_int]
    lib.zsw_group_context.restype = vp
    lib.zsw_group_last_error_string.argtypes = [vp]
    lib.zsw_group_last_error_string.restype = C.c_char_p
    lib.zsw_group_set_scoring.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int]
    lib.zsw_group_set_reference.argtypes = [vp, vp, C.c_size_t]
    lib.zsw_group_score_batch_from.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, vp, vp, vp]
    lib.zsw_group_score_batch_from_device.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    lib.zsw_group_align_batch_from.argtypes = [vp, C.POINTER(ZswBatch), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.zsw_group_align_3pass_batch_from.argtypes = lib.zsw_group_align_batch_from.argtypes
    _lib = lib
    return lib
