"""zoe_amd — MI355X (gfx950) implementation of CDCgov/zoe's striped Smith-Waterman hot path.

Hand-written HIP kernels behind a C ABI (include/zoe_sw.h), plus a host-side mirror of the
reference's profile / weight-matrix / sw_* interface (zoe_amd.alignment).
"""
from .alignment import (  # noqa: F401
    DNA_PROFILE_MAP, EMPTY, FILTER_NONE, FILTER_PASS, FILTER_REJECT, OVERFLOWED, SOME, UNMAPPED, AlignmentBatch, ByteIndexMap, LocalProfilesBatch, ProfileError,
    ReadBatch, ScoreBatch, SeqBatchSrc, SeqSrc, SharedProfilesBatch, SharedStripedProfile, StripedProfileBatch, SwContext, SwGroup, WeightMatrix, into_local_profile, into_shared_profile, sneaky_snake, validate_profile_args,
)
