//! `zoe_sw_gpu.rs` — Rust side of the MI355X striped Smith-Waterman library (`include/zoe_sw.h`,
//! `libzoe_sw_hip.so`). Meant to live in the Zoe crate as `src/alignment/sw/gpu.rs` behind a
//! `mi355x` cargo feature; it rebuilds Zoe's own result types, so callers keep the crate's API:
//!
//! | here                                   | replaces, one call per read                                   |
//! |----------------------------------------|----------------------------------------------------------------|
//! | `GpuContext::sw_score_batch`           | `StripedProfile::<T,N,S>::new(read,..)?.sw_score(reference)` (`sw/striped.rs:65`) |
//! | `GpuContext::sw_score_from_batch`      | `LocalProfiles::new_with_w*(read,..)?.sw_score_from_i{8,16,32}` (`profile_set.rs:71-107`) |
//! | `GpuContext::sw_score_ends_batch`      | `sw_simd_score_ends` (`sw/striped.rs:153`) -> `ScoreEnds<u32>` (`types/output.rs:201-208`) |
//! | `GpuContext::sw_score_ranges_batch`    | `sw_simd_score_ranges` (`sw/striped.rs:355`) -> `ScoreAndRanges<u32>` (`types/output.rs:212-219`) |
//! | `GpuContext::sw_align_batch`           | `sw_simd_align` (`sw/striped.rs:449`) -> `Alignment<u32>` (`types/output.rs:264-279`) |
//! | `GpuContext::sw_align_from_batch`      | `ProfileSets::sw_align_from_i{8,16,32}` (`profile_set.rs:124-179`) |
//! | `GpuContext::sw_align_3pass_batch`     | `sw_align_3pass` (`sw/three_pass.rs:21-104`) |
//! | `GpuContext::sneaky_snake_batch`       | `sneaky_snake` (`sneaky_snake.rs:78-131`) |
//! | `GpuGroup::sw_score_from_batch`        | the same over several GPUs (`zsw_group_*`) |
//!
//! NOT COMPILED in the repository this file ships in (the build image has no Rust toolchain);
//! `tests/test_rust_binding.py` checks every `extern "C"` item against `include/zoe_sw.h`
//! (names, arity, integer widths, pointer constness) so that the two cannot drift apart.
#![allow(clippy::too_many_arguments)]

use crate::alignment::{
    Alignment, AlignmentStates, MaybeAligned, ProfileError, ScoreAndRanges, ScoreEnds,
};
use crate::data::{cigar::Ciglet, matrices::WeightMatrix};
use std::ffi::{CStr, c_char, c_void};
use std::ptr;

// ---------------------------------------------------------------------------------------------
// C ABI (include/zoe_sw.h)
// ---------------------------------------------------------------------------------------------

/// `zsw_context` (opaque)
#[repr(C)]
pub struct ZswContext {
    _private: [u8; 0],
}

/// `zsw_group` (opaque)
#[repr(C)]
pub struct ZswGroup {
    _private: [u8; 0],
}

/// `zsw_batch`
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ZswBatch {
    pub bases:     *const u8,
    pub offsets:   *const u64,
    pub fixed_len: u32,
    pub n_reads:   u64,
    pub mem:       i32,
    pub encoding:  i32,
}

/// `zsw_alignment`
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct ZswAlignment {
    pub score:         u32,
    pub ref_start:     u32,
    pub ref_end:       u32,
    pub query_start:   u32,
    pub query_end:     u32,
    pub ref_len:       u32,
    pub query_len:     u32,
    pub n_ciglets:     u32,
    pub ciglet_offset: u64,
}

pub const ZSW_OK: i32 = 0;
pub const ZSW_MEM_HOST: i32 = 0;
pub const ZSW_ENCODING_BYTES: i32 = 0;
pub const ZSW_ENCODING_PACKED4: i32 = 1;
pub const ZSW_MEM_DEVICE: i32 = 1;
pub const ZSW_STATUS_SOME: u8 = 0;
pub const ZSW_STATUS_OVERFLOWED: u8 = 1;
pub const ZSW_STATUS_UNMAPPED: u8 = 2;
pub const ZSW_STATUS_EMPTY: u8 = 3;

/// `zsw_int_type`: T of `StripedProfile<T, N, S>` (`math/integer.rs:231-238`)
#[repr(i32)]
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum ZswIntType {
    I8  = 0,
    I16 = 1,
    I32 = 2,
    U8  = 3,
    U16 = 4,
    U32 = 5,
}

#[link(name = "zoe_sw_hip")]
unsafe extern "C" {
    fn zsw_pack4_host(ctx: *mut ZswContext, bases: *const u8, n_reads: u64, len: u32, out_packed: *mut u8) -> i32;
    fn zsw_create(device_id: i32, out: *mut *mut ZswContext) -> i32;
    fn zsw_destroy(ctx: *mut ZswContext);
    fn zsw_last_error_string(ctx: *const ZswContext) -> *const c_char;
    fn zsw_device_count() -> i32;
    fn zsw_set_scoring(ctx: *mut ZswContext, weights: *const i8, s: i32, index_map: *const u8, gap_open: i32, gap_extend: i32) -> i32;
    fn zsw_set_reference(ctx: *mut ZswContext, reference: *const u8, len: usize, mem: i32) -> i32;
    fn zsw_score_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, out_score: *mut u32, out_status: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, out_score: *mut u32, out_status: *mut u8, out_tier: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_ends_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, out_score: *mut u32, out_ref_end: *mut u32, out_query_end: *mut u32, out_status: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_ranges_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, out_score: *mut u32, out_ref_start: *mut u32, out_ref_end: *mut u32, out_query_start: *mut u32, out_query_end: *mut u32, out_status: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_ranges_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, out_score: *mut u32, out_ref_start: *mut u32, out_ref_end: *mut u32, out_query_start: *mut u32, out_query_end: *mut u32, out_status: *mut u8, out_tier: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_align_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_align_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_tier: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_align_3pass_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_align_3pass_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_tier: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_set_profile_sequence(ctx: *mut ZswContext, sequence: *const u8, len: usize, mem: i32) -> i32;
    fn zsw_score_shared_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, out_score: *mut u32, out_status: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_shared_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, out_score: *mut u32, out_status: *mut u8, out_tier: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_ends_shared_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, out_score: *mut u32, out_ref_end: *mut u32, out_query_end: *mut u32, out_status: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_ranges_shared_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, out_score: *mut u32, out_ref_start: *mut u32, out_ref_end: *mut u32, out_query_start: *mut u32, out_query_end: *mut u32, out_status: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_score_ranges_shared_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, out_score: *mut u32, out_ref_start: *mut u32, out_ref_end: *mut u32, out_query_start: *mut u32, out_query_end: *mut u32, out_status: *mut u8, out_tier: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_align_shared_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_align_shared_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_tier: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_align_3pass_shared_batch(ctx: *mut ZswContext, reads: *const ZswBatch, int_type: i32, lanes: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_align_3pass_shared_batch_from(ctx: *mut ZswContext, reads: *const ZswBatch, from_width: i32, preset_bits: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_tier: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64, stream: *mut c_void) -> i32;
    fn zsw_sneaky_snake_batch(ctx: *mut ZswContext, reads: *const ZswBatch, ref_start: *const u32, ref_len: *const u32, threshold: f32, out_pass: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_group_create(device_ids: *const i32, n_devices: i32, out: *mut *mut ZswGroup) -> i32;
    fn zsw_group_destroy(group: *mut ZswGroup);
    fn zsw_group_size(group: *const ZswGroup) -> i32;
    fn zsw_group_context(group: *mut ZswGroup, i: i32) -> *mut ZswContext;
    fn zsw_group_last_error_string(group: *const ZswGroup) -> *const c_char;
    fn zsw_group_set_scoring(group: *mut ZswGroup, weights: *const i8, s: i32, index_map: *const u8, gap_open: i32, gap_extend: i32) -> i32;
    fn zsw_group_set_reference(group: *mut ZswGroup, reference: *const u8, len: usize) -> i32;
    fn zsw_group_score_batch_from(group: *mut ZswGroup, reads: *const ZswBatch, from_width: i32, preset_bits: i32, out_score: *mut u32, out_status: *mut u8, out_tier: *mut u8) -> i32;
    fn zsw_group_align_batch_from(group: *mut ZswGroup, reads: *const ZswBatch, from_width: i32, preset_bits: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_tier: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64) -> i32;
    fn zsw_group_align_3pass_batch_from(group: *mut ZswGroup, reads: *const ZswBatch, from_width: i32, preset_bits: i32, invert: i32, out_aln: *mut ZswAlignment, out_status: *mut u8, out_tier: *mut u8, out_inc: *mut u32, out_op: *mut u8, ciglet_cap: u64, out_n_ciglets: *mut u64) -> i32;
    fn zsw_group_score_batch_from_device(group: *mut ZswGroup, shards: *const ZswBatch, from_width: i32, preset_bits: i32, out_score: *const *mut u32, out_status: *const *mut u8) -> i32;
    fn zsw_synth_reads(ctx: *mut ZswContext, seed: u64, first: u64, n: u64, len: u32, out_device: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_synth_reads_ragged(ctx: *mut ZswContext, seed: u64, first: u64, n: u64, min_len: u32, max_len: u32, offsets_device: *const u64, out_device: *mut u8, stream: *mut c_void) -> i32;
    fn zsw_synth_length(seed: u64, index: u64, min_len: u32, max_len: u32) -> u32;
    fn zsw_selftest(ctx: *mut ZswContext) -> i32;
    fn zsw_synth_reference_host(seed: u64, len: u64, out: *mut u8);
    fn zsw_synth_reads_host(seed: u64, first: u64, n: u64, len: u32, reference: *const u8, r: u32, out: *mut u8);
    fn zsw_synth_reads_ragged_host(seed: u64, first: u64, n: u64, min_len: u32, max_len: u32, offsets: *const u64, reference: *const u8, r: u32, out: *mut u8);
    fn zsw_timing_enable(ctx: *mut ZswContext, enable: i32) -> i32;
    fn zsw_timing_read(ctx: *mut ZswContext, seconds: *mut f64, launches: *mut u64) -> i32;
    fn zsw_timing_read_window(ctx: *mut ZswContext, seconds: *mut f64, launches: *mut u64) -> i32;
    fn zsw_debug_set(ctx: *mut ZswContext, flags: u32) -> i32;
    fn zsw_debug_band_records(ctx: *mut ZswContext, records: *mut i32) -> i32;
    fn zsw_prune_rescored(ctx: *mut ZswContext, out_reads: *mut u64) -> i32;
    fn zsw_set_option(ctx: *mut ZswContext, option: i32, value: i64) -> i32;
}

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------

/// What a call can fail with: one of Zoe's own [`ProfileError`]s (codes 1..=4 of `zsw_error`,
/// `alignment/errors.rs:6-15`) or a failure of the GPU library itself.
#[derive(Debug)]
pub enum GpuError {
    Profile(ProfileError),
    /// `ZSW_ERR_*` code < 0 and the library's message
    Library { code: i32, message: String },
}

impl From<ProfileError> for GpuError {
    fn from(e: ProfileError) -> Self {
        GpuError::Profile(e)
    }
}

fn c_message(p: *const c_char) -> String {
    if p.is_null() {
        String::new()
    } else {
        // SAFETY: the library returns a NUL-terminated string that lives until the next call
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

/// Codes 1..=4 are `ProfileError` exactly as `validate_profile_args` raises them
/// (`alignment/profile.rs:32-44`).
fn profile_error(code: i32, gap_open: i8, gap_extend: i8) -> Option<ProfileError> {
    match code {
        1 => Some(ProfileError::EmptySequence),
        2 => Some(ProfileError::GapOpenOutOfRange { gap_open }),
        3 => Some(ProfileError::GapExtendOutOfRange { gap_extend }),
        4 => Some(ProfileError::BadGapWeights { gap_open, gap_extend }),
        _ => None,
    }
}

// ---------------------------------------------------------------------------------------------
// inputs
// ---------------------------------------------------------------------------------------------

/// Scoring as passed to `StripedProfile::new`: the signed matrix, its byte map and the gap weights.
pub struct Scoring<'a, const S: usize> {
    pub matrix:     &'a WeightMatrix<'a, i8, S>,
    pub gap_open:   i8,
    pub gap_extend: i8,
}

impl<const S: usize> Scoring<'_, S> {
    /// `weights[r*S + q]`, row = reference residue (`data/matrices/mod.rs:242-244`)
    fn flat_weights(&self) -> Vec<i8> {
        self.matrix.weights.iter().flat_map(|row| row.iter().copied()).collect()
    }

    /// `ByteIndexMap::to_index` for every byte (`mappings/byte_index.rs:331-333`)
    fn index_map(&self) -> [u8; 256] {
        let mut out = [0u8; 256];
        for b in 0..=255u8 {
            out[b as usize] = self.matrix.mapping.to_index(b) as u8;
        }
        out
    }
}

/// Reads flattened for the C ABI: concatenated bases + `n+1` offsets (host memory).
struct HostBatch {
    bases:   Vec<u8>,
    offsets: Vec<u64>,
}

impl HostBatch {
    fn new<Q: AsRef<[u8]>>(reads: &[Q]) -> Self {
        let total: usize = reads.iter().map(|r| r.as_ref().len()).sum();
        let mut bases = Vec::with_capacity(total.max(1));
        let mut offsets = Vec::with_capacity(reads.len() + 1);
        offsets.push(0u64);
        for r in reads {
            bases.extend_from_slice(r.as_ref());
            offsets.push(bases.len() as u64);
        }
        if bases.is_empty() {
            bases.push(0); // a valid pointer for empty batches
        }
        HostBatch { bases, offsets }
    }

    fn as_c(&self) -> ZswBatch {
        ZswBatch {
            bases:     self.bases.as_ptr(),
            offsets:   self.offsets.as_ptr(),
            fixed_len: 0,
            n_reads:   (self.offsets.len() - 1) as u64,
            mem:       ZSW_MEM_HOST,
            encoding:  ZSW_ENCODING_BYTES,
        }
    }
}

// ---------------------------------------------------------------------------------------------
// outputs -> Zoe's types
// ---------------------------------------------------------------------------------------------

/// `(status, value)` -> `MaybeAligned<T>` (`types/output.rs:18-25`). A read of length 0 is the
/// `Err(ProfileError::EmptySequence)` that `StripedProfile::new` would have returned for it.
fn maybe<T>(status: u8, value: impl FnOnce() -> T) -> Result<MaybeAligned<T>, ProfileError> {
    match status {
        ZSW_STATUS_SOME => Ok(MaybeAligned::Some(value())),
        ZSW_STATUS_OVERFLOWED => Ok(MaybeAligned::Overflowed),
        ZSW_STATUS_UNMAPPED => Ok(MaybeAligned::Unmapped),
        _ => Err(ProfileError::EmptySequence),
    }
}

/// `zsw_alignment` + its slice of the ciglet arrays -> `Alignment<u32>` (`types/output.rs:264-279`,
/// `AlignmentStates::from_ciglets_unchecked`, `types/state.rs:259-268`).
fn alignment_of(rec: &ZswAlignment, inc: &[u32], op: &[u8]) -> Alignment<u32> {
    let first = rec.ciglet_offset as usize;
    let last = first + rec.n_ciglets as usize;
    let states = AlignmentStates::from_ciglets_unchecked(
        inc[first..last].iter().zip(&op[first..last]).map(|(&inc, &op)| Ciglet { inc: inc as usize, op }),
    );
    let mut aln = Alignment::<u32>::default(); // the struct is #[non_exhaustive]
    aln.score = rec.score;
    aln.ref_range = rec.ref_start as usize..rec.ref_end as usize;
    aln.query_range = rec.query_start as usize..rec.query_end as usize;
    aln.states = states;
    aln.ref_len = rec.ref_len as usize;
    aln.query_len = rec.query_len as usize;
    aln
}

/// Which profile the reference sequence is passed as (`SeqSrc`, `alignment/mod.rs:176-190`): with
/// `Query`, the library returns the alignment already passed through `Alignment::invert`.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum OtherSeq {
    Reference,
    Query,
}

/// Width the cascade starts at (`sw_*_from_i8 / _i16 / _i32`) and lane preset (`new_with_w128/256/512`).
#[derive(Clone, Copy, Debug)]
pub struct Cascade {
    pub from_width:  i32,
    pub preset_bits: i32,
}

impl Cascade {
    /// `Nucleotides::into_local_profile(..).sw_*_from_i8(..)` (`nucleotides/mod.rs:262-266`)
    pub const LOCAL_PROFILE_FROM_I8: Cascade = Cascade { from_width: 8, preset_bits: 256 };
}

// ---------------------------------------------------------------------------------------------
// one GPU
// ---------------------------------------------------------------------------------------------

/// One MI355X. Not `Sync`: the library allows one host thread per context at a time.
pub struct GpuContext {
    raw: *mut ZswContext,
}

unsafe impl Send for GpuContext {}

impl Drop for GpuContext {
    fn drop(&mut self) {
        // SAFETY: `raw` came from zsw_create and is destroyed once
        unsafe { zsw_destroy(self.raw) }
    }
}

impl GpuContext {
    /// Number of usable GPUs (0 when the HIP runtime finds none).
    #[must_use]
    pub fn device_count() -> usize {
        // SAFETY: no arguments
        unsafe { zsw_device_count() }.max(0) as usize
    }

    pub fn new(device_id: i32) -> Result<Self, GpuError> {
        let mut raw: *mut ZswContext = ptr::null_mut();
        // SAFETY: `raw` is a valid out-pointer
        let code = unsafe { zsw_create(device_id, &mut raw) };
        if code != ZSW_OK {
            // SAFETY: NULL asks for the reason the last zsw_create of this thread failed
            let message = c_message(unsafe { zsw_last_error_string(ptr::null()) });
            return Err(GpuError::Library { code, message });
        }
        Ok(GpuContext { raw })
    }

    fn check(&self, code: i32, gap_open: i8, gap_extend: i8) -> Result<(), GpuError> {
        if code == ZSW_OK {
            return Ok(());
        }
        if let Some(e) = profile_error(code, gap_open, gap_extend) {
            return Err(GpuError::Profile(e));
        }
        // SAFETY: `raw` is a live context
        let message = c_message(unsafe { zsw_last_error_string(self.raw) });
        Err(GpuError::Library { code, message })
    }

    /// `zsw_set_scoring` + `zsw_set_reference` (host memory).
    fn configure<const S: usize>(&self, scoring: &Scoring<'_, S>, reference: &[u8]) -> Result<(), GpuError> {
        let w = scoring.flat_weights();
        let map = scoring.index_map();
        let (go, ge) = (scoring.gap_open, scoring.gap_extend);
        // SAFETY: pointers are valid for S*S / 256 / reference.len() bytes for the duration of the calls
        self.check(unsafe { zsw_set_scoring(self.raw, w.as_ptr(), S as i32, map.as_ptr(), i32::from(go), i32::from(ge)) }, go, ge)?;
        let p = if reference.is_empty() { [0u8].as_ptr() } else { reference.as_ptr() };
        self.check(unsafe { zsw_set_reference(self.raw, p, reference.len(), ZSW_MEM_HOST) }, go, ge)
    }

    /// Per read: `StripedProfile::<T, N, S>::new(read, matrix, go, ge)?.sw_score(reference)`.
    /// The outer `Result` is a failure of the whole batch; the inner one is the per-read
    /// `Result<_, ProfileError>` of `StripedProfile::new` (an empty read).
    pub fn sw_score_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32,
    ) -> Result<Vec<Result<MaybeAligned<u32>, ProfileError>>, GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let (mut score, mut status) = (vec![0u32; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: output arrays hold n entries; NULL stream = default stream; host batches return when the results are written
        let code = unsafe { zsw_score_batch(self.raw, &batch.as_c(), int_type as i32, lanes, score.as_mut_ptr(), status.as_mut_ptr(), ptr::null_mut()) };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok((0..n).map(|i| maybe(status[i], || score[i])).collect())
    }

    /// Per read: `LocalProfiles::new_with_w{preset}(read, ..)?.sw_score_from_i{from_width}(reference)`;
    /// the second element is the integer width whose profile answered (8, 16 or 32).
    pub fn sw_score_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, cascade: Cascade,
    ) -> Result<Vec<(Result<MaybeAligned<u32>, ProfileError>, u8)>, GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let (mut score, mut status, mut tier) = (vec![0u32; n.max(1)], vec![0u8; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_batch_from(self.raw, &batch.as_c(), cascade.from_width, cascade.preset_bits, score.as_mut_ptr(), status.as_mut_ptr(), tier.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok((0..n).map(|i| (maybe(status[i], || score[i]), tier[i])).collect())
    }

    /// Per read: `profile.sw_score_ends(SeqSrc::Reference(reference))` -> `ScoreEnds<u32>`.
    pub fn sw_score_ends_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32,
    ) -> Result<Vec<Result<MaybeAligned<ScoreEnds<u32>>, ProfileError>>, GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let (mut score, mut r_end, mut q_end, mut status) = (vec![0u32; n.max(1)], vec![0u32; n.max(1)], vec![0u32; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_ends_batch(self.raw, &batch.as_c(), int_type as i32, lanes, score.as_mut_ptr(), r_end.as_mut_ptr(), q_end.as_mut_ptr(), status.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok((0..n)
            .map(|i| maybe(status[i], || ScoreEnds { score: score[i], ref_end: r_end[i] as usize, query_end: q_end[i] as usize }))
            .collect())
    }

    fn ranges_out(n: usize, status: &[u8], score: &[u32], rs: &[u32], re: &[u32], qs: &[u32], qe: &[u32]) -> Vec<Result<MaybeAligned<ScoreAndRanges<u32>>, ProfileError>> {
        (0..n)
            .map(|i| {
                maybe(status[i], || ScoreAndRanges {
                    score:       score[i],
                    ref_range:   rs[i] as usize..re[i] as usize,
                    query_range: qs[i] as usize..qe[i] as usize,
                })
            })
            .collect()
    }

    /// Per read: `profile.sw_score_ranges(SeqSrc::Reference(reference))` -> `ScoreAndRanges<u32>`.
    pub fn sw_score_ranges_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32,
    ) -> Result<Vec<Result<MaybeAligned<ScoreAndRanges<u32>>, ProfileError>>, GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let m = n.max(1);
        let (mut score, mut rs, mut re, mut qs, mut qe, mut status) = (vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u8; m]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_ranges_batch(self.raw, &batch.as_c(), int_type as i32, lanes, score.as_mut_ptr(), rs.as_mut_ptr(), re.as_mut_ptr(), qs.as_mut_ptr(), qe.as_mut_ptr(), status.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok(Self::ranges_out(n, &status, &score, &rs, &re, &qs, &qe))
    }

    /// Per read: `profiles.sw_score_ranges_from_i{from_width}(SeqSrc::Reference(reference))` (`profile_set.rs:313-362`).
    pub fn sw_score_ranges_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, cascade: Cascade,
    ) -> Result<Vec<Result<MaybeAligned<ScoreAndRanges<u32>>, ProfileError>>, GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let m = n.max(1);
        let (mut score, mut rs, mut re, mut qs, mut qe, mut status, mut tier) = (vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u8; m], vec![0u8; m]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_ranges_batch_from(self.raw, &batch.as_c(), cascade.from_width, cascade.preset_bits, score.as_mut_ptr(), rs.as_mut_ptr(), re.as_mut_ptr(), qs.as_mut_ptr(), qe.as_mut_ptr(), status.as_mut_ptr(), tier.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok(Self::ranges_out(n, &status, &score, &rs, &re, &qs, &qe))
    }

    // ---- the one-profile-many-sequences role: `sequence.into_shared_profile(..)` once, every read against it ----
    // (sw/mod.rs:63-67, profile_set.rs:552-560, nucleotides/mod.rs:295-299). In the results the READ is the reference of
    // sw_simd_* unless `other` is `OtherSeq::Query`, which is `SeqSrc::Query(read)` (alignment/mod.rs:176-190).

    fn configure_shared<const S: usize>(&self, scoring: &Scoring<'_, S>, sequence: &[u8]) -> Result<(), GpuError> {
        let w = scoring.flat_weights();
        let map = scoring.index_map();
        let (go, ge) = (scoring.gap_open, scoring.gap_extend);
        // SAFETY: pointers are valid for S*S / 256 / sequence.len() bytes for the duration of the calls
        self.check(unsafe { zsw_set_scoring(self.raw, w.as_ptr(), S as i32, map.as_ptr(), i32::from(go), i32::from(ge)) }, go, ge)?;
        let p = if sequence.is_empty() { [0u8].as_ptr() } else { sequence.as_ptr() };
        // an empty sequence comes back as ProfileError::EmptySequence through check()
        self.check(unsafe { zsw_set_profile_sequence(self.raw, p, sequence.len(), ZSW_MEM_HOST) }, go, ge)
    }

    /// Per read: `StripedProfile::<T, N, S>::new(sequence, ..)?.sw_score(read)`.
    pub fn sw_score_shared_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32,
    ) -> Result<Vec<Result<MaybeAligned<u32>, ProfileError>>, GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let (mut score, mut status) = (vec![0u32; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: output arrays hold n entries
        let code = unsafe { zsw_score_shared_batch(self.raw, &batch.as_c(), int_type as i32, lanes, score.as_mut_ptr(), status.as_mut_ptr(), ptr::null_mut()) };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok((0..n).map(|i| maybe(status[i], || score[i])).collect())
    }

    /// Per read: `shared_profiles.sw_score_from_i{from_width}(read)`; the second vector is the width that answered.
    pub fn sw_score_shared_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, cascade: Cascade,
    ) -> Result<(Vec<Result<MaybeAligned<u32>, ProfileError>>, Vec<u8>), GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let (mut score, mut status, mut tier) = (vec![0u32; n.max(1)], vec![0u8; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_shared_batch_from(self.raw, &batch.as_c(), cascade.from_width, cascade.preset_bits, score.as_mut_ptr(), status.as_mut_ptr(), tier.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok(((0..n).map(|i| maybe(status[i], || score[i])).collect(), tier))
    }

    /// Per read: `profile.sw_score_ends(SeqSrc::Reference(read))`: `ref_end` in the read, `query_end` in the sequence.
    pub fn sw_score_ends_shared_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32,
    ) -> Result<Vec<Result<MaybeAligned<ScoreEnds<u32>>, ProfileError>>, GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let (mut score, mut r_end, mut q_end, mut status) = (vec![0u32; n.max(1)], vec![0u32; n.max(1)], vec![0u32; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_ends_shared_batch(self.raw, &batch.as_c(), int_type as i32, lanes, score.as_mut_ptr(), r_end.as_mut_ptr(), q_end.as_mut_ptr(), status.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok((0..n)
            .map(|i| maybe(status[i], || ScoreEnds { score: score[i], ref_end: r_end[i] as usize, query_end: q_end[i] as usize }))
            .collect())
    }

    /// Per read: `profile.sw_score_ranges(SeqSrc::Reference(read))` at `<T, N>`.
    pub fn sw_score_ranges_shared_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32,
    ) -> Result<Vec<Result<MaybeAligned<ScoreAndRanges<u32>>, ProfileError>>, GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let m = n.max(1);
        let (mut score, mut rs, mut re, mut qs, mut qe, mut status) = (vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u8; m]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_ranges_shared_batch(self.raw, &batch.as_c(), int_type as i32, lanes, score.as_mut_ptr(), rs.as_mut_ptr(), re.as_mut_ptr(), qs.as_mut_ptr(), qe.as_mut_ptr(), status.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok(Self::ranges_out(n, &status, &score, &rs, &re, &qs, &qe))
    }

    /// Per read: `shared_profiles.sw_score_ranges_from_i{from_width}(SeqSrc::Reference(read))`.
    pub fn sw_score_ranges_shared_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, cascade: Cascade,
    ) -> Result<Vec<Result<MaybeAligned<ScoreAndRanges<u32>>, ProfileError>>, GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let m = n.max(1);
        let (mut score, mut rs, mut re, mut qs, mut qe, mut status, mut tier) = (vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u32; m], vec![0u8; m], vec![0u8; m]);
        // SAFETY: as above
        let code = unsafe {
            zsw_score_ranges_shared_batch_from(self.raw, &batch.as_c(), cascade.from_width, cascade.preset_bits, score.as_mut_ptr(), rs.as_mut_ptr(), re.as_mut_ptr(), qs.as_mut_ptr(), qe.as_mut_ptr(), status.as_mut_ptr(), tier.as_mut_ptr(), ptr::null_mut())
        };
        self.check(code, scoring.gap_open, scoring.gap_extend)?;
        Ok(Self::ranges_out(n, &status, &score, &rs, &re, &qs, &qe))
    }

    /// Per read: `profile.sw_align(SeqSrc::Query(read))` (or `SeqSrc::Reference`) with the profile of `sequence` at `<T, N>`.
    pub fn sw_align_shared_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32, other: OtherSeq,
    ) -> Result<Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        let (out, _) = self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, _tier, inc, op, cap, need| {
            // SAFETY: all arrays were sized by align_with
            unsafe { zsw_align_shared_batch(self.raw, &c, int_type as i32, lanes, invert, aln, st, inc, op, cap, need, ptr::null_mut()) }
        })?;
        Ok(out)
    }

    /// Per read: `sequence.into_shared_profile(..)?.sw_align_from_i{from_width}(SeqSrc::Query(read))`; the second vector is the
    /// width that answered.
    pub fn sw_align_shared_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, cascade: Cascade, other: OtherSeq,
    ) -> Result<(Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, Vec<u8>), GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, tier, inc, op, cap, need| {
            // SAFETY: as above
            unsafe { zsw_align_shared_batch_from(self.raw, &c, cascade.from_width, cascade.preset_bits, invert, aln, st, tier, inc, op, cap, need, ptr::null_mut()) }
        })
    }

    /// Per read: `StripedProfile::<T, N, S>::new(sequence, ..)?.sw_align_3pass(SeqSrc::Query(read), sequence, ..)` (profile.rs:536-552 ->
    /// three_pass.rs:21-104) with ONE profile for the whole batch.
    pub fn sw_align_3pass_shared_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, int_type: IntType, lanes: i32, other: OtherSeq,
    ) -> Result<Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        let (out, _) = self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, _tier, inc, op, cap, need| {
            // SAFETY: all arrays were sized by align_with
            unsafe { zsw_align_3pass_shared_batch(self.raw, &c, int_type as i32, lanes, invert, aln, st, inc, op, cap, need, ptr::null_mut()) }
        })?;
        Ok(out)
    }

    /// Per read: `sequence.into_shared_profile(..)?.sw_align_from_i{from_width}_3pass(SeqSrc::Query(read))` (profile_set.rs:212-283,
    /// 552-560); the second vector is the width that answered.
    pub fn sw_align_3pass_shared_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, sequence: &[u8], reads: &[Q], scoring: &Scoring<'_, S>, cascade: Cascade, other: OtherSeq,
    ) -> Result<(Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, Vec<u8>), GpuError> {
        self.configure_shared(scoring, sequence)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, tier, inc, op, cap, need| {
            // SAFETY: as above
            unsafe { zsw_align_3pass_shared_batch_from(self.raw, &c, cascade.from_width, cascade.preset_bits, invert, aln, st, tier, inc, op, cap, need, ptr::null_mut()) }
        })
    }

    /// Shared tail of the alignment calls: the library reports the number of ciglets it needs when the arrays
    /// are too small (ZSW_ERR_INVALID_ARGUMENT with `*out_n_ciglets` = required size), so the first call sizes them.
    fn align_with(
        &self, n: usize, gap_open: i8, gap_extend: i8,
        mut call: impl FnMut(*mut ZswAlignment, *mut u8, *mut u8, *mut u32, *mut u8, u64, *mut u64) -> i32,
    ) -> Result<(Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, Vec<u8>), GpuError> {
        let m = n.max(1);
        let (mut recs, mut status, mut tier) = (vec![ZswAlignment::default(); m], vec![0u8; m], vec![0u8; m]);
        let mut cap = (4 * n).max(16);
        loop {
            let (mut inc, mut op, mut needed) = (vec![0u32; cap], vec![0u8; cap], 0u64);
            let code = call(recs.as_mut_ptr(), status.as_mut_ptr(), tier.as_mut_ptr(), inc.as_mut_ptr(), op.as_mut_ptr(), cap as u64, &mut needed);
            if code == -1 && needed as usize > cap {
                cap = needed as usize; // capacity too small: the required size came back
                continue;
            }
            self.check(code, gap_open, gap_extend)?;
            let out = (0..n).map(|i| maybe(status[i], || alignment_of(&recs[i], &inc, &op))).collect();
            return Ok((out, tier));
        }
    }

    /// Per read: `profile.sw_align(SeqSrc::Reference(reference))` (or `SeqSrc::Query`) at `<T, N>`:
    /// score, ranges and CIGAR of exactly that instantiation (the striped traceback depends on `N`).
    pub fn sw_align_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32, other: OtherSeq,
    ) -> Result<Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        let (out, _) = self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, _tier, inc, op, cap, need| {
            // SAFETY: all arrays were sized by align_with
            unsafe { zsw_align_batch(self.raw, &c, int_type as i32, lanes, invert, aln, st, inc, op, cap, need, ptr::null_mut()) }
        })?;
        Ok(out)
    }

    /// Per read: `profiles.sw_align_from_i{from_width}(seq)`: the CIGAR is that of the first tier that does not
    /// overflow (each tier has its own lane count); the second vector is the width that answered.
    pub fn sw_align_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, cascade: Cascade, other: OtherSeq,
    ) -> Result<(Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, Vec<u8>), GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, tier, inc, op, cap, need| {
            // SAFETY: as above
            unsafe { zsw_align_batch_from(self.raw, &c, cascade.from_width, cascade.preset_bits, invert, aln, st, tier, inc, op, cap, need, ptr::null_mut()) }
        })
    }

    /// Per read: `profile.sw_align_3pass(seq, ..)` (`profile.rs:546-552`).
    pub fn sw_align_3pass_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, int_type: ZswIntType, lanes: i32, other: OtherSeq,
    ) -> Result<Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        let (out, _) = self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, _tier, inc, op, cap, need| {
            // SAFETY: as above
            unsafe { zsw_align_3pass_batch(self.raw, &c, int_type as i32, lanes, invert, aln, st, inc, op, cap, need, ptr::null_mut()) }
        })?;
        Ok(out)
    }

    /// Per read: `profiles.sw_align_from_i{from_width}_3pass(seq)` (`profile_set.rs:212-283`).
    pub fn sw_align_3pass_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, cascade: Cascade, other: OtherSeq,
    ) -> Result<(Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, Vec<u8>), GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        self.align_with(reads.len(), scoring.gap_open, scoring.gap_extend, |aln, st, tier, inc, op, cap, need| {
            // SAFETY: as above
            unsafe { zsw_align_3pass_batch_from(self.raw, &c, cascade.from_width, cascade.preset_bits, invert, aln, st, tier, inc, op, cap, need, ptr::null_mut()) }
        })
    }

    /// Per read: `sneaky_snake(&reference[start..start+len], read, threshold)` -> `Option<bool>`.
    pub fn sneaky_snake_batch<Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], windows: &[(u32, u32)], threshold: f32,
    ) -> Result<Vec<Option<bool>>, GpuError> {
        assert_eq!(reads.len(), windows.len());
        let p = if reference.is_empty() { [0u8].as_ptr() } else { reference.as_ptr() };
        // SAFETY: reference is valid for its length
        self.check(unsafe { zsw_set_reference(self.raw, p, reference.len(), ZSW_MEM_HOST) }, 0, 0)?;
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let start: Vec<u32> = windows.iter().map(|w| w.0).collect();
        let len: Vec<u32> = windows.iter().map(|w| w.1).collect();
        let mut pass = vec![0u8; n.max(1)];
        // SAFETY: arrays hold n entries
        let code = unsafe { zsw_sneaky_snake_batch(self.raw, &batch.as_c(), start.as_ptr(), len.as_ptr(), threshold, pass.as_mut_ptr(), ptr::null_mut()) };
        self.check(code, 0, 0)?;
        Ok(pass[..n].iter().map(|&p| match p { 0 => Some(false), 1 => Some(true), _ => None }).collect())
    }

    /// `zsw_selftest`: the instruction-level assumptions of the kernels, checked on the device.
    pub fn selftest(&self) -> Result<(), GpuError> {
        // SAFETY: live context
        self.check(unsafe { zsw_selftest(self.raw) }, 0, 0)
    }

    /// Kernel time of the calls since the last read (`zsw_timing_enable` / `zsw_timing_read`).
    pub fn timing(&self, enable: bool) -> Result<(f64, u64), GpuError> {
        let (mut seconds, mut launches) = (0f64, 0u64);
        // SAFETY: valid out-pointers
        self.check(unsafe { zsw_timing_read(self.raw, &mut seconds, &mut launches) }, 0, 0)?;
        self.check(unsafe { zsw_timing_enable(self.raw, i32::from(enable)) }, 0, 0)?;
        Ok((seconds, launches))
    }

    /// Time of the seeded pass's window kernel alone since the last read (`zsw_timing_read_window`).
    pub fn timing_window(&self) -> Result<(f64, u64), GpuError> {
        let (mut seconds, mut launches) = (0f64, 0u64);
        // SAFETY: valid out-pointers
        self.check(unsafe { zsw_timing_read_window(self.raw, &mut seconds, &mut launches) }, 0, 0)?;
        Ok((seconds, launches))
    }

    /// `zsw_debug_set`: kernel-selection overrides for parity tests (results never depend on them).
    pub fn debug_set(&self, flags: u32) -> Result<(), GpuError> {
        // SAFETY: live context
        self.check(unsafe { zsw_debug_set(self.raw, flags) }, 0, 0)
    }

    /// `zsw_pack4_host` + a score call on the packed batch (`ZSW_ENCODING_PACKED4`): `reads` are `n` contiguous reads of `len` bytes;
    /// they cross PCIe as two residue indices per byte. Same results as `sw_score_from_batch`.
    pub fn sw_score_from_packed_batch<const S: usize>(
        &self, reference: &[u8], reads: &[u8], len: u32, scoring: &Scoring<'_, S>, cascade: Cascade,
    ) -> Result<(Vec<Result<MaybeAligned<u32>, ProfileError>>, Vec<u8>), GpuError> {
        self.configure(scoring, reference)?;
        let n = if len == 0 { 0 } else { reads.len() / len as usize };
        let mut packed = vec![0u8; n * ((len as usize + 1) / 2)];
        // SAFETY: both buffers are sized as the library expects
        self.check(unsafe { zsw_pack4_host(self.raw, reads.as_ptr(), n as u64, len, packed.as_mut_ptr()) }, scoring.gap_open, scoring.gap_extend)?;
        let c = ZswBatch { bases: packed.as_ptr(), offsets: ptr::null(), fixed_len: len, n_reads: n as u64, mem: ZSW_MEM_HOST, encoding: ZSW_ENCODING_PACKED4 };
        let (mut score, mut status, mut tier) = (vec![0u32; n.max(1)], vec![0u8; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: output arrays hold n entries
        self.check(unsafe { zsw_score_batch_from(self.raw, &c, cascade.from_width, cascade.preset_bits, score.as_mut_ptr(), status.as_mut_ptr(), tier.as_mut_ptr(), ptr::null_mut()) },
                   scoring.gap_open, scoring.gap_extend)?;
        tier.truncate(n);
        Ok(((0..n).map(|i| maybe(status[i], || score[i])).collect(), tier))
    }

    /// `zsw_debug_band_records`: tests only — the banded seeded pass reports, per read, the values its decision rests on
    /// (8 `i32` per read in device memory; null = off).
    ///
    /// # Safety
    /// `records` must be null or device memory for `8 * n_reads` `i32` that outlives the following calls.
    pub unsafe fn debug_band_records(&self, records: *mut i32) -> Result<(), GpuError> {
        self.check(zsw_debug_band_records(self.raw, records), 0, 0)
    }

    /// `zsw_set_option(ZSW_OPTION_EXACT_PRUNING)`: the exact column-pruned first pass (same results for every input,
    /// about three times the rate on reads that resemble the reference).
    pub fn set_exact_pruning(&self, on: bool) -> Result<(), GpuError> {
        // SAFETY: live context
        self.check(unsafe { zsw_set_option(self.raw, 1, i64::from(on)) }, 0, 0)
    }

    /// `zsw_prune_rescored`: reads of the last column-pruned score call that were rescored over all their cells.
    pub fn prune_rescored(&self) -> Result<u64, GpuError> {
        let mut n = 0u64;
        // SAFETY: live context, valid out-pointer
        self.check(unsafe { zsw_prune_rescored(self.raw, &mut n) }, 0, 0)?;
        Ok(n)
    }
}

// ---------------------------------------------------------------------------------------------
// several GPUs
// ---------------------------------------------------------------------------------------------

/// Several GPUs behind one handle: reads shard into contiguous ranges `[i*n/G, (i+1)*n/G)`, one
/// host thread of the library drives each GPU, results land in place in the caller's arrays.
pub struct GpuGroup {
    raw: *mut ZswGroup,
}

unsafe impl Send for GpuGroup {}

impl Drop for GpuGroup {
    fn drop(&mut self) {
        // SAFETY: `raw` came from zsw_group_create and is destroyed once
        unsafe { zsw_group_destroy(self.raw) }
    }
}

impl GpuGroup {
    pub fn new(device_ids: &[i32]) -> Result<Self, GpuError> {
        let mut raw: *mut ZswGroup = ptr::null_mut();
        // SAFETY: the slice is valid for its length; `raw` is a valid out-pointer
        let code = unsafe { zsw_group_create(device_ids.as_ptr(), device_ids.len() as i32, &mut raw) };
        if code != ZSW_OK {
            let message = c_message(unsafe { zsw_last_error_string(ptr::null()) });
            return Err(GpuError::Library { code, message });
        }
        Ok(GpuGroup { raw })
    }

    #[must_use]
    pub fn len(&self) -> usize {
        // SAFETY: live group
        unsafe { zsw_group_size(self.raw) }.max(0) as usize
    }

    #[must_use]
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    /// The i-th context (owned by the group), e.g. for `zsw_timing_*`.
    #[must_use]
    pub fn context_ptr(&self, i: usize) -> *mut ZswContext {
        // SAFETY: live group; out-of-range indices return NULL
        unsafe { zsw_group_context(self.raw, i as i32) }
    }

    fn check(&self, code: i32, gap_open: i8, gap_extend: i8) -> Result<(), GpuError> {
        if code == ZSW_OK {
            return Ok(());
        }
        if let Some(e) = profile_error(code, gap_open, gap_extend) {
            return Err(GpuError::Profile(e));
        }
        // SAFETY: live group
        let message = c_message(unsafe { zsw_group_last_error_string(self.raw) });
        Err(GpuError::Library { code, message })
    }

    /// Per read: `LocalProfiles::new_with_w{preset}(read, ..)?.sw_score_from_i{from_width}(reference)`, over all GPUs of the group.
    pub fn sw_score_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, cascade: Cascade,
    ) -> Result<Vec<(Result<MaybeAligned<u32>, ProfileError>, u8)>, GpuError> {
        self.configure(scoring, reference)?;
        let (go, ge) = (scoring.gap_open, scoring.gap_extend);
        let batch = HostBatch::new(reads);
        let n = reads.len();
        let (mut score, mut status, mut tier) = (vec![0u32; n.max(1)], vec![0u8; n.max(1)], vec![0u8; n.max(1)]);
        // SAFETY: output arrays hold n entries
        let code = unsafe {
            zsw_group_score_batch_from(self.raw, &batch.as_c(), cascade.from_width, cascade.preset_bits, score.as_mut_ptr(), status.as_mut_ptr(), tier.as_mut_ptr())
        };
        self.check(code, go, ge)?;
        Ok((0..n).map(|i| (maybe(status[i], || score[i]), tier[i])).collect())
    }

    fn configure<const S: usize>(&self, scoring: &Scoring<'_, S>, reference: &[u8]) -> Result<(), GpuError> {
        let w = scoring.flat_weights();
        let map = scoring.index_map();
        let (go, ge) = (scoring.gap_open, scoring.gap_extend);
        // SAFETY: pointers valid for the duration of the calls
        self.check(unsafe { zsw_group_set_scoring(self.raw, w.as_ptr(), S as i32, map.as_ptr(), i32::from(go), i32::from(ge)) }, go, ge)?;
        let p = if reference.is_empty() { [0u8].as_ptr() } else { reference.as_ptr() };
        self.check(unsafe { zsw_group_set_reference(self.raw, p, reference.len()) }, go, ge)
    }

    /// Per read: `profiles.sw_align_from_i{from_width}(seq)` (or its `_3pass` form), the batch sharded over the
    /// group's GPUs; the second vector is the width that answered.
    pub fn sw_align_from_batch<const S: usize, Q: AsRef<[u8]>>(
        &self, reads: &[Q], reference: &[u8], scoring: &Scoring<'_, S>, cascade: Cascade, other: OtherSeq, three_pass: bool,
    ) -> Result<(Vec<Result<MaybeAligned<Alignment<u32>>, ProfileError>>, Vec<u8>), GpuError> {
        self.configure(scoring, reference)?;
        let batch = HostBatch::new(reads);
        let c = batch.as_c();
        let invert = i32::from(other == OtherSeq::Query);
        let n = reads.len();
        let m = n.max(1);
        let (mut recs, mut status, mut tier) = (vec![ZswAlignment::default(); m], vec![0u8; m], vec![0u8; m]);
        let mut cap = (4 * n).max(16);
        loop {
            let (mut inc, mut op, mut needed) = (vec![0u32; cap], vec![0u8; cap], 0u64);
            // SAFETY: the arrays hold n records / cap ciglets
            let code = unsafe {
                if three_pass {
                    zsw_group_align_3pass_batch_from(self.raw, &c, cascade.from_width, cascade.preset_bits, invert, recs.as_mut_ptr(),
                        status.as_mut_ptr(), tier.as_mut_ptr(), inc.as_mut_ptr(), op.as_mut_ptr(), cap as u64, &mut needed)
                } else {
                    zsw_group_align_batch_from(self.raw, &c, cascade.from_width, cascade.preset_bits, invert, recs.as_mut_ptr(),
                        status.as_mut_ptr(), tier.as_mut_ptr(), inc.as_mut_ptr(), op.as_mut_ptr(), cap as u64, &mut needed)
                }
            };
            if code == -1 && needed as usize > cap {
                cap = needed as usize; // capacity too small: the required size came back
                continue;
            }
            self.check(code, scoring.gap_open, scoring.gap_extend)?;
            let out = (0..n).map(|i| maybe(status[i], || alignment_of(&recs[i], &inc, &op))).collect();
            return Ok((out, tier));
        }
    }

    /// Device-resident shards (one per GPU, `mem = ZSW_MEM_DEVICE`) with the results gathered on every
    /// GPU by RCCL: `out_score[i]` / `out_status[i]` are device arrays on GPU i with room for all reads.
    ///
    /// # Safety
    /// Every pointer in `shards`, `out_score` and `out_status` must be device memory of the matching GPU.
    pub unsafe fn sw_score_from_device_shards(
        &self, shards: &[ZswBatch], cascade: Cascade, out_score: &[*mut u32], out_status: &[*mut u8],
    ) -> Result<(), GpuError> {
        assert!(shards.len() == self.len() && out_score.len() == shards.len() && out_status.len() == shards.len());
        // SAFETY: upheld by the caller
        let code = unsafe {
            zsw_group_score_batch_from_device(self.raw, shards.as_ptr(), cascade.from_width, cascade.preset_bits, out_score.as_ptr(), out_status.as_ptr())
        };
        self.check(code, 0, 0)
    }
}

// ---------------------------------------------------------------------------------------------
// the library's synthetic-read generator (bench / test utilities; not part of Zoe's surface)
// ---------------------------------------------------------------------------------------------

/// `zsw_synth_reference_host`
#[must_use]
pub fn synth_reference(seed: u64, len: usize) -> Vec<u8> {
    let mut out = vec![0u8; len.max(1)];
    // SAFETY: `out` holds `len` bytes
    unsafe { zsw_synth_reference_host(seed, len as u64, out.as_mut_ptr()) };
    out.truncate(len);
    out
}

/// `zsw_synth_reads_host`: reads `[first, first + n)` of length `len`, concatenated.
#[must_use]
pub fn synth_reads(seed: u64, first: u64, n: usize, len: u32, reference: &[u8]) -> Vec<u8> {
    let mut out = vec![0u8; (n * len as usize).max(1)];
    // SAFETY: `out` holds n*len bytes, `reference` is valid for its length
    unsafe { zsw_synth_reads_host(seed, first, n as u64, len, reference.as_ptr(), reference.len() as u32, out.as_mut_ptr()) };
    out.truncate(n * len as usize);
    out
}

/// `zsw_synth_length` + `zsw_synth_reads_ragged_host`: reads of lengths uniform in `[min_len, max_len]`.
#[must_use]
pub fn synth_reads_ragged(seed: u64, first: u64, n: usize, min_len: u32, max_len: u32, reference: &[u8]) -> (Vec<u8>, Vec<u64>) {
    let mut offsets = Vec::with_capacity(n + 1);
    offsets.push(0u64);
    for i in 0..n as u64 {
        // SAFETY: pure function
        let l = unsafe { zsw_synth_length(seed, first + i, min_len, max_len) };
        offsets.push(offsets[offsets.len() - 1] + u64::from(l));
    }
    let mut out = vec![0u8; (offsets[n] as usize).max(1)];
    // SAFETY: `out` holds offsets[n] bytes
    unsafe {
        zsw_synth_reads_ragged_host(seed, first, n as u64, min_len, max_len, offsets.as_ptr(), reference.as_ptr(), reference.len() as u32, out.as_mut_ptr());
    }
    out.truncate(offsets[n] as usize);
    (out, offsets)
}

/// Device-side generators (`zsw_synth_reads`, `zsw_synth_reads_ragged`): `out_device` is device memory.
///
/// # Safety
/// `out_device` (and `offsets_device`) must be device memory of the context's GPU, large enough for the reads.
pub unsafe fn synth_reads_device(ctx: &GpuContext, seed: u64, first: u64, n: u64, len: u32, out_device: *mut u8) -> Result<(), GpuError> {
    // SAFETY: upheld by the caller
    ctx.check(unsafe { zsw_synth_reads(ctx.raw, seed, first, n, len, out_device, ptr::null_mut()) }, 0, 0)
}

/// # Safety
/// See [`synth_reads_device`].
pub unsafe fn synth_reads_ragged_device(
    ctx: &GpuContext, seed: u64, first: u64, n: u64, min_len: u32, max_len: u32, offsets_device: *const u64, out_device: *mut u8,
) -> Result<(), GpuError> {
    // SAFETY: upheld by the caller
    ctx.check(unsafe { zsw_synth_reads_ragged(ctx.raw, seed, first, n, min_len, max_len, offsets_device, out_device, ptr::null_mut()) }, 0, 0)
}
