/* zoe_sw.h — C ABI of the MI355X (gfx950) implementation of Zoe's striped Smith-Waterman hot path.
 *
 * CDCgov/zoe is a pure-Rust crate with no FFI of its own; the boundary this header replaces is the
 * set of generic Rust functions below (file:line in the reference checkout).  Each entry point is the
 * *batched* form of one of them: "for every read i: build the profile from read i, run the function
 * against the context's reference".  Role convention (src/alignment/sw/mod.rs:119-120): the read is
 * the profile sequence ("query"), the context's sequence is `reference`.
 *
 *   StripedProfile::<T,N,S>::new(seq,&matrix,go,ge)          src/alignment/profile.rs:239-247
 *   sw_simd_score::<T,N,S>(reference,&profile)               src/alignment/sw/striped.rs:65-142
 *   sw_simd_score_ends::<T,N,S>(reference,&profile)          src/alignment/sw/striped.rs:153-162
 *   sw_simd_align::<T,N,S>(reference,&profile)               src/alignment/sw/striped.rs:449-598
 *   ProfileSets::sw_score_from_{i8,i16,i32}                  src/alignment/profile_set.rs:71-107
 *   ProfileSets::sw_align_from_{i8,i16,i32}                  src/alignment/profile_set.rs:124-179
 *   LocalProfiles::new_with_w{128,256,512}                   src/alignment/profile_set.rs:434-483
 *   Nucleotides::into_local_profile (w256)                   src/data/types/nucleotides/mod.rs:262-266
 *   SeqSrc::make_alignment / Alignment::invert               src/alignment/mod.rs:176-190, types/output.rs:396-425
 *
 * Plain pointers and sizes only; no hidden global state; one context per GPU (zsw_group for several); a context may
 * be used from one host thread at a time.  Every function returns a zsw_error and never aborts.  A call leaves the
 * calling thread's current HIP device as it found it.  At most 2^31-1 reads per call.
 * Results are bit-identical to the reference's CPU path: score and status for every read; for the
 * alignment calls also ranges and CIGAR of the stated <T,N> instantiation.
 */
#ifndef ZOE_SW_H
#define ZOE_SW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Codes 1..4 are ProfileError (src/alignment/errors.rs:6-15) as raised by validate_profile_args
 * (src/alignment/profile.rs:32-44). */
typedef enum zsw_error {
    ZSW_OK = 0,
    ZSW_ERR_EMPTY_SEQUENCE = 1,
    ZSW_ERR_GAP_OPEN_OUT_OF_RANGE = 2,
    ZSW_ERR_GAP_EXTEND_OUT_OF_RANGE = 3,
    ZSW_ERR_BAD_GAP_WEIGHTS = 4,
    ZSW_ERR_INVALID_ARGUMENT = -1,
    ZSW_ERR_HIP = -2,          /* a HIP runtime call failed; see zsw_last_error_string */
    ZSW_ERR_NO_DEVICE = -3,    /* no gfx950 device / HIP runtime unusable: the product path fails loudly */
    ZSW_ERR_UNSUPPORTED = -4,  /* valid in the reference but outside what the kernels cover (documented) */
    ZSW_ERR_NOT_CONFIGURED = -5
} zsw_error;

/* MaybeAligned<T> (src/alignment/types/output.rs:18-25), per read. ZSW_STATUS_EMPTY marks a read of
 * length 0, for which StripedProfile::new returns Err(ProfileError::EmptySequence). */
typedef enum zsw_status {
    ZSW_STATUS_SOME = 0,
    ZSW_STATUS_OVERFLOWED = 1,
    ZSW_STATUS_UNMAPPED = 2,
    ZSW_STATUS_EMPTY = 3
} zsw_status;

/* T of StripedProfile<T,N,S> (AlignableIntWidth, src/math/integer.rs:231-238). Unsigned types run
 * the biased algorithm on matrix.to_biased_matrix() (src/data/matrices/mod.rs:471-491). */
typedef enum zsw_int_type { ZSW_I8 = 0, ZSW_I16 = 1, ZSW_I32 = 2, ZSW_U8 = 3, ZSW_U16 = 4, ZSW_U32 = 5 } zsw_int_type;

typedef enum zsw_mem { ZSW_MEM_HOST = 0, ZSW_MEM_DEVICE = 1 } zsw_mem;

typedef struct zsw_context zsw_context;

/* A batch of profile sequences (reads). Either fixed-length (offsets == NULL, read i occupies
 * bases[i*fixed_len, (i+1)*fixed_len)) or ragged (offsets[n_reads+1], read i = bases[offsets[i], offsets[i+1])).
 * `mem` says where bases/offsets AND the output arrays of the call live. */
/* How `bases` spells the reads. ZSW_ENCODING_BYTES (0, the default): one byte per base, as in a Nucleotides buffer.
 * ZSW_ENCODING_PACKED4: two bases per byte as residue indices of the context's ByteIndexMap (index_map[byte], < 16), the first
 * base of a pair in the low nibble; read i occupies bases[i * ((fixed_len + 1) / 2) ...]. Fixed-length host batches only: a shard
 * that crosses PCIe at half the bytes (zsw_pack4_host packs a byte batch; the library unpacks on the device). */
typedef enum zsw_encoding { ZSW_ENCODING_BYTES = 0, ZSW_ENCODING_PACKED4 = 1 } zsw_encoding;

typedef struct zsw_batch {
    const uint8_t* bases;
    const uint64_t* offsets;
    uint32_t fixed_len;
    uint64_t n_reads;
    zsw_mem mem;
    zsw_encoding encoding;
} zsw_batch;

/* Per-read alignment record: Alignment<u32> (src/alignment/types/output.rs:264-279) with
 * AlignmentStates (src/alignment/types/state.rs:53) flattened into a ciglet array. */
typedef struct zsw_alignment {
    uint32_t score;
    uint32_t ref_start, ref_end;     /* ref_range   (0-based, end-exclusive) */
    uint32_t query_start, query_end; /* query_range (excludes clipped bases) */
    uint32_t ref_len, query_len;
    uint32_t n_ciglets;              /* number of (inc, op) pairs of this read */
    uint64_t ciglet_offset;          /* index of its first pair in the ciglet arrays */
} zsw_alignment;

/* ---- context ------------------------------------------------------------------------------- */
/* Host utility: packs n_reads reads of `len` bytes each (contiguous) into ZSW_ENCODING_PACKED4 with the context's ByteIndexMap
 * (zsw_set_scoring first; alphabets of up to 16 letters); out_packed: n_reads * ((len + 1) / 2) bytes. No GPU work. */
zsw_error zsw_pack4_host(zsw_context* ctx, const uint8_t* bases, uint64_t n_reads, uint32_t len, uint8_t* out_packed);

zsw_error zsw_create(int device_id, zsw_context** out);
void zsw_destroy(zsw_context* ctx);
/* The message of the context's last failing call. ctx == NULL: why the last zsw_create of the CALLING THREAD failed
 * (thread-local; no state is shared between threads). */
const char* zsw_last_error_string(const zsw_context* ctx);
int zsw_device_count(void);

/* WeightMatrix<i8,S> + ByteIndexMap<S> + gap penalties, validated exactly like validate_profile_args
 * (codes 2..4). weights[r*S + q]: row = reference residue, column = query residue
 * (src/data/matrices/mod.rs:242-244); index_map[256] = ByteIndexMap::to_index
 * (src/data/constants/mappings/byte_index.rs:331-333); 1 <= S <= 32. */
zsw_error zsw_set_scoring(zsw_context* ctx, const int8_t* weights, int S, const uint8_t* index_map, int gap_open,
                          int gap_extend);

/* The `reference: &[u8]` argument of sw_simd_*; copied into the context (replicated per GPU).
 * zsw_set_reference and zsw_set_scoring wait for the device to finish the work already queued (asynchronous score calls on
 * any stream may still be reading the previous tables) before they overwrite them; they are configuration calls, not
 * per-batch calls. */
zsw_error zsw_set_reference(zsw_context* ctx, const uint8_t* reference, size_t len, zsw_mem mem);

/* ---- score-only ---------------------------------------------------------------------------- */
/* out_score[i], out_status[i] = StripedProfile::<int_type, lanes, S>::new(read_i).sw_score(reference).
 * `lanes` must be a power of two in 2..64 (it does not change a score; it is checked and kept for
 * signature parity). `stream` is a hipStream_t (NULL = default stream); the call is asynchronous
 * when batch.mem == ZSW_MEM_DEVICE. */
zsw_error zsw_score_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes,
                          uint32_t* out_score, uint8_t* out_status, void* stream);

/* LocalProfiles::new_with_w{preset_bits}(read_i).sw_score_from_i{from_width}(reference):
 * from_width in {8,16,32}, preset_bits in {128,256,512}. out_tier (optional) = width that answered. */
zsw_error zsw_score_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits,
                               uint32_t* out_score, uint8_t* out_status, uint8_t* out_tier, void* stream);

/* ---- score + ends -------------------------------------------------------------------------- */
/* sw_simd_score_ends: 0-based exclusive ends; first row holding the maximum, then first column. */
zsw_error zsw_score_ends_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes,
                               uint32_t* out_score, uint32_t* out_ref_end, uint32_t* out_query_end,
                               uint8_t* out_status, void* stream);

/* sw_simd_score_ranges (src/alignment/sw/striped.rs:355-388): the truncated two-pass form — forward score+ends, then
 * sw_simd_score_ends_reverse on reference[..ref_end] with the profile of reverse(read[..query_end]). Ranges are
 * 0-based half-open. */
zsw_error zsw_score_ranges_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes,
                                 uint32_t* out_score, uint32_t* out_ref_start, uint32_t* out_ref_end,
                                 uint32_t* out_query_start, uint32_t* out_query_end, uint8_t* out_status, void* stream);

/* ProfileSets::sw_score_ranges_from_i{8,16,32} (src/alignment/profile_set.rs:313-362): the cascade over the tiers of the
 * w{preset_bits} preset; out_tier (optional) = width that answered. */
zsw_error zsw_score_ranges_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits,
                                      uint32_t* out_score, uint32_t* out_ref_start, uint32_t* out_ref_end,
                                      uint32_t* out_query_start, uint32_t* out_query_end, uint8_t* out_status, uint8_t* out_tier,
                                      void* stream);

/* ---- full alignment ------------------------------------------------------------------------ */
/* out_aln[i] / out_status[i] = StripedProfile::<int_type,lanes,S>::new(read_i).sw_align(SeqSrc::Reference(reference))
 * (invert != 0: SeqSrc::Query(reference), i.e. the result passed through Alignment::invert).
 * Ciglets of all reads are packed into out_inc/out_op (capacity ciglet_cap pairs); *out_n_ciglets
 * receives the total. If the capacity is too small the call returns ZSW_ERR_INVALID_ARGUMENT and
 * *out_n_ciglets holds the required size. This call synchronises the stream. */
zsw_error zsw_align_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert,
                          zsw_alignment* out_aln, uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op,
                          uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream);

/* LocalProfiles::new_with_w{preset_bits}(read_i).sw_align_from_i{from_width}(..): each read's CIGAR is
 * that of the first tier that does not overflow (each tier has its own lane count). */
zsw_error zsw_align_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                               zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                               uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream);

/* ---- 3-pass alignment ---------------------------------------------------------------------- */
/* sw_align_3pass (src/alignment/sw/three_pass.rs:21-104; profile.rs:546-552): score ranges first, then the no-gaps shortcut,
 * a doubling banded alignment (sw/banded.rs:40-133) or the scalar alignment (sw/scalar.rs:173-271) inside the bounding box.
 * Same outputs as zsw_align_batch; like in the reference the CIGAR may differ from sw_simd_align's where several optimal
 * alignments exist (the score and validity do not). */
zsw_error zsw_align_3pass_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert,
                                zsw_alignment* out_aln, uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op,
                                uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream);
/* ProfileSets::sw_align_from_i{8,16,32}_3pass (src/alignment/profile_set.rs:212-283) */
zsw_error zsw_align_3pass_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                     zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                                     uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream);

/* ---- shared profile: one profile, many sequences ------------------------------------------- */
/* The other role of the striped functions (sw/mod.rs:63-67: "the profile can be aligned against any number of different
 * sequences"; SharedProfiles, profile_set.rs:552-560; Nucleotides::into_shared_profile, nucleotides/mod.rs:295-299): the
 * profile is built ONCE from a sequence the context holds — typically the reference — and read i is the sequence
 * sw_simd_* walks row by row. Scores are the numbers of the entry points above; ends, ranges and CIGARs are not in tie
 * cases, because the tie rule (first row, then first column) and the striping <T, N, nv = ceil(len / N)> now run over the
 * other sequence. In the results `ref_*` is the non-profile sequence, i.e. the READ, and `query_*` the profile sequence, as
 * sw_simd_* return them; invert != 0 is SeqSrc::Query(read_i) (alignment/mod.rs:176-190), which hands the roles back.
 *
 * zsw_set_profile_sequence: StripedProfile::new's sequence argument; len == 0 -> ZSW_ERR_EMPTY_SEQUENCE (profile.rs:32-44).
 * An empty READ is an empty `reference` argument here: status ZSW_STATUS_UNMAPPED (striped.rs:219-221).
 * Reads of up to 6,800 bases; the alignment calls keep plen x (longest read) flag bytes per read in flight.
 * With ZSW_OPTION_EXACT_PRUNING (the default) the score calls, and the first pass of the ends / ranges / alignment calls, take the
 * seeded exact pass with the roles swapped (an index of the profile sequence under the transposed matrix, built with the first
 * such call): a read whose maximum sits in exactly one cell of its matrix has the same ends under either tie rule; every other
 * read is computed over all its cells under this role's own rule. Same results either way. */
zsw_error zsw_set_profile_sequence(zsw_context* ctx, const uint8_t* sequence, size_t len, zsw_mem mem);
/* StripedProfile::<int_type,lanes,S>::new(sequence).sw_score(read_i) / ProfileSets::sw_score_from_i{from_width} */
zsw_error zsw_score_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, uint32_t* out_score,
                                 uint8_t* out_status, void* stream);
zsw_error zsw_score_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, uint32_t* out_score,
                                      uint8_t* out_status, uint8_t* out_tier, void* stream);
/* sw_simd_score_ends(reference = read_i, profile of the sequence): out_ref_end = exclusive end in the READ, out_query_end =
 * exclusive end in the profile sequence; first read position holding the maximum, then the first sequence position. */
zsw_error zsw_score_ends_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, uint32_t* out_score,
                                      uint32_t* out_ref_end, uint32_t* out_query_end, uint8_t* out_status, void* stream);
/* sw_simd_score_ranges / ProfileSets::sw_score_ranges_from_i{from_width}: ref range = in the read, query range = in the sequence */
zsw_error zsw_score_ranges_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, uint32_t* out_score,
                                        uint32_t* out_ref_start, uint32_t* out_ref_end, uint32_t* out_query_start, uint32_t* out_query_end,
                                        uint8_t* out_status, void* stream);
zsw_error zsw_score_ranges_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, uint32_t* out_score,
                                             uint32_t* out_ref_start, uint32_t* out_ref_end, uint32_t* out_query_start, uint32_t* out_query_end,
                                             uint8_t* out_status, uint8_t* out_tier, void* stream);
/* sw_simd_align / ProfileSets::sw_align_from_i{from_width} with the shared profile; invert = 1 is the usual call,
 * `sequence.into_shared_profile(..).sw_align_from_i8(SeqSrc::Query(read_i))`. Outputs as zsw_align_batch(_from). */
zsw_error zsw_align_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert, zsw_alignment* out_aln,
                                 uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets,
                                 void* stream);
zsw_error zsw_align_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                      zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op,
                                      uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream);
/* StripedProfile::sw_align_3pass / ProfileSets::sw_align_from_i{from_width}_3pass with the shared profile (profile.rs:536-552,
 * profile_set.rs:212-283, 552-560 -> three_pass.rs:21-104): the shared role's ranges, then sw_banded_align / sw_scalar_align over
 * the bounding box with the roles of three_pass.rs (`reference` = read i, ScalarProfile over the profile sequence's sub-range) —
 * the form the reference documents for a large reference and small queries. Like the reference's, the CIGAR may differ from
 * zsw_align_shared_batch's where several optimal alignments exist. invert = 1: SeqSrc::Query(read_i). Outputs as zsw_align_batch(_from). */
zsw_error zsw_align_3pass_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert, zsw_alignment* out_aln,
                                       uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets,
                                       void* stream);
zsw_error zsw_align_3pass_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                            zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op,
                                            uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream);

/* ---- pre-alignment filter ------------------------------------------------------------------ */
/* out_pass[i] = sneaky_snake(&reference[ref_start[i] .. ref_start[i]+ref_len[i]], read_i, threshold)
 * (src/alignment/sneaky_snake.rs:78-131): the SneakySnake edit-distance filter between a read and a candidate window of the
 * context's reference; `threshold` is the allowed edits as a fraction of the read length. Bytes are compared raw, as in the
 * reference (no index map). ref_start / ref_len / out_pass live where `reads->mem` says. A window that leaves the reference
 * is ZSW_ERR_INVALID_ARGUMENT for host arrays and ZSW_FILTER_BAD_WINDOW in out_pass for device arrays. */
typedef enum zsw_filter_result {
    ZSW_FILTER_REJECT = 0,      /* Some(false) */
    ZSW_FILTER_PASS = 1,        /* Some(true)  */
    ZSW_FILTER_NONE = 2,        /* None: threshold outside [0,1] or |len difference| > allowed edits */
    ZSW_FILTER_BAD_WINDOW = 255
} zsw_filter_result;
zsw_error zsw_sneaky_snake_batch(zsw_context* ctx, const zsw_batch* reads, const uint32_t* ref_start, const uint32_t* ref_len,
                                 float threshold, uint8_t* out_pass, void* stream);

/* ---- several GPUs behind one handle --------------------------------------------------------- */
/* The batched counterpart of scoring many reads against SharedProfiles (src/alignment/profile_set.rs:552-560) from many
 * threads: one caller, one batch, n GPUs. Reads shard into contiguous index ranges [i*n/G, (i+1)*n/G), context i takes shard i
 * on device_ids[i] from its own host thread; scoring tables and reference are replicated; nothing is exchanged while the
 * kernels run. A device may be listed more than once (two contexts on one GPU). A group is used from one host thread at a time;
 * the contexts it owns are reachable through zsw_group_context (e.g. for zsw_timing_* or to run other entry points per shard). */
typedef struct zsw_group zsw_group;
zsw_error zsw_group_create(const int* device_ids, int n_devices, zsw_group** out);
void zsw_group_destroy(zsw_group* group);
int zsw_group_size(const zsw_group* group);
zsw_context* zsw_group_context(zsw_group* group, int i);
const char* zsw_group_last_error_string(const zsw_group* group);
zsw_error zsw_group_set_scoring(zsw_group* group, const int8_t* weights, int S, const uint8_t* index_map, int gap_open,
                                int gap_extend);
zsw_error zsw_group_set_reference(zsw_group* group, const uint8_t* reference, size_t len); /* host memory */

/* ProfileSets::sw_score_from_i{from_width} (src/alignment/profile_set.rs:71-107) for every read of a batch in HOST memory,
 * spread over the group's GPUs. out_score / out_status / out_tier (optional) are host arrays of reads->n_reads entries and are
 * written in place, shard by shard: no collective is involved. Synchronous. */
zsw_error zsw_group_score_batch_from(zsw_group* group, const zsw_batch* reads, int from_width, int preset_bits,
                                     uint32_t* out_score, uint8_t* out_status, uint8_t* out_tier);

/* ProfileSets::sw_align_from_i{from_width} (src/alignment/profile_set.rs:124-179) and sw_align_from_i{from_width}_3pass
 * (:212-283) for a batch in HOST memory spread over the group's GPUs; arguments as in zsw_align_batch_from /
 * zsw_align_3pass_batch_from. The shards are aligned in parallel; records, statuses and tiers land in read order and the
 * ciglets of the shards are laid out one after the other (ciglet_offset indexes the caller's arrays). If ciglet_cap is too
 * small the call returns ZSW_ERR_INVALID_ARGUMENT with the required size in *out_n_ciglets. Synchronous. */
zsw_error zsw_group_align_batch_from(zsw_group* group, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                     zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                                     uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets);
zsw_error zsw_group_align_3pass_batch_from(zsw_group* group, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                           zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                                           uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets);

/* zsw_group_score_batch_from with reads and results in DEVICE memory: shards[i] is the batch of context i on device_ids[i]; out_score[i] and
 * out_status[i] are arrays on device i with room for ALL reads (sum of the shards' n_reads). Context i writes its slice in
 * place and one RCCL all-gather over xGMI (grouped broadcasts: the shards need not be equal) completes the arrays on every
 * device, in shard order. librccl is opened at the first call; without it the call returns ZSW_ERR_UNSUPPORTED. Synchronous. */
zsw_error zsw_group_score_batch_from_device(zsw_group* group, const zsw_batch* shards, int from_width, int preset_bits,
                                            uint32_t* const* out_score, uint8_t* const* out_status);

/* ---- bench/test utilities (not part of the reference surface) ------------------------------ */
/* Counter-based synthetic reads: read i depends only on (seed, i, reference), so any
 * shard regenerates its own slice. Writes reads [first, first+n) of length `len` into out (device). */
zsw_error zsw_synth_reads(zsw_context* ctx, uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint8_t* out_device,
                          void* stream);
/* Ragged variant: lengths uniform in [min_len, max_len]; offsets_device[n+1] must hold the exclusive
 * prefix sum of zsw_synth_length(seed, first+i, ..) (see zoe_amd/synth.py). */
zsw_error zsw_synth_reads_ragged(zsw_context* ctx, uint64_t seed, uint64_t first, uint64_t n, uint32_t min_len,
                                 uint32_t max_len, const uint64_t* offsets_device, uint8_t* out_device, void* stream);
uint32_t zsw_synth_length(uint64_t seed, uint64_t index, uint32_t min_len, uint32_t max_len);

/* Runs a small on-device check of the instruction-level assumptions the kernels make (v_perm byte
 * order, cross-lane shuffle direction, packed saturation). Returns ZSW_OK or ZSW_ERR_HIP with the
 * failing check named in zsw_last_error_string. */
zsw_error zsw_selftest(zsw_context* ctx);

/* Host twins of the generator (no GPU needed): tests and the CPU baseline consume the same bytes. */
void zsw_synth_reference_host(uint64_t seed, uint64_t len, uint8_t* out);
void zsw_synth_reads_host(uint64_t seed, uint64_t first, uint64_t n, uint32_t len, const uint8_t* ref, uint32_t R,
                          uint8_t* out);
void zsw_synth_reads_ragged_host(uint64_t seed, uint64_t first, uint64_t n, uint32_t min_len, uint32_t max_len,
                                 const uint64_t* offsets, const uint8_t* ref, uint32_t R, uint8_t* out);

/* HIP-event timing of the kernels of each *_batch call (its whole first pass for score calls), recorded on the call's stream.
 * zsw_timing_read synchronises on the recorded events, returns the summed kernel seconds and the
 * number of launches since the last read, and resets. For bench.py's roofline line. */
zsw_error zsw_timing_enable(zsw_context* ctx, int enable);
zsw_error zsw_timing_read(zsw_context* ctx, double* seconds, uint64_t* launches);
/* The same for the one kernel that dominates a score call on the default path: seed_window_kernel of the seeded exact pass
 * (events around that launch alone, on its stream; a ragged batch has one launch per length class). 0 launches if the calls
 * since the last read took another path. For bench.py's roofline of that kernel. */
zsw_error zsw_timing_read_window(zsw_context* ctx, double* seconds, uint64_t* launches);

/* Options of a context. ZSW_OPTION_EXACT_PRUNING (value 0 / 1, default 1): the seeded exact first pass (zsw_score_seed.hip;
 * DESIGN.md 4.1e). sw_simd_score returns only the maximum of the DP matrix (striped.rs:65-142), so every entry point that starts
 * with a score pass (score, ends, ranges, alignment, 3-pass alignment) first looks a few k-mers of each read up in an index of
 * the reference, computes a band of diagonals around the one they agree on, and accepts the maximum found there only if upper bounds
 * show that no alignment elsewhere can reach it; every other read is scored over all its cells. Same results for every input;
 * roughly an order of magnitude fewer cells on reads that resemble the reference, the cost of the full pass plus a few per
 * cent on reads that do not. 28 bytes + 4 bits per base of device workspace per read, up to 0.3 GB of strip-boundary buffers per
 * call, and 8 * 4^K bytes of index (K = 8 for a 2 kb reference: 512 KiB; K = 10 for 30 kb: 8 MiB). Value 0 frees the workspace and computes every cell of every read.
 * Alphabets of 8..32 letters (amino-acid matrices: a substituted residue may cost as little as 1, which leaves the k-mer argument
 * nothing to prove with) take the column-pruned pass instead (zsw_score_prune.hip): the first 24 or 48 columns of
 * a read against every row of the reference, the other columns in a window of rows around the strip's best row, and bounds that
 * add each remaining column's own largest score; reads of 65..400 residues in batches of 98,304 or more, 8 bytes per read pair
 * and reference row of workspace for up to 2 M reads at a time; the first chip-full of a large batch decides whether the rest is
 * worth it (above 70 % handed back the others go straight to the full pass).
 * Unknown options or values return ZSW_ERR_INVALID_ARGUMENT. */
typedef enum zsw_option { ZSW_OPTION_EXACT_PRUNING = 1 } zsw_option;
zsw_error zsw_set_option(zsw_context* ctx, zsw_option option, int64_t value);

/* Kernel-selection overrides for the parity tests (every path below is bit-identical to the default one; the tests
 * prove it by running both). Results never depend on these bits, only which kernel produces them. They are kept apart
 * from the options above: zsw_debug_set(ctx, 0) does not switch an option off. The library reads no environment variable. */
typedef enum zsw_debug_flag {
    ZSW_DEBUG_SCORE_V1 = 1,          /* score: Zoe's signed-offset arithmetic (score_kernel) instead of the drift-domain kernel */
    ZSW_DEBUG_NO_TILES = 2,          /* score: reads longer than the widest strip configuration go to the exact 32-bit kernel */
    ZSW_DEBUG_NO_W32 = 4,            /* score: scores beyond the packed range go to the exact 32-bit kernel, not the 32-bit tile kernel */
    ZSW_DEBUG_NO_WIDE = 8,           /* score: 8..32-letter alphabets go to the exact 32-bit kernel */
    ZSW_DEBUG_NO_SIDE_STREAMS = 16,  /* score: length classes of a ragged batch run one after the other */
    ZSW_DEBUG_NO_PIPELINE = 32,      /* score: host batches are copied whole before the kernel */
    ZSW_DEBUG_ALIGN_NO_PACKED = 64,  /* align: the 32-bit one-read-per-lane-group kernel answers every group */
    ZSW_DEBUG_SCORE_PRUNE = 128,     /* the bit ZSW_OPTION_EXACT_PRUNING holds (set by default); here for completeness */
    ZSW_DEBUG_SCORE_PRUNE_ANY_SIZE = 256, /* pruned passes for batches of every size (by default batches under 1,024 reads take the full pass) */
    /* with SCORE_PRUNE: round 2's column-pruned pass (zsw_score_prune.hip) instead of the seeded one: a narrow
     * strip of query columns against every reference row, the other columns in a window of rows around the strip's best row,
     * bound checks, the full pass for the reads that fail one. Reads of 65..400 bases, batches of 98,304 reads or more (or
     * ANY_SIZE), up to 32 GiB of workspace. A cross-check of the seeded pass for the 5-letter tables; the default first pass of
     * 8..32-letter alphabets (no flag needed there). */
    ZSW_DEBUG_PRUNE_STRIP = 512,
    /* align: the second pass starts warmup_rows before the first kept row for every read (round 2), not at the row the seeded
     * first pass certifies (zsw_seed.hpp: seed_safe_start) */
    ZSW_DEBUG_ALIGN_LONG_WARMUP = 1024,
    /* the seeded pass computes whole rows around the anchor (seed_window_kernel) instead of the band of diagonals of
     * seed_band_kernel (zsw_score_band.hip) */
    ZSW_DEBUG_SEED_NO_BAND = 2048,
    /* score: the banded kernel walks every read in the full band at once (no narrow first band for short reads) */
    ZSW_DEBUG_SEED_WIDE_BAND = 4096,
    /* score: reads the seeded pass hands back walk every row of a long reference as one item each (by default they are cut
     * into chunks of rows that run as independent items, zsw_score_v2.hpp: ScoreArgsV2::chunk_rows) */
    ZSW_DEBUG_NO_ROW_CHUNKS = 8192,
    /* score + ranges: the reverse pass of sw_simd_score_ranges by the exact prefix kernel for every read (by default a second seeded
     * pass over the reversed sequences settles the reads whose maximum sits in one cell, forward and reversed) */
    ZSW_DEBUG_RANGES_EXACT_REVERSE = 16384,
    /* align: every read with an alignment goes through the second pass (the literal striped recurrence). By default a read with
     * exactly one optimal alignment that is a gapless diagonal — both maxima in one cell each, the diagonal's weights add up to the
     * score, the score beyond what any path with an insertion and a deletion between the same corners can reach — gets that
     * alignment without it (tests/models/align_gapless_cert.cpp) */
    ZSW_DEBUG_ALIGN_NO_CERTIFICATE = 32768
} zsw_debug_flag;
zsw_error zsw_debug_set(zsw_context* ctx, uint32_t flags);

/* Tests only (tests/test_gpu_bounds.py): the banded seeded pass (zsw_score_band.hip) writes, for every read it walks, the
 * values its decision rests on into records[8 * read]: [0] the band's maximum (doubled; odd = held by a path through a cell
 * outside the band), [1] / [2] the most a path that ends above / below the band can score, [3] / [4] the smaller / larger anchor
 * diagonal of the lane's two reads, [5] strips | rows above the anchor << 8 | below << 20, [6] the read's own anchor diagonal,
 * [7] 1 = accepted by this walk | columns per strip << 8. A later tier overwrites an earlier one. The host model (tests/models/seed_band.cpp) computes
 * the same quantities from the read and this geometry; the test requires equality. records: device memory for 8 int32 per read
 * of the following calls, NULL = off (the default). */
zsw_error zsw_debug_band_records(zsw_context* ctx, int32_t* records);

/* Reads of the context's last score call that the seeded (or column-pruned) pass handed back — no anchor, or a bound check
 * failed — and that were scored over all their cells (0 if the call did not take such a pass). Synchronises the device.
 * Diagnostics for tests and bench.py. */
zsw_error zsw_prune_rescored(zsw_context* ctx, uint64_t* out_reads);

#ifdef __cplusplus
}
#endif
#endif /* ZOE_SW_H */
