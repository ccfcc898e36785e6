// zsw_context.hpp — the context behind the C ABI and the helpers its translation units share (zsw_capi.hip: the entry points
// with the read as the profile; zsw_capi_shared.hip: the one-profile-many-sequences role). Not installed.
#pragma once
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "zsw_align.hpp"
#include "zsw_internal.hpp"
#include "zsw_score_prune.hpp"
#include "zsw_score_seed.hpp"
#include "zsw_shared.hpp"
#include "zsw_timer.hpp"

namespace zsw {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

}  // namespace zsw

struct zsw_context {
    int device = 0;
    uint32_t cu_count = 256;
    bool scoring_set = false, reference_set = false;
    zsw::ScoringDev h_sc{};
    int bias = 0;
    zsw::DevBuf d_sc, d_ref, d_fb_list, d_fb_count, d_scratch, d_maxlen, d_bucket_items, d_bucket_counts, d_tile_buf, d_tile_state;
    zsw::DevBuf d_prune, d_prune_list, d_prune_count;  // column-pruned score pass; the worklist and its counters also serve the seeded pass
    uint32_t prune_chunk = 0;
    // seeded exact score pass (zsw_score_seed.hip): index of the reference under the current matrix (built with the first batch
    // that can use it, rebuilt after zsw_set_scoring / zsw_set_reference), a host copy of the reference to build it from
    zsw::SeedIndex seed;
    // the same for the REVERSED reference (d_ref_rev): the reverse pass of sw_simd_score_ranges as a second seeded pass over the
    // reversed reads (zsw_capi.hip, ranges_device); built with the first such call
    zsw::SeedIndex seed_rev;
    zsw::DevBuf d_ref_rev;
    std::vector<uint8_t> h_ref;
    zsw::DevBuf d_seed_work, d_seed_gtab, d_chunk_keys;
    bool seed_ready = false;  // this call's batch takes the seeded pass (workspace and worklist are in place)
    bool chunk_ready = false;  // ... and d_chunk_keys holds a key per read (long references: the hand-back pass runs in chunks of rows)
    size_t ref_len = 0;
    uint32_t scratch_len = 0;
    size_t exact_slots = 0;
    // staging for host-memory batches
    zsw::DevBuf s_bases, s_packed, s_offsets, s_score, s_status, s_tier, s_rend, s_qend;
    // alignment workspace (zsw_align.hip)
    zsw::DevBuf a_ws[30];
    // score_ranges workspace
    zsw::DevBuf r_ws[24];
    zsw::KernelTimer timer;
    zsw::KernelTimer timer_window;  // seed_window_kernel launches alone
    std::string err;
    uint32_t debug = 0;    // zsw_debug_set (kernel-selection overrides for tests)
    int32_t* band_dbg = nullptr;  // zsw_debug_band_records (tests: the banded seeded pass reports the values it decides with)
    uint32_t options = ZSW_DEBUG_SCORE_PRUNE;  // zsw_set_option, as ZSW_DEBUG_* bits; exact pruning is on by default
    uint32_t flags() const { return debug | options; }
    // host batches: reads of chunk k+1 cross PCIe on this stream while chunk k computes
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> copy_events;
    // ragged batches: the length classes run on these (created with the first ragged batch)
    zsw::SideStreams* side = nullptr;
    // the one-profile-many-sequences role (zsw_capi_shared.hip): the sequence the shared profile is built from, the scoring with
    // the matrix transposed (score-only calls go through the ordinary kernels with the roles swapped), workspace
    zsw::DevBuf d_pseq, d_pseq_rev, d_sc_t, sh_ws[20];
    std::vector<uint8_t> h_pseq;
    size_t pseq_len = 0;
    bool pseq_set = false;
    bool shared_call = false;  // stage(): the call scores against d_pseq with the transposed matrix: its seeded pass uses seed_shared
    // index of the profile sequence under the transposed matrix: the seeded pass of the shared role's score calls (roles swapped:
    // the sequence is the ordinary kernels' reference); rebuilt after zsw_set_scoring / zsw_set_profile_sequence
    zsw::SeedIndex seed_shared;
    // the same for the REVERSED profile sequence (d_pseq_rev): the second pass of the shared role's sw_simd_score_ranges as a
    // seeded pass over the reversed reads (zsw_capi_shared.hip, run_ranges_shared); built with the first such call
    zsw::SeedIndex seed_shared_rev;
    bool shared_seedable = false;  // set by the shared entry points whose kernels can take the seeded pass (score; ends with MODE 3)
};

namespace zsw {
namespace capi {

constexpr size_t EXACT_SLOTS = 64 * 256;               // rows of the exact 32-bit kernel that run at once, at most
constexpr size_t EXACT_SCRATCH_BUDGET = size_t(1) << 30;  // bytes of its H/E rows, at most (longer reads get fewer slots)
constexpr uint32_t LONGEST_STRIP = 64 * 38;  // columns of the widest strip configuration (zsw_score_v2.hpp)

inline zsw_error fail(zsw_context* ctx, zsw_error code, const char* what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) {
            ctx->err += ": ";
            ctx->err += hipGetErrorString(e);
        }
    }
    return code;
}

// Makes the context's GPU current for the duration of a public call and puts the caller's device back afterwards: a
// single-process multi-GPU host (zsw_group, or PyTorch with several devices) must not find its current device changed.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const zsw_context* ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != ctx->device) (void)hipSetDevice(ctx->device);
        else prev = -1;  // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define ZSW_HIP(ctx, call)                                                     \
    do {                                                                       \
        hipError_t _e = (call);                                                \
        if (_e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, #call, _e);        \
    } while (0)

inline bool valid_lanes(int lanes) { return lanes == 2 || lanes == 4 || lanes == 8 || lanes == 16 || lanes == 32 || lanes == 64; }

inline uint64_t signed_thr(int bits) { return bits == 8 ? 255ull : bits == 16 ? 65535ull : 4294967295ull; }

// score_to_maybe_aligned (striped.rs:610-633) as a threshold on the true score:
//   signed T  : Overflowed  <=>  best >= T::MAX            <=>  s >= 2^bits - 1
//   unsigned T: Overflowed  <=>  best + bias + 1 > T::MAX  <=>  s >= T::MAX - bias
inline bool rule_direct(zsw_int_type t, int bias, ResultRule* r) {
    r->n_tiers = 1;
    switch (t) {
        case ZSW_I8: r->thr[0] = signed_thr(8); r->tier_code[0] = 8; return true;
        case ZSW_I16: r->thr[0] = signed_thr(16); r->tier_code[0] = 16; return true;
        case ZSW_I32: r->thr[0] = signed_thr(32); r->tier_code[0] = 32; return true;
        case ZSW_U8: r->thr[0] = 255ull - (uint64_t)bias; r->tier_code[0] = 8; return bias < 255;
        case ZSW_U16: r->thr[0] = 65535ull - (uint64_t)bias; r->tier_code[0] = 16; return true;
        case ZSW_U32: r->thr[0] = 4294967295ull - (uint64_t)bias; r->tier_code[0] = 32; return true;
    }
    return false;
}

// or_else_overflowed chain (profile_set.rs:71-107): i8 -> i16 -> i32 from `from_width`
inline bool rule_cascade(int from_width, ResultRule* r) {
    if (from_width != 8 && from_width != 16 && from_width != 32) return false;
    r->n_tiers = 0;
    for (int w = from_width; w <= 32; w *= 2) {
        r->thr[r->n_tiers] = signed_thr(w);
        r->tier_code[r->n_tiers] = (uint8_t)w;
        ++r->n_tiers;
    }
    return true;
}

struct Staged {
    BatchDev b{};
    uint32_t max_len = 0;
    uint32_t* d_score = nullptr;
    uint8_t* d_status = nullptr;
    uint8_t* d_tier = nullptr;
    uint32_t* d_rend = nullptr;
    uint32_t* d_qend = nullptr;
};

// zsw_capi.hip
zsw_error stage(zsw_context* ctx, const zsw_batch* reads, hipStream_t stream, bool want_tier, bool want_ends, uint32_t* out_score,
                uint8_t* out_status, uint8_t* out_tier, uint32_t* out_rend, uint32_t* out_qend, Staged* st, bool defer_bases_copy = false);
ScoreWorkspace score_ws(zsw_context* ctx);
zsw_error unstage(zsw_context* ctx, const zsw_batch* reads, hipStream_t stream, const Staged& st, uint32_t* out_score, uint8_t* out_status,
                  uint8_t* out_tier, uint32_t* out_rend, uint32_t* out_qend);
enum { WS_SCORE = 0, WS_STATUS, WS_TIER, WS_REND, WS_ITEMS, WS_RING, WS_CIG, WS_ALN, WS_CIGSTART, WS_CIGRAW, WS_BSUMS, WS_TOTAL,
       WS_FBLIST, WS_FBCOUNT, WS_OINC, WS_OOP, WS_CIG2, WS_RING2, WS_KEYS_IN, WS_KEYS_OUT, WS_VALS_IN, WS_SORT_TMP, WS_GTABLE, WS_FBMETA,
       WS_ITEMS2, WS_SAFE, WS_CERT_OK, WS_CERT_DONE, WS_CERT_STATUS };
enum { RW_FSCORE = 0, RW_FSTATUS, RW_FREND, RW_FQEND, RW_RSCORE, RW_RSTATUS, RW_RRS, RW_RQS, RW_QEM, RW_GTAB, RW_MIS, RW_O0, RW_O1,
       RW_O2, RW_O3, RW_O4, RW_O5, RW_FTIER, RW_UNIQ_F, RW_UNIQ_R, RW_RBASES, RW_ULIST, RW_UCOUNT };
zsw_error finish_alignments(zsw_context* ctx, DevBuf* ws, uint32_t n, bool host, const uint8_t* d_status, const uint8_t* d_tier, int invert,
                            zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op,
                            uint64_t ciglet_cap, uint64_t* out_n_ciglets, hipStream_t stream);
struct RangesDev {  // device arrays of sw_simd_score_ranges for every read (library workspace)
    uint32_t *score, *rs, *re, *qs, *qe;
    uint8_t *status, *tier;
};
// sw_align_3pass's third pass (zsw_threepass.hip) over ranges that are already on the device, results to the caller's arrays; the
// caller has opened ctx->timer's interval. pseq: non-null = the shared-profile role (ThreePassArgs::pseq).
zsw_error threepass_third_pass(zsw_context* ctx, const Staged& st, const RangesDev& rd, const uint8_t* pseq, uint32_t pseq_len, bool host, int invert,
                               zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap,
                               uint64_t* out_n_ciglets, hipStream_t stream);
// statuses as a literal second pass sees them: reads whose alignment a certificate pass wrote (done[i]) do not take part
hipError_t launch_cert_status(uint32_t n, const uint8_t* status, const uint8_t* done, uint8_t* out, hipStream_t stream);
// zsw_capi_shared.hip: the settlement of a reversed seeded pass (a read is
// done if both maxima sit in one cell each and the scores agree; the others are listed for the exact reverse kernel)
hipError_t launch_settle_reverse(const BatchDev& b, uint32_t n, uint32_t other_len, const uint8_t* uf, const uint8_t* ur, const uint32_t* fscore,
                                 const uint8_t* fstatus, const uint32_t* rscore, const uint8_t* rstatus, uint32_t* read_side, uint32_t* other_side,
                                 uint32_t* list, uint32_t* count, hipStream_t stream, uint8_t* settled = nullptr);  // settled[i] = 1: read i is done
// the and_then / map chain of sw_simd_score_ranges on device arrays (kernels of zsw_capi.hip)
hipError_t launch_ranges_prep(uint32_t n, const uint8_t* fstatus, const uint32_t* fqend, uint32_t* qe_masked, hipStream_t stream);
hipError_t launch_ranges_combine(uint32_t n, const uint32_t* fscore, const uint8_t* fstatus, const uint32_t* frend, const uint32_t* fqend,
                                 const uint32_t* rscore, const uint8_t* rstatus, const uint32_t* rrstart, const uint32_t* rqstart,
                                 uint32_t* out_score, uint32_t* out_rs, uint32_t* out_re, uint32_t* out_qs, uint32_t* out_qe,
                                 uint8_t* out_status, uint32_t* mismatch, hipStream_t stream);

}  // namespace capi
}  // namespace zsw
