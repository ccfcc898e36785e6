// zsw_capi_shared.hip — C ABI of the one-profile-many-sequences role (include/zoe_sw.h, "shared profile" section):
// zsw_set_profile_sequence and the *_shared_batch entry points. The profile is built from the sequence the context holds
// (Nucleotides::into_shared_profile, nucleotides/mod.rs:295-299; SharedProfiles, profile_set.rs:552-560) and read i is the
// sequence it is aligned against (sw/mod.rs:63-67), row by row. Kernels: zsw_shared.hip (score + ends, ranges),
// align_kernel<N, ., SHARED> (zsw_align.hip); the score-only calls swap the roles and use the ordinary kernels with the
// transposed matrix (the score of a pair does not depend on which sequence the profile is built from).
#include <map>
#include <utility>

#include "zsw_context.hpp"

using namespace zsw;
using namespace zsw::capi;

namespace {

enum { SH_SCORE = 0, SH_STATUS, SH_TIER, SH_REND, SH_QEND, SH_QEM, SH_RSCORE, SH_RSTATUS, SH_RRS, SH_RQS, SH_MIS, SH_LIST, SH_UNIQUE, SH_ULIST, SH_UCOUNT,
       SH_UNIQUE_R, SH_RBASES };

__global__ void empty_is_unmapped_kernel(uint32_t n, uint8_t* status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // an empty read is an empty `reference` here: sw_simd_score returns Unmapped (striped.rs:219-221), not a profile error
    if (i < n && status[i] == ZSW_STATUS_EMPTY) status[i] = ZSW_STATUS_UNMAPPED;
}

__global__ void iota_some_kernel(uint32_t n, const uint8_t* status, const uint8_t* tier, uint8_t want_tier, uint32_t* list, uint32_t* count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool take = i < n && status[i] == ZSW_STATUS_SOME && tier[i] == want_tier;
    const unsigned long long m = __ballot(take);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (take) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
    }
}

// reads whose ends the role-swapped seeded pass could not settle (unique[i] == 0): the list of the exact shared-role kernel
__global__ void select_not_unique_kernel(uint32_t n, const uint8_t* unique, uint32_t* list, uint32_t* count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool take = i < n && unique[i] == 0;
    const unsigned long long m = __ballot(take);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (take) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
    }
}

// every read of the batch reversed (same offsets): the other sequence of the reverse pass
// Second pass of sw_simd_score_ranges in the shared role, settled by the seeded pass over the REVERSED sequences: read i is done
// if its forward maximum sits in one cell (uf), the reversed problem's maximum sits in one cell (ur) and the two scores agree —
// then the starts are that cell's coordinates turned round (rs: position in rev(read) + 1 -> inclusive start in the read, qs
// likewise in the profile sequence). Everything else joins the list of the exact reverse kernel.
__global__ void settle_reverse_kernel(BatchDev b, uint32_t n, uint32_t plen, const uint8_t* uf, const uint8_t* ur, const uint32_t* fscore,
                                      const uint8_t* fstatus, const uint32_t* rscore, const uint8_t* rstatus, uint32_t* rs, uint32_t* qs,
                                      uint32_t* list, uint32_t* count, uint8_t* settled_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool settled = false;
    if (i < n && uf[i] && ur[i] && fstatus[i] == ZSW_STATUS_SOME && rstatus[i] == ZSW_STATUS_SOME && rscore[i] == fscore[i]) {
        const uint32_t len = b.offsets ? (uint32_t)(b.offsets[i + 1] - b.offsets[i]) : b.fixed_len;
        if (rs[i] >= 1 && rs[i] <= len && qs[i] >= 1 && qs[i] <= plen) {
            rs[i] = len - rs[i];
            qs[i] = plen - qs[i];
            settled = true;
        }
    }
    if (settled_out && i < n) settled_out[i] = settled ? 1 : 0;
    // one atomic per block of 1,024 reads (one per wavefront was 156,000 atomics on one word for 10 M reads: 1.6 ms)
    __shared__ uint32_t wave_base[16], block_base;
    const bool take = i < n && !settled;
    const unsigned long long m = __ballot(take);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_base[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            const uint32_t c = wave_base[w];
            wave_base[w] = run;
            run += c;
        }
        block_base = run ? atomicAdd(count, run) : 0u;
    }
    __syncthreads();
    if (take) list[block_base + wave_base[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

}  // namespace

namespace zsw {
namespace capi {
// (also used by the read-as-profile role's ranges, zsw_capi.hip: its reverse pass as a second seeded pass)
// read_side / other_side: the reversed pass's ends in the reversed read / in the reversed other sequence (of other_len residues); on
// return the inclusive starts in the sequences themselves for the settled reads
hipError_t launch_settle_reverse(const BatchDev& b, uint32_t n, uint32_t other_len, const uint8_t* uf, const uint8_t* ur, const uint32_t* fscore,
                                 const uint8_t* fstatus, const uint32_t* rscore, const uint8_t* rstatus, uint32_t* read_side, uint32_t* other_side,
                                 uint32_t* list, uint32_t* count, hipStream_t stream, uint8_t* settled) {
    if (n) hipLaunchKernelGGL(settle_reverse_kernel, dim3((n + 1023) / 1024), dim3(1024), 0, stream, b, n, other_len, uf, ur, fscore, fstatus, rscore, rstatus,
                              read_side, other_side, list, count, settled);
    return hipGetLastError();
}
}  // namespace capi
}  // namespace zsw

namespace {

zsw_error check_shared(zsw_context* ctx) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!ctx->scoring_set) return fail(ctx, ZSW_ERR_NOT_CONFIGURED, "scoring not set");
    if (!ctx->pseq_set) return fail(ctx, ZSW_ERR_NOT_CONFIGURED, "profile sequence not set (zsw_set_profile_sequence)");
    return ZSW_OK;
}

// stage() wants a reference: the shared calls have none of their own, the profile sequence stands in. `seedable`: the call's
// kernels can take the seeded pass (with the index of the profile sequence under the transposed matrix, ctx->seed_shared).
struct SharedStage {
    zsw_context* ctx;
    bool ref_was_set;
    size_t ref_len;
    explicit SharedStage(zsw_context* c, bool seedable = false) : ctx(c), ref_was_set(c->reference_set), ref_len(c->ref_len) {
        ctx->shared_call = true;
        ctx->shared_seedable = seedable;
        ctx->reference_set = true;
        ctx->ref_len = ctx->pseq_len;
    }
    ~SharedStage() {
        ctx->shared_call = false;
        ctx->shared_seedable = false;
        ctx->reference_set = ref_was_set;
        ctx->ref_len = ref_len;
    }
};

// score + ends of every read against the shared profile, on the device.
// With the seeded pass staged (SharedStage(ctx, true) + stage()): the roles are swapped — the read is the profile of the ordinary
// kernels, the shared sequence their reference, the matrix transposed — and the banded seeded pass runs at mode 3: score, both
// ends under ITS tie rule (first row of the sequence, then first column of the read) and whether the maximum sits in exactly one
// cell of the matrix. For such a read the tie rule does not matter and the ends are the shared role's with the names swapped
// (ref_end = row of the read, query_end = column of the profile sequence). Every other read — handed back by the seeded pass,
// ties, no alignment — is computed by shared_ends_kernel, which walks all cells under the shared role's own rule.
zsw_error shared_ends_device(zsw_context* ctx, const Staged& st, const ResultRule& rule, const ScoreOut& out, hipStream_t stream,
                             bool* seeded = nullptr /* out: the seeded route ran and sh_ws[SH_UNIQUE] holds its flags */) {
    if (seeded) *seeded = false;
    if (st.max_len > shared_max_rows()) return fail(ctx, ZSW_ERR_UNSUPPORTED, "read too long for the shared-profile kernels");
    const uint32_t n = st.b.n_items;
    BatchDev rest = st.b;
    const uint32_t* rest_count = nullptr;
    if (ctx->seed_ready && n > 0 && !st.b.items) {
        DevBuf* ws = ctx->sh_ws;
        ZSW_HIP(ctx, ws[SH_UNIQUE].ensure((size_t)n + 4));
        ZSW_HIP(ctx, ws[SH_ULIST].ensure((size_t)n * 4 + 4));
        ZSW_HIP(ctx, ws[SH_UCOUNT].ensure(16));
        ZSW_HIP(ctx, hipMemsetAsync(ws[SH_UNIQUE].p, 0, n, stream));
        ZSW_HIP(ctx, hipMemsetAsync(ws[SH_UCOUNT].p, 0, 4, stream));
        ScoringDev h_t = ctx->h_sc;
        for (int r = 0; r < h_t.S; ++r)
            for (int q = 0; q < h_t.S; ++q) h_t.w[r * h_t.S + q] = ctx->h_sc.w[q * h_t.S + r];
        ScoreOut o2 = out;
        o2.ref_end = out.query_end;   // rows of the swapped problem are positions of the profile sequence
        o2.query_end = out.ref_end;   // its columns positions of the read
        o2.fb_list = ctx->d_fb_list.as<uint32_t>();
        o2.fb_count = ctx->d_fb_count.as<uint32_t>();
        o2.unique = ws[SH_UNIQUE].as<uint8_t>();
        o2.skip_handed_back = true;
        hipError_t e = launch_score(ctx->d_sc_t.as<ScoringDev>(), h_t, st.b, st.max_len, ctx->d_pseq.as<uint8_t>(), (uint32_t)ctx->pseq_len, rule, o2,
                                    score_ws(ctx), stream, nullptr, 3);
        if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared ends: role-swapped seeded pass", e);
        hipLaunchKernelGGL(select_not_unique_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, ws[SH_UNIQUE].as<uint8_t>(),
                           ws[SH_ULIST].as<uint32_t>(), ws[SH_UCOUNT].as<uint32_t>());
        rest.items = ws[SH_ULIST].as<uint32_t>();
        rest_count = ws[SH_UCOUNT].as<uint32_t>();
        if (seeded) *seeded = true;
    }
    hipError_t e = launch_shared_ends(rest, std::max<uint32_t>(st.max_len, 1), ctx->d_pseq.as<uint8_t>(), (uint32_t)ctx->pseq_len,
                                      ctx->d_sc.as<ScoringDev>(), rule, out, nullptr, nullptr, stream, rest_count);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared ends kernel", e);
    return ZSW_OK;
}

zsw_error run_score_shared(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, uint32_t* out_score, uint8_t* out_status,
                           uint8_t* out_tier, void* stream_) {
    zsw_error ze = check_shared(ctx);
    if (ze != ZSW_OK) return ze;
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    SharedStage guard(ctx, true);  // the score does not depend on the roles: the seeded pass applies with the sequence as its reference
    Staged st;
    ze = stage(ctx, reads, stream, out_tier != nullptr, false, out_score, out_status, out_tier, nullptr, nullptr, &st);
    if (ze != ZSW_OK) return ze;
    if (reads->n_reads == 0) return ZSW_OK;
    // roles swapped: the read is the profile of the ordinary kernels, the shared sequence their reference, the matrix transposed
    ScoringDev h_t = ctx->h_sc;
    for (int r = 0; r < h_t.S; ++r)
        for (int q = 0; q < h_t.S; ++q) h_t.w[r * h_t.S + q] = ctx->h_sc.w[q * h_t.S + r];
    ScoreOut out;
    out.score = st.d_score;
    out.status = st.d_status;
    out.tier = st.d_tier;
    out.ref_end = nullptr;
    out.query_end = nullptr;
    out.fb_list = ctx->d_fb_list.as<uint32_t>();
    out.fb_count = ctx->d_fb_count.as<uint32_t>();
    hipError_t e = launch_score(ctx->d_sc_t.as<ScoringDev>(), h_t, st.b, st.max_len, ctx->d_pseq.as<uint8_t>(), (uint32_t)ctx->pseq_len, rule, out,
                                score_ws(ctx), stream, &ctx->timer, 0);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared score launch", e);
    const uint32_t n = (uint32_t)reads->n_reads;
    hipLaunchKernelGGL(empty_is_unmapped_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, st.d_status);
    return unstage(ctx, reads, stream, st, out_score, out_status, out_tier, nullptr, nullptr);
}

zsw_error run_ends_shared(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, uint32_t* out_score, uint32_t* out_rend,
                          uint32_t* out_qend, uint8_t* out_status, void* stream_) {
    zsw_error ze = check_shared(ctx);
    if (ze != ZSW_OK) return ze;
    if (!out_rend || !out_qend) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    SharedStage guard(ctx, true);
    Staged st;
    ze = stage(ctx, reads, stream, false, true, out_score, out_status, nullptr, out_rend, out_qend, &st);
    if (ze != ZSW_OK) return ze;
    if (reads->n_reads == 0) return ZSW_OK;
    ScoreOut out;
    out.score = st.d_score;
    out.status = st.d_status;
    out.tier = nullptr;
    out.ref_end = st.d_rend;
    out.query_end = st.d_qend;
    out.fb_list = nullptr;
    out.fb_count = nullptr;
    ctx->timer.begin(stream);
    ze = shared_ends_device(ctx, st, rule, out, stream);
    ctx->timer.end(stream);
    if (ze != ZSW_OK) return ze;
    return unstage(ctx, reads, stream, st, out_score, out_status, nullptr, out_rend, out_qend);
}

// sw_simd_score_ranges of the shared role for every read, on the device (rs / re: positions in the read, qs / qe: in the profile
// sequence); opens ctx->timer's interval and leaves it open.
zsw_error ranges_shared_device(zsw_context* ctx, const Staged& st, const ResultRule& rule, hipStream_t stream, RangesDev* out, uint8_t* settled = nullptr);

zsw_error run_ranges_shared(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, uint32_t* out_score, uint32_t* out_rs,
                            uint32_t* out_re, uint32_t* out_qs, uint32_t* out_qe, uint8_t* out_status, uint8_t* out_tier, void* stream_) {
    zsw_error ze = check_shared(ctx);
    if (ze != ZSW_OK) return ze;
    if (!reads || !out_score || !out_rs || !out_re || !out_qs || !out_qe || !out_status) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    SharedStage guard(ctx, true);
    Staged st;
    {
        uint32_t dummy_score = 0;
        uint8_t dummy_status = 0;
        ze = stage(ctx, reads, stream, false, false, &dummy_score, &dummy_status, nullptr, nullptr, nullptr, &st);
        if (ze != ZSW_OK) return ze;
    }
    const uint32_t n = (uint32_t)reads->n_reads;
    if (n == 0) return ZSW_OK;
    RangesDev rd;
    ze = ranges_shared_device(ctx, st, rule, stream, &rd);
    if (ze != ZSW_OK) return ze;
    ctx->timer.end(stream);
    const hipMemcpyKind kind = reads->mem == ZSW_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    uint32_t* outs[5] = {out_score, out_rs, out_re, out_qs, out_qe};
    const uint32_t* devs[5] = {rd.score, rd.rs, rd.re, rd.qs, rd.qe};
    for (int k = 0; k < 5; ++k) ZSW_HIP(ctx, hipMemcpyAsync(outs[k], devs[k], (size_t)n * 4, kind, stream));
    ZSW_HIP(ctx, hipMemcpyAsync(out_status, rd.status, n, kind, stream));
    if (out_tier) ZSW_HIP(ctx, hipMemcpyAsync(out_tier, rd.tier, n, kind, stream));
    if (reads->mem == ZSW_MEM_HOST) ZSW_HIP(ctx, hipStreamSynchronize(stream));
    return ZSW_OK;
}

// settled (optional, n bytes): 1 where both maxima of the read sit in one cell each (the two seeded passes agree), else 0
zsw_error ranges_shared_device(zsw_context* ctx, const Staged& st, const ResultRule& rule, hipStream_t stream, RangesDev* out, uint8_t* settled) {
    zsw_error ze = ZSW_OK;
    const uint32_t n = st.b.n_reads;
    DevBuf* ws = ctx->sh_ws;
    DevBuf* rw = ctx->r_ws;  // the output arrays of the ordinary ranges path
    for (int k : {SH_SCORE, SH_REND, SH_QEND, SH_QEM, SH_RSCORE, SH_RRS, SH_RQS}) ZSW_HIP(ctx, ws[k].ensure((size_t)n * 4 + 4));
    for (int k : {SH_STATUS, SH_TIER, SH_RSTATUS}) ZSW_HIP(ctx, ws[k].ensure((size_t)n + 4));
    ZSW_HIP(ctx, ws[SH_MIS].ensure(4));
    for (int k : {RW_O0, RW_O1, RW_O2, RW_O3, RW_O4}) ZSW_HIP(ctx, rw[k].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, rw[RW_O5].ensure((size_t)n + 4));
    ScoreOut fo;
    fo.score = ws[SH_SCORE].as<uint32_t>();
    fo.status = ws[SH_STATUS].as<uint8_t>();
    fo.tier = ws[SH_TIER].as<uint8_t>();
    fo.ref_end = ws[SH_REND].as<uint32_t>();
    fo.query_end = ws[SH_QEND].as<uint32_t>();
    fo.fb_list = nullptr;
    fo.fb_count = nullptr;
    ctx->timer.begin(stream);
    bool seeded = false;
    ze = shared_ends_device(ctx, st, rule, fo, stream, &seeded);
    if (ze != ZSW_OK) return ze;
    // reverse pass on the prefixes the forward pass found (reads without an alignment take part with empty prefixes)
    ZSW_HIP(ctx, launch_ranges_prep(n, fo.status, fo.query_end, ws[SH_QEM].as<uint32_t>(), stream));
    ScoreOut ro;
    ro.score = ws[SH_RSCORE].as<uint32_t>();
    ro.status = ws[SH_RSTATUS].as<uint8_t>();
    ro.tier = nullptr;
    ro.ref_end = ws[SH_RRS].as<uint32_t>();
    ro.query_end = ws[SH_RQS].as<uint32_t>();
    ro.fb_list = nullptr;
    ro.fb_count = nullptr;
    BatchDev rest = st.b;
    const uint32_t* rest_count = nullptr;
    if (settled) ZSW_HIP(ctx, hipMemsetAsync(settled, 0, n, stream));
    if (seeded) {
        // The reverse pass as a seeded pass of its own: reversed reads against the reversed profile sequence, whole sequences. If
        // the forward maximum sits in one cell, every alignment that scores it ends there, i.e. lies inside the prefixes the
        // reverse pass of striped.rs:355-388 is restricted to; the cells of the reversed matrix that hold the score are then the
        // same with or without the restriction, and if that is one cell too, it is the start under any tie rule.
        const size_t plen = ctx->pseq_len;
        ScoringDev h_t = ctx->h_sc;
        for (int r = 0; r < h_t.S; ++r)
            for (int q = 0; q < h_t.S; ++q) h_t.w[r * h_t.S + q] = ctx->h_sc.w[q * h_t.S + r];
        if (!ctx->seed_shared_rev.valid) {
            std::vector<uint8_t> rev(ctx->h_pseq.rbegin(), ctx->h_pseq.rend());
            ZSW_HIP(ctx, ctx->d_pseq_rev.ensure(plen + 16));
            ZSW_HIP(ctx, hipMemcpyAsync(ctx->d_pseq_rev.p, rev.data(), plen, hipMemcpyHostToDevice, stream));
            ZSW_HIP(ctx, hipStreamSynchronize(stream));  // `rev` goes out of scope
            ZSW_HIP(ctx, seed_index_update(&ctx->seed_shared_rev, h_t, rev.data(), plen));
        }
        if (ctx->seed_shared_rev.usable) {
            ZSW_HIP(ctx, ws[SH_UNIQUE_R].ensure((size_t)n + 4));
            ZSW_HIP(ctx, hipMemsetAsync(ws[SH_UNIQUE_R].p, 0, n, stream));
            BatchDev brev = st.b;  // the same bases: the seeded pass reads them back to front (ScoreOut::reads_reversed)
            ScoreOut o3 = ro;
            o3.ref_end = ro.query_end;   // rows of the swapped problem: positions of the reversed profile sequence
            o3.query_end = ro.ref_end;   // columns: positions of the reversed read
            o3.fb_list = ctx->d_fb_list.as<uint32_t>();
            o3.fb_count = ctx->d_fb_count.as<uint32_t>();
            o3.unique = ws[SH_UNIQUE_R].as<uint8_t>();
            o3.skip_handed_back = true;
            o3.reads_reversed = true;
            ScoreWorkspace w = score_ws(ctx);
            w.seed = &ctx->seed_shared_rev;
            hipError_t e3 = launch_score(ctx->d_sc_t.as<ScoringDev>(), h_t, brev, st.max_len, ctx->d_pseq_rev.as<uint8_t>(), (uint32_t)plen, rule, o3, w,
                                         stream, nullptr, 3);
            if (e3 != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared ranges: seeded reverse pass", e3);
            ZSW_HIP(ctx, hipMemsetAsync(ws[SH_UCOUNT].p, 0, 4, stream));
            hipLaunchKernelGGL(settle_reverse_kernel, dim3((n + 1023) / 1024), dim3(1024), 0, stream, st.b, n, (uint32_t)plen, ws[SH_UNIQUE].as<uint8_t>(),
                               ws[SH_UNIQUE_R].as<uint8_t>(), fo.score, fo.status, ro.score, ro.status, ro.ref_end, ro.query_end,
                               ws[SH_ULIST].as<uint32_t>(), ws[SH_UCOUNT].as<uint32_t>(), settled);
            rest.items = ws[SH_ULIST].as<uint32_t>();
            rest_count = ws[SH_UCOUNT].as<uint32_t>();
        }
    }
    hipError_t e = launch_shared_ends(rest, std::max<uint32_t>(st.max_len, 1), ctx->d_pseq.as<uint8_t>(), (uint32_t)ctx->pseq_len,
                                      ctx->d_sc.as<ScoringDev>(), rule, ro, fo.ref_end, ws[SH_QEM].as<uint32_t>(), stream, rest_count);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared ranges reverse pass", e);
    ZSW_HIP(ctx, hipMemsetAsync(ws[SH_MIS].p, 0, 4, stream));
    ZSW_HIP(ctx, launch_ranges_combine(n, fo.score, fo.status, fo.ref_end, fo.query_end, ro.score, ro.status, ro.ref_end, ro.query_end,
                                       rw[RW_O0].as<uint32_t>(), rw[RW_O1].as<uint32_t>(), rw[RW_O2].as<uint32_t>(), rw[RW_O3].as<uint32_t>(),
                                       rw[RW_O4].as<uint32_t>(), rw[RW_O5].as<uint8_t>(), ws[SH_MIS].as<uint32_t>(), stream));
    out->score = rw[RW_O0].as<uint32_t>();
    out->rs = rw[RW_O1].as<uint32_t>();
    out->re = rw[RW_O2].as<uint32_t>();
    out->qs = rw[RW_O3].as<uint32_t>();
    out->qe = rw[RW_O4].as<uint32_t>();
    out->status = rw[RW_O5].as<uint8_t>();
    out->tier = fo.tier;
    return ZSW_OK;
}

// SharedProfiles::sw_align_from_i*_3pass (profile_set.rs:212-283, 552-560; profile.rs:536-552 -> three_pass.rs:21-104): the shared
// role's ranges, then the third pass with `reference` = read i and `query` = the profile sequence (ScalarProfile over its sub-range).
zsw_error run_threepass_shared(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, int invert, zsw_alignment* out_aln, uint8_t* out_status,
                               uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream_) {
    zsw_error ze = check_shared(ctx);
    if (ze != ZSW_OK) return ze;
    if (!reads || !out_aln || !out_status || !out_n_ciglets || (ciglet_cap && (!out_inc || !out_op)))
        return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    SharedStage guard(ctx, true);
    Staged st;
    {
        uint32_t dummy_score = 0;
        uint8_t dummy_status = 0;
        ze = stage(ctx, reads, stream, false, false, &dummy_score, &dummy_status, nullptr, nullptr, nullptr, &st);
        if (ze != ZSW_OK) return ze;
    }
    *out_n_ciglets = 0;
    if (reads->n_reads == 0) return ZSW_OK;
    RangesDev rd;
    ze = ranges_shared_device(ctx, st, rule, stream, &rd);
    if (ze != ZSW_OK) return ze;
    return threepass_third_pass(ctx, st, rd, ctx->d_pseq.as<uint8_t>(), (uint32_t)ctx->pseq_len, reads->mem == ZSW_MEM_HOST, invert, out_aln, out_status, out_tier,
                                out_inc, out_op, ciglet_cap, out_n_ciglets, stream);
}

zsw_error run_align_shared(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, int lanes_w8, int lanes_w16, int lanes_w32, int invert,
                           zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap,
                           uint64_t* out_n_ciglets, void* stream_) {
    zsw_error ze = check_shared(ctx);
    if (ze != ZSW_OK) return ze;
    if (!reads || !out_aln || !out_status || !out_n_ciglets || (ciglet_cap && (!out_inc || !out_op)))
        return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    SharedStage guard(ctx, true);
    Staged st;
    {
        uint32_t dummy_score = 0;
        uint8_t dummy_status = 0;
        ze = stage(ctx, reads, stream, false, false, &dummy_score, &dummy_status, nullptr, nullptr, nullptr, &st);
        if (ze != ZSW_OK) return ze;
    }
    const uint32_t n = (uint32_t)reads->n_reads;
    *out_n_ciglets = 0;
    if (n == 0) return ZSW_OK;
    const bool host = reads->mem == ZSW_MEM_HOST;
    DevBuf* ws = ctx->a_ws;
    ZSW_HIP(ctx, ws[WS_SCORE].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ws[WS_STATUS].ensure((size_t)n + 4));
    ZSW_HIP(ctx, ws[WS_TIER].ensure((size_t)n + 4));
    ZSW_HIP(ctx, ws[WS_REND].ensure((size_t)n * 4 + 4));
    // pass 1: score, first row of the read holding it (sw_simd_align's `best` and `r_end`, striped.rs:449-598)
    ScoreOut so;
    so.score = ws[WS_SCORE].as<uint32_t>();
    so.status = ws[WS_STATUS].as<uint8_t>();
    so.tier = ws[WS_TIER].as<uint8_t>();
    so.ref_end = ws[WS_REND].as<uint32_t>();
    so.query_end = nullptr;
    so.fb_list = nullptr;
    so.fb_count = nullptr;
    // Certificate mode, as in the read-as-profile role (zsw_capi.hip run_align; tests/models/align_gapless_cert.cpp and
    // align_onegap_cert.cpp check the swapped roles too): the first pass is the whole of sw_simd_score_ranges in this role, whose two
    // seeded passes tell which reads have both maxima in one cell each; a read whose only optimal alignment is gapless or has one gap
    // run gets it from the classify pass of zsw_threepass.hip and never sees the literal recurrence, which in this role walks all the
    // vectors of the shared sequence's profile for every base of the read.
    const uint8_t* pass2_status = nullptr;
    const bool certify = !(ctx->flags() & ZSW_DEBUG_ALIGN_NO_CERTIFICATE) && ctx->h_sc.gap_open > 0;
    if (certify) {
        const uint32_t MAXC0 = 32;
        ZSW_HIP(ctx, ws[WS_CERT_OK].ensure((size_t)n + 4));
        ZSW_HIP(ctx, ws[WS_CERT_DONE].ensure((size_t)n + 4));
        ZSW_HIP(ctx, ws[WS_CERT_STATUS].ensure((size_t)n + 4));
        ZSW_HIP(ctx, ws[WS_ALN].ensure((size_t)n * sizeof(zsw_alignment)));
        ZSW_HIP(ctx, ws[WS_CIGSTART].ensure((size_t)n * 8));
        ZSW_HIP(ctx, ws[WS_CIGRAW].ensure((size_t)n * 4));
        ZSW_HIP(ctx, ws[WS_CIG].ensure((size_t)n * MAXC0 * 4));
        ZSW_HIP(ctx, ws[WS_FBLIST].ensure((size_t)n * 4 + 4));
        ZSW_HIP(ctx, ws[WS_FBCOUNT].ensure(16));
        RangesDev rd;
        ze = ranges_shared_device(ctx, st, rule, stream, &rd, ws[WS_CERT_OK].as<uint8_t>());
        ctx->timer.end(stream);
        if (ze != ZSW_OK) return ze;
        so.score = rd.score;
        so.status = rd.status;
        so.tier = rd.tier;
        so.ref_end = rd.re;
        int maxw = 0;
        for (int i = 0; i < ctx->h_sc.S * ctx->h_sc.S; ++i) maxw = std::max(maxw, (int)ctx->h_sc.w[i]);
        ThreePassArgs a;
        a.b = st.b;
        a.ref = nullptr;
        a.ref_len = 0;
        a.pseq = ctx->d_pseq.as<uint8_t>();  // three_pass.rs:21-26 with the roles of this call: `reference` = read i
        a.pseq_len = (uint32_t)ctx->pseq_len;
        a.sc = ctx->d_sc.as<ScoringDev>();
        a.score = rd.score;
        a.rs = rd.rs;
        a.re = rd.re;
        a.qs = rd.qs;
        a.qe = rd.qe;
        a.status = rd.status;
        a.list = nullptr;
        a.list_count = nullptr;
        a.dp_list = nullptr;
        a.dp_count = nullptr;
        a.dp_need_max = nullptr;
        a.scratch = nullptr;
        a.slots = 0;
        a.slot_bytes = 0;
        a.cig = ws[WS_CIG].as<uint32_t>();
        a.maxc = MAXC0;
        a.pool_base = 0;
        a.by_item = 0;
        a.cig_start = ws[WS_CIGSTART].as<uint64_t>();
        a.cig_raw = ws[WS_CIGRAW].as<uint32_t>();
        a.aln = ws[WS_ALN].as<zsw_alignment>();
        a.fb_list = ws[WS_FBLIST].as<uint32_t>();
        a.fb_count = ws[WS_FBCOUNT].as<uint32_t>();
        a.invert = invert;
        a.cert_ok = ws[WS_CERT_OK].as<uint8_t>();
        a.cert_done = ws[WS_CERT_DONE].as<uint8_t>();
        a.cert_maxw = maxw;
        a.cert_go = ctx->h_sc.gap_open;
        a.cert_ge = ctx->h_sc.gap_extend;
        // reads that need the sweeps over their two-run alternatives are listed and take a second, dense launch (ThreePassArgs::sweep_list)
        ZSW_HIP(ctx, ws[WS_ITEMS].ensure((size_t)n * 4 + 4));
        ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].as<uint32_t>() + 2, 0, 4, stream));
        a.sweep_list = ws[WS_ITEMS].as<uint32_t>();
        a.sweep_count = ws[WS_FBCOUNT].as<uint32_t>() + 2;
        hipError_t ce = launch_threepass(a, std::min<uint32_t>((n + 63) / 64, 65536u), stream);
        if (ce != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared align: certificate pass", ce);
        a.list = a.sweep_list;
        a.list_count = a.sweep_count;
        a.sweep_pass = true;
        ce = launch_threepass(a, std::min<uint32_t>((n / 4 + 63) / 64 + 1, 16384u), stream);
        if (ce != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared align: certificate pass (sweeps)", ce);
        ZSW_HIP(ctx, launch_cert_status(n, rd.status, ws[WS_CERT_DONE].as<uint8_t>(), ws[WS_CERT_STATUS].as<uint8_t>(), stream));
        pass2_status = ws[WS_CERT_STATUS].as<uint8_t>();
    } else {
        ze = shared_ends_device(ctx, st, rule, so, stream);
        if (ze != ZSW_OK) return ze;
        pass2_status = so.status;
    }
    // pass 2 per tier (each tier of the cascade has its own lane count, hence its own striping of the shared sequence)
    const int S = ctx->h_sc.S;
    const uint32_t plen = (uint32_t)ctx->pseq_len;
    const uint32_t MAXC = 32;
    const uint32_t W = std::max<uint32_t>(st.max_len, 1);
    ZSW_HIP(ctx, ws[WS_ALN].ensure((size_t)n * sizeof(zsw_alignment)));
    ZSW_HIP(ctx, ws[WS_CIGSTART].ensure((size_t)n * 8));
    ZSW_HIP(ctx, ws[WS_CIGRAW].ensure((size_t)n * 4));
    ZSW_HIP(ctx, ws[WS_FBLIST].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ws[WS_FBCOUNT].ensure(8));
    ZSW_HIP(ctx, ws[WS_CIG].ensure((size_t)n * MAXC * 4));
    ZSW_HIP(ctx, ws[WS_ITEMS].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ctx->sh_ws[SH_LIST].ensure(16));
    ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].p, 0, 4, stream));
    ctx->timer.begin(stream);
    struct Tier {
        uint8_t code;
        int N;
    };
    const Tier tiers[3] = {{8, lanes_w8}, {16, lanes_w16}, {32, lanes_w32}};
    auto run_tier = [&](int N, const uint32_t* d_items, uint32_t count, uint32_t maxc, int by_item, DevBuf& ringbuf, DevBuf& cigbuf) -> zsw_error {
        if (!count) return ZSW_OK;
        uint32_t grid = std::min<uint32_t>((count + (64 / (uint32_t)N) - 1) / (64 / (uint32_t)N), 4096u);
        // every row's flags of every read in flight: plen x W bytes per read (300 KB at 2 kb x 150); 4,096 wavefronts are four per SIMD
        while (grid > 1 && align_shared_ring_bytes(N, plen, W, grid, S) > (size_t(8) << 30)) grid /= 2;
        if (align_shared_ring_bytes(N, plen, W, grid, S) > (size_t(12) << 30)) return fail(ctx, ZSW_ERR_UNSUPPORTED, "shared profile too long for the flag ring");
        ZSW_HIP(ctx, ringbuf.ensure(align_shared_ring_bytes(N, plen, W, grid, S) + 64));
        if (by_item) ZSW_HIP(ctx, cigbuf.ensure((uint64_t)count * maxc * 4 + 64));
        BatchDev b = st.b;
        b.items = d_items;
        b.n_items = count;
        hipError_t he = align_pass2_shared(N, ctx->d_pseq.as<uint8_t>(), plen, b, ctx->d_sc.as<ScoringDev>(), S, so.score, so.ref_end, pass2_status, W, maxc,
                                           ringbuf.as<uint8_t>(), grid, cigbuf.as<uint32_t>(), 0, by_item, ws[WS_CIGSTART].as<uint64_t>(),
                                           ws[WS_CIGRAW].as<uint32_t>(), ws[WS_ALN].as<zsw_alignment>(), ws[WS_FBLIST].as<uint32_t>(),
                                           ws[WS_FBCOUNT].as<uint32_t>(), invert, stream,
                                           rule.n_tiers == 1 ? rule.tier_code[0] != 32 : N != lanes_w32);
        if (he != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "shared align pass 2", he);
        return ZSW_OK;
    };
    bool seen[65] = {false};
    for (const Tier& t : tiers) {
        if (rule.n_tiers == 1 ? t.code != rule.tier_code[0] : false) continue;
        bool in_rule = false;
        for (int k = 0; k < rule.n_tiers; ++k) in_rule = in_rule || rule.tier_code[k] == t.code;
        if (!in_rule) continue;
        (void)seen;
        uint32_t* d_list = ws[WS_ITEMS].as<uint32_t>();
        uint32_t* d_count = ctx->sh_ws[SH_LIST].as<uint32_t>();
        ZSW_HIP(ctx, hipMemsetAsync(d_count, 0, 4, stream));
        hipLaunchKernelGGL(iota_some_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, pass2_status, so.tier, t.code, d_list, d_count);
        uint32_t count = 0;
        ZSW_HIP(ctx, hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, stream));
        ZSW_HIP(ctx, hipStreamSynchronize(stream));
        ze = run_tier(t.N, d_list, count, MAXC, 0, ws[WS_RING], ws[WS_CIG]);
        if (ze != ZSW_OK) return ze;
        ZSW_HIP(ctx, hipStreamSynchronize(stream));  // the item list is reused by the next tier
    }
    ctx->timer.end(stream);
    // reads whose CIGAR needs more than 32 ciglets: rerun with room for any CIGAR (the window already holds every row)
    uint32_t n_fb = 0;
    ZSW_HIP(ctx, hipMemcpyAsync(&n_fb, ws[WS_FBCOUNT].p, 4, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    if (n_fb) {
        std::vector<uint32_t> fb(n_fb);
        std::vector<uint8_t> h_tier(n);
        ZSW_HIP(ctx, hipMemcpy(fb.data(), ws[WS_FBLIST].p, (size_t)n_fb * 4, hipMemcpyDeviceToHost));
        ZSW_HIP(ctx, hipMemcpy(h_tier.data(), so.tier, n, hipMemcpyDeviceToHost));
        std::map<int, std::vector<uint32_t>> by_n;
        for (uint32_t id : fb) by_n[h_tier[id] == 8 ? lanes_w8 : h_tier[id] == 16 ? lanes_w16 : lanes_w32].push_back(id);
        ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].p, 0, 4, stream));
        const uint32_t maxc_full = plen + W + 4;
        for (auto& g : by_n) {
            ZSW_HIP(ctx, ws[WS_ITEMS2].ensure(g.second.size() * 4 + 4));
            ZSW_HIP(ctx, hipMemcpy(ws[WS_ITEMS2].p, g.second.data(), g.second.size() * 4, hipMemcpyHostToDevice));
            ze = run_tier(g.first, ws[WS_ITEMS2].as<uint32_t>(), (uint32_t)g.second.size(), maxc_full, 1, ws[WS_RING2], ws[WS_CIG2]);
            if (ze != ZSW_OK) return ze;
            ZSW_HIP(ctx, hipStreamSynchronize(stream));
        }
        uint32_t again = 0;
        ZSW_HIP(ctx, hipMemcpy(&again, ws[WS_FBCOUNT].p, 4, hipMemcpyDeviceToHost));
        if (again) return fail(ctx, ZSW_ERR_HIP, "shared alignment traceback did not complete");
    }
    return finish_alignments(ctx, ws, n, host, so.status, so.tier, invert, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap, out_n_ciglets,
                             stream);
}

}  // namespace

extern "C" {

zsw_error zsw_set_profile_sequence(zsw_context* ctx, const uint8_t* sequence, size_t len, zsw_mem mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || (!sequence && len)) return ZSW_ERR_INVALID_ARGUMENT;
    if (len == 0) return ZSW_ERR_EMPTY_SEQUENCE;  // StripedProfile::new -> Err(ProfileError::EmptySequence) (profile.rs:32-44)
    if (len > 0x7fffffffull) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "profile sequence too long");
    // the sequence already set (callers set it before every call): nothing to do, and no device-wide synchronisation
    if (ctx->pseq_set && mem == ZSW_MEM_HOST && ctx->pseq_len == len && memcmp(ctx->h_pseq.data(), sequence, len) == 0) return ZSW_OK;
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->pseq_set) ZSW_HIP(ctx, hipDeviceSynchronize());  // queued kernels may still read the previous sequence
    ZSW_HIP(ctx, ctx->d_pseq.ensure(len + 16));
    ZSW_HIP(ctx, hipMemcpy(ctx->d_pseq.p, sequence, len, mem == ZSW_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
    ctx->h_pseq.resize(len);  // the seeded pass of the score calls builds its k-mer index from a host copy
    if (mem == ZSW_MEM_HOST) memcpy(ctx->h_pseq.data(), sequence, len);
    else ZSW_HIP(ctx, hipMemcpy(ctx->h_pseq.data(), sequence, len, hipMemcpyDeviceToHost));
    ctx->seed_shared.valid = false;
    ctx->seed_shared_rev.valid = false;
    ctx->pseq_len = len;
    ctx->pseq_set = true;
    return ZSW_OK;
}

zsw_error zsw_score_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, uint32_t* out_score,
                                 uint8_t* out_status, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_score_shared(ctx, reads, rule, out_score, out_status, nullptr, stream);
}

zsw_error zsw_score_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, uint32_t* out_score,
                                      uint8_t* out_status, uint8_t* out_tier, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_score_shared(ctx, reads, rule, out_score, out_status, out_tier, stream);
}

zsw_error zsw_score_ends_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, uint32_t* out_score,
                                      uint32_t* out_ref_end, uint32_t* out_query_end, uint8_t* out_status, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_ends_shared(ctx, reads, rule, out_score, out_ref_end, out_query_end, out_status, stream);
}

zsw_error zsw_score_ranges_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, uint32_t* out_score,
                                        uint32_t* out_ref_start, uint32_t* out_ref_end, uint32_t* out_query_start, uint32_t* out_query_end,
                                        uint8_t* out_status, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_ranges_shared(ctx, reads, rule, out_score, out_ref_start, out_ref_end, out_query_start, out_query_end, out_status, nullptr, stream);
}

zsw_error zsw_score_ranges_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, uint32_t* out_score,
                                             uint32_t* out_ref_start, uint32_t* out_ref_end, uint32_t* out_query_start, uint32_t* out_query_end,
                                             uint8_t* out_status, uint8_t* out_tier, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_ranges_shared(ctx, reads, rule, out_score, out_ref_start, out_ref_end, out_query_start, out_query_end, out_status, out_tier, stream);
}

zsw_error zsw_align_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert, zsw_alignment* out_aln,
                                 uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_align_shared(ctx, reads, rule, lanes, lanes, lanes, invert, out_aln, out_status, nullptr, out_inc, out_op, ciglet_cap, out_n_ciglets,
                            stream);
}

zsw_error zsw_align_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert, zsw_alignment* out_aln,
                                      uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap,
                                      uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_align_shared(ctx, reads, rule, preset_bits / 8, preset_bits / 16, preset_bits / 32, invert, out_aln, out_status, out_tier, out_inc,
                            out_op, ciglet_cap, out_n_ciglets, stream);
}

zsw_error zsw_align_3pass_shared_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert, zsw_alignment* out_aln,
                                       uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_threepass_shared(ctx, reads, rule, invert, out_aln, out_status, nullptr, out_inc, out_op, ciglet_cap, out_n_ciglets, stream);
}

zsw_error zsw_align_3pass_shared_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert, zsw_alignment* out_aln,
                                            uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap,
                                            uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_threepass_shared(ctx, reads, rule, invert, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap, out_n_ciglets, stream);
}

}  // extern "C"
