// zsw_capi.hip — the C ABI (include/zoe_sw.h): context, scoring/reference upload, batched entry points.
#include <map>
#include <new>
#include <utility>

#include "zsw_context.hpp"
#include "zsw_synth.h"

using namespace zsw;
using namespace zsw::capi;

namespace zsw {
namespace capi {


__global__ void maxlen_kernel(const uint64_t* offsets, uint32_t n, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m = 0;
    for (; i < n; i += gridDim.x * blockDim.x) m = max(m, (uint32_t)(offsets[i + 1] - offsets[i]));
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__global__ void synth_kernel(uint64_t seed, uint64_t first, uint64_t n, uint32_t len, const uint8_t* ref, uint32_t R,
                             uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) zsw_synth_read(seed, first + i, ref, R, len, out + i * len);
}

__global__ void synth_ragged_kernel(uint64_t seed, uint64_t first, uint64_t n, uint32_t min_len, uint32_t max_len,
                                    const uint64_t* offsets, const uint8_t* ref, uint32_t R, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) zsw_synth_read(seed, first + i, ref, R, zsw_synth_len(seed, first + i, min_len, max_len), out + offsets[i]);
}

__global__ void selftest_kernel(uint32_t* out) {
    const int lane = threadIdx.x;
    // 1. v_perm_b32 byte order: selector bytes 0-3 pick from the SECOND operand
    out[lane] = __builtin_amdgcn_perm(0x77665544u, 0x33221100u, 0x07040300u);
    // 2. shuffle direction within groups of 4
    out[64 + lane] = (uint32_t)__shfl_up(lane, 1, 4);
    // 3. packed saturation
    typedef short s2 __attribute__((ext_vector_type(2)));
    s2 a = {(short)32000, (short)-32000}, b = {(short)1000, (short)-1000};
    s2 c = __builtin_elementwise_add_sat(a, b);
    out[128 + lane] = __builtin_bit_cast(uint32_t, c);
    s2 d = __builtin_elementwise_sub_sat(b, a);  // 1000-32000, -1000+32000
    out[192 + lane] = __builtin_bit_cast(uint32_t, d);
}



// Brings a batch to the device (or validates device pointers), finds the longest read and sizes the workspace.
// ZSW_ENCODING_PACKED4: two residue indices per byte -> one byte per base (byte_of[index]: a byte the context's map sends to it)
struct Unpack4 {
    uint8_t byte_of[16];
};
__global__ void unpack4_kernel(const uint8_t* packed, uint64_t n, uint32_t len, Unpack4 u, uint8_t* out) {
    const uint32_t stride = (len + 1) / 2;
    const uint64_t total = n * stride;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t read = k / stride;
        const uint32_t c = 2 * (uint32_t)(k - read * stride);
        const uint8_t p = packed[k];
        out[read * len + c] = u.byte_of[p & 15u];
        if (c + 1 < len) out[read * len + c + 1] = u.byte_of[p >> 4];
    }
}
// reads [first, first + cnt) of a packed host batch: H2D copy of the packed bytes and their expansion into s_bases, on `stream`
hipError_t copy_packed_chunk(zsw_context* ctx, const zsw_batch* reads, size_t first, size_t cnt, hipStream_t stream) {
    const uint32_t L = reads->fixed_len;
    const size_t stride = (L + 1) / 2;
    hipError_t e = hipMemcpyAsync(ctx->s_packed.as<uint8_t>() + first * stride, reads->bases + first * stride, cnt * stride, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess || cnt == 0) return e;
    Unpack4 u;
    for (int k = 0; k < 16; ++k) u.byte_of[k] = 0;
    bool seen[16] = {false};
    for (int b = 255; b >= 0; --b) {  // the smallest byte of each index (upper case before lower case)
        const int idx = ctx->h_sc.index_map[b];
        if (idx < 16) {
            u.byte_of[idx] = (uint8_t)b;
            seen[idx] = true;
        }
    }
    (void)seen;
    const uint64_t total = (uint64_t)cnt * stride;
    hipLaunchKernelGGL(unpack4_kernel, dim3((unsigned)std::min<uint64_t>((total + 255) / 256, 65536)), dim3(256), 0, stream,
                       ctx->s_packed.as<uint8_t>() + first * stride, (uint64_t)cnt, L, u, ctx->s_bases.as<uint8_t>() + first * L);
    return hipGetLastError();
}

zsw_error stage(zsw_context* ctx, const zsw_batch* reads, hipStream_t stream, bool want_tier, bool want_ends,
                uint32_t* out_score, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_rend, uint32_t* out_qend,
                Staged* st, bool defer_bases_copy) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!ctx->scoring_set || !ctx->reference_set) return fail(ctx, ZSW_ERR_NOT_CONFIGURED, "scoring/reference not set");
    if (!reads || !out_score || !out_status) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    if (reads->n_reads > 0x7fffffffull) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "n_reads > 2^31-1 per call");
    if (reads->n_reads && !reads->bases) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null bases");
    if (!reads->offsets && reads->fixed_len == 0 && reads->n_reads) return ZSW_ERR_EMPTY_SEQUENCE;
    const bool packed = reads->encoding == ZSW_ENCODING_PACKED4;
    if (reads->encoding != ZSW_ENCODING_BYTES && !packed) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "unknown zsw_batch.encoding");
    if (packed && (reads->mem != ZSW_MEM_HOST || reads->offsets || ctx->h_sc.S > 16))
        return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "ZSW_ENCODING_PACKED4: fixed-length host batches, alphabets of up to 16 letters");
    const uint32_t n = (uint32_t)reads->n_reads;
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    st->b.n_reads = n;
    st->b.n_items = n;
    st->b.items = nullptr;
    st->b.fixed_len = reads->fixed_len;
    if (reads->mem == ZSW_MEM_HOST) {
        size_t total = reads->offsets ? (size_t)reads->offsets[n] : (size_t)n * reads->fixed_len;
        ZSW_HIP(ctx, ctx->s_bases.ensure(total + 16));
        if (packed) {  // half the bytes cross PCIe; the device spells them out as bytes again (one representative byte per residue index)
            ZSW_HIP(ctx, ctx->s_packed.ensure((size_t)n * ((reads->fixed_len + 1) / 2) + 16));
            if (!defer_bases_copy) ZSW_HIP(ctx, copy_packed_chunk(ctx, reads, 0, n, stream));
        } else if (!defer_bases_copy) ZSW_HIP(ctx, hipMemcpyAsync(ctx->s_bases.p, reads->bases, total, hipMemcpyHostToDevice, stream));
        st->b.bases = ctx->s_bases.as<uint8_t>();
        st->b.offsets = nullptr;
        st->max_len = reads->fixed_len;
        if (reads->offsets) {
            ZSW_HIP(ctx, ctx->s_offsets.ensure((size_t)(n + 1) * 8));
            ZSW_HIP(ctx, hipMemcpyAsync(ctx->s_offsets.p, reads->offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, stream));
            st->b.offsets = ctx->s_offsets.as<uint64_t>();
            uint32_t m = 0;
            for (uint32_t i = 0; i < n; ++i) {
                if (reads->offsets[i + 1] < reads->offsets[i]) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "offsets not monotone");
                m = std::max<uint32_t>(m, (uint32_t)(reads->offsets[i + 1] - reads->offsets[i]));
            }
            st->max_len = m;
        }
        ZSW_HIP(ctx, ctx->s_score.ensure((size_t)n * 4 + 4));
        ZSW_HIP(ctx, ctx->s_status.ensure((size_t)n + 4));
        st->d_score = ctx->s_score.as<uint32_t>();
        st->d_status = ctx->s_status.as<uint8_t>();
        if (want_tier && out_tier) {
            ZSW_HIP(ctx, ctx->s_tier.ensure((size_t)n + 4));
            st->d_tier = ctx->s_tier.as<uint8_t>();
        }
        if (want_ends) {
            ZSW_HIP(ctx, ctx->s_rend.ensure((size_t)n * 4 + 4));
            ZSW_HIP(ctx, ctx->s_qend.ensure((size_t)n * 4 + 4));
            st->d_rend = ctx->s_rend.as<uint32_t>();
            st->d_qend = ctx->s_qend.as<uint32_t>();
        }
    } else {
        st->b.bases = reads->bases;
        st->b.offsets = reads->offsets;
        st->max_len = reads->fixed_len;
        if (reads->offsets && n) {
            ZSW_HIP(ctx, ctx->d_maxlen.ensure(4));
            ZSW_HIP(ctx, hipMemsetAsync(ctx->d_maxlen.p, 0, 4, stream));
            hipLaunchKernelGGL(maxlen_kernel, dim3(std::min<uint32_t>(1024, (n + 255) / 256)), dim3(256), 0, stream,
                               reads->offsets, n, ctx->d_maxlen.as<uint32_t>());
            uint32_t m = 0;
            ZSW_HIP(ctx, hipMemcpyAsync(&m, ctx->d_maxlen.p, 4, hipMemcpyDeviceToHost, stream));
            ZSW_HIP(ctx, hipStreamSynchronize(stream));
            st->max_len = m;
        }
        st->d_score = out_score;
        st->d_status = out_status;
        st->d_tier = want_tier ? out_tier : nullptr;
        st->d_rend = out_rend;
        st->d_qend = out_qend;
    }
    // workspace
    ZSW_HIP(ctx, ctx->d_fb_list.ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ctx->d_fb_count.ensure(4));
    if (reads->offsets || st->max_len > LONGEST_STRIP) ZSW_HIP(ctx, ctx->d_bucket_items.ensure((size_t)n * 4 + 4));
    if (reads->offsets && !ctx->side) {
        SideStreams* p = new (std::nothrow) SideStreams();
        if (p) {
            bool ok = hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) == hipSuccess;
            for (int i = 0; ok && i < SideStreams::N; ++i)
                ok = hipStreamCreateWithFlags(&p->s[i], hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&p->join[i], hipEventDisableTiming) == hipSuccess;
            if (ok) ctx->side = p;  // otherwise the classes simply run in sequence
            else delete p;
        }
    }
    if (st->max_len > LONGEST_STRIP) {  // tile-by-tile scoring of long reads: boundary buffers for as many read pairs as 1 GiB holds
        const size_t per_pair = std::max<size_t>(ctx->ref_len, 1) * 8, pairs = ((size_t)n + 1) / 2;
        const size_t want = 2 * std::min<size_t>(pairs * per_pair, std::max<size_t>(per_pair, size_t(512) << 20));
        ZSW_HIP(ctx, ctx->d_tile_buf.ensure(want));
        ZSW_HIP(ctx, ctx->d_tile_state.ensure((size_t)n * 16 + 16));
    }
    ZSW_HIP(ctx, ctx->d_bucket_counts.ensure(64 * 4));
    ctx->prune_chunk = 0;
    ctx->seed_ready = false;
    ctx->chunk_ready = false;
    const uint32_t flags = ctx->flags();
    const bool any_size = (flags & ZSW_DEBUG_SCORE_PRUNE_ANY_SIZE) != 0;
    // a shared-role call (zsw_capi_shared.hip) scores the reads against the profile sequence with the roles swapped and the matrix
    // transposed: its index is seed_shared, over h_pseq
    const bool shared = ctx->shared_call;
    const size_t seq_len = shared ? ctx->pseq_len : ctx->ref_len;
    if ((!shared || ctx->shared_seedable) && (flags & ZSW_DEBUG_SCORE_PRUNE) && !(flags & ZSW_DEBUG_PRUNE_STRIP) && n > 0 && (n >= SEED_MIN_READS || any_size) && seq_len > 0 &&
        st->max_len >= SEED_MIN_LEN) {
        // the seeded exact pass: index of the reference (first use after a change of reference or matrix), 28 bytes + 4 bits per base
        // of workspace per read, the banded kernel's strip-boundary buffers and the worklist of the reads it hands back. If the device cannot spare them the full pass runs.
        SeedIndex& index = shared ? ctx->seed_shared : ctx->seed;
        if (!index.valid) {
            if (shared) {
                ScoringDev t = ctx->h_sc;
                for (int r = 0; r < t.S; ++r)
                    for (int q = 0; q < t.S; ++q) t.w[r * t.S + q] = ctx->h_sc.w[q * t.S + r];
                ZSW_HIP(ctx, seed_index_update(&index, t, ctx->h_pseq.data(), seq_len));
            } else {
                ZSW_HIP(ctx, seed_index_update(&index, ctx->h_sc, ctx->h_ref.data(), seq_len));
            }
        }
        if (index.usable) {
            // ragged batches: a region per length class (the banded pass's buffers are sized per region, by the reads in it)
            // (ragged: the classes' sort temporaries and round-ups, and SEED_BAND_CLASS_MIN_GRID blocks of boundary buffer each)
            const size_t want = seed_workspace_bytes(n, st->max_len) + 24 * seed_workspace_bytes(0, st->max_len) +
                                (reads->offsets ? 24 * seed_workspace_bytes(std::min<uint32_t>(n, 2 * SEED_BAND_CLASS_MIN_GRID * 256), st->max_len, SEED_BAND_CLASS_MIN_GRID) : 0);
            if (ctx->d_seed_work.ensure(want) == hipSuccess && ctx->d_seed_gtab.ensure(2 * (seq_len + 2 * SEED_GTAB_PAD) * 8) == hipSuccess &&
                ctx->d_prune_list.ensure((size_t)n * 4 + 4) == hipSuccess &&
                ctx->d_prune_count.ensure(64 * 4) == hipSuccess) {
                ZSW_HIP(ctx, hipMemsetAsync(ctx->d_prune_count.p, 0, 64 * 4, stream));  // [1] the call's total, [0], [2..] lists in flight
                ctx->seed_ready = true;
                // long references: the reads handed back are scored in chunks of rows (one key per read collects the chunks' results)
                if (seq_len >= 8192) {
                    ctx->chunk_ready = ctx->d_chunk_keys.ensure((size_t)n * 8 + 8) == hipSuccess;
                    if (!ctx->chunk_ready) (void)hipGetLastError();
                }
            } else {
                (void)hipGetLastError();
                ctx->d_seed_work.release();
                ctx->err = "seeded pass skipped: workspace allocation failed (every cell is computed instead)";
            }
        }
    }
    // the column-pruned pass (strip + window): the default first pass of 8..32-letter alphabets (the seeded pass's k-mer argument
    // needs substitutions that cost something: it serves the 5-letter tables), a cross-check for the others (ZSW_DEBUG_PRUNE_STRIP)
    const bool wide_alphabet = ctx->h_sc.S > 7 && ctx->h_sc.S <= 32 && !(flags & ZSW_DEBUG_NO_WIDE);
    if (!ctx->shared_call && (flags & ZSW_DEBUG_SCORE_PRUNE) && ((flags & ZSW_DEBUG_PRUNE_STRIP) || wide_alphabet) && n > 0 && (n >= PR_MIN_READS || any_size) &&
        ctx->ref_len > 0 && (reads->offsets ? st->max_len > 64 : prune_class_for(st->max_len) >= 0)) {
        const uint32_t chunk = prune_chunk_reads((uint32_t)n, (uint32_t)ctx->ref_len);
        // a reference so long that the boundary streams of a round's reads no longer fill the chip: the full pass
        if (chunk >= std::min<uint32_t>((uint32_t)n, PR_MIN_READS) || any_size) {
            // the workspace is large (8 B per read pair and reference row): if the device cannot spare it, the full pass runs
            if (ctx->d_prune.ensure(prune_workspace_bytes(chunk, (uint32_t)ctx->ref_len)) == hipSuccess &&
                ctx->d_prune_list.ensure((size_t)n * 4 + 4) == hipSuccess && ctx->d_prune_count.ensure(64 * 4) == hipSuccess) {
                ZSW_HIP(ctx, hipMemsetAsync(ctx->d_prune_count.p, 0, 64 * 4, stream));  // [0] the class in flight, [1] the call's total
                ctx->prune_chunk = chunk;
            } else {
                (void)hipGetLastError();
                ctx->d_prune.release();
                ctx->err = "column-pruned pass skipped: workspace allocation failed (every cell is computed instead)";
            }
        }
    }
    uint32_t need = std::max<uint32_t>(512, (st->max_len + 127) / 128 * 128);
    if (need > ctx->scratch_len || !ctx->d_scratch.p) {
        // slots from a byte budget: 16,384 for reads up to 8 kb, down to 64 for genome-sized reads (2 rows of `need` ints per slot)
        size_t slots = EXACT_SCRATCH_BUDGET / (2 * (size_t)need * sizeof(int32_t)) / 64 * 64;
        slots = std::min(EXACT_SLOTS, std::max<size_t>(64, slots));
        ZSW_HIP(ctx, ctx->d_scratch.ensure(2 * slots * (size_t)need * sizeof(int32_t)));
        ctx->scratch_len = need;
        ctx->exact_slots = slots;
    }
    return ZSW_OK;
}

ScoreWorkspace score_ws(zsw_context* ctx) {
    ScoreWorkspace w;
    w.scratch = ctx->d_scratch.as<int32_t>();
    w.slots = ctx->exact_slots;
    w.scratch_len = ctx->scratch_len;
    w.bucket_items = ctx->d_bucket_items.as<uint32_t>();
    w.bucket_counts = ctx->d_bucket_counts.as<uint32_t>();
    w.tile_buf = ctx->d_tile_buf.as<uint2>();
    w.tile_bytes = ctx->d_tile_buf.cap / 16 * 16;
    w.tile_state = ctx->d_tile_state.as<uint4>();
    w.side = ctx->side;
    w.debug = ctx->flags();
    if (ctx->seed_ready) {
        w.seed = ctx->shared_call ? &ctx->seed_shared : &ctx->seed;
        w.seed_work = ctx->d_seed_work.as<uint8_t>();
        w.seed_bytes = ctx->d_seed_work.cap;
        w.seed_gtab = ctx->d_seed_gtab.as<uint2>();
        w.band_dbg = ctx->band_dbg;
        w.chunk_keys = ctx->chunk_ready ? ctx->d_chunk_keys.as<unsigned long long>() : nullptr;
        w.window_timer = &ctx->timer_window;
        w.prune_fail_list = ctx->d_prune_list.as<uint32_t>();
        w.prune_fail_count = ctx->d_prune_count.as<uint32_t>();
    }
    if (ctx->prune_chunk) {
        w.prune_work = ctx->d_prune.as<uint8_t>();
        w.prune_bytes = ctx->d_prune.cap;
        w.prune_chunk = ctx->prune_chunk;
        w.prune_fail_list = ctx->d_prune_list.as<uint32_t>();
        w.prune_fail_count = ctx->d_prune_count.as<uint32_t>();
    }
    return w;
}

zsw_error unstage(zsw_context* ctx, const zsw_batch* reads, hipStream_t stream, const Staged& st, uint32_t* out_score,
                  uint8_t* out_status, uint8_t* out_tier, uint32_t* out_rend, uint32_t* out_qend) {
    if (reads->mem != ZSW_MEM_HOST) return ZSW_OK;
    const size_t n = reads->n_reads;
    ZSW_HIP(ctx, hipMemcpyAsync(out_score, st.d_score, n * 4, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipMemcpyAsync(out_status, st.d_status, n, hipMemcpyDeviceToHost, stream));
    if (st.d_tier && out_tier) ZSW_HIP(ctx, hipMemcpyAsync(out_tier, st.d_tier, n, hipMemcpyDeviceToHost, stream));
    if (st.d_rend && out_rend) ZSW_HIP(ctx, hipMemcpyAsync(out_rend, st.d_rend, n * 4, hipMemcpyDeviceToHost, stream));
    if (st.d_qend && out_qend) ZSW_HIP(ctx, hipMemcpyAsync(out_qend, st.d_qend, n * 4, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    return ZSW_OK;
}

constexpr uint32_t PIPE_CHUNK = 4'000'000;  // reads per chunk of the host-batch pipeline (600 MB at 150 bp; fewer, larger chunks keep the kernels near their large-batch rate)
constexpr uint32_t PIPE_FIRST = 500'000;    // ... and of its first chunk, whose copy nothing hides

zsw_error run_score(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, bool want_ends, uint32_t* out_score,
                    uint8_t* out_status, uint8_t* out_tier, uint32_t* out_rend, uint32_t* out_qend, void* stream_) {
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    Staged st;
    // fixed-length host batches of more than one chunk: the H2D copy of chunk k+1 overlaps the kernel of chunk k
    const bool pipelined = reads && reads->mem == ZSW_MEM_HOST && !reads->offsets && reads->fixed_len > 0 &&
                           reads->n_reads > PIPE_CHUNK && !(ctx && (ctx->flags() & ZSW_DEBUG_NO_PIPELINE));
    zsw_error ze = stage(ctx, reads, stream, out_tier != nullptr, want_ends, out_score, out_status, out_tier, out_rend,
                         out_qend, &st, pipelined);
    if (ze != ZSW_OK) return ze;
    if (reads->n_reads == 0) return ZSW_OK;
    ScoreOut out;
    out.score = st.d_score;
    out.status = st.d_status;
    out.tier = st.d_tier;
    out.ref_end = st.d_rend;
    out.query_end = st.d_qend;
    out.fb_list = ctx->d_fb_list.as<uint32_t>();
    out.fb_count = ctx->d_fb_count.as<uint32_t>();
    if (!pipelined) {
        hipError_t e = launch_score(ctx->d_sc.as<ScoringDev>(), ctx->h_sc, st.b, st.max_len, ctx->d_ref.as<uint8_t>(),
                                    (uint32_t)ctx->ref_len, rule, out, score_ws(ctx), stream, &ctx->timer, want_ends ? 2 : 0);
        if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "score launch", e);
        return unstage(ctx, reads, stream, st, out_score, out_status, out_tier, out_rend, out_qend);
    }
    // chunks: a small first one (its copy is the only one nothing hides), then PIPE_CHUNK reads each. Packed reads (half the bytes: the
    // kernels of a chunk take about as long as the copy of the next one, twice as large) grow 0.5 - 1 - 2 - 4 M instead, so the device
    // does not wait 5 ms for the second chunk: 338 -> 371 M reads/s; byte reads are bound by the copies and lose 3 % to more chunks
    const uint32_t n = (uint32_t)reads->n_reads, L = reads->fixed_len;
    std::vector<uint32_t> starts;
    const bool grow = reads->encoding == ZSW_ENCODING_PACKED4;
    for (uint32_t first = 0, size = PIPE_FIRST; first < n; first += size, size = grow ? std::min(2 * size, PIPE_CHUNK) : PIPE_CHUNK) starts.push_back(first);
    starts.push_back(n);
    const uint32_t n_chunks = (uint32_t)starts.size() - 1;
    if (!ctx->copy_stream) ZSW_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    while (ctx->copy_events.size() < 2 * (size_t)n_chunks) {
        hipEvent_t ev;
        ZSW_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->copy_events.push_back(ev);
    }
    auto copy_chunk = [&](uint32_t k) -> hipError_t {
        const size_t first = starts[k], cnt = starts[k + 1] - starts[k];
        hipError_t e = reads->encoding == ZSW_ENCODING_PACKED4
                           ? copy_packed_chunk(ctx, reads, first, cnt, ctx->copy_stream)
                           : hipMemcpyAsync(ctx->s_bases.as<uint8_t>() + first * L, reads->bases + first * L, cnt * L, hipMemcpyHostToDevice, ctx->copy_stream);
        if (e != hipSuccess) return e;
        return hipEventRecord(ctx->copy_events[k], ctx->copy_stream);
    };
    // the results of chunk k go back on the copy stream once its kernels are done (under the kernels of chunk k + 1)
    auto results_back = [&](uint32_t k) -> hipError_t {
        const size_t first = starts[k], cnt = starts[k + 1] - starts[k];
        hipError_t e = hipEventRecord(ctx->copy_events[n_chunks + k], stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, ctx->copy_events[n_chunks + k], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(out_score + first, st.d_score + first, cnt * 4, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_status + first, st.d_status + first, cnt, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (e == hipSuccess && st.d_tier && out_tier) e = hipMemcpyAsync(out_tier + first, st.d_tier + first, cnt, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (e == hipSuccess && st.d_rend && out_rend) e = hipMemcpyAsync(out_rend + first, st.d_rend + first, cnt * 4, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (e == hipSuccess && st.d_qend && out_qend) e = hipMemcpyAsync(out_qend + first, st.d_qend + first, cnt * 4, hipMemcpyDeviceToHost, ctx->copy_stream);
        return e;
    };
    ZSW_HIP(ctx, copy_chunk(0));
    for (uint32_t k = 0; k < n_chunks; ++k) {
        const uint32_t first = starts[k], cnt = starts[k + 1] - starts[k];
        ZSW_HIP(ctx, hipStreamWaitEvent(stream, ctx->copy_events[k], 0));
        BatchDev bk = st.b;
        bk.bases = st.b.bases + (size_t)first * L;
        bk.n_reads = bk.n_items = cnt;
        ScoreOut ok = out;
        ok.score = out.score + first;
        ok.status = out.status + first;
        if (out.tier) ok.tier = out.tier + first;
        if (out.ref_end) ok.ref_end = out.ref_end + first;
        if (out.query_end) ok.query_end = out.query_end + first;
        hipError_t e = launch_score(ctx->d_sc.as<ScoringDev>(), ctx->h_sc, bk, st.max_len, ctx->d_ref.as<uint8_t>(),
                                    (uint32_t)ctx->ref_len, rule, ok, score_ws(ctx), stream, &ctx->timer, want_ends ? 2 : 0);
        if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "score launch", e);
        if (k + 1 < n_chunks) ZSW_HIP(ctx, copy_chunk(k + 1));  // issued after the launch: it runs under chunk k's kernel
        ZSW_HIP(ctx, results_back(k));
    }
    ZSW_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    return ZSW_OK;
}


__global__ void ranges_prep_kernel(uint32_t n, const uint8_t* fstatus, const uint32_t* fqend, uint32_t* qe_masked) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) qe_masked[i] = fstatus[i] == ZSW_STATUS_SOME ? fqend[i] : 0;
}

// and_then / map chain of sw_simd_score_ranges (striped.rs:361-387)
__global__ void ranges_combine_kernel(uint32_t n, const uint32_t* fscore, const uint8_t* fstatus, const uint32_t* frend,
                                      const uint32_t* fqend, const uint32_t* rscore, const uint8_t* rstatus,
                                      const uint32_t* rrstart, const uint32_t* rqstart, uint32_t* out_score,
                                      uint32_t* out_rs, uint32_t* out_re, uint32_t* out_qs, uint32_t* out_qe,
                                      uint8_t* out_status, uint32_t* mismatch) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t st = fstatus[i];
    if (st == ZSW_STATUS_SOME && rstatus[i] != ZSW_STATUS_SOME) st = rstatus[i] == ZSW_STATUS_EMPTY ? (uint8_t)ZSW_STATUS_UNMAPPED : rstatus[i];
    const bool some = st == ZSW_STATUS_SOME;
    if (some && rscore[i] != fscore[i]) atomicAdd(mismatch, 1u);  // debug_assert_eq!(score, score2) (striped.rs:379)
    out_status[i] = st;
    out_score[i] = some ? fscore[i] : 0;
    out_rs[i] = some ? rrstart[i] : 0;
    out_re[i] = some ? frend[i] : 0;
    out_qs[i] = some ? rqstart[i] : 0;
    out_qe[i] = some ? fqend[i] : 0;
}


// can this call's batch take the reverse pass of the ranges as a second seeded pass (stage() has run)?
static bool seeded_reverse_possible(zsw_context* ctx, const Staged& st) {
    return ctx->seed_ready && !ctx->shared_call && !(ctx->flags() & ZSW_DEBUG_RANGES_EXACT_REVERSE) && ctx->ref_len > 0 && ctx->seed.valid && ctx->seed.usable &&
           st.max_len <= SEED_MAX_LEN;
}

// run_align's first pass in certificate mode: the forward and the reversed seeded pass only (no exact reverse pass, no combine)
struct RangesCert {
    uint32_t* safe_row;  // in: where the forward pass leaves the late-start rows of sw_simd_align's second pass
    uint8_t* settled;    // in: n bytes; out: 1 = both maxima of the read sit in one cell each, rs / qs hold its starts
    ScoreOut fwd;        // out: the forward pass's arrays (score, status, tier, ref_end, query_end)
    uint32_t *rs, *qs;   // out
};

// forward score+ends (MODE 2), reverse pass on the prefixes, combine; everything stays on the device
zsw_error ranges_device(zsw_context* ctx, const Staged& st, const ResultRule& rule, hipStream_t stream, RangesDev* out, RangesCert* cert = nullptr) {
    const uint32_t n = st.b.n_reads;
    DevBuf* ws = ctx->r_ws;
    for (int k : {RW_FSCORE, RW_FREND, RW_FQEND, RW_RSCORE, RW_RRS, RW_RQS, RW_QEM, RW_O0, RW_O1, RW_O2, RW_O3, RW_O4})
        ZSW_HIP(ctx, ws[k].ensure((size_t)n * 4 + 4));
    for (int k : {RW_FSTATUS, RW_RSTATUS, RW_O5, RW_FTIER}) ZSW_HIP(ctx, ws[k].ensure((size_t)n + 4));
    ZSW_HIP(ctx, ws[RW_GTAB].ensure(ctx->ref_len * 8 + 8));
    ZSW_HIP(ctx, ws[RW_MIS].ensure(4));
    ScoreOut fo;
    fo.score = ws[RW_FSCORE].as<uint32_t>();
    fo.status = ws[RW_FSTATUS].as<uint8_t>();
    fo.tier = ws[RW_FTIER].as<uint8_t>();
    fo.ref_end = ws[RW_FREND].as<uint32_t>();
    fo.query_end = ws[RW_FQEND].as<uint32_t>();
    fo.fb_list = ctx->d_fb_list.as<uint32_t>();
    fo.fb_count = ctx->d_fb_count.as<uint32_t>();
    // The reverse pass as a second seeded pass (as the shared role does it, zsw_capi_shared.hip): when this batch takes the seeded
    // pass, the forward pass also reports whether a read's maximum sits in ONE cell (mode 3). Then every alignment that scores it
    // ends there, i.e. lies inside the prefixes the reverse pass of striped.rs:355-388 is restricted to, and the cells of the
    // reversed matrix that hold the score are the same with or without the restriction (tests/models/reverse_unique.cpp): a seeded
    // pass over the reversed reads and the reversed reference, whole sequences, finds them; if that is one cell too, it is the
    // start. Every other read goes to the exact reverse kernel below.
    const bool seeded_reverse = seeded_reverse_possible(ctx, st);
    if (cert && !seeded_reverse) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "internal: certificate pass without the seeded reverse pass");
    if (cert) fo.safe_row = cert->safe_row;
    if (seeded_reverse) {
        ZSW_HIP(ctx, ws[RW_UNIQ_F].ensure((size_t)n + 4));
        ZSW_HIP(ctx, hipMemsetAsync(ws[RW_UNIQ_F].p, 0, n, stream));
        fo.unique = ws[RW_UNIQ_F].as<uint8_t>();
    }
    hipError_t e = launch_score(ctx->d_sc.as<ScoringDev>(), ctx->h_sc, st.b, st.max_len, ctx->d_ref.as<uint8_t>(),
                                (uint32_t)ctx->ref_len, rule, fo, score_ws(ctx), stream, nullptr, seeded_reverse ? 3 : 2);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "ranges forward pass", e);
    const uint32_t g256 = (n + 255) / 256;
    hipLaunchKernelGGL(ranges_prep_kernel, dim3(g256), dim3(256), 0, stream, n, fo.status, fo.query_end, ws[RW_QEM].as<uint32_t>());
    ScoreOut ro;
    ro.score = ws[RW_RSCORE].as<uint32_t>();
    ro.status = ws[RW_RSTATUS].as<uint8_t>();
    ro.tier = nullptr;
    ro.ref_end = ws[RW_RRS].as<uint32_t>();
    ro.query_end = ws[RW_RQS].as<uint32_t>();
    ro.fb_list = ctx->d_fb_list.as<uint32_t>();
    ro.fb_count = ctx->d_fb_count.as<uint32_t>();
    BatchDev rest = st.b;
    bool run_exact = true;
    if (seeded_reverse) {
        const size_t R = ctx->ref_len;
        if (!ctx->seed_rev.valid) {
            std::vector<uint8_t> rev(ctx->h_ref.rbegin(), ctx->h_ref.rend());
            ZSW_HIP(ctx, ctx->d_ref_rev.ensure(R + 16));
            ZSW_HIP(ctx, hipMemcpyAsync(ctx->d_ref_rev.p, rev.data(), R, hipMemcpyHostToDevice, stream));
            ZSW_HIP(ctx, hipStreamSynchronize(stream));  // `rev` goes out of scope
            ZSW_HIP(ctx, seed_index_update(&ctx->seed_rev, ctx->h_sc, rev.data(), R));
        }
        if (ctx->seed_rev.usable) {
            ZSW_HIP(ctx, ws[RW_UNIQ_R].ensure((size_t)n + 4));
            ZSW_HIP(ctx, ws[RW_ULIST].ensure((size_t)n * 4 + 4));
            ZSW_HIP(ctx, ws[RW_UCOUNT].ensure(4));
            ZSW_HIP(ctx, hipMemsetAsync(ws[RW_UNIQ_R].p, 0, n, stream));
            BatchDev brev = st.b;  // the same bases: the seeded pass reads them back to front (ScoreOut::reads_reversed)
            ScoreOut o3 = ro;  // rows: positions of the reversed reference; columns: positions of the reversed read
            o3.unique = ws[RW_UNIQ_R].as<uint8_t>();
            o3.skip_handed_back = true;
            o3.reads_reversed = true;
            // the exact reverse kernel walks a read's prefixes for what the second tier costs per read, and always answers; only the
            // certificate of run_align, whose unsettled reads take the literal second pass, is worth the second tier
            o3.narrow_only = cert == nullptr;
            ScoreWorkspace w = score_ws(ctx);
            w.seed = &ctx->seed_rev;
            w.band_dbg = nullptr;
            hipError_t e3 = launch_score(ctx->d_sc.as<ScoringDev>(), ctx->h_sc, brev, st.max_len, ctx->d_ref_rev.as<uint8_t>(), (uint32_t)R, rule, o3, w, stream,
                                         nullptr, 3);
            if (e3 != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "ranges: seeded reverse pass", e3);
            ZSW_HIP(ctx, hipMemsetAsync(ws[RW_UCOUNT].p, 0, 4, stream));
            ZSW_HIP(ctx, launch_settle_reverse(st.b, n, (uint32_t)R, ws[RW_UNIQ_F].as<uint8_t>(), ws[RW_UNIQ_R].as<uint8_t>(), fo.score, fo.status, ro.score,
                                               ro.status, ro.query_end, ro.ref_end, ws[RW_ULIST].as<uint32_t>(), ws[RW_UCOUNT].as<uint32_t>(), stream,
                                               cert ? cert->settled : nullptr));
            if (cert) {  // the caller takes it from here (the unsettled reads go to sw_simd_align's own second pass)
                cert->fwd = fo;
                cert->rs = ro.ref_end;
                cert->qs = ro.query_end;
                return ZSW_OK;
            }
            uint32_t left = 0;
            ZSW_HIP(ctx, hipMemcpyAsync(&left, ws[RW_UCOUNT].p, 4, hipMemcpyDeviceToHost, stream));
            ZSW_HIP(ctx, hipStreamSynchronize(stream));
            rest.items = ws[RW_ULIST].as<uint32_t>();
            rest.n_items = left;
            run_exact = left > 0;
        } else if (cert) {
            ZSW_HIP(ctx, hipMemsetAsync(cert->settled, 0, n, stream));  // no index of the reversed reference: nothing is settled
            cert->fwd = fo;
            cert->rs = ro.ref_end;
            cert->qs = ro.query_end;
            return ZSW_OK;
        }
    }
    if (run_exact) {
        e = launch_score_rev(ctx->d_sc.as<ScoringDev>(), ctx->h_sc, rest, st.max_len, ctx->d_ref.as<uint8_t>(), (uint32_t)ctx->ref_len,
                             rule, ro, score_ws(ctx), fo.ref_end, ws[RW_QEM].as<uint32_t>(), fo.score, ws[RW_GTAB].as<uint2>(), stream);
        if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "ranges reverse pass", e);
    }
    out->score = ws[RW_O0].as<uint32_t>();
    out->rs = ws[RW_O1].as<uint32_t>();
    out->re = ws[RW_O2].as<uint32_t>();
    out->qs = ws[RW_O3].as<uint32_t>();
    out->qe = ws[RW_O4].as<uint32_t>();
    out->status = ws[RW_O5].as<uint8_t>();
    out->tier = fo.tier;
    ZSW_HIP(ctx, hipMemsetAsync(ws[RW_MIS].p, 0, 4, stream));
    hipLaunchKernelGGL(ranges_combine_kernel, dim3(g256), dim3(256), 0, stream, n, fo.score, fo.status, fo.ref_end, fo.query_end,
                       ro.score, ro.status, ro.ref_end, ro.query_end, out->score, out->rs, out->re, out->qs, out->qe, out->status,
                       ws[RW_MIS].as<uint32_t>());
    ZSW_HIP(ctx, hipGetLastError());
    return ZSW_OK;
}

zsw_error run_ranges(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, uint32_t* out_score, uint32_t* out_rs,
                     uint32_t* out_re, uint32_t* out_qs, uint32_t* out_qe, uint8_t* out_status, uint8_t* out_tier, void* stream_) {
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!reads || !out_score || !out_rs || !out_re || !out_qs || !out_qe || !out_status)
        return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    const uint32_t n = (uint32_t)reads->n_reads;
    Staged st;
    {
        uint32_t dummy_score = 0;
        uint8_t dummy_status = 0;
        zsw_error ze = stage(ctx, reads, stream, false, false, &dummy_score, &dummy_status, nullptr, nullptr, nullptr, &st);
        if (ze != ZSW_OK) return ze;
    }
    if (n == 0) return ZSW_OK;
    RangesDev rd;
    ctx->timer.begin(stream);
    zsw_error ze = ranges_device(ctx, st, rule, stream, &rd);
    ctx->timer.end(stream);
    if (ze != ZSW_OK) return ze;
    const hipMemcpyKind kind = reads->mem == ZSW_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    uint32_t* outs[5] = {out_score, out_rs, out_re, out_qs, out_qe};
    uint32_t* devs[5] = {rd.score, rd.rs, rd.re, rd.qs, rd.qe};
    for (int k = 0; k < 5; ++k) ZSW_HIP(ctx, hipMemcpyAsync(outs[k], devs[k], (size_t)n * 4, kind, stream));
    ZSW_HIP(ctx, hipMemcpyAsync(out_status, rd.status, n, kind, stream));
    if (out_tier) ZSW_HIP(ctx, hipMemcpyAsync(out_tier, rd.tier, n, kind, stream));
    if (reads->mem == ZSW_MEM_HOST) ZSW_HIP(ctx, hipStreamSynchronize(stream));
    return ZSW_OK;
}


// statuses as sw_simd_align's second pass sees them: a read whose alignment the certificate pass wrote does not take part
__global__ void cert_status_kernel(uint32_t n, const uint8_t* status, const uint8_t* done, uint8_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = done[i] ? (uint8_t)ZSW_STATUS_UNMAPPED : status[i];
}

hipError_t launch_cert_status(uint32_t n, const uint8_t* status, const uint8_t* done, uint8_t* out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(cert_status_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, status, done, out);
    return hipGetLastError();
}

// (grid-stride, one atomic per wavefront at the end: 10 M statuses used to be 156,000 atomics on one word, 1.8 ms)
__global__ void count_some_kernel(const uint8_t* status, uint32_t n, uint32_t* out) {
    uint32_t mine = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) mine += status[i] == ZSW_STATUS_SOME ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

// (tier, length) of the listed reads
__global__ void gather_meta_kernel(BatchDev b, const uint8_t* tier, const uint32_t* ids, uint32_t n_ids, uint32_t* meta) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_ids) return;
    const uint32_t id = ids[k];
    meta[2 * k] = tier[id];
    meta[2 * k + 1] = b.offsets ? (uint32_t)(b.offsets[id + 1] - b.offsets[id]) : b.fixed_len;
}

// Shared tail of the alignment entry points: count + scan + packed write of the ciglets (reversal and SeqSrc::Query
// inversion happen in the writer), then the records/statuses to the caller's arrays.
zsw_error finish_alignments(zsw_context* ctx, DevBuf* ws, uint32_t n, bool host, const uint8_t* d_status, const uint8_t* d_tier,
                            int invert, zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                            uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, hipStream_t stream) {
    const uint32_t nblocks = (n + 1023) / 1024;
    ZSW_HIP(ctx, ws[WS_BSUMS].ensure((size_t)nblocks * 8 + 8));
    ZSW_HIP(ctx, ws[WS_TOTAL].ensure(8));
    hipError_t e = align_finalize(ws[WS_ALN].as<zsw_alignment>(), d_status, n, ws[WS_BSUMS].as<uint64_t>(), ws[WS_TOTAL].as<uint64_t>(),
                                  ws[WS_CIGSTART].as<uint64_t>(), ws[WS_CIGRAW].as<uint32_t>(), invert, nullptr, nullptr, 0, true, stream);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "align count", e);
    uint64_t total = 0;
    ZSW_HIP(ctx, hipMemcpyAsync(&total, ws[WS_TOTAL].p, 8, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    *out_n_ciglets = total;
    if (total > ciglet_cap) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "ciglet capacity too small; required size returned");
    uint32_t* d_inc = out_inc;
    uint8_t* d_op = out_op;
    if (host) {
        ZSW_HIP(ctx, ws[WS_OINC].ensure(total * 4 + 4));
        ZSW_HIP(ctx, ws[WS_OOP].ensure(total + 4));
        d_inc = ws[WS_OINC].as<uint32_t>();
        d_op = ws[WS_OOP].as<uint8_t>();
    }
    e = align_finalize(ws[WS_ALN].as<zsw_alignment>(), d_status, n, ws[WS_BSUMS].as<uint64_t>(), ws[WS_TOTAL].as<uint64_t>(),
                       ws[WS_CIGSTART].as<uint64_t>(), ws[WS_CIGRAW].as<uint32_t>(), invert, d_inc, d_op, ciglet_cap, false, stream);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "align write", e);
    const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    ZSW_HIP(ctx, hipMemcpyAsync(out_aln, ws[WS_ALN].p, (size_t)n * sizeof(zsw_alignment), kind, stream));
    ZSW_HIP(ctx, hipMemcpyAsync(out_status, d_status, n, kind, stream));
    if (out_tier && d_tier) ZSW_HIP(ctx, hipMemcpyAsync(out_tier, d_tier, n, kind, stream));
    if (host && total) {
        ZSW_HIP(ctx, hipMemcpyAsync(out_inc, d_inc, total * 4, kind, stream));
        ZSW_HIP(ctx, hipMemcpyAsync(out_op, d_op, total, kind, stream));
    }
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    return ZSW_OK;
}

// lanes_w*: lane count of the profile whose width answered (direct call: the caller's N for its one width).
zsw_error run_align(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, int lanes_w8, int lanes_w16, int lanes_w32,
                    int invert, zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                    uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream_) {
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!reads || !out_aln || !out_status || !out_n_ciglets || (ciglet_cap && (!out_inc || !out_op)))
        return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    const uint32_t n = (uint32_t)reads->n_reads;
    Staged st;
    {
        uint32_t dummy_score = 0;
        uint8_t dummy_status = 0;
        zsw_error ze = stage(ctx, reads, stream, false, false, &dummy_score, &dummy_status, nullptr, nullptr, nullptr, &st);
        if (ze != ZSW_OK) return ze;
    }
    *out_n_ciglets = 0;
    if (n == 0) return ZSW_OK;
    const bool host = reads->mem == ZSW_MEM_HOST;
    DevBuf* ws = ctx->a_ws;
    ZSW_HIP(ctx, ws[WS_SCORE].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ws[WS_STATUS].ensure((size_t)n + 4));
    ZSW_HIP(ctx, ws[WS_TIER].ensure((size_t)n + 4));
    ZSW_HIP(ctx, ws[WS_REND].ensure((size_t)n * 4 + 4));
    // pass 1: packed score kernel with the reference-end row (sw_simd_align's `best` and `r_end`)
    ScoreOut so;
    so.score = ws[WS_SCORE].as<uint32_t>();
    so.status = ws[WS_STATUS].as<uint8_t>();
    so.tier = ws[WS_TIER].as<uint8_t>();
    so.ref_end = ws[WS_REND].as<uint32_t>();
    so.query_end = nullptr;
    so.fb_list = ctx->d_fb_list.as<uint32_t>();
    so.fb_count = ctx->d_fb_count.as<uint32_t>();
    // the seeded first pass leaves, per read it accepts, the row from which pass 2 may start with a zero state
    ZSW_HIP(ctx, ws[WS_SAFE].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, hipMemsetAsync(ws[WS_SAFE].p, 0xff, (size_t)n * 4, stream));
    so.safe_row = (ctx->flags() & ZSW_DEBUG_ALIGN_LONG_WARMUP) ? nullptr : ws[WS_SAFE].as<uint32_t>();
    // Certificate mode (tests/models/align_gapless_cert.cpp): the first pass is the forward and the reversed seeded pass of the
    // ranges (mode 3: is the maximum in one cell?); a read with exactly one optimal alignment, a gapless diagonal, gets it from the
    // classify pass of zsw_threepass.hip and never sees the literal striped recurrence below.
    const uint8_t* pass2_status = nullptr;  // statuses as the grouping sees them: certified reads do not take part
    hipError_t e = hipSuccess;
    const bool certify = seeded_reverse_possible(ctx, st) && !(ctx->flags() & ZSW_DEBUG_ALIGN_NO_CERTIFICATE) && ctx->h_sc.gap_open > 0;
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
        RangesCert cert;
        cert.safe_row = so.safe_row;
        cert.settled = ws[WS_CERT_OK].as<uint8_t>();
        RangesDev unused;
        zsw_error ze = ranges_device(ctx, st, rule, stream, &unused, &cert);
        if (ze != ZSW_OK) return ze;
        so.score = cert.fwd.score;
        so.status = cert.fwd.status;
        so.tier = cert.fwd.tier;
        so.ref_end = cert.fwd.ref_end;
        int maxw = 0;
        for (int i = 0; i < ctx->h_sc.S * ctx->h_sc.S; ++i) maxw = std::max(maxw, (int)ctx->h_sc.w[i]);
        ThreePassArgs a;
        a.b = st.b;
        a.ref = ctx->d_ref.as<uint8_t>();
        a.ref_len = (uint32_t)ctx->ref_len;
        a.sc = ctx->d_sc.as<ScoringDev>();
        a.score = so.score;
        a.rs = cert.rs;
        a.re = cert.fwd.ref_end;
        a.qs = cert.qs;
        a.qe = cert.fwd.query_end;
        a.status = so.status;
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
        a.cert_ok = cert.settled;
        a.cert_done = ws[WS_CERT_DONE].as<uint8_t>();
        a.cert_maxw = maxw;
        a.cert_go = ctx->h_sc.gap_open;
        a.cert_ge = ctx->h_sc.gap_extend;
        // reads that need the sweeps over their two-run alternatives are listed and take a second, dense launch (ThreePassArgs::sweep_list)
        ZSW_HIP(ctx, ws[WS_ITEMS].ensure((size_t)n * 4 + 4));
        ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].as<uint32_t>() + 2, 0, 4, stream));
        a.sweep_list = ws[WS_ITEMS].as<uint32_t>();
        a.sweep_count = ws[WS_FBCOUNT].as<uint32_t>() + 2;
        e = launch_threepass(a, std::min<uint32_t>((n + 63) / 64, 65536u), stream);
        if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "align: certificate pass", e);
        a.list = a.sweep_list;
        a.list_count = a.sweep_count;
        a.sweep_pass = true;
        e = launch_threepass(a, std::min<uint32_t>((n / 4 + 63) / 64 + 1, 16384u), stream);
        if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "align: certificate pass (sweeps)", e);
        hipLaunchKernelGGL(cert_status_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, so.status, ws[WS_CERT_DONE].as<uint8_t>(), ws[WS_CERT_STATUS].as<uint8_t>());
        pass2_status = ws[WS_CERT_STATUS].as<uint8_t>();
    } else {
        e = launch_score(ctx->d_sc.as<ScoringDev>(), ctx->h_sc, st.b, st.max_len, ctx->d_ref.as<uint8_t>(), (uint32_t)ctx->ref_len, rule, so, score_ws(ctx), stream,
                         nullptr, 1);
        if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "align pass 1", e);
        pass2_status = so.status;
    }

    // Group the reads that have an alignment by the <N, nv> of the instantiation that answered and order each group by its
    // reference end row — on the device (zsw_group.hip); only the table of group starts comes back to the host.
    // Reads that share a wavefront walk the reference rows together: ordering by r_end makes their high-scoring rows (where
    // Zoe's lazy-F loop runs long) coincide and bounds the rows a wave computes by nearly the same r_end for all its reads.
    struct GroupRun {
        int N;
        uint32_t nv, start, count;
        bool wide;  // scores beyond the packed kernel's 16-bit lanes
    };
    const uint32_t TABLE_CAP = 1u << 16;
    ZSW_HIP(ctx, ws[WS_ITEMS].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ws[WS_KEYS_IN].ensure((size_t)n * 8 + 8));
    ZSW_HIP(ctx, ws[WS_KEYS_OUT].ensure((size_t)n * 8 + 8));
    ZSW_HIP(ctx, ws[WS_VALS_IN].ensure((size_t)n * 4 + 4));
    const size_t sort_bytes = group_temp_bytes(n);
    ZSW_HIP(ctx, ws[WS_SORT_TMP].ensure(sort_bytes + 256));
    ZSW_HIP(ctx, ws[WS_GTABLE].ensure((size_t)TABLE_CAP * 8 + 8));
    e = group_reads(st.b, pass2_status, so.tier, so.ref_end, so.score, so.safe_row, lanes_w8, lanes_w16, lanes_w32, ws[WS_KEYS_IN].as<uint64_t>(),
                    ws[WS_KEYS_OUT].as<uint64_t>(), ws[WS_VALS_IN].as<uint32_t>(), ws[WS_ITEMS].as<uint32_t>(), ws[WS_SORT_TMP].p,
                    sort_bytes, ws[WS_GTABLE].as<uint32_t>() + 2, ws[WS_GTABLE].as<uint32_t>(), TABLE_CAP - 1, stream);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "align grouping", e);
    uint32_t n_groups = 0;
    ZSW_HIP(ctx, hipMemcpyAsync(&n_groups, ws[WS_GTABLE].p, 4, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    if (n_groups > TABLE_CAP - 1) return fail(ctx, ZSW_ERR_UNSUPPORTED, "more than 65535 distinct <lanes, vectors> groups in one call");
    std::vector<uint32_t> h_table((size_t)n_groups * 2 + 2);
    uint32_t n_some = 0;
    if (n_groups) ZSW_HIP(ctx, hipMemcpy(h_table.data(), ws[WS_GTABLE].as<uint32_t>() + 2, (size_t)n_groups * 8, hipMemcpyDeviceToHost));
    std::vector<GroupRun> groups;
    {
        // sort the table by start (atomic slots are unordered) and derive the counts; the last group ends where the reads
        // without an alignment (key ~0) begin, i.e. at the number of SOME statuses (counted on the device)
        std::vector<std::pair<uint32_t, uint32_t>> t;  // (start, N<<24 | wide<<23 | nv)
        for (uint32_t k = 0; k < n_groups; ++k) t.emplace_back(h_table[2 * k + 1], h_table[2 * k]);
        std::sort(t.begin(), t.end());
        if (!t.empty()) {
            ZSW_HIP(ctx, ws[WS_FBCOUNT].ensure(8));  // fallback counter + the packed kernel's work counter
            ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].p, 0, 4, stream));
            hipLaunchKernelGGL(count_some_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 1024u)), dim3(256), 0, stream, pass2_status, n, ws[WS_FBCOUNT].as<uint32_t>());
            ZSW_HIP(ctx, hipMemcpyAsync(&n_some, ws[WS_FBCOUNT].p, 4, hipMemcpyDeviceToHost, stream));
            ZSW_HIP(ctx, hipStreamSynchronize(stream));
        }
        for (size_t k = 0; k < t.size(); ++k) {
            const uint32_t end = k + 1 < t.size() ? t[k + 1].first : n_some;
            groups.push_back(GroupRun{(int)(t[k].second >> 24), t[k].second & 0x7fffffu, t[k].first, end - t[k].first, ((t[k].second >> 23) & 1u) != 0});
        }
    }
    const int S = ctx->h_sc.S;
    for (auto& g : groups)
        if (align_lds_need(g.nv, S) + 9216 > 160 * 1024)
            return fail(ctx, ZSW_ERR_UNSUPPORTED, "read too long for the alignment kernel's LDS-resident profile");

    const uint32_t MAXC = 32;
    ZSW_HIP(ctx, ws[WS_ALN].ensure((size_t)n * sizeof(zsw_alignment)));
    ZSW_HIP(ctx, ws[WS_CIGSTART].ensure((size_t)n * 8));
    ZSW_HIP(ctx, ws[WS_CIGRAW].ensure((size_t)n * 4));
    ZSW_HIP(ctx, ws[WS_FBLIST].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ws[WS_FBCOUNT].ensure(8));  // fallback counter + the packed kernel's work counter
    ZSW_HIP(ctx, ws[WS_CIG].ensure((size_t)n * MAXC * 4));
    ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].p, 0, 4, stream));

    auto window_of = [&](const GroupRun& g, bool full) {
        const uint32_t lpad = g.nv * (uint32_t)g.N;
        uint32_t W = full ? (uint32_t)ctx->ref_len : std::min<uint32_t>((uint32_t)ctx->ref_len, lpad + std::max<uint32_t>(16, lpad / 8));
        return W ? W : 1u;
    };
    // the packed kernel (two reads per lane group, zsw_align_pk_kernel.hpp) answers the groups it covers; the 32-bit kernels
    // take scores beyond 16-bit lanes, more than 32 vectors, large alphabets and the full-window reruns
    auto packed = [&](const GroupRun& g, bool full) {
        return !full && !g.wide && !(ctx->flags() & ZSW_DEBUG_ALIGN_NO_PACKED) && align_pk_supported(g.N, g.nv, S);
    };
    auto ring_bytes_of = [&](const GroupRun& g, bool full, uint32_t grid) {
        return packed(g, full) ? align_pk_ring_bytes(g.N, g.nv, window_of(g, full), grid) : align_ring_bytes(g.N, g.nv, window_of(g, full), grid);
    };
    auto grid_of = [&](const GroupRun& g, bool full) {
        uint32_t grid;
        if (packed(g, full)) {
            grid = align_pk_grid(g.N, g.nv, S, g.count, ctx->cu_count);
        } else {
            const uint32_t rpw = 64 / (uint32_t)g.N;
            grid = std::min<uint32_t>((g.count + rpw - 1) / rpw, full ? 256u : 4096u);
        }
        while (grid > 1 && ring_bytes_of(g, full, grid) > (size_t(3) << 30)) grid /= 2;
        return grid;
    };
    // one ring per pass, shared by its launches (they run in order on one stream): size it for the largest group up front
    auto run_groups = [&](const std::vector<GroupRun>& gs, const uint32_t* d_items, bool full, DevBuf& ringbuf, DevBuf& cigbuf) -> zsw_error {
        if (gs.empty()) return ZSW_OK;
        size_t ring_need = 0;
        uint64_t pool_need = 0;
        for (auto& g : gs) {
            ring_need = std::max(ring_need, ring_bytes_of(g, full, grid_of(g, full)));
            pool_need += (uint64_t)g.count * ((uint64_t)g.N * g.nv + ctx->ref_len + 4);
        }
        ZSW_HIP(ctx, ringbuf.ensure(ring_need + 64));
        if (full) ZSW_HIP(ctx, cigbuf.ensure(pool_need * 4));
        uint64_t pool = 0;
        for (auto& g : gs) {
            const uint32_t W = window_of(g, full);
            const uint32_t maxc = full ? g.nv * (uint32_t)g.N + (uint32_t)ctx->ref_len + 4 : MAXC;
            BatchDev b = st.b;
            b.items = d_items + g.start;
            b.n_items = g.count;
            hipError_t he = (packed(g, full) ? align_pass2_pk : align_pass2)(g.N, g.nv, b, ctx->d_ref.as<uint8_t>(), (uint32_t)ctx->ref_len, ctx->d_sc.as<ScoringDev>(), S,
                                        so.score, so.ref_end, so.status, W, maxc, ringbuf.as<uint8_t>(), grid_of(g, full),
                                        cigbuf.as<uint32_t>(), pool, full ? 1 : 0, ws[WS_CIGSTART].as<uint64_t>(),
                                        ws[WS_CIGRAW].as<uint32_t>(), ws[WS_ALN].as<zsw_alignment>(),
                                        ws[WS_FBLIST].as<uint32_t>(), ws[WS_FBCOUNT].as<uint32_t>(), invert, stream, full ? nullptr : so.safe_row);
            if (he != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "align pass 2", he);
            if (full) pool += (uint64_t)g.count * maxc;
        }
        return ZSW_OK;
    };
    ctx->timer.begin(stream);
    zsw_error ze = run_groups(groups, ws[WS_ITEMS].as<uint32_t>(), false, ws[WS_RING], ws[WS_CIG]);
    ctx->timer.end(stream);
    if (ze != ZSW_OK) return ze;
    // reads whose traceback left the retained window (or overflowed their ciglet slots): rerun keeping every row
    uint32_t n_fb = 0;
    ZSW_HIP(ctx, hipMemcpyAsync(&n_fb, ws[WS_FBCOUNT].p, 4, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    if (n_fb) {
        // few reads: fetch their (tier, length) and group them on the host
        ZSW_HIP(ctx, ws[WS_FBMETA].ensure((size_t)n_fb * 8));
        hipLaunchKernelGGL(gather_meta_kernel, dim3((n_fb + 255) / 256), dim3(256), 0, stream, st.b, so.tier, ws[WS_FBLIST].as<uint32_t>(),
                           n_fb, ws[WS_FBMETA].as<uint32_t>());
        std::vector<uint32_t> fb(n_fb), meta((size_t)n_fb * 2);
        ZSW_HIP(ctx, hipMemcpyAsync(fb.data(), ws[WS_FBLIST].p, (size_t)n_fb * 4, hipMemcpyDeviceToHost, stream));
        ZSW_HIP(ctx, hipMemcpyAsync(meta.data(), ws[WS_FBMETA].p, (size_t)n_fb * 8, hipMemcpyDeviceToHost, stream));
        ZSW_HIP(ctx, hipStreamSynchronize(stream));
        std::map<std::pair<int, uint32_t>, std::vector<uint32_t>> g2;
        for (uint32_t k = 0; k < n_fb; ++k) {
            const int N = meta[2 * k] == 8 ? lanes_w8 : meta[2 * k] == 16 ? lanes_w16 : lanes_w32;
            g2[std::make_pair(N, (meta[2 * k + 1] + (uint32_t)N - 1) / (uint32_t)N)].push_back(fb[k]);
        }
        std::vector<uint32_t> ids;
        std::vector<GroupRun> runs;
        for (auto& g : g2) {
            runs.push_back(GroupRun{g.first.first, g.first.second, (uint32_t)ids.size(), (uint32_t)g.second.size(), true});
            ids.insert(ids.end(), g.second.begin(), g.second.end());
        }
        ZSW_HIP(ctx, ws[WS_ITEMS2].ensure(ids.size() * 4 + 4));
        ZSW_HIP(ctx, hipMemcpy(ws[WS_ITEMS2].p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
        ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].p, 0, 4, stream));
        ze = run_groups(runs, ws[WS_ITEMS2].as<uint32_t>(), true, ws[WS_RING2], ws[WS_CIG2]);
        if (ze != ZSW_OK) return ze;
        uint32_t again = 0;
        ZSW_HIP(ctx, hipMemcpyAsync(&again, ws[WS_FBCOUNT].p, 4, hipMemcpyDeviceToHost, stream));
        ZSW_HIP(ctx, hipStreamSynchronize(stream));
        if (again) return fail(ctx, ZSW_ERR_HIP, "alignment traceback did not complete with the full window");
    }

    return finish_alignments(ctx, ws, n, host, so.status, so.tier, invert, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap,
                             out_n_ciglets, stream);
}

// sw_align_3pass per read (three_pass.rs:21-104): ranges on the device, then the third pass (zsw_threepass.hip)
zsw_error run_threepass(zsw_context* ctx, const zsw_batch* reads, const ResultRule& rule, int invert, zsw_alignment* out_aln,
                        uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap,
                        uint64_t* out_n_ciglets, void* stream_) {
    DeviceGuard device_guard(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!reads || !out_aln || !out_status || !out_n_ciglets || (ciglet_cap && (!out_inc || !out_op)))
        return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    const uint32_t n = (uint32_t)reads->n_reads;
    Staged st;
    {
        uint32_t dummy_score = 0;
        uint8_t dummy_status = 0;
        zsw_error ze = stage(ctx, reads, stream, false, false, &dummy_score, &dummy_status, nullptr, nullptr, nullptr, &st);
        if (ze != ZSW_OK) return ze;
    }
    *out_n_ciglets = 0;
    if (n == 0) return ZSW_OK;
    RangesDev rd;
    ctx->timer.begin(stream);
    zsw_error ze = ranges_device(ctx, st, rule, stream, &rd);
    if (ze != ZSW_OK) return ze;
    return threepass_third_pass(ctx, st, rd, nullptr, 0, reads->mem == ZSW_MEM_HOST, invert, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap,
                                out_n_ciglets, stream);
}

// The third pass over ranges that are already on the device (the caller opened ctx->timer's interval). pseq: non-null = the
// shared-profile role (ThreePassArgs::pseq).
zsw_error threepass_third_pass(zsw_context* ctx, const Staged& st, const RangesDev& rd, const uint8_t* pseq, uint32_t pseq_len, bool host, int invert,
                               zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap,
                               uint64_t* out_n_ciglets, hipStream_t stream) {
    const uint32_t n = st.b.n_reads;
    zsw_error ze = ZSW_OK;
    DevBuf* ws = ctx->a_ws;
    const uint32_t MAXC = 32;
    ZSW_HIP(ctx, ws[WS_ALN].ensure((size_t)n * sizeof(zsw_alignment)));
    ZSW_HIP(ctx, ws[WS_CIGSTART].ensure((size_t)n * 8));
    ZSW_HIP(ctx, ws[WS_CIGRAW].ensure((size_t)n * 4));
    ZSW_HIP(ctx, ws[WS_FBLIST].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ws[WS_ITEMS].ensure((size_t)n * 4 + 4));
    ZSW_HIP(ctx, ws[WS_FBCOUNT].ensure(16));
    ZSW_HIP(ctx, ws[WS_CIG].ensure((size_t)n * MAXC * 4));
    ZSW_HIP(ctx, hipMemsetAsync(ws[WS_FBCOUNT].p, 0, 16, stream));
    uint32_t* counters = ws[WS_FBCOUNT].as<uint32_t>();  // [0] rerun count, [1] dp count, [2] largest slot need
    ThreePassArgs a;
    a.b = st.b;
    a.ref = ctx->d_ref.as<uint8_t>();
    a.ref_len = (uint32_t)ctx->ref_len;
    a.sc = ctx->d_sc.as<ScoringDev>();
    a.score = rd.score;
    a.rs = rd.rs;
    a.re = rd.re;
    a.qs = rd.qs;
    a.qe = rd.qe;
    a.status = rd.status;
    a.list = nullptr;
    a.list_count = nullptr;
    a.dp_list = ws[WS_ITEMS].as<uint32_t>();
    a.dp_count = counters + 1;
    a.dp_need_max = counters + 2;
    a.scratch = nullptr;
    a.slots = 0;
    a.slot_bytes = 0;
    a.cig = ws[WS_CIG].as<uint32_t>();
    a.maxc = MAXC;
    a.pool_base = 0;
    a.by_item = 0;
    a.cig_start = ws[WS_CIGSTART].as<uint64_t>();
    a.cig_raw = ws[WS_CIGRAW].as<uint32_t>();
    a.aln = ws[WS_ALN].as<zsw_alignment>();
    a.fb_list = ws[WS_FBLIST].as<uint32_t>();
    a.fb_count = counters;
    a.invert = invert;
    a.pseq = pseq;
    a.pseq_len = pseq_len;
    hipError_t e = launch_threepass(a, std::min<uint32_t>((n + 63) / 64, 65536u), stream);  // classify + no-gaps shortcut
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "3-pass classify", e);
    uint32_t h_cnt[3] = {0, 0, 0};
    ZSW_HIP(ctx, hipMemcpyAsync(h_cnt, counters, 12, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    auto dp_pass = [&](const uint32_t* d_list, uint32_t* d_list_count, uint32_t count, uint64_t slot_bytes, uint32_t maxc, int by_item,
                       DevBuf& cigbuf, DevBuf& scratchbuf) -> zsw_error {
        if (!count) return ZSW_OK;
        slot_bytes = (slot_bytes + 63) & ~63ull;
        uint64_t budget = 2ull << 30;
        uint32_t slots = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(budget / slot_bytes, 64), 131072);
        slots = std::min<uint32_t>(slots, (count + 63) / 64 * 64);
        slots = std::max<uint32_t>(64, slots / 64 * 64);
        ZSW_HIP(ctx, scratchbuf.ensure((uint64_t)slots * slot_bytes + 64));
        if (by_item) ZSW_HIP(ctx, cigbuf.ensure((uint64_t)count * maxc * 4 + 64));
        ThreePassArgs d = a;
        d.list = d_list;
        d.list_count = d_list_count;
        d.scratch = scratchbuf.as<uint8_t>();
        d.slots = slots;
        d.slot_bytes = slot_bytes;
        d.cig = cigbuf.as<uint32_t>();
        d.maxc = maxc;
        d.by_item = by_item;
        hipError_t he = launch_threepass(d, slots / 64, stream);
        if (he != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "3-pass DP", he);
        return ZSW_OK;
    };
    const uint64_t SLOT_CAP = 96 * 1024;
    ze = dp_pass(ws[WS_ITEMS].as<uint32_t>(), counters + 1, h_cnt[1], std::min<uint64_t>(std::max<uint32_t>(h_cnt[2], 64), SLOT_CAP), MAXC, 0,
                 ws[WS_CIG], ws[WS_RING]);
    if (ze != ZSW_OK) return ze;
    ZSW_HIP(ctx, hipMemcpyAsync(h_cnt, counters, 4, hipMemcpyDeviceToHost, stream));
    ZSW_HIP(ctx, hipStreamSynchronize(stream));
    if (h_cnt[0]) {
        // reads whose box is larger than the slot cap or whose CIGAR needs more than 32 ciglets: rerun with full-size resources
        const uint32_t n_fb = h_cnt[0];
        if ((uint64_t)h_cnt[2] > (1ull << 31)) return fail(ctx, ZSW_ERR_UNSUPPORTED, "3-pass bounding box too large");
        ZSW_HIP(ctx, ws[WS_ITEMS2].ensure((size_t)n_fb * 4 + 4));
        ZSW_HIP(ctx, hipMemcpyAsync(ws[WS_ITEMS2].p, ws[WS_FBLIST].p, (size_t)n_fb * 4, hipMemcpyDeviceToDevice, stream));
        ZSW_HIP(ctx, ws[WS_FBMETA].ensure(8));
        ZSW_HIP(ctx, hipMemcpyAsync(ws[WS_FBMETA].p, &n_fb, 4, hipMemcpyHostToDevice, stream));
        ZSW_HIP(ctx, hipMemsetAsync(counters, 0, 4, stream));
        const uint32_t maxc_fb = st.max_len + (pseq ? pseq_len : (uint32_t)ctx->ref_len) + 4;
        ze = dp_pass(ws[WS_ITEMS2].as<uint32_t>(), ws[WS_FBMETA].as<uint32_t>(), n_fb, std::max<uint32_t>(h_cnt[2], 64), maxc_fb, 1,
                     ws[WS_CIG2], ws[WS_RING2]);
        if (ze != ZSW_OK) return ze;
        uint32_t again = 0;
        ZSW_HIP(ctx, hipMemcpyAsync(&again, counters, 4, hipMemcpyDeviceToHost, stream));
        ZSW_HIP(ctx, hipStreamSynchronize(stream));
        if (again) return fail(ctx, ZSW_ERR_HIP, "3-pass alignment did not complete with full-size resources");
    }
    ctx->timer.end(stream);
    return finish_alignments(ctx, ws, n, host, rd.status, rd.tier, invert, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap,
                             out_n_ciglets, stream);
}

hipError_t launch_ranges_prep(uint32_t n, const uint8_t* fstatus, const uint32_t* fqend, uint32_t* qe_masked, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(ranges_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, fstatus, fqend, qe_masked);
    return hipGetLastError();
}

hipError_t launch_ranges_combine(uint32_t n, const uint32_t* fscore, const uint8_t* fstatus, const uint32_t* frend, const uint32_t* fqend,
                                 const uint32_t* rscore, const uint8_t* rstatus, const uint32_t* rrstart, const uint32_t* rqstart,
                                 uint32_t* out_score, uint32_t* out_rs, uint32_t* out_re, uint32_t* out_qs, uint32_t* out_qe,
                                 uint8_t* out_status, uint32_t* mismatch, hipStream_t stream) {
    if (n)
        hipLaunchKernelGGL(ranges_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, fscore, fstatus, frend, fqend, rscore, rstatus,
                           rrstart, rqstart, out_score, out_rs, out_re, out_qs, out_qe, out_status, mismatch);
    return hipGetLastError();
}

}  // namespace capi
}  // namespace zsw

extern "C" {

zsw_error zsw_align_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert,
                          zsw_alignment* out_aln, uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op,
                          uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_align(ctx, reads, rule, lanes, lanes, lanes, invert, out_aln, out_status, nullptr, out_inc, out_op, ciglet_cap,
                     out_n_ciglets, stream);
}

zsw_error zsw_align_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                               zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                               uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_align(ctx, reads, rule, preset_bits / 8, preset_bits / 16, preset_bits / 32, invert, out_aln, out_status, out_tier,
                     out_inc, out_op, ciglet_cap, out_n_ciglets, stream);
}

int zsw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static thread_local std::string g_create_error;

zsw_error zsw_create(int device_id, zsw_context** out) {
    if (!out) return ZSW_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("hipGetDeviceCount: ") + (e != hipSuccess ? hipGetErrorString(e) : "0 devices");
        return ZSW_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "device_id out of range";
        return ZSW_ERR_INVALID_ARGUMENT;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return ZSW_ERR_NO_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {  // kernels are built for gfx950 only
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950";
        return ZSW_ERR_NO_DEVICE;
    }
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    e = hipSetDevice(device_id);  // fails here, not in the first batch call, if the device cannot be used
    if (prev_device >= 0 && prev_device != device_id) (void)hipSetDevice(prev_device);  // the caller's current device stays
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return ZSW_ERR_NO_DEVICE;
    }
    zsw_context* c = new (std::nothrow) zsw_context();
    if (!c) return ZSW_ERR_INVALID_ARGUMENT;
    c->device = device_id;
    c->cu_count = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    *out = c;
    return ZSW_OK;
}

void zsw_destroy(zsw_context* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    DevBuf* bufs[] = {&ctx->d_sc, &ctx->d_ref, &ctx->d_fb_list, &ctx->d_fb_count, &ctx->d_scratch, &ctx->d_maxlen, &ctx->d_bucket_items, &ctx->d_bucket_counts, &ctx->d_tile_buf, &ctx->d_tile_state, &ctx->d_prune, &ctx->d_prune_list, &ctx->d_prune_count, &ctx->d_seed_work, &ctx->d_seed_gtab, &ctx->d_chunk_keys, &ctx->d_pseq, &ctx->d_pseq_rev, &ctx->d_sc_t,
                      &ctx->s_bases, &ctx->s_packed, &ctx->s_offsets, &ctx->s_score, &ctx->s_status, &ctx->s_tier, &ctx->s_rend, &ctx->s_qend};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : ctx->a_ws) b.release();
    for (DevBuf& b : ctx->r_ws) b.release();
    for (DevBuf& b : ctx->sh_ws) b.release();
    seed_index_release(&ctx->seed);
    seed_index_release(&ctx->seed_rev);
    ctx->d_ref_rev.release();
    seed_index_release(&ctx->seed_shared);
    seed_index_release(&ctx->seed_shared_rev);
    ctx->timer.destroy();
    ctx->timer_window.destroy();
    if (ctx->side) {
        (void)hipEventDestroy(ctx->side->fork);
        for (int i = 0; i < SideStreams::N; ++i) {
            (void)hipStreamDestroy(ctx->side->s[i]);
            (void)hipEventDestroy(ctx->side->join[i]);
        }
        delete ctx->side;
    }
    for (hipEvent_t ev : ctx->copy_events) (void)hipEventDestroy(ev);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

// ctx == NULL: the reason the last zsw_create on this thread failed
const char* zsw_last_error_string(const zsw_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

zsw_error zsw_set_scoring(zsw_context* ctx, const int8_t* weights, int S, const uint8_t* index_map, int gap_open,
                          int gap_extend) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !weights || !index_map) return ZSW_ERR_INVALID_ARGUMENT;
    if (S < 1 || S > MAX_S) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "S out of range");
    // validate_profile_args (profile.rs:32-44), sequence checks are per read
    if (gap_open < -127 || gap_open > 0) return ZSW_ERR_GAP_OPEN_OUT_OF_RANGE;
    if (gap_extend < -127 || gap_extend > 0) return ZSW_ERR_GAP_EXTEND_OUT_OF_RANGE;
    if (gap_extend < gap_open) return ZSW_ERR_BAD_GAP_WEIGHTS;
    for (int b = 0; b < 256; ++b)
        if (index_map[b] >= S) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "index_map entry >= S");
    ScoringDev& s = ctx->h_sc;
    memset(&s, 0, sizeof(s));
    memcpy(s.index_map, index_map, 256);
    int mn = 0;
    for (int i = 0; i < S * S; ++i) {
        s.w[i] = weights[i];
        mn = std::min<int>(mn, weights[i]);
    }
    s.S = S;
    s.gap_open = -gap_open;
    s.gap_extend = -gap_extend;
    ctx->bias = -mn;  // WeightMatrix::get_bias / to_biased_matrix (matrices/mod.rs:452-491)
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    // asynchronous score calls (on any stream, blocking or not) may still be reading the previous tables
    if (ctx->scoring_set) ZSW_HIP(ctx, hipDeviceSynchronize());
    ZSW_HIP(ctx, ctx->d_sc.ensure(sizeof(ScoringDev)));
    ZSW_HIP(ctx, hipMemcpy(ctx->d_sc.p, &s, sizeof(ScoringDev), hipMemcpyHostToDevice));
    {  // the same scoring with the matrix transposed, for the score-only calls of the shared-profile role
        ScoringDev t = s;
        for (int r = 0; r < S; ++r)
            for (int q = 0; q < S; ++q) t.w[r * S + q] = s.w[q * S + r];
        ZSW_HIP(ctx, ctx->d_sc_t.ensure(sizeof(ScoringDev)));
        ZSW_HIP(ctx, hipMemcpy(ctx->d_sc_t.p, &t, sizeof(ScoringDev), hipMemcpyHostToDevice));
    }
    ctx->scoring_set = true;
    ctx->seed.valid = false;  // the index spells k-mers with the matrix's good residues
    ctx->seed_rev.valid = false;
    ctx->seed_shared.valid = false;
    ctx->seed_shared_rev.valid = false;
    return ZSW_OK;
}

zsw_error zsw_set_reference(zsw_context* ctx, const uint8_t* reference, size_t len, zsw_mem mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || (!reference && len)) return ZSW_ERR_INVALID_ARGUMENT;
    if (len > 0x7fffffffull) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "reference too long");
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->reference_set) ZSW_HIP(ctx, hipDeviceSynchronize());  // queued kernels may still read the previous reference
    ZSW_HIP(ctx, ctx->d_ref.ensure(len + 16));
    if (len)
        ZSW_HIP(ctx, hipMemcpy(ctx->d_ref.p, reference, len, mem == ZSW_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
    ctx->h_ref.resize(len);  // the seeded pass builds its k-mer index from a host copy
    if (len) {
        if (mem == ZSW_MEM_HOST) memcpy(ctx->h_ref.data(), reference, len);
        else ZSW_HIP(ctx, hipMemcpy(ctx->h_ref.data(), reference, len, hipMemcpyDeviceToHost));
    }
    ctx->seed.valid = false;
    ctx->seed_rev.valid = false;
    ctx->ref_len = len;
    ctx->reference_set = true;
    return ZSW_OK;
}

zsw_error zsw_score_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes,
                          uint32_t* out_score, uint8_t* out_status, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_score(ctx, reads, rule, false, out_score, out_status, nullptr, nullptr, nullptr, stream);
}

zsw_error zsw_score_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits,
                               uint32_t* out_score, uint8_t* out_status, uint8_t* out_tier, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_score(ctx, reads, rule, false, out_score, out_status, out_tier, nullptr, nullptr, stream);
}

zsw_error zsw_score_ends_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes,
                               uint32_t* out_score, uint32_t* out_ref_end, uint32_t* out_query_end,
                               uint8_t* out_status, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!out_ref_end || !out_query_end) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_score(ctx, reads, rule, true, out_score, out_status, nullptr, out_ref_end, out_query_end, stream);
}

zsw_error zsw_score_ranges_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes,
                                 uint32_t* out_score, uint32_t* out_ref_start, uint32_t* out_ref_end,
                                 uint32_t* out_query_start, uint32_t* out_query_end, uint8_t* out_status, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_ranges(ctx, reads, rule, out_score, out_ref_start, out_ref_end, out_query_start, out_query_end, out_status, nullptr, stream);
}

zsw_error zsw_score_ranges_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits,
                                      uint32_t* out_score, uint32_t* out_ref_start, uint32_t* out_ref_end,
                                      uint32_t* out_query_start, uint32_t* out_query_end, uint8_t* out_status, uint8_t* out_tier,
                                      void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_ranges(ctx, reads, rule, out_score, out_ref_start, out_ref_end, out_query_start, out_query_end, out_status, out_tier,
                      stream);
}

zsw_error zsw_align_3pass_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert,
                                zsw_alignment* out_aln, uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op,
                                uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!valid_lanes(lanes)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "lanes must be a power of two in 2..64");
    ResultRule rule;
    if (!rule_direct(int_type, ctx->bias, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "bad int_type");
    return run_threepass(ctx, reads, rule, invert, out_aln, out_status, nullptr, out_inc, out_op, ciglet_cap, out_n_ciglets, stream);
}

zsw_error zsw_align_3pass_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                     zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                                     uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (preset_bits != 128 && preset_bits != 256 && preset_bits != 512) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "preset_bits");
    ResultRule rule;
    if (!rule_cascade(from_width, &rule)) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "from_width");
    return run_threepass(ctx, reads, rule, invert, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap, out_n_ciglets, stream);
}

zsw_error zsw_sneaky_snake_batch(zsw_context* ctx, const zsw_batch* reads, const uint32_t* ref_start, const uint32_t* ref_len,
                                 float threshold, uint8_t* out_pass, void* stream_) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!ctx->reference_set) return fail(ctx, ZSW_ERR_NOT_CONFIGURED, "reference not set");
    if (!reads || !ref_start || !ref_len || !out_pass) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    if (reads->n_reads > 0x7fffffffull) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "n_reads > 2^31-1 per call");
    if (reads->n_reads && !reads->bases) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "null bases");
    const uint32_t n = (uint32_t)reads->n_reads;
    if (n == 0) return ZSW_OK;
    hipStream_t stream = (hipStream_t)stream_;
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    BatchDev b{};
    b.n_reads = b.n_items = n;
    b.fixed_len = reads->fixed_len;
    const uint32_t *d_rs = ref_start, *d_rl = ref_len;
    uint8_t* d_out = out_pass;
    if (reads->mem == ZSW_MEM_HOST) {
        for (uint32_t i = 0; i < n; ++i)
            if ((uint64_t)ref_start[i] + ref_len[i] > ctx->ref_len) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "window leaves the reference");
        const size_t total = reads->offsets ? (size_t)reads->offsets[n] : (size_t)n * reads->fixed_len;
        ZSW_HIP(ctx, ctx->s_bases.ensure(total + 16));
        ZSW_HIP(ctx, hipMemcpyAsync(ctx->s_bases.p, reads->bases, total, hipMemcpyHostToDevice, stream));
        b.bases = ctx->s_bases.as<uint8_t>();
        if (reads->offsets) {
            ZSW_HIP(ctx, ctx->s_offsets.ensure((size_t)(n + 1) * 8));
            ZSW_HIP(ctx, hipMemcpyAsync(ctx->s_offsets.p, reads->offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, stream));
            b.offsets = ctx->s_offsets.as<uint64_t>();
        }
        ZSW_HIP(ctx, ctx->s_rend.ensure((size_t)n * 4 + 4));
        ZSW_HIP(ctx, ctx->s_qend.ensure((size_t)n * 4 + 4));
        ZSW_HIP(ctx, ctx->s_status.ensure((size_t)n + 4));
        ZSW_HIP(ctx, hipMemcpyAsync(ctx->s_rend.p, ref_start, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        ZSW_HIP(ctx, hipMemcpyAsync(ctx->s_qend.p, ref_len, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        d_rs = ctx->s_rend.as<uint32_t>();
        d_rl = ctx->s_qend.as<uint32_t>();
        d_out = ctx->s_status.as<uint8_t>();
    } else {
        b.bases = reads->bases;
        b.offsets = reads->offsets;
    }
    ctx->timer.begin(stream);
    hipError_t e = launch_sneaky(b, ctx->d_ref.as<uint8_t>(), (uint32_t)ctx->ref_len, d_rs, d_rl, threshold, d_out, stream);
    ctx->timer.end(stream);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "sneaky_kernel launch", e);
    if (reads->mem == ZSW_MEM_HOST) {
        ZSW_HIP(ctx, hipMemcpyAsync(out_pass, d_out, n, hipMemcpyDeviceToHost, stream));
        ZSW_HIP(ctx, hipStreamSynchronize(stream));
    }
    return ZSW_OK;
}

zsw_error zsw_synth_reads(zsw_context* ctx, uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint8_t* out_device,
                          void* stream) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !out_device || len == 0) return ZSW_ERR_INVALID_ARGUMENT;
    if (!ctx->reference_set) return fail(ctx, ZSW_ERR_NOT_CONFIGURED, "reference not set");
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return ZSW_OK;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, first, n,
                       len, ctx->d_ref.as<uint8_t>(), (uint32_t)ctx->ref_len, out_device);
    ZSW_HIP(ctx, hipGetLastError());
    return ZSW_OK;
}

zsw_error zsw_synth_reads_ragged(zsw_context* ctx, uint64_t seed, uint64_t first, uint64_t n, uint32_t min_len,
                                 uint32_t max_len, const uint64_t* offsets_device, uint8_t* out_device, void* stream) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !out_device || !offsets_device || min_len == 0 || max_len < min_len) return ZSW_ERR_INVALID_ARGUMENT;
    if (!ctx->reference_set) return fail(ctx, ZSW_ERR_NOT_CONFIGURED, "reference not set");
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return ZSW_OK;
    hipLaunchKernelGGL(synth_ragged_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       first, n, min_len, max_len, offsets_device, ctx->d_ref.as<uint8_t>(), (uint32_t)ctx->ref_len,
                       out_device);
    ZSW_HIP(ctx, hipGetLastError());
    return ZSW_OK;
}

uint32_t zsw_synth_length(uint64_t seed, uint64_t index, uint32_t min_len, uint32_t max_len) {
    return zsw_synth_len(seed, index, min_len, max_len);
}

// Host twins of the generator (no GPU needed): tests and the CPU baseline consume the same bytes.
void zsw_synth_reference_host(uint64_t seed, uint64_t len, uint8_t* out) {
    for (uint64_t j = 0; j < len; ++j) out[j] = zsw_synth_ref_base(seed, j);
}
void zsw_synth_reads_host(uint64_t seed, uint64_t first, uint64_t n, uint32_t len, const uint8_t* ref, uint32_t R,
                          uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i) zsw_synth_read(seed, first + i, ref, R, len, out + i * len);
}
void zsw_synth_reads_ragged_host(uint64_t seed, uint64_t first, uint64_t n, uint32_t min_len, uint32_t max_len,
                                 const uint64_t* offsets, const uint8_t* ref, uint32_t R, uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i)
        zsw_synth_read(seed, first + i, ref, R, zsw_synth_len(seed, first + i, min_len, max_len), out + offsets[i]);
}

zsw_error zsw_selftest(zsw_context* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t* d = nullptr;
    ZSW_HIP(ctx, hipMalloc(&d, 256 * 4));
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[256];
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, ZSW_ERR_HIP, "selftest copy", e);
    for (int l = 0; l < 64; ++l) {
        if (h[l] != 0x77443300u) return fail(ctx, ZSW_ERR_HIP, "selftest: v_perm_b32 byte order");
        uint32_t want = (l % 4 == 0) ? (uint32_t)l : (uint32_t)(l - 1);
        if (h[64 + l] != want) return fail(ctx, ZSW_ERR_HIP, "selftest: __shfl_up direction");
        if (h[128 + l] != 0x80007fffu) return fail(ctx, ZSW_ERR_HIP, "selftest: packed saturating add");
        if (h[192 + l] != (uint32_t)(((uint32_t)(uint16_t)(int16_t)31000 << 16) | (uint16_t)(int16_t)-31000))
            return fail(ctx, ZSW_ERR_HIP, "selftest: packed saturating sub");
    }
    return ZSW_OK;
}

zsw_error zsw_debug_set(zsw_context* ctx, uint32_t flags) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    ctx->debug = flags;
    return ZSW_OK;
}

zsw_error zsw_pack4_host(zsw_context* ctx, const uint8_t* bases, uint64_t n_reads, uint32_t len, uint8_t* out_packed) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (!ctx->scoring_set) return fail(ctx, ZSW_ERR_NOT_CONFIGURED, "scoring not set");
    if ((n_reads && len && (!bases || !out_packed)) || ctx->h_sc.S > 16) return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "zsw_pack4_host: null argument, or an alphabet above 16 letters");
    const uint8_t* map = ctx->h_sc.index_map;
    const uint32_t stride = (len + 1) / 2;
    for (uint64_t i = 0; i < n_reads; ++i) {
        const uint8_t* r = bases + i * len;
        uint8_t* o = out_packed + i * stride;
        uint32_t c = 0;
        for (; c + 1 < len; c += 2) o[c / 2] = (uint8_t)(map[r[c]] | (map[r[c + 1]] << 4));
        if (c < len) o[c / 2] = map[r[c]];
    }
    return ZSW_OK;
}

zsw_error zsw_debug_band_records(zsw_context* ctx, int32_t* records) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    ctx->band_dbg = records;
    return ZSW_OK;
}

zsw_error zsw_set_option(zsw_context* ctx, zsw_option option, int64_t value) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    if (option == ZSW_OPTION_EXACT_PRUNING && (value == 0 || value == 1)) {
        ctx->options = value ? (ctx->options | ZSW_DEBUG_SCORE_PRUNE) : (ctx->options & ~(uint32_t)ZSW_DEBUG_SCORE_PRUNE);
        if (!value) {  // the workspaces of the pruned passes go back to the device
            DeviceGuard device_guard(ctx);
            ZSW_HIP(ctx, hipDeviceSynchronize());
            ctx->d_prune.release();
            ctx->d_seed_work.release();
        }
        return ZSW_OK;
    }
    return fail(ctx, ZSW_ERR_INVALID_ARGUMENT, "unknown option or value");
}

zsw_error zsw_prune_rescored(zsw_context* ctx, uint64_t* out_reads) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !out_reads) return ZSW_ERR_INVALID_ARGUMENT;
    *out_reads = 0;
    if ((!ctx->prune_chunk && !ctx->seed_ready) || !ctx->d_prune_count.p) return ZSW_OK;
    uint32_t cnt = 0;
    ZSW_HIP(ctx, hipDeviceSynchronize());
    ZSW_HIP(ctx, hipMemcpy(&cnt, ctx->d_prune_count.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost));
    *out_reads = cnt;
    return ZSW_OK;
}

zsw_error zsw_timing_enable(zsw_context* ctx, int enable) {
    if (!ctx) return ZSW_ERR_INVALID_ARGUMENT;
    ctx->timer.enabled = ctx->timer_window.enabled = enable != 0;
    ctx->timer.used = ctx->timer_window.used = 0;
    return ZSW_OK;
}

zsw_error zsw_timing_read_window(zsw_context* ctx, double* seconds, uint64_t* launches) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !seconds || !launches) return ZSW_ERR_INVALID_ARGUMENT;
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    ZSW_HIP(ctx, ctx->timer_window.collect(seconds, launches));
    return ZSW_OK;
}

zsw_error zsw_timing_read(zsw_context* ctx, double* seconds, uint64_t* launches) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !seconds || !launches) return ZSW_ERR_INVALID_ARGUMENT;
    ZSW_HIP(ctx, hipSetDevice(ctx->device));
    ZSW_HIP(ctx, ctx->timer.collect(seconds, launches));
    return ZSW_OK;
}

}  // extern "C"
