// zsw_score_seed.hip — the seeded exact score pass (sw_simd_score, striped.rs:65-142: only the maximum of the DP matrix is
// returned). Per read: (1) seed_kernel looks a few k-mers of the read up in an index of the reference (first / last occurrence
// per k-mer), votes an anchor diagonal and derives two upper bounds on what any alignment far from that diagonal can score;
// (2) the reads are ordered by anchor (hipCUB radix sort) so that the two reads of a lane and the reads of a block share their
// rows; (3) seed_window_kernel (zsw_score_seed_kernel.hpp) computes all query columns for the ~len + 56 reference rows around
// the anchor and accepts its maximum if the bounds allow no better path elsewhere. Reads without an anchor and reads whose
// bounds fail go to a device-side list and are scored over all their cells by score_kernel_v2 (the caller does that), so the
// results are the full pass's for every input. The argument and its arithmetic: zsw_seed.hpp; host model with the full Gotoh
// matrix as the truth: tests/models/seed_bounds.cpp.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "zsw_score_seed.hpp"
#include "zsw_score_v2.hpp"
#include "zsw_timer.hpp"

namespace zsw {

namespace {

struct SeedArgs {
    BatchDev b;
    const ScoringDev* sc;
    SeedParams sp;
    const uint2* table;
    uint32_t first, n;
    uint32_t* keys;
    uint32_t* ids;
    uint32_t* info;
    uint32_t* masks;
    uint32_t* band_masks;  // banded pass: fa_mask | fb_mask << 16, d_fb, and the read as 4-bit residue codes (cs dwords per item)
    uint8_t* band_dfb;
    uint32_t* codes;
    uint32_t cs;
    uint32_t key_bias, fail_key;
    uint32_t ref_len;
    uint32_t* fail_list;
    uint32_t* fail_count;
    bool reversed;  // column c of a read is its base len - 1 - c
};

// One thread per read. STAGED (contiguous fixed-length reads of at most SEED_STAGE_LEN bases): the block's reads arrive in LDS
// by 16-byte loads first — 150 single-byte loads per thread, each touching 64 different cache lines per wavefront, were most of
// this kernel's time.
constexpr uint32_t SEED_STAGE_LEN = 160;

template <bool STAGED>
__global__ __launch_bounds__(256) void seed_kernel(SeedArgs a) {
    __shared__ uint16_t cell_lut[256];  // read byte -> seed_cell (potential, 2-bit code) of its residue | residue index << 12
    __shared__ __attribute__((aligned(16))) uint8_t sbytes[STAGED ? 256 * SEED_STAGE_LEN + 16 : 16];
    {  // bits 0-7 potential, 8-9 code (bits 8-11 = 0xf: not a good residue), 12-15 residue index (S <= 7)
        const int res = (int)a.sc->index_map[threadIdx.x];
        const uint32_t cellv = seed_cell(a.sp, res);
        cell_lut[threadIdx.x] = (uint16_t)((cellv & 0xffu) | (((cellv >> 8) == 0xffu ? 0xfu : ((cellv >> 8) & 3u)) << 8) | ((uint32_t)(res & 15) << 12));
    }
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const bool valid = k < a.n;
    uint32_t id = 0, len = 0;
    uint64_t off = 0;
    if (valid) {
        id = a.b.items ? a.b.items[a.first + k] : a.first + k;
        off = a.b.offsets ? a.b.offsets[id] : (uint64_t)id * a.b.fixed_len;
        len = a.b.offsets ? (uint32_t)(a.b.offsets[id + 1] - off) : a.b.fixed_len;
    }
    if (STAGED) {  // bytes [block_first * L, min(n, block_first + 256) * L) of the batch, 16 at a time, the tail byte by byte
        const uint32_t L = a.b.fixed_len;
        const uint64_t b0 = (uint64_t)(a.first + blockIdx.x * 256u) * L;
        const uint32_t cnt = min(256u, a.n - blockIdx.x * 256u) * L;
        const uint8_t* src = a.b.bases + b0;
        const uint32_t head = (uint32_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u)) & 15u);  // bytes before the first aligned 16
        // LDS offset = global offset + shift, so that aligned global loads land on aligned LDS stores
        const uint32_t shift = (16u - head) & 15u;
        for (uint32_t i = threadIdx.x; i < min(head, cnt); i += 256) sbytes[shift + i] = src[i];
        const uint32_t n16 = cnt > head ? (cnt - head) / 16 : 0;
        for (uint32_t i = threadIdx.x; i < n16; i += 256)
            *reinterpret_cast<uint4*>(&sbytes[shift + head + 16 * i]) = *reinterpret_cast<const uint4*>(src + head + 16 * i);
        for (uint32_t i = head + 16 * n16 + threadIdx.x; i < cnt; i += 256) sbytes[shift + i] = src[i];
        __syncthreads();
        off = (uint64_t)shift + (uint64_t)threadIdx.x * L;  // this thread's read inside sbytes
    } else {
        __syncthreads();
    }
    SeedRead sr;
    sr.ok = 0;
    sr.t_all = sr.d_fa = sr.d_bl = 0;
    sr.bl_mask = 0;
    if (valid && len >= SEED_MIN_LEN && len < SEED_KEY_BIAS) {
        const uint8_t* bases = STAGED ? sbytes + off : a.b.bases + off;
        const bool rev = a.reversed;
        const uint2* table = a.table;
        // seed_read asks for every column exactly once, in order: the cell functor also packs the residue codes for the banded pass
        uint32_t* code_out = a.codes ? a.codes + (size_t)k * a.cs : nullptr;
        uint32_t acc = 0;
        sr = seed_read(
            a.sp, (int)len,
            [&](int c) {
                const uint32_t v = cell_lut[bases[rev ? (int)len - 1 - c : c]];
                if (code_out) {
                    acc |= (v >> 12) << (4 * (c & 7));
                    if ((c & 7) == 7) {
                        code_out[c >> 3] = acc;
                        acc = 0;
                    }
                }
                return (v & 0xffu) | (((v >> 8) & 0xfu) == 0xfu ? 0xff00u : (v & 0x300u));
            },
            [&](uint32_t code, uint32_t* f1, uint32_t* l1) {
                const uint2 e = table[code];
                *f1 = e.x;
                *l1 = e.y;
            });
    }
    const bool fail = valid && !sr.ok;
    if (valid) {
        a.keys[k] = sr.ok ? (uint32_t)(sr.dt + (int)a.key_bias) : a.fail_key;
        a.ids[k] = k;
        a.info[k] = (uint32_t)sr.t_all | ((uint32_t)sr.d_fa << 16) | ((uint32_t)sr.d_bl << 24);
        a.masks[k] = sr.bl_mask;
        if (a.codes) {
            a.band_masks[k] = sr.fa_mask | (sr.fb_mask << 16);
            a.band_dfb[k] = (uint8_t)sr.d_fb;
            // the last, partial dword of the codes and the rest of the row: padding (reads the seed did not sweep are all padding)
            uint32_t* code_row = a.codes + (size_t)k * a.cs;
            const bool swept = len >= SEED_MIN_LEN && len < SEED_KEY_BIAS;
            const uint32_t full = swept ? len >> 3 : 0;
            if (full < a.cs) {
                uint32_t tail = 0xffffffffu;
                if (swept && (len & 7u)) {
                    const uint8_t* bases = STAGED ? sbytes + off : a.b.bases + off;
                    tail = 0;
                    for (uint32_t c = full * 8; c < full * 8 + 8; ++c)
                        tail |= (c < len ? (uint32_t)(cell_lut[bases[a.reversed ? len - 1 - c : c]] >> 12) : 15u) << (4 * (c & 7));
                }
                code_row[full] = tail;
                for (uint32_t d = full + 1; d < a.cs; ++d) code_row[d] = 0xffffffffu;
            }
        }
    }
    // reads without an anchor are scored over all their cells: one atomic per wavefront
    const unsigned long long m = __ballot(fail);
    if (m) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(a.fail_count, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (fail) a.fail_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = id;
    }
}

__global__ void seed_gtab_kernel(const uint8_t* ref, uint32_t ref_len, const ScoringDev* sc, ScoreArgsV2 a, uint2* gtab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ref_len + 2 * SEED_GTAB_PAD) return;
    const int row = (int)i - SEED_GTAB_PAD;
    const int idx = (row >= 0 && row < (int)ref_len) ? (int)sc->index_map[ref[row]] : NEUTRAL;
    gtab[i] = make_uint2(a.wtab[idx][0], a.wtab[idx][1]);
}

size_t round256(size_t x) { return (x + 255) & ~(size_t)255; }

// temporary storage of the radix sort and of the selection of the reads the full band walks again (the larger of the two)
size_t sort_temp_bytes(uint32_t n) {
    size_t bytes = 0, sel = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)n, 0, 32, (hipStream_t)0);
    (void)hipcub::DeviceSelect::Flagged(nullptr, sel, (const uint32_t*)nullptr, (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                        (int)n, (hipStream_t)0);
    return std::max(bytes, sel);
}

}  // namespace

uint32_t seed_codes_stride(uint32_t max_len) { return (max_len + 7) / 8 + 1; }

// keys, sorted keys, ids, order, info, masks, band masks (u32 per item), d_fb (u8), packed residue codes, sort temp, band buffer
size_t seed_workspace_bytes(uint32_t n, uint32_t max_len, uint32_t band_grid_cap) {
    SeedParams p{};
    p.M1 = SEED_M1;
    p.M1_per8 = SEED_M1_PER8;
    p.Wd = SEED_WD;
    p.Wd_per16 = SEED_WD_PER16;
    return 7 * round256((size_t)n * 4 + 8) + round256((size_t)n + 8) + round256((size_t)n * seed_codes_stride(max_len) * 4 + 8) +
           round256(sort_temp_bytes(n)) + round256(seed_band_buffer_bytes(p, n, max_len, band_grid_cap)) + 256;
}

bool seed_applicable(const SeedIndex& ix, uint32_t max_len, uint32_t ref_len, uint32_t limit) {
    if (!ix.valid || !ix.usable || !ix.d_table || ref_len == 0 || ref_len >= (1u << 24)) return false;
    if (max_len < SEED_MIN_LEN || max_len > SEED_MAX_LEN) return false;
    return (uint64_t)ix.params.maxw * max_len + 8 < limit;  // no score can leave the packed range
}

hipError_t launch_score_seeded(const ScoreArgsV2& a2, int G, int C, uint32_t max_len, const SeedIndex& ix, uint8_t* work, size_t work_bytes,
                               uint2* gtab, uint32_t* fail_list, uint32_t* fail_count, int mode, const ScoreArgsV2* band_tabs, const uint2* gtab_band, int32_t* band_dbg,
                               uint32_t narrow_min_reads, uint32_t band_grid_cap, hipStream_t stream, KernelTimer* window_timer, bool narrow_only, bool reads_reversed) {
    const uint32_t n = a2.b.n_items;
    if (n == 0) return hipSuccess;
    if (!work || !gtab || work_bytes < seed_workspace_bytes(n, max_len, band_grid_cap)) return hipErrorNotSupported;
    // the banded kernel (zsw_score_band.hip) when a strip's rows fit one drift period of the doubled scoring
    const bool band = band_tabs != nullptr && gtab_band != nullptr && seed_band_applicable(ix.params, max_len, band_tabs->K, band_tabs->limit);
    const size_t per = round256((size_t)n * 4 + 8);
    uint32_t* keys = reinterpret_cast<uint32_t*>(work);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(work + per);
    uint32_t* ids = reinterpret_cast<uint32_t*>(work + 2 * per);
    uint32_t* order = reinterpret_cast<uint32_t*>(work + 3 * per);
    uint32_t* info = reinterpret_cast<uint32_t*>(work + 4 * per);
    uint32_t* masks = reinterpret_cast<uint32_t*>(work + 5 * per);
    uint32_t* band_masks = reinterpret_cast<uint32_t*>(work + 6 * per);
    uint8_t* p8 = work + 7 * per;
    uint8_t* band_dfb = p8;
    p8 += round256((size_t)n + 8);
    const uint32_t cs = seed_codes_stride(max_len);
    uint32_t* codes = reinterpret_cast<uint32_t*>(p8);
    p8 += round256((size_t)n * cs * 4 + 8);
    void* temp = p8;
    p8 += round256(sort_temp_bytes(n));
    uint2* band_buf = reinterpret_cast<uint2*>(p8);
    size_t temp_bytes = sort_temp_bytes(n);
    int key_bits = 1;
    while ((1ull << key_bits) < (uint64_t)a2.ref_len + 2ull * SEED_KEY_BIAS + 2 && key_bits < 32) ++key_bits;
    const uint32_t fail_key = key_bits >= 32 ? 0xffffffffu : (1u << key_bits) - 1u;

    SeedArgs s;
    s.b = a2.b;
    s.sc = a2.sc;
    s.sp = ix.params;
    s.table = reinterpret_cast<const uint2*>(ix.d_table);
    s.first = 0;
    s.n = n;
    s.keys = keys;
    s.ids = ids;
    s.info = info;
    s.masks = masks;
    s.band_masks = band_masks;
    s.band_dfb = band_dfb;
    s.codes = band ? codes : nullptr;
    s.cs = cs;
    s.key_bias = SEED_KEY_BIAS;
    s.fail_key = fail_key;
    s.ref_len = a2.ref_len;
    s.fail_list = fail_list;
    s.fail_count = fail_count;
    s.reversed = reads_reversed;
    if (!s.b.offsets && !s.b.items && s.b.fixed_len <= SEED_STAGE_LEN)
        hipLaunchKernelGGL(seed_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, stream, s);
    else
        hipLaunchKernelGGL(seed_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, stream, s);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint32_t*)keys, keys_out, (const uint32_t*)ids, order, (int)n, 0,
                                                      key_bits, stream);
    if (e != hipSuccess) return e;

    if (band) {
        SeedBandArgs b;
        b.b = a2.b;
        b.ref_len = a2.ref_len;
        b.ge2 = band_tabs->ge2;
        b.gd2 = band_tabs->gd2;
        b.floor0 = band_tabs->floor0;
        b.rule = a2.rule;
        b.out = a2.out;
        b.sp = ix.params;
        b.n = n;
        b.order = order;
        b.keys = keys;
        b.info = info;
        b.band_masks = band_masks;
        b.band_dfb = band_dfb;
        b.codes = codes;
        b.cs = cs;
        b.gtab = gtab_band;
        b.bnd = band_buf;
        b.dbg = band_dbg;
        b.nb = seed_band_rows(ix.params, max_len);
        b.grid = seed_band_grid(n, band_grid_cap);
        b.key_bias = SEED_KEY_BIAS;
        b.fail_key = fail_key;
        b.fail_list = fail_list;
        b.fail_count = fail_count;
        b.wu0 = ix.params.M1;
        b.wu_per16 = 2 * ix.params.M1_per8;
        b.wd0 = ix.params.Wd;
        b.wd_per32 = 2 * ix.params.Wd_per16;
        b.n_dev = nullptr;
        b.retry = nullptr;
        // work-queue counters of the (up to) two launches: the spare words behind `order` and `keys`
        uint32_t* queue1 = reinterpret_cast<uint32_t*>(work + 3 * per + (size_t)n * 4) + 1;
        uint32_t* queue2 = reinterpret_cast<uint32_t*>(work + (size_t)n * 4);
        e = hipMemsetAsync(queue1, 0, 4, stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(queue2, 0, 8, stream);  // (and the first tier's count of accepted reads behind it)
        if (e != hipSuccess) return e;
        uint32_t* accepted = queue2 + 1;
        b.next_pair = queue2;
        if (window_timer) window_timer->begin(stream);
        if (max_len <= SEED_NARROW_MAX_LEN && n >= narrow_min_reads) {
            // two tiers: every read in a narrow band; the reads whose bounds fail there, still in anchor order, in the full band.
            // The sort's input values and output keys are free by now: flags and the second tier's order.
            uint8_t* retry = reinterpret_cast<uint8_t*>(ids);
            uint32_t* order2 = keys_out;
            uint32_t* n2 = reinterpret_cast<uint32_t*>(work + 3 * per + (size_t)n * 4);  // the 8 spare bytes behind `order`
            e = hipMemsetAsync(retry, 0, n, stream);
            if (e != hipSuccess) return e;
            SeedBandArgs b1 = b;
            b1.wu0 = SEED_NARROW_WU;
            b1.wu_per16 = SEED_NARROW_WU_PER16;
            b1.wd0 = SEED_NARROW_WD;
            b1.wd_per32 = SEED_NARROW_WD_PER32;
            b1.retry = narrow_only ? nullptr : retry;  // no second tier: what fails joins the worklist at once
            b1.accepted = narrow_only ? nullptr : accepted;
            b1.next_pair = queue1;
            e = launch_seed_band(b1, mode, true, stream);
            if (e != hipSuccess) return e;
            if (narrow_only) {
                if (window_timer) window_timer->end(stream);
                return hipSuccess;
            }
            e = hipcub::DeviceSelect::Flagged(temp, temp_bytes, (const uint32_t*)order, (const uint8_t*)retry, order2, n2, (int)n, stream);
            if (e != hipSuccess) return e;
            b.order = order2;
            b.n_dev = n2;
            b.wu0 = SEED_SECOND_WU;
            b.wu_per16 = SEED_SECOND_WU_PER16;
            b.wd0 = SEED_SECOND_WD;
            b.wd_per32 = SEED_SECOND_WD_PER32;
            // measured (tools/bench_divergence.py): at 8 % divergence the first tier accepts 12 % of the anchored reads and the second
            // a third of the rest (worth it); at 10 % 3 % and an eighth (break-even); at 12 % 0.4 % and 3 % (4 ms spent to save 1)
            b.accepted = accepted;
            b.bail_check = true;
            b.bail_below = band_dbg ? 0u : SEED_BAIL_BELOW;  // (the bounds test wants the second tier's records)
        }
        e = launch_seed_band(b, mode, false, stream);
        if (window_timer) window_timer->end(stream);
        return e;
    }
    SeedWindowArgs w;
    w.b = a2.b;
    w.ref = a2.ref;
    w.ref_len = a2.ref_len;
    w.sc = a2.sc;
    for (int r = 0; r < 9; ++r) {
        w.wtab[r][0] = a2.wtab[r][0];
        w.wtab[r][1] = a2.wtab[r][1];
    }
    w.ge2 = a2.ge2;
    w.gd2 = a2.gd2;
    w.floor0 = a2.floor0;
    w.K = a2.K;
    w.rule = a2.rule;
    w.out = a2.out;
    w.sp = ix.params;
    w.first = 0;
    w.n = n;
    w.order = order;
    w.keys = keys;
    w.info = info;
    w.masks = masks;
    w.gtab = gtab;
    w.key_bias = SEED_KEY_BIAS;
    w.fail_key = fail_key;
    w.fail_list = fail_list;
    w.fail_count = fail_count;
    w.reversed = reads_reversed;
    if (window_timer) window_timer->begin(stream);
    e = mode == 0 ? launch_seed_window_m0(w, G, C, stream) : mode == 1 ? launch_seed_window_m1(w, G, C, stream) : launch_seed_window_m2(w, G, C, stream);
    if (window_timer) window_timer->end(stream);
    return e;
}

hipError_t seed_build_gtab(const ScoreArgsV2& a2, uint2* gtab, hipStream_t stream) {
    hipLaunchKernelGGL(seed_gtab_kernel, dim3((a2.ref_len + 2 * SEED_GTAB_PAD + 255) / 256), dim3(256), 0, stream, a2.ref, a2.ref_len, a2.sc, a2, gtab);
    return hipGetLastError();
}

hipError_t seed_index_update(SeedIndex* ix, const ScoringDev& sc, const uint8_t* h_ref, size_t ref_len) {
    ix->valid = true;
    ix->usable = false;
    if (ref_len == 0 || ref_len >= (size_t(1) << 24)) return hipSuccess;
    std::vector<uint8_t> res(ref_len);
    bool ref_has[32] = {false};
    for (size_t i = 0; i < ref_len; ++i) {
        res[i] = sc.index_map[h_ref[i]];
        ref_has[res[i] & 31] = true;
    }
    const int K = seed_k_for(ref_len);
    SeedParams p{};
    if (sc.S > 7 || !seed_analyze(sc.S, sc.w, sc.gap_open, sc.gap_extend, ref_has, K, &p)) return hipSuccess;
    p.M1 = SEED_M1;
    p.M1_per8 = SEED_M1_PER8;
    p.M2 = SEED_M2;
    p.Dn = SEED_DN;
    p.Dm = SEED_DM;
    p.Wd = SEED_WD;
    p.Wd_per16 = SEED_WD_PER16;
    p.tol = SEED_TOL;
    const size_t entries = size_t(1) << (2 * K);
    std::vector<uint32_t> table(2 * entries, 0u);
    seed_index_build(p, res.data(), ref_len, table.data());
    const size_t bytes = table.size() * sizeof(uint32_t);
    if (bytes > ix->table_bytes) {
        if (ix->d_table) (void)hipFree(ix->d_table);
        ix->d_table = nullptr;
        ix->table_bytes = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ix->d_table), bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return hipSuccess;  // no index: every cell is computed
        }
        ix->table_bytes = bytes;
    }
    hipError_t e = hipMemcpy(ix->d_table, table.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    ix->params = p;
    ix->usable = true;
    return hipSuccess;
}

void seed_index_release(SeedIndex* ix) {
    if (ix->d_table) (void)hipFree(ix->d_table);
    ix->d_table = nullptr;
    ix->table_bytes = 0;
    ix->valid = false;
    ix->usable = false;
}

}  // namespace zsw
