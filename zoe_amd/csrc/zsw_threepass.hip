// zsw_threepass.hip — third pass of sw_align_3pass (src/alignment/sw/three_pass.rs:21-104), batched.
//
// Passes 1 and 2 are sw_simd_score_ranges (zsw_score.hip: forward score+ends, reverse pass on the prefixes). What is left
// per read is small and sequential, so one GPU thread owns one read:
//   * ranges of equal length whose diagonal weights add up to the score -> `new_no_gaps` (state.rs:201-208);
//   * else sw_banded_align (banded.rs:40-133) on the bounding box, band |dr - dq| + 1 doubling while <= (q-1)/2, accepted when
//     it reproduces the score; else sw_scalar_align (scalar.rs:173-271) on the box;
//   * the outer soft clips are then added exactly as the reference does (prepend_soft_clip / soft_clip merge with the inner
//     alignment's own clips, three_pass.rs:85-92).
// Flag bytes and the two DP rows of a read live in a per-thread slot of HBM scratch (a 150 x 160 box is 24 KB); a slot is
// sized for the attempt in progress (band first), see the DP pass.
#include "zsw_align.hpp"

namespace zsw {

constexpr uint8_t BT_UP = 1, BT_UP_EXT = 2, BT_LEFT = 4, BT_LEFT_EXT = 8, BT_STOP = 16;  // backtrack.rs:18-34

struct CigWriter {  // AlignmentStates::add_ciglet (state.rs:142-152) into the pool, traceback order
    uint32_t* cig;
    uint32_t maxc, ncig = 0, cur_op = 0, cur_inc = 0, n_nons = 0;
    bool overflow = false;
    __device__ void push(uint32_t inc, uint32_t op) {
        if (inc == 0) return;
        if (cur_inc && cur_op == op) {
            cur_inc += inc;
            return;
        }
        flush();
        cur_op = op;
        cur_inc = inc;
        if (op != 'S') ++n_nons;
    }
    __device__ void flush() {
        if (cur_inc) {
            if (ncig < maxc) cig[ncig] = (cur_inc << 8) | cur_op;
            else overflow = true;
            ++ncig;
        }
        cur_inc = 0;
    }
};

__device__ __forceinline__ uint32_t tp_read_len(const BatchDev& b, uint32_t id, uint64_t* off) {
    if (b.offsets) {
        *off = b.offsets[id];
        return (uint32_t)(b.offsets[id + 1] - *off);
    }
    *off = (uint64_t)id * b.fixed_len;
    return b.fixed_len;
}

// bytes of a slot that holds the two DP rows (16-byte aligned) and the flag bytes of the scalar alignment of the whole box
__device__ __forceinline__ uint64_t slot_need(uint32_t rlen, uint32_t qlen) { return (uint64_t)rlen * qlen + ((8ull * qlen + 15) & ~15ull) + 32; }

// BackTrackable::to_alignment (backtrack.rs:290-342) over a cell functor; emits into `w` in traceback order, including the outer
// clips of three_pass.rs:85-92. Returns false when the reference would index outside its banded matrix.
template <typename CellFn>
__device__ bool tp_traceback(CellFn cell, int r_end, int c_end, uint32_t qlen_box, uint32_t q_off, uint32_t query_len, CigWriter& w,
                             int* r_out, int* c_out) {
    bool ok = true;
    uint32_t f = cell(r_end, c_end, &ok);
    int r = r_end + 1, c = c_end + 1;
    // inner 3' clip (query_len_box - c) merged with the outer one (query.len() - adjusted_end)
    w.push((qlen_box - (uint32_t)c) + (query_len - ((uint32_t)c + q_off)), 'S');
    uint32_t op = 0;
    while (ok && !(f & BT_STOP) && r > 0 && c > 0) {
        if (op == 'D' && (f & BT_UP_EXT)) {
            r -= 1;
        } else if (op == 'I' && (f & BT_LEFT_EXT)) {
            c -= 1;
        } else if (f & BT_UP) {
            op = 'D';
            r -= 1;
        } else if (f & BT_LEFT) {
            op = 'I';
            c -= 1;
        } else {
            op = 'M';
            r -= 1;
            c -= 1;
        }
        w.push(1, op);
        f = cell(r > 0 ? r - 1 : 0, c > 0 ? c - 1 : 0, &ok);
    }
    // inner 5' clip (c) merged with the outer one (adjusted start = c + q_off): the reference adds both (three_pass.rs:91)
    w.push((uint32_t)c + ((uint32_t)c + q_off), 'S');
    w.flush();
    *r_out = r;
    *c_out = c;
    return ok;
}

__global__ __launch_bounds__(64) void threepass_kernel(ThreePassArgs a) {
    __shared__ uint8_t lut[256];
    __shared__ int32_t wsh[MAX_S * MAX_S];
    for (int i = threadIdx.x; i < 256; i += 64) lut[i] = a.sc->index_map[i];
    for (int i = threadIdx.x; i < MAX_S * MAX_S; i += 64) wsh[i] = a.sc->w[i];
    __syncthreads();
    const int S = a.sc->S;
    const int go = -a.sc->gap_open, ge = -a.sc->gap_extend;  // negative, as in ScalarProfile (profile.rs:110-115)
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    const uint32_t nthreads = gridDim.x * 64;
    const uint32_t n = a.list ? *a.list_count : a.b.n_reads;
    auto wt = [&](uint8_t rb, uint8_t qb) -> int32_t { return wsh[lut[rb] * S + lut[qb]]; };

    for (uint32_t item = tid; item < n; item += nthreads) {
        const uint32_t id = a.list ? a.list[item] : item;
        if (a.status[id] != ZSW_STATUS_SOME) continue;
        uint64_t off = 0;
        const uint32_t read_len = tp_read_len(a.b, id, &off);
        // three_pass.rs:21-26: `reference` is the sequence the profile walks, `query` the sequence the profile was built from
        const uint32_t query_len = a.pseq ? a.pseq_len : read_len;
        const uint8_t* query = a.pseq ? a.pseq : a.b.bases + off;
        const uint8_t* reference = a.pseq ? a.b.bases + off : a.ref;
        const uint32_t reference_len = a.pseq ? read_len : a.ref_len;
        const uint32_t score = a.score[id];
        const uint32_t rs = a.rs[id], re = a.re[id], qs = a.qs[id], qe = a.qe[id];
        const uint32_t rlen = re - rs, qlen = qe - qs;
        uint32_t* cig = a.cig + a.pool_base + (uint64_t)(a.by_item ? item : id) * a.maxc;
        CigWriter w;
        w.cig = cig;
        w.maxc = a.maxc;
        zsw_alignment out;
        out.score = score;
        out.ref_len = reference_len;
        out.query_len = query_len;
        out.ciglet_offset = 0;
        bool done = false;
        if (!a.list || a.sweep_pass) {
            // classify pass: the no-gaps shortcut (three_pass.rs:37-58) is resolved here, the rest is queued for the DP pass
            // (certificate mode: only reads with one optimal alignment, which must be this diagonal; the rest is not touched)
            // Certificate mode (run_align): a.cert_ok[id] = both maxima of the read sit in one cell each, so every alignment that
            // scores `score` runs from (rs, qs) to (re - 1, qe - 1). Equal ranges: it must be the diagonal, and no path with an
            // insertion and a deletion may reach it (tests/models/align_gapless_cert.cpp). Ranges that differ by g: one placement
            // of ONE gap run of g (or a run of adjacent placements) must reach it, and no path with two runs may (tests/models/align_onegap_cert.cpp).
            const bool cert = a.cert_ok != nullptr;
            const bool defer = a.sweep_list != nullptr && !a.sweep_pass;  // reads that need the sweeps wait for the second launch
            bool deferred = false;
            const bool uniq = !cert || (a.cert_ok[id] && re > rs && qe > qs);
            if (a.cert_done) a.cert_done[id] = 0;
            // (gapless certificate: three or more gap runs are ruled out by the potential — at most n - 1 pairs, 3 * gap_open —, two runs,
            // an insertion and a deletion of the same length k in either order, by the potential too or by the sweep below)
            if (uniq && qlen == rlen && (!cert || (long long)score > (long long)a.cert_maxw * ((long long)rlen - 1) - 3ll * a.cert_go)) {
                int64_t sum = 0;
                uint32_t k = 0;
                for (; k + 4 <= qlen; k += 4) {  // four residues per (unaligned) load
                    uint32_t rw, qw;
                    __builtin_memcpy(&rw, reference + rs + k, 4);
                    __builtin_memcpy(&qw, query + qs + k, 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum += wt((uint8_t)(rw >> (8 * j)), (uint8_t)(qw >> (8 * j)));
                }
                for (; k < qlen; ++k) sum += wt(reference[rs + k], query[qs + k]);
                bool only = (sum < 0 ? 0u : (uint32_t)sum) == score;
                if (only && cert && defer && (long long)a.cert_maxw * ((long long)rlen - 1) - 2ll * a.cert_go >= (long long)score) {
                    only = false;  // (k = 1 is not ruled out by the potential: sweeps needed)
                    deferred = true;
                }
                if (only && cert) {
                    // paths with two runs of k: pairs 0 .. i-1 on the diagonal, run, pairs i .. j-1 on the diagonal k rows (dir 0) or
                    // k columns (dir 1) away, run, pairs j+k .. n-1 on the diagonal again: score - 2go - 2ge(k-1) + A(j) - B(i) with
                    // A(j) = Q(j) - P0(j+k), B(i) = Q(i) - P0(i), 1 <= i <= j <= n-k-1 (tests/models/align_gapless_cert.cpp)
                    const long long n_ = (long long)rlen, S_ = (long long)score;
                    for (uint32_t k = 1; k < rlen && only; ++k) {
                        if ((long long)a.cert_maxw * (n_ - k) - 2ll * a.cert_go - 2ll * a.cert_ge * ((long long)k - 1) < S_) break;
                        for (int dir = 0; dir < 2 && only; ++dir) {
                            const uint8_t* r1 = reference + rs + (dir == 0 ? k : 0);
                            const uint8_t* q1 = query + qs + (dir == 0 ? 0 : k);
                            int64_t qv = 0, p0j = 0, p0jk = 0, low = INT64_MAX, best_alt = INT64_MIN;
                            for (uint32_t t = 0; t < k; ++t) p0jk += wt(reference[rs + t], query[qs + t]);
                            for (uint32_t j = 1; j + k + 1 <= rlen; ++j) {
                                qv += wt(r1[j - 1], q1[j - 1]);
                                p0j += wt(reference[rs + j - 1], query[qs + j - 1]);
                                p0jk += wt(reference[rs + j + k - 1], query[qs + j + k - 1]);
                                const int64_t bj = qv - p0j;
                                low = bj < low ? bj : low;
                                const int64_t v = qv - p0jk - low;
                                best_alt = v > best_alt ? v : best_alt;
                            }
                            if (best_alt != INT64_MIN && best_alt - 2ll * a.cert_go - 2ll * a.cert_ge * ((long long)k - 1) >= 0) only = false;
                        }
                    }
                }
                if (only) {
                    w.push(query_len - qe, 'S');
                    w.push(qe - qs, 'M');
                    w.push(qs, 'S');
                    w.flush();
                    out.ref_start = rs;
                    out.ref_end = re;
                    out.query_start = qs;
                    out.query_end = qe;
                    done = true;
                }
            } else if (cert && uniq && qlen != rlen) {
                const bool del = rlen > qlen;  // the run consumes reference rows
                const uint32_t g = del ? rlen - qlen : qlen - rlen, m = del ? qlen : rlen;
                // three or more runs: ruled out by the potential; two runs: by the potential or by the sweeps below
                const long long three_runs = (long long)a.cert_maxw * m - 3ll * a.cert_go - (long long)(g > 3 ? g - 3 : 0) * a.cert_ge;
                if (m >= 2 && a.cert_ge > 0 && (long long)score > three_runs) {
                    // second diagonal: the pairs behind the run
                    const uint8_t* r1 = reference + rs + (del ? g : 0);
                    const uint8_t* q1 = query + qs + (del ? 0 : g);
                    int64_t t1 = 0;
                    for (uint32_t k = 0; k < m; ++k) t1 += wt(r1[k], q1[k]);
                    const int64_t gap = (int64_t)a.cert_go + (int64_t)(g - 1) * a.cert_ge;
                    int64_t p0 = 0, p1 = 0, best = INT64_MIN;
                    uint32_t best_p = 0, first_p = 0, n_best = 0;
                    for (uint32_t p = 1; p < m; ++p) {
                        p0 += wt(reference[rs + p - 1], query[qs + p - 1]);
                        p1 += wt(r1[p - 1], q1[p - 1]);
                        const int64_t sc = p0 + (t1 - p1) - gap;
                        if (sc > best) {
                            best = sc;
                            best_p = first_p = p;
                            n_best = 1;
                        } else if (sc == best) {
                            ++n_best;
                            best_p = p;
                        }
                    }
                    // one placement, or adjacent ones (a gap inside a homopolymer run): the walk from the end takes the last
                    bool only = best_p - first_p == n_best - 1 && best == (int64_t)score;
                    if (only) {
                        // two runs of signed lengths ra and rb = gs - ra (a deletion counts +, an insertion -): i pairs on the first
                        // diagonal, run ra, j - i pairs on the diagonal ra away, run rb, the rest on the last diagonal:
                        // P0(i) + Pa(j) - Pa(i) + Pz(M) - Pz(j) - cost, 1 <= i <= j <= M - 1 (tests/models/align_onegap_cert.cpp)
                        const long long gs = (long long)rlen - (long long)qlen, S_ = (long long)score;
                        // |ra| + |rb| <= X or the potential rules the pair out; |ra| + |gs - ra| >= 2 |ra| - g
                        const long long X = ((long long)a.cert_maxw * m - 2ll * a.cert_go - S_) / a.cert_ge + 2;
                        const long long amax = X >= 0 ? (X + g) / 2 + 1 : 0;
                        for (long long ra = -amax; ra <= amax && only; ++ra) {
                            const long long rb = gs - ra;
                            if (ra == 0 || rb == 0) continue;
                            const long long ap = ra > 0 ? ra : 0, an = ra < 0 ? -ra : 0, bp = rb > 0 ? rb : 0, bn = rb < 0 ? -rb : 0;
                            const long long M = (long long)rlen - ap - bp;
                            if (M < 2) continue;
                            const long long cost = 2ll * a.cert_go + (long long)a.cert_ge * (ap + an + bp + bn - 2);
                            if ((long long)a.cert_maxw * M - cost < S_) continue;
                            if (defer) {
                                only = false;
                                deferred = true;
                                break;
                            }
                            const uint8_t* ra_r = reference + rs + ap;
                            const uint8_t* ra_q = query + qs + an;
                            const uint8_t* rz_r = reference + rs + ap + bp;
                            const uint8_t* rz_q = query + qs + an + bn;
                            int64_t pzM = 0;
                            for (long long t = 0; t < M; ++t) pzM += wt(rz_r[t], rz_q[t]);
                            int64_t s0 = 0, sa = 0, sz = 0, low = INT64_MAX, best_alt = INT64_MIN;
                            for (long long j = 1; j <= M - 1; ++j) {
                                s0 += wt(reference[rs + j - 1], query[qs + j - 1]);
                                sa += wt(ra_r[j - 1], ra_q[j - 1]);
                                sz += wt(rz_r[j - 1], rz_q[j - 1]);
                                const int64_t bj = sa - s0;
                                low = bj < low ? bj : low;
                                const int64_t v = sa - sz - low;
                                best_alt = v > best_alt ? v : best_alt;
                            }
                            if (pzM - cost + best_alt >= S_) only = false;
                        }
                    }
                    if (only) {
                        w.push(query_len - qe, 'S');
                        w.push(m - best_p, 'M');
                        w.push(g, del ? 'D' : 'I');
                        w.push(best_p, 'M');
                        w.push(qs, 'S');
                        w.flush();
                        out.ref_start = rs;
                        out.ref_end = re;
                        out.query_start = qs;
                        out.query_end = qe;
                        done = true;
                    }
                }
            }
            if (deferred) a.sweep_list[atomicAdd(a.sweep_count, 1u)] = id;
            if (!done && a.cert_ok) continue;
            if (!done) {
                const uint32_t k = atomicAdd(a.dp_count, 1u);
                a.dp_list[k] = id;
                const uint64_t need = slot_need(rlen, qlen);
                atomicMax(a.dp_need_max, (uint32_t)(need > 0xffffffffull ? 0xffffffffull : need));
                continue;
            }
        } else {
            // The slot holds the two DP rows and the flag bytes of the attempt in progress: rlen * (2*band + 1) for a banded
            // attempt, rlen * qlen only for the scalar fallback. An attempt that does not fit sends the read to the rerun with
            // full-size slots (which starts over and takes the same decisions), so long reads with few indels — a narrow
            // band — are served by this launch's many small slots instead of a handful of box-sized ones.
            const uint64_t rows_bytes = ((8ull * qlen + 15) & ~15ull) + 32;
            bool too_big = rows_bytes > a.slot_bytes;
            if (too_big) {
                const uint32_t k = atomicAdd(a.fb_count, 1u);
                a.fb_list[k] = id;
                continue;
            }
            // The 64 slots of a block are interleaved word by word (word i of lane l at word i * 64 + l of the block's region): the
            // lanes walk their rows and flag bytes in step, so a wavefront's accesses to "its i-th word" fall into four cache lines
            // instead of sixty-four.
            uint32_t* const region = reinterpret_cast<uint32_t*>(a.scratch + (uint64_t)blockIdx.x * 64 * a.slot_bytes) + threadIdx.x;
            const uint32_t e_off = qlen;                                          // words: H row, then E row, then the flag bytes
            const uint64_t bt_off = (((8ull * qlen + 15) & ~15ull) >> 2);
            auto h_row = [&](uint32_t c) -> int32_t& { return reinterpret_cast<int32_t*>(region)[(uint64_t)c * 64]; };
            auto e_row = [&](uint32_t c) -> int32_t& { return reinterpret_cast<int32_t*>(region)[(uint64_t)(e_off + c) * 64]; };
            auto bt = [&](uint64_t k) -> uint8_t& { return reinterpret_cast<uint8_t*>(region + (bt_off + (k >> 2)) * 64)[k & 3]; };
            const uint8_t* refb = reference + rs;
            const uint8_t* qb = query + qs;
            int r_fin = 0, c_fin = 0, r_end = 0, c_end = 0;
            bool have = false;
            uint32_t band = (rlen > qlen ? rlen - qlen : qlen - rlen) + 1;
            const uint32_t max_band = (qlen - 1) / 2;
            while (!have && band <= max_band) {
                // sw_banded_align (banded.rs:40-133)
                const uint32_t full = 2 * band + 1;
                if (rows_bytes + (uint64_t)rlen * full > a.slot_bytes) {
                    too_big = true;
                    break;
                }
                for (uint32_t c = 0; c < qlen; ++c) {
                    h_row(c) = 0;
                    e_row(c) = go;
                }
                for (uint64_t k = 0; k < ((uint64_t)rlen * full + 3) / 4; ++k) region[(bt_off + k) * 64] = 0;  // the flag bytes, a word at a time
                int32_t best = 0, h_store = 0;
                for (uint32_t r = 0; r < rlen; ++r) {
                    int32_t f = go, h = h_store;
                    const uint32_t start_col = r > band ? r - band : 0;
                    const uint32_t end_col = min(r + band + 1, qlen);
                    if (start_col >= end_col) break;
                    if (start_col + band == r) h_store = max(max(h + wt(refb[r], qb[start_col]), e_row(start_col)), 0);
                    for (uint32_t c = start_col; c < end_col; ++c) {
                        uint8_t cell = 0;
                        h += wt(refb[r], qb[c]);
                        int32_t e = e_row(c);
                        h = max(max(max(h, e), f), 0);
                        if (h > best) {
                            best = h;
                            r_end = (int)r;
                            c_end = (int)c;
                        }
                        if (e == h) cell |= BT_UP;
                        if (f == h) cell |= BT_LEFT;
                        if (h == 0) cell = BT_STOP;
                        const int32_t next_diag = h_row(c);
                        h_row(c) = h;
                        h += go;
                        e = max(e + ge, h);
                        f = max(f + ge, h);
                        if (h != go) {
                            if (e > h) cell |= BT_UP_EXT;
                            if (f > h) cell |= BT_LEFT_EXT;
                        }
                        h = next_diag;
                        e_row(c) = e;
                        bt((uint64_t)r * full + (c - start_col)) = cell;
                    }
                }
                if (best > 0 && (uint32_t)best == score) {
                    // BandedBacktrackMatrix::move_to (backtrack.rs:629-634)
                    auto cell = [&](int rr, int cc, bool* ok) -> uint32_t {
                        const uint32_t skipped = (uint32_t)rr > band ? (uint32_t)rr - band : 0;
                        if ((uint32_t)cc < skipped) {
                            *ok = false;
                            return BT_STOP;
                        }
                        const uint64_t cur = (uint64_t)rr * full + ((uint32_t)cc - skipped);
                        if (cur >= (uint64_t)rlen * full) {
                            *ok = false;
                            return BT_STOP;
                        }
                        return bt(cur);
                    };
                    CigWriter trial = w;
                    if (tp_traceback(cell, r_end, c_end, qlen, qs, query_len, trial, &r_fin, &c_fin)) {
                        w = trial;
                        have = true;
                    }
                }
                if (!have) band *= 2;
            }
            if (!have && !too_big && rows_bytes + (uint64_t)rlen * qlen > a.slot_bytes) too_big = true;
            if (too_big) {
                const uint32_t k = atomicAdd(a.fb_count, 1u);
                a.fb_list[k] = id;
                continue;
            }
            if (!have) {
                // sw_scalar_align on the box (scalar.rs:173-271)
                for (uint32_t c = 0; c < qlen; ++c) {
                    h_row(c) = 0;
                    e_row(c) = go;
                }
                int32_t best = 0;
                for (uint32_t r = 0; r < rlen; ++r) {
                    int32_t f = go, h = 0;
                    for (uint32_t c = 0; c < qlen; ++c) {
                        uint8_t cell = 0;
                        h += wt(refb[r], qb[c]);
                        int32_t e = e_row(c);
                        h = max(max(max(h, e), f), 0);
                        if (h > best) {
                            best = h;
                            r_end = (int)r;
                            c_end = (int)c;
                        }
                        if (e == h) cell |= BT_UP;
                        if (f == h) cell |= BT_LEFT;
                        if (h == 0) cell = BT_STOP;
                        const int32_t next_diag = h_row(c);
                        h_row(c) = h;
                        h += go;
                        e = max(e + ge, h);
                        f = max(f + ge, h);
                        if (h != go) {
                            if (e > h) cell |= BT_UP_EXT;
                            if (f > h) cell |= BT_LEFT_EXT;
                        }
                        h = next_diag;
                        e_row(c) = e;
                        bt((uint64_t)r * qlen + c) = cell;
                    }
                }
                auto cell = [&](int rr, int cc, bool* ok) -> uint32_t { return bt((uint64_t)rr * qlen + (uint32_t)cc); };
                tp_traceback(cell, r_end, c_end, qlen, qs, query_len, w, &r_fin, &c_fin);
            }
            out.ref_start = (uint32_t)r_fin + rs;
            out.ref_end = (uint32_t)(r_end + 1) + rs;
            out.query_start = (uint32_t)c_fin + qs;
            out.query_end = (uint32_t)(c_end + 1) + qs;
        }
        if (w.overflow) {
            const uint32_t k = atomicAdd(a.fb_count, 1u);
            a.fb_list[k] = id;
            continue;
        }
        out.n_ciglets = a.invert ? w.n_nons + (out.ref_start > 0 ? 1u : 0u) + (reference_len > out.ref_end ? 1u : 0u) : w.ncig;
        a.aln[id] = out;
        a.cig_start[id] = (uint64_t)(uintptr_t)cig;
        a.cig_raw[id] = w.ncig;
        if (a.cert_done) a.cert_done[id] = 1;
    }
}

hipError_t launch_threepass(const ThreePassArgs& a, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(threepass_kernel, dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace zsw
