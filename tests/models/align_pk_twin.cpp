// Host twin of the packed alignment kernel's row update — TEST INFRASTRUCTURE.
//
// zoe_amd/csrc/zsw_align_pk.hpp is written over a wave-value type. Here V is 64 explicit lanes and every op is the
// plain-C meaning of the gfx950 instruction the device build uses (v_pk_*_u16/i16, v_lshl_or_b32, v_bfi_b32, DPP shifts),
// so the kernel's arithmetic runs on the CPU exactly as written and is compared, cell by cell and row by row, with the
// oracle's literal restatement of sw_simd_align (oracle/zoe_oracle.hpp, flags_out): 2*64/N different reads per "wavefront"
// (different lengths with equal nv, different last rows), all N, vector counts 1..32, random scoring schemes, and with the first
// rows run through the flag-less scan path. Exit code 0 = identical everywhere.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../oracle/zoe_oracle.hpp"
#include "../../zoe_amd/csrc/zsw_align_pk.hpp"

using namespace zor;

namespace {

struct HV {
    uint32_t l[64];
};

struct HostOps {
    using V = HV;
    static V splat(uint32_t x) {
        V r;
        for (auto& e : r.l) e = x;
        return r;
    }
    template <class F> static V map2(const V& a, const V& b, F f) {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = f(a.l[i], b.l[i]);
        return r;
    }
    template <class F> static V pk2(const V& a, const V& b, F f) {  // per 16-bit half
        return map2(a, b, [&](uint32_t x, uint32_t y) {
            const uint32_t lo = f(x & 0xffffu, y & 0xffffu) & 0xffffu, hi = f(x >> 16, y >> 16) & 0xffffu;
            return lo | (hi << 16);
        });
    }
    static V add(const V& a, const V& b) { return pk2(a, b, [](uint32_t x, uint32_t y) { return x + y; }); }
    static V sub(const V& a, const V& b) { return pk2(a, b, [](uint32_t x, uint32_t y) { return x - y; }); }
    static V add_sat(const V& a, const V& b) { return pk2(a, b, [](uint32_t x, uint32_t y) { return x + y > 0xffffu ? 0xffffu : x + y; }); }
    static V sub_sat(const V& a, const V& b) { return pk2(a, b, [](uint32_t x, uint32_t y) { return x > y ? x - y : 0u; }); }
    static V max_u(const V& a, const V& b) { return pk2(a, b, [](uint32_t x, uint32_t y) { return x > y ? x : y; }); }
    static V min_u(const V& a, const V& b) { return pk2(a, b, [](uint32_t x, uint32_t y) { return x < y ? x : y; }); }
    static V max_i(const V& a, const V& b) {
        return pk2(a, b, [](uint32_t x, uint32_t y) { return (int16_t)x > (int16_t)y ? x : y; });
    }
    static V mul(const V& a, const V& b) { return pk2(a, b, [](uint32_t x, uint32_t y) { return x * y; }); }
    static V mad(const V& a, const V& b, const V& c) { return add(mul(a, b), c); }
    static V and_(const V& a, const V& b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x & y; }); }
    static V xor_(const V& a, const V& b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x ^ y; }); }
    static V and_or(const V& a, const V& b, const V& c) {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = (a.l[i] & b.l[i]) | c.l[i];
        return r;
    }
    static V bfi(const V& m, const V& a, const V& b) {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = (m.l[i] & a.l[i]) | (~m.l[i] & b.l[i]);
        return r;
    }
    static V lshl_or(const V& a, int n, const V& b) {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = (a.l[i] << n) | b.l[i];
        return r;
    }
    static V shr(const V& a, int n) {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = a.l[i] >> n;
        return r;
    }
    template <int N> static V shr1(const V& x, const V&) {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = (i % N == 0) ? 0u : x.l[i - 1];
        return r;
    }
    template <int N> static V shr_d(const V& x, int d) {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = (i % N >= d) ? x.l[i - d] : 0u;
        return r;
    }
    template <int N> static V group_or(const V& x) {
        V r;
        for (int g = 0; g < 64 / N; ++g) {
            uint32_t o = 0;
            for (int i = 0; i < N; ++i) o |= x.l[g * N + i];
            for (int i = 0; i < N; ++i) r.l[g * N + i] = o;
        }
        return r;
    }
    static bool any(const V& x) {
        for (auto e : x.l)
            if (e) return true;
        return false;
    }
    static V lead_ones(const V& x, int nv) {
        V r;
        for (int i = 0; i < 64; ++i) {
            int c = 0;
            while (c < nv && ((x.l[i] >> (nv - 1 - c)) & 1u)) ++c;
            r.l[i] = (uint32_t)c;
        }
        return r;
    }
};

struct Read {
    std::vector<uint8_t> seq;
    std::vector<uint8_t> flags;  // oracle: [R][nv][N]
    size_t rend;                 // rows r <= rend are compared (the kernel stops a read at its r_end)
};

template <int N, int NV>
bool run_wave(const std::vector<uint8_t>& ref, std::vector<Read>& reads, const ProfileWeights& pw, const ByteIndexMap& map, int go, int ge,
              size_t first_flag_row, const char* tag) {
    using O = HostOps;
    using V = HV;
    constexpr int RPW = 2 * 64 / N;
    const size_t R = ref.size();
    zsw_pk::Consts<O, NV> c;
    c.go2 = O::splat((uint32_t)go * zsw_pk::ONE2);
    c.ge2 = O::splat((uint32_t)ge * zsw_pk::ONE2);
    c.nvge2 = O::splat((uint32_t)(NV * ge) * zsw_pk::ONE2);
    c.keep = O::splat(0);
    c.one = O::splat(zsw_pk::ONE2);
    c.ge = (uint32_t)ge;
    zsw_pk::State<O, NV> st;
    for (int v = 0; v < NV; ++v) st.H[v] = st.E[v] = O::splat(0);
    // residue codes per (v, lane, half)
    std::vector<int> kq(NV * 64 * 2, -1);
    for (int lane = 0; lane < 64; ++lane)
        for (int half = 0; half < 2; ++half) {
            const int rd = (lane / N) * 2 + half;
            if (rd >= (int)reads.size()) continue;
            for (int v = 0; v < NV; ++v) {
                const size_t q = (size_t)v + (size_t)(lane % N) * NV;
                if (q < reads[rd].seq.size()) kq[(v * 64 + lane) * 2 + half] = (int)map.to_index(reads[rd].seq[q]);
            }
        }
    for (size_t r = 0; r < R; ++r) {
        const int ri = (int)map.to_index(ref[r]);
        V p[NV], flg[NV], act;
        for (int v = 0; v < NV; ++v)
            for (int lane = 0; lane < 64; ++lane) {
                uint32_t w = 0;
                for (int half = 0; half < 2; ++half) {
                    const int k = kq[(v * 64 + lane) * 2 + half];
                    const int s = k < 0 ? 0 : pw.w[ri][k] - pw.bias;
                    w |= ((uint32_t)(uint16_t)(int16_t)s) << (16 * half);
                }
                p[v].l[lane] = w;
            }
        for (int lane = 0; lane < 64; ++lane) {
            uint32_t a = 0;
            for (int half = 0; half < 2; ++half) {
                const int rd = (lane / N) * 2 + half;
                if (rd < (int)reads.size() && r <= reads[rd].rend) a |= 0xffffu << (16 * half);
            }
            act.l[lane] = a;
        }
        if (r >= first_flag_row) {
            zsw_pk::row<O, N, NV, true>(st, p, act, c, flg, [] {});
            for (int rd = 0; rd < (int)reads.size() && rd < RPW; ++rd) {
                if (r > reads[rd].rend) continue;
                const int g = rd / 2, half = rd % 2;
                for (int v = 0; v < NV; ++v)
                    for (int li = 0; li < N; ++li) {
                        const uint32_t got = (flg[v].l[g * N + li] >> (16 * half)) & 0xffffu;
                        const uint32_t want = reads[rd].flags[(r * NV + v) * N + li];
                        if (got != want) {
                            fprintf(stderr, "%s N=%d NV=%d read %d (len %zu) row %zu v %d lane %d: oracle %u kernel-twin %u\n", tag, N, NV, rd,
                                    reads[rd].seq.size(), r, v, li, want, got);
                            return false;
                        }
                    }
            }
        } else {
            zsw_pk::row<O, N, NV, false>(st, p, act, c, flg, [] {});
        }
    }
    return true;
}

template <int N, int NV>
bool check(std::mt19937_64& rng, int iter) {
    std::uniform_int_distribution<int> pick(0, 1 << 20);
    const ByteIndexMap map = ByteIndexMap::dna_profile_map();
    const int schemes[][5] = {{2, -5, -10, -1, 1}, {4, -2, -3, -1, 1}, {3, -1, -4, -1, 1}, {2, -3, 0, 0, 1},  {1, -1, -2, -2, 0},
                              {5, -4, -1, 0, 1},  {2, -5, -5, -5, 1}, {10, -10, -5, -5, 1}, {1, -3, -6, -2, 0}, {2, -2, -1, -1, 1}};
    const int* sc = schemes[pick(rng) % 10];
    WeightMatrixI8 wm = WeightMatrixI8::make(map, (int8_t)sc[0], (int8_t)sc[1], sc[4] ? 'N' : -1);
    ProfileWeights pw = ProfileWeights::from(wm, true);
    const size_t R = 30 + pick(rng) % 150;
    std::vector<uint8_t> ref(R);
    const char* alpha = "ACGTN";
    const int mode = pick(rng) % 3;
    for (auto& ch : ref) ch = (uint8_t)alpha[mode == 1 ? pick(rng) % 2 : pick(rng) % 4];
    constexpr int RPW = 2 * 64 / N;
    std::vector<Read> reads(RPW - (pick(rng) % 3 == 0 ? pick(rng) % RPW : 0));
    if (reads.empty()) reads.resize(1);
    for (auto& rd : reads) {
        const size_t lo = (size_t)(NV - 1) * N + 1, hi = (size_t)NV * N;
        const size_t L = lo + pick(rng) % (hi - lo + 1);
        rd.seq.resize(L);
        size_t ppos = pick(rng) % R;
        const int style = pick(rng) % 4;
        for (size_t i = 0; i < L; ++i) {
            const int d = pick(rng) % 100;
            if (style == 3) {
                rd.seq[i] = (uint8_t)alpha[pick(rng) % 5];
                continue;
            }
            if (d < 5) ppos += 1 + pick(rng) % 3;
            if (d >= 5 && d < 10 && ppos > 0) --ppos;
            uint8_t chx = ref[ppos % R];
            if (d >= 10 && d < 18) chx = (uint8_t)alpha[pick(rng) % 5];
            rd.seq[i] = chx;
            ++ppos;
        }
        auto prof = StripedProfile<int32_t, N>::make(rd.seq.data(), L, pw, map, sc[2], sc[3]);
        sw_simd_align<int32_t, N>(ref.data(), R, prof, &rd.flags);
        rd.rend = pick(rng) % 3 == 0 ? pick(rng) % R : R - 1;  // some reads stop early, like reads past their r_end
    }
    char tag[96];
    snprintf(tag, sizeof tag, "iter %d scheme %d/%d/%d/%d", iter, sc[0], sc[1], sc[2], sc[3]);
    if (!run_wave<N, NV>(ref, reads, pw, map, -sc[2], -sc[3], 0, tag)) return false;
    // the first rows through the flag-less scan path: the state they leave must give the same flags afterwards
    return run_wave<N, NV>(ref, reads, pw, map, -sc[2], -sc[3], 1 + pick(rng) % (R - 1), tag);
}

template <int N>
bool check_all_nv(std::mt19937_64& rng, int iter) {
    return check<N, 1>(rng, iter) && check<N, 2>(rng, iter) && check<N, 3>(rng, iter) && check<N, 4>(rng, iter) &&
           check<N, 5>(rng, iter) && check<N, 7>(rng, iter) && check<N, 10>(rng, iter) && check<N, 13>(rng, iter) &&
           check<N, 16>(rng, iter) && check<N, 17>(rng, iter) && check<N, 25>(rng, iter) && check<N, 32>(rng, iter);
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 20261004ull);
    for (int it = 0; it < iters; ++it)
        if (!check_all_nv<8>(rng, it) || !check_all_nv<16>(rng, it) || !check_all_nv<32>(rng, it) || !check_all_nv<64>(rng, it) ||
            !check_all_nv<4>(rng, it) || !check_all_nv<2>(rng, it))
            return 1;
    printf("packed row update (zsw_align_pk.hpp) == oracle flags on %d x 6 lane counts x 12 vector counts x 2 runs\n", iters);
    return 0;
}
