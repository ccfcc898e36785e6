// ubench_mix.hip — does a fast 32-bit VALU op (v_sub_u32, ~2 cycles alone) keep its rate when it is interleaved with 4-cycle packed
// ops in dependent chains, at 3 and at 8 waves per SIMD? Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mix.hip -o tools/ubench_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITERS 4096

// NCH independent chains; each chain step = PAT applied to its register. 16 chain-steps per loop iteration.
#define CH8(P) P("%0") P("%1") P("%2") P("%3") P("%4") P("%5") P("%6") P("%7")
#define DEF(NAME, PAT, SRC)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, unsigned seed) {                           \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,  \
                 a6 = a0 * 17, a7 = a0 * 19;                                                                  \
        unsigned b = seed * 0x9E3779B9u + 12345u, c = seed ^ 0x5bd1e995u;                                     \
        for (int i = 0; i < ITERS; ++i) {                                                                     \
            asm volatile(CH8(PAT) CH8(PAT)                                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), SRC(c));                                                                   \
        }                                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                          \
    }
#define VSRC(x) "v"(x)
#define SSRC(x) "s"(x)
// patterns: number of instructions per chain-step noted as INS
#define P_MAX3(X) "v_pk_maximum3_f16 " X ", " X ", %8, %8\n"
#define P_SUBV(X) "v_subrev_u32 " X ", %9, " X "\n"
#define P_SUBS(X) "v_subrev_u32 " X ", %9, " X "\n"
#define P_PKSUB(X) "v_pk_sub_i16 " X ", " X ", %8\n"
#define P_MIXV(X) P_MAX3(X) P_SUBV(X)
#define P_MIXS(X) P_MAX3(X) P_SUBS(X)
#define P_MIXP(X) P_MAX3(X) P_PKSUB(X)
#define P_COL(X) P_MAX3(X) P_SUBS(X) P_MAX3(X) P_MAX3(X) P_SUBS(X) "v_pk_add_u16 " X ", " X ", %8\n" "v_perm_b32 " X ", " X ", %8, %8\n"
#define P_COLP(X) P_MAX3(X) P_PKSUB(X) P_MAX3(X) P_MAX3(X) P_PKSUB(X) "v_pk_add_u16 " X ", " X ", %8\n" "v_perm_b32 " X ", " X ", %8, %8\n"

DEF(max3, P_MAX3, VSRC)
DEF(sub_v, P_SUBV, VSRC)
DEF(sub_s, P_SUBS, SSRC)
DEF(mix_max3_subv, P_MIXV, VSRC)
DEF(mix_max3_subs, P_MIXS, SSRC)
DEF(mix_max3_pksub, P_MIXP, VSRC)
DEF(column_with_sub32, P_COL, SSRC)
DEF(column_all_packed, P_COLP, VSRC)

// patterns over PAIRS of chains (X, Y): 4 pair-steps cover the 8 chains
#define PAIRS4(P) P("%0", "%1") P("%2", "%3") P("%4", "%5") P("%6", "%7")
#define DEFP(NAME, PAT, SRC)                                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, unsigned seed) {                           \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,  \
                 a6 = a0 * 17, a7 = a0 * 19;                                                                  \
        unsigned b = seed * 0x9E3779B9u + 12345u, c = seed ^ 0x5bd1e995u;                                     \
        for (int i = 0; i < ITERS; ++i) {                                                                     \
            asm volatile(PAIRS4(PAT) PAIRS4(PAT)                                                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), SRC(c));                                                                   \
        }                                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                          \
    }
#define M3(X) "v_pk_maximum3_f16 " X ", " X ", %8, %8\n"
#define SB(X) "v_subrev_u32 " X ", %9, " X "\n"
#define AD(X) "v_add_u32 " X ", %9, " X "\n"
// two packed, then two adjacent independent fast ops
#define Q_ADJ(X, Y) M3(X) M3(Y) SB(X) SB(Y)
// fast ops isolated between packed ops
#define Q_ISO(X, Y) M3(X) SB(Y) M3(Y) SB(X)
// column shape: h=max3 | hg=sub, hd=add (adjacent) | E=max3, G=max3 | F=sub, addr=add (adjacent)   [X: state chain, Y: side chain]
#define Q_COL(X, Y) M3(X) SB(X) AD(Y) M3(Y) M3(X) SB(X) AD(Y)
DEFP(pair_adjacent_fast, Q_ADJ, VSRC)
DEFP(pair_isolated_fast, Q_ISO, VSRC)
DEFP(column_paired_fast, Q_COL, VSRC)

typedef void (*kfn)(unsigned*, unsigned);
struct Entry { const char* name; kfn fn; double ins; };

int main() {
    Entry es[] = {{"max3", k_max3, 1}, {"sub_u32 vgpr", k_sub_v, 1}, {"sub_u32 sgpr", k_sub_s, 1}, {"max3+sub(vgpr)", k_mix_max3_subv, 2},
                  {"max3+sub(sgpr)", k_mix_max3_subs, 2}, {"max3+pk_sub", k_mix_max3_pksub, 2}, {"column: 3 max3 + 2 sub32 + add + perm", k_column_with_sub32, 7},
                  {"column: all packed", k_column_all_packed, 7},
                  {"2 max3 + 2 ADJACENT sub32 (x4 pairs)", k_pair_adjacent_fast, 2},   /* 4 instr per pair-step, 4 pair-steps = 16 = 8 chains x 2 */
                  {"2 max3 + 2 ISOLATED sub32", k_pair_isolated_fast, 2},
                  {"column: 3 max3 + (sub,add) + (sub,add)", k_column_paired_fast, 3.5}};
    unsigned* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps : {3, 8}) {
        const int blocks = 256 * wps;  // wps blocks of 4 waves per CU = wps waves/SIMD
        printf("---- %d waves per SIMD\n", wps);
        for (auto& e : es) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u + rep);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double wave_instr = (double)blocks * 4 * ITERS * 16 * e.ins;
            const double cyc = 2.38e9 * (best * 1e-3) * 1024.0 / wave_instr;  // SIMD cycles per wave-instruction at 2.38 GHz
            printf("%-40s %8.3f ms  %.2f cycles/instr  (%.1f cycles per pattern)\n", e.name, best, cyc, cyc * e.ins);
        }
    }
    return 0;
}
