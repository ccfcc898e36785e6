// ubench.hip — issue-rate microbenchmark for the VALU instructions the SW kernels could use (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
// Prints wave-instructions per cycle-equivalent relative to v_fma_f32, per SIMD, at 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>

#define ITERS 4096
#define UNR 16  // instructions per loop iteration (8 independent chains x 2)

#define DEF_KERNEL(NAME, ASM)                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, unsigned seed) {            \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                                      \
        unsigned b = seed * 0x9E3779B9u + 12345u, c = seed ^ 0x5bd1e995u;                       \
        for (int i = 0; i < ITERS; ++i) {                                                       \
            asm volatile(ASM("%0") "\n" ASM("%1") "\n" ASM("%2") "\n" ASM("%3") "\n" ASM("%4") "\n" ASM("%5") "\n" ASM("%6") "\n" ASM("%7") "\n" \
                         ASM("%0") "\n" ASM("%1") "\n" ASM("%2") "\n" ASM("%3") "\n" ASM("%4") "\n" ASM("%5") "\n" ASM("%6") "\n" ASM("%7")      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c));                                                     \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;            \
    }

#define A_FMA(X) "v_fma_f32 " X ", " X ", %8, %9"
#define A_PKFMA32(X) "v_fma_f32 " X ", " X ", %8, %9"
#define A_ADDU32(X) "v_add_u32 " X ", " X ", %8"
#define A_MAXI32(X) "v_max_i32 " X ", " X ", %8"
#define A_MAXF32(X) "v_max_f32 " X ", " X ", %8"
#define A_MAX3I32(X) "v_max3_i32 " X ", " X ", %8, %9"
#define A_ADD3(X) "v_add3_u32 " X ", " X ", %8, %9"
#define A_PKADDI16(X) "v_pk_add_i16 " X ", " X ", %8"
#define A_PKADDI16C(X) "v_pk_add_i16 " X ", " X ", %8 clamp"
#define A_PKSUBI16C(X) "v_pk_sub_i16 " X ", " X ", %8 clamp"
#define A_PKADDU16C(X) "v_pk_add_u16 " X ", " X ", %8 clamp"
#define A_PKMAXI16(X) "v_pk_max_i16 " X ", " X ", %8"
#define A_PKMAXU16(X) "v_pk_max_u16 " X ", " X ", %8"
#define A_PKMAXF16(X) "v_pk_max_f16 " X ", " X ", %8"
#define A_PKADDF16(X) "v_pk_add_f16 " X ", " X ", %8"
#define A_PKFMAF16(X) "v_pk_fma_f16 " X ", " X ", %8, %9"
#define A_PKMADI16(X) "v_pk_mad_i16 " X ", " X ", %8, %9"
#define A_PERM(X) "v_perm_b32 " X ", " X ", %8, %9"
#define A_BFI(X) "v_bfi_b32 " X ", " X ", %8, %9"
#define A_ANDOR(X) "v_and_or_b32 " X ", " X ", %8, %9"
#define A_MOV(X) "v_mov_b32 " X ", %8"
#define A_MAXI16(X) "v_max_i16 " X ", " X ", %8"
#define A_MAX3I16(X) "v_max3_i16 " X ", " X ", %8, %9"
#define A_MAX3F16(X) "v_max3_f16 " X ", " X ", %8, %9"
#define A_MAX3F32(X) "v_max3_f32 " X ", " X ", %8, %9"
#define A_PKMAX3F16(X) "v_pk_maximum3_f16 " X ", " X ", %8, %9"
#define A_PKMINF16(X) "v_pk_min_f16 " X ", " X ", %8"
#define A_SUBU32(X) "v_sub_u32 " X ", " X ", %8"
#define A_XOR(X) "v_xor_b32 " X ", " X ", %8"
#define A_LSHLADD(X) "v_lshl_add_u32 " X ", " X ", 1, %8"
#define A_PKFMAF32(X) "v_pk_mul_f32 " X ", " X ", " X

DEF_KERNEL(fma_f32, A_FMA)
DEF_KERNEL(add_u32, A_ADDU32)
DEF_KERNEL(sub_u32, A_SUBU32)
DEF_KERNEL(xor_b32, A_XOR)
DEF_KERNEL(max_i32, A_MAXI32)
DEF_KERNEL(max_f32, A_MAXF32)
DEF_KERNEL(max3_i32, A_MAX3I32)
DEF_KERNEL(max3_f32, A_MAX3F32)
DEF_KERNEL(add3_u32, A_ADD3)
DEF_KERNEL(lshl_add_u32, A_LSHLADD)
DEF_KERNEL(pk_add_i16, A_PKADDI16)
DEF_KERNEL(pk_add_i16_clamp, A_PKADDI16C)
DEF_KERNEL(pk_sub_i16_clamp, A_PKSUBI16C)
DEF_KERNEL(pk_add_u16_clamp, A_PKADDU16C)
DEF_KERNEL(pk_max_i16, A_PKMAXI16)
DEF_KERNEL(pk_max_u16, A_PKMAXU16)
DEF_KERNEL(pk_max_f16, A_PKMAXF16)
DEF_KERNEL(pk_min_f16, A_PKMINF16)
DEF_KERNEL(pk_add_f16, A_PKADDF16)
DEF_KERNEL(pk_fma_f16, A_PKFMAF16)
DEF_KERNEL(pk_mad_i16, A_PKMADI16)
DEF_KERNEL(perm_b32, A_PERM)
DEF_KERNEL(bfi_b32, A_BFI)
DEF_KERNEL(and_or_b32, A_ANDOR)
DEF_KERNEL(mov_b32, A_MOV)
DEF_KERNEL(max_i16, A_MAXI16)
DEF_KERNEL(max3_i16, A_MAX3I16)
DEF_KERNEL(max3_f16, A_MAX3F16)
DEF_KERNEL(pk_maximum3_f16, A_PKMAX3F16)

typedef void (*kfn)(unsigned*, unsigned);
struct Entry { const char* name; kfn fn; };

int main() {
    Entry es[] = {
#define E(N) {#N, k_##N}
        E(fma_f32), E(add_u32), E(sub_u32), E(xor_b32), E(max_i32), E(max_f32), E(max3_i32), E(max3_f32), E(add3_u32), E(lshl_add_u32),
        E(pk_add_i16), E(pk_add_i16_clamp), E(pk_sub_i16_clamp), E(pk_add_u16_clamp), E(pk_max_i16), E(pk_max_u16), E(pk_max_f16),
        E(pk_min_f16), E(pk_add_f16), E(pk_fma_f16), E(pk_mad_i16), E(perm_b32), E(bfi_b32), E(and_or_b32), E(mov_b32), E(max_i16),
        E(max3_i16), E(max3_f16), E(pk_maximum3_f16)};
    unsigned* d;
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves/SIMD
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double base = 0;
    for (auto& e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u + rep);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double wave_instr = (double)blocks * 4 * ITERS * UNR;
        double per_simd_per_s = wave_instr / (best * 1e-3) / 1024.0;
        if (base == 0) base = per_simd_per_s;
        // cycles per wave-instruction assuming v_fma_f32 = 2 cycles
        printf("%-20s %8.3f ms  %7.1f M wave-instr/s/SIMD  rel_to_fma %.2fx  (~%.2f cyc if fma=2)\n", e.name, best,
               per_simd_per_s / 1e6, per_simd_per_s / base, 2.0 * base / per_simd_per_s);
    }
    return 0;
}
