// valu_rate_probe.hip — issue cost of the VALU instructions used by the 4-bit decode, relative to v_mul_f32.
// One wave per SIMD (256-thread workgroups, one per CU), 8 independent chains per instruction, s_memtime deltas.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM)                                                                               \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, unsigned long long *cyc, int n) {      \
        uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 * 11 + 4, \
                 r5 = r0 * 13 + 5, r6 = r0 * 17 + 6, r7 = r0 * 19 + 7;                                  \
        uint32_t s = 0x3f800001u + threadIdx.x, t = 0x00070503u;                                      \
        unsigned long long t0, t1;                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
        for (int i = 0; i < n; i++) {                                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                         : "v"(s), "v"(t));                                                           \
        }                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                    \
        if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345678u) out[threadIdx.x] = r0;            \
    }

#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n\t"
#define A_PKMUL(i) "v_pk_mul_f32 v[10:11], v[10:11], v[12:13]\n\t"
#define A_CVTBF(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %8\n\t"
#define A_CVTF16(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8\n\t"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 4, 8\n\t"
#define A_SDWA(i) "v_lshlrev_b32_sdwa %" #i ", %9, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define A_DOT2(i) "v_dot2_f32_bf16 %" #i ", %8, %9, %" #i "\n\t"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n\t"

BODY(k_mul, A_MUL)
BODY(k_cvtbf, A_CVTBF)
BODY(k_cvtf16, A_CVTF16)
BODY(k_perm, A_PERM)
BODY(k_bfe, A_BFE)
BODY(k_sdwa, A_SDWA)
BODY(k_and, A_AND)
BODY(k_dot2, A_DOT2)
BODY(k_fma, A_FMA)
BODY(k_add3, A_ADD3)
BODY(k_lshlor, A_LSHLOR)

// v_pk_mul_f32 needs aligned register pairs: written with explicit pairs
__global__ __launch_bounds__(256) void k_pkmul(uint32_t *out, unsigned long long *cyc, int n) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[8];
    for (int i = 0; i < 8; i++) r[i] = f2{1.0f + threadIdx.x * 1e-3f + i, 2.0f + i};
    f2 s = f2{1.0000001f, 0.9999999f};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[j]) : "v"(s));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += r[i][0] + r[i][1];
    if (acc == 123.456f) out[threadIdx.x] = 1;
}

int main() {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    const int n = 2000;
    struct { const char *name; void (*k)(uint32_t *, unsigned long long *, int); } ks[] = {
        {"v_mul_f32", k_mul}, {"v_pk_mul_f32", k_pkmul}, {"v_cvt_pk_bf16_f32", k_cvtbf}, {"v_cvt_pkrtz_f16_f32", k_cvtf16},
        {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe}, {"v_lshlrev_b32_sdwa", k_sdwa}, {"v_and_b32", k_and},
        {"v_dot2_f32_bf16", k_dot2}, {"v_fma_f32", k_fma}, {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshlor}};
    for (auto &k : ks) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k.k, dim3(256), dim3(256), 0, 0, out, cyc, n);
            hipDeviceSynchronize();
        }
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on gfx9 (constant clock): report ns per instruction and relative cost
        printf("%-24s %8.3f ticks/instr\n", k.name, (double)c / (n * 32.0));
    }
    return 0;
}
