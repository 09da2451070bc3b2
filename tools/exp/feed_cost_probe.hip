// feed_cost_probe.hip — what does it cost a wave that issues MFMAs back to back (one wave per SIMD, 256 workgroups) to ALSO feed LDS, by feeder type?
// Per iteration: 32 v_mfma_f32_16x16x32_bf16 (8 accumulators, asm, one per fenced slot) + P feeder operations spread over the slots:
//   mode 0 none;  1 LDS-DMA piece (s_mov m0 + s_nop + buffer_load_dwordx4 .. lds);  2 LDS-DMA with ONE m0 write per four pieces (instruction offsets);
//   3 global_load_dwordx4 into registers + ds_write_b128 of the piece loaded an iteration earlier;  4 global_load_dwordx4 only;  5 ds_write_b128 only.
// Source: an L2-resident window shared by all workgroups.  Prints shader cycles per iteration and per feeder operation beyond mode 0.
//   hipcc -O3 --offload-arch=gfx950 -o feed_cost_probe feed_cost_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int... I, class F> __device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F &&f) { sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f)); }

template <int MODE, int P, int BIG = 0>
__global__ __launch_bounds__(256, 1) void k(const char *src, int iters, uint64_t *cycles, float *sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];     // 4 waves x 2 stages x P KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    bf16x8 ab, bb;
    for (int e = 0; e < 8; e++) { ab[e] = (__bf16)((float)((int)((h >> e) & 255) - 128) * 0.01f); bb[e] = (__bf16)((float)((int)((h >> (e + 8)) & 255) - 128) * 0.01f); }
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 big[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) big[i][e] = 0.0f;
    u32x4 rd[2];
    rd[0] = rd[1] = u32x4{0, 0, 0, 0};
    const uint64_t pa = reinterpret_cast<uint64_t>(src);
    i32x4 rs = i32x4{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), 1 << 21, 0x00020000};
    for (int e = 0; e < 4; e++) rs[e] = __builtin_amdgcn_readfirstlane(rs[e]);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    const uint32_t lds_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + wave * (2 * P * 1024)));
    int voff[P];
    for (int p = 0; p < P; p++) voff[p] = (wave * P + p) * 8192 + lane * 16 + ((MODE == 2) ? -(p & 3) * 1024 + (p & 3) * 1024 : 0);
    u32x4 regs[P];
    for (int p = 0; p < P; p++) regs[p] = u32x4{h, h + 1, h + 2, h + 3};
    char *lw = smem + wave * (2 * P * 1024) + lane * 16;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const int stage = it & 1;
        const int soff = __builtin_amdgcn_readfirstlane((it & 15) * 65536);
        sfor<32>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            if constexpr (BIG == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[t & 7]) : "v"(ab), "v"(bb));
            else if constexpr ((t & 1) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(big[(t >> 1) & 3]) : "v"(ab), "v"(bb));
            constexpr int every = 32 / P;
            if constexpr (MODE != 0 && (t % every) == every - 1 && t / every < P) {
                constexpr int p = t / every;
                const uint32_t dst = lds_wave + (uint32_t)(stage * P * 1024 + p * 1024);
                if constexpr (MODE == 1) {
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff[p]), "s"(rs), "s"(soff) : "memory", "m0");
                } else if constexpr (MODE == 2) {
                    if constexpr ((p & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff[p]), "s"(rs), "s"(soff) : "memory", "m0");
                    else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff[p]), "s"(rs), "s"(soff), "n"((p & 3) * 1024) : "memory", "m0");
                } else if constexpr (MODE == 3) {
                    u32x4 old = regs[p];
                    asm volatile("ds_write_b128 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(lw + stage * P * 1024 + p * 1024)), "v"(old) : "memory");
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(regs[p]) : "v"(voff[p]), "s"(rs), "s"(soff) : "memory");
                } else if constexpr (MODE == 4) {
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(regs[p]) : "v"(voff[p]), "s"(rs), "s"(soff) : "memory");
                } else if constexpr (MODE == 6) {
                    u32x4 &r = rd[p & 1];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(lw + ((p * 1024) % (2 * P * 1024)))) : "memory");
                } else if constexpr (MODE == 5) {
                    asm volatile("ds_write_b128 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(lw + stage * P * 1024 + p * 1024)), "v"(regs[p]) : "memory");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // everything issued an iteration ago has landed (as a pipelined kernel would ask before its barrier)
        if constexpr (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        if constexpr (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int p = 0; p < P; p++) { u32x4 &r = regs[p]; asm volatile("" : "+v"(r)); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 17) cycles[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; i++) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[i][0])); s += v; }
    for (int p = 0; p < P; p++) s += (float)regs[p][0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    s += (float)(rd[0][0] ^ rd[1][1]);
    for (int i = 0; i < 4; i++) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(big[i][0])); s += v; }
    if (s == 123.456f) sink[0] = s + *reinterpret_cast<float *>(smem + threadIdx.x * 4);
}
template <int MODE, int P, int BIG = 0> double run(const char *src, uint64_t *cyc, float *sink) {
    const int iters = 4000, lds = 4 * 2 * P * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, P, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<MODE, P, BIG>), dim3(256), dim3(256), lds, 0, src, iters, cyc, sink);
    hipDeviceSynchronize();
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    return (double)h / iters;
}
template <int P> void row(const char *src, uint64_t *cyc, float *sink) {
    const double base = run<0, P>(src, cyc, sink);
    const double m1 = run<1, P>(src, cyc, sink), m2 = run<2, P>(src, cyc, sink), m3 = run<3, P>(src, cyc, sink), m4 = run<4, P>(src, cyc, sink), m5 = run<5, P>(src, cyc, sink);
    printf("%d feeder ops per 32 MFMAs (%.0f cycles bare): per op beyond bare -- LDS-DMA %.1f, LDS-DMA one m0 per four %.1f, load + ds_write %.1f (load alone %.1f, ds_write alone %.1f)\n",
           P, base, (m1 - base) / P, (m2 - base) / P, (m3 - base) / P, (m4 - base) / P, (m5 - base) / P);
}
template <int P> void row2(const char *src, uint64_t *cyc, float *sink) {
    const double b0 = run<0, P, 0>(src, cyc, sink), b1 = run<0, P, 1>(src, cyc, sink);
    printf("%d ops per iteration: 32 x 16x16x32 bare %.0f | ds_read_b128 %.1f, LDS-DMA (grouped m0) %.1f per op   ||   16 x 32x32x16 bare %.0f | ds_read_b128 %.1f, LDS-DMA (grouped m0) %.1f per op\n", P, b0,
           (run<6, P, 0>(src, cyc, sink) - b0) / P, (run<2, P, 0>(src, cyc, sink) - b0) / P, b1, (run<6, P, 1>(src, cyc, sink) - b1) / P, (run<2, P, 1>(src, cyc, sink) - b1) / P);
}
int main() {
    char *src; uint64_t *cyc; float *sink;
    hipMalloc(&src, 4 << 20); hipMemset(src, 1, 4 << 20); hipMalloc(&cyc, 64); hipMalloc(&sink, 64);
    row<4>(src, cyc, sink);
    row<8>(src, cyc, sink);
    row<16>(src, cyc, sink);
    row2<8>(src, cyc, sink);
    row2<16>(src, cyc, sink);
    return 0;
}
