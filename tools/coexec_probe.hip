// coexec_probe.hip — does a wave's VALU / LDS / LDS-DMA work overlap with its SIMD partner's MFMAs?
// 512-thread workgroups (8 waves); waves 0-3 run role A, waves 4-7 role B, per mode.
//   build: hipcc -O3 --offload-arch=gfx950 tools/coexec_probe.hip -o tools/coexec_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void do_mfma(f32x16 (&acc)[8], bf16x8 a, bf16x8 b, int n) {
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
}
__device__ __forceinline__ float do_valu(float x, int n) {
    float v[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = v[i] * 1.0001f;
    }
    return v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
}
__device__ __forceinline__ float do_lds(const float *lut, unsigned idx, int n) {
    float s = 0;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float t = lut[(idx + i) & 15];
            idx = idx * 5 + 1 + (unsigned)(t > 10.f);
            s += t;
        }
    }
    return s;
}

// mode bits: roleA (waves 0-3) and roleB (waves 4-7): 0 idle, 1 MFMA, 2 VALU, 3 LDS lookups, 4 MFMA+VALU interleaved in one stream
__global__ __launch_bounds__(512, 2) void probe(float *out, int roleA, int roleB, int n) {
    __shared__ float lut[16];
    if (threadIdx.x < 16) lut[threadIdx.x] = threadIdx.x * 0.25f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? roleA : roleB;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; e++) { a[e] = (__bf16)(0.5f + threadIdx.x * 1e-3f); b[e] = (__bf16)(0.25f + e * 1e-2f); }
    float r = 0;
    if (role == 1) do_mfma(acc, a, b, n);
    else if (role == 2) r = do_valu((float)threadIdx.x, 4 * n);
    else if (role == 3) r = do_lds(lut, threadIdx.x, 2 * n);
    else if (role == 4) {
        float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) v[(i + q) & 7] = v[(i + q) & 7] * 1.0001f;
            }
        }
        r = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
    }
    else if (role == 5 || role == 6) {
        // one stream: 8 MFMAs, then (role 5) 6 x ds_read_b128 issued BEFORE the next 8 MFMAs and consumed after them;
        // role 6: the same reads issued and waited for immediately (no overlap allowed)
        __shared__ __attribute__((aligned(16))) char tile[32768];
        u32x4 f[6];
        unsigned off = (threadIdx.x & 63) * 16;
        unsigned accum = 0;
        for (int it = 0; it < n; it++) {
            if (role == 5) {
#pragma unroll
                for (int q = 0; q < 6; q++) f[q] = *reinterpret_cast<const u32x4 *>(tile + ((off + q * 4096 + it * 64) & 32767 & ~15u));
            }
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (role == 6) {
#pragma unroll
                for (int q = 0; q < 6; q++) f[q] = *reinterpret_cast<const u32x4 *>(tile + ((off + q * 4096 + it * 64) & 32767 & ~15u));
            }
#pragma unroll
            for (int q = 0; q < 6; q++) accum += f[q][0] ^ f[q][3];
            __builtin_amdgcn_sched_barrier(0);
        }
        r = (float)accum;
    }
    else if (role >= 7 && role <= 10) {
        // LDS mix of one k-step of k_gemm256p per wave: 24 ds_read_b128 (conflict-free rows), 32 ds_read_b32 from a
        // 16-entry table, 4 ds_write_b128.  role 7: all three; 8: only the b128 reads; 9: only the table reads; 10: only writes
        __shared__ __attribute__((aligned(16))) char img[65536];
        const int lane = threadIdx.x & 63;
        const int fr = lane & 31, fh = lane >> 5;
        unsigned faddr = fr * 128 + ((fh ^ ((fr >> 1) & 7)) << 4) + (wave & 1) * 8192;
        unsigned waddr = (threadIdx.x >> 1) * 128 + (((threadIdx.x & 1) * 4) << 4);
        unsigned accum = 0; float facc = 0; unsigned idx = threadIdx.x * 7;
        for (int it = 0; it < n; it++) {
            if (role == 7 || role == 8) {
#pragma unroll
                for (int q = 0; q < 24; q++) {
                    u32x4 f = *reinterpret_cast<const u32x4 *>(img + ((faddr + (q & 3) * 4096 + (q >> 2) * 32 + (it & 1) * 32768) & 65535));
                    accum += f[0] ^ f[3];
                }
            }
            if (role == 7 || role == 9) {
#pragma unroll
                for (int q = 0; q < 32; q++) { facc += lut[(idx >> (q & 15)) & 15]; }
                idx = idx * 1664525u + 1013904223u;
            }
            if (role == 7 || role == 10) {
#pragma unroll
                for (int q = 0; q < 4; q++) *reinterpret_cast<u32x4 *>(img + ((waddr + q * 16 + (it & 1) * 32768) & 65535)) = u32x4{accum, idx, accum, idx};
            }
        }
        r = (float)accum + facc;
    }
    else if (role >= 11 && role <= 14) {
        // the decode writes of k_gemm256p with its real (swizzled) addresses: 4 x 16 B per lane and k-step;
        // role 11: ds_write_b128, 12: 2 x ds_write_b64, 13: 4 x ds_write_b32, 14: ds_write_b128 to a linear (tid * 64) image
        __shared__ __attribute__((aligned(16))) char img2[65536];
        const int lane = threadIdx.x & 63, l32 = lane & 31;
        const int b_row = 32 * wave + 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);
        const int b_half = l32 >> 4;
        unsigned off[4];
        for (int d = 0; d < 4; d++) {
            const int chunk = 4 * b_half + d;
            off[d] = role == 14 ? threadIdx.x * 64 + d * 16 : b_row * 128 + ((chunk ^ ((b_row >> 1) & 7)) << 4);
        }
        unsigned x = threadIdx.x;
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                char *p = img2 + off[d] + (it & 1) * 32768;
                if (role == 11 || role == 14) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)p), "v"(u32x4{x, x, x, x}) : "memory");
                else if (role == 12) {
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:8" ::"v"((unsigned)(size_t)p), "v"(u32x2{x, x}) : "memory");
                } else {
                    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:4\n\tds_write_b32 %0, %1 offset:8\n\tds_write_b32 %0, %1 offset:12" ::"v"((unsigned)(size_t)p), "v"(x) : "memory");
                }
            }
            x += it;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        r = (float)x;
    }
    float s = r;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    const int n = 2000;
    struct { int a, b; const char *name; } modes[] = {
        {1, 0, "MFMA | idle"}, {0, 1, "idle | MFMA"}, {1, 1, "MFMA | MFMA"}, {2, 0, "VALU | idle"}, {2, 2, "VALU | VALU"},
        {1, 2, "MFMA | VALU"}, {2, 1, "VALU | MFMA"}, {3, 0, "LDS  | idle"}, {3, 3, "LDS  | LDS"}, {1, 3, "MFMA | LDS"},
        {3, 1, "LDS  | MFMA"}, {4, 0, "MFMA+VALU one stream | idle"}, {4, 4, "MFMA+VALU | MFMA+VALU"},
        {5, 0, "MFMA+6xb128 prefetched | idle"}, {5, 5, "MFMA+6xb128 prefetched | same"},
        {6, 0, "MFMA+6xb128 in place | idle"}, {6, 6, "MFMA+6xb128 in place | same"},
        {7, 7, "LDS k-step mix (all 8 waves)"}, {8, 8, "  only 24 x ds_read_b128"}, {9, 9, "  only 32 x ds_read_b32 table"},
        {10, 10, "  only 4 x ds_write_b128"}, {11, 11, "4 x ds_write_b128 real swizzle"}, {12, 12, "8 x ds_write_b64 real swizzle"},
        {13, 13, "16 x ds_write_b32 real swizzle"}, {14, 14, "4 x ds_write_b128 linear"}, {11, 0, "4 x ds_write_b128 real, 4 waves"}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &m : modes) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, m.a, m.b, n);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, m.a, m.b, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us\n", m.name, ms / 5 * 1e3);
    }
    return 0;
}
