// lds_write_probe.hip — cost of the decoded-weight LDS writes of k_gemm256p (4 x 16 B per lane and k-step, real
// swizzled addresses), one workgroup per CU.  MODE 0: ds_write_b128 x4, constant data; 1: data rewritten by VALU
// after every write (WAR on the data registers); 2: 8 x ds_write_b64; 3: writes interleaved with 24 ds_read_b128
// (the fragment reads of a k-step); 4: only the 24 reads.
//   hipcc -O3 --offload-arch=gfx950 lds_write_probe.hip -o lds_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int WAVES> __global__ __launch_bounds__(WAVES * 64) void probe(unsigned *out, int n) {
    __shared__ __attribute__((aligned(16))) char img[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l32 = lane & 31;
    const int b_row = 32 * wave + 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);
    const int b_half = l32 >> 4;
    unsigned off[4];
#pragma unroll
    for (int d = 0; d < 4; d++) off[d] = b_row * 128 + (((4 * b_half + d) ^ ((b_row >> 1) & 7)) << 4);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned faddr = fr * 128 + ((fh ^ ((fr >> 1) & 7)) << 4) + (wave & 1) * 8192;
    u32x4 x[4];
#pragma unroll
    for (int d = 0; d < 4; d++) x[d] = u32x4{threadIdx.x + d, 1u, 2u, 3u};
    unsigned accum = 0;
    for (int it = 0; it < n; it++) {
        const unsigned st = (it & 1) * 32768;
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int q = 0; q < 24; q++) {
                u32x4 f;
                asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"((faddr + (q & 3) * 4096 + (q >> 2) * 32 + st) & 65535u) : "memory");
                if (q % 6 == 5 && MODE == 3) {
                    const int d = q / 6;
                    asm volatile("ds_write_b128 %0, %1" ::"v"(off[d] + (st ^ 32768)), "v"(x[d]) : "memory");
                }
                if (q == 23) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); accum += f[0]; }
            }
        } else {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (MODE == 2) {
                    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" ::"v"(off[d] + st), "v"(u32x2{x[d][0], x[d][1]}), "v"(u32x2{x[d][2], x[d][3]}) : "memory");
                } else {
                    asm volatile("ds_write_b128 %0, %1" ::"v"(off[d] + st), "v"(x[d]) : "memory");
                }
                if (MODE == 1) {
#pragma unroll
                    for (int e = 0; e < 4; e++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[d][e]) : "v"(it));
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (accum + x[0][0] + x[1][1] + x[2][2] + x[3][3] == 0x12345678u) out[threadIdx.x] = accum;
}

template <int MODE, int WAVES> void run(const char *name, unsigned *out) {
    const int n = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-56s %d waves: %7.1f ns per k-step\n", name, WAVES, ms / 3 / n * 1e6);
}

int main() {
    unsigned *out; hipMalloc(&out, 4096);
    run<0, 8>("4 x ds_write_b128, constant data", out);
    run<0, 4>("4 x ds_write_b128, constant data", out);
    run<1, 8>("4 x ds_write_b128, data rewritten after each write", out);
    run<1, 4>("4 x ds_write_b128, data rewritten after each write", out);
    run<2, 8>("8 x ds_write_b64", out);
    run<4, 8>("24 x ds_read_b128", out);
    run<3, 8>("24 x ds_read_b128 + 4 x ds_write_b128 interleaved", out);
    return 0;
}
