// gemm256w.h — k_gemm256w: Linear8bit.forward (W8A16) on the skeleton of k_gemm256p (gemm256.h):
//   out[M, N] = X[M, K] . RNE_T(W_i8[N, K] * (scales[n] / 127))^T + bias            (nn/linear8bit.py:70-102)
// 256 x 256 x 64 tile, 8 waves, one workgroup per CU; activations and the raw int8 weights arrive by LDS-DMA
// (activations with the bank swizzle applied to the source address, each lane's own 32 weight bytes three k-steps
// ahead into a private raw slot); decode = sign-extend -> f32 -> * (scale/127) -> RNE 16 bit (the bits
// dequantize_rowwise produces, functional.py:628-636) -> ds_write_b128 into the next stage, interleaved with
// v_mfma_f32_32x32x16 on the current one; one barrier per k-step between MFMA groups 2 and 3; counted vmcnt.
// The row scale is constant along K, so there is no absmax traffic and no table.  LDS: 4 x 32 KiB images +
// 2 x 16 KiB raw slots = 160 KiB exactly (no static LDS).  Needs K % 64 == 0 and 16-bit types.
#pragma once
#include "gemm256.h"

namespace mbnb {

template <typename T, int WF = W8_INT8>
__global__ __launch_bounds__(512, 2) void k_gemm256w(const T *__restrict__ X, typename I8ProducerRT<T, WF>::Params wp,
                                                     const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                     int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    constexpr int RAWW = 16384;   // one raw slot: 8 waves x 2 pieces x 1 KiB
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;

    const int64_t tiles_m = (M + 255) >> 8, tiles_n = (N + 255) >> 8;
    const int64_t nwg = tiles_m * tiles_n;
    int64_t bid = blockIdx.x;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    int64_t tm, tn;
    if ((tiles_m % 4 == 0) && (tiles_n % 8 == 0)) {
        const int64_t patch = bid >> 5, within = bid & 31;
        const int64_t patches_m = tiles_m >> 2;
        tm = (patch % patches_m) * 4 + (within & 3);
        tn = (patch / patches_m) * 8 + (within >> 2);
    } else {
        tm = bid % tiles_m;
        tn = bid / tiles_m;
    }
    const int64_t m0 = tm << 8, n0 = tn << 8;

    // ---- activation pieces: wave w moves pieces 4w .. 4w+3 (8 rows x 128 B each), swizzle on the source
    const T *a_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_src[i] = X + m * K + 8 * c;
    }
    auto issue_a = [&](int stage, int64_t k0, int first, int count) {
#pragma unroll
        for (int i = first; i < first + count; i++) {
            auto g = (const __attribute__((address_space(1))) void *)(a_src[i] + k0);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_A + stage * P_IMG + (wave * 4 + i) * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };

    // ---- weight decode role (as k_gemm256p): rows 32*wave .. +31 of the tile, 2 k-halves of 32
    const int l32 = lane & 31;
    const int b_row = 32 * wave + 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);
    const int b_half = l32 >> 4;
    int64_t bn = n0 + b_row;
    bn = bn < N ? bn : N - 1;
    const int8_t *w_src = wp.w + bn * wp.K_weight + 32 * b_half;
    const float sc = w8_row_scale<WF>(wp.scales[bn]);   // int8: q.float() * (scales / 127.0); fp8: decode(byte) * scale
    auto issue_raw = [&](int rs, int64_t k0) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            auto g = (const __attribute__((address_space(1))) void *)(w_src + k0 + 16 * h);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_RAW + rs * RAWW + wave * 2048 + h * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };
    u32x4 rw[2][2];   // raw bytes of the tile being decoded, by tile parity: 32 int8
    auto load_raw = [&](auto pp, int rs) {
        constexpr int P = decltype(pp)::value;
#pragma unroll
        for (int h = 0; h < 2; h++)
            rw[P][h] = *reinterpret_cast<const u32x4 *>(smem + P_RAW + rs * RAWW + wave * 2048 + h * 1024 + lane * 16);
    };
    int bw_off[4];
#pragma unroll
    for (int d = 0; d < 4; d++) bw_off[d] = P_B + swz_off(b_row, 4 * b_half + d);
    // quarter d = 8 consecutive k: bytes 8d .. 8d+7 of the lane's 32
    auto decode_q = [&](const u32x4 (&r)[2], int d, int stage) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = r[d >> 1][2 * (d & 1) + (j >> 1)];
            const float q0 = w8_decode_sel<WF>(w, 2 * (j & 1));
            const float q1 = w8_decode_sel<WF>(w, 2 * (j & 1) + 1);
            o[j] = pack2<T>(q0 * sc, q1 * sc);
        }
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[d]) = o;
    };

    // ---- fragment reads
    const int fr = lane & 31, fh = lane >> 5;
    int fw[4], fx[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = fr * ROW_BYTES + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
        fw[s] = P_B + wn * 128 * ROW_BYTES + f;
        fx[s] = P_A + wm * 64 * ROW_BYTES + f;
    }
    auto read_frags = [&](int stage, int s, Frag (&wf)[4], Frag (&xf)[2]) {
#pragma unroll
        for (int i = 0; i < 4; i++) wf[i] = *reinterpret_cast<const Frag *>(smem + fw[s] + stage * P_IMG + i * 32 * ROW_BYTES);
#pragma unroll
        for (int j = 0; j < 2; j++) xf[j] = *reinterpret_cast<const Frag *>(smem + fx[s] + stage * P_IMG + j * 32 * ROW_BYTES);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    auto mfma_group = [&](const Frag (&wf)[4], const Frag (&xf)[2]) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = Mfma<T>::run(wf[i], xf[j], acc[i][j]);
    };

    const int64_t nk = K >> 6;
    const int64_t k_last = (nk - 1) << 6;
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // ---- prologue: stage 0 <- tile 0; raw(1) in registers; A(1), raw(2) in flight
    issue_a(0, 0, 0, 4);
    issue_raw(0, 0);
    issue_raw(1, kclamp(1));
    MBNB_VMCNT(0);
    __syncthreads();
    load_raw(P0{}, 0);
#pragma unroll
    for (int d = 0; d < 4; d++) decode_q(rw[0], d, 0);
    load_raw(P1{}, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_a(1, kclamp(1), 0, 4);
    issue_raw(0, kclamp(2));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frag wfA[4], xfA[2], wfB[4], xfB[2];
    read_frags(0, 0, wfA, xfA);

    // one k-step with compile-time stage parity C: stage C holds tile j; tile j+1 (raw registers rw[Nn]) is decoded
    // into stage Nn, quarter 0 in group 0, 1 in group 1, 2 and 3 in group 2
    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        using PC = std::integral_constant<int, C>;
        // group 0
        read_frags(C, 1, wfB, xfB);
        mfma_group(wfA, xfA);
        decode_q(rw[Nn], 0, Nn);
        if (j > 0) issue_a(Nn, kclamp(j + 1), 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        // group 1
        read_frags(C, 2, wfA, xfA);
        mfma_group(wfB, xfB);
        decode_q(rw[Nn], 1, Nn);
        if (j > 0) issue_a(Nn, kclamp(j + 1), 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        // group 2
        read_frags(C, 3, wfB, xfB);
        mfma_group(wfA, xfA);
        decode_q(rw[Nn], 2, Nn);
        decode_q(rw[Nn], 3, Nn);
        issue_raw(Nn, kclamp(j + 3));
        __builtin_amdgcn_sched_barrier(0);
        MBNB_VMCNT(2);                                        // all but raw(j+3) landed: A(j+1), raw(j+2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // own decode writes + fragment reads done
        __builtin_amdgcn_s_barrier();                         // stage Nn complete, stage C free
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // group 3
        read_frags(Nn, 0, wfA, xfA);
        load_raw(PC{}, C);  // raw(j+2) sits in slot (j+2) & 1 = C
        mfma_group(wfB, xfB);
        issue_a(C, kclamp(j + 2), 0, 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    MBNB_VMCNT(0);

    // ---- epilogue (as k_gemm256p)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (out_dtype != MBNB_F32) {
        char *wave_lds = smem + wave * 64 * 264;
        if (out_dtype == MBNB_F16)
            epilogue_staged<T, f16_t>(acc, wave_lds, bias, static_cast<f16_t *>(out_v), M, N, m0 + wm * 64, n0 + wn * 128, lane);
        else
            epilogue_staged<T, bf16_t>(acc, wave_lds, bias, static_cast<bf16_t *>(out_v), M, N, m0 + wm * 64, n0 + wn * 128, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t m = m0 + wm * 64 + j * 32 + fr;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t nn = n0 + wn * 128 + i * 32 + 8 * g + 4 * fh;
                if (m >= M || nn >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float s = acc[i][j][4 * g + e];
                    if (bias != nullptr && nn + e < N) s += to_f32(bias[nn + e]);
                    v[e] = to_f32(from_f32<T>(s));
                }
                store4(static_cast<float *>(out_v) + m * N + nn, v, nn, N);
            }
        }
}

constexpr int gemm256w_lds_bytes() { return P_RAW + 2 * 16384; }

}  // namespace mbnb
