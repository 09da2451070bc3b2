// gemm_dq.h — k_gemm_dq: matmul_4bit for large M in ONE launch, the 4-bit weight decoded ONCE per launch (reference:
// functional.py:680-773; above M = 512 the reference itself dequantises once, then multiplies, :753-767).
//
// The two-launch form (dequantize_4bit into a scratch, then k_gemm_dense) pays a write-bound pass (10 us at 4096^2) plus the
// boundary between the launches before the first MFMA; the fused kernels (k_gemm256s, k_gemm_fused4) decode every weight tile
// once per 256 activation rows -- 16 x the work at M = 4096, and the decode's table lookups saturate the LDS
// (profiles/r03_fused4_ablation.txt).  Here the 256 workgroups of ONE launch are k_gemm_dense (same tile, pipeline, fragment
// schedule, epilogue) AND, on the side, the dequantise pass: the 16 workgroups of a tile column (same 256 weight rows, tm =
// 0 .. 15) each decode 16 of those rows into the scratch Wd, one SLAB of 512 k (8 k-steps) at a time, three slabs ahead of the
// k-loop that consumes them, and hand them to each other through agent-scope flags:
//   producer  thread t: row 16 tm + (t >> 4), 32 values (16 packed bytes, one absmax) per slab; per loop iteration (two
//             k-steps) one packed dword -> 8 values -> one 16-byte WRITE-THROUGH store (sc0 sc1); after the slab's last store,
//             the next k-step's vmcnt + barrier (every wave's stores have left), then ONE lane adds 1 to flag[tn][slab];
//   consumer  one lane polls flag[tn][slab] (sc1 load, issued one iteration before it is looked at) until it reads 16, before
//             the k-step that issues the first LDS-DMA of the slab; a workgroup barrier lies between that and the DMA.
// Every line of Wd is written once per launch and read only behind its flag, so no cache can hold a stale copy of it
// (MI355X_MICROARCH.md, inter-workgroup visibility: write-through payload, drained, flag by one lane behind a barrier, sc1
// poll, barrier, loads).  The slabs 0-2 are produced in the prologue (startup: one hand-off before the first MFMA).
// Spins are bounded: on a timeout the workgroup sets sync[error] and goes on (wrong numbers, never a hang).
// Flags are counters that start at ZERO (caller's contract) and are zeroed again by the last workgroup of each column.
// Requirements (launcher): blocksize 64, plain f32 absmax, tiles_m == 16 (3840 < M <= 4096), tiles <= CUs (every workgroup
// resident: the hand-off needs it), K % 512 == 0, K_weight == K, 16-byte aligned operands.
// Output bits: those of dequantize_4bit + k_gemm_dense (same Wd bits, same pipeline).
#pragma once
#include "gemm_dense.h"

namespace mbnb {

constexpr int GQ_SLAB = 8;                 // k-steps per slab
constexpr int GQ_AHEAD = 3;                // slabs produced in the prologue; the loop decodes slab GQ_AHEAD + it / 4
constexpr int GQ_COL_WORDS = 66;           // per tile column: 64 slab flags, the done counter, one pad word
constexpr int GQ_MAX_SLABS = 64;
constexpr int GQ_LDS = GD_LDS + 4 * 2 * 1152;   // k_gemm_dense's images + the producers' wave-private exchange areas
constexpr int64_t gq_sync_bytes(int64_t tiles_n) { return (tiles_n * GQ_COL_WORDS + 2) * 4; }

// ABL (diagnostic builds under tools/exp; the product instantiates 0): 1 no side work in the loop at all, 2 no decode, 4 no stores,
// 8 no flag adds, 16 no polls, 32 no next-slab loads
template <typename T, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_gemm_dq(const T *__restrict__ X, const uint8_t *__restrict__ packed,
                                                    const float *__restrict__ absmax, int qt, T *__restrict__ Wd,
                                                    uint32_t *__restrict__ sync, const T *__restrict__ bias, void *__restrict__ out_v,
                                                    int out_dtype, int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma16<T>::frag;
    using Plan = GdPlan<8>;
    constexpr int FM = 8, TM = 256, PM = 4, PN = 8;
    __shared__ __attribute__((aligned(2048))) float s_lut2[512];
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;

    // ---- tile -> workgroup map (k_gemm_dense)
    const int64_t tiles_m = (M + TM - 1) / TM, tiles_n = (N + 255) >> 8;
    const int64_t nwg = tiles_m * tiles_n;
    int64_t bid = blockIdx.x;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    int64_t tm, tn;
    if ((tiles_m % PM == 0) && (tiles_n % PN == 0)) {
        const int64_t patch = bid >> 5, within = bid & 31;
        const int64_t patches_m = tiles_m / PM;
        tm = (patch % patches_m) * PM + (within % PM);
        tn = (patch / patches_m) * PN + (within / PM);
    } else {
        tm = bid % tiles_m;
        tn = bid / tiles_m;
    }
    const int64_t m0 = tm * TM, n0 = tn << 8;
    const int nk = (int)(K >> 6);
    const int nslab = nk / GQ_SLAB;
    const int64_t ldw = K;

    // ---- byte table: entry b = (code[b & 15], code[b >> 4])
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int e = tid * 2 + h, b = e >> 1, nib = (e & 1) ? (b >> 4) : (b & 15);
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (nib == i) v = (qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
        s_lut2[e] = v;
    }

    // ---- LDS-DMA of both operands (k_gemm_dense); B comes from the scratch this launch fills
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a, rs_b;
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K), pb = reinterpret_cast<uint64_t>(Wd + n0 * ldw);
        const int64_t rows_a = M - m0 < TM ? M - m0 : TM, rows_b = N - n0 < 256 ? N - n0 : 256;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K * 2), 0x00020000};
        rs_b = i32x4_t{(int)(uint32_t)pb, (int)(uint32_t)(pb >> 32), (int)(rows_b * ldw * 2), 0x00020000};
    }
    int voff_a[FM], voff_b[8];
#pragma unroll
    for (int pl = 0; pl < 8; pl++) {
        const int row = 8 * (8 * wave + pl) + (lane >> 3);
        voff_b[pl] = (int)(row * ldw * 2) + 16 * ((lane & 7) ^ ((row >> 1) & 7));
        voff_a[pl] = (int)(row * K * 2) + 16 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    struct DmaCtx { i32x4_t ra, rb; uint32_t lwa, lwb; };
    auto dma_ctx = [&]() {
        DmaCtx c;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c.ra[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
            c.rb[e] = __builtin_amdgcn_readfirstlane(rs_b[e]);
        }
        c.lwa = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(P_A + wave * FM * 1024)));
        c.lwb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(P_B + wave * 8192)));
        return c;
    };
    auto issue_piece = [&](auto qq, int stage, int kb, const DmaCtx &c) {
        constexpr int q = decltype(qq)::value, pl = q < FM ? q : q - FM;
        const uint32_t dst = (q < FM ? c.lwa : c.lwb) + (uint32_t)(stage * P_IMG + pl * 1024);
        const int vo = (q < FM) ? voff_a[pl] : voff_b[pl];
        const i32x4_t rs = (q < FM) ? c.ra : c.rb;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(kb) : "memory", "m0");
    };

    // ---- producer role.  Wave w owns weight rows 16 tm + 4 w + r (r = lane >> 4) of the column's 256; per slab and row 256 packed
    // bytes + 8 absmax.  LOAD: lane (r, p = lane & 15) fetches the row's packed bytes 16 p .. 16 p + 15 (one 16-byte load per
    // slab) and, for p < 8, absmax p of the slab; both go through a wave-private LDS area (1 KiB + 128 B per slab parity), because
    // the STORES want another distribution: in iteration q of the slab lane (r, p) decodes the row's packed dword 16 q + p
    // (k = 128 q + 8 p .. + 7) and stores 16 bytes at Wd[row][512 slab + 128 q + 8 p] -- the 16 lanes of a row write 256
    // contiguous bytes, whole 128-byte lines per write-through store (16-byte pieces at a 64-byte stride, which the
    // one-segment-per-lane form produced, cost 18 us in the loop and 8 us in the prologue: profiles/r03_dq_ablation.txt).
    const int pr = lane >> 4, pp = lane & 15;
    int64_t prow = n0 + 16 * tm + 4 * wave + pr;
    prow = prow < N ? prow : N - 1;                        // rows past N: a duplicate of the last row (same bytes, same address)
    const int64_t rowb = K >> 1, nblk = K >> 6;
    const uint8_t *p_src = packed + prow * rowb + 16 * pp;                  // + 256 slab
    const float *m_src = absmax + prow * nblk + (pp & 7);                   // + 8 slab
    T *w_dst = Wd + prow * ldw + 8 * pp;                                    // + 512 slab + 128 q   (elements)
    constexpr int GQ_ST = GD_LDS, GQ_ST_SLOT = 1152, GQ_ST_WAVE = 2 * GQ_ST_SLOT;
    const int st_w = GQ_ST + wave * GQ_ST_WAVE + pr * 256 + 16 * pp;                 // own 16 packed bytes      (+ slot)
    const int st_a = GQ_ST + wave * GQ_ST_WAVE + 1024 + pr * 32 + 4 * (pp & 7);       // own absmax               (+ slot)
    const int st_rw = GQ_ST + wave * GQ_ST_WAVE + pr * 256 + 4 * pp;                  // packed dword 16 q + p    (+ slot + 64 q)
    const int st_ra = GQ_ST + wave * GQ_ST_WAVE + 1024 + pr * 32 + 4 * (pp >> 3);     // absmax 2 q + (p >> 3)    (+ slot + 8 q)
    uint32_t *col_sync = sync + tn * GQ_COL_WORDS;
    uint32_t *err_word = sync + tiles_n * GQ_COL_WORDS;
    const uint32_t want = (uint32_t)tiles_m;
    const char *lut2 = reinterpret_cast<const char *>(s_lut2);

    // s_nop 1 behind the store: a store of more than 8 bytes reads its data registers for a few cycles after issue, and the
    // hazard recognizer that normally keeps the next VALU write away from them does not look into inline assembly (seen: the
    // first dword of a 16-byte store replaced by the next pack's output in lanes 12-15 of every 16)
    auto store_wt = [&](T *dst, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory"); };
    // 8 values of packed dword w (k order: low nibble first) scaled by am, rounded to T: dequantize_4bit's bits
    auto decode8 = [&](uint32_t w, float am) {
        u32x4 o;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((w >> (8 * b)) & 0xFFu) << 3));
            float p0, p1;
            asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(v[0]), "v"(am));
            asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(v[1]), "v"(am));
            o[b] = pack2<T>(p0, p1);
        }
        return o;
    };
    // one relaxed agent-scope poll of a flag word (one lane's value, broadcast)
    auto poll_once = [&](const uint32_t *p) {
        uint32_t v;
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    // a slab's raw data from registers into the wave's LDS area (slot = slab parity); the wave's LDS operations execute in
    // order, so its own later reads see it without a barrier
    auto stage_put = [&](int slot, u32x4 w, float a) {
        *reinterpret_cast<u32x4 *>(smem + st_w + slot * GQ_ST_SLOT) = w;
        *reinterpret_cast<float *>(smem + st_a + slot * GQ_ST_SLOT) = a;
    };

    // ---- prologue, part 1: activations of tiles 0 and 1 on their way; slabs 0 .. GQ_AHEAD-1 decoded, stored, flagged
    {
        const DmaCtx c0 = dma_ctx();
        gd_static_for<FM>([&](auto q) { issue_piece(q, 0, 0, c0); });
        gd_static_for<FM>([&](auto q) { issue_piece(q, 1, (nk > 1 ? 1 : 0) << 7, c0); });
    }
    u32x4 rwn;       // raw bytes / absmax of the slab AFTER the one being decoded in the loop (requested a slab ahead)
    float amn = 0.0f;
    {
        u32x4 r0[GQ_AHEAD + 1];
        float a0[GQ_AHEAD + 1];
#pragma unroll
        for (int u = 0; u <= GQ_AHEAD; u++) {
            const int uc = u < nslab ? u : nslab - 1;
            r0[u] = *reinterpret_cast<const u32x4 *>(p_src + 256 * uc);
            a0[u] = m_src[8 * uc];
        }
        {
            const int uc = GQ_AHEAD + 1 < nslab ? GQ_AHEAD + 1 : nslab - 1;
            rwn = *reinterpret_cast<const u32x4 *>(p_src + 256 * uc);
            amn = m_src[8 * uc];
        }
        __syncthreads();                                  // byte table
#pragma unroll
        for (int u = 0; u < GQ_AHEAD; u++) {
            stage_put(u & 1, r0[u], a0[u]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(smem + st_rw + (u & 1) * GQ_ST_SLOT + 64 * q);
                const float a = *reinterpret_cast<const float *>(smem + st_ra + (u & 1) * GQ_ST_SLOT + 8 * q);
                const u32x4 o = decode8(w, a);
                if (u < nslab) store_wt(w_dst + 512 * u + 128 * q, o);
            }
        }
        stage_put(GQ_AHEAD & 1, r0[GQ_AHEAD], a0[GQ_AHEAD]);     // the slab the loop decodes first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left (and the activations have landed)
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int u = 0; u < GQ_AHEAD; u++)
                if (u < nslab) asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(col_sync + u), "v"(1u) : "memory");
        }
        if (wave == 0) {     // slab 0 of every producer of the column
            int spins = 0;
            while (poll_once(col_sync) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) {
                    if (lane == 0) atomicOr(err_word, 1u);
                    break;
                }
            }
        }
        __syncthreads();
        const DmaCtx c0 = dma_ctx();
        gd_static_for<8>([&](auto q) { issue_piece(std::integral_constant<int, FM + decltype(q)::value>{}, 0, 0, c0); });
        gd_static_for<8>([&](auto q) { issue_piece(std::integral_constant<int, FM + decltype(q)::value>{}, 1, (nk > 1 ? 1 : 0) << 7, c0); });
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // everything but the weight pieces of tile 1
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- fragment reads (k_gemm_dense)
    const int r16 = lane & 15, fq = lane >> 4;
    int fw[2], fx[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const int f = r16 * ROW_BYTES + (((4 * ks + fq) ^ (r16 >> 1)) << 4);
        fw[ks] = P_B + wn * 128 * ROW_BYTES + f;
        fx[ks] = P_A + wm * 16 * FM * ROW_BYTES + f;
    }
    Frag wf[2][8], xf[2][FM];
    auto read_one = [&](int stage, auto kk, auto nn) {
        constexpr int ks = decltype(kk)::value, n = decltype(nn)::value;
        if constexpr (n == 0) wf[ks][0] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * P_IMG);
        else if constexpr (n <= FM) xf[ks][n - 1] = *reinterpret_cast<const Frag *>(smem + fx[ks] + stage * P_IMG + (n - 1) * 16 * ROW_BYTES);
        else wf[ks][n - FM] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * P_IMG + (n - FM) * 16 * ROW_BYTES);
    };
    f32x4 acc[8][FM];
    // MFMAs from assembly, accumulators pinned to their AGPR tuples (gemm_fused4.h: with the producer's live ranges in the loop
    // the allocator otherwise moves accumulators between the files)
    auto mfma_acc = [&](f32x4 &c, const Frag &a, const Frag &b) {
        if constexpr (std::is_same_v<T, bf16_t>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    };
    auto mfma_zero = [&](f32x4 &c, const Frag &a, const Frag &b) {
        if constexpr (std::is_same_v<T, bf16_t>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    };
    auto kbytes = [&](int t) { return (t < nk ? t : nk - 1) << 7; };
    gd_static_for<Plan::NR>([&](auto n) { read_one(0, std::integral_constant<int, 0>{}, n); });

    // producer / consumer state of the loop
    uint32_t dw = 0u;        // the iteration's packed dword
    float dam = 0.0f;        // ... and its absmax
    float Lr[4][2];
    u32x4 ob = {0u, 0u, 0u, 0u};
    uint32_t pollv = 0u;
    const uint32_t *flag_base = col_sync;
    {   // what k-step b of "iteration -1" would have done: iteration 0's dword, absmax and lookups
        dw = *reinterpret_cast<const uint32_t *>(smem + st_rw + (GQ_AHEAD & 1) * GQ_ST_SLOT);
        dam = *reinterpret_cast<const float *>(smem + st_ra + (GQ_AHEAD & 1) * GQ_ST_SLOT);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((dw >> (8 * b)) & 0xFFu) << 3));
            Lr[b][0] = v[0]; Lr[b][1] = v[1];
        }
    }

    // ---- one k-step (k_gemm_dense's slots) + the side work behind barrier 2.  HALF 0 = first k-step of a loop iteration,
    // 1 = second, -1 = the peeled first / last k-steps (no side work).  it = iteration number.
    auto kstep = [&](auto cc, auto first, auto wo_, auto half_, int j, int it, const DmaCtx &dc) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1, WO = decltype(wo_)::value, HALF = decltype(half_)::value;
        constexpr bool FIRST = decltype(first)::value;
        const int kb2 = __builtin_amdgcn_readfirstlane(kbytes(j + 2));
        const int q = it & 3;                          // packed dword of the slab decoded in this iteration
        const int u_dec = GQ_AHEAD + (it >> 2);        // slab decoded in this iteration
        const int u_poll = (it >> 2) + 1;              // slab whose first LDS-DMA comes up in iteration 4 u_poll - 2
        // the side operations of k-step a behind barrier 2 (wave-uniform conditions)
        const int c_flag = (!(ABL & 8) && q == 0 && it >= 4 && u_dec - 1 < nslab && wave == 0) ? 1 : 0;
        const int c_poll = (!(ABL & 16) && q == 0 && u_poll < nslab && wave == 0) ? 1 : 0;
        const int c_chk = (!(ABL & 16) && q == 1 && u_poll < nslab && wave == 0) ? 1 : 0;
        const int c_next = (!(ABL & 32) && q == 3) ? 1 : 0;
        const int c_store = (!(ABL & 4) && u_dec < nslab) ? 1 : 0;
        const int n_side = __builtin_amdgcn_readfirstlane(c_flag + c_poll + 2 * c_next + c_store);
        gd_static_for<Plan::NS>([&](auto tt) {
            constexpr int t = decltype(tt)::value, ks = t / 64, f = (t % 64) / FM, g = t % FM;
            if constexpr (t == Plan::B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (t == Plan::B2) {
                // k-step b: the side operations of k-step a (issued behind its last piece: flag add, poll, next-slab loads, the
                // iteration's write-through store) may stay in flight -- the acknowledgement of a write-through store or of an
                // agent-scope atomic comes from the memory side and takes longer than one k-step under load (waiting for the store
                // here cost 13 us).  Their number n_side is wave-uniform but varies, and the wait must leave EXACTLY the 16 pieces
                // of this k-step + n_side operations in flight (one more would be a piece of tile j+1), so the count is chosen by a
                // scalar branch inside the asm statement.  k-step a of the next iteration (vmcnt(16)) waits for all of them.
                if constexpr (HALF == 1 && !(ABL & 1)) {
                    asm volatile("s_cmp_lt_u32 %0, 1\n\ts_cbranch_scc1 .Lgq_w0%=\n\ts_cmp_lt_u32 %0, 2\n\ts_cbranch_scc1 .Lgq_w1%=\n\t"
                                 "s_cmp_lt_u32 %0, 3\n\ts_cbranch_scc1 .Lgq_w2%=\n\ts_waitcnt vmcnt(19)\n\ts_branch .Lgq_we%=\n"
                                 ".Lgq_w2%=:\n\ts_waitcnt vmcnt(18)\n\ts_branch .Lgq_we%=\n"
                                 ".Lgq_w1%=:\n\ts_waitcnt vmcnt(17)\n\ts_branch .Lgq_we%=\n"
                                 ".Lgq_w0%=:\n\ts_waitcnt vmcnt(16)\n"
                                 ".Lgq_we%=:" ::"s"(n_side) : "scc", "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Plan::NP) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (FIRST && ks == 0) mfma_zero(acc[f][g], wf[ks][f], xf[ks][g]);
            else mfma_acc(acc[f][g], wf[ks][f], xf[ks][g]);
            if constexpr ((t % Plan::RS1) == 0 && t / Plan::RS1 < Plan::NR)
                read_one(C, std::integral_constant<int, 1>{}, std::integral_constant<int, (t / Plan::RS1) % Plan::NR>{});
            if constexpr (t >= Plan::R0 && t < Plan::R0 + Plan::NR)
                read_one(Nn, std::integral_constant<int, 0>{}, std::integral_constant<int, (t - Plan::R0) % Plan::NR>{});
            if constexpr (t >= Plan::D0 && t < Plan::D0 + Plan::NP * Plan::DS && ((t - Plan::D0) % Plan::DS) == WO)
                issue_piece(std::integral_constant<int, ((t - Plan::D0) / Plan::DS) % Plan::NP>{}, C, kb2, dc);
            if constexpr (HALF >= 0 && !(ABL & 1)) {
                // ---- side work.  Every scalar condition is tested INSIDE its asm statement (a branch the compiler sees splits the
                // k-step into basic blocks and wrecks the register allocation, gemm_fused4.h).  LDS reads of the side work sit
                // between the two barriers, where the k-step has none of its own (behind barrier 2 a use of theirs would also wait
                // for the 16 fragment reads in front of it: that form cost 12 us).
                //   k-step a (HALF 0): the iteration's dword and absmax out of the wave's LDS area, 4 lookups, 8 products, 4 packs,
                //                      one 16-byte write-through store of a quarter of the slab's rows (whole lines per row);
                //                      behind barrier 2: poll issue (q == 0) / poll check (q == 1)
                //   k-step b (HALF 1): behind barrier 2 (the iteration's store is older than this k-step's pieces: every wave has
                //                      waited for it): q == 3 -> flag of the finished slab, the next slab's raw data into the
                //                      LDS area, the slab after that requested
                // The decode of an iteration is spread over TWO k-steps so that no LDS round trip is exposed (with the read, the
                // lookups and the products in one k-step the loop lost 10 us): k-step b of iteration it-1 reads the dword and looks
                // its four bytes up, k-step a of iteration it multiplies, packs and stores.
                if constexpr (HALF == 0) {
                    if constexpr ((t == 42 || t == 50 || t == 58 || t == 66) && !(ABL & 2)) {
                        constexpr int b = (t - 42) / 8;
                        const float l0 = Lr[b][0], l1 = Lr[b][1], a = dam;
                        float p0, p1;
                        asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(l0), "v"(a));
                        asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(l1), "v"(a));
                        ob[b] = pack2<T>(p0, p1);
                    }
                    if constexpr (t == 126 && !(ABL & 4)) {
                        const int go = __builtin_amdgcn_readfirstlane(c_store);
                        T *dst = w_dst + 512 * (u_dec < nslab ? u_dec : 0) + 128 * q;
                        asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 .Lgq_skip_s%=\n\tglobal_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1\n.Lgq_skip_s%=:"
                                     ::"v"(dst), "v"(ob), "s"(go) : "scc", "memory");
                    }
                    if constexpr (t == Plan::B2 + 16 && !(ABL & 8)) {
                        // flag of slab u_dec - 1, finished in the previous iteration (its last store is two k-steps old: barrier 2
                        // above has waited for it in every wave), by one lane of wave 0
                        const int fl = __builtin_amdgcn_readfirstlane(c_flag);
                        const uint32_t *fp = flag_base + (u_dec - 1 < nslab ? u_dec - 1 : 0);
                        uint64_t sv_exec;
                        asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lgq_skip_d%=\n\t"
                                     "s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %1, %2, off sc1\n\ts_mov_b64 exec, %0\n"
                                     ".Lgq_skip_d%=:"
                                     : "=&s"(sv_exec) : "v"(fp), "v"(1u), "s"(fl) : "scc", "memory");
                    }
                    if constexpr (t == Plan::B2 + 17) {
                        if constexpr (!(ABL & 16)) {
                            // q == 0: ask for the flag of the slab whose first LDS-DMA comes up in iteration 4 u_poll - 2 (looked at one
                            // iteration later: it is older than the next k-steps' pieces, so their vmcnt covers it)
                            const int pl = __builtin_amdgcn_readfirstlane(c_poll);
                            const uint32_t *pq = flag_base + (u_poll < nslab ? u_poll : 0);
                            asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 .Lgq_skip_e%=\n\tglobal_load_dword %0, %1, off sc1\n.Lgq_skip_e%=:"
                                         : "+v"(pollv) : "v"(pq), "s"(pl) : "scc", "memory");
                            // q == 1: that poll has landed; not all producers there yet -> poll in place (bounded)
                            const int ck = __builtin_amdgcn_readfirstlane(c_chk);
                            uint32_t cnt, cur;
                            asm volatile("s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 .Lgq_done%=\n\t"
                                         "s_mov_b32 %1, 0\n"
                                         ".Lgq_again%=:\n\t"
                                         "v_readfirstlane_b32 %2, %0\n\t"
                                         "s_cmp_ge_u32 %2, %5\n\ts_cbranch_scc1 .Lgq_done%=\n\t"
                                         "s_add_u32 %1, %1, 1\n\ts_cmp_gt_u32 %1, 0x400000\n\ts_cbranch_scc1 .Lgq_fail%=\n\t"
                                         "s_sleep 2\n\tglobal_load_dword %0, %3, off sc1\n\ts_waitcnt vmcnt(0)\n\ts_branch .Lgq_again%=\n"
                                         ".Lgq_fail%=:\n\tglobal_atomic_or %4, %7, off sc1\n"
                                         ".Lgq_done%=:"
                                         : "+v"(pollv), "=&s"(cnt), "=&s"(cur) : "v"(pq), "v"(err_word), "s"(want), "s"(ck), "v"(2u) : "scc", "memory");
                        }
                        if constexpr (!(ABL & 32)) {
                            // q == 3: the next slab's raw data (requested a slab ago) into the wave's LDS area, where k-step b of this
                            // iteration starts reading it; then the slab after it is requested (landed by the next barrier 2)
                            const int last = __builtin_amdgcn_readfirstlane(c_next);
                            const int so = __builtin_amdgcn_readfirstlane(((u_dec + 1) & 1) * GQ_ST_SLOT);
                            // asm LDS addresses are absolute: the dynamic region starts behind the static byte table
                            const int a_w = (int)smem_base + st_w + so, a_a = (int)smem_base + st_a + so;
                            asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 .Lgq_skip_w%=\n\tds_write_b128 %0, %1\n\tds_write_b32 %2, %3\n.Lgq_skip_w%=:"
                                         ::"v"(a_w), "v"(rwn), "v"(a_a), "v"(amn), "s"(last) : "scc", "memory");
                            const int un = u_dec + 2 < nslab ? u_dec + 2 : nslab - 1;
                            const uint8_t *ps = p_src + 256 * un;
                            const float *ms = m_src + 8 * un;
                            asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 .Lgq_skip_c%=\n\t"
                                         "s_nop 1\n\tglobal_load_dwordx4 %0, %2, off\n\tglobal_load_dword %1, %3, off\n"
                                         ".Lgq_skip_c%=:"
                                         : "+v"(rwn), "+v"(amn) : "v"(ps), "v"(ms), "s"(last) : "scc", "memory");
                        }
                    }
                } else {
                    // k-step b: the NEXT iteration's dword and absmax, then its four lookups
                    if constexpr (t == 40 && !(ABL & 2)) {
                        const int itn = it + 1, un_ = GQ_AHEAD + (itn >> 2);
                        const int so = __builtin_amdgcn_readfirstlane((un_ & 1) * GQ_ST_SLOT + 64 * (itn & 3));
                        const int sa = __builtin_amdgcn_readfirstlane((un_ & 1) * GQ_ST_SLOT + 8 * (itn & 3));
                        dw = *reinterpret_cast<const uint32_t *>(smem + st_rw + so);
                        dam = *reinterpret_cast<const float *>(smem + st_ra + sa);
                    }
                    if constexpr ((t == 60 || t == 64 || t == 68 || t == 72) && !(ABL & 2)) {
                        constexpr int b = (t - 60) / 4;
                        const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((dw >> (8 * b)) & 0xFFu) << 3));
                        Lr[b][0] = v[0]; Lr[b][1] = v[1];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IM = std::integral_constant<int, -1>;
    auto main_loop = [&](auto wo) {
        const DmaCtx dc = dma_ctx();
        kstep(I0{}, std::true_type{}, wo, IM{}, 0, 0, dc);
        int j = 1, it = 0;
        for (; j + 1 < nk; j += 2, it++) {
            kstep(I1{}, std::false_type{}, wo, I0{}, j, it, dc);
            kstep(I0{}, std::false_type{}, wo, I1{}, j + 1, it, dc);
        }
        if (j < nk) kstep(I1{}, std::false_type{}, wo, IM{}, j, 0, dc);
    };
    if (wave == 0) main_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) main_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) main_loop(std::integral_constant<int, 2>{});
    else main_loop(std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue (k_gemm_dense, 16-bit weights)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the column's flags go back to zero: the last workgroup of the column to get here clears them (all 16 have passed every
    // poll of theirs by now)
    if (threadIdx.x == 0) {
        const uint32_t prev = atomicAdd(col_sync + 64, 1u);
        if (prev == want - 1) {
            for (int u = 0; u < nslab; u++) __hip_atomic_store(col_sync + u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(col_sync + 64, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane_e = tid2 & 63, er16 = lane_e & 15, efq = lane_e >> 4;
    const int64_t n_base = n0 + wn * 128;
    if (out_dtype == MBNB_F32) {
        float *o = static_cast<float *>(out_v);
#pragma unroll
        for (int f = 0; f < 8; f++)
#pragma unroll
            for (int g = 0; g < FM; g++) {
                const int64_t m = m0 + wm * 16 * FM + 16 * g + er16, nn = n_base + 16 * f + 4 * efq;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][g][e]));
                    if (bias != nullptr && nn + e < N) sv += to_f32(bias[nn + e]);
                    v[e] = to_f32(from_f32<T>(sv));
                }
                if (m < M && nn < N) store4(o + m * N + nn, v, nn, N);
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
    constexpr int ROWB = 264;
    char *wave_lds = smem + wave * 64 * ROWB;
    uint16_t *out = static_cast<uint16_t *>(out_v);
    const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out_v) & 15) == 0);
    const bool same_out = out_dtype == (std::is_same_v<T, f16_t> ? MBNB_F16 : MBNB_BF16);
    u32x2 bias_all[8];
    if (bias != nullptr) {
        const uint16_t *bp = reinterpret_cast<const uint16_t *>(bias);
#pragma unroll
        for (int f = 0; f < 8; f++) {
            const int64_t n = n_base + 16 * f + 4 * efq;
            if (n + 4 <= N && (reinterpret_cast<uintptr_t>(bp + n) & 7) == 0) bias_all[f] = *reinterpret_cast<const u32x2 *>(bp + n);
            else {
                uint32_t t[4];
#pragma unroll
                for (int e = 0; e < 4; e++) t[e] = bp[n + e < N ? n + e : N - 1];
                bias_all[f] = u32x2{t[0] | (t[1] << 16), t[2] | (t[3] << 16)};
            }
        }
    }
    auto epilogue16 = [&](auto wb_t) {
        constexpr bool WB = decltype(wb_t)::value;
        gd_static_for<FM / 4>([&](auto hh) {
            constexpr int H = decltype(hh)::value;
            const int64_t m_base = m0 + wm * 16 * FM + 64 * H;
#pragma unroll
            for (int f = 0; f < 8; f++) {
                const int nl = 16 * f + 4 * efq;
                float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (WB) {
#pragma unroll
                    for (int e = 0; e < 4; e++) bv[e] = unpack_lo<T>(bias_all[f][e >> 1] >> (16 * (e & 1)));
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float sv;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][4 * H + g][e]));
                        v[e] = sv + bv[e];
                    }
                    if (!same_out) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = to_f32(from_f32<T>(v[e]));
                    }
                    u32x2 pk;
                    if (out_dtype == MBNB_F16) pk = u32x2{pack2<f16_t>(v[0], v[1]), pack2<f16_t>(v[2], v[3])};
                    else pk = u32x2{pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3])};
                    *reinterpret_cast<u32x2 *>(wave_lds + (16 * g + er16) * ROWB + nl * 2) = pk;
                }
            }
            const int ch = lane_e & 15;
            u32x4 piece[16];
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const char *srcp = wave_lds + (p * 4 + (lane_e >> 4)) * ROWB + ch * 16;
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(srcp), hi = *reinterpret_cast<const u32x2 *>(srcp + 8);
                piece[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            const int64_t n = n_base + ch * 8;
            if (n < N) {
                if (vec_ok && n + 8 <= N) {
#pragma unroll
                    for (int p = 0; p < 16; p++) {
                        const int64_t m = m_base + p * 4 + (lane_e >> 4);
                        if (m < M) store_out16_nt(reinterpret_cast<u32x4 *>(out + m * N + n), piece[p]);
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 16; p++) {
                        const int64_t m = m_base + p * 4 + (lane_e >> 4);
                        if (m >= M) continue;
#pragma unroll
                        for (int e = 0; e < 8; e++)
                            if (n + e < N) out[m * N + n + e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
    };
    if (bias != nullptr) epilogue16(std::true_type{});
    else epilogue16(std::false_type{});
}

}  // namespace mbnb
