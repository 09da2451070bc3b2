// gemm_beside.h — matmul_4bit for large M with the dequantise pass running BESIDE the GEMM (round 3; reference:
// functional.py:680-773 -- above M = 512 the reference dequantises once, then multiplies, :753-767).
//
// The two-launch form (dequantize_4bit into a scratch, then k_gemm_dense) pays a write-bound pass (10 us at 4096^2) and the
// boundary between the launches before the first MFMA.  k_gemm_dq (gemm_dq.h) let the GEMM's own waves do the pass: every
// load, write-through store and flag operation of it sits in the same in-order vmcnt queue as the wave's LDS-DMA pieces and
// costs the k-loop what it hides (profiles/r03_dq_ablation.txt).  Here the pass has its OWN waves -- and its own vmcnt queues --
// on the same SIMDs: k_gemm_dense takes 476 of a SIMD's 512 registers (480 allocated) and 128 of the CU's 160 KiB of LDS, so
// a SECOND kernel of <= 32 registers, one wave per SIMD, is resident next to it (tools/exp/ab_cores.py,
// profiles/r03_coresident_probe.txt: a 32-register helper streaming 8 -> 32 MB beside the GEMM stretches the pair to 102.5 us
// against 100.0 for the GEMM alone and 111.5 back to back; a 64-register helper does not fit and runs behind it).  Two
// launches on two streams (fork / join of events inside the call):
//   k_decode_beside  grid = N / 16 workgroups of 4 waves, <= 32 VGPRs: workgroup d decodes weight rows 16 d .. 16 d + 15, one SLAB
//                    of 512 k at a time (slab-major: the k-loop's order), wave w rows 16 d + 4 w + r (r = lane >> 4).  Lane
//                    (r, p): one 16-byte load of packed bytes + one absmax per slab, through a wave-private LDS area, because
//                    the stores want another distribution: dword 16 q + p of the row -> 8 values -> 16 bytes at
//                    Wd[row][512 u + 128 q + 8 p]: the 16 lanes of a row write whole 128-byte lines per WRITE-THROUGH store.
//                    Behind the slab's last store: vmcnt(0), barrier, ONE lane adds 1 to flag[column][slab] (agent scope).
//   k_gemm_gated     k_gemm_dense (same tile, pipeline, fragment schedule, epilogue; B = the scratch) whose wave 0 polls
//                    flag[tn][slab] (sc1 load, issued one loop iteration before it is looked at) until it reads the number of
//                    decoder workgroups of its tile column, before the k-step that issues the slab's first LDS-DMA piece; a
//                    workgroup barrier lies between that and the piece.
// Every line of Wd is written once per call and read only behind its flag, and the GEMM launch begins behind the previous
// call's last read of the scratch, so no cache holds a stale copy of it (MI355X_MICROARCH.md, inter-workgroup visibility:
// write-through payload, drained, flag by one lane behind a barrier, sc1 poll, barrier, loads).  The decoder never waits for
// the GEMM, so it always finishes; the GEMM's spins are bounded: on a timeout the workgroup sets sync[error] and goes on
// (wrong numbers, never a hang).  Flags are counters that start at ZERO (caller's contract) and are zeroed again by the last
// workgroup of each tile column.  Any M; no residency requirement (a GEMM workgroup waits for decoders only).
// Requirements (launcher): blocksize 64, plain or double-quantised absmax, K % 512 == 0, K <= 32768, 16-byte aligned operands.
// Output bits: those of dequantize_4bit + k_gemm_dense (same Wd bits, same pipeline).
#pragma once
#include "gemm_dense.h"

namespace mbnb {

constexpr int GB_SLAB = 8;                 // k-steps per slab
constexpr int GB_COL_WORDS = 66;           // per tile column: 64 slab flags, the done counter, one pad word
constexpr int GB_MAX_SLABS = 64;
constexpr int GB_ST_SLOT = 1152, GB_ST_WAVE = 2 * GB_ST_SLOT;      // decoder: wave-private exchange area, two slab parities
constexpr int GB_DEC_LDS = 4 * GB_ST_WAVE;
constexpr int64_t gb_sync_bytes(int64_t tiles_n) { return (tiles_n * GB_COL_WORDS + 2) * 4; }

// ------------------------------------------------------------------------------------------------------------------------
// The decoder.  amdgpu_num_vgpr(32): the kernel must fit the 32 registers k_gemm_dense leaves on a SIMD.
template <typename T, bool NESTED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) void k_decode_beside(
    const uint8_t *__restrict__ packed, AbsmaxView am, int qt, T *__restrict__ Wd, uint32_t *__restrict__ sync, int64_t N, int64_t K,
    int64_t K_weight, int u_begin, int u_end, uint32_t go_value) {        // slabs [u_begin, u_end) of the K / 512; go_value: see k_gemm_gated
    __shared__ __attribute__((aligned(2048))) float s_lut2[512];   // byte table: entry b = (code[b & 15], code[b >> 4])
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = lane >> 4, pp = lane & 15;
    if (tid == 0 && go_value != 0u) __hip_atomic_store(sync + ((N + 255) >> 8) * GB_COL_WORDS + 1, go_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int e = tid * 2 + h, b = e >> 1, nib = (e & 1) ? (b >> 4) : (b & 15);
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (nib == i) v = (qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
        s_lut2[e] = v;
    }
    // 32-bit offsets from the (uniform) kernel arguments: the decoder has 32 registers (launcher: N * K_weight * 2 < 2^31)
    int row = (int)blockIdx.x * 16 + 4 * wave + pr;
    row = row < (int)N ? row : (int)N - 1;                // rows past N: a duplicate of the last row (same bytes, same address)
    const int nblk = (int)(K_weight >> 6);
    const uint32_t src_off = (uint32_t)row * (uint32_t)(K_weight >> 1) + 16u * pp;       // + 256 slab                (bytes)
    const uint32_t blk_off = (uint32_t)row * (uint32_t)nblk + (uint32_t)(pp & 7);        // + 8 slab                  (blocks)
    const uint32_t dst_off = ((uint32_t)row * (uint32_t)K_weight + 8u * pp) * 2u;        // + 1024 slab + 256 q       (bytes)
    uint32_t *flag = sync + (((int64_t)blockIdx.x * 16) >> 8) * GB_COL_WORDS;
    const int st_w = wave * GB_ST_WAVE + pr * 256 + 16 * pp;                 // own 16 packed bytes      (+ slot)
    const int st_a = wave * GB_ST_WAVE + 1024 + pr * 32 + 4 * (pp & 7);      // own absmax               (+ slot)
    const int st_rw = wave * GB_ST_WAVE + pr * 256 + 4 * pp;                 // packed dword 16 q + p    (+ slot + 64 q)
    const int st_ra = wave * GB_ST_WAVE + 1024 + pr * 32 + 4 * (pp >> 3);    // absmax 2 q + (p >> 3)    (+ slot + 8 q)
    const char *lut2 = reinterpret_cast<const char *>(s_lut2);
    const int bs2_shift = NESTED ? __builtin_ctz((unsigned)am.bs2) : 0;      // launcher: bs2 is a power of two

    auto load_abs = [&](int u) -> float {
        const uint32_t bi = blk_off + 8u * (uint32_t)u;
        if constexpr (NESTED) return (float)(int)am.i8[bi] * (am.am2[bi >> bs2_shift] / 127.0f);   // dequantize_blockwise (functional.py:592-594)
        else return am.f32[bi];
    };
#ifdef GB_STAMPS
    uint64_t *stamp = reinterpret_cast<uint64_t *>(sync) + 2048 + 12 * blockIdx.x;
    if (tid == 0) stamp[0] = wall_clock64();
#endif
    u32x4 rw = *reinterpret_cast<const u32x4 *>(packed + (src_off + 256u * (uint32_t)u_begin));
    float ra = load_abs(u_begin);
    __syncthreads();                                      // byte table
    for (int u = u_begin; u < u_end; u++) {
        const int slot = (u & 1) * GB_ST_SLOT;
        // the slab's raw data into the wave's LDS area (the wave's LDS operations execute in order: its own reads below see it,
        // and its reads of this slot two slabs ago are done), the next slab requested
        *reinterpret_cast<u32x4 *>(smem + st_w + slot) = rw;
        *reinterpret_cast<float *>(smem + st_a + slot) = ra;
        const int un = u + 1 < u_end ? u + 1 : u;
        rw = *reinterpret_cast<const u32x4 *>(packed + (src_off + 256u * (uint32_t)un));
        ra = load_abs(un);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(smem + st_rw + slot + 64 * q);
            const float a = *reinterpret_cast<const float *>(smem + st_ra + slot + 8 * q);
            u32x4 o;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((w >> (8 * b)) & 0xFFu) << 3));
                float p0, p1;
                asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(v[0]), "v"(a));       // two IEEE products (no contraction, no packing)
                asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(v[1]), "v"(a));
                o[b] = pack2<T>(p0, p1);
            }
            const uint32_t doff = dst_off + 1024u * (uint32_t)u + 256u * (uint32_t)q;
            // s_nop 1: a store of more than 8 bytes reads its data registers for a few cycles after issue (gemm_dq.h)
            asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(doff), "v"(o), "s"(Wd) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left
        __syncthreads();
        if (tid == 0) asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(flag + u), "v"(1u) : "memory");
#ifdef GB_STAMPS
        if (tid == 0 && u < 8) stamp[1 + u] = wall_clock64();
#endif
    }
#ifdef GB_STAMPS
    if (tid == 0) stamp[9] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// The GEMM: k_gemm_dense<T, false, 8> with the slab gates.
// ABL (diagnostic builds under tools/exp; the product instantiates 0): 1 no wait for slab 0, 2 no spin at the in-loop gates (the
// poll is still issued), 4 no in-loop polls at all.  GB_STAMPS (tools/exp): s_memrealtime stamps into the sync area from byte 16384.
template <typename T, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_gemm_gated(const T *__restrict__ X, const T *__restrict__ Wd, uint32_t *__restrict__ sync,
                                                       const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype, int64_t M,
                                                       int64_t N, int64_t K, int64_t ldw, uint32_t go_value) {
    using Frag = typename Mfma16<T>::frag;
    using Plan = GdPlan<8>;
    constexpr int FM = 8, TM = 256, PM = 4, PN = 8;
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
    const int64_t full_n = (tiles_m % PM == 0) ? (tiles_n / PN) * PN : 0;
    if (bid < tiles_m * full_n) {
        const int64_t patch = bid >> 5, within = bid & 31;
        const int64_t patches_m = tiles_m / PM;
        tm = (patch % patches_m) * PM + (within % PM);
        tn = (patch / patches_m) * PN + (within / PM);
    } else {
        const int64_t r = bid - tiles_m * full_n;
        tm = r % tiles_m;
        tn = full_n + r / tiles_m;
    }
    const int64_t m0 = tm * TM, n0 = tn << 8;
    const int nk = (int)(K >> 6);
    const int nslab = nk / GB_SLAB;
    // go_value != 0: the decoder sits on another stream behind hipStreamWaitValue32(go == go_value).  This launch is in the caller's
    // stream, so its first instruction runs behind everything the decoder must wait for (the previous user of the scratch, whatever
    // produced the packed weight): every workgroup publishes the word, the command processor starts the decoder ~1.2 us later
    // (profiles/r03_waitvalue_probe.txt; an event between the two queues costs 7-10 us).
    if (tid == 0 && go_value != 0u) __hip_atomic_store(sync + tiles_n * GB_COL_WORDS + 1, go_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef GB_STAMPS
    uint64_t *stamp = reinterpret_cast<uint64_t *>(sync) + 2048 + 4096 + 4 * blockIdx.x;
    if (tid == 0) stamp[0] = wall_clock64();
#endif

    // ---- LDS-DMA of both operands (k_gemm_dense); B comes from the scratch the decoder fills meanwhile
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

    uint32_t *col_sync = sync + tn * GB_COL_WORDS;
    uint32_t *err_word = sync + tiles_n * GB_COL_WORDS;
    const int64_t rows_col = N - n0 < 256 ? N - n0 : 256;
    const uint32_t want = (uint32_t)__builtin_amdgcn_readfirstlane((int)((rows_col + 15) >> 4));       // decoder workgroups of this tile column
    // one relaxed agent-scope poll of a flag word (one lane's value, broadcast)
    auto poll_once = [&](const uint32_t *p) {
        uint32_t v;
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };

    // ---- prologue: activations of tiles 0 and 1 on their way; slab 0 of the column awaited; then the weight pieces
    {
        const DmaCtx c0 = dma_ctx();
        gd_static_for<FM>([&](auto q) { issue_piece(q, 0, 0, c0); });
        gd_static_for<FM>([&](auto q) { issue_piece(q, 1, (nk > 1 ? 1 : 0) << 7, c0); });
        if (wave == 0 && !(ABL & 1)) {
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
#ifdef GB_STAMPS
        if (tid == 0) stamp[1] = wall_clock64();
#endif
        const DmaCtx c1 = dma_ctx();
        gd_static_for<8>([&](auto q) { issue_piece(std::integral_constant<int, FM + decltype(q)::value>{}, 0, 0, c1); });
        gd_static_for<8>([&](auto q) { issue_piece(std::integral_constant<int, FM + decltype(q)::value>{}, 1, (nk > 1 ? 1 : 0) << 7, c1); });
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
    // MFMAs from assembly, accumulators pinned to their AGPR tuples (gemm_fused4.h)
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

    uint32_t pollv = 0u;
    const uint32_t *flag_base = col_sync;

    // ---- one k-step (k_gemm_dense's slots) + the gate behind barrier 2 of k-step a.  HALF 0 = first k-step of a loop
    // iteration, 1 = second, -1 = the peeled first / last k-steps.  it = iteration number; a slab = 4 iterations.
    auto kstep = [&](auto cc, auto first, auto wo_, auto half_, int j, int it, const DmaCtx &dc) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1, WO = decltype(wo_)::value, HALF = decltype(half_)::value;
        constexpr bool FIRST = decltype(first)::value;
        const int kb2 = __builtin_amdgcn_readfirstlane(kbytes(j + 2));
        const int q = it & 3;
        const int u_poll = (it >> 2) + 1;              // slab whose first LDS-DMA comes up in iteration 4 u_poll - 2
        const int c_poll = (!(ABL & 4) && q == 0 && u_poll < nslab && wave == 0) ? 1 : 0;
        const int c_chk = (!(ABL & 6) && q == 1 && u_poll < nslab && wave == 0) ? 1 : 0;
        const int n_side = __builtin_amdgcn_readfirstlane(c_poll);
        gd_static_for<Plan::NS>([&](auto tt) {
            constexpr int t = decltype(tt)::value, ks = t / 64, f = (t % 64) / FM, g = t % FM;
            if constexpr (t == Plan::B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (t == Plan::B2) {
                // k-step b: the poll issued behind barrier 2 of k-step a is YOUNGER than the pieces this wait is for and may stay
                // in flight (a system-coherent load takes longer than a piece): the count that leaves exactly the 16 pieces of
                // this k-step + that poll is chosen by a scalar branch inside the asm statement (a branch the compiler sees splits
                // the k-step into basic blocks and wrecks the register allocation, gemm_fused4.h).  k-step a of the next iteration
                // (vmcnt(16)) waits for it.
                if constexpr (HALF == 1) {
                    asm volatile("s_cmp_lt_u32 %0, 1\n\ts_cbranch_scc1 .Lgb_w0%=\n\ts_waitcnt vmcnt(17)\n\ts_branch .Lgb_we%=\n"
                                 ".Lgb_w0%=:\n\ts_waitcnt vmcnt(16)\n"
                                 ".Lgb_we%=:" ::"s"(n_side) : "scc", "memory");
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
            if constexpr (HALF == 0 && t == Plan::B2 + 17) {
                // q == 0: ask for the flag of the slab whose first LDS-DMA comes up in iteration 4 u_poll - 2 (looked at one
                // iteration later: it is older than the next k-steps' pieces, so their vmcnt covers it)
                const int pl = __builtin_amdgcn_readfirstlane(c_poll);
                const uint32_t *pq = flag_base + (u_poll < nslab ? u_poll : 0);
                asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 .Lgb_skip_e%=\n\tglobal_load_dword %0, %1, off sc1\n.Lgb_skip_e%=:"
                             : "+v"(pollv) : "v"(pq), "s"(pl) : "scc", "memory");
                // q == 1: that poll has landed; not every decoder of the column there yet -> poll in place (bounded)
                const int ck = __builtin_amdgcn_readfirstlane(c_chk);
                uint32_t cnt, cur;
                asm volatile("s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 .Lgb_done%=\n\t"
                             "s_mov_b32 %1, 0\n"
                             ".Lgb_again%=:\n\t"
                             "v_readfirstlane_b32 %2, %0\n\t"
                             "s_cmp_ge_u32 %2, %5\n\ts_cbranch_scc1 .Lgb_done%=\n\t"
                             "s_add_u32 %1, %1, 1\n\ts_cmp_gt_u32 %1, 0x400000\n\ts_cbranch_scc1 .Lgb_fail%=\n\t"
                             "s_sleep 2\n\tglobal_load_dword %0, %3, off sc1\n\ts_waitcnt vmcnt(0)\n\ts_branch .Lgb_again%=\n"
                             ".Lgb_fail%=:\n\tglobal_atomic_or %4, %7, off sc1\n"
                             ".Lgb_done%=:"
                             : "+v"(pollv), "=&s"(cnt), "=&s"(cur) : "v"(pq), "v"(err_word), "s"(want), "s"(ck), "v"(2u) : "scc", "memory");
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
#ifdef GB_STAMPS
    if (tid == 0) stamp[2] = wall_clock64();
#endif

    // ---- epilogue (k_gemm_dense, 16-bit weights)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the column's flags go back to zero: the last of the column's tiles_m workgroups to get here clears them (all of them have
    // passed every poll of theirs by now; the decoder's last add to them is older than the last poll that succeeded)
    if (threadIdx.x == 0) {
        const uint32_t prev = atomicAdd(col_sync + 64, 1u);
        if (prev == (uint32_t)tiles_m - 1) {
            for (int u = 0; u < nslab; u++) __hip_atomic_store(col_sync + u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the epoch word too (WaitValue arrangement): the decoder was launched long ago; the area is all zero again after the call
            __hip_atomic_store(sync + tiles_n * GB_COL_WORDS + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
#ifdef GB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) stamp[3] = wall_clock64();
#endif
}

}  // namespace mbnb
