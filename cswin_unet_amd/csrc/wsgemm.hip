// Weight-stationary fp32 MFMA GEMM for the tall-skinny Linears of CSWin-UNet (gfx950 / CDNA4).
//
//   C[m][n] = epilogue( sum_r A[m][r] * Bm(r, n) ),   M = B*L tokens (1e3 .. 1e5),  R and N = 64 .. 1024
//     forward        (networks/cswin_unet.py:169,177,23-27):  Bm(r, n) = W[n][r]   (W = nn.Linear weight [N][R])
//     data gradient  (their autograd backward):               Bm(r, n) = W[r][n]   (W = nn.Linear weight [R][N])
//
// Every Linear of the model multiplies a long token matrix by a SMALL weight (<= 1 M floats).  The tiled family in gemm.hip
// re-stages a 64 x 32 slice of the weight together with every 64 x 32 slice of the activations: 16 B of LDS staging per
// cycle and CU at MFMA peak, a prologue and an epilogue per 64 x 64 tile, and 300 .. 1200 short-lived workgroups per launch.
// Measured (profiles/round1_notes.md): with the matrix cores made 16x faster the step only gained 17 % -- those kernels
// are bound by staging, barriers and launch ramps, not by MFMA.  Here the operand roles follow the shape instead:
//
//   * one PERSISTENT workgroup per CU (grid <= 256): wave (wn, wk) keeps Bm[wk*KS .. +KS) x [32 wn .. +32) of the weight in
//     REGISTERS for the whole launch (KS/2 VGPRs: lane (j, h) holds the B operand of every v_mfma_f32_32x32x2_f32 of its
//     column block), loaded once from L2;
//   * the workgroup walks its contiguous range of token rows; activations are the only thing that moves: 64 rows x 32 k per
//     wave-group and step, global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no ds_write) into a
//     3-4 deep ring, XOR-swizzled on the SOURCE address so that the ds_read_b128 A fragments are bank-conflict free;
//   * one raw s_barrier per step with a counted s_waitcnt vmcnt (the DMA of the next stages stays in flight across it);
//   * k-groups (wk) sum their accumulators through LDS once per 64-row tile; the epilogue is gemm_epilogue.h's.
// Activation staging drops to 2-4 B per cycle and CU, the weight is read once per CU, and the MFMA stream of a wave is
// 32 back-to-back instructions per step between two barriers.
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <utility>

#include "gemm_epilogue.h"

namespace {

constexpr int WS_BM = 64;            // rows per tile (two 32-row MFMA blocks)
constexpr int WS_BK = 32;            // k per step and k-group
constexpr int WS_STAGE_FLOATS = WS_BM * WS_BK;   // one k-group's stage: 8 KB

struct WsParams {
    const float* A; long lda;        // [M][R]
    const float* W; long ldw;        // forward: [N][R]; data gradient: [R][N]
    int M, N, R;
    int n_groups, slots;             // grid = n_groups * slots
    long long* stamps;               // debug (cswin_debug_set_ws_stamps): [workgroup][16] s_memtime stamps of wave 0, or NULL
    Epilogue epi;
};

#define WS_STAMP(k)                                                                                             \
    do {                                                                                                        \
        if (p.stamps && threadIdx.x == 0 && (k) < 16) p.stamps[(long)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to lds_dst + 16 * lane (wave-uniform lds_dst).  hipcc does not
// count this load: completion is tracked by hand with ws_wait_vmcnt (guide 5.7).
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int... Is, class F>
__device__ __forceinline__ void ws_static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void ws_static_for(F&& f) { ws_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// First tile of a workgroup: operations YOUNGER than the DMA of step kc + 1 when step kc's MFMAs have been issued (the counted
// wait may leave that many in flight).  Issue order: prologue = DMA(0 .. D-2), weight chunks 0 and 1, bias; step j = DMA(j + D - 1),
// weight chunk j + 2 (chunks exist for indices < nkc; q DMA and nwc weight load instructions each).  vmcnt is a 6-bit field.
constexpr int ws_first_tile_inflight(int kc, int nkc, int d, int q, int nwc) {
    int n = 0;
    if (kc + 1 <= d - 2) {
        n = (d - 2 - (kc + 1)) * q + (nkc >= 2 ? 2 : 1) * nwc + 1;
        for (int j = 0; j <= kc; ++j) n += q + (j + 2 < nkc ? nwc : 0);
    } else {
        const int j0 = kc + 2 - d;
        n = (j0 + 2 < nkc ? nwc : 0);
        for (int j = j0 + 1; j <= kc; ++j) n += q + (j + 2 < nkc ? nwc : 0);
    }
    return n > 63 ? 63 : n;
}

// ring depth: 8-wave workgroups own the CU's LDS (<= 133 KB); two 4-wave workgroups per CU get <= 80 KB each
constexpr int ws_ring_depth(int nwn, int nwk) { return nwn * nwk == 4 ? (nwk >= 2 ? 3 : 4) : (nwk >= 4 ? 3 : 4); }

template <int N>
__device__ __forceinline__ void ws_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void ws_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// Epilogue operands of one 64 x 32 wave tile in the store layout (lane: row rrow + 8 p of block i, columns rcol .. rcol + 3).
// They are loaded at the top of the tile's LAST step, retired together with the LDS-DMA by one explicit vmcnt(0) after that
// step's MFMAs, and only then consumed: inside the tile loop hipcc never sees a load whose result is still pending, so its
// s_waitcnt pass inserts no vmcnt(0) of its own (it would drain the stores of the tile and every DMA in flight).
template <int EPI>
struct WsEpiRegs {
    static constexpr bool HAS_AUX = EPI == EPI_RES || EPI == EPI_GELUBWD;
    f32x4 aux[HAS_AUX ? 2 : 1][HAS_AUX ? 4 : 1];
    float rs[2][4];
};

template <int EPI>
__device__ __forceinline__ void ws_epi_prefetch(WsEpiRegs<EPI>& r, const Epilogue& e, int M, int N, int mb, int nb, int lane, bool full) {
    const int rrow = lane >> 3, n = nb + (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = mb + i * 32 + rrow + 8 * p;
            const bool ok = m < M && n < N && (i == 0 || full);
            r.rs[i][p] = (e.row_scale && ok) ? e.row_scale[m / e.rows_per_sample] : 1.0f;
            if constexpr (WsEpiRegs<EPI>::HAS_AUX) {
                const float* ap = EPI == EPI_RES ? e.residual : e.gelu_pre;
                const long ld = EPI == EPI_RES ? e.ldres : e.ldpre;
                r.aux[i][p] = ok ? *reinterpret_cast<const f32x4*>(ap + (long)m * ld + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
}

// acc (native MFMA C/D layout) -> wave-private LDS patch -> 16-B stores; the arithmetic of gemm_epilogue.h's vector path
template <int EPI>
__device__ __forceinline__ void ws_epi_store(const WsEpiRegs<EPI>& r, const Epilogue& e, f32x16 (&acc)[2][1], f32x4 bias_v, int M, int N,
                                             int mb, int nb, int lane, float* wbuf, bool full) {
    const int li = lane & 31, lh = lane >> 5;
    const int rrow = lane >> 3, rcol = (lane & 7) * 4;
    const int n = nb + rcol;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !full) break;
#pragma unroll
        for (int g = 0; g < 16; ++g) wbuf[((g & 3) + 8 * (g >> 2) + 4 * lh) * EP_LD + li] = acc[i][0][g];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = mb + i * 32 + rrow + 8 * p;
            f32x4 o = *reinterpret_cast<const f32x4*>(&wbuf[(rrow + 8 * p) * EP_LD + rcol]) + bias_v;
            if (EPI == EPI_GELUBWD) {
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] *= gelu_grad_f(r.aux[i][p][c]);
            }
            o *= r.rs[i][p];
            if (EPI == EPI_RES) o += r.aux[i][p];
            if (m < M && n < N) {
                *reinterpret_cast<f32x4*>(e.C + (long)m * e.ldc + n) = o;
                if (EPI == EPI_ACT) {
                    f32x4 a;
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] = gelu_f(o[c]);
                    *reinterpret_cast<f32x4*>(e.Cact + (long)m * e.ldact + n) = a;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int KS, int NWN, int NWK, bool BTRANS, int EPI>
__global__ __launch_bounds__(64 * NWN * NWK, 2) void ws_gemm_kernel(WsParams p) {      // 8 waves per CU: one workgroup of 8 or two of 4
    constexpr int NWAVES = NWN * NWK;
    constexpr int NKC = KS / WS_BK;                       // steps per tile
    constexpr int D = ws_ring_depth(NWN, NWK);            // ring depth
    constexpr int Q = 8 / NWN;                            // LDS-DMA instructions per wave and stage (8 per k-group)
    constexpr int STAGE = NWK * WS_STAGE_FLOATS;
    constexpr int PATCH = 2 * 16 * 64;                    // one wave's two accumulator blocks in native layout
    constexpr int RED = NWK > 1 ? (NWK / 2) * NWN * PATCH : 0;
    constexpr int EPIW = NWAVES * EP_WAVE_FLOATS;
    constexpr int SCRATCH = RED > EPIW ? RED : EPIW;
    static_assert(KS % WS_BK == 0 && 8 % NWN == 0 && NWAVES % 4 == 0, "unsupported wave layout");
    extern __shared__ __attribute__((aligned(1024))) float lds[];   // [D][NWK][64][32] ring | scratch
    float* scratch = lds + D * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NWN, wk = wave / NWN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware order: blocks b, b + 8, ... share an XCD (private L2).  Consecutive logical ids = the n-groups of one row
    // range, so the activations of a row range are fetched into one L2 only.
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int lb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int slot = lb / p.n_groups, ng = lb - slot * p.n_groups;
    const int units = (p.M + 31) >> 5;
    const int u0 = (int)((long)slot * units / p.slots), u1 = (int)((long)(slot + 1) * units / p.slots);
    const int n_units = u1 - u0;
    const int ntiles = (n_units + 1) >> 1;
    const int total = ntiles * NKC;
    const int row_base = u0 * 32;
    const int row_end = min(p.M, u1 * 32);               // rows this workgroup owns
    const int ncol0 = (ng * NWN + wn) * 32;               // this wave's output columns
    const bool active = ncol0 < p.N;                       // wave-uniform

    WS_STAMP(0);
    // ---- LDS-DMA geometry: instruction q of this wave covers k-group gq, rows 8*rb .. +8 of the tile (8 lanes x 16 B per row) ----
    // lane -> (row = 8 rb + lane/8, physical chunk pc = lane & 7) holds logical chunk pc ^ ((row >> 1) & 7) of that row
    int q_row[Q];
    int q_off[Q];
    unsigned q_lds[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = wave * Q + q;                       // 0 .. 8 NWK - 1
        const int gq = i >> 3, rb = i & 7;
        const int row = 8 * rb + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        q_row[q] = row;
        q_off[q] = gq * KS + 4 * c;
        q_lds[q] = (unsigned)(i * 1024);                  // byte offset of the 1 KB piece inside a stage
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;   // LDS byte address of the ring

    auto issue = [&](int s) {                              // LDS-DMA of step s into stage s % D
        const int tile = s / NKC, kc = s - tile * NKC;
        const unsigned dst = lds_base + (unsigned)((s % D) * STAGE * 4);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            int m = row_base + tile * WS_BM + q_row[q];
            m = m < p.M ? m : p.M - 1;                     // rows past the end re-read the last row (masked in the epilogue)
            glds16(p.A + (long)m * p.lda + q_off[q] + kc * WS_BK, __builtin_amdgcn_readfirstlane(dst + q_lds[q]));
        }
    };

    // read-side address of this lane inside a stage: row-half hm, k-offset kk (0, 8, 16, 24)
    const int sw = (li >> 1) & 7;                          // (row >> 1) & 7 with row = 32 hm + li: hm adds 16 -> same low bits
    auto a_frag = [&](const float* st, int hm, int kk) -> f32x4 {
        const int pc = ((kk >> 2) + lh) ^ sw;
        return *reinterpret_cast<const f32x4*>(st + wk * WS_STAGE_FLOATS + (32 * hm + li) * WS_BK + 4 * pc);
    };

    // ---- prologue.  Issue order = vmcnt order: [DMA of steps 0 .. D-2] [NWL weight loads] [1 bias load].  The activations of
    // the first steps are therefore NOT queued behind the 32 KB of weights of this wave, and the first tile (peeled below)
    // starts its MFMAs as soon as step 0 and the first sixteen weight registers have landed: hipcc waits for its own loads
    // register by register, and every load between the DMA and the first counted wait is unconditional, so their number
    // is a constant the hand-written waits can add to their counts.
    if (total > 0) {
#pragma unroll
        for (int s = 0; s < D - 1; ++s)
            if (s < total) issue(s);
    }
    // every wave's DMA is in the CU's memory queue before any wave's weight loads (the texture path serves requests in arrival
    // order: without this barrier the DMA of the last waves sit behind the 32-KB weight streams of the first)
    ws_barrier();
    // The weight slice of this wave lives in registers for the whole launch: breg[4 * (koff / 8) + s] = Bm(wk*KS + koff + 4 lh + s,
    // ncol0 + li).  It is STREAMED in: the 32-k chunk of step c is requested two steps ahead (chunks 0 and 1 here, chunk c + 2
    // at the top of step c of the first tile), so the launch does not open with every CU pulling its whole 32 - 256 KB slice
    // through L2 at once (measured: that burst delayed the first activations of every workgroup to ~15 k cycles).
    // Columns past N (last n-group) re-read column N - 1: their products are never stored.  No select on the loaded values:
    // it would make hipcc wait for the load on the spot instead of at the MFMA that first uses it.
    constexpr int NWC = BTRANS ? 16 : 4;                   // weight load instructions per lane and chunk
    float breg[KS / 2];
    const int wn_col = min(ncol0 + li, p.N - 1);
    const float* wbase = BTRANS ? p.W + (long)(wk * KS + 4 * lh) * p.ldw + wn_col : p.W + (long)wn_col * p.ldw + wk * KS + 4 * lh;
    auto load_w_chunk = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c < NKC) {
#pragma unroll
            for (int g = 4 * c; g < 4 * c + 4; ++g) {
                if constexpr (!BTRANS) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) breg[4 * g + e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) breg[4 * g + e] = wbase[(long)(8 * g + e) * p.ldw];
                }
            }
        }
    };
    load_w_chunk(std::integral_constant<int, 0>{});
    load_w_chunk(std::integral_constant<int, 1>{});
    // bias of this wave's columns in the epilogue's lane layout (lane owns columns ncol0 + 4 (lane & 7) .. +3): loaded once,
    // unconditionally (from the weight matrix when there is no bias: the value is then discarded)
    f32x4 bias_raw;
    bool has_bias;
    {
        const int n = ncol0 + (lane & 7) * 4;
        has_bias = p.epi.bias && n < p.N;
        bias_raw = *reinterpret_cast<const f32x4*>(has_bias ? p.epi.bias + n : p.W);     // selected where it is used (epilogue)
    }
    constexpr int NW_PRO = (NKC >= 2 ? 2 : 1) * NWC + 1;   // weight chunks 0, 1 and the bias load
    constexpr int N_PRO = (D - 2) * Q + NW_PRO > 63 ? 63 : (D - 2) * Q + NW_PRO;       // vmcnt is a 6-bit field
    if (total > D - 2) ws_wait_vmcnt<N_PRO>();             // my DMA of step 0 has landed (everything younger may be in flight)
    else ws_wait_vmcnt<0>();
    WS_STAMP(1);
    ws_barrier();                                          // step 0 has landed for every wave
    WS_STAMP(2);

    // Step protocol (one barrier per step): issue the DMA of step s + D - 1 into the stage that step s - 1 just released ->
    // MFMAs of step s -> wait for MY DMA of step s + 1 (counted: the later stages stay in flight) -> [tile end: epilogue]
    // -> barrier.  The wait sits BEFORE the epilogue's stores: vmcnt retires in issue order, so a counted wait behind
    // freshly issued stores would wait for them too.  FIRST (the peeled first tile): the weight loads are still in the queue
    // behind the prologue's DMA, so waits for steps 1 .. D-2 allow N_PRO operations in flight.
    int s = 0;
    auto run_tile = [&](auto first_tag, int tile) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const bool full = 2 * tile + 1 < n_units;          // second 32-row block belongs to this workgroup
        f32x16 acc[2][1];
        WsEpiRegs<EPI> er;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] = acc[1][0][e] = 0.f;
        ws_static_for<NKC>([&](auto kct) {
            constexpr int kc = decltype(kct)::value;
            {
                int si = s + D - 1;                        // laundered: keeps hipcc from specialising (and hoisting) the DMA
                asm volatile("" : "+s"(si));               // addresses of every unrolled step
                if (si < total) issue(si);
            }
            if constexpr (FIRST) load_w_chunk(std::integral_constant<int, kc + 2>{});      // two steps ahead of its MFMAs
            if (kc == NKC - 1 && wk == 0 && active)
                ws_epi_prefetch<EPI>(er, p.epi, row_end, p.N, row_base + tile * WS_BM, ncol0, lane, full);
            if (active) {
                const float* st = lds + (s % D) * STAGE;
#pragma unroll
                for (int hm = 0; hm < 2; ++hm) {
                    if (hm == 1 && !full) break;
                    f32x4 af[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) af[kk] = a_frag(st, hm, 8 * kk);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[hm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][e], breg[kc * 16 + kk * 4 + e], acc[hm][0], 0, 0, 0);
                }
            }
            // my DMA of step s + 1 has landed once at most the (D - 2) later stages (first tile: plus the weight chunks requested
            // behind it) are outstanding.  The tile's last step retires everything instead (the epilogue operands just loaded
            // sit behind the DMA in the in-order counter).
            if constexpr (kc == NKC - 1) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            } else {
                if (s + D - 1 >= total) ws_wait_vmcnt<0>();
                else if constexpr (FIRST) ws_wait_vmcnt<ws_first_tile_inflight(kc, NKC, D, Q, NWC)>();
                else ws_wait_vmcnt<(D - 2) * Q>();
                ws_barrier();                              // (the tile's last barrier comes after the epilogue)
            }
            ++s;
        });
        WS_STAMP(3 + 2 * tile);
        // ---- k-groups -> one accumulator (binary tree through LDS, native C/D layout) ----
        if constexpr (NWK > 1) {
#pragma unroll
            for (int half = NWK / 2; half >= 1; half >>= 1) {
                if (wk >= half && wk < 2 * half) {
                    float* dst = scratch + ((wk - half) * NWN + wn) * PATCH + lane;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 16; ++g) dst[(i * 16 + g) * 64] = acc[i][0][g];
                }
                ws_barrier();
                if (wk < half) {
                    const float* src = scratch + (wk * NWN + wn) * PATCH + lane;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 16; ++g) acc[i][0][g] += src[(i * 16 + g) * 64];
                }
                ws_barrier();
            }
        }
        if (wk == 0 && active)
            ws_epi_store<EPI>(er, p.epi, acc, has_bias ? bias_raw : f32x4{0.f, 0.f, 0.f, 0.f}, row_end, p.N, row_base + tile * WS_BM, ncol0, lane, scratch + wave * EP_WAVE_FLOATS, full);
        WS_STAMP(4 + 2 * tile);
        ws_barrier();
    };
    if (ntiles > 0) run_tile(std::true_type{}, 0);
    for (int tile = 1; tile < ntiles; ++tile) run_tile(std::false_type{}, tile);
    WS_STAMP(15);
}

template <int KS, int NWN, int NWK>
constexpr size_t ws_lds_bytes() {
    constexpr int D = ws_ring_depth(NWN, NWK);
    constexpr int RED = NWK > 1 ? (NWK / 2) * NWN * 2 * 16 * 64 : 0;
    constexpr int EPIW = NWN * NWK * EP_WAVE_FLOATS;
    return (size_t)(D * NWK * WS_STAGE_FLOATS + (RED > EPIW ? RED : EPIW)) * sizeof(float);
}

template <int KS, int NWN, int NWK, bool BTRANS, int EPI>
int ws_launch(const WsParams& p, hipStream_t st) {
    constexpr size_t lds = ws_lds_bytes<KS, NWN, NWK>();
    auto kern = ws_gemm_kernel<KS, NWN, NWK, BTRANS, EPI>;
    static std::once_flag once;         // once per process and instantiation, thread-safe (a function attribute, not a stream operation)
    static hipError_t status = hipSuccess;
    std::call_once(once, [&] { status = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    if (status != hipSuccess) { cswin_set_error("ws_gemm: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(status)); return CSWIN_ERR_HIP; }
    hipLaunchKernelGGL(kern, dim3(p.n_groups * p.slots), dim3(64 * NWN * NWK), lds, st, p);
    return CSWIN_OK;
}

struct WsCfg { int ks, nwn, nwk; double cost; };

// Cost model (cycles; constants from in-kernel stamps, tools/ws_stamps.py): the launch ends with the most loaded workgroup.
// A workgroup of n units (32 rows) runs n * KS/2 MFMAs of 64 cycles per wave, two waves per SIMD; every 64-row tile pays one
// epilogue (~3000 cycles, not overlapped: the waves of a workgroup move in lock-step); the weight load and the first DMA
// cost ~6000 cycles.  Layouts whose register allocation spills (hipcc -Rpass-analysis=kernel-resource-usage) are left out.
bool ws_choose(int M, int N, int R, bool aux, WsCfg* out) {
    static const int ks_list[] = {64, 128, 192, 256};
    static const int lay[][2] = {{8, 1}, {4, 2}, {2, 4}, {4, 1}, {2, 2}};
    const int units = (M + 31) / 32;
    bool found = false;
    WsCfg best = {0, 0, 0, 1e300};
    for (int ks : ks_list)
        for (auto& l : lay) {
            const int nwn = l[0], nwk = l[1];
            if (ks * nwk != R) continue;
            if ((nwk == 4 && ks == 256) || (nwn == 2 && nwk == 2 && ks == 256)) continue;                  // spill
            if (aux && ((nwk == 4 && ks >= 192) || (nwk == 2 && ks == 256))) continue;                      // spill
            const int ng = (N + 32 * nwn - 1) / (32 * nwn);
            const int wgs = nwn * nwk == 4 ? 512 : 256;           // two 4-wave workgroups per CU, or one of 8
            if (ng > wgs) continue;
            int slots = wgs / ng;
            if (slots > units) slots = units;
            if (slots < 1) continue;
            const int upw = (units + slots - 1) / slots;
            const double cost = (double)upw * (ks / 2) * 128.0 + ((upw + 1) / 2) * (3000.0 + (nwk > 1 ? 800.0 * nwk : 0.0)) + 6000.0;
            if (cost < best.cost) { best = {ks, nwn, nwk, cost}; found = true; }
        }
    if (found) *out = best;
    return found;
}

template <bool BTRANS, int EPI>
int ws_dispatch(const WsCfg& c, WsParams& p, hipStream_t st) {
    const int ng = (p.N + 32 * c.nwn - 1) / (32 * c.nwn);
    const int units = (p.M + 31) / 32;
    int slots = (c.nwn * c.nwk == 4 ? 512 : 256) / ng;
    if (slots > units) slots = units;
    p.n_groups = ng;
    p.slots = slots;
#define WS_CASE(KS_, NWN_, NWK_) \
    if (c.ks == KS_ && c.nwn == NWN_ && c.nwk == NWK_) return ws_launch<KS_, NWN_, NWK_, BTRANS, EPI>(p, st);
    WS_CASE(64, 8, 1) WS_CASE(128, 8, 1) WS_CASE(192, 8, 1) WS_CASE(256, 8, 1)
    WS_CASE(64, 4, 2) WS_CASE(128, 4, 2) WS_CASE(192, 4, 2) WS_CASE(256, 4, 2)
    WS_CASE(64, 2, 4) WS_CASE(128, 2, 4) WS_CASE(192, 2, 4) WS_CASE(256, 2, 4)
    WS_CASE(64, 4, 1) WS_CASE(128, 4, 1) WS_CASE(192, 4, 1) WS_CASE(256, 4, 1)
    WS_CASE(64, 2, 2) WS_CASE(128, 2, 2) WS_CASE(192, 2, 2) WS_CASE(256, 2, 2)
#undef WS_CASE
    return 1;
}

}  // namespace

// Entry used by gemm.hip's cswin_linear_fwd / cswin_linear_bwd_data (internal: not part of the C ABI).
// Returns 1 when the shape is not served here (the caller falls back to the tiled family), 0 on launch, < 0 on error.
// mode: 0 = forward (W [N][R]), 1 = data gradient (W [R][N]).  epi_mode: EPI_PLAIN / EPI_ACT / EPI_RES / EPI_GELUBWD.
static long long* g_ws_stamps = nullptr;
extern "C" void cswin_debug_set_ws_stamps(void* p) { g_ws_stamps = (long long*)p; }
static int g_ws_override = -1;      // debug hook only (cswin_debug_set_ws_gemm, tools/ws_gemm_check.py): A/B the two families in one process
extern "C" void cswin_debug_set_ws_gemm(int on) { g_ws_override = on; }

int cswin_ws_gemm(int mode, int epi_mode, const float* A, const float* W, const void* epilogue, int M, int N, int R, void* stream) {
    // CSWIN_WS_GEMM / cswin_debug_set_ws_gemm: 0 (default) = never, 1 = wherever a layout exists (A/B runs), 2 = where the cost
    // model predicts this family within 1.7x of the pure-MFMA time and the output is at least 128 columns wide -- the shapes on
    // which it measured faster than the tiled family stand-alone (profiles/round2_ws_gemm_vs_tiled.txt: -0.19 ms per step
    // picking the better of the two per shape).  Inside the training step that gain does not materialise (13.25 / 13.28 /
    // 13.46 ms per step for modes 0 / 2 / 1), so the default path stays the tiled family.
    const int ws_mode = g_ws_override >= 0 ? g_ws_override : cswin_tuning().ws_gemm;
    if (!ws_mode) return 1;
    if (M < 512 || N % 4 != 0 || R % 4 != 0 || !aligned16(A) || !aligned16(W)) return 1;
    WsCfg c;
    if (!ws_choose(M, N, R, epi_mode == EPI_RES || epi_mode == EPI_GELUBWD, &c)) return 1;
    if (ws_mode == 2) {
        const double ideal = 2.0 * M * (double)N * R / (256.0 * 256.0);       // cycles at 64 flop / clk / SIMD on 256 CUs
        if (N < 128 || c.cost > 1.7 * ideal) return 1;
    }
    if (cswin_tuning().ws_nwk > 0) {                        // tuning aid CSWIN_WS_LAYOUT = "nwn,nwk": forces a wave layout where it fits R
        const int a = cswin_tuning().ws_nwn, b = cswin_tuning().ws_nwk;
        if (R % b == 0) {
            const int ks = R / b;
            if (ks == 64 || ks == 128 || ks == 192 || ks == 256) c = WsCfg{ks, a, b, 0.0};
            else return 1;
        }
    }
    WsParams p = {};
    p.A = A; p.lda = R;
    p.W = W; p.ldw = mode == 0 ? R : N;
    p.M = M; p.N = N; p.R = R;
    p.stamps = g_ws_stamps;
    p.epi = *(const Epilogue*)epilogue;
    p.epi.vec_store = epilogue_vec_ok(p.epi, N);
    if (!p.epi.vec_store) return 1;                    // the persistent kernel compiles the 16-B epilogue only
    hipStream_t st = (hipStream_t)stream;
    int rc = 1;
    if (mode == 0) {
        if (epi_mode == EPI_PLAIN) rc = ws_dispatch<false, EPI_PLAIN>(c, p, st);
        else if (epi_mode == EPI_ACT) rc = ws_dispatch<false, EPI_ACT>(c, p, st);
        else if (epi_mode == EPI_RES) rc = ws_dispatch<false, EPI_RES>(c, p, st);
    } else {
        if (epi_mode == EPI_PLAIN) rc = ws_dispatch<true, EPI_PLAIN>(c, p, st);
        else if (epi_mode == EPI_GELUBWD) rc = ws_dispatch<true, EPI_GELUBWD>(c, p, st);
        else if (epi_mode == EPI_RES) rc = ws_dispatch<true, EPI_RES>(c, p, st);
    }
    return rc;
}
