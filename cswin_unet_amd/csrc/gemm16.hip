// Forward and data-gradient Linears of the bf16 mode whose BOTH operands are stored as bf16 (activations: cswin_linear_fwd /
// cswin_linear_bwd_data io_bf16 bit 0; weights: the optimiser's bf16 shadow, io_bf16 bit 2), gfx950 / CDNA4.
//
//   forward      : C[m][n] = sum_k A[m][k] W[n][k]       (cswin_unet.py:169,177,23-27 nn.Linear; A (M, K), W (N, K))
//   data gradient: C[m][k] = sum_n A[m][n] W[n][k]       (its autograd backward;          A = dy (M, N), W (N, K))
//
// With bf16 MFMAs (v_mfma_f32_32x32x16_bf16: 16x the fp32 rate) a 64 x 64 x 64 step of the tiled family (gemm.hip) holds 128
// cycles of matrix work and costs ~1800: one global-load round trip per step, because its register prefetch reaches one step
// ahead and the data then still has to be converted and written to LDS (profiles/round2_notes.md, in-kernel stamps).
// Operands that are ALREADY bf16 in memory need neither: here every step's tiles travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, no VGPRs, no ds_write), up to four steps in flight per workgroup -- the whole K = 256 panel of the
// C x C and C x 3C Linears is requested before the first MFMA --, and the only per-step synchronisation is one counted
// s_waitcnt vmcnt + s_barrier.
//
// LDS images (per step and operand: 64 rows x 128 B, written by the DMA in lane order, so the XOR swizzles are applied to the
// SOURCE addresses):
//   A, and W of the forward (rows = output column n, 64 reduction values each): 16-B chunk c of row r sits in slot c ^ ((r >> 1) & 7);
//     the MFMA fragments are ds_read_b128 (8 reduction values of one row), conflict free.
//   W of the data gradient (rows = reduction index n, 64 output columns k each): slot c ^ (4 * ((r >> 1) & 1)), the image gemm.hip
//     uses for its transposing reads (ds_read_b64_tr_b16 delivers the reduction-contiguous fragments).
// One workgroup = 4 waves = one 64 x 64 output tile (wave: 32 x 32), or 8 waves = 128 x 64 where there are many tiles.  Reduction
// lengths n x 64 or n x 64 + 32 (cswin_base: C = 96): a half last step rereads in-range chunks and skips its upper two MFMAs.
#include <mutex>
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"

namespace {

typedef __bf16 g16_bf16x8 __attribute__((ext_vector_type(8)));
typedef short g16_s16x4 __attribute__((ext_vector_type(4)));
typedef short g16_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) g16_s16x4 g16_lds_s16x4;

constexpr int G16_T = 64;                 // tile edge (rows, columns) and reduction step
constexpr int G16_IMG = G16_T * 128;      // bytes of one operand image of one step
constexpr int G16_STAGE = 2 * G16_IMG;

struct G16Params {
    const __bf16* A; long lda;            // (M, R) rows
    const __bf16* B; long ldb;            // forward: (NO, R) rows;  data gradient: (R, NO) rows
    int M, NO, R;                         // output rows / columns, reduction length (a multiple of 64)
    int tiles_m, tiles_n, nblk;
    Epilogue epi;
};

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to lds_dst + 16 * lane (lds_dst wave-uniform).  hipcc does not count
// these loads; completion is tracked by hand (g16_wait) -- nothing else in the main loop touches vmcnt.
__device__ __forceinline__ void g16_dma(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void g16_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void g16_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// S: LDS stages = ring slots in flight (2 .. 4).  BT: data gradient (W rows run along the reduction index).
// (A two-k-group variant -- eight waves, alternate 64-k steps, accumulators meeting in LDS -- was shorter stand-alone on long
// reductions and 0.06 ms slower in the step; removed in round 3, profiles/round2_notes.md.)
// MODE 2: a 128 x 64 tile on eight waves (outputs with many tiles: half the workgroups to dispatch, the W tile shared by twice
// the rows); a ring slot is the 16-KB A image + the 8-KB W image.
template <int EPI, bool BT, int S, bool PRE16, int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512) void gemm16_kernel(G16Params p) {
    constexpr int KG = 1;                            // k-groups (one; see above)
    constexpr int TM = MODE == 2 ? 2 * G16_T : G16_T;  // tile rows
    extern __shared__ __attribute__((aligned(16))) unsigned char g16_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = MODE == 2 ? wave8 : (wave8 & 3), kgrp = 0;
    const int li = lane & 31, lh = lane >> 5;
    constexpr int A_IMG = TM * 128;                  // bytes of the A image of a step
    constexpr int SLOT = KG * (A_IMG + G16_IMG);     // bytes of a ring slot
    // XCD-aware block order (as gemm.hip): each XCD takes a contiguous range of tiles, ordered [m-tile][n-tile]
    const int bid = blockIdx.x, xq = p.nblk >> 3, xr = p.nblk & 7, xcd = bid & 7;
    const int lb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int m0 = (lb / p.tiles_n) * TM, n0 = (lb % p.tiles_n) * G16_T;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;         // MODE 2: wave 0 .. 7 -> rows 0 .. 96
    const int nsteps = (p.R + G16_T * KG - 1) / (G16_T * KG);        // loop iterations (KG steps of 64 each)
    const int tail = p.R % G16_T;                    // 0, or 32 valid reduction values in the last step (KG = 1 only)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)g16_lds;   // LDS byte address of the ring

    // ---- DMA source addresses: wave w moves row groups 2w and 2w + 1 (8 rows each) of both images of a step (four DMAs per wave
    // and iteration; the 128-row tile: three)
    const int drow = lane >> 3, slot = lane & 7;
    const unsigned char* a_src[2];
    const unsigned char* b_src[2];
    int a_td[2], b_td[2];                            // byte deltas of the sources for a half last step: stay inside the row / the matrix
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int row = (2 * wave + g) * 8 + drow;                       // row of the A image
        const int brow = MODE == 2 ? wave8 * 8 + drow : row;             // row of the W image (MODE 2: one group per wave)
        const int ca = slot ^ ((row >> 1) & 7);                          // source chunk of this lane's slot
        const int am = min(m0 + row, p.M - 1);                           // clamped rows are never stored
        a_src[g] = reinterpret_cast<const unsigned char*>(p.A + (long)am * p.lda + 8 * ca);
        a_td[g] = 16 * ((ca & 3) - ca);             // chunks 4 .. 7 of a half step lie beyond the row: reread 0 .. 3 (never used)
        if (!BT) {
            const int bn = min(n0 + brow, p.NO - 1);
            const int cbf = slot ^ ((brow >> 1) & 7);
            b_src[g] = reinterpret_cast<const unsigned char*>(p.B + (long)bn * p.ldb + 8 * cbf);
            b_td[g] = 16 * ((cbf & 3) - cbf);
        } else {
            b_td[g] = (int)((long)((brow & 31) - brow) * p.ldb * 2);      // rows 32 .. 63 of a half step lie beyond the matrix
            int cb = slot ^ (((brow >> 1) & 1) << 2);
            cb = min(cb, (p.NO - n0) / 8 - 1);                           // columns beyond NO: any in-range chunk (never stored)
            b_src[g] = reinterpret_cast<const unsigned char*>(p.B + (long)brow * p.ldb + n0 + 8 * cb);
        }
    }
    auto issue = [&](int it) {
        const bool half = tail != 0 && it == nsteps - 1;                  // wave-uniform
        if constexpr (MODE == 2) {
            const unsigned st = lds0 + (unsigned)(it % S) * SLOT;
#pragma unroll
            for (int g = 0; g < 2; ++g) g16_dma(a_src[g] + (long)it * (G16_T * 2) + (half ? a_td[g] : 0), st + (unsigned)wave8 * 2048 + g * 1024);
            const unsigned char* src = BT ? b_src[0] + (long)it * G16_T * p.ldb * 2 : b_src[0] + (long)it * (G16_T * 2);
            g16_dma(src + (half ? b_td[0] : 0), st + A_IMG + (unsigned)wave8 * 1024);
        } else {
            const unsigned st = lds0 + (unsigned)(it % S) * SLOT + (unsigned)wave * 2048;
#pragma unroll
            for (int g = 0; g < 2; ++g) g16_dma(a_src[g] + (long)it * (G16_T * 2) + (half ? a_td[g] : 0), st + g * 1024);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const unsigned char* src = BT ? b_src[g] + (long)it * G16_T * p.ldb * 2 : b_src[g] + (long)it * (G16_T * 2);
                g16_dma(src + (half ? b_td[g] : 0), st + G16_IMG + g * 1024);
            }
        }
    };

    // ---- epilogue operands, requested BEFORE the first DMA (older in the in-order vmcnt queue, so the counted waits below are
    // unaffected): bias, row factors and the residual / GELU' rows of this lane's store positions (row erow + 8 p, columns ecol..+3)
    const Epilogue& e = p.epi;
    const int erow = lane >> 3, ecol = n0 + wn0 + (lane & 7) * 4;
    const bool ecol_ok = ecol < p.NO;
    f32x4 bias_v = {0.f, 0.f, 0.f, 0.f};
    if (e.bias && ecol_ok) bias_v = *reinterpret_cast<const f32x4*>(e.bias + ecol);
    float rs[4];
    f32x4 aux[4];
    u32x2 aux16[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm0 + erow + 8 * q;
        const bool ok = m < p.M && ecol_ok;
        rs[q] = (e.row_scale && ok) ? e.row_scale[m / e.rows_per_sample] : 1.0f;
        aux[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        aux16[q] = u32x2{0u, 0u};
        if (EPI == EPI_RES && ok) aux[q] = *reinterpret_cast<const f32x4*>(e.residual + (long)m * e.ldres + ecol);
        if (EPI == EPI_GELUBWD && ok) {
            if constexpr (PRE16) aux16[q] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(e.gelu_pre) + (long)m * e.ldpre + ecol);
            else aux[q] = *reinterpret_cast<const f32x4*>(e.gelu_pre + (long)m * e.ldpre + ecol);
        }
    }

    f32x16 acc0, acc1;                    // two independent MFMA chains (alternate 16-k chunks), summed before the epilogue
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    long long* stamps = e.stamps;            // debug (cswin_debug_set_stamps): the tiled family's record layout
    if (stamps && tid == 0) { stamps[8L * bid + 0] = __builtin_amdgcn_s_memtime(); stamps[8L * bid + 4] = __builtin_amdgcn_s_getreg(6164); stamps[8L * bid + 5] = __builtin_amdgcn_s_memrealtime(); }

    for (int s = 0; s < S - 1 && s < nsteps; ++s) issue(s);
    for (int step = 0; step < nsteps; ++step) {
        // this step's 4 DMAs (per wave) have landed when at most 4 * (groups issued after it) are outstanding
        const int later = min(S - 2, nsteps - 1 - step);
        constexpr int PW = MODE == 2 ? 3 : 4;          // DMAs per wave and iteration
        if (S >= 4 && later >= 2) g16_wait<2 * PW>();
        else if (S >= 3 && later >= 1) g16_wait<PW>();
        else g16_wait<0>();
        g16_barrier();                     // every wave's part of the step is in LDS; the stage read in step - 1 is free again
        if (step + S - 1 < nsteps) issue(step + S - 1);
        if (stamps && tid == 0 && step == 0) stamps[8L * bid + 1] = __builtin_amdgcn_s_memtime();
        const unsigned char* aimg = g16_lds + (step % S) * SLOT + kgrp * G16_STAGE;
        const unsigned char* bimg = aimg + A_IMG;
        const int kend = (tail != 0 && step == nsteps - 1) ? tail : G16_T;      // a half last step holds 32 reduction values
#pragma unroll
        for (int kk = 0; kk < G16_T; kk += 16) {
            if (kk >= kend) break;
            const int c = (kk >> 3) + lh;                                 // 16-B chunk of this lane's 8 reduction values
            const int ar = wm0 + li;
            const g16_bf16x8 af = *reinterpret_cast<const g16_bf16x8*>(aimg + ar * 128 + 16 * (c ^ ((ar >> 1) & 7)));
            g16_bf16x8 bf;
            if (!BT) {
                const int br = wn0 + li;
                bf = *reinterpret_cast<const g16_bf16x8*>(bimg + br * 128 + 16 * (c ^ ((br >> 1) & 7)));
            } else {
                // lane (q, p) of its 16-lane group supplies row r + q, columns c0 + 4 p .. + 3; it receives column c0 + (lane & 15) of
                // rows r .. r + 3: two reads give the 8 reduction values of this lane's output column (gemm.hip, same image)
                const int col = wn0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                const int row = kk + 8 * lh + ((lane & 15) >> 2);
                const int sw = ((row >> 1) & 1) << 2;                     // rows row and row + 4 share bit 1
                const unsigned char* a0 = bimg + row * 128 + 16 * ((col >> 3) ^ sw) + 2 * (col & 7);
                const g16_s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g16_lds_s16x4*)a0);
                const g16_s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g16_lds_s16x4*)(a0 + 4 * 128));
                const g16_s16x8 f = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                bf = __builtin_bit_cast(g16_bf16x8, f);
            }
            if ((kk >> 4) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc0, 0, 0, 0);
        }
    }
    if (stamps && tid == 0) stamps[8L * bid + 2] = __builtin_amdgcn_s_memtime();
    g16_barrier();                          // the stage ring becomes the epilogue's per-wave transpose patches
    // ---- epilogue (the vector path of gemm_epilogue.h's run_epilogue for one 32 x 32 fragment, operands already in registers):
    // C/D layout (lane = column, registers = rows) -> private [32][36] LDS patch -> lane owns 4 consecutive columns of 4 rows
    acc0 += acc1;
    float* wbuf = reinterpret_cast<float*>(g16_lds) + wave * EP_WAVE_FLOATS;      // MODE 2: eight patches (wave = 0 .. 7)
#pragma unroll
    for (int g = 0; g < 16; ++g) wbuf[((g & 3) + 8 * (g >> 2) + 4 * lh) * EP_LD + li] = acc0[g];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm0 + erow + 8 * q;
        if (m >= p.M || !ecol_ok) continue;
        f32x4 o = *reinterpret_cast<const f32x4*>(&wbuf[(erow + 8 * q) * EP_LD + (lane & 7) * 4]) + bias_v;
        if (EPI == EPI_GELUBWD) {
            const f32x4 pre = PRE16 ? widen_bf16x4(aux16[q]) : aux[q];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] *= gelu_grad_f(pre[c]);
        }
        o *= rs[q];
        if (EPI == EPI_RES) o += aux[q];
        epi_store4(e.C, (long)m * e.ldc + ecol, o, e.c_bf16);
        if (EPI == EPI_ACT) {
            f32x4 a;
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = gelu_f(o[c]);
            epi_store4(e.Cact, (long)m * e.ldact + ecol, a, e.c_bf16);
        }
    }
    if (stamps && tid == 0) { stamps[8L * bid + 3] = __builtin_amdgcn_s_memtime(); stamps[8L * bid + 6] = __builtin_amdgcn_s_memrealtime(); }
}

template <int EPI, bool BT, bool PRE16>
int g16_launch(const G16Params& p, hipStream_t st) {
    const int nsteps = (p.R + G16_T - 1) / G16_T;
    const int forced = cswin_tuning().gemm16_stages;                                                        // tuning aid: 2 .. 4
    const int forced_tm = cswin_tuning().gemm16_tm;                                                         // tuning aid: 64 / 128
    const bool tm128 = forced_tm == 128 || (forced_tm == 0 && p.nblk > 768);
    int S = nsteps >= 3 ? 3 : 2;          // 48 KB: three workgroups per CU (measured against 2 and 4 stages: profiles/round2_notes.md)
    if (forced >= 2 && forced <= 4) S = forced;
    static_assert(2 * G16_STAGE >= 4 * EP_WAVE_FLOATS * (int)sizeof(float), "the smallest ring must hold the epilogue patches");
    // dynamic-LDS opt-in of the instantiations: once per process, thread-safe (a function attribute, not a stream operation)
    static std::once_flag once;
    static hipError_t status = hipSuccess;
    std::call_once(once, [&] {
        status = hipFuncSetAttribute((const void*)gemm16_kernel<EPI, BT, 4, PRE16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G16_STAGE);
        if (status == hipSuccess) status = hipFuncSetAttribute((const void*)gemm16_kernel<EPI, BT, 3, PRE16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * G16_STAGE);
        if (status == hipSuccess) status = hipFuncSetAttribute((const void*)gemm16_kernel<EPI, BT, 3, PRE16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (3 * G16_IMG));
    });
    if (status != hipSuccess) return 1;
    if (tm128) {                           // 128 x 64 tiles: three slots of 24 KB, two workgroups per CU
        G16Params q = p;
        q.tiles_m = cdiv(p.M, 2 * G16_T);
        q.nblk = q.tiles_m * q.tiles_n;
        hipLaunchKernelGGL((gemm16_kernel<EPI, BT, 3, PRE16, 2>), dim3(q.nblk), dim3(512), 3 * (3 * G16_IMG), st, q);
        return 0;
    }
    const size_t lds = (size_t)S * G16_STAGE;
    if (S == 4) hipLaunchKernelGGL((gemm16_kernel<EPI, BT, 4, PRE16, 0>), dim3(p.nblk), dim3(256), lds, st, p);
    else if (S == 3) hipLaunchKernelGGL((gemm16_kernel<EPI, BT, 3, PRE16, 0>), dim3(p.nblk), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemm16_kernel<EPI, BT, 2, PRE16, 0>), dim3(p.nblk), dim3(256), lds, st, p);
    return 0;
}

}  // namespace

// Internal entry used by gemm.hip's cswin_linear_fwd / cswin_linear_bwd_data (precision 1, io_bf16 bits 0 and 2 both set, plain
// single-source forms).  mode 0: forward (A (M, R), B (NO, R));  mode 1: data gradient (A (M, R), B (R, NO)).
// Returns 0 when launched, 1 when the shape is not covered (the caller falls back to the tiled family).
int cswin_gemm16(int mode, int epi_mode, const void* A, const void* B, const void* epilogue, int M, int NO, int R, void* stream) {
    const bool off = !cswin_tuning().gemm16_on;                                                             // tuning aid
    if (off || (R % G16_T != 0 && R % G16_T != 32) || R < G16_T || NO % 8 != 0 || M < 1) return 1;    // reduction: n x 64 (+ 32)
    if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return 1;
    G16Params p;
    p.A = (const __bf16*)A; p.lda = R;
    p.B = (const __bf16*)B; p.ldb = mode == 0 ? R : NO;
    p.M = M; p.NO = NO; p.R = R;
    p.tiles_m = cdiv(M, G16_T); p.tiles_n = cdiv(NO, G16_T);
    p.nblk = p.tiles_m * p.tiles_n;
    p.epi = *(const Epilogue*)epilogue;
    p.epi.vec_store = epilogue_vec_ok(p.epi, NO);
    if (!p.epi.vec_store) return 1;          // bf16-stored outputs / auxiliaries need the 16-B epilogue path anyway
    if (p.epi.rm_on || p.epi.C2 || p.epi.colsum || p.epi.split_stride) return 1;      // forms the kernel's own epilogue does not implement
    hipStream_t st = (hipStream_t)stream;
    const bool pre16 = p.epi.pre_bf16 != 0;
    if (mode == 0) {
        switch (epi_mode) {
            case EPI_PLAIN: return g16_launch<EPI_PLAIN, false, false>(p, st);
            case EPI_ACT: return g16_launch<EPI_ACT, false, false>(p, st);
            case EPI_RES: return g16_launch<EPI_RES, false, false>(p, st);
            default: return 1;
        }
    }
    switch (epi_mode) {
        case EPI_PLAIN: return g16_launch<EPI_PLAIN, true, false>(p, st);
        case EPI_GELUBWD: return pre16 ? g16_launch<EPI_GELUBWD, true, true>(p, st) : g16_launch<EPI_GELUBWD, true, false>(p, st);
        default: return 1;
    }
}
