// Epilogue shared by the GEMM kernels of libcswin_hip (gemm.hip: tiled family; gemm16.hip: LDS-DMA bf16 kernel).
#pragma once
#include "common.h"

namespace {

typedef __bf16 epi_bf16x4 __attribute__((ext_vector_type(4)));
// widen 4 bf16 held as two raw dwords.  Raw bf16 bits travel in INTEGER vectors: hipcc 7.2 miscompiles bit casts of the
// elements of a 2-float vector (element 1 reads element 0).
__device__ __forceinline__ f32x4 widen_bf16x4(u32x2 raw) {
    return f32x4{__builtin_bit_cast(float, raw[0] << 16), __builtin_bit_cast(float, raw[0] & 0xffff0000u),
                 __builtin_bit_cast(float, raw[1] << 16), __builtin_bit_cast(float, raw[1] & 0xffff0000u)};
}
__device__ __forceinline__ void epi_store4(float* base, long idx, f32x4 v, int bf16) {
    if (bf16) *reinterpret_cast<epi_bf16x4*>(reinterpret_cast<__bf16*>(base) + idx) = __builtin_convertvector(v, epi_bf16x4);
    else *reinterpret_cast<f32x4*>(base + idx) = v;
}

// ------------------------------------------------------------------------------------
// epilogue
// ------------------------------------------------------------------------------------
enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_RES = 2, EPI_GELUBWD = 3, EPI_SPLIT2 = 4 };

struct Epilogue {
    float* C; long ldc;
    float* C2; long ldc2; int col_split;      // EPI_SPLIT2: columns >= col_split are written to C2[m][n - col_split]
    const float* bias;                        // [N] or NULL (any mode)
    float* Cact; long ldact;                  // EPI_ACT: C = acc + bias (pre-activation), Cact = gelu(C)
    const float* residual; long ldres;        // EPI_RES: C = residual + row_scale * (acc + bias)
    const float* row_scale; int rows_per_sample;   // any mode: per-row multiplier (NULL = 1)
    const float* gelu_pre; long ldpre;        // EPI_GELUBWD: C = row_scale * acc * gelu'(gelu_pre[m][n])
    long split_stride;                        // C += split * split_stride (split-R partial slabs)
    float* colsum; int colsum_stride;         // TN only: partial column sums of A (dbias), [split][M]
    int rm_on, rm_H, rm_W, rm_H2, rm_W2, rm_py, rm_px;   // output row m is a parity-class pixel index -> full (b, iy, ix) row
    int vec_store;                            // 1: every output / auxiliary row is 16-B aligned and N % 4 == 0
    int c_bf16;                               // C and Cact are stored as bf16 (vector path only; bf16 activation storage)
    int pre_bf16;                             // gelu_pre is stored as bf16
    long long* stamps;                        // debug: per-workgroup s_memtime stamps [nblk][8] (NULL in production)
};

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), i.e. a lane owns one
// COLUMN of the fragment.  Storing from that layout is 16 dword stores per fragment (store-issue bound, measured 20-26 %
// of a workgroup's life).  Each wave therefore transposes its fragment through a private [32][36] LDS patch so that a
// lane owns 4 consecutive columns of one row: bias / residual / gelu' operands are read and the result is written
// with 16-B accesses, 8 full 128-B lines per wave instruction.
constexpr int EP_LD = 36;
constexpr int EP_WAVE_FLOATS = 32 * EP_LD;

// PRE16: gelu_pre is stored as bf16 (EPI_GELUBWD) -- compile-time, so that the four auxiliary loads of a fragment stay in one
// basic block and are issued back to back.
template <int EPI, int FM, int FN, bool PRE16 = false>
__device__ __forceinline__ void run_epilogue(const Epilogue& e, f32x16 (&acc)[FM][FN], int M, int N, int mb, int nb,
                                             int lane, float* wbuf, bool vec) {
    const int li = lane & 31, lh = lane >> 5;
    const int rrow = lane >> 3, rcol = (lane & 7) * 4;
    const bool has_rs = e.row_scale != nullptr;
    auto out_row = [&](int m) -> long {           // class-local pixel -> row of the full (B, H*W, C) token matrix
        if (!e.rm_on) return m;
        const int hw = e.rm_H2 * e.rm_W2;
        const int b = m / hw, rem = m - b * hw, jy = rem / e.rm_W2;
        return ((long)b * e.rm_H + 2 * jy + e.rm_py) * e.rm_W + 2 * (rem - jy * e.rm_W2) + e.rm_px;
    };
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int g = 0; g < 16; ++g) wbuf[((g & 3) + 8 * (g >> 2) + 4 * lh) * EP_LD + li] = acc[i][j][g];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int n = nb + j * 32 + rcol;
            if (vec) {
                f32x4 bias_v = {0.f, 0.f, 0.f, 0.f};
                if (e.bias && n < N) bias_v = *reinterpret_cast<const f32x4*>(e.bias + n);
                f32x4 v[4], aux[4];
                u32x2 aux16[4];
                float rs[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int m = mb + i * 32 + rrow + 8 * p;
                    const bool ok = m < M && n < N;
                    v[p] = *reinterpret_cast<const f32x4*>(&wbuf[(rrow + 8 * p) * EP_LD + rcol]);
                    rs[p] = (has_rs && ok) ? e.row_scale[m / e.rows_per_sample] : 1.0f;
                    if (EPI == EPI_RES || EPI == EPI_GELUBWD) {
                        const float* ap = EPI == EPI_RES ? e.residual : e.gelu_pre;
                        const long ld = EPI == EPI_RES ? e.ldres : e.ldpre;
                        aux[p] = f32x4{0.f, 0.f, 0.f, 0.f};
                        aux16[p] = u32x2{0u, 0u};
                        if (ok) {
                            if constexpr (PRE16) aux16[p] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(ap) + (long)m * ld + n);
                            else aux[p] = *reinterpret_cast<const f32x4*>(ap + (long)m * ld + n);
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int m = mb + i * 32 + rrow + 8 * p;
                    if (m >= M || n >= N) continue;
                    f32x4 o = v[p] + bias_v;
                    if (EPI == EPI_GELUBWD) {
                        const f32x4 pre = PRE16 ? widen_bf16x4(aux16[p]) : aux[p];
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] *= gelu_grad_f(pre[c]);
                    }
                    if (has_rs) o *= rs[p];
                    if (EPI == EPI_RES) o += aux[p];
                    if (EPI == EPI_SPLIT2 && n >= e.col_split)
                        *reinterpret_cast<f32x4*>(e.C2 + (long)m * e.ldc2 + (n - e.col_split)) = o;
                    else
                        epi_store4(e.C, out_row(m) * e.ldc + n, o, e.c_bf16);
                    if (EPI == EPI_ACT) {
                        f32x4 a;
#pragma unroll
                        for (int c = 0; c < 4; ++c) a[c] = gelu_f(o[c]);
                        epi_store4(e.Cact, (long)m * e.ldact + n, a, e.c_bf16);
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int m = mb + i * 32 + rrow + 8 * p;
                    if (m >= M) continue;
                    const float rsv = has_rs ? e.row_scale[m / e.rows_per_sample] : 1.0f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int nn = n + c;
                        if (nn >= N) continue;
                        float o = wbuf[(rrow + 8 * p) * EP_LD + rcol + c] + (e.bias ? e.bias[nn] : 0.f);
                        if (EPI == EPI_GELUBWD) o *= gelu_grad_f(e.gelu_pre[(long)m * e.ldpre + nn]);
                        o *= rsv;
                        if (EPI == EPI_RES) o += e.residual[(long)m * e.ldres + nn];
                        if (EPI == EPI_SPLIT2 && nn >= e.col_split) e.C2[(long)m * e.ldc2 + (nn - e.col_split)] = o;
                        else e.C[out_row(m) * e.ldc + nn] = o;
                        if (EPI == EPI_ACT) e.Cact[(long)m * e.ldact + nn] = gelu_f(o);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}


inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 16-B epilogue accesses are legal when every row of every output / auxiliary operand starts 16-B aligned
inline int epilogue_vec_ok(const Epilogue& e, int n_out) {
    auto ok = [](const void* p, long ld) { return !p || (aligned16(p) && ld % 4 == 0); };
    return n_out % 4 == 0 && aligned16(e.C) && e.ldc % 4 == 0 && ok(e.C2, e.ldc2) && (!e.C2 || e.col_split % 4 == 0) &&
           ok(e.bias, 4) && ok(e.Cact, e.ldact) && ok(e.residual, e.ldres) && ok(e.gelu_pre, e.ldpre) &&
           e.split_stride % 4 == 0;
}

}  // namespace
