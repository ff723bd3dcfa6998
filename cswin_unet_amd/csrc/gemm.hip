// fp32 MFMA GEMM family for the CSWin-UNet hot path (gfx950 / CDNA4).
//
//   C[m][n] = epilogue( sum_r A(m, r) * B(n, r) )
//
// One kernel template serves every dense contraction of the model:
//   * Linear forward        (A = X  [M,K] r-contiguous, B = W [N,K] r-contiguous)      "NT"
//   * Linear data-gradient  (A = dY [M,N] r-contiguous, B = W [N,K] row-contiguous)    "NN"
//   * Linear weight-gradient(A = dY, B = X, both row-contiguous, split over M)         "TN"
//   * 3x3 / strided convolutions on the (B, L, C) token layout as implicit GEMM: the
//     A operand is gathered on the fly from NHWC tokens (no im2col buffer, no NCHW
//     transposes -- replaces the transpose/contiguous/conv/transpose chain of
//     networks/cswin_unet.py:214-217, 235-241).
//
// Matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TF/s chip peak).
// The k index inside an MFMA is arbitrary as long as A and B agree, so an r-contiguous
// operand is read from its [row][r] LDS image with ONE ds_read_b128 per four MFMA steps
// (lane half h takes r = kk + 4h + s), and a row-contiguous operand from its [r][row]
// image with conflict-free ds_read_b32.  Global loads are 16 B per lane; LDS is single
// buffered with register prefetch of the next tile (4 workgroups of 4 waves co-reside per CU
// and cover each other's barriers).
#include "common.h"

namespace {

constexpr int BK = 32;        // reduction tile
constexpr int LDR = BK + 4;   // [row][r] image: 144-B rows -> conflict-free b128 reads

// ------------------------------------------------------------------------------------
// operand sources: a logical matrix S(i, j) whose fast (contiguous) index is j
// ------------------------------------------------------------------------------------
struct PlainSrc {
    const float* p; long ld; int rows, cols;
    const float* row_scale; int rows_per_sample;     // optional per-row multiplier (DropPath backward)
    struct Row { const float* base; float s; };
    __device__ Row row(int i) const {
        Row r;
        r.base = (i < rows) ? p + (long)i * ld : nullptr;
        r.s = (row_scale && i < rows) ? row_scale[i / rows_per_sample] : 1.0f;
        return r;
    }
    __device__ const float* ptr(const Row& r, int j) const { return (r.base && j < cols) ? r.base + j : nullptr; }
};

// cat([p0 (c0 cols), p1 (cols - c0)], dim=-1) without materialising it (skip-concat, cswin_unet.py:509-510)
struct ConcatSrc {
    const float* p0; const float* p1; long ld0, ld1; int rows, cols, c0;
    struct Row { const float* b0; const float* b1; float s; };
    __device__ Row row(int i) const {
        Row r;
        r.b0 = (i < rows) ? p0 + (long)i * ld0 : nullptr;
        r.b1 = (i < rows) ? p1 + (long)i * ld1 : nullptr;
        r.s = 1.0f;
        return r;
    }
    __device__ const float* ptr(const Row& r, int j) const {
        if (!r.b0 || j >= cols) return nullptr;
        return j < c0 ? r.b0 + j : r.b1 + (j - c0);
    }
};

// implicit im2col of NHWC tokens: i = output pixel (b, oy, ox), j = tap * C + ci
struct ConvSrc {
    const float* x; int B, H, W, C, OH, OW, ks, stride, pad; int rows, cols;
    struct Row { int b, iy0, ix0; float s; };
    __device__ Row row(int i) const {
        Row r;
        r.s = 1.0f;
        if (i >= rows) { r.b = -1; r.iy0 = r.ix0 = 0; return r; }
        int ohw = OH * OW;
        r.b = i / ohw;
        int rem = i - r.b * ohw;
        int oy = rem / OW;
        r.iy0 = oy * stride - pad;
        r.ix0 = (rem - oy * OW) * stride - pad;
        return r;
    }
    __device__ const float* ptr(const Row& r, int j) const {
        if (r.b < 0 || j >= cols) return nullptr;
        int tap = j / C, ci = j - tap * C;
        int ky = tap / ks, kx = tap - ky * ks;
        int iy = r.iy0 + ky, ix = r.ix0 + kx;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return nullptr;
        return x + ((long)(r.b * H + iy) * W + ix) * C + ci;
    }
};

// transposed gather for the conv data-gradient: i = input pixel (b, iy, ix), j = tap * C + co over dy (NHWC, OHxOW)
struct ConvTSrc {
    const float* dy; int B, H, W, C, OH, OW, ks, stride, pad; int rows, cols;   // C = Cout here
    struct Row { int b, iy, ix; float s; };
    __device__ Row row(int i) const {
        Row r;
        r.s = 1.0f;
        if (i >= rows) { r.b = -1; r.iy = r.ix = 0; return r; }
        int hw = H * W;
        r.b = i / hw;
        int rem = i - r.b * hw;
        r.iy = rem / W;
        r.ix = rem - r.iy * W;
        return r;
    }
    __device__ const float* ptr(const Row& r, int j) const {
        if (r.b < 0 || j >= cols) return nullptr;
        int tap = j / C, co = j - tap * C;
        int ky = tap / ks, kx = tap - ky * ks;
        int ty = r.iy + pad - ky, tx = r.ix + pad - kx;
        if (ty < 0 || tx < 0) return nullptr;
        int oy = ty / stride, ox = tx / stride;
        if (oy * stride != ty || ox * stride != tx || oy >= OH || ox >= OW) return nullptr;
        return dy + ((long)(r.b * OH + oy) * OW + ox) * C + co;
    }
};

// ------------------------------------------------------------------------------------
// epilogue
// ------------------------------------------------------------------------------------
struct Epilogue {
    float* C; long ldc;
    float* C2; long ldc2; int col_split;      // columns >= col_split are written to C2[m][n - col_split]
    const float* bias;                        // [N]
    float* Cact; long ldact;                  // gelu(acc + bias) second output (C keeps the pre-activation)
    const float* residual; long ldres;        // C = residual + row_scale * (acc + bias)
    const float* row_scale; int rows_per_sample;
    const float* gelu_pre; long ldpre;        // C = acc * gelu'(gelu_pre[m][n])
    long split_stride;                        // C += split * split_stride (split-R partial slabs)
    float* colsum; int colsum_stride;         // TN only: partial column sums of A (dbias), [split][M]

    __device__ __forceinline__ void store(int m, int n, float v) const {
        if (bias) v += bias[n];
        if (gelu_pre) v *= gelu_grad_f(gelu_pre[(long)m * ldpre + n]);
        if (row_scale) v *= row_scale[m / rows_per_sample];
        if (residual) v += residual[(long)m * ldres + n];
        if (C2 && n >= col_split) C2[(long)m * ldc2 + (n - col_split)] = v;
        else C[(long)m * ldc + n] = v;
        if (Cact) Cact[(long)m * ldact + n] = gelu_f(v);
    }
};

template <int VEC>
__device__ __forceinline__ f32x4 load_chunk(const float* p) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p) {
        if (VEC == 4) v = *reinterpret_cast<const f32x4*>(p);
    }
    return v;
}

// ------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------
template <int BM, int BN, bool A_RC, bool B_RC, int VEC, class ASrc, class BSrc>
__global__ __launch_bounds__(256) void gemm_kernel(ASrc A, BSrc B, Epilogue epi, int M, int N, int R,
                                                    int r_per_split, int tiles_m) {
    constexpr int WM = BM / 2, WN = BN / 2;          // 4 waves as 2 x 2
    constexpr int FM = WM / 32, FN = WN / 32;        // 32x32 fragments per wave
    constexpr int LDA = A_RC ? LDR : BM + 4;
    constexpr int LDB = B_RC ? LDR : BN + 4;
    constexpr int A_ELEMS = A_RC ? BM * LDR : BK * (BM + 4);
    constexpr int B_ELEMS = B_RC ? BN * LDR : BK * (BN + 4);
    __shared__ __attribute__((aligned(16))) float lds[A_ELEMS + B_ELEMS];
    float* As = lds;
    float* Bs = lds + A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x;
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
    const int split = blockIdx.y;
    const int r_begin = split * r_per_split;
    const int r_end = min(R, r_begin + r_per_split);
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    // loader geometry
    constexpr int QA = BM / 32, QB = BN / 32;               // chunks (16 B) per thread per tile
    constexpr int A_CPR = A_RC ? 8 : BM / 4;                // chunks per LDS row
    constexpr int B_CPR = B_RC ? 8 : BN / 4;
    constexpr int A_RPP = 256 / A_CPR, B_RPP = 256 / B_CPR; // rows per pass
    const int a_c = tid % A_CPR, a_r = tid / A_CPR;
    const int b_c = tid % B_CPR, b_r = tid / B_CPR;

    typename ASrc::Row arow[QA];
    typename BSrc::Row brow[QB];
    if (A_RC) {
#pragma unroll
        for (int q = 0; q < QA; ++q) arow[q] = A.row(m0 + a_r + q * A_RPP);
    }
    if (B_RC) {
#pragma unroll
        for (int q = 0; q < QB; ++q) brow[q] = B.row(n0 + b_r + q * B_RPP);
    }

    f32x4 pa[QA], pb[QB];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            if (A_RC) {
                int j = r0 + 4 * a_c;
                if (VEC == 4) {
                    pa[q] = load_chunk<4>(j < r_end ? A.ptr(arow[q], j) : nullptr) * arow[q].s;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* p = (j + e < r_end) ? A.ptr(arow[q], j + e) : nullptr;
                        pa[q][e] = p ? *p * arow[q].s : 0.f;
                    }
                }
            } else {
                int r = r0 + a_r + q * A_RPP;
                typename ASrc::Row rr = A.row(r < r_end ? r : 0x7fffffff);
                int j = m0 + 4 * a_c;
                if (VEC == 4) {
                    pa[q] = load_chunk<4>(A.ptr(rr, j)) * rr.s;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* p = A.ptr(rr, j + e);
                        pa[q][e] = p ? *p * rr.s : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if (B_RC) {
                int j = r0 + 4 * b_c;
                if (VEC == 4) {
                    pb[q] = load_chunk<4>(j < r_end ? B.ptr(brow[q], j) : nullptr) * brow[q].s;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* p = (j + e < r_end) ? B.ptr(brow[q], j + e) : nullptr;
                        pb[q][e] = p ? *p * brow[q].s : 0.f;
                    }
                }
            } else {
                int r = r0 + b_r + q * B_RPP;
                typename BSrc::Row rr = B.row(r < r_end ? r : 0x7fffffff);
                int j = n0 + 4 * b_c;
                if (VEC == 4) {
                    pb[q] = load_chunk<4>(B.ptr(rr, j)) * rr.s;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* p = B.ptr(rr, j + e);
                        pb[q][e] = p ? *p * rr.s : 0.f;
                    }
                }
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < QA; ++q)
            *reinterpret_cast<f32x4*>(&As[(a_r + q * A_RPP) * LDA + 4 * a_c]) = pa[q];
#pragma unroll
        for (int q = 0; q < QB; ++q)
            *reinterpret_cast<f32x4*>(&Bs[(b_r + q * B_RPP) * LDB + 4 * b_c]) = pb[q];
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float csum = 0.f;   // dbias partial (TN, A row-contiguous image: column tid of the A tile)

    if (r_begin < r_end) {
        fetch(r_begin);
        stash();
        __syncthreads();
        for (int r0 = r_begin; r0 < r_end; r0 += BK) {
            const bool more = r0 + BK < r_end;
            if (more) fetch(r0 + BK);
            if (!A_RC && epi.colsum && n0 == 0 && tid < BM) {
#pragma unroll 8
                for (int r = 0; r < BK; ++r) csum += As[r * LDA + tid];
            }
#pragma unroll
            for (int kk = 0; kk < BK; kk += 8) {
                f32x4 af[FM], bf[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if (A_RC) {
                        af[i] = *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + li) * LDA + kk + 4 * lh]);
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) af[i][s] = As[(kk + 4 * lh + s) * LDA + wm0 + i * 32 + li];
                    }
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if (B_RC) {
                        bf[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + j * 32 + li) * LDB + kk + 4 * lh]);
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) bf[j][s] = Bs[(kk + 4 * lh + s) * LDB + wn0 + j * 32 + li];
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                stash();
                __syncthreads();
            }
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    Epilogue e = epi;
    e.C += (long)split * e.split_stride;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn0 + j * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int m = m0 + wm0 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (m < M && n < N) e.store(m, n, acc[i][j][g]);
            }
        }
    if (!A_RC && epi.colsum && n0 == 0 && tid < BM && m0 + tid < M)
        epi.colsum[(long)split * epi.colsum_stride + m0 + tid] = csum;
}

// out[i] = sum_s part[s][i]     (split-R slab reduction; deterministic order)
__global__ void reduce_slabs_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int splits,
                                    long stride) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(long)k * stride + i];
    out[i] = s;
}

template <int BM, int BN, bool A_RC, bool B_RC, int VEC, class ASrc, class BSrc>
void launch_cfg(const ASrc& A, const BSrc& B, const Epilogue& epi, int M, int N, int R, int splits, int r_per_split,
                hipStream_t st) {
    int tm = cdiv(M, BM), tn = cdiv(N, BN);
    dim3 grid(tm * tn, splits);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, A_RC, B_RC, VEC, ASrc, BSrc>), grid, dim3(256), 0, st, A, B, epi, M, N, R,
                       r_per_split, tm);
}

// tile choice: biggest tile that still gives >= ~2 workgroups per CU; prefer BN that divides N
template <bool A_RC, bool B_RC, int VEC, class ASrc, class BSrc>
void launch_gemm(const ASrc& A, const BSrc& B, const Epilogue& epi, int M, int N, int R, int splits, int r_per_split,
                 hipStream_t st) {
    auto blocks = [&](int bm, int bn) { return (long)cdiv(M, bm) * cdiv(N, bn) * splits; };
    auto waste = [&](int bn) { return (double)cdiv(N, bn) * bn / N; };
    const long want = 512;
    if (blocks(128, 128) >= want && waste(128) <= 1.15)
        launch_cfg<128, 128, A_RC, B_RC, VEC>(A, B, epi, M, N, R, splits, r_per_split, st);
    else if (blocks(128, 64) >= want)
        launch_cfg<128, 64, A_RC, B_RC, VEC>(A, B, epi, M, N, R, splits, r_per_split, st);
    else
        launch_cfg<64, 64, A_RC, B_RC, VEC>(A, B, epi, M, N, R, splits, r_per_split, st);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

Epilogue plain_epilogue(float* C, long ldc) {
    Epilogue e = {};
    e.C = C;
    e.ldc = ldc;
    return e;
}

// split count for the M-reduction of a weight gradient: enough workgroups to fill the chip
void choose_split(int M, int out_rows, int out_cols, int* splits, int* r_per_split) {
    long tiles = (long)cdiv(out_rows, 64) * cdiv(out_cols, 64);
    int s = (int)((768 + tiles - 1) / tiles);
    int max_s = cdiv(M, 4 * BK);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int rps = cdiv(cdiv(M, s), BK) * BK;
    *splits = cdiv(M, rps);
    *r_per_split = rps;
}

}  // namespace

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" {

int cswin_linear_fwd(const float* x, const float* x2, int k_split, const float* w, const float* bias, float* y,
                     float* y_act, const float* residual, const float* row_scale, int rows_per_sample, int M, int N,
                     int K, void* stream) {
    CSWIN_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0, CSWIN_ERR_SHAPE, "linear_fwd: bad arguments M=%d N=%d K=%d", M, N, K);
    CSWIN_REQUIRE(!x2 || (k_split > 0 && k_split < K), CSWIN_ERR_SHAPE, "linear_fwd: bad concat split %d of K=%d", k_split, K);
    CSWIN_REQUIRE(!row_scale || rows_per_sample > 0, CSWIN_ERR_SHAPE, "linear_fwd: rows_per_sample must be > 0");
    hipStream_t st = (hipStream_t)stream;
    Epilogue e = plain_epilogue(y, N);
    e.bias = bias;
    e.Cact = y_act; e.ldact = N;
    e.residual = residual; e.ldres = N;
    e.row_scale = row_scale; e.rows_per_sample = rows_per_sample;
    PlainSrc B = {w, K, N, K, nullptr, 1};
    if (x2) {
        ConcatSrc A = {x, x2, k_split, K - k_split, M, K, k_split};
        bool vec = (k_split % 4 == 0) && (K % 4 == 0) && aligned16(x) && aligned16(x2) && aligned16(w);
        if (vec) launch_gemm<true, true, 4>(A, B, e, M, N, K, 1, cdiv(K, BK) * BK, st);
        else launch_gemm<true, true, 1>(A, B, e, M, N, K, 1, cdiv(K, BK) * BK, st);
    } else {
        PlainSrc A = {x, K, M, K, nullptr, 1};
        bool vec = (K % 4 == 0) && aligned16(x) && aligned16(w);
        if (vec) launch_gemm<true, true, 4>(A, B, e, M, N, K, 1, cdiv(K, BK) * BK, st);
        else launch_gemm<true, true, 1>(A, B, e, M, N, K, 1, cdiv(K, BK) * BK, st);
    }
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// dx[M,K] = (row_scale . dy)[M,N] @ w[N,K]   (optionally * gelu'(gelu_pre), + add; optionally split into dx | dx2)
int cswin_linear_bwd_data(const float* dy, const float* w, float* dx, float* dx2, int k_split, const float* gelu_pre,
                          const float* row_scale, int rows_per_sample, const float* add, int M, int N, int K,
                          void* stream) {
    CSWIN_REQUIRE(dy && w && dx && M > 0 && N > 0 && K > 0, CSWIN_ERR_SHAPE, "linear_bwd_data: bad arguments");
    CSWIN_REQUIRE(!dx2 || (k_split > 0 && k_split < K), CSWIN_ERR_SHAPE, "linear_bwd_data: bad concat split");
    CSWIN_REQUIRE(!row_scale || rows_per_sample > 0, CSWIN_ERR_SHAPE, "linear_bwd_data: rows_per_sample must be > 0");
    hipStream_t st = (hipStream_t)stream;
    Epilogue e = plain_epilogue(dx, dx2 ? k_split : K);
    e.C2 = dx2; e.ldc2 = K - k_split; e.col_split = dx2 ? k_split : 0;
    e.gelu_pre = gelu_pre; e.ldpre = K;
    e.row_scale = row_scale; e.rows_per_sample = rows_per_sample;
    e.residual = add; e.ldres = K;
    PlainSrc A = {dy, N, M, N, nullptr, 1};
    PlainSrc B = {w, K, N, K, nullptr, 1};     // S(i = n (reduction), j = k): row-contiguous image
    bool vec = (N % 4 == 0) && (K % 4 == 0) && aligned16(dy) && aligned16(w);
    // output rows = M, output cols = K, reduction = N
    if (vec) launch_gemm<true, false, 4>(A, B, e, M, K, N, 1, cdiv(N, BK) * BK, st);
    else launch_gemm<true, false, 1>(A, B, e, M, K, N, 1, cdiv(N, BK) * BK, st);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

size_t cswin_linear_bwd_weight_workspace(int M, int N, int K) {
    int splits, rps;
    choose_split(M, N, K, &splits, &rps);
    return (size_t)splits * ((size_t)N * K + N) * sizeof(float);
}

// dw[N,K] = (row_scale . dy)^T @ [x | x2];  dbias[N] = colsum(row_scale . dy)
int cswin_linear_bwd_weight(const float* dy, const float* x, const float* x2, int k_split, const float* row_scale,
                            int rows_per_sample, float* dw, float* dbias, void* workspace, size_t ws_bytes, int M,
                            int N, int K, void* stream) {
    CSWIN_REQUIRE(dy && x && dw && M > 0 && N > 0 && K > 0, CSWIN_ERR_SHAPE, "linear_bwd_weight: bad arguments");
    CSWIN_REQUIRE(!x2 || (k_split > 0 && k_split < K), CSWIN_ERR_SHAPE, "linear_bwd_weight: bad concat split");
    CSWIN_REQUIRE(!row_scale || rows_per_sample > 0, CSWIN_ERR_SHAPE, "linear_bwd_weight: rows_per_sample must be > 0");
    size_t need = cswin_linear_bwd_weight_workspace(M, N, K);
    CSWIN_REQUIRE(workspace && ws_bytes >= need, CSWIN_ERR_WORKSPACE, "linear_bwd_weight: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    int splits, rps;
    choose_split(M, N, K, &splits, &rps);
    float* slab = (float*)workspace;
    float* bslab = slab + (size_t)splits * N * K;
    Epilogue e = plain_epilogue(slab, K);
    e.split_stride = (long)N * K;
    e.colsum = dbias ? bslab : nullptr;
    e.colsum_stride = N;
    PlainSrc A = {dy, N, M, N, row_scale, rows_per_sample};      // S(i = m (reduction), j = n)
    if (x2) {
        ConcatSrc B = {x, x2, k_split, K - k_split, M, K, k_split};
        bool vec = (N % 4 == 0) && (K % 4 == 0) && (k_split % 4 == 0) && aligned16(dy) && aligned16(x) && aligned16(x2);
        if (vec) launch_gemm<false, false, 4>(A, B, e, N, K, M, splits, rps, st);
        else launch_gemm<false, false, 1>(A, B, e, N, K, M, splits, rps, st);
    } else {
        PlainSrc B = {x, K, M, K, nullptr, 1};
        bool vec = (N % 4 == 0) && (K % 4 == 0) && aligned16(dy) && aligned16(x);
        if (vec) launch_gemm<false, false, 4>(A, B, e, N, K, M, splits, rps, st);
        else launch_gemm<false, false, 1>(A, B, e, N, K, M, splits, rps, st);
    }
    CSWIN_LAUNCH_CHECK();
    long n = (long)N * K;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, slab, dw, n, splits, n);
    if (dbias) hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, bslab, dbias, (long)N, splits, (long)N);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// -------- convolutions on the (B, H*W, C) token layout (NHWC), implicit GEMM -----------------------------------
// w_perm: [Cout][ks*ks][Cin]  (made by cswin_conv_weight_permute from the nn.Conv2d [Cout][Cin][ks][ks] parameter)
int cswin_conv_tok_fwd(const float* x, const float* w_perm, const float* bias, float* y, int B, int H, int W, int Cin,
                       int Cout, int ks, int stride, int pad, void* stream) {
    CSWIN_REQUIRE(x && w_perm && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, CSWIN_ERR_SHAPE, "conv_tok_fwd: bad arguments");
    CSWIN_REQUIRE(Cin % 4 == 0 && aligned16(x) && aligned16(w_perm), CSWIN_ERR_ALIGN, "conv_tok_fwd: Cin %% 4 and 16-B alignment required");
    int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    int M = B * OH * OW, R = ks * ks * Cin;
    ConvSrc A = {x, B, H, W, Cin, OH, OW, ks, stride, pad, M, R};
    PlainSrc Bm = {w_perm, R, Cout, R, nullptr, 1};
    Epilogue e = plain_epilogue(y, Cout);
    e.bias = bias;
    launch_gemm<true, true, 4>(A, Bm, e, M, Cout, R, 1, cdiv(R, BK) * BK, (hipStream_t)stream);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// w_permT: [ks*ks][Cout][Cin]
int cswin_conv_tok_bwd_data(const float* dy, const float* w_permT, float* dx, int B, int H, int W, int Cin, int Cout,
                            int ks, int stride, int pad, void* stream) {
    CSWIN_REQUIRE(dy && w_permT && dx, CSWIN_ERR_SHAPE, "conv_tok_bwd_data: null pointer");
    CSWIN_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && aligned16(dy) && aligned16(w_permT), CSWIN_ERR_ALIGN, "conv_tok_bwd_data: channels %% 4 and 16-B alignment required");
    int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    int M = B * H * W, R = ks * ks * Cout;
    ConvTSrc A = {dy, B, H, W, Cout, OH, OW, ks, stride, pad, M, R};
    PlainSrc Bm = {w_permT, Cin, R, Cin, nullptr, 1};          // S(i = (tap, co), j = ci)
    Epilogue e = plain_epilogue(dx, Cin);
    launch_gemm<true, false, 4>(A, Bm, e, M, Cin, R, 1, cdiv(R, BK) * BK, (hipStream_t)stream);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

size_t cswin_conv_tok_bwd_weight_workspace(int B, int H, int W, int Cin, int Cout, int ks, int stride, int pad) {
    int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    return cswin_linear_bwd_weight_workspace(B * OH * OW, Cout, ks * ks * Cin);
}

// dw_perm: [Cout][ks*ks][Cin]; dbias: [Cout]
int cswin_conv_tok_bwd_weight(const float* dy, const float* x, float* dw_perm, float* dbias, void* workspace,
                              size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ks, int stride, int pad,
                              void* stream) {
    CSWIN_REQUIRE(dy && x && dw_perm, CSWIN_ERR_SHAPE, "conv_tok_bwd_weight: null pointer");
    CSWIN_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && aligned16(dy) && aligned16(x), CSWIN_ERR_ALIGN, "conv_tok_bwd_weight: channels %% 4 and 16-B alignment required");
    int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    int M = B * OH * OW, K = ks * ks * Cin;
    size_t need = cswin_linear_bwd_weight_workspace(M, Cout, K);
    CSWIN_REQUIRE(workspace && ws_bytes >= need, CSWIN_ERR_WORKSPACE, "conv_tok_bwd_weight: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    int splits, rps;
    choose_split(M, Cout, K, &splits, &rps);
    float* slab = (float*)workspace;
    float* bslab = slab + (size_t)splits * Cout * K;
    Epilogue e = plain_epilogue(slab, K);
    e.split_stride = (long)Cout * K;
    e.colsum = dbias ? bslab : nullptr;
    e.colsum_stride = Cout;
    PlainSrc A = {dy, Cout, M, Cout, nullptr, 1};
    ConvSrc Bm = {x, B, H, W, Cin, OH, OW, ks, stride, pad, M, K};   // S(i = pixel m (reduction), j = (tap, ci))
    launch_gemm<false, false, 4>(A, Bm, e, Cout, K, M, splits, rps, st);
    CSWIN_LAUNCH_CHECK();
    long n = (long)Cout * K;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, slab, dw_perm, n, splits, n);
    if (dbias) hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(Cout, 256)), dim3(256), 0, st, bslab, dbias, (long)Cout, splits, (long)Cout);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

}  // extern "C"
