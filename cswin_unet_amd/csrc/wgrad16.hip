// bf16-operand weight gradient of nn.Linear for the bf16 matmul mode (BASELINE configs[2..4]), gfx950 / CDNA4.
//
//   dW[n][k] = sum_m (row_scale . dy)[m][n] * x[m][k]        dbias[n] = sum_m (row_scale . dy)[m][n]
//   (autograd backward of cswin_unet.py:169,177,23-27; dy (M, N) and x (M, K) fp32 in memory, both m-major)
//
// Both MFMA operands of this product run along the REDUCTION index m, while both matrices are stored with m as the slow
// index.  The tiled family (gemm.hip) transposes them while staging, with one 2-byte LDS store per element: with bf16
// MFMAs (16x the fp32 rate) that staging is the whole kernel (profiles/round1_gemm_bench_bf16_operands.txt: 3.07 ms per
// step against 3.30 ms in fp32).  Here the tiles are stored in LDS the way they arrive -- [m][128 columns] bf16 rows, one
// ds_write_b64 per 16-B global chunk -- and CDNA4's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a
// 4-row x 16-column block and each lane receives one COLUMN of it) delivers the m-contiguous fragments that
// v_mfma_f32_32x32x16_bf16 wants.  256-B rows are XOR-swizzled per 16-B chunk (guide T10, image (b)) so that the
// transposed reads are bank-conflict free.
//
// One workgroup = 4 waves = one 128 (n) x 128 (k) tile of dW over one slice of m; partial tiles go to the same
// [split][N*K + N] slabs as the fp32 path, reduced by cswin_rows_sum_multi.  fp32 accumulation throughout.
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int W16_T = 128;          // tile edge (n and k)
constexpr int W16_MS = 32;          // m rows per step
constexpr int W16_MAXP = 4;

struct W16Problem {
    const float* dy; const float* x; const float* row_scale;
    float* slab;                    // [splits][N*K + N]
    int M, N, K, rows_per_sample, rows_per_split, tiles_n, tiles_k, has_bias;
    int dy_bf16, x_bf16;            // the operand is stored as bf16 (bf16 activation storage)
    long slab_stride;
};
struct W16Batch {
    W16Problem p[W16_MAXP];
    int first[W16_MAXP + 1];
    int n;
    long long* stamps;              // debug (cswin_debug_set_stamps): [workgroup][8] s_memtime / realtime stamps of thread 0, or NULL
};

// byte offset of 16-B chunk `ch` (8 bf16) of row `row` inside a [rows][128] bf16 image
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

constexpr int W16_LDS = 2 * 2 * W16_MS * 256 + 8 * W16_T * 4;      // 2 stages x (dy | x) images + bias sums

// One problem's tile.  DY16 / X16: that operand is STORED as bf16 -- compile-time copies of the body (chosen per workgroup in
// the kernel below): a run-time choice between 8-B and 16-B loads inside fetch() makes the prefetch loads wait for one another.
template <bool DY16, bool X16>
__device__ __forceinline__ void wgrad16_tile(const W16Problem& P, const int lb, unsigned char* lds, long long* st) {
    const int tiles = P.tiles_n * P.tiles_k;
    const int split = lb / tiles, tile = lb - split * tiles;
    const int nb = (tile / P.tiles_k) * W16_T, kb = (tile % P.tiles_k) * W16_T;
    const int m_begin = split * P.rows_per_split;
    const int m_end = min(P.M, m_begin + P.rows_per_split);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;                    // wave tile: n in [64 wi, +64), k in [64 wj, +64)
    const int li = lane & 31, lh = lane >> 5;

    // loader: thread -> 16-B fp32 chunk c (4 columns) of rows lr + 8 j, j = 0..3, of both tiles
    const int lc = tid & 31, lr = tid >> 5;
    const bool n_ok = nb + 4 * lc < P.N, k_ok = kb + 4 * lc < P.K;
    const float* dyp = P.dy + nb + 4 * lc;
    const float* xp = P.x + kb + 4 * lc;
    f32x4 gdy[4], gx[4];
    u32x2 gdyh[4], gxh[4];          // raw chunks of a bf16-stored operand (INTEGER vectors: see widen_bf16x4 in gemm_epilogue.h)
    float grs[4];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + lr + 8 * j;
            const bool ok = m < m_end;
            gdy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            gx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            gdyh[j] = u32x2{0u, 0u};
            gxh[j] = u32x2{0u, 0u};
            // bf16-stored operands arrive as raw bits in the two low lanes and are widened (dy) or passed through (x) in stash():
            // a conversion here would make every prefetch wait for its own data
            if (ok && n_ok) {
                if constexpr (DY16) {
                    gdyh[j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(P.dy) + (long)m * P.N + nb + 4 * lc);
                } else {
                    gdy[j] = *reinterpret_cast<const f32x4*>(dyp + (long)m * P.N);
                }
            }
            if (ok && k_ok) {
                if constexpr (X16) {
                    gxh[j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(P.x) + (long)m * P.K + kb + 4 * lc);
                } else {
                    gx[j] = *reinterpret_cast<const f32x4*>(xp + (long)m * P.K);
                }
            }
            grs[j] = (P.row_scale && ok) ? P.row_scale[m / P.rows_per_sample] : 1.0f;
        }
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto stash = [&](int stage) {
        unsigned char* dimg = lds + stage * (2 * W16_MS * 256);
        unsigned char* ximg = dimg + W16_MS * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = lr + 8 * j;
            f32x4 d = gdy[j];
            if constexpr (DY16)
                d = f32x4{__builtin_bit_cast(float, gdyh[j][0] << 16), __builtin_bit_cast(float, gdyh[j][0] & 0xffff0000u),
                          __builtin_bit_cast(float, gdyh[j][1] << 16), __builtin_bit_cast(float, gdyh[j][1] & 0xffff0000u)};
            d *= grs[j];
            bsum += d;
            const int off = img_off(r, lc >> 1) + 8 * (lc & 1);
            *reinterpret_cast<bf16x4v*>(dimg + off) = __builtin_convertvector(d, bf16x4v);
            if constexpr (X16) *reinterpret_cast<u32x2*>(ximg + off) = gxh[j];
            else *reinterpret_cast<bf16x4v*>(ximg + off) = __builtin_convertvector(gx[j], bf16x4v);
        }
    };

    // transposed-read addresses: lane (q, p) of its 16-lane group supplies row r0 + q, columns c0 + 4 p .. +3 of the block;
    // it receives column c0 + (lane & 15) of rows r0 .. r0 + 3.  Fragment of a 32-column block cb (columns cb .. cb + 31):
    // group (lane >> 4) & 1 takes columns cb + 16 * that, lane half lh takes rows 8 lh + 4 t (t = 0, 1) of the k16 step.
    const int q = (lane & 15) >> 2, pp = lane & 3;
    auto tr_frag = [&](const unsigned char* img, int cb, int r16) -> bf16x8v {
        const int col = cb + 16 * ((lane >> 4) & 1) + 4 * pp;
        s16x4 v0, v1;
        {
            const int row = r16 + 8 * lh + q;
            v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(row, col >> 3) + 2 * (col & 7)));
            v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(row + 4, col >> 3) + 2 * (col & 7)));
        }
        s16x8 f = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8v, f);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

    if (m_begin < m_end) {
        fetch(m_begin);
        stash(0);
        __syncthreads();
        if (st && threadIdx.x == 0) st[1] = __builtin_amdgcn_s_memtime();
        int stage = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += W16_MS) {
            const bool more = m0 + W16_MS < m_end;
            if (more) fetch(m0 + W16_MS);
            const unsigned char* dimg = lds + stage * (2 * W16_MS * 256);
            const unsigned char* ximg = dimg + W16_MS * 256;
#pragma unroll
            for (int r16 = 0; r16 < W16_MS; r16 += 16) {
                bf16x8v af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = tr_frag(dimg, 64 * wi + 32 * a, r16);
#pragma unroll
                for (int c = 0; c < 2; ++c) bf[c] = tr_frag(ximg, 64 * wj + 32 * c, r16);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[c], acc[a][c], 0, 0, 0);
            }
            if (more) stash(stage ^ 1);          // the other stage was last read one step ago, before the barrier below
            __syncthreads();
            stage ^= 1;
        }
    }

    if (st && threadIdx.x == 0) st[2] = __builtin_amdgcn_s_memtime();
    // ---- partial tile -> slab (C/D layout: lane = column k, registers = rows n) ----
    float* slab = P.slab + (long)split * P.slab_stride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = kb + 64 * wj + 32 * c + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = nb + 64 * wi + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (n < P.N && k < P.K) slab[(long)n * P.K + k] = acc[a][c][g];
            }
        }
    // ---- bias gradient partial: column sums of the dy tile (only the k-tile 0 workgroups own them) ----
    if (P.has_bias && kb == 0) {
        float* red = reinterpret_cast<float*>(lds + 2 * 2 * W16_MS * 256);      // [8][128]
        *reinterpret_cast<f32x4*>(&red[lr * W16_T + 4 * lc]) = bsum;
        __syncthreads();
        if (tid < W16_T && nb + tid < P.N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += red[r * W16_T + tid];
            slab[(long)P.N * P.K + nb + tid] = s;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad16_kernel(W16Batch b) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[W16_LDS];
    int pi = 0;
    while (pi + 1 < b.n && (int)blockIdx.x >= b.first[pi + 1]) ++pi;
    const W16Problem& P = b.p[pi];
    const int lb = (int)blockIdx.x - b.first[pi];
    long long* st = b.stamps ? b.stamps + 8L * blockIdx.x : nullptr;
    if (st && threadIdx.x == 0) { st[0] = __builtin_amdgcn_s_memtime(); st[4] = __builtin_amdgcn_s_getreg(6164); st[5] = __builtin_amdgcn_s_memrealtime(); }
    if (P.dy_bf16) {
        if (P.x_bf16) wgrad16_tile<true, true>(P, lb, lds, st);
        else wgrad16_tile<true, false>(P, lb, lds, st);
    } else {
        if (P.x_bf16) wgrad16_tile<false, true>(P, lb, lds, st);
        else wgrad16_tile<false, false>(P, lb, lds, st);
    }
    if (st && threadIdx.x == 0) { st[3] = __builtin_amdgcn_s_memtime(); st[6] = __builtin_amdgcn_s_memrealtime(); }
}

}  // namespace

// Internal entry used by gemm.hip's cswin_linear_bwd_weight_batch in bf16 matmul mode.  Problems must be 16-B aligned with
// N % 4 == K % 4 == 0 (the caller checks).  splits[i] <= what problem i's workspace holds.  Returns 0.
int cswin_wgrad16_batch(const cswin_wgrad_desc* d, int n, const int* splits, const int* rows_per_split, void* stream, long long* stamps) {
    W16Batch b = {};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        W16Problem& P = b.p[i];
        P.dy = d[i].dy; P.x = d[i].x; P.row_scale = d[i].row_scale;
        P.slab = (float*)d[i].workspace;
        P.M = d[i].M; P.N = d[i].N; P.K = d[i].K;
        P.rows_per_sample = d[i].row_scale ? d[i].rows_per_sample : 1;
        P.rows_per_split = rows_per_split[i];
        P.tiles_n = cdiv(P.N, W16_T); P.tiles_k = cdiv(P.K, W16_T);
        P.has_bias = d[i].dbias != nullptr;
        P.dy_bf16 = d[i].io_bf16 & 1;
        P.x_bf16 = (d[i].io_bf16 >> 1) & 1;
        P.slab_stride = (long)P.N * P.K + P.N;
        b.first[i] = blocks;
        blocks += P.tiles_n * P.tiles_k * splits[i];
    }
    b.first[n] = blocks;
    b.n = n;
    b.stamps = stamps;
    hipLaunchKernelGGL(wgrad16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    return 0;
}
