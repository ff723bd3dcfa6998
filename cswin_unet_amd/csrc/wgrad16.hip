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
    int dma;                        // both stored as bf16, M % 32 == 0, N % 8 == K % 8 == 0: LDS-DMA tile (below)
    long slab_stride;
};
struct W16Batch {
    W16Problem p[W16_MAXP];
    int first[W16_MAXP + 1];
    int n;
    long long* stamps;              // debug (cswin_debug_set_stamps): [workgroup][8] s_memtime / realtime stamps of thread 0, or NULL
    ReduceRiders r;                 // pending slab reductions of earlier launches: the grid's last workgroups (common.h)
};

// byte offset of 16-B chunk `ch` (8 bf16) of row `row` inside a [rows][128] bf16 image
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

constexpr int W16_DMA_S = 3;                                        // LDS-DMA tile: steps in flight
constexpr int W16_STEP = 2 * W16_MS * 256;                          // bytes of one step's (dy | x) images
constexpr int W16_SC_SAMPLES = 64;                                  // DropPath factors of the samples a workgroup's rows touch
constexpr int W16_SC_BYTES = (W16_SC_SAMPLES + W16_DMA_S * W16_MS) * 4;
constexpr int W16_LDS = W16_DMA_S * W16_STEP + W16_SC_BYTES;        // 48.6 KB (register path: 2 stages + bias sums = 36 KB): 3 workgroups per CU
static_assert(W16_LDS >= 2 * W16_STEP + 8 * W16_T * 4, "the register path's stages and bias sums must fit");

// Partial 64 x 64 wave tile -> slab.  The 32x32 accumulators hold a COLUMN per lane (lane = k, registers = rows n): stored from
// that layout a wave issues 64 dword stores.  Each fragment goes through a private [32][36] LDS patch instead (as the GEMM
// epilogue does) so that a lane owns 4 consecutive k of one row: 16 stores of 16 B.  Call after a workgroup barrier (the patches
// overlay the operand images); K % 4 == 0 and 16-B aligned slabs are the caller's contract.
constexpr int W16_PATCH = 32 * 36;
__device__ __forceinline__ void w16_store_tile(f32x16 (&acc)[2][2], float* slab, int N, int K, int n0, int k0, int lane, float* patch) {
    const int li = lane & 31, lh = lane >> 5, rrow = lane >> 3, rcol = (lane & 7) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int g = 0; g < 16; ++g) patch[((g & 3) + 8 * (g >> 2) + 4 * lh) * 36 + li] = acc[a][c][g];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int k = k0 + 32 * c + rcol;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 32 * a + rrow + 8 * q;
                if (n < N && k < K) *reinterpret_cast<f32x4*>(slab + (long)n * K + k) = *reinterpret_cast<const f32x4*>(&patch[(rrow + 8 * q) * 36 + rcol]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
}

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
    // ---- partial tile -> slab ----
    float* slab = P.slab + (long)split * P.slab_stride;
    __syncthreads();                        // every wave is done with the operand images: they become the store patches
    w16_store_tile(acc, slab, P.N, P.K, nb + 64 * wi, kb + 64 * wj, lane, reinterpret_cast<float*>(lds) + wave * W16_PATCH);
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

// ---- LDS-DMA tile: both operands stored as bf16 and used as they are (no row scale) ----------------------------------------
// The register path above spends ~3 k cycles per 32-row step on 256 cycles of MFMA work (tools/wgrad16_stamps.py): one
// global-load round trip per step, its prefetch reaching one step ahead, and a second register set costs a workgroup per CU.
// Here the [m][128] images are written by global_load_lds_dwordx4 (no registers: three steps in flight at the same occupancy),
// the XOR swizzle of img_off applied to the SOURCE chunk, one counted s_waitcnt vmcnt + s_barrier per step.  The bias gradient
// (column sums of dy) is one more MFMA per fragment against a constant ones operand instead of a pass over the tile.
// SCALE (DropPath row factors on dy): the factors of the samples this workgroup's rows touch are read once, a 32-entry per-row
// table is refreshed per step in LDS, and the dy fragments are scaled as they leave LDS (widen, multiply, round: the same
// arithmetic, element by element, as the register path's scale-then-round).
__device__ __forceinline__ void w16_dma(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void w16_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void w16_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool SCALE>
__device__ __forceinline__ void wgrad16_dma_tile(const W16Problem& P, const int lb, unsigned char* lds, long long* st) {
    const int tiles = P.tiles_n * P.tiles_k;
    const int split = lb / tiles, tile = lb - split * tiles;
    const int nb = (tile / P.tiles_k) * W16_T, kb = (tile % P.tiles_k) * W16_T;
    const int m_begin = split * P.rows_per_split;
    const int m_end = min(P.M, m_begin + P.rows_per_split);
    const int nsteps = (m_end - m_begin) / W16_MS;              // whole steps: M and the split size are multiples of 32

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;                    // wave tile: n in [64 wi, +64), k in [64 wj, +64)
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // DMA: a wave instruction moves 4 image rows (1 KB); wave w moves row groups 2 w, 2 w + 1 of both images of a step
    const int drow = lane >> 4, slot = lane & 15;
    const unsigned char* dy_src[2];
    const unsigned char* x_src[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int row = (2 * wave + g) * 4 + drow;
        const int ch = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));                    // source chunk of this lane's slot (img_off)
        const int cn = min(ch, (P.N - nb) / 8 - 1), ck = min(ch, (P.K - kb) / 8 - 1);   // columns beyond N / K: in-range data, never stored
        dy_src[g] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const __bf16*>(P.dy) + (long)(m_begin + row) * P.N + nb + 8 * cn);
        x_src[g] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const __bf16*>(P.x) + (long)(m_begin + row) * P.K + kb + 8 * ck);
    }
    auto issue = [&](int step) {
        const unsigned stg = lds0 + (unsigned)(step % W16_DMA_S) * W16_STEP + (unsigned)wave * 2048;
#pragma unroll
        for (int g = 0; g < 2; ++g) w16_dma(dy_src[g] + (long)step * W16_MS * P.N * 2, stg + g * 1024);
#pragma unroll
        for (int g = 0; g < 2; ++g) w16_dma(x_src[g] + (long)step * W16_MS * P.K * 2, stg + W16_MS * 256 + g * 1024);
    };

    const int q = (lane & 15) >> 2, pp = lane & 3;
    auto tr_frag = [&](const unsigned char* img, int cb, int r16) -> bf16x8v {
        const int col = cb + 16 * ((lane >> 4) & 1) + 4 * pp;
        const int row = r16 + 8 * lh + q;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(row, col >> 3) + 2 * (col & 7)));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(row + 4, col >> 3) + 2 * (col & 7)));
        const s16x8 f = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8v, f);
    };

    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[a][e] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
    }
    const bool do_bias = P.has_bias && kb == 0 && wj == 0;      // wave-uniform
    // DropPath factors: samples b0 .. of this workgroup's rows (read BEFORE the first DMA: older in the vmcnt queue)
    float* sc_tab = reinterpret_cast<float*>(lds + W16_DMA_S * W16_STEP);
    float* sc_row = sc_tab + W16_SC_SAMPLES;                     // [stage][32]
    const int b0 = SCALE ? m_begin / P.rows_per_sample : 0;
    if constexpr (SCALE) {
        const int nb_s = (m_end - 1) / P.rows_per_sample - b0 + 1;
        if (tid < nb_s) sc_tab[tid] = P.row_scale[b0 + tid];
        __syncthreads();
    }
    auto fill_rows = [&](int step) {                            // per-row factors of a step (LDS only)
        if (tid < W16_MS) sc_row[(step % W16_DMA_S) * W16_MS + tid] = sc_tab[(m_begin + step * W16_MS + tid) / P.rows_per_sample - b0];
    };
    const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};      // bf16 1.0
    const bf16x8v ones = __builtin_bit_cast(bf16x8v, ones_s);

    for (int s = 0; s < W16_DMA_S - 1 && s < nsteps; ++s) {
        issue(s);
        if constexpr (SCALE) fill_rows(s);
    }
    for (int step = 0; step < nsteps; ++step) {
        const int later = min(W16_DMA_S - 2, nsteps - 1 - step);
        if (later >= 1) w16_wait<4>();
        else w16_wait<0>();
        w16_barrier();                      // the step is in LDS for every wave; the stage read one step ago is free again
        if (step + W16_DMA_S - 1 < nsteps) {
            issue(step + W16_DMA_S - 1);
            if constexpr (SCALE) fill_rows(step + W16_DMA_S - 1);
        }
        if (st && threadIdx.x == 0 && step == 0) st[1] = __builtin_amdgcn_s_memtime();
        const unsigned char* dimg = lds + (step % W16_DMA_S) * W16_STEP;
        const unsigned char* ximg = dimg + W16_MS * 256;
#pragma unroll
        for (int r16 = 0; r16 < W16_MS; r16 += 16) {
            bf16x8v af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = tr_frag(dimg, 64 * wi + 32 * a, r16);
#pragma unroll
            for (int c = 0; c < 2; ++c) bf[c] = tr_frag(ximg, 64 * wj + 32 * c, r16);
            if constexpr (SCALE) {
                // a lane's 8 values are rows r16 + 8 lh + 0 .. 7 of its column
                const float* sr = &sc_row[(step % W16_DMA_S) * W16_MS + r16 + 8 * lh];
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sr), s1 = *reinterpret_cast<const f32x4*>(sr + 4);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const u32x4 raw = __builtin_bit_cast(u32x4, af[a]);
                    f32x4 lo = {__builtin_bit_cast(float, raw[0] << 16), __builtin_bit_cast(float, raw[0] & 0xffff0000u),
                                __builtin_bit_cast(float, raw[1] << 16), __builtin_bit_cast(float, raw[1] & 0xffff0000u)};
                    f32x4 hi = {__builtin_bit_cast(float, raw[2] << 16), __builtin_bit_cast(float, raw[2] & 0xffff0000u),
                                __builtin_bit_cast(float, raw[3] << 16), __builtin_bit_cast(float, raw[3] & 0xffff0000u)};
                    af[a] = __builtin_shufflevector(__builtin_convertvector(lo * s0, bf16x4v), __builtin_convertvector(hi * s1, bf16x4v), 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[c], acc[a][c], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int a = 0; a < 2; ++a) accb[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], ones, accb[a], 0, 0, 0);
            }
        }
    }
    if (st && threadIdx.x == 0) st[2] = __builtin_amdgcn_s_memtime();
    float* slab = P.slab + (long)split * P.slab_stride;
    w16_barrier();                          // every wave is done with the operand images: they become the store patches
    w16_store_tile(acc, slab, P.N, P.K, nb + 64 * wi, kb + 64 * wj, lane, reinterpret_cast<float*>(lds) + wave * W16_PATCH);
    if (do_bias && li == 0) {               // every column of the ones product holds the row sums: take column 0
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = nb + 64 * wi + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (n < P.N) slab[(long)P.N * P.K + n] = accb[a][g];
            }
    }
}

__global__ __launch_bounds__(256, 3) void wgrad16_kernel(W16Batch b) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[W16_LDS];
    if ((int)blockIdx.x >= b.first[b.n]) {
        static_assert(W16_LDS >= (int)sizeof(float) * RS_G * (RS_COLS + 1), "reduction scratch");
        rows_sum_dispatch(b.r.j, b.r.first_block, b.r.njobs, (int)blockIdx.x - b.first[b.n], reinterpret_cast<float(*)[RS_COLS + 1]>(lds));
        return;
    }
    int pi = 0;
    while (pi + 1 < b.n && (int)blockIdx.x >= b.first[pi + 1]) ++pi;
    const W16Problem& P = b.p[pi];
    const int lb = (int)blockIdx.x - b.first[pi];
    long long* st = b.stamps ? b.stamps + 8L * blockIdx.x : nullptr;
    if (st && threadIdx.x == 0) { st[0] = __builtin_amdgcn_s_memtime(); st[4] = __builtin_amdgcn_s_getreg(6164); st[5] = __builtin_amdgcn_s_memrealtime(); }
    if (P.dma) {
        if (P.row_scale) wgrad16_dma_tile<true>(P, lb, lds, st);
        else wgrad16_dma_tile<false>(P, lb, lds, st);
    } else if (P.dy_bf16) {
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
// N % 4 == K % 4 == 0 (the caller checks).  splits[i] <= what problem i's workspace holds.  pending[0..npending): reductions of
// earlier launches that ride as this grid's last workgroups.  Returns 0 (1: a bad pending job).
int cswin_wgrad16_batch(const cswin_wgrad_desc* d, int n, const int* splits, const int* rows_per_split, const cswin_reduce_job* pending,
                        int npending, void* stream, long long* stamps) {
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
        const bool dma_off = !cswin_tuning().w16_dma;                                                      // tuning aid
        P.dma = !dma_off && P.dy_bf16 && P.x_bf16 && P.M % W16_MS == 0 && P.rows_per_split % W16_MS == 0 &&
                P.N % 8 == 0 && P.K % 8 == 0 && ((((uintptr_t)P.dy) | ((uintptr_t)P.x)) & 15) == 0 &&
                (!P.row_scale || P.rows_per_split / P.rows_per_sample + 2 <= W16_SC_SAMPLES);
        b.first[i] = blocks;
        blocks += P.tiles_n * P.tiles_k * splits[i];
    }
    b.first[n] = blocks;
    b.n = n;
    b.stamps = stamps;
    int rblocks = 0;
    if (npending > 0) {
        rblocks = fill_reduce_table(pending, npending, b.r.j, b.r.first_block);
        if (rblocks < 0) return 1;
        b.r.njobs = npending;
    }
    static_assert(sizeof(W16Batch) <= 4096, "kernel argument block");
    hipLaunchKernelGGL(wgrad16_kernel, dim3(blocks + rblocks), dim3(256), 0, (hipStream_t)stream, b);
    return 0;
}
