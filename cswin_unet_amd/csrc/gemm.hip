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
#include <stdlib.h>

#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"

// both operands stored as bf16 (gemm16.hip); 0 = launched, 1 = not covered
int cswin_gemm16(int mode, int epi_mode, const void* A, const void* B, const void* epilogue, int M, int NO, int R, void* stream);
// wgrad16.hip: bf16-operand weight gradients with transposing LDS reads (bf16 matmul mode)
int cswin_wgrad16_batch(const cswin_wgrad_desc* d, int n, const int* splits, const int* rows_per_split, const cswin_reduce_job* pending,
                        int npending, void* stream, long long* stamps);

namespace {

constexpr int BKMAX = 64;     // largest reduction tile (split sizes are multiples of it)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// matmul precision is an ARGUMENT of every entry point (no library state): 0 = exact fp32 MFMA (the parity path);
// 1 = operands rounded to bf16 while they are staged into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulation and fp32
// storage everywhere (BASELINE configs 3-5 name bf16)
#define CSWIN_CHECK_PRECISION(p, who) CSWIN_REQUIRE((p) == 0 || (p) == 1, CSWIN_ERR_UNSUPPORTED, who ": precision %d (0 = fp32, 1 = bf16 operands)", (p))

// ------------------------------------------------------------------------------------
// operand sources: a logical matrix S(i, j) whose fast (contiguous) index is j
// ------------------------------------------------------------------------------------
struct PlainSrc {
    const float* p; long ld; int rows, cols;
    const float* row_scale; int rows_per_sample;     // optional per-row multiplier (DropPath backward)
    int bf16;                                        // the matrix is stored as bf16 (p is then only an element-indexed handle)
    struct Row { const float* base; float s; };
    __device__ Row row(int i) const {
        Row r;
        r.base = (i < rows) ? p + (long)i * ld : nullptr;
        r.s = (row_scale && i < rows) ? row_scale[i / rows_per_sample] : 1.0f;
        return r;
    }
    __device__ const float* ptr(const Row& r, int j) const { return (r.base && j < cols) ? r.base + j : nullptr; }
};

// bf16 STORAGE of an operand (activations of the bf16 mode): only plain matrices; every source computes addresses in elements,
// so the element offset of the fp32-typed pointer is re-applied to the bf16 base.
template <class S> __device__ __forceinline__ bool src_is_bf16(const S&) { return false; }
__device__ __forceinline__ bool src_is_bf16(const PlainSrc& s) { return s.bf16 != 0; }
// 4 bf16 as two raw dwords (an INTEGER vector: see widen_bf16x4); they go to the LDS image as they are (stash below)
template <class S> __device__ __forceinline__ u32x2 src_load4_bf16(const S&, const float*) { return u32x2{0u, 0u}; }
__device__ __forceinline__ u32x2 src_load4_bf16(const PlainSrc& s, const float* p) {
    return *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(s.p) + (p - s.p));
}

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

// Stride-2 3x3 pad-1 data gradient, one input-pixel parity class (py, px) at a time: an input pixel only ever meets the
// taps with ky = (iy + 1) mod 2 (+2), kx likewise, i.e. 1, 2, 2 or 4 of the 9 taps depending on its class.  Gathering
// over all 9 taps (ConvTSrc) feeds 75 % structural zeros to the MFMAs; per class there are none.
// i = class-local pixel (b, jy, jx) <-> (iy, ix) = (2 jy + py, 2 jx + px);  j = tap_local * C + co.
struct ConvTS2Src {
    const float* dy; int B, H, W, C, OH, OW, py, px, H2, W2, kys, kxs; int rows, cols;   // kys/kxs: 2 bits per local tap
    struct Row { int b, iy, ix; float s; };
    __device__ Row row(int i) const {
        Row r;
        r.s = 1.0f;
        if (i >= rows) { r.b = -1; r.iy = r.ix = 0; return r; }
        const int hw = H2 * W2;
        r.b = i / hw;
        const int rem = i - r.b * hw;
        const int jy = rem / W2;
        r.iy = 2 * jy + py;
        r.ix = 2 * (rem - jy * W2) + px;
        return r;
    }
    __device__ const float* ptr(const Row& r, int j) const {
        if (r.b < 0 || j >= cols) return nullptr;
        const int t = j / C, co = j - t * C;
        const int oy = (r.iy + 1 - ((kys >> (2 * t)) & 3)) >> 1, ox = (r.ix + 1 - ((kxs >> (2 * t)) & 3)) >> 1;
        if (oy >= OH || ox >= OW) return nullptr;
        return dy + ((long)(r.b * OH + oy) * OW + ox) * C + co;
    }
};

// rows (tap_local, co) of the [9][Cout][Cin] weight image that belong to one parity class
struct TapRowsSrc {
    const float* w; int Cout, Cin, taps; int rows, cols;      // taps: 4 bits per local tap (index into the 9 taps)
    struct Row { const float* base; float s; };
    __device__ Row row(int i) const {
        Row r;
        r.s = 1.0f;
        if (i >= rows) { r.base = nullptr; return r; }
        const int t = i / Cout, co = i - t * Cout;
        r.base = w + ((long)((taps >> (4 * t)) & 15) * Cout + co) * Cin;
        return r;
    }
    __device__ const float* ptr(const Row& r, int j) const { return (r.base && j < cols) ? r.base + j : nullptr; }
};

// ------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------
// SCALE_A: multiply the A operand rows by their Row::s (DropPath factor) when staging (weight gradients only)
// KW: wave groups that split each reduction tile between them (workgroup = 4*KW waves; group g multiplies the k-slice
// [g*BK/KW, (g+1)*BK/KW) of every tile, the groups' accumulators are summed through LDS at the end).  When M*N is too
// small to give every SIMD >= 2 independent MFMA chains (stage-3/4 shapes with long K) this doubles / quadruples the
// resident waves per workgroup without a global split-K reduction.
// floats of LDS one tile configuration needs (operand images, or the waves' epilogue patches if those are larger).  The kernels own
// the buffer and hand it to gemm_body, so that a launch which mixes two configurations (gemm_block_tail_kernel) holds ONE buffer
// of the larger size instead of two static arrays.
template <int BM, int BN, int BK, int KW, bool A_RC, bool B_RC>
constexpr int gemm_lds_floats() {
    constexpr int NW = (BM >= 64 ? 2 : 1) * (BN >= 64 ? 2 : 1);
    constexpr int LDR = BK + 4;
    constexpr int A_ELEMS = A_RC ? BM * LDR : BK * (BM + 4);
    constexpr int B_ELEMS = B_RC ? BN * LDR : BK * (BN + 4);
    return (A_ELEMS + B_ELEMS) > NW * EP_WAVE_FLOATS ? (A_ELEMS + B_ELEMS) : NW * EP_WAVE_FLOATS;
}

template <int BM, int BN, int BK, int KW, bool A_RC, bool B_RC, int VEC, int EPI, bool SCALE_A, int PREC, class ASrc, class BSrc>
__device__ __forceinline__ void gemm_body(float* lds, const ASrc& A, const BSrc& B, const Epilogue& epi, int M, int N, int R,
                                          int r_per_split, int tiles_m, int tiles_n, int bid, int nblk, int stamp_row) {
    // waves of one k-group: 2 x 2 for 64 x 64 (and larger) tiles, 2 x 1 for 64 x 32: the narrow tile
    // exists for launches whose 64 x 64 tile count is a poor multiple of the 256 CUs (every workgroup is resident at once,
    // so the kernel ends with the most loaded CU)
    constexpr int WGM = BM >= 64 ? 2 : 1, WGN = BN >= 64 ? 2 : 1, NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 32, FN = WN / 32;        // 32x32 fragments per wave
    constexpr int LDR = BK + 4;                      // [row][r] image: 16-B padded rows -> conflict-free b128 reads
    constexpr int LDA = A_RC ? LDR : BM + 4;
    constexpr int LDB = B_RC ? LDR : BN + 4;
    constexpr int A_ELEMS = A_RC ? BM * LDR : BK * (BM + 4);
    constexpr int B_ELEMS = B_RC ? BN * LDR : BK * (BN + 4);
    constexpr int LDS_FLOATS = gemm_lds_floats<BM, BN, BK, KW, A_RC, B_RC>();
    static_assert(LDS_FLOATS >= A_ELEMS + B_ELEMS && LDS_FLOATS >= NW * EP_WAVE_FLOATS, "gemm_lds_floats out of step with the tile geometry");
    float* As = lds;
    float* Bs = lds + A_ELEMS;
    // PREC == 1: both operands as [row][r] bf16 images (r-contiguous: one ds_read_b128 = the 8 k-values of a lane)
    constexpr int LD16 = BK + 8;                     // in bf16 elements: 16-B aligned rows, 36-dword stride at BK = 64
    unsigned short* As16 = reinterpret_cast<unsigned short*>(lds);
    unsigned short* Bs16 = As16 + BM * LD16;
    static_assert(PREC == 0 || ((BM + BN) * LD16 * 2 <= LDS_FLOATS * 4 && (BK / KW) % 16 == 0), "bf16 tile does not fit");

    constexpr int NTH = 64 * NW * KW;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % NW, kg = tid / (64 * NW);
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so give every XCD a
    // CONTIGUOUS range of logical blocks, ordered [split][m-tile][n-tile]: the blocks that re-read one A row panel
    // (all n-tiles of an m-tile; all tiles of a split) then share one L2 instead of fetching it eight times.
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int lb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tiles = tiles_m * tiles_n;
    const int split = lb / tiles;
    const int tile = lb - split * tiles;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int r_begin = split * r_per_split;
    const int r_end = min(R, r_begin + r_per_split);
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;

    // loader geometry
    constexpr int QA = BM * BK / (4 * NTH), QB = BN * BK / (4 * NTH); // chunks (16 B) per thread per tile
    static_assert(QA >= 1 && QB >= 1 && (BK / KW) % 8 == 0, "tile too small for this many wave groups");
    constexpr int A_CPR = A_RC ? BK / 4 : BM / 4;           // chunks per LDS row
    constexpr int B_CPR = B_RC ? BK / 4 : BN / 4;
    constexpr int A_RPP = NTH / A_CPR, B_RPP = NTH / B_CPR; // rows per pass
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

    // register prefetch of the next tile: raw loads only -- nothing may consume pa/pb before stash(), or the
    // compiler waits for the loads in front of the MFMA section
    f32x4 pa[QA], pb[QB];
    u32x2 pa16[QA], pb16[QB];                        // an operand stored as bf16: its raw chunks (pa / pb is then unused)
    float pas[QA];
    // a16 (std::true_type / false_type): A is STORED as bf16.  A compile-time copy of the main loop per storage type: a run-time
    // choice between the 8-B and the 16-B load inside fetch() splits it into basic blocks and the loads end up waiting for one another.
    auto fetch = [&](int r0, auto a16, auto b16) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            typename ASrc::Row rr;
            int j;
            if (A_RC) {
                rr = arow[q];
                j = r0 + 4 * a_c;
            } else {
                const int r = r0 + a_r + q * A_RPP;
                rr = A.row(r < r_end ? r : 0x7fffffff);
                j = m0 + 4 * a_c;
            }
            if (SCALE_A) pas[q] = rr.s;
            if (VEC == 4) {
                const float* p = (!A_RC || j < r_end) ? A.ptr(rr, j) : nullptr;
                if constexpr (decltype(a16)::value) {
                    pa16[q] = u32x2{0u, 0u};
                    if (p) pa16[q] = src_load4_bf16(A, p);
                } else {
                    pa[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p) pa[q] = *reinterpret_cast<const f32x4*>(p);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* p = (!A_RC || j + e < r_end) ? A.ptr(rr, j + e) : nullptr;
                    pa[q][e] = p ? *p : 0.f;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            typename BSrc::Row rr;
            int j;
            if (B_RC) {
                rr = brow[q];
                j = r0 + 4 * b_c;
            } else {
                const int r = r0 + b_r + q * B_RPP;
                rr = B.row(r < r_end ? r : 0x7fffffff);
                j = n0 + 4 * b_c;
            }
            if (VEC == 4) {
                const float* p = (!B_RC || j < r_end) ? B.ptr(rr, j) : nullptr;
                if constexpr (decltype(b16)::value) {
                    pb16[q] = u32x2{0u, 0u};
                    if (p) pb16[q] = src_load4_bf16(B, p);
                } else {
                    pb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p) pb[q] = *reinterpret_cast<const f32x4*>(p);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* p = (!B_RC || j + e < r_end) ? B.ptr(rr, j + e) : nullptr;
                    pb[q][e] = p ? *p : 0.f;
                }
            }
        }
    };
    auto stash = [&](auto a16, auto b16) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!decltype(a16)::value) v = pa[q];
            if constexpr (PREC == 1) {
                // the 4 bf16 of this chunk as two packed dwords (plain integers: hipcc 7.2 miscompiles element extraction from a
                // bf16 vector that is merged from two branches -- all four stores received element 0)
                unsigned h01, h23;
                if constexpr (decltype(a16)::value) {    // stored as bf16: the bits go to the image unchanged
                    h01 = pa16[q][0];
                    h23 = pa16[q][1];
                    if (SCALE_A) {
                        const u32x2 u = __builtin_bit_cast(u32x2, __builtin_convertvector(widen_bf16x4(pa16[q]) * pas[q], bf16x4));
                        h01 = u[0]; h23 = u[1];
                    }
                } else {
                    if (SCALE_A) v *= pas[q];
                    const u32x2 u = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
                    h01 = u[0]; h23 = u[1];
                }
                if (A_RC) {
                    *reinterpret_cast<u32x2*>(&As16[(a_r + q * A_RPP) * LD16 + 4 * a_c]) = u32x2{h01, h23};
                } else {                                 // source is row(m)-contiguous: transpose into the [m][r] image
                    unsigned short* col = &As16[4 * a_c * LD16 + a_r + q * A_RPP];
                    col[0] = (unsigned short)(h01 & 0xffffu);
                    col[LD16] = (unsigned short)(h01 >> 16);
                    col[2 * LD16] = (unsigned short)(h23 & 0xffffu);
                    col[3 * LD16] = (unsigned short)(h23 >> 16);
                }
            } else {
                if (SCALE_A) v *= pas[q];
                *reinterpret_cast<f32x4*>(&As[(a_r + q * A_RPP) * LDA + 4 * a_c]) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if constexpr (PREC == 1) {
                u32x2 hb;                                // 4 bf16 as two dwords (stored bf16: unchanged bits)
                if constexpr (decltype(b16)::value) hb = pb16[q];
                else hb = __builtin_bit_cast(u32x2, __builtin_convertvector(pb[q], bf16x4));
                if (B_RC) {
                    *reinterpret_cast<u32x2*>(&Bs16[(b_r + q * B_RPP) * LD16 + 4 * b_c]) = hb;
                } else {
                    // source rows run along the REDUCTION index (data gradient: W[r][k]): keep them as they arrive, [r][BN] bf16
                    // rows of 128 B, and let ds_read_b64_tr_b16 do the transpose when the fragments are read (below).  16-B
                    // chunks are XOR-swizzled with bit 1 of the row so that the 4-row x 16-column transposed reads of a
                    // 32-lane half fall into 64 distinct banks.
                    static_assert(B_RC || BN == 64, "transposed-read B image is laid out for 64-column tiles");
                    const int row = b_r + q * B_RPP;
                    *reinterpret_cast<u32x2*>(&Bs16[row * BN + 8 * ((b_c >> 1) ^ (((row >> 1) & 1) << 2)) + 4 * (b_c & 1)]) = hb;
                }
            } else {
                *reinterpret_cast<f32x4*>(&Bs[(b_r + q * B_RPP) * LDB + 4 * b_c]) = pb[q];
            }
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float csum = 0.f;   // dbias partial (TN, A row-contiguous image: column tid of the A tile)
    const bool do_colsum = !A_RC && epi.colsum && n0 == 0 && tid < BM;

    if (epi.stamps && tid == 0) { epi.stamps[8L * stamp_row + 0] = __builtin_amdgcn_s_memtime(); epi.stamps[8L * stamp_row + 4] = __builtin_amdgcn_s_getreg(6164); epi.stamps[8L * stamp_row + 5] = __builtin_amdgcn_s_memrealtime(); }
    auto main_loop = [&](auto a16, auto b16) {
        fetch(r_begin, a16, b16);
        stash(a16, b16);
        __syncthreads();
        if (epi.stamps && tid == 0) epi.stamps[8L * stamp_row + 1] = __builtin_amdgcn_s_memtime();
        for (int r0 = r_begin; r0 < r_end; r0 += BK) {
            const bool more = r0 + BK < r_end;
            if (more) fetch(r0 + BK, a16, b16);
            if (do_colsum) {
                if constexpr (PREC == 1) {
#pragma unroll 8
                    for (int r = 0; r < BK; ++r)
                        csum += __builtin_bit_cast(float, (unsigned)As16[tid * LD16 + r] << 16);
                } else {
#pragma unroll 8
                    for (int r = 0; r < BK; ++r) csum += As[r * LDA + tid];
                }
            }
            if constexpr (PREC == 1) {
#pragma unroll
                for (int kk = kg * (BK / KW); kk < (kg + 1) * (BK / KW); kk += 16) {
                    bf16x8 af[FM], bf[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i)
                        af[i] = *reinterpret_cast<const bf16x8*>(&As16[(wm0 + i * 32 + li) * LD16 + kk + 8 * lh]);
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        if constexpr (B_RC) {
                            bf[j] = *reinterpret_cast<const bf16x8*>(&Bs16[(wn0 + j * 32 + li) * LD16 + kk + 8 * lh]);
                        } else {
                            // lane (q, p) of its 16-lane group supplies row r0 + q, columns c0 + 4 p .. + 3; it receives column
                            // c0 + (lane & 15) of rows r0 .. r0 + 3: two reads give the 8 reduction values of this lane's column
                            typedef short s16x4 __attribute__((ext_vector_type(4)));
                            typedef short s16x8 __attribute__((ext_vector_type(8)));
                            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                            const int col = wn0 + j * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                            const int row = kk + 8 * lh + ((lane & 15) >> 2);
                            const int sw = ((row >> 1) & 1) << 2;               // rows row and row + 4 share bit 1
                            const unsigned short* a0 = &Bs16[row * BN + 8 * ((col >> 3) ^ sw) + (col & 7)];
                            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
                            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * BN));
                            const s16x8 f = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                            bf[j] = __builtin_bit_cast(bf16x8, f);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int kk = kg * (BK / KW); kk < (kg + 1) * (BK / KW); kk += 8) {
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
                stash(a16, b16);
                __syncthreads();
            }
        }
    };
    if (r_begin < r_end) {
        // storage variants (bf16 mode, forward / data-gradient Linears only): A (activations) and / or B (the weights' bf16 shadow)
        constexpr bool CAN_A16 = PREC == 1 && A_RC && VEC == 4 && std::is_same<ASrc, PlainSrc>::value;
        constexpr bool CAN_B16 = PREC == 1 && A_RC && VEC == 4 && std::is_same<BSrc, PlainSrc>::value;
        const bool a_is16 = CAN_A16 && src_is_bf16(A), b_is16 = CAN_B16 && src_is_bf16(B);
        if constexpr (CAN_A16 && CAN_B16) {
            if (a_is16 && b_is16) main_loop(std::true_type{}, std::true_type{});
            else if (a_is16) main_loop(std::true_type{}, std::false_type{});
            else if (b_is16) main_loop(std::false_type{}, std::true_type{});
            else main_loop(std::false_type{}, std::false_type{});
        } else if constexpr (CAN_B16) {
            if (b_is16) main_loop(std::false_type{}, std::true_type{});
            else main_loop(std::false_type{}, std::false_type{});
        } else {
            main_loop(std::false_type{}, std::false_type{});
        }
    }

    if (epi.stamps && tid == 0) epi.stamps[8L * stamp_row + 2] = __builtin_amdgcn_s_memtime();
    if (KW > 1) {
        // sum the wave groups' accumulators (native C/D layout, lane-contiguous LDS patches, binary tree)
        static_assert(KW == 1 || (KW / 2) * NW * FM * FN * 16 * 64 <= LDS_FLOATS, "LDS too small for the k-group reduction");
        constexpr int PATCH = FM * FN * 16 * 64;
#pragma unroll
        for (int half = KW / 2; half >= 1; half >>= 1) {
            if (kg >= half && kg < 2 * half) {
                float* dst = lds + ((kg - half) * NW + wave) * PATCH + lane;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int g = 0; g < 16; ++g) dst[((i * FN + j) * 16 + g) * 64] = acc[i][j][g];
            }
            __syncthreads();
            if (kg < half) {
                const float* src = lds + (kg * NW + wave) * PATCH + lane;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int g = 0; g < 16; ++g) acc[i][j][g] += src[((i * FN + j) * 16 + g) * 64];
            }
            __syncthreads();
        }
        if (kg != 0) return;
    }
    Epilogue e = epi;
    e.C += (long)split * e.split_stride;
    if constexpr (EPI == EPI_GELUBWD) {
        if (e.pre_bf16) run_epilogue<EPI, FM, FN, true>(e, acc, M, N, m0 + wm0, n0 + wn0, lane, lds + wave * EP_WAVE_FLOATS, e.vec_store != 0);
        else run_epilogue<EPI, FM, FN, false>(e, acc, M, N, m0 + wm0, n0 + wn0, lane, lds + wave * EP_WAVE_FLOATS, e.vec_store != 0);
    } else {
        run_epilogue<EPI, FM, FN>(e, acc, M, N, m0 + wm0, n0 + wn0, lane, lds + wave * EP_WAVE_FLOATS, e.vec_store != 0);
    }
    if (epi.stamps && tid == 0) { epi.stamps[8L * stamp_row + 3] = __builtin_amdgcn_s_memtime(); epi.stamps[8L * stamp_row + 6] = __builtin_amdgcn_s_memrealtime(); }
    if (do_colsum && m0 + tid < M) epi.colsum[(long)split * epi.colsum_stride + m0 + tid] = csum;
}

template <int BM, int BN, int BK, int KW, bool A_RC, bool B_RC, int VEC, int EPI, bool SCALE_A, int PREC, class ASrc, class BSrc>
__global__ __launch_bounds__(64 * (BM >= 64 ? 2 : 1) * (BN >= 64 ? 2 : 1) * KW) void gemm_kernel(ASrc A, BSrc B, Epilogue epi, int M, int N, int R,
                                                    int r_per_split, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, BK, KW, A_RC, B_RC>()];
    gemm_body<BM, BN, BK, KW, A_RC, B_RC, VEC, EPI, SCALE_A, PREC, ASrc, BSrc>(lds, A, B, epi, M, N, R, r_per_split, tiles_m, tiles_n,
                                                                              (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.x);
}

// Up to four independent weight-gradient problems (the four Linears of a CSWinBlock: same kernel configuration, operands all
// alive at the end of the block's backward) in ONE launch: an empty launch costs ~3 us in-stream, and the four grids' tails
// fill each other.  Each problem keeps its own XCD-aware block order inside its block range.
constexpr int WGRAD_BATCH = 4;
template <class ASrc, class BSrc>
struct GemmBatch {
    ASrc A[WGRAD_BATCH];
    BSrc B[WGRAD_BATCH];
    Epilogue e[WGRAD_BATCH];
    int M[WGRAD_BATCH], N[WGRAD_BATCH], R[WGRAD_BATCH], rps[WGRAD_BATCH], tm[WGRAD_BATCH], tn[WGRAD_BATCH];
    int first[WGRAD_BATCH + 1];
    int n;
};
typedef GemmBatch<PlainSrc, PlainSrc> WgradBatch;

template <int PREC>
__global__ __launch_bounds__(512) void gemm_wgrad_batch_kernel(WgradBatch b) {
    int p = 0;
    while (p + 1 < b.n && (int)blockIdx.x >= b.first[p + 1]) ++p;
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<64, 64, 64, 2, false, false>()];
    gemm_body<64, 64, 64, 2, false, false, 4, EPI_PLAIN, true, PREC, PlainSrc, PlainSrc>(
        lds, b.A[p], b.B[p], b.e[p], b.M[p], b.N[p], b.R[p], b.rps[p], b.tm[p], b.tn[p], (int)blockIdx.x - b.first[p],
        b.first[p + 1] - b.first[p], (int)blockIdx.x);
}

// The tail of a CSWinBlock's backward in ONE launch: the data gradient of the qkv Linear (dqkv . Wqkv, the last Linear whose input
// gradient is still missing) beside the block's four weight gradients.  Both only wait for dqkv, neither depends on the other, and
// alone the data gradient keeps the matrix pipe busy a third of the time (profiles/round3_sq_counters_fp32.txt: 0.34 against the
// batch's 0.50): in one launch its tiles' load waits, prologues and epilogues sit beside the weight gradients' long k-loops.
// Workgroups [0, nd) are the data gradient's 64 x 64 tiles (two k-groups of four waves, as the weight-gradient workgroups).
struct BlockTail {
    PlainSrc dA, dB;
    Epilogue de;
    int dM, dN, dR, drps, dtm, dtn, nd;
    WgradBatch w;
    ReduceRiders r;          // pending slab reductions of EARLIER launches (the previous block's): the grid's last workgroups
};
__global__ __launch_bounds__(512) void gemm_block_tail_kernel(BlockTail t) {
    constexpr int L0 = gemm_lds_floats<64, 64, 64, 2, true, false>(), L1 = gemm_lds_floats<64, 64, 64, 2, false, false>();
    __shared__ __attribute__((aligned(16))) float lds[L0 > L1 ? L0 : L1];
    const WgradBatch& b = t.w;
    const int nw = b.first[b.n];
    if ((int)blockIdx.x >= nw + t.nd) {    // riders: 256-thread reduction workgroups (the upper four waves leave)
        if (threadIdx.x >= 256) return;
        static_assert(sizeof(lds) >= sizeof(float) * RS_G * (RS_COLS + 1), "reduction scratch");
        rows_sum_dispatch(t.r.j, t.r.first_block, t.r.njobs, (int)blockIdx.x - nw - t.nd, reinterpret_cast<float(*)[RS_COLS + 1]>(lds));
        return;
    }
    if ((int)blockIdx.x >= nw) {           // the data gradient's tiles come after the weight gradients': they fill their tail
        gemm_body<64, 64, 64, 2, true, false, 4, EPI_PLAIN, false, 0, PlainSrc, PlainSrc>(
            lds, t.dA, t.dB, t.de, t.dM, t.dN, t.dR, t.drps, t.dtm, t.dtn, (int)blockIdx.x - nw, t.nd, (int)blockIdx.x);
        return;
    }
    const int bid = (int)blockIdx.x;
    int p = 0;
    while (p + 1 < b.n && bid >= b.first[p + 1]) ++p;
    gemm_body<64, 64, 64, 2, false, false, 4, EPI_PLAIN, true, 0, PlainSrc, PlainSrc>(
        lds, b.A[p], b.B[p], b.e[p], b.M[p], b.N[p], b.R[p], b.rps[p], b.tm[p], b.tn[p], bid - b.first[p],
        b.first[p + 1] - b.first[p], (int)blockIdx.x);
}

// The four parity classes of a stride-2 3x3 convolution's data gradient (ConvTS2Src) are independent GEMMs over a quarter of
// the pixels each: alone they fill 1.1 - 1.2 workgroups per CU (the launch ends with the CUs that got two); together 4.6.
template <int KW, int PREC>
__global__ __launch_bounds__(256 * KW) void gemm_conv_s2_dgrad_batch_kernel(GemmBatch<ConvTS2Src, TapRowsSrc> b) {
    int p = 0;
    while (p + 1 < b.n && (int)blockIdx.x >= b.first[p + 1]) ++p;
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<64, 64, (KW == 1 ? 32 : 64), KW, true, false>()];
    gemm_body<64, 64, (KW == 1 ? 32 : 64), KW, true, false, 4, EPI_PLAIN, false, PREC, ConvTS2Src, TapRowsSrc>(
        lds, b.A[p], b.B[p], b.e[p], b.M[p], b.N[p], b.R[p], b.rps[p], b.tm[p], b.tn[p], (int)blockIdx.x - b.first[p],
        b.first[p + 1] - b.first[p], (int)blockIdx.x);
}

template <int BM, int BN, int BK, int KW, bool A_RC, bool B_RC, int VEC, int EPI, bool SCALE_A, int PREC, class ASrc, class BSrc>
void launch_cfg(const ASrc& A, const BSrc& B, const Epilogue& epi, int M, int N, int R, int splits, int r_per_split,
                hipStream_t st) {
    int tm = cdiv(M, BM), tn = cdiv(N, BN);
    dim3 grid(tm * tn * splits);
    const int pad_lds = cswin_tuning().gemm_pad_lds;                                                    // tuning aid: caps residency
    hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, KW, A_RC, B_RC, VEC, EPI, SCALE_A, PREC, ASrc, BSrc>), grid, dim3(64 * (BM >= 64 ? 2 : 1) * (BN >= 64 ? 2 : 1) * KW), pad_lds, st, A, B, epi,
                       M, N, R, r_per_split, tm, tn);
}

// tile choice (measured on MI355X, tools/gemm_bench.py): the largest tile that still yields >= ~1.5 workgroups per CU

template <bool A_RC, bool B_RC, int VEC, int EPI, bool SCALE_A, class ASrc, class BSrc>
void launch_gemm(const ASrc& A, const BSrc& B, const Epilogue& epi_in, int M, int N, int R, int splits, int r_per_split,
                 int precision, hipStream_t st) {
    Epilogue epi = epi_in;
    epi.vec_store = epilogue_vec_ok(epi, N);
    auto blocks = [&](int bm, int bn) { return (long)cdiv(M, bm) * cdiv(N, bn) * splits; };
    const int forced_kw = cswin_tuning().gemm_kw;                                                               // tuning aid
    // Tile choice.  64 x 64 is the most efficient tile (profiles/round1_gemm_bench.txt), but every workgroup of these
    // launches is resident at once and the kernel ends with the most loaded CU: with t tiles the critical CU does
    // ceil(t / 256) of them.  When that is a poor multiple (stage 3: 296 tiles -> 2 where 1.16 would do) a smaller tile
    // shortens the critical path although it is a little less efficient per flop (penalties measured with gemm_bench).
    const int forced_tile = cswin_tuning().gemm_tile;                                                // tuning aid: 1 = 64x64, 2 = 64x32
    // measured (profiles/round1_gemm_tiles.txt): a 64 x 32 tile costs ~1.2x per flop, so it only pays where it cuts the
    // critical CU's share by more than that (296 -> 592 tiles: 2 -> 1.5 units); 32 x 32 single-wave tiles never paid.
    const double pen2 = cswin_tuning().gemm_pen2;
    auto cost = [&](int bm, int bn, double pen) { return (double)((blocks(bm, bn) + 255) / 256) * bm * bn * pen; };
    int tile = 1;
    if (splits == 1 && cost(64, 32, pen2) < cost(64, 64, 1.0)) tile = 2;
    if (forced_tile) tile = forced_tile;
    const int r_len = r_per_split < R ? r_per_split : R;
    if (precision == 1) {
        // bf16 operands: 16x fewer MFMA cycles per tile, the kernel is bound by staging and barriers: one k-tile of 64,
        // and two wave groups only where a long reduction meets few workgroups
        const long nb = blocks(64, 64);
        if ((nb < 640 && r_len >= 256) || (!A_RC && !B_RC && r_len >= 256))
            launch_cfg<64, 64, 64, 2, A_RC, B_RC, VEC, EPI, SCALE_A, 1>(A, B, epi, M, N, R, splits, r_per_split, st);
        else launch_cfg<64, 64, 64, 1, A_RC, B_RC, VEC, EPI, SCALE_A, 1>(A, B, epi, M, N, R, splits, r_per_split, st);
        return;
    }
    if (tile == 2) {
        if (blocks(64, 32) < 640 && r_len >= 256)
            return launch_cfg<64, 32, 64, 2, A_RC, B_RC, VEC, EPI, SCALE_A, 0>(A, B, epi, M, N, R, splits, r_per_split, st);
        return launch_cfg<64, 32, 32, 1, A_RC, B_RC, VEC, EPI, SCALE_A, 0>(A, B, epi, M, N, R, splits, r_per_split, st);
    }
    // 64 x 64: with fewer than ~2 workgroups per CU and a long reduction, split the k-range of each tile over 2 or 4 wave
    // groups so every SIMD still has independent MFMA chains.
    const long nb = blocks(64, 64);
    int kw = 1;
    if (nb < 640 && r_len >= 256) kw = 2;
    if (nb < 320 && r_len >= 512) kw = 4;
    if (!A_RC && !B_RC && r_len >= 256 && kw < 2) kw = 2;      // weight gradients: measured 5-8 % faster
    if (forced_kw) kw = forced_kw;
    if (kw == 4) launch_cfg<64, 64, 64, 4, A_RC, B_RC, VEC, EPI, SCALE_A, 0>(A, B, epi, M, N, R, splits, r_per_split, st);
    else if (kw == 2) launch_cfg<64, 64, 64, 2, A_RC, B_RC, VEC, EPI, SCALE_A, 0>(A, B, epi, M, N, R, splits, r_per_split, st);
    else launch_cfg<64, 64, 32, 1, A_RC, B_RC, VEC, EPI, SCALE_A, 0>(A, B, epi, M, N, R, splits, r_per_split, st);
}

long long* g_stamps = nullptr;     // debug only (cswin_debug_set_stamps)

Epilogue plain_epilogue(float* C, long ldc) {
    Epilogue e = {};
    e.C = C;
    e.ldc = ldc;
    e.stamps = g_stamps;
    return e;
}

// split count for the M-reduction of a weight gradient: enough workgroups to fill the chip
void choose_split(int M, int out_rows, int out_cols, int* splits, int* r_per_split, int target_override = 0) {
    // Every workgroup is resident at once and the kernel ends with the most loaded CU, so aim at a workgroup count
    // that is a whole multiple of the 256 CUs (3 per CU): slices need not be multiples of the k-tile (the loaders mask
    // the ragged last tile), only of 8.
    const int target_env = cswin_tuning().gemm_split_wgs;                                                       // tuning aid
    const int target = target_override > 0 ? target_override : target_env;
    long tiles = (long)cdiv(out_rows, 64) * cdiv(out_cols, 64);
    int s = (int)(target / tiles);
    if (s < 1) s = 1;
    int max_s = cdiv(M, 2 * BKMAX);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int rps = cdiv(cdiv(M, s), 8) * 8;
    *splits = cdiv(M, rps);
    *r_per_split = rps;
}

}  // namespace

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" {

// debug aid (not part of include/cswin_hip.h): device buffer [nblk][4] of int64 that GEMM workgroups stamp with s_memtime
void cswin_debug_set_stamps(void* p) { g_stamps = (long long*)p; }

int cswin_linear_fwd(const float* x, const float* x2, int k_split, const float* w, const float* bias, float* y,
                     float* y_act, const float* residual, const float* row_scale, int rows_per_sample, int M, int N,
                     int K, int precision, int io_bf16, void* stream) {
    CSWIN_CHECK_PRECISION(precision, "linear_fwd");
    CSWIN_REQUIRE(io_bf16 == 0 || (precision == 1 && (!x2 || !(io_bf16 & 1)) && (io_bf16 & ~7) == 0 && K % 4 == 0 && N % 4 == 0 && !((io_bf16 & 2) && residual)), CSWIN_ERR_UNSUPPORTED,
                  "linear_fwd: bf16 storage (io_bf16 = %d: 1 = x, 2 = y / y_act, 4 = w) needs precision 1, a plain input for bit 1, K and N multiples of 4; a residual output stays fp32", io_bf16);
    CSWIN_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0, CSWIN_ERR_SHAPE, "linear_fwd: bad arguments M=%d N=%d K=%d", M, N, K);
    CSWIN_REQUIRE(!x2 || (k_split > 0 && k_split < K), CSWIN_ERR_SHAPE, "linear_fwd: bad concat split %d of K=%d", k_split, K);
    CSWIN_REQUIRE(!row_scale || rows_per_sample > 0, CSWIN_ERR_SHAPE, "linear_fwd: rows_per_sample must be > 0");
    hipStream_t st = (hipStream_t)stream;
    Epilogue e = plain_epilogue(y, N);
    e.bias = bias;
    e.Cact = y_act; e.ldact = N;
    e.residual = residual; e.ldres = N;
    e.row_scale = row_scale; e.rows_per_sample = rows_per_sample;
    e.c_bf16 = (io_bf16 & 2) != 0;
    PlainSrc B = {w, K, N, K, nullptr, 1, (io_bf16 >> 2) & 1};
    CSWIN_REQUIRE(!(y_act && residual), CSWIN_ERR_UNSUPPORTED, "linear_fwd: activation and residual epilogues are exclusive");
    const int rk = cdiv(K, BKMAX) * BKMAX;
    if (x2) {
        CSWIN_REQUIRE(!y_act, CSWIN_ERR_UNSUPPORTED, "linear_fwd: concat input does not support the activation epilogue");
        ConcatSrc A = {x, x2, k_split, K - k_split, M, K, k_split};
        bool vec = (k_split % 4 == 0) && (K % 4 == 0) && aligned16(x) && aligned16(x2) && aligned16(w);
        CSWIN_REQUIRE(io_bf16 == 0 || (io_bf16 == 4 && vec), CSWIN_ERR_UNSUPPORTED, "linear_fwd: a concat input takes only the bf16 weight flag (4), with 16-B aligned operands");
        if (residual) {
            if (vec) launch_gemm<true, true, 4, EPI_RES, false>(A, B, e, M, N, K, 1, rk, precision, st);
            else launch_gemm<true, true, 1, EPI_RES, false>(A, B, e, M, N, K, 1, rk, precision, st);
        } else {
            if (vec) launch_gemm<true, true, 4, EPI_PLAIN, false>(A, B, e, M, N, K, 1, rk, precision, st);
            else launch_gemm<true, true, 1, EPI_PLAIN, false>(A, B, e, M, N, K, 1, rk, precision, st);
        }
    } else {
        if (precision == 1 && (io_bf16 & 5) == 5) {             // both operands stored as bf16: LDS-DMA kernel (gemm16.hip)
            if (cswin_gemm16(0, y_act ? EPI_ACT : (residual ? EPI_RES : EPI_PLAIN), x, w, &e, M, N, K, stream) == 0) {
                CSWIN_LAUNCH_CHECK();
                return CSWIN_OK;
            }
        }
        PlainSrc A = {x, K, M, K, nullptr, 1, io_bf16 & 1};
        bool vec = (K % 4 == 0) && aligned16(x) && aligned16(w);
        CSWIN_REQUIRE(io_bf16 == 0 || (vec && aligned16(y) && (!y_act || aligned16(y_act)) && (!bias || aligned16(bias))), CSWIN_ERR_ALIGN,
                      "linear_fwd: bf16 storage needs 16-B aligned operands");
        if (y_act) {
            if (vec) launch_gemm<true, true, 4, EPI_ACT, false>(A, B, e, M, N, K, 1, rk, precision, st);
            else launch_gemm<true, true, 1, EPI_ACT, false>(A, B, e, M, N, K, 1, rk, precision, st);
        } else if (residual) {
            if (vec) launch_gemm<true, true, 4, EPI_RES, false>(A, B, e, M, N, K, 1, rk, precision, st);
            else launch_gemm<true, true, 1, EPI_RES, false>(A, B, e, M, N, K, 1, rk, precision, st);
        } else {
            if (vec) launch_gemm<true, true, 4, EPI_PLAIN, false>(A, B, e, M, N, K, 1, rk, precision, st);
            else launch_gemm<true, true, 1, EPI_PLAIN, false>(A, B, e, M, N, K, 1, rk, precision, st);
        }
    }
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// dx[M,K] = (row_scale . dy)[M,N] @ w[N,K]   (optionally * gelu'(gelu_pre), + add; optionally split into dx | dx2)
int cswin_linear_bwd_data(const float* dy, const float* w, float* dx, float* dx2, int k_split, const float* gelu_pre,
                          const float* row_scale, int rows_per_sample, const float* add, int M, int N, int K,
                          int precision, int io_bf16, void* stream) {
    CSWIN_CHECK_PRECISION(precision, "linear_bwd_data");
    CSWIN_REQUIRE(io_bf16 == 0 || (precision == 1 && ((!dx2 && !add) || (io_bf16 & ~4) == 0) && (io_bf16 & ~15) == 0 && K % 4 == 0 && N % 4 == 0), CSWIN_ERR_UNSUPPORTED,
                  "linear_bwd_data: bf16 storage (io_bf16 = %d: 1 = dy, 2 = dx, 4 = w, 8 = gelu_pre) needs precision 1, K and N multiples of 4; split / add outputs take only the weight flag", io_bf16);
    CSWIN_REQUIRE(dy && w && dx && M > 0 && N > 0 && K > 0, CSWIN_ERR_SHAPE, "linear_bwd_data: bad arguments");
    CSWIN_REQUIRE(!dx2 || (k_split > 0 && k_split < K), CSWIN_ERR_SHAPE, "linear_bwd_data: bad concat split");
    CSWIN_REQUIRE(!row_scale || rows_per_sample > 0, CSWIN_ERR_SHAPE, "linear_bwd_data: rows_per_sample must be > 0");
    hipStream_t st = (hipStream_t)stream;
    Epilogue e = plain_epilogue(dx, dx2 ? k_split : K);
    e.C2 = dx2; e.ldc2 = K - k_split; e.col_split = dx2 ? k_split : 0;
    e.gelu_pre = gelu_pre; e.ldpre = K;
    e.row_scale = row_scale; e.rows_per_sample = rows_per_sample;
    e.residual = add; e.ldres = K;
    e.c_bf16 = (io_bf16 & 2) != 0;
    e.pre_bf16 = (io_bf16 & 8) != 0;
    CSWIN_REQUIRE(io_bf16 == 0 || (aligned16(dy) && aligned16(w) && aligned16(dx) && (!gelu_pre || aligned16(gelu_pre))), CSWIN_ERR_ALIGN,
                  "linear_bwd_data: bf16 storage needs 16-B aligned operands");
    PlainSrc A = {dy, N, M, N, nullptr, 1, io_bf16 & 1};
    PlainSrc B = {w, K, N, K, nullptr, 1, (io_bf16 >> 2) & 1};     // S(i = n (reduction), j = k): row-contiguous image
    bool vec = (N % 4 == 0) && (K % 4 == 0) && aligned16(dy) && aligned16(w);
    // output rows = M, output cols = K, reduction = N
    const int rn = cdiv(N, BKMAX) * BKMAX;
    const int modes = (dx2 != nullptr) + (gelu_pre != nullptr) + (add != nullptr);
    CSWIN_REQUIRE(modes <= 1, CSWIN_ERR_UNSUPPORTED, "linear_bwd_data: dx2 / gelu_pre / add are mutually exclusive");
    if (precision == 1 && !dx2 && !add && (io_bf16 & 5) == 5) {  // both operands stored as bf16: LDS-DMA kernel (gemm16.hip)
        if (cswin_gemm16(1, gelu_pre ? EPI_GELUBWD : EPI_PLAIN, dy, w, &e, M, K, N, stream) == 0) {
            CSWIN_LAUNCH_CHECK();
            return CSWIN_OK;
        }
    }
    if (dx2) {
        if (vec) launch_gemm<true, false, 4, EPI_SPLIT2, false>(A, B, e, M, K, N, 1, rn, precision, st);
        else launch_gemm<true, false, 1, EPI_SPLIT2, false>(A, B, e, M, K, N, 1, rn, precision, st);
    } else if (gelu_pre) {
        if (vec) launch_gemm<true, false, 4, EPI_GELUBWD, false>(A, B, e, M, K, N, 1, rn, precision, st);
        else launch_gemm<true, false, 1, EPI_GELUBWD, false>(A, B, e, M, K, N, 1, rn, precision, st);
    } else if (add) {
        if (vec) launch_gemm<true, false, 4, EPI_RES, false>(A, B, e, M, K, N, 1, rn, precision, st);
        else launch_gemm<true, false, 1, EPI_RES, false>(A, B, e, M, K, N, 1, rn, precision, st);
    } else {
        if (vec) launch_gemm<true, false, 4, EPI_PLAIN, false>(A, B, e, M, K, N, 1, rn, precision, st);
        else launch_gemm<true, false, 1, EPI_PLAIN, false>(A, B, e, M, K, N, 1, rn, precision, st);
    }
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

size_t cswin_linear_bwd_weight_workspace(int M, int N, int K) {
    // covers the stand-alone split policy and the batched one (up to 1024 workgroups for a single problem)
    int s0, s1, rps;
    choose_split(M, N, K, &s0, &rps);
    choose_split(M, N, K, &s1, &rps, 1024);
    return (size_t)(s0 > s1 ? s0 : s1) * ((size_t)N * K + N) * sizeof(float);
}

int cswin_linear_bwd_weight_batch(const cswin_wgrad_desc* d, int n, cswin_reduce_job* deferred, const cswin_reduce_job* pending, int npending,
                                  void* stream);
int cswin_rows_sum_multi(const cswin_reduce_job* jobs, int njobs, void* stream);
}  // extern "C"
namespace {
// what may ride at the end of the weight-gradient batch's grid: a plain fp32 data gradient dx[M,K] = dy[M,N] @ w[N,K] (dy == NULL:
// none; cswin_linear_bwd_tail) and reductions left pending by earlier launches
struct TailExtras { const float* dy; const float* w; float* dx; int M, N, K; const cswin_reduce_job* jobs; int njobs; };
int wgrad_batch_impl(const cswin_wgrad_desc* d, int n, cswin_reduce_job* deferred, const TailExtras* extra, void* stream);
}  // namespace
extern "C" {

// dw[N,K] = (row_scale . dy)^T @ [x | x2];  dbias[N] = colsum(row_scale . dy)
int cswin_linear_bwd_weight(const float* dy, const float* x, const float* x2, int k_split, const float* row_scale,
                            int rows_per_sample, float* dw, float* dbias, void* workspace, size_t ws_bytes, int M,
                            int N, int K, cswin_reduce_job* deferred, int precision, void* stream) {
    CSWIN_CHECK_PRECISION(precision, "linear_bwd_weight");
    CSWIN_REQUIRE(dy && x && dw && M > 0 && N > 0 && K > 0, CSWIN_ERR_SHAPE, "linear_bwd_weight: bad arguments");
    CSWIN_REQUIRE(!x2 || (k_split > 0 && k_split < K), CSWIN_ERR_SHAPE, "linear_bwd_weight: bad concat split");
    CSWIN_REQUIRE(!row_scale || rows_per_sample > 0, CSWIN_ERR_SHAPE, "linear_bwd_weight: rows_per_sample must be > 0");
    size_t need = cswin_linear_bwd_weight_workspace(M, N, K);
    CSWIN_REQUIRE(workspace && ws_bytes >= need, CSWIN_ERR_WORKSPACE, "linear_bwd_weight: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    if (precision == 1 && !x2 && N % 4 == 0 && K % 4 == 0 && aligned16(dy) && aligned16(x) && aligned16(workspace)) {
        // bf16 mode: the transposing-read kernel (wgrad16.hip) through the batch entry, as a batch of one
        cswin_wgrad_desc d1 = {dy, x, row_scale, dw, dbias, workspace, ws_bytes, rows_per_sample, M, N, K, 1, 0};
        cswin_reduce_job job;
        int rc = cswin_linear_bwd_weight_batch(&d1, 1, &job, nullptr, 0, stream);
        if (rc) return rc;
        reduce_now_or_defer(job, deferred, st);
        CSWIN_LAUNCH_CHECK();
        return CSWIN_OK;
    }
    int splits, rps;
    choose_split(M, N, K, &splits, &rps);
    // slab s = [dw partial (N*K) | dbias partial (N)]: one reduction launch serves both
    float* slab = (float*)workspace;
    const long slab_stride = (long)N * K + N;
    Epilogue e = plain_epilogue(slab, K);
    e.split_stride = slab_stride;
    e.colsum = dbias ? slab + (long)N * K : nullptr;
    e.colsum_stride = (int)slab_stride;
    PlainSrc A = {dy, N, M, N, row_scale, rows_per_sample};      // S(i = m (reduction), j = n)
    if (x2) {
        ConcatSrc B = {x, x2, k_split, K - k_split, M, K, k_split};
        bool vec = (N % 4 == 0) && (K % 4 == 0) && (k_split % 4 == 0) && aligned16(dy) && aligned16(x) && aligned16(x2);
        if (vec) launch_gemm<false, false, 4, EPI_PLAIN, true>(A, B, e, N, K, M, splits, rps, precision, st);
        else launch_gemm<false, false, 1, EPI_PLAIN, true>(A, B, e, N, K, M, splits, rps, precision, st);
    } else {
        PlainSrc B = {x, K, M, K, nullptr, 1};
        bool vec = (N % 4 == 0) && (K % 4 == 0) && aligned16(dy) && aligned16(x);
        if (row_scale) {
            if (vec) launch_gemm<false, false, 4, EPI_PLAIN, true>(A, B, e, N, K, M, splits, rps, precision, st);
            else launch_gemm<false, false, 1, EPI_PLAIN, true>(A, B, e, N, K, M, splits, rps, precision, st);
        } else {
            if (vec) launch_gemm<false, false, 4, EPI_PLAIN, false>(A, B, e, N, K, M, splits, rps, precision, st);
            else launch_gemm<false, false, 1, EPI_PLAIN, false>(A, B, e, N, K, M, splits, rps, precision, st);
        }
    }
    CSWIN_LAUNCH_CHECK();
    long n = (long)N * K;
    cswin_reduce_job job = {slab, dw, dbias, n, n + (dbias ? N : 0), slab_stride, splits, 0};
    reduce_now_or_defer(job, deferred, st);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// n (<= 4) weight gradients without concat sources in one launch (see gemm_wgrad_batch_kernel); deferred[i] receives problem
// i's slab reduction (run them with cswin_rows_sum_multi).  Falls back to separate launches when a problem is not 16-B
// aligned / a multiple of 4 in N and K.
int cswin_linear_bwd_weight_batch(const cswin_wgrad_desc* d, int n, cswin_reduce_job* deferred, const cswin_reduce_job* pending, int npending,
                                  void* stream) {
    CSWIN_REQUIRE(npending >= 0 && npending <= CSWIN_TAIL_RIDER_JOBS && (npending == 0 || pending), CSWIN_ERR_SHAPE,
                  "linear_bwd_weight_batch: 0..%d pending reductions", CSWIN_TAIL_RIDER_JOBS);
    TailExtras r = {nullptr, nullptr, nullptr, 0, 0, 0, pending, npending};
    return wgrad_batch_impl(d, n, deferred, npending > 0 ? &r : nullptr, stream);
}

// The tail of a CSWinBlock's backward: dx[M,K] = dy[M,N] @ w[N,K] (the qkv Linear's data gradient, plain fp32, no epilogue extras)
// and the block's weight gradients as cswin_linear_bwd_weight_batch takes them.  In fp32 with aligned operands both run in ONE
// launch (gemm_block_tail_kernel); otherwise the data gradient is launched first and the batch follows -- same results either way.
int cswin_linear_bwd_tail(const float* dy, const float* w, float* dx, int M, int N, int K, const cswin_wgrad_desc* d, int n,
                          cswin_reduce_job* deferred, const cswin_reduce_job* pending, int npending, void* stream) {
    CSWIN_REQUIRE(dy && w && dx && M > 0 && N > 0 && K > 0, CSWIN_ERR_SHAPE, "linear_bwd_tail: bad data-gradient arguments");
    CSWIN_REQUIRE(npending >= 0 && npending <= CSWIN_TAIL_RIDER_JOBS && (npending == 0 || pending), CSWIN_ERR_SHAPE,
                  "linear_bwd_tail: 0..%d pending reductions", CSWIN_TAIL_RIDER_JOBS);
    TailExtras r = {dy, w, dx, M, N, K, pending, npending};
    return wgrad_batch_impl(d, n, deferred, &r, stream);
}

}  // extern "C"
namespace {
int wgrad_batch_impl(const cswin_wgrad_desc* d, int n, cswin_reduce_job* deferred, const TailExtras* extra, void* stream) {
    CSWIN_REQUIRE(d && deferred && n >= 1 && n <= WGRAD_BATCH, CSWIN_ERR_SHAPE, "linear_bwd_weight_batch: 1..%d problems and their deferred slots", WGRAD_BATCH);
    bool fast = true;
    const int precision = d[0].precision;
    CSWIN_CHECK_PRECISION(precision, "linear_bwd_weight_batch");
    for (int i = 0; i < n; ++i) {
        CSWIN_REQUIRE(d[i].precision == precision, CSWIN_ERR_UNSUPPORTED, "linear_bwd_weight_batch: problems of one launch share one precision");
        CSWIN_REQUIRE(d[i].io_bf16 == 0 || (precision == 1 && (d[i].io_bf16 & ~3) == 0), CSWIN_ERR_UNSUPPORTED, "linear_bwd_weight_batch: bf16 storage needs precision 1");
        CSWIN_REQUIRE(d[i].dy && d[i].x && d[i].dw && d[i].M > 0 && d[i].N > 0 && d[i].K > 0, CSWIN_ERR_SHAPE, "linear_bwd_weight_batch: bad problem %d", i);
        CSWIN_REQUIRE(!d[i].row_scale || d[i].rows_per_sample > 0, CSWIN_ERR_SHAPE, "linear_bwd_weight_batch: rows_per_sample must be > 0");
        size_t need = cswin_linear_bwd_weight_workspace(d[i].M, d[i].N, d[i].K);
        CSWIN_REQUIRE(d[i].workspace && d[i].ws_bytes >= need, CSWIN_ERR_WORKSPACE, "linear_bwd_weight_batch: workspace %zu < %zu", d[i].ws_bytes, need);
        fast = fast && d[i].N % 4 == 0 && d[i].K % 4 == 0 && aligned16(d[i].dy) && aligned16(d[i].x) && aligned16(d[i].workspace);
    }
    const bool has_dgrad = extra && extra->dy;
    const bool merge_on = cswin_tuning().gemm_tail_merge != 0;                                                  // tuning aid
    const bool ride = extra && merge_on && fast && precision == 0 &&
                      (!has_dgrad || (extra->N % 4 == 0 && extra->K % 4 == 0 && aligned16(extra->dy) && aligned16(extra->w) && aligned16(extra->dx)));
    const bool w16_path = fast && precision == 1 && (cswin_tuning().wgrad16_on || d[0].io_bf16 || (n > 1 && d[1].io_bf16) || (n > 2 && d[2].io_bf16) || (n > 3 && d[3].io_bf16));
    const bool jobs_ride16 = extra && merge_on && w16_path && extra->njobs > 0;       // bf16 mode: the reductions ride in wgrad16's grid
    if (extra && !ride) {                   // the data gradient (and the pending reductions) as launches of their own, then the batch as usual
        if (has_dgrad) {
            int rc = cswin_linear_bwd_data(extra->dy, extra->w, extra->dx, nullptr, 0, nullptr, nullptr, 1, nullptr, extra->M, extra->N,
                                           extra->K, precision == 1 ? 1 : 0, 0, stream);
            if (rc) return rc;
        }
        if (extra->njobs > 0 && !jobs_ride16) {
            int rc = cswin_rows_sum_multi(extra->jobs, extra->njobs, stream);
            if (rc) return rc;
        }
    }
    if (!fast) {
        for (int i = 0; i < n; ++i) CSWIN_REQUIRE(d[i].io_bf16 == 0, CSWIN_ERR_ALIGN, "linear_bwd_weight_batch: bf16 storage needs N, K multiples of 4 and 16-B alignment");
        for (int i = 0; i < n; ++i) {
            int rc = cswin_linear_bwd_weight(d[i].dy, d[i].x, nullptr, 0, d[i].row_scale, d[i].rows_per_sample, d[i].dw, d[i].dbias,
                                             d[i].workspace, d[i].ws_bytes, d[i].M, d[i].N, d[i].K, &deferred[i], precision, stream);
            if (rc) return rc;
        }
        return CSWIN_OK;
    }
    if (w16_path) {
        // bf16 operands: 128 x 128 tiles, ~3 workgroups per CU over the whole batch (load-bound: see wgrad16.hip)
        // Workgroups are shared out in proportion to the work (rows x tiles), so that every workgroup of the launch walks the same
        // number of rows: with equal shares per problem the C x C problem's workgroups finished after 10 k cycles and the C x 4C
        // ones after 37 k (in-kernel stamps, tools/wgrad16_stamps.py), and the launch lasts as long as its slowest workgroup.
        int splits[WGRAD_BATCH], rps[WGRAD_BATCH];
        double work_total = 0.0;
        for (int i = 0; i < n; ++i) work_total += (double)d[i].M * cdiv(d[i].N, 128) * cdiv(d[i].K, 128);
        for (int i = 0; i < n; ++i) {
            const int M = d[i].M, N = d[i].N, K = d[i].K;
            const long slab = ((long)N * K + N) * (long)sizeof(float);
            const int tiles = cdiv(N, 128) * cdiv(K, 128);
            const int w16_wgs = cswin_tuning().w16_wgs, w16_even = cswin_tuning().w16_even;                // tuning aids (1 = equal share per problem)
            int s = w16_even ? (w16_wgs / n) / tiles : (int)(w16_wgs * ((double)M * tiles / work_total) / tiles + 0.5);
            const int cap = (int)(d[i].ws_bytes / slab);
            if (s > cap) s = cap;
            if (s > M / 64) s = M / 64;
            if (s < 1) s = 1;
            rps[i] = cdiv(cdiv(M, s), 32) * 32;
            splits[i] = cdiv(M, rps[i]);
            const long nk = (long)N * K;
            deferred[i] = cswin_reduce_job{(const float*)d[i].workspace, d[i].dw, d[i].dbias, nk, nk + (d[i].dbias ? N : 0), nk + N, splits[i], 0};
        }
        CSWIN_REQUIRE(cswin_wgrad16_batch(d, n, splits, rps, jobs_ride16 ? extra->jobs : nullptr, jobs_ride16 ? extra->njobs : 0, stream, g_stamps) == 0,
                      CSWIN_ERR_SHAPE, "linear_bwd_weight_batch: bad pending reduction");
        CSWIN_LAUNCH_CHECK();
        return CSWIN_OK;
    }
    WgradBatch b = {};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const int M = d[i].M, N = d[i].N, K = d[i].K;
        int splits, rps;
        // 1024 workgroups per launch, i.e. exactly 4 per CU, shared by the problems (measured with four problems: 192 / 256 /
        // 320 per problem -> 13.80 / 13.55 / 14.09 ms per step); a quarter of the slab traffic of four stand-alone launches
        const int batch_env = cswin_tuning().gemm_batch_wgs;                                                      // tuning aid
        // Equal shares per problem (the C x C problem then has 294-row workgroups beside the 1176-row ones of the C x 4C problems).
        // Shares in proportion to the work (CSWIN_GEMM_BATCH_EVEN=0), which pays for the load-bound bf16 kernel above, measured
        // SLOWER here: 13.70 against 13.28 ms/step -- this kernel is matrix-pipe bound and the extra slabs cost more than the tail.
        const int batch_even = cswin_tuning().gemm_batch_even;                                                      // tuning aid
        const int total_wgs = batch_env > 0 ? (batch_env < 1024 ? batch_env : 1024) * n : 1024;
        int batch_target = total_wgs / n;                                                            // the workspace query covers <= 1024
        if (!batch_even) {
            double work = 0.0;
            for (int j = 0; j < n; ++j) work += (double)d[j].M * cdiv(d[j].N, 64) * cdiv(d[j].K, 64);
            batch_target = (int)(total_wgs * ((double)M * cdiv(N, 64) * cdiv(K, 64) / work) + 0.5);
            if (batch_target > 1024) batch_target = 1024;
            if (batch_target < 1) batch_target = 1;
        }
        choose_split(M, N, K, &splits, &rps, batch_target);
        CSWIN_REQUIRE((size_t)splits * ((size_t)N * K + N) * sizeof(float) <= d[i].ws_bytes, CSWIN_ERR_WORKSPACE,
                      "linear_bwd_weight_batch: %d slabs do not fit problem %d's workspace", splits, i);
        float* slab = (float*)d[i].workspace;
        const long slab_stride = (long)N * K + N;
        Epilogue e = plain_epilogue(slab, K);
        e.split_stride = slab_stride;
        e.colsum = d[i].dbias ? slab + (long)N * K : nullptr;
        e.colsum_stride = (int)slab_stride;
        e.vec_store = epilogue_vec_ok(e, K);
        b.A[i] = PlainSrc{d[i].dy, N, M, N, d[i].row_scale, d[i].row_scale ? d[i].rows_per_sample : 1};   // S(i = m (reduction), j = n)
        b.B[i] = PlainSrc{d[i].x, K, M, K, nullptr, 1};
        b.e[i] = e;
        b.M[i] = N; b.N[i] = K; b.R[i] = M; b.rps[i] = rps;
        b.tm[i] = cdiv(N, 64); b.tn[i] = cdiv(K, 64);
        b.first[i] = blocks;
        blocks += b.tm[i] * b.tn[i] * splits;
        const long nk = (long)N * K;
        deferred[i] = cswin_reduce_job{slab, d[i].dw, d[i].dbias, nk, nk + (d[i].dbias ? N : 0), slab_stride, splits, 0};
    }
    b.first[n] = blocks;
    b.n = n;
    if (ride) {
        BlockTail t = {};
        if (has_dgrad) {
            Epilogue e = plain_epilogue(extra->dx, extra->K);
            e.vec_store = epilogue_vec_ok(e, extra->K);
            t.dA = PlainSrc{extra->dy, extra->N, extra->M, extra->N, nullptr, 1, 0};
            t.dB = PlainSrc{extra->w, extra->K, extra->N, extra->K, nullptr, 1, 0};      // S(i = n (reduction), j = k)
            t.de = e;
            t.dM = extra->M; t.dN = extra->K; t.dR = extra->N; t.drps = cdiv(extra->N, BKMAX) * BKMAX;
            t.dtm = cdiv(extra->M, 64); t.dtn = cdiv(extra->K, 64);
            t.nd = t.dtm * t.dtn;
        }
        t.w = b;
        int rblocks = 0;
        if (extra->njobs > 0) {
            rblocks = fill_reduce_table(extra->jobs, extra->njobs, t.r.j, t.r.first_block);
            CSWIN_REQUIRE(rblocks >= 0, CSWIN_ERR_SHAPE, "linear_bwd_tail: bad pending reduction");
            t.r.njobs = extra->njobs;
        }
        static_assert(sizeof(BlockTail) <= 4096, "kernel argument block");
        hipLaunchKernelGGL(gemm_block_tail_kernel, dim3(blocks + t.nd + rblocks), dim3(512), 0, (hipStream_t)stream, t);
    } else if (precision == 1) {
        hipLaunchKernelGGL(gemm_wgrad_batch_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, b);
    } else {
        hipLaunchKernelGGL(gemm_wgrad_batch_kernel<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, b);
    }
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}
}  // namespace
extern "C" {

// -------- convolutions on the (B, H*W, C) token layout (NHWC), implicit GEMM -----------------------------------
// w_perm: [Cout][ks*ks][Cin]  (made by cswin_conv_weight_permute from the nn.Conv2d [Cout][Cin][ks][ks] parameter)
int cswin_conv_tok_fwd(const float* x, const float* w_perm, const float* bias, float* y, int B, int H, int W, int Cin,
                       int Cout, int ks, int stride, int pad, int precision, void* stream) {
    CSWIN_CHECK_PRECISION(precision, "conv_tok_fwd");
    CSWIN_REQUIRE(x && w_perm && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, CSWIN_ERR_SHAPE, "conv_tok_fwd: bad arguments");
    CSWIN_REQUIRE(Cin % 4 == 0 && aligned16(x) && aligned16(w_perm), CSWIN_ERR_ALIGN, "conv_tok_fwd: Cin %% 4 and 16-B alignment required");
    int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    int M = B * OH * OW, R = ks * ks * Cin;
    ConvSrc A = {x, B, H, W, Cin, OH, OW, ks, stride, pad, M, R};
    PlainSrc Bm = {w_perm, R, Cout, R, nullptr, 1};
    Epilogue e = plain_epilogue(y, Cout);
    e.bias = bias;
    launch_gemm<true, true, 4, EPI_PLAIN, false>(A, Bm, e, M, Cout, R, 1, cdiv(R, BKMAX) * BKMAX, precision, (hipStream_t)stream);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// w_permT: [ks*ks][Cout][Cin]
int cswin_conv_tok_bwd_data(const float* dy, const float* w_permT, float* dx, int B, int H, int W, int Cin, int Cout,
                            int ks, int stride, int pad, int precision, void* stream) {
    CSWIN_CHECK_PRECISION(precision, "conv_tok_bwd_data");
    CSWIN_REQUIRE(dy && w_permT && dx, CSWIN_ERR_SHAPE, "conv_tok_bwd_data: null pointer");
    CSWIN_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && aligned16(dy) && aligned16(w_permT), CSWIN_ERR_ALIGN, "conv_tok_bwd_data: channels %% 4 and 16-B alignment required");
    int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    if (ks == 3 && stride == 2 && pad == 1) {
        // four parity classes, each a dense GEMM over only the taps that reach it (Merge_Block.conv, cswin_unet.py:208),
        // launched together (gemm_conv_s2_dgrad_batch_kernel)
        GemmBatch<ConvTS2Src, TapRowsSrc> b = {};
        int blocks = 0, n = 0, rmax = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int H2 = (H - py + 1) / 2, W2 = (W - px + 1) / 2;
                if (H2 <= 0 || W2 <= 0) continue;
                int kys = 0, kxs = 0, taps = 0, nt = 0;
                for (int ky = (py ? 0 : 1); ky < 3; ky += 2)
                    for (int kx = (px ? 0 : 1); kx < 3; kx += 2) {
                        kys |= ky << (2 * nt);
                        kxs |= kx << (2 * nt);
                        taps |= (ky * 3 + kx) << (4 * nt);
                        ++nt;
                    }
                const int Mc = B * H2 * W2, Rc = nt * Cout;
                b.A[n] = ConvTS2Src{dy, B, H, W, Cout, OH, OW, py, px, H2, W2, kys, kxs, Mc, Rc};
                b.B[n] = TapRowsSrc{w_permT, Cout, Cin, taps, Rc, Cin};
                Epilogue e = plain_epilogue(dx, Cin);
                e.rm_on = 1; e.rm_H = H; e.rm_W = W; e.rm_H2 = H2; e.rm_W2 = W2; e.rm_py = py; e.rm_px = px;
                e.vec_store = epilogue_vec_ok(e, Cin);
                b.e[n] = e;
                b.M[n] = Mc; b.N[n] = Cin; b.R[n] = Rc; b.rps[n] = cdiv(Rc, BKMAX) * BKMAX;
                b.tm[n] = cdiv(Mc, 64); b.tn[n] = cdiv(Cin, 64);
                b.first[n] = blocks;
                blocks += b.tm[n] * b.tn[n];
                rmax = Rc > rmax ? Rc : rmax;
                ++n;
            }
        if (n > 0) {
            b.first[n] = blocks;
            b.n = n;
            hipStream_t st = (hipStream_t)stream;
            // few workgroups and a long reduction: split each k-tile over wave groups (same rule as launch_gemm)
            const int kw = (blocks < 320 && rmax >= 512) ? 4 : ((blocks < 640 && rmax >= 256) ? 2 : 1);
            if (precision == 1) {
                hipLaunchKernelGGL((gemm_conv_s2_dgrad_batch_kernel<2, 1>), dim3(blocks), dim3(512), 0, st, b);
            } else if (kw == 4) {
                hipLaunchKernelGGL((gemm_conv_s2_dgrad_batch_kernel<4, 0>), dim3(blocks), dim3(1024), 0, st, b);
            } else if (kw == 2) {
                hipLaunchKernelGGL((gemm_conv_s2_dgrad_batch_kernel<2, 0>), dim3(blocks), dim3(512), 0, st, b);
            } else {
                hipLaunchKernelGGL((gemm_conv_s2_dgrad_batch_kernel<1, 0>), dim3(blocks), dim3(256), 0, st, b);
            }
        }
        CSWIN_LAUNCH_CHECK();
        return CSWIN_OK;
    }
    int M = B * H * W, R = ks * ks * Cout;
    ConvTSrc A = {dy, B, H, W, Cout, OH, OW, ks, stride, pad, M, R};
    PlainSrc Bm = {w_permT, Cin, R, Cin, nullptr, 1};          // S(i = (tap, co), j = ci)
    Epilogue e = plain_epilogue(dx, Cin);
    launch_gemm<true, false, 4, EPI_PLAIN, false>(A, Bm, e, M, Cin, R, 1, cdiv(R, BKMAX) * BKMAX, precision, (hipStream_t)stream);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

size_t cswin_conv_tok_bwd_weight_workspace(int B, int H, int W, int Cin, int Cout, int ks, int stride, int pad) {
    int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    return cswin_linear_bwd_weight_workspace(B * OH * OW, Cout, ks * ks * Cin);
}

// dw_perm: [Cout][ks*ks][Cin], or the nn.Conv2d parameter layout [Cout][Cin][ks][ks] when torch_layout != 0; dbias: [Cout]
int cswin_conv_tok_bwd_weight(const float* dy, const float* x, float* dw_perm, float* dbias, void* workspace,
                              size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ks, int stride, int pad,
                              int torch_layout, cswin_reduce_job* deferred, int precision, void* stream) {
    CSWIN_CHECK_PRECISION(precision, "conv_tok_bwd_weight");
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
    const long slab_stride = (long)Cout * K + Cout;
    Epilogue e = plain_epilogue(slab, K);
    e.split_stride = slab_stride;
    e.colsum = dbias ? slab + (long)Cout * K : nullptr;
    e.colsum_stride = (int)slab_stride;
    PlainSrc A = {dy, Cout, M, Cout, nullptr, 1};
    ConvSrc Bm = {x, B, H, W, Cin, OH, OW, ks, stride, pad, M, K};   // S(i = pixel m (reduction), j = (tap, ci))
    launch_gemm<false, false, 4, EPI_PLAIN, false>(A, Bm, e, Cout, K, M, splits, rps, precision, st);
    CSWIN_LAUNCH_CHECK();
    long n = (long)Cout * K;
    cswin_reduce_job job = {slab, dw_perm, dbias, n, n + (dbias ? Cout : 0), slab_stride, splits, 0, torch_layout ? ks * ks : 0, torch_layout ? Cin : 0};
    if (deferred) {
        *deferred = job;
        return CSWIN_OK;
    }
    if (torch_layout) {
        job.reserved = reduce_job_vec_ok(job);
        hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)((job.n + RS_COLS - 1) / RS_COLS)), dim3(256), 0, st, job);
    } else {
        launch_rows_sum(slab, dw_perm, dbias, n, n + (dbias ? Cout : 0), splits, slab_stride, st);
    }
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// jobs: HOST array of njobs (<= CSWIN_MAX_REDUCE_JOBS = 48) reductions left pending by *_bwd_weight / layernorm_bwd calls with `deferred` set
int cswin_rows_sum_multi(const cswin_reduce_job* jobs, int njobs, void* stream) {
    CSWIN_REQUIRE(jobs && njobs > 0 && njobs <= CSWIN_MAX_REDUCE_JOBS, CSWIN_ERR_SHAPE, "rows_sum_multi: 1..%d jobs", CSWIN_MAX_REDUCE_JOBS);
    ReduceJobs J = {};
    const int blocks = fill_reduce_table(jobs, njobs, J.j, J.first_block);
    CSWIN_REQUIRE(blocks >= 0, CSWIN_ERR_SHAPE, "rows_sum_multi: bad job (part / out / n / rows, or conv_kk without conv_cin)");
    J.njobs = njobs;
    hipLaunchKernelGGL(rows_sum_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

}  // extern "C"
