// Fused cross-shaped-window attention (LePEAttention, networks/cswin_unet.py:31-109) for gfx950.
//
// One workgroup = one (branch, window, head).  Nothing of the reference's intermediate tensors
// exists: q/k/v rows are gathered straight from the (B, L, 3C) output of the qkv Linear with the
// stripe index map  l = (ih*H_sp + r)*W + iw*W_sp + c  (img2windows / im2cswin /
// get_lepe never materialise), K/V stripes are staged in LDS, S = QK^T and PV run on
// v_mfma_f32_16x16x4_f32 (exact fp32) with the KEY index on MFMA rows so that
//   * softmax row statistics are per-lane reductions over registers + two cross-group shuffles,
//   * the P tile in its accumulator layout is directly the B operand of the PV product
//     (no LDS round trip for P),
// the LePE 3x3 depthwise conv (zero padded at the WINDOW border) is evaluated from the V tile
// already in LDS, and the epilogue scatters to (B, L, C) at the branch's channel offset
// (windows2img + torch.cat fused).  Both branches of a block run in ONE launch.
//
// Backward recomputes S from q,k and the saved row log-sum-exp.  Each wave owns 16
// keys: dK^T and dV^T accumulate in registers over all query tiles (P / dS accumulator tiles
// are again direct MFMA B operands), dS crosses LDS once for dQ, LePE^T(dO) is added to dV, and
// the depthwise-conv weight/bias gradients are written as per-workgroup partial slabs (a standard
// cswin_reduce_job, reduced deterministically by rows_sum).  delta = rowsum(dO o y0) with y0 = P V, which
// the forward saves beside y.  Two variants by window size:
//   attn_bwd3_kernel  N <= 112: one workgroup per unit; Q/K aliased, V/dS aliased (37 KB at N = 56, 80 KB at N = 98: four / two
//                               workgroups per CU); S tiles formed while the dO / y0 / v loads are in flight; reduction-free LePE
//                               weight gradient; one fused dP / dV / dK loop
//   attn_delta / attn_bwd_kv / attn_bwd_q / lepe_wgrad   N > 112: two passes, 64 x 64 at a time
// Forward: attn_fwd3_kernel (N <= 128; optional query split over two workgroups), attn_fwd_kernel for larger windows.
// (Persistent workgroups with cross-unit register prefetch were built and measured in round 3: 7-8 % slower at stages 1-2, equal at
// stages 3-4; the non-matrix phases need the resident waves that the prefetch registers cost -- profiles/round3_notes.md.)
// Head dims 8 / 16 / 24 / 32 share the kernels (tiles zero-padded to HD = 32).
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

constexpr int HD = 32;          // head dim (64/2 = 128/4 = 256/8 = 512/16)
constexpr int LDT = HD + 4;     // LDS row stride of the [token][d] images

struct AttnBranch {
    int c0;          // first channel of this branch inside C
    int heads;       // heads of this branch
    int head0;       // global index of its first head (for the lse layout)
    int H_sp, W_sp, nW, nWin;
    int wg_begin;    // first workgroup of this branch
    unsigned m_heads, m_nWin, m_nW, m_Wsp;   // ceil(2^32 / d) of the four divisors of the index arithmetic (fdiv below)
    const float* lepe_w;   // [Cb][9]
    const float* lepe_b;   // [Cb]
    float* dw_part;        // backward: partial slabs [b * nWin + win][Cb * 9 (channel-major, tap minor) | Cb] of the
                           // LePE conv weight / bias gradient: a standard cswin_reduce_job over B * nWin rows
};

struct AttnParams {
    const float* qkv;      // (B, L, 3C)
    float* y;              // (B, L, C)        forward output
    float* y0;             // (B, L, C)        forward output without the LePE term (P V), saved for the backward's delta; or NULL
    float* lse;            // (B, heads_total, L)
    const float* dy;       // (B, L, C)        backward input
    const float* y_in;     // (B, L, C)        backward: the forward's y0 = P V (delta = rowsum(dO o y0) = rowsum(P o dP))
    float* delta;          // (B, heads_total, L) workspace (large-window backward only)
    float* dqkv;           // (B, L, 3C)       backward output
    int B, reso, C, heads_total;
    int hd;                // real head dim (8, 16, 24 or 32); LDS images and MFMA tiles are zero-padded to HD = 32
    float scale;
    float drop_p, drop_scale;   // attention-probability dropout (cswin_unet.py:101): p (0 = off) and 1 / (1 - p)
    unsigned drop_thresh;       // keep iff 24 hash bits >= p * 2^24
    unsigned long long drop_seed;
    const unsigned long long* drop_epoch;   // device-resident step counter added to drop_seed (NULL: none)
    int nbranch;
    int ds_stride;         // fused backward: LDS row stride of the dS image (>= N, = 4 mod 8: conflict-free column writes)
    int slab_rows;         // LePE gradient slab rows per window (1 on every current path)
    int vs_floats;         // fused backward: floats of the V / dS region (see attn_bwd3_kernel)
    long long* stamps;     // debug (cswin_debug_set_attn_stamps): [workgroup][8] s_memtime stamps of wave 0, or NULL
    int qkv_bf16;          // storage mode: bit 0 = qkv and dqkv, bit 1 = y are STORED as bf16 (bf16 activation storage; 0, 1 or 3);
                           // all arithmetic stays fp32
    AttnBranch br[2];
};

// q / k / v and their gradients may be stored as bf16.  Addresses are computed in ELEMENTS on fp32-typed pointers (a plain
// fp32 access when qkv_bf16 == 0); for bf16 storage the element offset is re-applied to the bf16 base.
typedef __bf16 attn_bf16x4 __attribute__((ext_vector_type(4)));
// Storage is a COMPILE-TIME parameter of every kernel that touches q / k / v (Q16): a run-time choice between 8-B and 16-B loads
// splits the load sequences into basic blocks whose loads then wait for one another.  Loads come in two halves: ldq_raw issues
// the load (4 fp32, or 4 bf16 as two raw dwords) and qcv widens it where the value is consumed.  The raw bf16 pair is an
// INTEGER vector on purpose: hipcc 7.2 miscompiles bit casts of the elements of a 2-float vector (element 1 reads element 0).
template <bool Q16> struct QRaw { typedef f32x4 type; };
template <> struct QRaw<true> { typedef u32x2 type; };
template <bool Q16> __device__ __forceinline__ typename QRaw<Q16>::type ldq_raw(const AttnParams& p, const float* elem_ptr) {
    if constexpr (Q16) return *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.qkv) + (elem_ptr - p.qkv));
    else return *reinterpret_cast<const f32x4*>(elem_ptr);
}
__device__ __forceinline__ f32x4 qcv(f32x4 raw) { return raw; }
__device__ __forceinline__ f32x4 qcv(u32x2 raw) {
    return f32x4{__builtin_bit_cast(float, raw[0] << 16), __builtin_bit_cast(float, raw[0] & 0xffff0000u),
                 __builtin_bit_cast(float, raw[1] << 16), __builtin_bit_cast(float, raw[1] & 0xffff0000u)};
}
template <bool Y16> __device__ __forceinline__ typename QRaw<Y16>::type ldy_raw(const AttnParams& p, const float* elem_ptr) {   // y (backward)
    if constexpr (Y16) return *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.y_in) + (elem_ptr - p.y_in));
    else return *reinterpret_cast<const f32x4*>(elem_ptr);
}
template <bool Q16> __device__ __forceinline__ f32x4 ldq(const AttnParams& p, const float* elem_ptr) { return qcv(ldq_raw<Q16>(p, elem_ptr)); }
template <bool Q16> __device__ __forceinline__ void stdq(const AttnParams& p, float* elem_ptr, f32x4 v) {
    if constexpr (Q16) *reinterpret_cast<attn_bf16x4*>(reinterpret_cast<__bf16*>(p.dqkv) + (elem_ptr - p.dqkv)) = __builtin_convertvector(v, attn_bf16x4);
    else *reinterpret_cast<f32x4*>(elem_ptr) = v;
}

struct WgInfo {
    int bi, b, win, g, N, ih, iw;
};

// n / d for 0 <= n < 2^20 and 0 < d < 2^12 by one multiply-high with m = ceil(2^32 / d) (the integer division the compiler
// emits is ~35 VALU instructions; the index arithmetic of a unit has eight of them and these kernels are issue-bound outside
// their MFMA loops)
__device__ __forceinline__ int fdiv(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }
static inline unsigned fdiv_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

// nn.Dropout on the attention probabilities (cswin_unet.py:101, attn_drop_rate > 0; no reference config uses it): the keep factor
// (0 or 1 / (1 - p)) of probability (query tq, key tk) of (batch, head, window) unit `uid` is a counter-based hash of
// (seed, element index), so the backward kernels regenerate the forward's mask instead of storing an N x N tensor per head.
//   forward : y = ((P o M) V) + LePE                     (the softmax statistics are those of the undropped P)
//   backward: dV = (P o M)^T dO,   dS = P o ((dO V^T) o M - delta),   delta = rowsum(dO o (y - LePE)) as without dropout
__device__ __forceinline__ unsigned long long attn_mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ long attn_unit_id(const AttnParams& p, const AttnBranch& br, int b, int g, int win) {
    return ((long)b * p.heads_total + br.head0 + g) * br.nWin + win;
}
__device__ __forceinline__ float attn_keep(const AttnParams& p, long uid, int N, int tq, int tk) {
    const unsigned long long i = ((unsigned long long)uid * (unsigned)N + (unsigned)tq) * (unsigned)N + (unsigned)tk;
    const unsigned long long r = attn_mix64(attn_mix64(p.drop_seed + (p.drop_epoch ? *p.drop_epoch : 0ull)) ^ i);
    return (unsigned)(r >> 40) >= p.drop_thresh ? p.drop_scale : 0.f;
}

// the keep decisions of NB (query, key) pairs as ONE bit mask, built by a rolled loop: unrolled, the 64-bit hashes of a whole tile
// row interleave and cost the persistent kernels 100 VGPRs (spills) for a path no reference configuration enables
template <int NB, typename F>
__device__ __forceinline__ unsigned attn_keep_bits(const AttnParams& p, long uid, int N, F&& qk) {
    unsigned bits = 0;
#pragma unroll 1
    for (int i = 0; i < NB; ++i) {
        int tq, tk;
        qk(i, tq, tk);
        bits |= (attn_keep(p, uid, N, tq, tk) != 0.f ? 1u : 0u) << i;
    }
    return bits;
}

__device__ __forceinline__ WgInfo decode_wg(const AttnParams& p, int wg) {
    WgInfo w;
    w.bi = (p.nbranch > 1 && wg >= p.br[1].wg_begin) ? 1 : 0;
    const AttnBranch& br = p.br[w.bi];
    const int loc = wg - br.wg_begin;
    const int t = br.heads > 1 ? fdiv(loc, br.m_heads) : loc;
    w.g = loc - t * br.heads;
    w.b = br.nWin > 1 ? fdiv(t, br.m_nWin) : t;
    w.win = t - w.b * br.nWin;
    w.ih = br.nW > 1 ? fdiv(w.win, br.m_nW) : w.win;
    w.iw = w.win - w.ih * br.nW;
    w.N = br.H_sp * br.W_sp;
    return w;
}

// in-window token t -> image token l
__device__ __forceinline__ int token_of(const AttnBranch& br, const WgInfo& w, int reso, int t) {
    const int r = br.W_sp > 1 ? fdiv(t, br.m_Wsp) : t, c = t - r * br.W_sp;
    return (w.ih * br.H_sp + r) * reso + w.iw * br.W_sp + c;
}

// partial-slab row of workgroup (b, win, head g): element i = tap * HD + d of the [10][HD] scratch (tap 9 = bias)
__device__ __forceinline__ void store_lepe_partial(const AttnParams& p, const AttnBranch& br, const WgInfo& w, int sub, int i, float v) {
    const int tap = i / HD, d = i - tap * HD;
    if (d >= p.hd) return;
    const int cb = br.heads * p.hd;                         // channels of this branch
    float* row = br.dw_part + (((long)w.b * br.nWin + w.win) * p.slab_rows + sub) * (cb * 10);
    const int ch = w.g * p.hd + d;
    if (tap < 9) row[ch * 9 + tap] = v;
    else row[cb * 9 + ch] = v;
}

#define ATTN_STAMP(k)                                                                                   \
    do {                                                                                                \
        if (p.stamps && threadIdx.x == 0) p.stamps[(long)blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); \
    } while (0)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bf16 matrix instructions of storage mode 7 (bf16 MFMA: BASELINE configs[2..4] pin the softmax / matmul dtype of the attention
// to bf16, cswin_unet.py:100).  Fragments are built from the SAME fp32 registers / LDS images as the fp32 path and rounded
// (v_cvt_pk_bf16_f32) on the way in; accumulation, softmax statistics and everything stored stay as before.
//   mfma32: v_mfma_f32_16x16x32_bf16, a lane holds 8 consecutive k = 8 (lane >> 4) .. + 7  (one instruction = the eight
//           16x16x4 steps over the 32 head channels);
//   mfma16: v_mfma_f32_16x16x16_bf16, a lane holds 4 consecutive k = 4 (lane >> 4) .. + 3  (= the four rows 4 kq + r of an
//           accumulator tile: P / dS tiles are B operands as they stand).
typedef short attn_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 attn_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ attn_s16x4 pk4(f32x4 v) { return __builtin_bit_cast(attn_s16x4, __builtin_convertvector(v, attn_bf16x4)); }
__device__ __forceinline__ attn_bf16x8 pk8(f32x4 lo, f32x4 hi) {
    return __builtin_shufflevector(__builtin_convertvector(lo, attn_bf16x4), __builtin_convertvector(hi, attn_bf16x4), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 mfma16(attn_s16x4 a, attn_s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma32(attn_bf16x8 a, attn_bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ---- LePE taps ------------------------------------------------------------------------------------------------
// The 3x3 depthwise conv is zero padded at the WINDOW border.  Reads use a clamped address and a select instead of a
// branch, so that the LDS reads of all taps are in flight together.  THIN: the stripe is one token high or wide
// (split_size 1, stage 1): only the three taps along the stripe can stay inside; they are picked with wave-uniform
// arithmetic (tap j = (1, j) for a 1 x W stripe, (j, 1) for an H x 1 stripe) and the other six are never issued.
// SIGN = +1: neighbour (r + ky - 1, c + kx - 1) (forward conv, weight gradient); -1: (r - ky + 1, c - kx + 1) (transpose).
template <bool THIN, int SIGN>
__device__ __forceinline__ f32x4 lepe_taps4(const AttnBranch& br, const float* __restrict__ src, const float* __restrict__ Wl,
                                            int rr, int cc, int self_t, int d0, f32x4 acc) {
    if constexpr (THIN) {
        // a 1 x W or H x 1 stripe is a line: the neighbours of token t are t - 1, t, t + 1 (rr / cc are not used)
        const bool row = br.H_sp == 1;
        const int n = br.H_sp * br.W_sp;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tap = row ? 3 + j : 3 * j + 1;
            const int t2 = self_t + SIGN * (j - 1);
            const bool ok = (unsigned)t2 < (unsigned)n;
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wl[tap * HD + d0]);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(&src[(ok ? t2 : self_t) * LDT + d0]);
            acc += (ok ? 1.f : 0.f) * wv * vv;
        }
    } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int r2 = rr + SIGN * (ky - 1), c2 = cc + SIGN * (kx - 1);
                const bool ok = (unsigned)r2 < (unsigned)br.H_sp && (unsigned)c2 < (unsigned)br.W_sp;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wl[(ky * 3 + kx) * HD + d0]);
                const f32x4 vv = *reinterpret_cast<const f32x4*>(&src[(ok ? r2 * br.W_sp + c2 : self_t) * LDT + d0]);
                acc += (ok ? 1.f : 0.f) * wv * vv;
            }
    }
    return acc;
}

// weight-gradient accumulation of one token: a[tap] += g * V[neighbour(tap)][d]  (a[9] += g is done by the caller)
template <bool THIN>
__device__ __forceinline__ void lepe_wgrad_taps(const AttnBranch& br, const float* __restrict__ Vs, int rr, int cc, int t, int d,
                                                float g, float* a) {
    if constexpr (THIN) {
        const int n = br.H_sp * br.W_sp;
#pragma unroll
        for (int j = 0; j < 3; ++j) {                           // a[j] holds tap (1, j) or (j, 1); expanded by the caller
            const int t2 = t + j - 1;                           // the stripe is a line (rr / cc are not used)
            const bool ok = (unsigned)t2 < (unsigned)n;
            const float vv = Vs[(ok ? t2 : t) * LDT + d];
            a[j] += ok ? g * vv : 0.f;
        }
    } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int r2 = rr + ky - 1, c2 = cc + kx - 1;
                const bool ok = (unsigned)r2 < (unsigned)br.H_sp && (unsigned)c2 < (unsigned)br.W_sp;
                const float vv = Vs[(ok ? r2 * br.W_sp + c2 : t) * LDT + d];
                a[ky * 3 + kx] += ok ? g * vv : 0.f;
            }
    }
}

// =====================================================================================
// forward
// =====================================================================================
// QS = 2: the query tiles of a (window, head) are split over two workgroups (each stages the whole K / V stripe): with
// 384 units on 256 CUs half the CUs would otherwise carry two whole units and set the kernel time; 768 half-units are
// three per CU.  The two halves are `units` apart in the grid, i.e. on the same XCD / L2 when units % 8 == 0.
template <int NT, int QS, int ST>
__global__ __launch_bounds__(64 * (QS == 2 ? (NT + 1) / 2 : (NT < 8 ? NT : 8))) void attn_fwd_kernel(AttnParams p, int units) {
    constexpr bool Q16 = (ST & 1) != 0, Y16 = (ST & 2) != 0, M16 = (ST & 4) != 0;      // storage of qkv / dqkv and of y, bf16 MFMAs (see AttnParams)
    (void)Q16; (void)Y16; (void)M16;
    constexpr int NP = 16 * NT;
    constexpr int NW = QS == 2 ? (NT + 1) / 2 : (NT < 8 ? NT : 8);
    static_assert(QS == 1 || NT <= 18, "query split: one query tile per wave, at most 9 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                  // [NP][LDT]
    float* Vs = Ks + NP * LDT;         // [NP][LDT]
    float* Wl = Vs + NP * LDT;         // [10][32]: 9 taps + bias of this head's channels

    const int half = QS == 2 ? (int)blockIdx.x / units : 0;
    const WgInfo w = decode_wg(p, QS == 2 ? (int)blockIdx.x - half * units : (int)blockIdx.x);
    const AttnBranch& br = p.br[w.bi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt0 = half * NW + wave;             // this wave's first (QS = 2: only) query tile
    const int li = lane & 15, kq = lane >> 4;
    const int L = p.reso * p.reso, C3 = 3 * p.C;
    const int ch0 = br.c0 + w.g * p.hd;           // first channel of this head inside C
    const int N = w.N;
    const float* qkv_b = p.qkv + (long)w.b * L * C3;
    const bool thin = br.H_sp == 1 || br.W_sp == 1;       // wave-uniform

    ATTN_STAMP(0);
    // this wave's first query tile: issue its Q loads first so their latency overlaps the K/V staging below
    typename QRaw<Q16>::type q0_pre = {}, q1_pre = {};      // raw (see ldq_raw)
    int lq_pre = 0;
    {
        const int tq = 16 * qt0 + li;
        if (qt0 < NT && tq < N) lq_pre = token_of(br, w, p.reso, tq);
        if (qt0 < NT && tq < N && 8 * kq < p.hd) {
            const float* src = qkv_b + (long)lq_pre * C3 + ch0 + 8 * kq;
            q0_pre = ldq_raw<Q16>(p, src);
            q1_pre = ldq_raw<Q16>(p, src + 4);
        }
    }

    for (int idx = tid; idx < NP * 8; idx += 64 * NW) {
        const int row = idx >> 3, c4 = idx & 7;
        typename QRaw<Q16>::type kv = {}, vv = {};
        if (row < N && 4 * c4 < p.hd) {
            const float* src = qkv_b + (long)token_of(br, w, p.reso, row) * C3 + ch0 + 4 * c4;
            kv = ldq_raw<Q16>(p, src + p.C);
            vv = ldq_raw<Q16>(p, src + 2 * p.C);
        }
        *reinterpret_cast<f32x4*>(&Ks[row * LDT + 4 * c4]) = qcv(kv);
        *reinterpret_cast<f32x4*>(&Vs[row * LDT + 4 * c4]) = qcv(vv);
    }
    for (int i = tid; i < 10 * HD; i += 64 * NW) {
        const int tap = i / HD, ch = i - tap * HD;
        const int cb = ch0 - br.c0 + ch;        // channel inside the branch
        Wl[i] = ch >= p.hd ? 0.f : (tap < 9 ? br.lepe_w[cb * 9 + tap] : br.lepe_b[cb]);
    }
    __syncthreads();
    ATTN_STAMP(1);

    for (int qt = qt0; qt < NT; qt += QS == 2 ? NT : NW) {
        const int tq = 16 * qt + li;
        const bool qvalid = tq < N;
        int lq = lq_pre;
        float qr[8];
        {
            typename QRaw<Q16>::type q0r = q0_pre, q1r = q1_pre;
            if (qt != qt0) {                        // only when a wave owns more than one query tile (N > 128)
                q0r = {};
                q1r = {};
                lq = qvalid ? token_of(br, w, p.reso, tq) : 0;
                if (qvalid && 8 * kq < p.hd) {
                    const float* src = qkv_b + (long)lq * C3 + ch0 + 8 * kq;
                    q0r = ldq_raw<Q16>(p, src);
                    q1r = ldq_raw<Q16>(p, src + 4);
                }
            }
            const f32x4 q0 = qcv(q0r), q1 = qcv(q1r);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qr[e] = q0[e] * p.scale;
                qr[4 + e] = q1[e] * p.scale;
            }
        }
        // S^T tiles: rows = keys (16 kt + 4 kq + reg), col = query li
        const attn_bf16x8 qb = pk8(f32x4{qr[0], qr[1], qr[2], qr[3]}, f32x4{qr[4], qr[5], qr[6], qr[7]});     // M16 only
        f32x4 s[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const float* kp = &Ks[(16 * kt + li) * LDT + 8 * kq];
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(kp);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(kp + 4);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (M16) {
                acc = mfma32(pk8(k0, k1), qb, acc);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma4(k0[e], qr[e], acc);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma4(k1[e], qr[4 + e], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kt + 4 * kq + r >= N) acc[r] = -INFINITY;
                mx = fmaxf(mx, acc[r]);
            }
            s[kt] = acc;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (p.drop_p > 0.f) {                      // attention dropout: P o M goes into P V, the statistics stay those of P
            const long uid = attn_unit_id(p, br, w.b, w.g, w.win);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] *= attn_keep(p, uid, N, tq, 16 * kt + 4 * kq + r);
        }
        if (qt == qt0) ATTN_STAMP(2);

        // O^T[d][q] = sum_key V[key][d] * P^T[key][q]; the P accumulator tile is the B operand as it stands
        f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if constexpr (M16) {
                const float* vp = &Vs[(16 * kt + 4 * kq) * LDT + li];
                const attn_s16x4 pb = pk4(s[kt]);
                o[0] = mfma16(pk4(f32x4{vp[0], vp[LDT], vp[2 * LDT], vp[3 * LDT]}), pb, o[0]);
                o[1] = mfma16(pk4(f32x4{vp[16], vp[LDT + 16], vp[2 * LDT + 16], vp[3 * LDT + 16]}), pb, o[1]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* vp = &Vs[(16 * kt + 4 * kq + r) * LDT + li];
                    o[0] = mfma4(vp[0], s[kt][r], o[0]);
                    o[1] = mfma4(vp[16], s[kt][r], o[1]);
                }
            }
        }
        // lane now holds O^T[d = 16 df + 4 kq + e][q = li]
        if (qt == qt0) ATTN_STAMP(3);
        if (qvalid) {
            const int rr = thin ? 0 : fdiv(tq, br.m_Wsp), cc = thin ? 0 : tq - rr * br.W_sp;      // thin stripes: unused
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                const int d0 = 16 * df + 4 * kq;
                f32x4 acc = *reinterpret_cast<const f32x4*>(&Wl[9 * HD + d0]);     // bias
                acc = thin ? lepe_taps4<true, 1>(br, Vs, Wl, rr, cc, tq, d0, acc) : lepe_taps4<false, 1>(br, Vs, Wl, rr, cc, tq, d0, acc);
                const f32x4 out0 = o[df] * inv, out = out0 + acc;
                if (d0 < p.hd) {
                    const long yi = ((long)w.b * L + lq) * p.C + ch0 + d0;
                    if constexpr (Y16) *reinterpret_cast<attn_bf16x4*>(reinterpret_cast<__bf16*>(p.y) + yi) = __builtin_convertvector(out, attn_bf16x4);
                    else *reinterpret_cast<f32x4*>(p.y + yi) = out;
                    if (p.y0) {
                        if constexpr (Y16) *reinterpret_cast<attn_bf16x4*>(reinterpret_cast<__bf16*>(p.y0) + yi) = __builtin_convertvector(out0, attn_bf16x4);
                        else *reinterpret_cast<f32x4*>(p.y0 + yi) = out0;
                    }
                }
            }
            if (kq == 0) p.lse[((long)w.b * p.heads_total + br.head0 + w.g) * L + lq] = mx + __logf(sum);
        }
    }
    ATTN_STAMP(4);
}

// =====================================================================================
// forward, windows of up to 128 tokens
// =====================================================================================
// An ITEM is one (branch, window, head) unit (QS = 1) or one half of its query tiles (QS = 2: `units` apart in the grid, i.e.
// on the same XCD when units is a multiple of 8); one workgroup per item, one query tile per wave.  All q / k / v loads are
// issued first; the K stripe goes to LDS at once, the V stripe only after the S / softmax phase, so that its load latency hides
// behind that phase.  (Round 3 also built persistent workgroups that prefetch their next item into registers while computing
// the current one: with 2 - 3 resident workgroups per CU instead of 4 - 6 and ~40 more live registers it was 5 - 8 % slower on
// every stage than one workgroup per item, whose neighbours on the CU hide the same latencies; profiles/round3_notes.md.)
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }

template <int NT, int QS, int ST>
__global__ __launch_bounds__(64 * (QS == 2 ? (NT + 1) / 2 : NT), 4) void attn_fwd3_kernel(AttnParams p, int units) {
    constexpr bool Q16 = (ST & 1) != 0, Y16 = (ST & 2) != 0, M16 = (ST & 4) != 0;
    (void)Q16; (void)Y16; (void)M16;
    static_assert(NT <= 8, "one query tile per wave");
    constexpr int NP = 16 * NT;
    constexpr int NW = QS == 2 ? (NT + 1) / 2 : NT;
    constexpr int T = 64 * NW;
    constexpr int NLD = (NP * 8 + T - 1) / T;          // 16-B row chunks of K (and of V) staged per thread
    static_assert(10 * HD <= 2 * T, "LePE taps: at most two per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                  // [NP][LDT]
    float* Vs = Ks + NP * LDT;         // [NP][LDT]
    float* Wl = Vs + NP * LDT;         // [10][32]: 9 taps + bias of this head's channels

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int L = p.reso * p.reso, C3 = 3 * p.C;
    typedef typename QRaw<Q16>::type raw_t;

    // registers of the item being fetched
    raw_t kraw[NLD], vraw[NLD], q0r = {}, q1r = {};
    float wl0 = 0.f, wl1 = 0.f;
    int lq_r = 0;
    auto item_info = [&](int item, int& half) {
        half = QS == 2 ? item / units : 0;
        return decode_wg(p, QS == 2 ? item - half * units : item);
    };
    auto issue_kq = [&](int item) {
        int half;
        const WgInfo w = item_info(item, half);
        const AttnBranch& br = p.br[w.bi];
        const int ch0 = br.c0 + w.g * p.hd, N = w.N;
        const float* qkv_b = p.qkv + (long)w.b * L * C3;
        const int qt0 = half * NW + wave, tq = 16 * qt0 + li;
        q0r = {};
        q1r = {};
        lq_r = 0;
        if (qt0 < NT && tq < N) lq_r = token_of(br, w, p.reso, tq);
        if (qt0 < NT && tq < N && 8 * kq < p.hd) {
            const float* src = qkv_b + (long)lq_r * C3 + ch0 + 8 * kq;
            q0r = ldq_raw<Q16>(p, src);
            q1r = ldq_raw<Q16>(p, src + 4);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * T, row = idx >> 3, c4 = idx & 7;
            kraw[i] = {};
            if (idx < NP * 8 && row < N && 4 * c4 < p.hd)
                kraw[i] = ldq_raw<Q16>(p, qkv_b + (long)token_of(br, w, p.reso, row) * C3 + ch0 + 4 * c4 + p.C);
        }
        {
            const int tap = tid / HD, ch = tid - tap * HD, cb = ch0 - br.c0 + ch;
            wl0 = (tid >= 10 * HD || ch >= p.hd) ? 0.f : (tap < 9 ? br.lepe_w[cb * 9 + tap] : br.lepe_b[cb]);
            const int i2 = tid + T, tap2 = i2 / HD, ch2 = i2 - tap2 * HD, cb2 = ch0 - br.c0 + ch2;
            wl1 = (i2 >= 10 * HD || ch2 >= p.hd) ? 0.f : (tap2 < 9 ? br.lepe_w[cb2 * 9 + tap2] : br.lepe_b[cb2]);
        }
    };
    auto issue_v = [&](int item) {
        int half;
        const WgInfo w = item_info(item, half);
        const AttnBranch& br = p.br[w.bi];
        const int ch0 = br.c0 + w.g * p.hd, N = w.N;
        const float* qkv_b = p.qkv + (long)w.b * L * C3;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * T, row = idx >> 3, c4 = idx & 7;
            vraw[i] = {};
            if (idx < NP * 8 && row < N && 4 * c4 < p.hd)
                vraw[i] = ldq_raw<Q16>(p, qkv_b + (long)token_of(br, w, p.reso, row) * C3 + ch0 + 4 * c4 + 2 * p.C);
        }
    };

    const int item = blockIdx.x;
    ATTN_STAMP(0);
    issue_kq(item);
    issue_v(item);
    {
        int half;
        const WgInfo w = item_info(item, half);
        const AttnBranch& br = p.br[w.bi];
        const int ch0 = br.c0 + w.g * p.hd, N = w.N;
        const bool thin = br.H_sp == 1 || br.W_sp == 1;       // wave-uniform
        const int qt = half * NW + wave;                      // this wave's query tile
        const int tq = 16 * qt + li;
        const bool qvalid = qt < NT && tq < N;
        const int lq = lq_r;

        // ---- K stripe, LePE taps -> LDS; this wave's Q fragment ----
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * T, row = idx >> 3, c4 = idx & 7;
            if (idx < NP * 8) *reinterpret_cast<f32x4*>(&Ks[row * LDT + 4 * c4]) = qcv(kraw[i]);
        }
        if (tid < 10 * HD) Wl[tid] = wl0;
        if (tid + T < 10 * HD) Wl[tid + T] = wl1;
        float qr[8];
        {
            const f32x4 q0 = qcv(q0r), q1 = qcv(q1r);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qr[e] = q0[e] * p.scale;
                qr[4 + e] = q1[e] * p.scale;
            }
        }
        lds_barrier();
        ATTN_STAMP(1);

        // ---- S^T tiles: rows = keys (16 kt + 4 kq + reg), col = query li; softmax over registers + two shuffles ----
        f32x4 s[NT];
        float mx = -INFINITY, sum = 0.f;
        if (qt < NT) {
            const attn_bf16x8 qb = pk8(f32x4{qr[0], qr[1], qr[2], qr[3]}, f32x4{qr[4], qr[5], qr[6], qr[7]});     // M16 only
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const float* kp = &Ks[(16 * kt + li) * LDT + 8 * kq];
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(kp);
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(kp + 4);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if constexpr (M16) {
                    acc = mfma32(pk8(k0, k1), qb, acc);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma4(k0[e], qr[e], acc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma4(k1[e], qr[4 + e], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * kt + 4 * kq + r >= N) acc[r] = -INFINITY;
                    mx = fmaxf(mx, acc[r]);
                }
                s[kt] = acc;
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[kt][r] - mx);
                    s[kt][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (p.drop_p > 0.f) {                      // attention dropout: P o M goes into P V, the statistics stay those of P
                const unsigned keep = attn_keep_bits<4 * NT>(p, attn_unit_id(p, br, w.b, w.g, w.win), N,
                                                             [&](int i, int& q_, int& k_) { q_ = tq; k_ = 16 * (i >> 2) + 4 * kq + (i & 3); });
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][r] *= (keep >> (4 * kt + r)) & 1u ? p.drop_scale : 0.f;
            }
        }
        ATTN_STAMP(2);

        // ---- V stripe -> LDS ----
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * T, row = idx >> 3, c4 = idx & 7;
            if (idx < NP * 8) *reinterpret_cast<f32x4*>(&Vs[row * LDT + 4 * c4]) = qcv(vraw[i]);
        }
        lds_barrier();

        // ---- O^T[d][q] = sum_key V[key][d] * P^T[key][q]; the P accumulator tile is the B operand as it stands ----
        if (qt < NT) {
            const float inv = 1.0f / sum;
            f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                if constexpr (M16) {
                    const float* vp = &Vs[(16 * kt + 4 * kq) * LDT + li];
                    const attn_s16x4 pb = pk4(s[kt]);
                    o[0] = mfma16(pk4(f32x4{vp[0], vp[LDT], vp[2 * LDT], vp[3 * LDT]}), pb, o[0]);
                    o[1] = mfma16(pk4(f32x4{vp[16], vp[LDT + 16], vp[2 * LDT + 16], vp[3 * LDT + 16]}), pb, o[1]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* vp = &Vs[(16 * kt + 4 * kq + r) * LDT + li];
                        o[0] = mfma4(vp[0], s[kt][r], o[0]);
                        o[1] = mfma4(vp[16], s[kt][r], o[1]);
                    }
                }
            }
            ATTN_STAMP(3);
            // lane holds O^T[d = 16 df + 4 kq + e][q = li]: LePE from the V image, scatter to (B, L, C)
            if (qvalid) {
                const int rr = thin ? 0 : fdiv(tq, br.m_Wsp), cc = thin ? 0 : tq - rr * br.W_sp;      // thin stripes: unused
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const int d0 = 16 * df + 4 * kq;
                    f32x4 acc = *reinterpret_cast<const f32x4*>(&Wl[9 * HD + d0]);     // bias
                    acc = thin ? lepe_taps4<true, 1>(br, Vs, Wl, rr, cc, tq, d0, acc) : lepe_taps4<false, 1>(br, Vs, Wl, rr, cc, tq, d0, acc);
                    const f32x4 out0 = o[df] * inv, out = out0 + acc;
                    if (d0 < p.hd) {
                        const long yi = ((long)w.b * L + lq) * p.C + ch0 + d0;
                        if constexpr (Y16) *reinterpret_cast<attn_bf16x4*>(reinterpret_cast<__bf16*>(p.y) + yi) = __builtin_convertvector(out, attn_bf16x4);
                        else *reinterpret_cast<f32x4*>(p.y + yi) = out;
                        if (p.y0) {
                            if constexpr (Y16) *reinterpret_cast<attn_bf16x4*>(reinterpret_cast<__bf16*>(p.y0) + yi) = __builtin_convertvector(out0, attn_bf16x4);
                            else *reinterpret_cast<f32x4*>(p.y0 + yi) = out0;
                        }
                    }
                }
                if (kq == 0) p.lse[((long)w.b * p.heads_total + br.head0 + w.g) * L + lq] = mx + __logf(sum);
            }
        }
        ATTN_STAMP(4);
    }
}

// =====================================================================================
// backward
// =====================================================================================
// sum over the 8 lanes that share lane >> 3 (quad swap, pair-of-quads swap, half-row mirror): pure VALU
__device__ __forceinline__ float oct_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    return v;
}

// v[lane] + v[lane ^ 16] and v[lane] + v[lane ^ 32] by the gfx950 row / half swaps: VALU only (a __shfl_xor is a ds_bpermute round trip)
__device__ __forceinline__ float xor16_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// sum over the 16 lanes of a DPP row (the lanes that share lane >> 4): pure VALU, every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v = oct_sum(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

// Fused backward, windows of up to 112 tokens: one workgroup per (branch, window, head) unit.
// LDS: QK [NP][LDT] (Q, later K) | Ds [NP][LDT] (dO) | VS (V, later the dS image [N][ds_stride] + pad) | lse, delta [NP] |
// LePE taps + bias [10][HD]: 73 KB for N = 98, two workgroups per CU.  Wave w owns the 16 keys of key tile w.  Per item:
//   A  q, lse -> LDS and the S tiles of every query tile (MFMA) while the other loads are in flight; then v, dO, y0, taps ->
//      LDS; delta[q] = sum_d dO[q][d] y0[q][d] (= rowsum(P o dP): y0 = P V is the forward's
//      output WITHOUT the LePE term, saved by the forward for exactly this) from the registers being staged, an 8-lane DPP sum
//      per token; K / V fragments of the wave's keys -> registers; barrier
//   B  LePE conv weight / bias gradient: wave w takes taps w, w + NT, ...; a lane = (token slice, 16-B channel chunk), the eight
//      slices meet by DPP / shuffles, lanes 0-7 store the (window, head) partial -- no LDS scratch, no barrier
//   C  barrier (V image dead); per query tile: dP (MFMA) -> P, dS (VALU) -> dV^T += dO^T P, dK^T += Q^T dS (MFMA, the P / dS
//      accumulator tiles are the B operands as they stand); dS -> LDS over the V image.  The dP product of tile qt + 1 is
//      issued before the VALU work of tile qt.  Then dV += LePE^T(dO); dK, dV -> global
//   D  barrier; K fragments -> LDS over the dead Q image; barrier; dQ^T = K^T dS^T per query tile (one per wave) -> global
template <int NT, int ST>
__global__ __launch_bounds__(64 * NT, 4) void attn_bwd3_kernel(AttnParams p) {
    constexpr bool Q16 = (ST & 1) != 0, Y16 = (ST & 2) != 0, M16 = (ST & 4) != 0;
    (void)Q16; (void)Y16; (void)M16;
    constexpr int NP = 16 * NT;
    constexpr int T = 64 * NT;                 // NP * 8 = 2 T: two 16-B row chunks of each image per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = p.ds_stride;
    float* QK = smem;                       // [NP][LDT]
    float* Ds = QK + NP * LDT;              // [NP][LDT]
    float* VS = Ds + NP * LDT;              // V [NP][LDT], then dS [N][S] (+ NP finite floats that row N - 1's tail reads)
    float* lse_s = VS + p.vs_floats;
    float* del_s = lse_s + NP;
    float* Wl = del_s + NP;                 // [10][HD]
    float* Gl = Wl + 10 * HD;               // [9 + NT][HD]: LePE weight gradient of this unit (taps by the waves that own them), per-wave bias gradients

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int L = p.reso * p.reso, C3 = 3 * p.C;
    typedef typename QRaw<Q16>::type raw_t;
    typedef typename QRaw<Y16>::type rawy_t;

    const int item = blockIdx.x;
    // the tail behind the dS image is only ever read (by padded query rows, against zero K rows): make it finite once
    for (int i = tid; i < NP; i += T) VS[p.vs_floats - NP + i] = 0.f;

    // registers of the item being fetched
    raw_t qv[2], vv[2], k0r = {}, k1r = {};
    rawy_t yv[2];
    f32x4 dv[2];
    int bbits[2];              // border bits of the staged rows: 1 = row >= 1, 2 = row <= H_sp - 2, 4 = col >= 1, 8 = col <= W_sp - 2 (0: no token)
    float lsev = 0.f, wl0 = 0.f, wl1 = 0.f;
    auto issue = [&](int it_) {
        const WgInfo w = decode_wg(p, it_);
        const AttnBranch& br = p.br[w.bi];
        const int ch0 = br.c0 + w.g * p.hd, N = w.N;
        const float* qkv_b = p.qkv + (long)w.b * L * C3;
        const float* dy_b = p.dy + (long)w.b * L * p.C;
        const float* y_b = p.y_in + (long)w.b * L * p.C;
        // loads in the order of their use: q rows, this wave's K fragments and lse feed the S products, which run while the rest
        // (dO, y0, v) is still on its way
        int lrow[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * T, row = idx >> 3, c4 = idx & 7;
            qv[it] = {};
            bbits[it] = 0;
            lrow[it] = -1;
            if (row < N) {
                const int r = br.W_sp > 1 ? fdiv(row, br.m_Wsp) : row, c = row - r * br.W_sp;
                bbits[it] = (r >= 1 ? 1 : 0) | (r <= br.H_sp - 2 ? 2 : 0) | (c >= 1 ? 4 : 0) | (c <= br.W_sp - 2 ? 8 : 0) | 16;
                if (4 * c4 < p.hd) {
                    lrow[it] = (w.ih * br.H_sp + r) * p.reso + w.iw * br.W_sp + c;
                    qv[it] = ldq_raw<Q16>(p, qkv_b + (long)lrow[it] * C3 + ch0 + 4 * c4);
                }
            }
        }
        k0r = {};
        k1r = {};
        const int tk = 16 * wave + li;
        if (tk < N && 8 * kq < p.hd) {
            const float* src = qkv_b + (long)token_of(br, w, p.reso, tk) * C3 + p.C + ch0 + 8 * kq;
            k0r = ldq_raw<Q16>(p, src);
            k1r = ldq_raw<Q16>(p, src + 4);
        }
        lsev = INFINITY;                                     // +inf -> P = 0 on padded query rows
        if (tid < N) lsev = p.lse[((long)w.b * p.heads_total + br.head0 + w.g) * L + token_of(br, w, p.reso, tid)];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c4 = (tid + it * T) & 7;
            vv[it] = {};
            yv[it] = {};
            dv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lrow[it] >= 0) {
                dv[it] = *reinterpret_cast<const f32x4*>(dy_b + (long)lrow[it] * p.C + ch0 + 4 * c4);
                yv[it] = ldy_raw<Y16>(p, y_b + (long)lrow[it] * p.C + ch0 + 4 * c4);
                vv[it] = ldq_raw<Q16>(p, qkv_b + (long)lrow[it] * C3 + ch0 + 4 * c4 + 2 * p.C);
            }
        }
        {
            const int tap = tid / HD, ch = tid - tap * HD, cb = ch0 - br.c0 + ch;
            wl0 = (tid >= 10 * HD || ch >= p.hd) ? 0.f : (tap < 9 ? br.lepe_w[cb * 9 + tap] : br.lepe_b[cb]);
            const int i2 = tid + T, tap2 = i2 / HD, ch2 = i2 - tap2 * HD, cb2 = ch0 - br.c0 + ch2;
            wl1 = (i2 >= 10 * HD || ch2 >= p.hd) ? 0.f : (tap2 < 9 ? br.lepe_w[cb2 * 9 + tap2] : br.lepe_b[cb2]);
        }
    };

    ATTN_STAMP(0);
    issue(item);
    {
        const WgInfo w = decode_wg(p, item);
        const AttnBranch& br = p.br[w.bi];
        const int ch0 = br.c0 + w.g * p.hd, N = w.N;
        float* dqkv_b = p.dqkv + (long)w.b * L * C3;
        const bool thin = br.H_sp == 1 || br.W_sp == 1;       // wave-uniform
        const int kw = wave;                               // key tile owned by this wave
        const int tk = 16 * kw + li;                       // this lane's key token
        const bool kvalid = tk < N;
        const int lk = kvalid ? token_of(br, w, p.reso, tk) : 0;

        // ---- A1: Q image, lse, K fragments ----
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * T, row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<f32x4*>(&QK[row * LDT + 4 * c4]) = qcv(qv[it]);
        }
        if (tid < NP) lse_s[tid] = lsev;
        float kf[8], vf[8];
        {
            const f32x4 k0 = qcv(k0r), k1 = qcv(k1r);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kf[e] = k0[e];
                kf[4 + e] = k1[e];
            }
        }
        lds_barrier();
        ATTN_STAMP(1);
        // ---- S tiles of all query tiles x this wave's keys (MFMA): they need q and k only, and run in the shadow of the dO / y0 /
        // v loads (at kernel start every workgroup of the launch waits for its loads at the same time: nothing else hides them)
        const attn_bf16x8 kb = pk8(f32x4{kf[0], kf[1], kf[2], kf[3]}, f32x4{kf[4], kf[5], kf[6], kf[7]});       // M16 only
        f32x4 Sq[NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            const float* qp = &QK[(16 * qt + li) * LDT + 8 * kq];
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qp), q1 = *reinterpret_cast<const f32x4*>(qp + 4);
            f32x4 sa = {0.f, 0.f, 0.f, 0.f};
            if constexpr (M16) {
                sa = mfma32(pk8(q0, q1), kb, sa);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) sa = mfma4(q0[e], kf[e], sa);          // S[q][key] = sum_d Q[q][d] K[key][d]
#pragma unroll
                for (int e = 0; e < 4; ++e) sa = mfma4(q1[e], kf[4 + e], sa);
            }
            Sq[qt] = sa;
        }

        // ---- A2: dO, V images; delta; border bits; zero row; LePE taps; bias gradient ----
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * T, row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<f32x4*>(&VS[row * LDT + 4 * c4]) = qcv(vv[it]);
            *reinterpret_cast<f32x4*>(&Ds[row * LDT + 4 * c4]) = dv[it];
            const f32x4 y4 = qcv(yv[it]);
            const float part = oct_sum(dv[it][0] * y4[0] + dv[it][1] * y4[1] + dv[it][2] * y4[2] + dv[it][3] * y4[3]);
            if (c4 == 0) {
                del_s[row] = part;                          // rows >= N: 0
                reinterpret_cast<int*>(Ds)[row * LDT + HD] = bbits[it];
            }
        }
        if (tid < 8) *reinterpret_cast<f32x4*>(&VS[NP * LDT + 4 * tid]) = f32x4{0.f, 0.f, 0.f, 0.f};      // the zero row behind the V image
        {
            // LePE bias gradient = sum of dO over the window's tokens: this thread's two rows share its channel chunk (lane & 7); the
            // wave's eight row groups meet by DPP / swaps, the waves in the [NT][HD] patch behind the taps (summed when stored)
            f32x4 bs = dv[0] + dv[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bs[e];
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
                bs[e] = xor32_sum(xor16_sum(v));
            }
            if (lane < 8) *reinterpret_cast<f32x4*>(&Gl[(9 + wave) * HD + 4 * lane]) = bs;
        }
        if (tid < 10 * HD) Wl[tid] = wl0;
        if (tid + T < 10 * HD) Wl[tid + T] = wl1;
        lds_barrier();
        ATTN_STAMP(2);
        {
            const float* vp = &VS[tk * LDT + 8 * kq];
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(vp), v1 = *reinterpret_cast<const f32x4*>(vp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vf[e] = v0[e];
                vf[4 + e] = v1[e];
            }
        }

        // ---- B: LePE conv weight gradient of this (window, head): dW[tap][ch] = sum_t dO[t][ch] V[nbr(t, tap)][ch] ----
        // The live taps (all nine, or the three along a one-token-wide stripe) are dealt to the waves round robin.  A lane =
        // (token 8 i + j, 16-B channel chunk c4), j = lane >> 3: per step i the wave covers eight consecutive image rows, so every
        // LDS address is a lane constant plus a compile-time offset.  Whether the neighbour (r + ky - 1, c + kx - 1) lies inside
        // the window comes from the token's border bits (pad column of the dO image, written while staging); a neighbour outside
        // reads the zero row instead.  The step loop is branch-free and its reads are issued in batches: as a loop of dependent
        // ds_read -> wait -> fma steps this phase took 14 - 21 k cycles per unit.
        {
            const int c4 = lane & 7, tj = lane >> 3;
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            constexpr int NTAP = (9 + NT - 1) / NT;              // taps per wave (at most)
            const int nlive = thin ? 3 : 9;
            int ntap = 0;                                        // wave-uniform
            int off[NTAP], need[NTAP], tapid[NTAP];
#pragma unroll
            for (int k = 0; k < NTAP; ++k) {
                const int j = wv + k * NT;                       // index into the live taps
                const int tap = !thin ? j : (br.H_sp == 1 ? 3 + j : 3 * j + 1);
                const int ky = tap / 3, kx = tap - 3 * ky;
                tapid[k] = tap;
                off[k] = ((ky - 1) * br.W_sp + (kx - 1)) * LDT;
                need[k] = (ky == 0 ? 1 : 0) | (ky == 2 ? 2 : 0) | (kx == 0 ? 4 : 0) | (kx == 2 ? 8 : 0) | 16;
                if (j < nlive) ntap = k + 1;
            }
            const float* gbase = &Ds[tj * LDT + 4 * c4];
            const float* vbase = &VS[tj * LDT + 4 * c4];
            const float* zrow = &VS[NP * LDT + 4 * c4];
            const int* mbase = reinterpret_cast<const int*>(&Ds[tj * LDT + HD]);
            f32x4 acc[NTAP];
#pragma unroll
            for (int k = 0; k < NTAP; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto taps = [&](auto kc) {
                constexpr int KC = decltype(kc)::value;
                const float* gp = gbase;
                const float* vp0 = vbase;
                const int* mp = mbase;
#pragma unroll 1
                for (int i0 = 0; i0 < 2 * NT; i0 += 2) {         // two steps at a time
                    f32x4 g4[2], v4[2][KC];
                    int m[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        g4[u] = *reinterpret_cast<const f32x4*>(gp + 8 * u * LDT);
                        m[u] = mp[8 * u * LDT];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int k = 0; k < KC; ++k) {
                            const bool ok = (m[u] & need[k]) == need[k];
                            v4[u][k] = *reinterpret_cast<const f32x4*>(ok ? vp0 + 8 * u * LDT + off[k] : zrow);
                        }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int k = 0; k < KC; ++k) acc[k] += g4[u] * v4[u][k];
                    gp += 16 * LDT;
                    vp0 += 16 * LDT;
                    mp += 16 * LDT;
                }
            };
            if (ntap == 1) taps(std::integral_constant<int, 1>{});
            else if (NTAP >= 2 && ntap == 2) taps(std::integral_constant<int, (NTAP >= 2 ? 2 : 1)>{});
            else if (NTAP >= 3 && ntap == 3) taps(std::integral_constant<int, (NTAP >= 3 ? 3 : 1)>{});
            // the eight token slots (lanes that share lane & 7) meet: DPP inside the row, row / half swaps across
#pragma unroll
            for (int k = 0; k < NTAP; ++k) {
                if (k >= ntap) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[k][e];
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
                    acc[k][e] = xor32_sum(xor16_sum(v));
                }
                if (lane < 8) *reinterpret_cast<f32x4*>(&Gl[tapid[k] * HD + 4 * lane]) = acc[k];
            }
            if (thin && wv == 3 % NT) {                          // the six taps that never fall inside a thin stripe
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const bool dead = br.H_sp == 1 ? ky != 1 : kx != 1;
                    if (dead && lane < 8) *reinterpret_cast<f32x4*>(&Gl[tap * HD + 4 * lane]) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        ATTN_STAMP(3);
        lds_barrier();                                      // V image dead (VS becomes the dS image); the unit's LePE gradient is complete
        {
            // partial-slab row of (b, window): columns [channel of the branch][tap] then [bias]; this head's 9 hd + hd values are
            // contiguous, so thread i stores element i (coalesced) from the [tap][d] patch
            const int cb = br.heads * p.hd;
            float* row = br.dw_part + ((long)w.b * br.nWin + w.win) * p.slab_rows * (cb * 10);
#pragma unroll
            for (int i = tid; i < 10 * HD; i += T) {
                if (i < 9 * p.hd) {
                    const int d = i / 9, tap = i - 9 * d;
                    row[w.g * p.hd * 9 + i] = Gl[tap * HD + d];
                } else if (i >= 9 * HD && i < 9 * HD + p.hd) {
                    float bsum = 0.f;
#pragma unroll
                    for (int k = 0; k < NT; ++k) bsum += Gl[(9 + k) * HD + (i - 9 * HD)];
                    row[cb * 9 + w.g * p.hd + (i - 9 * HD)] = bsum;
                }
            }
        }

        // ---- C: fused S / dP -> P, dS -> dV^T, dK^T ----
        f32x4 dVt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 dKt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const bool colok = 16 * kw + li < S;               // dS columns beyond the stride would land in the next row
        const attn_bf16x8 vb = pk8(f32x4{vf[0], vf[1], vf[2], vf[3]}, f32x4{vf[4], vf[5], vf[6], vf[7]});       // M16 only
        auto dp_tile = [&](int qt, f32x4& da) {
            const float* dp = &Ds[(16 * qt + li) * LDT + 8 * kq];
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
            da = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (M16) {
                da = mfma32(pk8(d0, d1), vb, da);
                return;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) da = mfma4(d0[e], vf[e], da);               // dP[q][key] = sum_d dO[q][d] V[key][d]
#pragma unroll
            for (int e = 0; e < 4; ++e) da = mfma4(d1[e], vf[4 + e], da);
        };
        unsigned keep = ~0u;                                // attention dropout only: bit 4 qt + r = keep (query 16 qt + 4 kq + r, key tk)
        if (p.drop_p > 0.f)
            keep = attn_keep_bits<4 * NT>(p, attn_unit_id(p, br, w.b, w.g, w.win), N,
                                          [&](int i, int& q_, int& k_) { q_ = 16 * (i >> 2) + 4 * kq + (i & 3); k_ = tk; });
        {
            f32x4 da;
            dp_tile(0, da);
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                f32x4 dn = da;
                if (qt + 1 < NT) dp_tile(qt + 1, dn);
                const f32x4 sa = Sq[qt];
                const f32x4 ls = *reinterpret_cast<const f32x4*>(&lse_s[16 * qt + 4 * kq]);
                const f32x4 de = *reinterpret_cast<const f32x4*>(&del_s[16 * qt + 4 * kq]);
                f32x4 pr, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr[r] = kvalid ? __expf(sa[r] * p.scale - ls[r]) : 0.f;
                    ds[r] = pr[r] * (da[r] - de[r]);
                }
                if (p.drop_p > 0.f) {                      // dS = P o (dP o M - delta); dV below takes P o M (pr is not used after it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m = (keep >> (4 * qt + r)) & 1u ? p.drop_scale : 0.f;
                        ds[r] = pr[r] * (da[r] * m - de[r]);
                        pr[r] *= m;
                    }
                }
                const int q0row = 16 * qt + 4 * kq;
                if constexpr (M16) {
                    const float* dop = &Ds[q0row * LDT + li];
                    const float* qp = &QK[q0row * LDT + li];
                    const attn_s16x4 prb = pk4(pr), dsb = pk4(ds);
                    dVt[0] = mfma16(pk4(f32x4{dop[0], dop[LDT], dop[2 * LDT], dop[3 * LDT]}), prb, dVt[0]);
                    dVt[1] = mfma16(pk4(f32x4{dop[16], dop[LDT + 16], dop[2 * LDT + 16], dop[3 * LDT + 16]}), prb, dVt[1]);
                    dKt[0] = mfma16(pk4(f32x4{qp[0], qp[LDT], qp[2 * LDT], qp[3 * LDT]}), dsb, dKt[0]);
                    dKt[1] = mfma16(pk4(f32x4{qp[16], qp[LDT + 16], qp[2 * LDT + 16], qp[3 * LDT + 16]}), dsb, dKt[1]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* dop = &Ds[(q0row + r) * LDT + li];
                        const float* qp = &QK[(q0row + r) * LDT + li];
                        dVt[0] = mfma4(dop[0], pr[r], dVt[0]);
                        dVt[1] = mfma4(dop[16], pr[r], dVt[1]);
                        dKt[0] = mfma4(qp[0], ds[r], dKt[0]);
                        dKt[1] = mfma4(qp[16], ds[r], dKt[1]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (colok && q0row + r < N) VS[(q0row + r) * S + 16 * kw + li] = ds[r];
                da = dn;
            }
        }
        // lane holds dV^T / dK^T [d = 16 df + 4 kq + e][key = tk]: add LePE^T(dO) to dV and store
        if (kvalid) {
            const int rr = thin ? 0 : fdiv(tk, br.m_Wsp), cc = thin ? 0 : tk - rr * br.W_sp;          // thin stripes: unused
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                const int d0 = 16 * df + 4 * kq;
                const f32x4 acc = thin ? lepe_taps4<true, -1>(br, Ds, Wl, rr, cc, tk, d0, dVt[df])
                                       : lepe_taps4<false, -1>(br, Ds, Wl, rr, cc, tk, d0, dVt[df]);
                if (d0 < p.hd) {
                    float* dst = dqkv_b + (long)lk * C3 + ch0 + d0;
                    stdq<Q16>(p, dst + p.C, dKt[df] * p.scale);
                    stdq<Q16>(p, dst + 2 * p.C, acc);
                }
            }
        }
        ATTN_STAMP(4);
        lds_barrier();                                      // dS complete; Q image dead

        // ---- D: K image over Q, then dQ ----
        {
            float* kp = &QK[tk * LDT + 8 * kq];
            *reinterpret_cast<f32x4*>(kp) = f32x4{kf[0], kf[1], kf[2], kf[3]};
            *reinterpret_cast<f32x4*>(kp + 4) = f32x4{kf[4], kf[5], kf[6], kf[7]};
        }
        lds_barrier();
        ATTN_STAMP(5);
        {
            const int qt = wave;
            const int tq = 16 * qt + li;
            const int qrow = min(tq, N - 1);                // padded queries re-read the last row (their columns are discarded)
            f32x4 dQt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                // columns beyond the stride read the head of the next row / the zeroed tail (finite) against K rows that are zero
                const f32x4 ds = *reinterpret_cast<const f32x4*>(&VS[qrow * S + 16 * kt + 4 * kq]);
                if constexpr (M16) {
                    const float* kp = &QK[(16 * kt + 4 * kq) * LDT + li];
                    const attn_s16x4 dsb = pk4(ds);
                    dQt[0] = mfma16(pk4(f32x4{kp[0], kp[LDT], kp[2 * LDT], kp[3 * LDT]}), dsb, dQt[0]);
                    dQt[1] = mfma16(pk4(f32x4{kp[16], kp[LDT + 16], kp[2 * LDT + 16], kp[3 * LDT + 16]}), dsb, dQt[1]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* kp = &QK[(16 * kt + 4 * kq + r) * LDT + li];
                        dQt[0] = mfma4(kp[0], ds[r], dQt[0]);
                        dQt[1] = mfma4(kp[16], ds[r], dQt[1]);
                    }
                }
            }
            if (tq < N) {
                float* dst = dqkv_b + (long)token_of(br, w, p.reso, tq) * C3 + ch0 + 4 * kq;
                if (4 * kq < p.hd) stdq<Q16>(p, dst, dQt[0] * p.scale);
                if (16 + 4 * kq < p.hd) stdq<Q16>(p, dst + 16, dQt[1] * p.scale);
            }
        }
        ATTN_STAMP(6);
    }
}

// =====================================================================================
// backward for windows larger than 112 tokens (384x384 inputs: N = 144, 288): two-pass, any N
// =====================================================================================
// The fused kernel above keeps Q, K, V, dO and the N x N dS of a window in LDS; beyond N = 112 that does not fit.  The
// large-window path never holds more than 64 x 64 of anything:
//   attn_delta_kernel   delta[q] = sum_d dO[q][d] (y[q][d] - lepe[q][d])      (= rowsum(P o dP), from the saved output)
//   attn_bwd_kv_kernel  one workgroup per 64 keys: dK, dV complete in registers over all query chunks (+ LePE^T(dO))
//   attn_bwd_q_kernel   one workgroup per 64 queries: dQ complete in registers over all key chunks (S, dP recomputed)
//   lepe_wgrad_kernel   depthwise-conv weight/bias gradient partial slabs from global memory
// 56 MFMAs per 16x16 tile pair instead of 40, no cross-workgroup reduction, deterministic.

struct BigWg { int bi, b, win, g, blk, ih, iw, N; };

__device__ __forceinline__ BigWg decode_big(const AttnParams& p, int wg, int nblk) {
    BigWg w;
    w.blk = wg % nblk;
    const WgInfo i = decode_wg(p, wg / nblk);
    w.bi = i.bi; w.b = i.b; w.win = i.win; w.g = i.g; w.ih = i.ih; w.iw = i.iw; w.N = i.N;
    return w;
}

__device__ __forceinline__ int token_of2(const AttnBranch& br, int ih, int iw, int reso, int t) {
    const int r = br.W_sp > 1 ? fdiv(t, br.m_Wsp) : t, c = t - r * br.W_sp;
    return (ih * br.H_sp + r) * reso + iw * br.W_sp + c;
}

// grid: ceil(B * L * heads_total / 32) blocks of 256 threads; 8 lanes per (token, head)
template <int ST>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
    constexpr bool Q16 = (ST & 1) != 0, Y16 = (ST & 2) != 0, M16 = (ST & 4) != 0;      // storage of qkv / dqkv and of y, bf16 MFMAs (see AttnParams)
    (void)Q16; (void)Y16; (void)M16;
    const int L = p.reso * p.reso;
    const long item = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int j = threadIdx.x & 7;
    const long total = (long)p.B * p.heads_total * L;
    float part = 0.f;
    long out_idx = -1;
    if (item < total) {
        const int l = (int)(item % L);
        const int hg = (int)((item / L) % p.heads_total);
        const int b = (int)(item / ((long)L * p.heads_total));
        const int bi = (p.nbranch > 1 && hg >= p.br[1].head0) ? 1 : 0;
        const AttnBranch& br = p.br[bi];
        const int g = hg - br.head0;
        const int ch0 = br.c0 + g * p.hd + 4 * j;
        if (4 * j < p.hd) {
            const f32x4 yv = qcv(ldy_raw<Y16>(p, p.y_in + ((long)b * L + l) * p.C + ch0));
            const f32x4 dv = *reinterpret_cast<const f32x4*>(p.dy + ((long)b * L + l) * p.C + ch0);
#pragma unroll
            for (int e = 0; e < 4; ++e) part += dv[e] * yv[e];
        }
        out_idx = ((long)b * p.heads_total + hg) * L + l;
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (j == 0 && out_idx >= 0) p.delta[out_idx] = part;
}

// one workgroup (4 waves) = 64 keys of one (branch, window, head); wave w owns keys [k0 + 16 w, +16)
template <int ST>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnParams p, int nblk) {
    constexpr bool Q16 = (ST & 1) != 0, Y16 = (ST & 2) != 0, M16 = (ST & 4) != 0;      // storage of qkv / dqkv and of y, bf16 MFMAs (see AttnParams)
    (void)Q16; (void)Y16; (void)M16;
    __shared__ __attribute__((aligned(16))) float Qc[64 * LDT];
    __shared__ __attribute__((aligned(16))) float Dc[64 * LDT];
    __shared__ __attribute__((aligned(16))) float lse_c[64];
    __shared__ __attribute__((aligned(16))) float del_c[64];
    const BigWg w = decode_big(p, blockIdx.x, nblk);
    const AttnBranch& br = p.br[w.bi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int L = p.reso * p.reso, C3 = 3 * p.C, N = w.N;
    const int ch0 = br.c0 + w.g * p.hd;
    const float* qkv_b = p.qkv + (long)w.b * L * C3;
    const float* dy_b = p.dy + (long)w.b * L * p.C;
    float* dqkv_b = p.dqkv + (long)w.b * L * C3;
    const long stat_base = ((long)w.b * p.heads_total + br.head0 + w.g) * L;
    const int tk = 64 * w.blk + 16 * wave + li;
    const bool kvalid = tk < N;
    const int lk = kvalid ? token_of2(br, w.ih, w.iw, p.reso, tk) : 0;
    float kf[8], vf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) kf[e] = vf[e] = 0.f;
    if (kvalid && 8 * kq < p.hd) {
        const float* src = qkv_b + (long)lk * C3 + ch0 + 8 * kq;
        const auto k0r = ldq_raw<Q16>(p, src + p.C), k1r = ldq_raw<Q16>(p, src + p.C + 4);
        const auto v0r = ldq_raw<Q16>(p, src + 2 * p.C), v1r = ldq_raw<Q16>(p, src + 2 * p.C + 4);
        const f32x4 k0 = qcv(k0r), k1 = qcv(k1r), v0 = qcv(v0r), v1 = qcv(v1r);
#pragma unroll
        for (int e = 0; e < 4; ++e) { kf[e] = k0[e]; kf[4 + e] = k1[e]; vf[e] = v0[e]; vf[4 + e] = v1[e]; }
    }
    f32x4 dVt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 dKt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const attn_bf16x8 kb = pk8(f32x4{kf[0], kf[1], kf[2], kf[3]}, f32x4{kf[4], kf[5], kf[6], kf[7]});       // M16 only
    const attn_bf16x8 vb = pk8(f32x4{vf[0], vf[1], vf[2], vf[3]}, f32x4{vf[4], vf[5], vf[6], vf[7]});
    for (int q0 = 0; q0 < N; q0 += 64) {
        for (int idx = tid; idx < 64 * 8; idx += 256) {
            const int row = idx >> 3, c4 = idx & 7, tq = q0 + row;
            typename QRaw<Q16>::type qv = {};
            f32x4 dv = {0.f, 0.f, 0.f, 0.f};
            if (tq < N && 4 * c4 < p.hd) {
                const int l = token_of2(br, w.ih, w.iw, p.reso, tq);
                qv = ldq_raw<Q16>(p, qkv_b + (long)l * C3 + ch0 + 4 * c4);
                dv = *reinterpret_cast<const f32x4*>(dy_b + (long)l * p.C + ch0 + 4 * c4);
            }
            *reinterpret_cast<f32x4*>(&Qc[row * LDT + 4 * c4]) = qcv(qv);
            *reinterpret_cast<f32x4*>(&Dc[row * LDT + 4 * c4]) = dv;
        }
        if (tid < 128) {
            const int row = tid & 63, tq = q0 + row;
            const bool ok = tq < N;
            const int l = ok ? token_of2(br, w.ih, w.iw, p.reso, tq) : 0;
            if (tid < 64) lse_c[row] = ok ? p.lse[stat_base + l] : INFINITY;
            else del_c[row] = ok ? p.delta[stat_base + l] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const float* qp = &Qc[(16 * qt + li) * LDT + 8 * kq];
            const float* dp = &Dc[(16 * qt + li) * LDT + 8 * kq];
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = sa;
            if constexpr (M16) {
                sa = mfma32(pk8(*reinterpret_cast<const f32x4*>(qp), *reinterpret_cast<const f32x4*>(qp + 4)), kb, sa);
                da = mfma32(pk8(*reinterpret_cast<const f32x4*>(dp), *reinterpret_cast<const f32x4*>(dp + 4)), vb, da);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sa = mfma4(qp[e], kf[e], sa);
                    da = mfma4(dp[e], vf[e], da);
                }
            }
            const f32x4 ls = *reinterpret_cast<const f32x4*>(&lse_c[16 * qt + 4 * kq]);
            const f32x4 de = *reinterpret_cast<const f32x4*>(&del_c[16 * qt + 4 * kq]);
            f32x4 pv, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = kvalid ? __expf(sa[r] * p.scale - ls[r]) : 0.f;
                ds[r] = pv[r] * (da[r] - de[r]);
            }
            if (p.drop_p > 0.f) {
                const long uid = attn_unit_id(p, br, w.b, w.g, w.win);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m = attn_keep(p, uid, N, q0 + 16 * qt + 4 * kq + r, tk);
                    ds[r] = pv[r] * (da[r] * m - de[r]);
                    pv[r] *= m;
                }
            }
            if constexpr (M16) {
                const float* dop = &Dc[(16 * qt + 4 * kq) * LDT + li];
                const float* qq = &Qc[(16 * qt + 4 * kq) * LDT + li];
                const attn_s16x4 pvb = pk4(pv), dsb = pk4(ds);
                dVt[0] = mfma16(pk4(f32x4{dop[0], dop[LDT], dop[2 * LDT], dop[3 * LDT]}), pvb, dVt[0]);
                dVt[1] = mfma16(pk4(f32x4{dop[16], dop[LDT + 16], dop[2 * LDT + 16], dop[3 * LDT + 16]}), pvb, dVt[1]);
                dKt[0] = mfma16(pk4(f32x4{qq[0], qq[LDT], qq[2 * LDT], qq[3 * LDT]}), dsb, dKt[0]);
                dKt[1] = mfma16(pk4(f32x4{qq[16], qq[LDT + 16], qq[2 * LDT + 16], qq[3 * LDT + 16]}), dsb, dKt[1]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qrow = 16 * qt + 4 * kq + r;
                    const float* dop = &Dc[qrow * LDT + li];
                    const float* qq = &Qc[qrow * LDT + li];
                    dVt[0] = mfma4(dop[0], pv[r], dVt[0]);
                    dVt[1] = mfma4(dop[16], pv[r], dVt[1]);
                    dKt[0] = mfma4(qq[0], ds[r], dKt[0]);
                    dKt[1] = mfma4(qq[16], ds[r], dKt[1]);
                }
            }
        }
        __syncthreads();
    }
    if (kvalid) {
        const int rr = br.W_sp > 1 ? fdiv(tk, br.m_Wsp) : tk, cc = tk - rr * br.W_sp;
#pragma unroll
        for (int df = 0; df < 2; ++df) {
            const int d0 = 16 * df + 4 * kq, cb = w.g * p.hd + d0;
            f32x4 acc = dVt[df];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int r2 = rr - ky + 1, c2 = cc - kx + 1;
                    if (d0 < p.hd && (unsigned)r2 < (unsigned)br.H_sp && (unsigned)c2 < (unsigned)br.W_sp) {
                        const int l2 = token_of2(br, w.ih, w.iw, p.reso, r2 * br.W_sp + c2);
                        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy_b + (long)l2 * p.C + ch0 + d0);
                        const int tap = ky * 3 + kx;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] += br.lepe_w[(cb + e) * 9 + tap] * dv[e];
                    }
                }
            float* dst = dqkv_b + (long)lk * C3 + ch0 + d0;
            if (d0 < p.hd) {
                stdq<Q16>(p, dst + p.C, dKt[df] * p.scale);
                stdq<Q16>(p, dst + 2 * p.C, acc);
            }
        }
    }
}

// one workgroup (4 waves) = 64 queries of one (branch, window, head); wave w owns queries [q0 + 16 w, +16)
template <int ST>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnParams p, int nblk) {
    constexpr bool Q16 = (ST & 1) != 0, Y16 = (ST & 2) != 0, M16 = (ST & 4) != 0;      // storage of qkv / dqkv and of y, bf16 MFMAs (see AttnParams)
    (void)Q16; (void)Y16; (void)M16;
    __shared__ __attribute__((aligned(16))) float Kc[64 * LDT];
    __shared__ __attribute__((aligned(16))) float Vc[64 * LDT];
    const BigWg w = decode_big(p, blockIdx.x, nblk);
    const AttnBranch& br = p.br[w.bi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int L = p.reso * p.reso, C3 = 3 * p.C, N = w.N;
    const int ch0 = br.c0 + w.g * p.hd;
    const float* qkv_b = p.qkv + (long)w.b * L * C3;
    const float* dy_b = p.dy + (long)w.b * L * p.C;
    float* dqkv_b = p.dqkv + (long)w.b * L * C3;
    const long stat_base = ((long)w.b * p.heads_total + br.head0 + w.g) * L;
    const int tq = 64 * w.blk + 16 * wave + li;
    const bool qvalid = tq < N;
    const int lq = qvalid ? token_of2(br, w.ih, w.iw, p.reso, tq) : 0;
    float qr[8], dor[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] = dor[e] = 0.f;
    float lse_q = INFINITY, del_q = 0.f;
    if (qvalid) {
        lse_q = p.lse[stat_base + lq];
        del_q = p.delta[stat_base + lq];
    }
    if (qvalid && 8 * kq < p.hd) {
        const float* src = qkv_b + (long)lq * C3 + ch0 + 8 * kq;
        const auto q0r = ldq_raw<Q16>(p, src), q1r = ldq_raw<Q16>(p, src + 4);
        const float* dsrc = dy_b + (long)lq * p.C + ch0 + 8 * kq;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(dsrc), d1 = *reinterpret_cast<const f32x4*>(dsrc + 4);
        const f32x4 q0 = qcv(q0r), q1 = qcv(q1r);
#pragma unroll
        for (int e = 0; e < 4; ++e) { qr[e] = q0[e] * p.scale; qr[4 + e] = q1[e] * p.scale; dor[e] = d0[e]; dor[4 + e] = d1[e]; }
    }
    f32x4 dQt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const long uid = attn_unit_id(p, br, w.b, w.g, w.win);      // attention dropout only
    const attn_bf16x8 qb = pk8(f32x4{qr[0], qr[1], qr[2], qr[3]}, f32x4{qr[4], qr[5], qr[6], qr[7]});       // M16 only
    const attn_bf16x8 dob = pk8(f32x4{dor[0], dor[1], dor[2], dor[3]}, f32x4{dor[4], dor[5], dor[6], dor[7]});
    for (int k0 = 0; k0 < N; k0 += 64) {
        for (int idx = tid; idx < 64 * 8; idx += 256) {
            const int row = idx >> 3, c4 = idx & 7, tk = k0 + row;
            typename QRaw<Q16>::type kv = {}, vv = {};
            if (tk < N && 4 * c4 < p.hd) {
                const float* src = qkv_b + (long)token_of2(br, w.ih, w.iw, p.reso, tk) * C3 + ch0 + 4 * c4;
                kv = ldq_raw<Q16>(p, src + p.C);
                vv = ldq_raw<Q16>(p, src + 2 * p.C);
            }
            *reinterpret_cast<f32x4*>(&Kc[row * LDT + 4 * c4]) = qcv(kv);
            *reinterpret_cast<f32x4*>(&Vc[row * LDT + 4 * c4]) = qcv(vv);
        }
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float* kp = &Kc[(16 * kt + li) * LDT + 8 * kq];
            const float* vp = &Vc[(16 * kt + li) * LDT + 8 * kq];
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = sa;       // S^T / dP^T tiles: rows = keys, col = this lane's query
            if constexpr (M16) {
                sa = mfma32(pk8(*reinterpret_cast<const f32x4*>(kp), *reinterpret_cast<const f32x4*>(kp + 4)), qb, sa);
                da = mfma32(pk8(*reinterpret_cast<const f32x4*>(vp), *reinterpret_cast<const f32x4*>(vp + 4)), dob, da);
                f32x4 ds4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kok = k0 + 16 * kt + 4 * kq + r < N;
                    const float pv = kok ? __expf(sa[r] - lse_q) : 0.f;
                    const float m = p.drop_p > 0.f ? attn_keep(p, uid, N, tq, k0 + 16 * kt + 4 * kq + r) : 1.f;
                    ds4[r] = pv * (da[r] * m - del_q);
                }
                const float* kk = &Kc[(16 * kt + 4 * kq) * LDT + li];
                const attn_s16x4 dsb = pk4(ds4);
                dQt[0] = mfma16(pk4(f32x4{kk[0], kk[LDT], kk[2 * LDT], kk[3 * LDT]}), dsb, dQt[0]);
                dQt[1] = mfma16(pk4(f32x4{kk[16], kk[LDT + 16], kk[2 * LDT + 16], kk[3 * LDT + 16]}), dsb, dQt[1]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sa = mfma4(kp[e], qr[e], sa);
                    da = mfma4(vp[e], dor[e], da);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kok = k0 + 16 * kt + 4 * kq + r < N;
                    const float pv = kok ? __expf(sa[r] - lse_q) : 0.f;
                    const float m = p.drop_p > 0.f ? attn_keep(p, uid, N, tq, k0 + 16 * kt + 4 * kq + r) : 1.f;
                    const float ds = pv * (da[r] * m - del_q);
                    const float* kk = &Kc[(16 * kt + 4 * kq + r) * LDT + li];
                    dQt[0] = mfma4(kk[0], ds, dQt[0]);
                    dQt[1] = mfma4(kk[16], ds, dQt[1]);
                }
            }
        }
        __syncthreads();
    }
    if (qvalid) {
        float* dst = dqkv_b + (long)lq * C3 + ch0 + 4 * kq;
        if (4 * kq < p.hd) stdq<Q16>(p, dst, dQt[0] * p.scale);
        if (16 + 4 * kq < p.hd) stdq<Q16>(p, dst + 16, dQt[1] * p.scale);
    }
}

// LePE conv weight / bias gradient partial slabs from global memory (large-window path).  One workgroup per (branch, window,
// head, token slice): LW_SUB slices per window, each with its own slab row.  A thread owns ONE output group -- (tap, 16-B
// channel chunk): tid & 7 = chunk, (tid >> 3) & 15 = tap slot (10 of 16 used; tap 9 = bias) -- and walks the tokens of its half
// of the slice, so nothing is reduced across lanes; the two halves meet in LDS.  v is re-read nine times by its own window
// only (L1 / L2 resident).  (The first version -- one thread per (token slot, channel), scalar loads -- took 69 us per launch
// at 384 x 384, 8 % of that step.)
constexpr int LW_SUB = 4;
template <int ST>
__global__ __launch_bounds__(256) void lepe_wgrad_kernel(AttnParams p) {
    constexpr bool Q16 = (ST & 1) != 0, Y16 = (ST & 2) != 0, M16 = (ST & 4) != 0;      // storage of qkv / dqkv and of y, bf16 MFMAs (see AttnParams)
    (void)Q16; (void)Y16; (void)M16;
    __shared__ __attribute__((aligned(16))) float red[16 * 8 * 4];
    const int sub = (int)blockIdx.x % LW_SUB;
    const WgInfo w = decode_wg(p, (int)blockIdx.x / LW_SUB);
    const AttnBranch& br = p.br[w.bi];
    const int tid = threadIdx.x;
    const int c4 = tid & 7, tap = (tid >> 3) & 15, half = tid >> 7;
    const int L = p.reso * p.reso, C3 = 3 * p.C, N = w.N;
    const int ch0 = br.c0 + w.g * p.hd;
    const bool live = tap < 10 && 4 * c4 < p.hd;
    const float* v_b = p.qkv + (long)w.b * L * C3 + 2 * p.C + ch0 + 4 * c4;
    const float* dy_b = p.dy + (long)w.b * L * p.C + ch0 + 4 * c4;
    const int ky = tap / 3, kx = tap - ky * 3;            // (tap 9: ky = 3, unused)
    const int n_part = (N + 2 * LW_SUB - 1) / (2 * LW_SUB);
    const int t0 = min(N, (2 * sub + half) * n_part), t1 = min(N, t0 + n_part);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll 4
        for (int t = t0; t < t1; ++t) {
            const int rr = br.W_sp > 1 ? fdiv(t, br.m_Wsp) : t, cc = t - rr * br.W_sp;
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(dy_b + (long)token_of(br, w, p.reso, t) * p.C);
            if (tap == 9) {
                acc += g4;
            } else {
                const int r2 = rr + ky - 1, c2 = cc + kx - 1;
                if ((unsigned)r2 < (unsigned)br.H_sp && (unsigned)c2 < (unsigned)br.W_sp)
                    acc += g4 * ldq<Q16>(p, v_b + (long)token_of(br, w, p.reso, r2 * br.W_sp + c2) * C3);
            }
        }
    }
    if (half == 1) *reinterpret_cast<f32x4*>(&red[(tap * 8 + c4) * 4]) = acc;
    __syncthreads();
    if (half == 0 && live) {
        acc += *reinterpret_cast<const f32x4*>(&red[(tap * 8 + c4) * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) store_lepe_partial(p, br, w, sub, tap * HD + 4 * c4 + e, acc[e]);
    }
}

// =====================================================================================
// standalone index-only window gather / scatter (img2windows / windows2img, cswin_unet.py:184-202)
// =====================================================================================
__global__ void img2windows_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int C, int H, int W,
                                   int H_sp, int W_sp) {
    // img (B, C, H, W) -> out (B*nH*nW, H_sp*W_sp, C)
    const long total = (long)B * C * H * W;
    const int nW = W / W_sp, nH = H / H_sp, N = H_sp * W_sp;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        long t1 = o / C;
        const int t = (int)(t1 % N);
        long wi = t1 / N;
        const int iw = (int)(wi % nW);
        wi /= nW;
        const int ih = (int)(wi % nH);
        const int b = (int)(wi / nH);
        const int r = t / W_sp, cc = t - r * W_sp;
        out[o] = img[(((long)b * C + c) * H + ih * H_sp + r) * W + iw * W_sp + cc];
    }
}

__global__ void windows2img_kernel(const float* __restrict__ win, float* __restrict__ out, int B, int C, int H, int W,
                                   int H_sp, int W_sp) {
    // win (B*nH*nW, H_sp*W_sp, C) -> out (B, H, W, C)
    const long total = (long)B * C * H * W;
    const int nW = W / W_sp, nH = H / H_sp, N = H_sp * W_sp;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        long t1 = o / C;
        const int x = (int)(t1 % W);
        t1 /= W;
        const int yy = (int)(t1 % H);
        const int b = (int)(t1 / H);
        const int ih = yy / H_sp, r = yy - ih * H_sp, iw = x / W_sp, cc = x - iw * W_sp;
        out[o] = win[((((long)b * nH + ih) * nW + iw) * N + r * W_sp + cc) * C + c];
    }
}

struct HostBranch { int idx; int heads; const float* w; const float* b; float* dw; float* db; };

int fill_params(AttnParams& p, const char* who, int B, int reso, int C, int nbranch, const int* heads,
                const int* idx, int split, float scale, int* ntile, int* nwg) {
    CSWIN_REQUIRE(B > 0 && reso > 0 && C > 0 && (nbranch == 1 || nbranch == 2), CSWIN_ERR_SHAPE, "%s: bad arguments", who);
    const int Cb = C / nbranch;
    int heads_total = 0, wg = 0, N0 = -1;
    for (int i = 0; i < nbranch; ++i) {
        const int hd = heads[i] > 0 ? Cb / heads[i] : 0;
        CSWIN_REQUIRE(heads[i] > 0 && Cb == heads[i] * hd && hd >= 8 && hd <= HD && hd % 8 == 0 && (i == 0 || hd == p.hd),
                      CSWIN_ERR_UNSUPPORTED, "%s: head dim %d unsupported (8, 16, 24 or 32, equal in both branches)", who, hd);
        p.hd = hd;
        int H_sp, W_sp;
        if (idx[i] == -1) { H_sp = reso; W_sp = reso; }
        else if (idx[i] == 0) { H_sp = reso; W_sp = split; }
        else if (idx[i] == 1) { H_sp = split; W_sp = reso; }
        else { cswin_set_error("%s: ERROR MODE %d", who, idx[i]); return CSWIN_ERR_SHAPE; }
        CSWIN_REQUIRE(reso % H_sp == 0 && reso % W_sp == 0, CSWIN_ERR_SHAPE,
                      "%s: resolution %d not divisible by window %dx%d", who, reso, H_sp, W_sp);
        AttnBranch& br = p.br[i];
        br.c0 = i * Cb;
        br.heads = heads[i];
        br.head0 = heads_total;
        br.H_sp = H_sp; br.W_sp = W_sp;
        br.nW = reso / W_sp;
        br.nWin = (reso / H_sp) * (reso / W_sp);
        br.wg_begin = wg;
        br.m_heads = fdiv_magic(br.heads); br.m_nWin = fdiv_magic(br.nWin); br.m_nW = fdiv_magic(br.nW); br.m_Wsp = fdiv_magic(W_sp);
        CSWIN_REQUIRE((long)B * br.nWin * heads[i] < (1 << 20) && H_sp * W_sp < (1 << 12) && br.nWin < (1 << 12) && heads[i] < (1 << 12),
                      CSWIN_ERR_UNSUPPORTED, "%s: %ld units of %d tokens exceed the index arithmetic's range", who,
                      (long)B * br.nWin * heads[i], H_sp * W_sp);
        wg += B * br.nWin * heads[i];
        heads_total += heads[i];
        if (N0 < 0) N0 = H_sp * W_sp;
        CSWIN_REQUIRE(N0 == H_sp * W_sp, CSWIN_ERR_SHAPE, "%s: branches with different window sizes", who);
    }
    p.B = B; p.reso = reso; p.C = C; p.heads_total = heads_total; p.nbranch = nbranch;
    p.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)p.hd);
    *ntile = (N0 + 15) / 16;
    *nwg = wg;
    return CSWIN_OK;
}

template <int NT, int Q16>
int launch_fwd_q(const AttnParams& p, int nwg, hipStream_t st) {
    constexpr int NW = NT < 8 ? NT : 8;
    const size_t lds = (size_t)(2 * 16 * NT * LDT + 10 * HD) * sizeof(float);
    if (lds > 64 * 1024) {
        // dynamic-LDS opt-in of this instantiation: once per process, thread-safe, idempotent (a function attribute, not a
        // stream operation: it stays out of graph captures); the size is a constant of the template
        static std::once_flag once;
        static hipError_t status = hipSuccess;
        std::call_once(once, [&] {
            status = hipFuncSetAttribute((const void*)attn_fwd_kernel<NT, 1, Q16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (status == hipSuccess) status = hipFuncSetAttribute((const void*)attn_fwd_kernel<NT, 2, Q16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        if (status != hipSuccess) { cswin_set_error("attn_fwd: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(status)); return CSWIN_ERR_HIP; }
    }
    {
        // few units relative to the 256 CUs: split the query tiles over two workgroups per unit (see the kernel).  Large windows
        // (NT > 8: 384 x 384 inputs, 128 units at batch 8) otherwise leave half the chip idle and give a wave 2 - 3 query tiles.
        const int force = cswin_tuning().attn_fwd_qsplit;                                // tuning aid: 1 or 2
        const bool split = force ? force == 2 : (NT <= 8 ? (nwg < 1024 && nwg % 256 != 0 && NT >= 6)   // measured: pays at N = 98, not at N = 49
                                                         : nwg <= 256);
        if (split) {
            hipLaunchKernelGGL((attn_fwd_kernel<NT, 2, Q16>), dim3(2 * nwg), dim3(64 * ((NT + 1) / 2)), lds, st, p, nwg);
            return CSWIN_OK;
        }
    }
    hipLaunchKernelGGL((attn_fwd_kernel<NT, 1, Q16>), dim3(nwg), dim3(64 * NW), lds, st, p, nwg);
    return CSWIN_OK;
}

template <int NT>
int launch_fwd(const AttnParams& p, int nwg, hipStream_t st) {
    return p.qkv_bf16 == 7 ? launch_fwd_q<NT, 7>(p, nwg, st) : p.qkv_bf16 == 3 ? launch_fwd_q<NT, 3>(p, nwg, st)
         : p.qkv_bf16 ? launch_fwd_q<NT, 1>(p, nwg, st) : launch_fwd_q<NT, 0>(p, nwg, st);
}

// forward for windows of up to 128 tokens: items = units, or 2 x units with the query tiles split over two workgroups
template <int NT, int ST>
int launch_fwd3_q(const AttnParams& p, int nwg, hipStream_t st) {
    const size_t lds = (size_t)(2 * 16 * NT * LDT + 10 * HD) * sizeof(float);
    static_assert((2 * 16 * NT * LDT + 10 * HD) * sizeof(float) <= 64 * 1024, "no dynamic-LDS opt-in on this path");
    if constexpr (NT >= 6) {
        // few units relative to the CUs (stage 3: 384 units of 7 query tiles): split the query tiles over two workgroups per unit
        const int force = cswin_tuning().attn_fwd_qsplit;                            // tuning aid: 1 or 2
        if (force ? force == 2 : (nwg < 1024 && nwg % 256 != 0)) {
            hipLaunchKernelGGL((attn_fwd3_kernel<NT, 2, ST>), dim3(2 * nwg), dim3(64 * ((NT + 1) / 2)), lds, st, p, nwg);
            return CSWIN_OK;
        }
    }
    hipLaunchKernelGGL((attn_fwd3_kernel<NT, 1, ST>), dim3(nwg), dim3(64 * NT), lds, st, p, nwg);
    return CSWIN_OK;
}

template <int NT>
int launch_fwd3(const AttnParams& p, int nwg, hipStream_t st) {
    return p.qkv_bf16 == 7 ? launch_fwd3_q<NT, 7>(p, nwg, st) : p.qkv_bf16 == 3 ? launch_fwd3_q<NT, 3>(p, nwg, st)
         : p.qkv_bf16 ? launch_fwd3_q<NT, 1>(p, nwg, st) : launch_fwd3_q<NT, 0>(p, nwg, st);
}

inline int ds_stride_for(int N) {            // smallest stride >= N with stride = 4 (mod 8): 16-B aligned rows, and the four
    int s = (N + 3) / 4 * 4;                 // row groups of a dS column write land in distinct banks
    while (s % 8 != 4) s += 4;
    return s;
}

inline int vs_floats_for(int NT, int N, int S) {     // V image [16 NT + 1][LDT] (last row zero), later dS [N][S] + 16 NT finite floats behind it
    const int NP = 16 * NT;
    const int a = N * S + NP, b = (NP + 1) * LDT;
    return ((a > b ? a : b) + 3) / 4 * 4;
}

template <int NT, int ST>
int launch_bwd3_q(const AttnParams& p, int nwg, hipStream_t st) {
    constexpr int NP = 16 * NT;
    const size_t lds = (size_t)(2 * NP * LDT + p.vs_floats + 2 * NP + (19 + NT) * HD) * sizeof(float);
    if (lds > 64 * 1024) {
        // once per process and instantiation, thread-safe: reserve the largest footprint this NT can ask for (N = 16 NT tokens)
        static std::once_flag once;
        static hipError_t status = hipSuccess;
        constexpr int SMAX = ((NP + 3) / 4 * 4) + 8;
        constexpr size_t lds_max = (size_t)(2 * NP * LDT + (NP * SMAX + NP + 4 > (NP + 1) * LDT ? NP * SMAX + NP + 4 : (NP + 1) * LDT) + 2 * NP + (19 + NT) * HD) * sizeof(float);
        std::call_once(once, [&] {
            status = hipFuncSetAttribute((const void*)attn_bwd3_kernel<NT, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        });
        if (status != hipSuccess) { cswin_set_error("attn_bwd: cannot reserve %zu B LDS: %s", lds_max, hipGetErrorString(status)); return CSWIN_ERR_HIP; }
    }
    hipLaunchKernelGGL((attn_bwd3_kernel<NT, ST>), dim3(nwg), dim3(64 * NT), lds, st, p);
    return CSWIN_OK;
}

template <int NT>
int launch_bwd3(const AttnParams& p, int nwg, hipStream_t st) {
    return p.qkv_bf16 == 7 ? launch_bwd3_q<NT, 7>(p, nwg, st) : p.qkv_bf16 == 3 ? launch_bwd3_q<NT, 3>(p, nwg, st)
         : p.qkv_bf16 ? launch_bwd3_q<NT, 1>(p, nwg, st) : launch_bwd3_q<NT, 0>(p, nwg, st);
}

template <int Q16>
void launch_bwd_two_pass(const AttnParams& p, long items, int nwg, int nblk, hipStream_t st) {
    hipLaunchKernelGGL(attn_delta_kernel<Q16>, dim3((unsigned)((items * 8 + 255) / 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(attn_bwd_kv_kernel<Q16>, dim3(nwg * nblk), dim3(256), 0, st, p, nblk);
    hipLaunchKernelGGL(attn_bwd_q_kernel<Q16>, dim3(nwg * nblk), dim3(256), 0, st, p, nblk);
    hipLaunchKernelGGL(lepe_wgrad_kernel<Q16>, dim3(nwg * LW_SUB), dim3(256), 0, st, p);
}

long long* g_attn_stamps = nullptr;     // debug only (cswin_debug_set_attn_stamps)

inline bool force_two_pass() {          // tuning aid: the large-window backward for every window size
    const bool f = cswin_tuning().attn_bwd_two_pass != 0;
    return f;
}

}  // namespace

extern "C" {

// debug aid (not part of include/cswin_hip.h): device buffer [workgroups][8] of int64 stamped by wave 0 of the attention kernels
void cswin_debug_set_attn_stamps(void* p) { g_attn_stamps = (long long*)p; }

// qkv (B, L, 3C) -> y (B, L, C), lse (B, heads_total, L).  nbranch = 2: branch i uses channels
// [i*C/2, (i+1)*C/2) with stripe mode idx[i]; nbranch = 1: whole C, idx[0] (normally -1).
int cswin_attn_fwd(const float* qkv, const float* const* lepe_w, const float* const* lepe_b, float* y, float* y0, float* lse,
                   int B, int reso, int C, int nbranch, const int* heads, const int* idx, int split, float scale, float drop_p,
                   unsigned long long drop_seed, const unsigned long long* drop_epoch, int qkv_bf16, void* stream) {
    AttnParams p = {};
    CSWIN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, CSWIN_ERR_UNSUPPORTED, "attn_fwd: dropout probability %g outside [0, 1)", (double)drop_p);
    p.drop_p = drop_p; p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_thresh = (unsigned)(drop_p * 16777216.0f); p.drop_seed = drop_seed; p.drop_epoch = drop_epoch;
    CSWIN_REQUIRE(qkv_bf16 == 0 || qkv_bf16 == 1 || qkv_bf16 == 3 || qkv_bf16 == 7, CSWIN_ERR_UNSUPPORTED,
                  "attn: mode %d (0 = fp32, 1 = qkv / dqkv stored as bf16, 3 = also y, 7 = also bf16 MFMAs)", qkv_bf16);
    p.qkv_bf16 = qkv_bf16;
    int nt, nwg;
    int rc = fill_params(p, "attn_fwd", B, reso, C, nbranch, heads, idx, split, scale, &nt, &nwg);
    if (rc) return rc;
    CSWIN_REQUIRE(qkv && y && lse && lepe_w && lepe_b, CSWIN_ERR_SHAPE, "attn_fwd: null pointer");
    for (int i = 0; i < nbranch; ++i) { p.br[i].lepe_w = lepe_w[i]; p.br[i].lepe_b = lepe_b[i]; }
    p.qkv = qkv; p.y = y; p.y0 = y0; p.lse = lse;
    p.stamps = g_attn_stamps;
    hipStream_t st = (hipStream_t)stream;
    switch (nt) {
        case 1: case 2: case 3: case 4: rc = launch_fwd3<4>(p, nwg, st); break;
        case 5: case 6: rc = launch_fwd3<6>(p, nwg, st); break;
        case 7: rc = launch_fwd3<7>(p, nwg, st); break;
        case 8: case 9: rc = launch_fwd<9>(p, nwg, st); break;
        case 10: case 11: case 12: rc = launch_fwd<12>(p, nwg, st); break;
        case 13: case 14: case 15: rc = launch_fwd<15>(p, nwg, st); break;
        case 16: case 17: case 18: rc = launch_fwd<18>(p, nwg, st); break;
        default:
            cswin_set_error("attn_fwd: window of %d tokens unsupported", p.br[0].H_sp * p.br[0].W_sp);
            return CSWIN_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

size_t cswin_attn_bwd_workspace(int B, int reso, int C, int nbranch, const int* heads, const int* idx, int split) {
    AttnParams p = {};
    int nt, nwg;
    if (fill_params(p, "attn_bwd_workspace", B, reso, C, nbranch, heads, idx, split, 0.f, &nt, &nwg)) return 0;
    // LePE partial slabs (LW_SUB rows per window on the two-pass path) + delta (B, heads, L) (used by the two-pass path)
    size_t n = (size_t)nwg * 10 * HD * ((nt > 7 || force_two_pass()) ? LW_SUB : 1);
    n += (size_t)B * p.heads_total * reso * reso;
    return n * sizeof(float);
}

// dqkv (B, L, 3C) is fully overwritten; dlepe_w[i] (Cb,9) and dlepe_b[i] (Cb) are overwritten.
int cswin_attn_bwd(const float* qkv, const float* const* lepe_w, const float* const* lepe_b, const float* lse,
                   const float* y0, const float* dy, float* dqkv, float* const* dlepe_w, float* const* dlepe_b,
                   void* workspace, size_t ws_bytes, int B, int reso, int C, int nbranch, const int* heads, const int* idx,
                   int split, float scale, cswin_reduce_job* deferred, float drop_p, unsigned long long drop_seed,
                   const unsigned long long* drop_epoch, int qkv_bf16, void* stream) {
    AttnParams p = {};
    CSWIN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, CSWIN_ERR_UNSUPPORTED, "attn_bwd: dropout probability %g outside [0, 1)", (double)drop_p);
    p.drop_p = drop_p; p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_thresh = (unsigned)(drop_p * 16777216.0f); p.drop_seed = drop_seed; p.drop_epoch = drop_epoch;
    CSWIN_REQUIRE(qkv_bf16 == 0 || qkv_bf16 == 1 || qkv_bf16 == 3 || qkv_bf16 == 7, CSWIN_ERR_UNSUPPORTED,
                  "attn: mode %d (0 = fp32, 1 = qkv / dqkv stored as bf16, 3 = also y, 7 = also bf16 MFMAs)", qkv_bf16);
    p.qkv_bf16 = qkv_bf16;
    int nt, nwg;
    int rc = fill_params(p, "attn_bwd", B, reso, C, nbranch, heads, idx, split, scale, &nt, &nwg);
    if (rc) return rc;
    CSWIN_REQUIRE(qkv && lse && dy && dqkv && lepe_w && dlepe_w && dlepe_b, CSWIN_ERR_SHAPE, "attn_bwd: null pointer");
    CSWIN_REQUIRE(workspace && ws_bytes >= cswin_attn_bwd_workspace(B, reso, C, nbranch, heads, idx, split), CSWIN_ERR_WORKSPACE, "attn_bwd: workspace too small");
    CSWIN_REQUIRE(y0 && lepe_b, CSWIN_ERR_SHAPE, "attn_bwd: y0 (the forward's output without the LePE term) and lepe_b are required");
    const bool two_pass = nt > 7 || force_two_pass();
    p.slab_rows = two_pass ? LW_SUB : 1;
    p.ds_stride = ds_stride_for(p.br[0].H_sp * p.br[0].W_sp);
    for (int i = 0; i < nbranch; ++i) {
        p.br[i].lepe_w = lepe_w[i];
        p.br[i].lepe_b = lepe_b[i];
        p.br[i].dw_part = (float*)workspace + (size_t)p.br[i].wg_begin * 10 * HD * p.slab_rows;
    }
    p.stamps = g_attn_stamps;
    p.y_in = y0;
    p.delta = (float*)workspace + (size_t)nwg * 10 * HD * p.slab_rows;
    p.qkv = qkv; p.lse = const_cast<float*>(lse); p.dy = dy; p.dqkv = dqkv;
    hipStream_t st = (hipStream_t)stream;
    if (two_pass) {
        // windows of more than 112 tokens (384x384: N = 144, 288): two-pass path
        const int N = p.br[0].H_sp * p.br[0].W_sp, nblk = (N + 63) / 64;
        const long items = (long)B * p.heads_total * reso * reso;
        if (p.qkv_bf16 == 7) launch_bwd_two_pass<7>(p, items, nwg, nblk, st);
        else if (p.qkv_bf16 == 3) launch_bwd_two_pass<3>(p, items, nwg, nblk, st);
        else if (p.qkv_bf16) launch_bwd_two_pass<1>(p, items, nwg, nblk, st);
        else launch_bwd_two_pass<0>(p, items, nwg, nblk, st);
        rc = CSWIN_OK;
    } else {
        const int Ntok = p.br[0].H_sp * p.br[0].W_sp;
        switch (nt) {
            case 1: case 2: case 3: case 4: p.vs_floats = vs_floats_for(4, Ntok, p.ds_stride); rc = launch_bwd3<4>(p, nwg, st); break;
            case 5: case 6: p.vs_floats = vs_floats_for(6, Ntok, p.ds_stride); rc = launch_bwd3<6>(p, nwg, st); break;
            default: p.vs_floats = vs_floats_for(7, Ntok, p.ds_stride); rc = launch_bwd3<7>(p, nwg, st); break;
        }
    }
    if (rc) return rc;
    CSWIN_LAUNCH_CHECK();
    // the slabs of each branch are a standard reduction job: rows = B * nWin, columns = [Cb * 9 | Cb]
    ReduceJobs J = {};
    int blocks = 0;
    for (int i = 0; i < nbranch; ++i) {
        const AttnBranch& br = p.br[i];
        const long cb = (long)br.heads * p.hd;
        cswin_reduce_job job = {br.dw_part, dlepe_w[i], dlepe_b[i], cb * 9, cb * 10, cb * 10, B * br.nWin * p.slab_rows, 0};
        if (deferred) {
            deferred[i] = job;
            continue;
        }
        job.reserved = reduce_job_vec_ok(job);
        J.j[i] = job;
        J.first_block[i] = blocks;
        blocks += (int)((job.n + RS_COLS - 1) / RS_COLS);
    }
    if (!deferred) {
        J.first_block[nbranch] = blocks;
        J.njobs = nbranch;
        hipLaunchKernelGGL(rows_sum_multi_kernel, dim3(blocks), dim3(256), 0, st, J);
        CSWIN_LAUNCH_CHECK();
    }
    return CSWIN_OK;
}

int cswin_img2windows(const float* img, float* out, int B, int C, int H, int W, int H_sp, int W_sp, void* stream) {
    CSWIN_REQUIRE(img && out && B > 0 && C > 0 && H_sp > 0 && W_sp > 0 && H % H_sp == 0 && W % W_sp == 0, CSWIN_ERR_SHAPE,
                  "img2windows: shape '[%d, %d, %d, %d]' is invalid for windows %dx%d", B, C, H, W, H_sp, W_sp);
    long total = (long)B * C * H * W;
    hipLaunchKernelGGL(img2windows_kernel, dim3(min((long)cdiv(total, 256), 4096L)), dim3(256), 0, (hipStream_t)stream, img, out, B, C, H, W, H_sp, W_sp);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

int cswin_windows2img(const float* win, float* out, int B, int C, int H, int W, int H_sp, int W_sp, void* stream) {
    CSWIN_REQUIRE(win && out && B > 0 && C > 0 && H_sp > 0 && W_sp > 0 && H % H_sp == 0 && W % W_sp == 0, CSWIN_ERR_SHAPE,
                  "windows2img: bad shape");
    long total = (long)B * C * H * W;
    hipLaunchKernelGGL(windows2img_kernel, dim3(min((long)cdiv(total, 256), 4096L)), dim3(256), 0, (hipStream_t)stream, win, out, B, C, H, W, H_sp, W_sp);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

}  // extern "C"
