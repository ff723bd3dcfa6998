// CARAFE content-aware reassembly (networks/cswin_unet.py:222-319) in its closed form:
//
//   Wt[b,hw,k,s] = softmax_k e[b,hw,k*S^2+s]
//   out[b, (hS+sy)(SW) + (wS+sx), c] = bias[c] + sum_k Wt[b,hw,k,s] * z[b, nbr_k(h,w), c]      (zero outside the map)
//
// where z = x @ W_out^T is the `out` 1x1 convolution applied BEFORE the reassembly, at LOW
// resolution: the 1x1 conv is linear and the reassembly weights sum over pixels only, so the two
// commute exactly (S^2 x fewer GEMM FLOPs, and the (B,C,SH,SW) tensor of the reference never exists).
// pixel_shuffle / unfold / pad / permute of the reference all collapse into index arithmetic here.
// Everything is on the (B, L, C) token layout.  HBM-bound: one pass over e and z (z neighbours
// come from L1/L2), one coalesced write of out.
#include "common.h"
#include <stdlib.h>

namespace {

// lanes per (pixel, sub-pixel) item: the largest power of two <= min(Cz / 4, 64); they stride over the Cz / 4 chunks
__host__ __device__ inline int carafe_lpr(int Cz) {
    int l = 1;
    while (2 * l <= Cz / 4 && 2 * l <= 64) l *= 2;
    return l;
}

// one group of LPR lanes (Cz = 4*LPR*VPL... here VPL folded into a loop) handles one (low-res pixel, sub-pixel s)
template <int S>
__global__ __launch_bounds__(256) void carafe_fwd_kernel(const float* __restrict__ e, const float* __restrict__ z,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          float* __restrict__ wt_save, int B, int H, int W, int Cz) {
    constexpr int S2 = S * S;
    const int lpr = carafe_lpr(Cz);                   // lanes per item
    const int groups = 256 / lpr;
    const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
    const long items = (long)B * H * W * S2;
    for (long it = (long)blockIdx.x * groups + grp; it < items; it += (long)gridDim.x * groups) {
        const int s = (int)(it % S2);
        const long pix = it / S2;                    // b*H*W + h*W + w
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int b = (int)(pix / ((long)W * H));
        float wt[9];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            wt[k] = e[pix * (9 * S2) + k * S2 + s];
            mx = fmaxf(mx, wt[k]);
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            wt[k] = __expf(wt[k] - mx);
            sum += wt[k];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] *= inv;
        if (wt_save && sub == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wt_save[pix * (9 * S2) + k * S2 + s] = wt[k];
        }
        const int sy = s / S, sx = s - sy * S;
        const long orow = ((long)b * H * S + h * S + sy) * (W * S) + w * S + sx;
        for (int c = 4 * sub; c < Cz; c += 4 * lpr) {
            f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int hh = h + k / 3 - 1, ww = w + k % 3 - 1;
                if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
                    acc += wt[k] * *reinterpret_cast<const f32x4*>(z + (((long)b * H + hh) * W + ww) * Cz + c);
            }
            *reinterpret_cast<f32x4*>(out + orow * Cz + c) = acc;
        }
    }
}

// de[b,hw,k*S2+s] = Wt[k] * (dWt[k] - sum_j Wt[j] dWt[j]),  dWt[k] = sum_c dout[pix(s), c] * z[nbr_k, c]
// dbias_part != NULL (Cz <= 512): the kernel streams over dout anyway, so it also leaves the per-workgroup column sums of dout
// in dbias_part[block][Cz] (dbias = their sum), which saves the separate column-sum pass over dout.
template <int S>
__global__ __launch_bounds__(256) void carafe_bwd_e_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                            const float* __restrict__ wt_save, float* __restrict__ de,
                                                            float* __restrict__ dbias_part, int B, int H, int W, int Cz) {
    __shared__ float bred[2 * 1024];                 // [chunk][group][4 * lpr]: groups * 4 * lpr = 1024 floats per chunk
    f32x4 bacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    constexpr int S2 = S * S;
    const int lpr = carafe_lpr(Cz);
    const int groups = 256 / lpr;
    const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
    const long items = (long)B * H * W * S2;
    const long items_pad = (items + groups - 1) / groups * groups;     // keep whole groups alive for the shuffles
    for (long it0 = (long)blockIdx.x * groups + grp; it0 < items_pad; it0 += (long)gridDim.x * groups) {
        const bool live = it0 < items;
        const long it = live ? it0 : items - 1;
        const int s = (int)(it % S2);
        const long pix = it / S2;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int b = (int)(pix / ((long)W * H));
        const int sy = s / S, sx = s - sy * S;
        const long orow = ((long)b * H * S + h * S + sy) * (W * S) + w * S + sx;
        float dwt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) dwt[k] = 0.f;
        int ci = 0;
        for (int c = 4 * sub; c < Cz; c += 4 * lpr, ++ci) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dout + orow * Cz + c);
            if (live && dbias_part) {
                if (ci == 0) bacc[0] += g;
                else bacc[1] += g;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int hh = h + k / 3 - 1, ww = w + k % 3 - 1;
                if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) {
                    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (((long)b * H + hh) * W + ww) * Cz + c);
                    dwt[k] += g[0] * zv[0] + g[1] * zv[1] + g[2] * zv[2] + g[3] * zv[3];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
            for (int o = lpr >> 1; o > 0; o >>= 1) dwt[k] += __shfl_xor(dwt[k], o, 64);
        if (sub == 0 && live) {
            float wt[9], dot = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                wt[k] = wt_save[pix * (9 * S2) + k * S2 + s];
                dot += wt[k] * dwt[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) de[pix * (9 * S2) + k * S2 + s] = wt[k] * (dwt[k] - dot);
        }
    }
    if (dbias_part) {                                 // column sums of this workgroup's rows of dout
        const int span = 4 * lpr;                     // channels covered by one chunk
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bred[(q * groups + grp) * span + 4 * sub + e] = bacc[q][e];
        __syncthreads();
        for (int c = threadIdx.x; c < Cz; c += 256) {
            const int q = c / span, cc = c - q * span;
            float t = 0.f;
            for (int g2 = 0; g2 < groups; ++g2) t += bred[(q * groups + g2) * span + cc];
            dbias_part[(long)blockIdx.x * Cz + c] = t;
        }
    }
}

// dz[b,n,c] = sum_k sum_s Wt[n - off_k][k][s] * dout[pix(n - off_k, s), c]
template <int S>
__global__ __launch_bounds__(256) void carafe_bwd_z_kernel(const float* __restrict__ dout,
                                                            const float* __restrict__ wt_save, float* __restrict__ dz,
                                                            int B, int H, int W, int Cz) {
    constexpr int S2 = S * S;
    const int lpr = carafe_lpr(Cz);
    const int groups = 256 / lpr;
    const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
    const long items = (long)B * H * W;
    for (long pix = (long)blockIdx.x * groups + grp; pix < items; pix += (long)gridDim.x * groups) {
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int b = (int)(pix / ((long)W * H));
        for (int c = 4 * sub; c < Cz; c += 4 * lpr) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                // source pixel (h2, w2) whose k-th neighbour is (h, w):  h2 + k/3 - 1 = h
                const int h2 = h - (k / 3 - 1), w2 = w - (k % 3 - 1);
                if ((unsigned)h2 < (unsigned)H && (unsigned)w2 < (unsigned)W) {
                    const long p2 = ((long)b * H + h2) * W + w2;
#pragma unroll
                    for (int s = 0; s < S2; ++s) {
                        const float wv = wt_save[p2 * (9 * S2) + k * S2 + s];
                        const long orow = ((long)b * H * S + h2 * S + s / S) * (W * S) + w2 * S + s % S;
                        acc += wv * *reinterpret_cast<const f32x4*>(dout + orow * Cz + c);
                    }
                }
            }
            *reinterpret_cast<f32x4*>(dz + pix * Cz + c) = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused backward for the model's last stage (CARAFE4 + fused head: S = 4, Cz = 16; 224 x 224 logits at B = 24 are 77 MB of
// dout).  The three generic kernels below read dout three times (de, dz, dbias) and gather it with 4-B / 16-B pieces; here
// one workgroup owns an 8 x 8 tile of low-resolution pixels and reads the 16 x 16 block D[p] = dout[4h..4h+3][4w..4w+3][0..15]
// of every pixel of the tile + a one-pixel halo exactly once per use, as 256-B row segments, and the per-pixel
// contractions run on the matrix pipes (v_mfma_f32_16x16x4_f32, exact fp32):
//   G[p]   (9 x 16) = Wt[p] (9 taps x 16 sub-pixels) . D[p] (16 sub-pixels x 16 channels)     -> LDS, tile + halo
//   dWt[p] (9 x 16) = Znbr[p] (9 taps x 16 channels) . D[p]^T                                   -> softmax backward -> de
//   dz[n][c] = sum_k G[n - off_k][k][c]   (9 LDS reads per output),   dbias[c] = sum_{p, s} D[p][s][c]
// ---------------------------------------------------------------------------------------------------------------------
constexpr int C4_T = 8;                       // tile edge (low-resolution pixels)
constexpr int C4_HALO = C4_T + 2;
constexpr int C4_NP = C4_HALO * C4_HALO;      // 100 pixels whose G is needed

__device__ __forceinline__ f32x4 carafe_mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int C4_WAVES = 8;                   // waves per workgroup (two workgroups of 64 KB LDS share a CU)

__global__ __launch_bounds__(64 * C4_WAVES) void carafe4_bwd_fused_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                                 const float* __restrict__ wt_save, float* __restrict__ de,
                                                                 float* __restrict__ dz, float* __restrict__ dbias_part,
                                                                 int B, int H, int W, int tiles_x, int tiles_y) {
    constexpr int S = 4, S2 = 16, Cz = 16;
    __shared__ __attribute__((aligned(16))) float Gs[C4_NP * 9 * Cz];      // [pixel][tap][channel]   57.6 KB
    __shared__ __attribute__((aligned(16))) float zt[C4_NP * Cz];          // z of the tile + halo       6.4 KB
    __shared__ float bred[C4_WAVES][Cz];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int h0 = ty * C4_T - 1, w0 = tx * C4_T - 1;           // top-left of the halo region

    for (int i = tid; i < C4_NP * (Cz / 4); i += 64 * C4_WAVES) {
        const int pl = i >> 2, c4 = i & 3;
        const int h = h0 + pl / C4_HALO, w = w0 + pl % C4_HALO;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
            v = *reinterpret_cast<const f32x4*>(z + (((long)b * H + h) * W + w) * Cz + 4 * c4);
        *reinterpret_cast<f32x4*>(&zt[pl * Cz + 4 * c4]) = v;
    }
    __syncthreads();

    float bsum = 0.f;                                            // dbias partial: lane (li = channel), its kq's share
    // Each wave walks its pixels four at a time: all global loads of the four (reassembly weights as A operand, the D block
    // in both operand layouts, the weights again in the accumulator layout) are issued before any of them is consumed, so
    // a wave has ~1 KB x 4 in flight instead of one dependent load chain per pixel.
    constexpr int PB = 4;
    for (int pl0 = wave * PB; pl0 < C4_NP; pl0 += C4_WAVES * PB) {
        bool inside[PB], interior[PB];
        long pix[PB];
        float av[PB][4], dv[PB][4], wv[PB][4];
        f32x4 db[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int pl = pl0 + u;
            const int ph = pl / C4_HALO, pw = pl - ph * C4_HALO;
            const int h = h0 + ph, w = w0 + pw;
            inside[u] = pl < C4_NP && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;        // wave-uniform
            interior[u] = inside[u] && ph >= 1 && ph <= C4_T && pw >= 1 && pw <= C4_T;
            pix[u] = ((long)b * H + h) * W + w;
            db[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) av[u][j] = dv[u][j] = wv[u][j] = 0.f;
            if (!inside[u]) continue;
            const float* wp = wt_save + pix[u] * (9 * S2);
            const float* drow = dout + (((long)b * H * S + h * S) * (W * S) + w * S) * Cz;   // hi-res pixel (4h, 4w)
            const long hstride = (long)W * S * Cz;                                           // one hi-res row down
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (li < 9) av[u][j] = wp[li * S2 + 4 * j + kq];                              // A: tap li, sub-pixel 4j + kq
                dv[u][j] = drow[j * hstride + kq * Cz + li];      // B: channel li, sub-pixel (row j, column kq): 256-B segments
            }
            if (interior[u]) {
                db[u] = *reinterpret_cast<const f32x4*>(drow + (li >> 2) * hstride + (li & 3) * Cz + 4 * kq);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * kq + r < 9) wv[u][r] = wp[(4 * kq + r) * S2 + li];
            }
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int pl = pl0 + u;
            if (pl >= C4_NP) continue;
            float* Gp = &Gs[pl * 9 * Cz];
            if (!inside[u]) {
                for (int i = lane; i < 9 * Cz; i += 64) Gp[i] = 0.f;
                continue;
            }
            // ---- G = Wt . D ----
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) g = carafe_mfma4(av[u][j], dv[u][j], g);
            // C layout: lane holds G[tap = 4 kq + r][channel = li]
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kq + r < 9) Gp[(4 * kq + r) * Cz + li] = g[r];
            if (!interior[u]) continue;
            bsum += dv[u][0] + dv[u][1] + dv[u][2] + dv[u][3];
            // ---- dWt = Znbr . D^T : A lane (tap li, channels 4 kq + j), B lane (sub-pixel li, channels 4 kq + j) ----
            const int ph = pl / C4_HALO, pw = pl - ph * C4_HALO;
            f32x4 za = {0.f, 0.f, 0.f, 0.f};
            if (li < 9) {
                const int nh = ph + li / 3 - 1, nw = pw + li % 3 - 1;                       // neighbour inside the halo region
                za = *reinterpret_cast<const f32x4*>(&zt[(nh * C4_HALO + nw) * Cz + 4 * kq]);   // zeros outside the image
            }
            f32x4 dw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) dw = carafe_mfma4(za[j], db[u][j], dw);
            // lane holds dWt[tap = 4 kq + r][sub-pixel = li]; softmax backward over the 9 taps of this sub-pixel
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot += wv[u][r] * dw[r];
            dot += __shfl_xor(dot, 16, 64);
            dot += __shfl_xor(dot, 32, 64);
            float* dep = de + pix[u] * (9 * S2);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kq + r < 9) dep[(4 * kq + r) * S2 + li] = wv[u][r] * (dw[r] - dot);
        }
    }
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < Cz) bred[wave][lane] = bsum;
    __syncthreads();
    if (tid < Cz && dbias_part) {
        float t2 = 0.f;
#pragma unroll
        for (int k = 0; k < C4_WAVES; ++k) t2 += bred[k][tid];
        dbias_part[(long)blockIdx.x * Cz + tid] = t2;
    }

    // ---- dz of the 64 interior pixels: 4 lanes x 16 B per pixel ----
    if (tid < 256) {
        const int n = tid >> 2, c4 = tid & 3;
        const int ph = 1 + n / C4_T, pw = 1 + n % C4_T;
        const int h = h0 + ph, w = w0 + pw;
        if (h < H && w < W) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                // source pixel whose k-th neighbour is (h, w): (h - (k/3 - 1), w - (k%3 - 1)); outside the image its G is 0
                const int sh = ph - (k / 3 - 1), sw = pw - (k % 3 - 1);
                acc += *reinterpret_cast<const f32x4*>(&Gs[((sh * C4_HALO + sw) * 9 + k) * Cz + 4 * c4]);
            }
            *reinterpret_cast<f32x4*>(dz + (((long)b * H + h) * W + w) * Cz + 4 * c4) = acc;
        }
    }
}

// column sums of a (rows, C) matrix: partial[blk][C] then a second pass
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                              long rows, int C) {
    __shared__ float red[256];
    const int lanes = min(C, 256);
    const int rgroups = 256 / lanes;
    const int c0 = threadIdx.x % lanes, rg = threadIdx.x / lanes;
    for (int c = c0; c < C; c += lanes) {
        float s = 0.f;
        if (rg < rgroups)
            for (long r = (long)blockIdx.x * rgroups + rg; r < rows; r += (long)gridDim.x * rgroups) s += x[r * C + c];
        red[threadIdx.x] = s;
        __syncthreads();
        if (rg == 0) {
            float t = 0.f;
            for (int k = 0; k < rgroups; ++k) t += red[k * lanes + c0];
            partial[(long)blockIdx.x * C + c] = t;
        }
        __syncthreads();
    }
}

int colsum_blocks(long rows, int C) {
    int lanes = C < 256 ? C : 256;
    int rg = 256 / lanes;
    long b = (rows + rg * 64 - 1) / (rg * 64);
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

int grid_for(long items, int groups) {
    long b = (items + groups - 1) / groups;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

bool carafe_args_ok(int B, int H, int W, int Cz, int S) {
    if (B <= 0 || H <= 0 || W <= 0 || Cz <= 0 || (S != 2 && S != 4)) return false;
    return Cz % 4 == 0 && Cz >= 4;
}

}  // namespace

extern "C" {

// e (B, H*W, 9*S*S), z (B, H*W, Cz), bias (Cz) or NULL -> out (B, (S*H)*(S*W), Cz); wt_save (B, H*W, 9*S*S) or NULL
int cswin_carafe_fwd(const float* e, const float* z, const float* bias, float* out, float* wt_save, int B, int H, int W,
                     int Cz, int S, void* stream) {
    CSWIN_REQUIRE(e && z && out, CSWIN_ERR_SHAPE, "carafe_fwd: null pointer");
    CSWIN_REQUIRE(carafe_args_ok(B, H, W, Cz, S), CSWIN_ERR_UNSUPPORTED, "carafe_fwd: unsupported shape B=%d H=%d W=%d Cz=%d S=%d (Cz %% 4 == 0, S in {2,4})", B, H, W, Cz, S);
    const int groups = 256 / carafe_lpr(Cz);
    hipStream_t st = (hipStream_t)stream;
    const long items = (long)B * H * W * S * S;
    if (S == 2) hipLaunchKernelGGL(carafe_fwd_kernel<2>, dim3(grid_for(items, groups)), dim3(256), 0, st, e, z, bias, out, wt_save, B, H, W, Cz);
    else hipLaunchKernelGGL(carafe_fwd_kernel<4>, dim3(grid_for(items, groups)), dim3(256), 0, st, e, z, bias, out, wt_save, B, H, W, Cz);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

static bool carafe4_fused_ok(int H, int W, int Cz, int S) { return S == 4 && Cz == 16 && H % C4_T == 0 && W % C4_T == 0; }

size_t cswin_carafe_bwd_workspace(int B, int H, int W, int Cz, int S) {
    size_t generic = (size_t)colsum_blocks((long)B * H * W * S * S, Cz) * Cz * sizeof(float);
    size_t in_e = Cz <= 512 ? (size_t)grid_for((long)B * H * W * S * S, 256 / carafe_lpr(Cz)) * Cz * sizeof(float) : 0;
    if (in_e > generic) generic = in_e;
    size_t fused = carafe4_fused_ok(H, W, Cz, S) ? (size_t)B * (H / C4_T) * (W / C4_T) * Cz * sizeof(float) : 0;
    return generic > fused ? generic : fused;
}

// dout (B, (S*H)*(S*W), Cz) -> de (B, H*W, 9*S*S), dz (B, H*W, Cz), dbias (Cz) (may be NULL)
int cswin_carafe_bwd(const float* dout, const float* z, const float* wt_save, float* de, float* dz, float* dbias,
                     void* workspace, size_t ws_bytes, int B, int H, int W, int Cz, int S, cswin_reduce_job* deferred, void* stream) {
    if (deferred) *deferred = cswin_reduce_job{};          // part == NULL: nothing pending (no bias)
    CSWIN_REQUIRE(dout && z && wt_save && de && dz, CSWIN_ERR_SHAPE, "carafe_bwd: null pointer");
    CSWIN_REQUIRE(carafe_args_ok(B, H, W, Cz, S), CSWIN_ERR_UNSUPPORTED, "carafe_bwd: unsupported shape");
    CSWIN_REQUIRE(!dbias || (workspace && ws_bytes >= cswin_carafe_bwd_workspace(B, H, W, Cz, S)), CSWIN_ERR_WORKSPACE, "carafe_bwd: workspace too small");
    const int groups = 256 / carafe_lpr(Cz);
    hipStream_t st = (hipStream_t)stream;
    const long items = (long)B * H * W * S * S, pixels = (long)B * H * W;
    const bool no_fused = cswin_tuning().carafe_generic != 0;                                // tuning aid
    if (carafe4_fused_ok(H, W, Cz, S) && !no_fused) {
        const int tx = W / C4_T, ty = H / C4_T, nblk = B * tx * ty;
        hipLaunchKernelGGL(carafe4_bwd_fused_kernel, dim3(nblk), dim3(64 * C4_WAVES), 0, st, dout, z, wt_save, de, dz,
                           dbias ? (float*)workspace : nullptr, B, H, W, tx, ty);
        CSWIN_LAUNCH_CHECK();
        if (dbias) {
            reduce_now_or_defer(cswin_reduce_job{(const float*)workspace, dbias, nullptr, 0, Cz, Cz, nblk, 0, 0, 0}, deferred, st);
            CSWIN_LAUNCH_CHECK();
        }
        return CSWIN_OK;
    }
    const bool bias_in_e = dbias && Cz <= 512;        // at most two 16-B chunks per lane: the column sums ride along in bwd_e
    float* bpart = bias_in_e ? (float*)workspace : nullptr;
    const int eblk = grid_for(items, groups);
    if (S == 2) {
        hipLaunchKernelGGL(carafe_bwd_e_kernel<2>, dim3(eblk), dim3(256), 0, st, dout, z, wt_save, de, bpart, B, H, W, Cz);
        hipLaunchKernelGGL(carafe_bwd_z_kernel<2>, dim3(grid_for(pixels, groups)), dim3(256), 0, st, dout, wt_save, dz, B, H, W, Cz);
    } else {
        hipLaunchKernelGGL(carafe_bwd_e_kernel<4>, dim3(eblk), dim3(256), 0, st, dout, z, wt_save, de, bpart, B, H, W, Cz);
        hipLaunchKernelGGL(carafe_bwd_z_kernel<4>, dim3(grid_for(pixels, groups)), dim3(256), 0, st, dout, wt_save, dz, B, H, W, Cz);
    }
    CSWIN_LAUNCH_CHECK();
    if (bias_in_e) {
        reduce_now_or_defer(cswin_reduce_job{(const float*)workspace, dbias, nullptr, 0, Cz, Cz, eblk, 0, 0, 0}, deferred, st);
        CSWIN_LAUNCH_CHECK();
    } else if (dbias) {
        int nblk = colsum_blocks(items, Cz);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, dout, (float*)workspace, items, Cz);
        reduce_now_or_defer(cswin_reduce_job{(const float*)workspace, dbias, nullptr, 0, Cz, Cz, nblk, 0, 0, 0}, deferred, st);
        CSWIN_LAUNCH_CHECK();
    }
    return CSWIN_OK;
}

}  // extern "C"
