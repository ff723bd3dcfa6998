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
template <int S>
__global__ __launch_bounds__(256) void carafe_bwd_e_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                            const float* __restrict__ wt_save, float* __restrict__ de,
                                                            int B, int H, int W, int Cz) {
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
        for (int c = 4 * sub; c < Cz; c += 4 * lpr) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dout + orow * Cz + c);
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

size_t cswin_carafe_bwd_workspace(int B, int H, int W, int Cz, int S) {
    return (size_t)colsum_blocks((long)B * H * W * S * S, Cz) * Cz * sizeof(float);
}

// dout (B, (S*H)*(S*W), Cz) -> de (B, H*W, 9*S*S), dz (B, H*W, Cz), dbias (Cz) (may be NULL)
int cswin_carafe_bwd(const float* dout, const float* z, const float* wt_save, float* de, float* dz, float* dbias,
                     void* workspace, size_t ws_bytes, int B, int H, int W, int Cz, int S, void* stream) {
    CSWIN_REQUIRE(dout && z && wt_save && de && dz, CSWIN_ERR_SHAPE, "carafe_bwd: null pointer");
    CSWIN_REQUIRE(carafe_args_ok(B, H, W, Cz, S), CSWIN_ERR_UNSUPPORTED, "carafe_bwd: unsupported shape");
    CSWIN_REQUIRE(!dbias || (workspace && ws_bytes >= cswin_carafe_bwd_workspace(B, H, W, Cz, S)), CSWIN_ERR_WORKSPACE, "carafe_bwd: workspace too small");
    const int groups = 256 / carafe_lpr(Cz);
    hipStream_t st = (hipStream_t)stream;
    const long items = (long)B * H * W * S * S, pixels = (long)B * H * W;
    if (S == 2) {
        hipLaunchKernelGGL(carafe_bwd_e_kernel<2>, dim3(grid_for(items, groups)), dim3(256), 0, st, dout, z, wt_save, de, B, H, W, Cz);
        hipLaunchKernelGGL(carafe_bwd_z_kernel<2>, dim3(grid_for(pixels, groups)), dim3(256), 0, st, dout, wt_save, dz, B, H, W, Cz);
    } else {
        hipLaunchKernelGGL(carafe_bwd_e_kernel<4>, dim3(grid_for(items, groups)), dim3(256), 0, st, dout, z, wt_save, de, B, H, W, Cz);
        hipLaunchKernelGGL(carafe_bwd_z_kernel<4>, dim3(grid_for(pixels, groups)), dim3(256), 0, st, dout, wt_save, dz, B, H, W, Cz);
    }
    CSWIN_LAUNCH_CHECK();
    if (dbias) {
        int nblk = colsum_blocks(items, Cz);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, dout, (float*)workspace, items, Cz);
        launch_rows_sum((const float*)workspace, dbias, nullptr, 0, Cz, nblk, Cz, st);
        CSWIN_LAUNCH_CHECK();
    }
    return CSWIN_OK;
}

}  // extern "C"
