// LayerNorm over the channel axis of (M, C) token matrices (cswin_unet.py:168,179,218,341,497,533).
// HBM-bound: one pass over x forward (16-B loads, a row lives in registers), one pass over (dy, x)
// backward with the residual-path gradient add fused in and dgamma/dbeta reduced through
// per-workgroup partial slabs (deterministic, no atomics).
#include "common.h"

namespace {

// y may be STORED as bf16 (bf16 activation storage: the output feeds only GEMMs, which round it to bf16 anyway)
typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ln_store4(float* y, long idx, f32x4 v, int y_bf16) {
    if (y_bf16) *reinterpret_cast<ln_bf16x4*>(reinterpret_cast<__bf16*>(y) + idx) = __builtin_convertvector(v, ln_bf16x4);
    else *reinterpret_cast<f32x4*>(y + idx) = v;
}

// LPR lanes cooperate on one row, each holding VPL float4 (C = 4 * LPR * VPL)
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                      float eps, int y_bf16) {
    constexpr int C = 4 * LPR * VPL;
    constexpr int RPB = 256 / LPR;                    // rows per block pass
    const int sub = threadIdx.x % LPR, rib = threadIdx.x / LPR;
    f32x4 g[VPL], b[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        g[v] = *reinterpret_cast<const f32x4*>(gamma + 4 * (sub + v * LPR));
        b[v] = *reinterpret_cast<const f32x4*>(beta + 4 * (sub + v * LPR));
    }
    for (long row = (long)blockIdx.x * RPB + rib; row < M; row += (long)gridDim.x * RPB) {
        f32x4 xv[VPL];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            xv[v] = *reinterpret_cast<const f32x4*>(x + row * C + 4 * (sub + v * LPR));
            s += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mu = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = xv[v][e] - mu;
                q += d * d;
            }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rs = rsqrtf(q * (1.0f / C) + eps);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            f32x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (xv[v][e] - mu) * rs * g[v][e] + b[v][e];
            ln_store4(y, row * C + 4 * (sub + v * LPR), o4, y_bf16);
        }
        if (sub == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// dx = dres + rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat));  partial[blk] = {sum dy*xhat, sum dy}
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ dres,
                                                      float* __restrict__ dx, float* __restrict__ partial, int M,
                                                      __bf16* __restrict__ dx16) {
    constexpr int C = 4 * LPR * VPL;
    constexpr int RPB = 256 / LPR;
    __shared__ float red[2 * RPB * C];
    const int sub = threadIdx.x % LPR, rib = threadIdx.x / LPR;
    f32x4 g[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        g[v] = *reinterpret_cast<const f32x4*>(gamma + 4 * (sub + v * LPR));
        dg[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long row = (long)blockIdx.x * RPB + rib; row < M; row += (long)gridDim.x * RPB) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[VPL], gy[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const long off = row * C + 4 * (sub + v * LPR);
            f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
            f32x4 dv = *reinterpret_cast<const f32x4*>(dy + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[v][e] = (xv[e] - mu) * rs;
                gy[v][e] = dv[e] * g[v][e];
                s1 += gy[v][e];
                s2 += gy[v][e] * xh[v][e];
                dg[v][e] += dv[e] * xh[v][e];
                db[v][e] += dv[e];
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        s1 *= (1.0f / C);
        s2 *= (1.0f / C);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const long off = row * C + 4 * (sub + v * LPR);
            f32x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = rs * (gy[v][e] - s1 - xh[v][e] * s2);
            if (dres) o4 += *reinterpret_cast<const f32x4*>(dres + off);
            *reinterpret_cast<f32x4*>(dx + off) = o4;
            if (dx16) *reinterpret_cast<ln_bf16x4*>(dx16 + off) = __builtin_convertvector(o4, ln_bf16x4);      // bf16 twin for the GEMMs
        }
    }
    // reduce the RPB row-groups of this block, then write one partial slab
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (sub + v * LPR) + e;
            red[rib * C + c] = dg[v][e];
            red[RPB * C + rib * C + c] = db[v][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int which = c / C, cc = c - which * C;
        float s = 0.f;
        for (int r = 0; r < RPB; ++r) s += red[which * RPB * C + r * C + cc];
        partial[(long)blockIdx.x * 2 * C + c] = s;
    }
}

// ---- any C with C % 4 == 0 (cswin_base: 96, 192, 384, 768): one wave per row, lanes stride over 16-B chunks ----
constexpr int GEN_MAXV = 4;     // chunks per lane kept in registers: C <= 4 * 64 * GEN_MAXV = 1024

__global__ __launch_bounds__(256) void ln_fwd_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              float* __restrict__ mean, float* __restrict__ rstd, int M, int C,
                                                              float eps, int y_bf16) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, nchunk = C >> 2;
    for (long row = (long)blockIdx.x * 4 + wib; row < M; row += (long)gridDim.x * 4) {
        f32x4 xv[GEN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < GEN_MAXV; ++v) {
            const int ch = lane + 64 * v;
            xv[v] = ch < nchunk ? *reinterpret_cast<const f32x4*>(x + row * C + 4 * ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < GEN_MAXV; ++v)
            if (lane + 64 * v < nchunk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) q += (xv[v][e] - mu) * (xv[v][e] - mu);
            }
        const float rs = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
        for (int v = 0; v < GEN_MAXV; ++v) {
            const int ch = lane + 64 * v;
            if (ch < nchunk) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * ch), b = *reinterpret_cast<const f32x4*>(beta + 4 * ch);
                f32x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (xv[v][e] - mu) * rs * g[e] + b[e];
                ln_store4(y, row * C + 4 * ch, o4, y_bf16);
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// partial[blk][2C]: per-workgroup {sum dy*xhat, sum dy}; LDS accumulators per wave, combined at the end
__global__ __launch_bounds__(256) void ln_bwd_generic_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ dres,
                                                              float* __restrict__ dx, float* __restrict__ partial, int M, int C,
                                                              __bf16* __restrict__ dx16) {
    extern __shared__ float red[];        // [4 waves][2C]
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, nchunk = C >> 2;
    f32x4 dg[GEN_MAXV], db[GEN_MAXV];
#pragma unroll
    for (int v = 0; v < GEN_MAXV; ++v) dg[v] = db[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * 4 + wib; row < M; row += (long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[GEN_MAXV], gy[GEN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < GEN_MAXV; ++v) {
            const int ch = lane + 64 * v;
            xh[v] = gy[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ch < nchunk) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * C + 4 * ch);
                const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + row * C + 4 * ch);
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[v][e] = (xv[e] - mu) * rs;
                    gy[v][e] = dv[e] * g[e];
                    s1 += gy[v][e];
                    s2 += gy[v][e] * xh[v][e];
                    dg[v][e] += dv[e] * xh[v][e];
                    db[v][e] += dv[e];
                }
            }
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
#pragma unroll
        for (int v = 0; v < GEN_MAXV; ++v) {
            const int ch = lane + 64 * v;
            if (ch < nchunk) {
                f32x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = rs * (gy[v][e] - s1 - xh[v][e] * s2);
                if (dres) o4 += *reinterpret_cast<const f32x4*>(dres + row * C + 4 * ch);
                *reinterpret_cast<f32x4*>(dx + row * C + 4 * ch) = o4;
                if (dx16) *reinterpret_cast<ln_bf16x4*>(dx16 + row * C + 4 * ch) = __builtin_convertvector(o4, ln_bf16x4);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < GEN_MAXV; ++v) {
        const int ch = lane + 64 * v;
        if (ch < nchunk) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[wib * 2 * C + 4 * ch + e] = dg[v][e];
                red[wib * 2 * C + C + 4 * ch + e] = db[v][e];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256)
        partial[(long)blockIdx.x * 2 * C + c] = red[c] + red[2 * C + c] + red[4 * C + c] + red[6 * C + c];
}

int ln_grid(int M, int rpb) { return min(cdiv(M, rpb), 512); }

template <int LPR, int VPL>
void launch_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int M, float eps,
                int y_bf16, hipStream_t st) {
    hipLaunchKernelGGL((ln_fwd_kernel<LPR, VPL>), dim3(ln_grid(M, 256 / LPR)), dim3(256), 0, st, x, g, b, y, mean, rstd, M, eps, y_bf16);
}
template <int LPR, int VPL>
void launch_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* g,
                const float* dres, float* dx, float* partial, int M, __bf16* dx16, hipStream_t st) {
    hipLaunchKernelGGL((ln_bwd_kernel<LPR, VPL>), dim3(ln_grid(M, 256 / LPR)), dim3(256), 0, st, dy, x, mean, rstd, g, dres, dx, partial, M, dx16);
}

bool ln_fast(int C) { return C == 64 || C == 128 || C == 256 || C == 512 || C == 32 || C == 1024; }
bool ln_supported(int C) { return C > 0 && C % 4 == 0 && C <= 256 * GEN_MAXV; }

}  // namespace

extern "C" {

int cswin_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int M, int C, float eps, int y_bf16, void* stream) {
    CSWIN_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0, CSWIN_ERR_SHAPE, "layernorm_fwd: bad arguments");
    y_bf16 = y_bf16 != 0;
    CSWIN_REQUIRE(ln_supported(C), CSWIN_ERR_UNSUPPORTED, "layernorm: C=%d must be a multiple of 4, at most 1024", C);
    hipStream_t st = (hipStream_t)stream;
    if (!ln_fast(C)) {
        hipLaunchKernelGGL(ln_fwd_generic_kernel, dim3(ln_grid(M, 4)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, C, eps, y_bf16);
        CSWIN_LAUNCH_CHECK();
        return CSWIN_OK;
    }
    switch (C) {
        case 32: launch_fwd<8, 1>(x, gamma, beta, y, mean, rstd, M, eps, y_bf16, st); break;
        case 64: launch_fwd<16, 1>(x, gamma, beta, y, mean, rstd, M, eps, y_bf16, st); break;
        case 128: launch_fwd<32, 1>(x, gamma, beta, y, mean, rstd, M, eps, y_bf16, st); break;
        case 256: launch_fwd<64, 1>(x, gamma, beta, y, mean, rstd, M, eps, y_bf16, st); break;
        case 512: launch_fwd<64, 2>(x, gamma, beta, y, mean, rstd, M, eps, y_bf16, st); break;
        case 1024: launch_fwd<64, 4>(x, gamma, beta, y, mean, rstd, M, eps, y_bf16, st); break;
    }
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

size_t cswin_layernorm_bwd_workspace(int M, int C) {
    if (!ln_fast(C)) return (size_t)ln_grid(M, 4) * 2 * C * sizeof(float);
    int lpr = C / 4 > 64 ? 64 : C / 4;
    return (size_t)ln_grid(M, 256 / lpr) * 2 * C * sizeof(float);
}

// dres may be NULL; dx may alias dres.  dgamma/dbeta are overwritten.
int cswin_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes,
                        int M, int C, cswin_reduce_job* deferred, void* dx_bf16, void* stream) {
    __bf16* dx16 = (__bf16*)dx_bf16;
    CSWIN_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && M > 0, CSWIN_ERR_SHAPE, "layernorm_bwd: bad arguments");
    CSWIN_REQUIRE(ln_supported(C), CSWIN_ERR_UNSUPPORTED, "layernorm: C=%d must be a multiple of 4, at most 1024", C);
    CSWIN_REQUIRE(workspace && ws_bytes >= cswin_layernorm_bwd_workspace(M, C), CSWIN_ERR_WORKSPACE, "layernorm_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    int lpr = C / 4 > 64 ? 64 : C / 4;
    int nblk = ln_fast(C) ? ln_grid(M, 256 / lpr) : ln_grid(M, 4);
    if (!ln_fast(C))
        hipLaunchKernelGGL(ln_bwd_generic_kernel, dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), st, dy, x, mean, rstd, gamma, dres, dx, partial, M, C, dx16);
    else switch (C) {
        case 32: launch_bwd<8, 1>(dy, x, mean, rstd, gamma, dres, dx, partial, M, dx16, st); break;
        case 64: launch_bwd<16, 1>(dy, x, mean, rstd, gamma, dres, dx, partial, M, dx16, st); break;
        case 128: launch_bwd<32, 1>(dy, x, mean, rstd, gamma, dres, dx, partial, M, dx16, st); break;
        case 256: launch_bwd<64, 1>(dy, x, mean, rstd, gamma, dres, dx, partial, M, dx16, st); break;
        case 512: launch_bwd<64, 2>(dy, x, mean, rstd, gamma, dres, dx, partial, M, dx16, st); break;
        case 1024: launch_bwd<64, 4>(dy, x, mean, rstd, gamma, dres, dx, partial, M, dx16, st); break;
    }
    CSWIN_LAUNCH_CHECK();
    cswin_reduce_job job = {partial, dgamma, dbeta, C, 2LL * C, 2LL * C, nblk, 0};
    reduce_now_or_defer(job, deferred, st);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

}  // extern "C"
