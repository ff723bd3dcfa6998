// Layout adapters at the two ends of the token pipeline and for nn.Conv2d-shaped parameters.
//   * image (B, C, H, W)  ->  NHWC tokens (B, H*W, Cpad) with zero padded channels (patch-embed input)
//   * tokens (B, H*W, Cpad) -> (B, C, H, W) taking the first C channels (segmentation logits)
//   * Conv2d weight [Cout][Cin][ks][ks] <-> implicit-GEMM images [Cout][ks*ks][Cpad] and [ks*ks][Cout][Cpad]
// All index-only (bit-exact).
#include "common.h"

namespace {

__global__ void nchw_to_tok_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, long HW, int Cpad) {
    const long total = (long)B * HW * Cpad;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int c = (int)(o % Cpad);
        const long bp = o / Cpad;
        const long p = bp % HW, b = bp / HW;
        y[o] = c < C ? x[(b * C + c) * HW + p] : 0.f;
    }
}

// y (B, C, HW) <- x (B, HW, Cpad)[..., :C]; tiled through LDS so both sides are coalesced
__global__ __launch_bounds__(256) void tok_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                                           long HW, int Cpad) {
    __shared__ float tile[64][17];
    const long ptiles = (HW + 63) / 64;
    for (long t = blockIdx.x; t < (long)B * ptiles; t += gridDim.x) {
        const long b = t / ptiles, p0 = (t % ptiles) * 64;
        for (int c0 = 0; c0 < C; c0 += 16) {
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                const int pp = i / 16, cc = i % 16;
                tile[pp][cc] = (p0 + pp < HW && c0 + cc < Cpad) ? x[(b * HW + p0 + pp) * Cpad + c0 + cc] : 0.f;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                const int cc = i / 64, pp = i % 64;
                if (p0 + pp < HW && c0 + cc < C) y[(b * C + c0 + cc) * HW + p0 + pp] = tile[pp][cc];
            }
            __syncthreads();
        }
    }
}

__global__ void conv_w_permute_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wpt,
                                      int Cout, int Cin, int kk, int Cpad) {
    const long total = (long)Cout * kk * Cpad;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(o % Cpad);
        const int tap = (int)((o / Cpad) % kk);
        const int co = (int)(o / ((long)Cpad * kk));
        const float v = ci < Cin ? w[((long)co * Cin + ci) * kk + tap] : 0.f;
        if (wp) wp[o] = v;
        if (wpt) wpt[((long)tap * Cout + co) * Cpad + ci] = v;
    }
}

__global__ void conv_w_unpermute_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin, int kk,
                                        int Cpad) {
    const long total = (long)Cout * Cin * kk;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(o % kk);
        const int ci = (int)((o / kk) % Cin);
        const int co = (int)(o / ((long)kk * Cin));
        dw[o] = dwp[((long)co * kk + tap) * Cpad + ci];
    }
}

// wf[ci][tap'][co] = w[co][ci][kk - 1 - tap']: the weight image with which the DATA gradient of a stride-1 "same" convolution
// is itself a forward convolution of dy (Cout -> Cin channels, taps mirrored), i.e. runs on the forward implicit-GEMM kernel
__global__ void conv_w_flipT_kernel(const float* __restrict__ w, float* __restrict__ wf, int Cout, int Cin, int kk) {
    const long total = (long)Cin * kk * Cout;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int co = (int)(o % Cout);
        const int tap = (int)((o / Cout) % kk);
        const int ci = (int)(o / ((long)Cout * kk));
        wf[o] = w[((long)co * Cin + ci) * kk + (kk - 1 - tap)];
    }
}

int grid1d(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---- dropout -------------------------------------------------------------------------------------------------------------
// counter-based generator: two rounds of a 64-bit mix (splitmix64 finaliser) of (seed, element index / 4); each 64-bit result
// gives the four 16-bit uniforms of a 16-B chunk.  Stateless, so backward regenerates the forward mask exactly.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                       const float* __restrict__ row_scale, float* __restrict__ y, long n,
                                                       long elems_per_sample, float p, unsigned long long seed,
                                                       const unsigned long long* __restrict__ epoch) {
    if (epoch) seed += *epoch;                                 // device-resident step counter: a replayed hipGraph draws a new mask
    const unsigned thr = (unsigned)(p * 65536.0f);             // keep iff u16 >= thr  (P(drop) = thr / 65536)
    const float inv_keep = 1.0f / (1.0f - p);
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const unsigned long long r = mix64(mix64(seed) ^ (unsigned long long)i);
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o = residual ? reinterpret_cast<const f32x4*>(residual)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        const float rs = (row_scale ? row_scale[(4 * i) / elems_per_sample] : 1.0f) * inv_keep;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += (((unsigned)(r >> (16 * e)) & 0xFFFFu) >= thr) ? rs * xv[e] : 0.f;
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

}  // namespace

extern "C" {

int cswin_dropout(const float* x, const float* residual, const float* row_scale, float* y, long n, long elems_per_sample,
                  float p, unsigned long long seed, const unsigned long long* seed_epoch, void* stream) {
    CSWIN_REQUIRE(x && y && n > 0 && n % 4 == 0 && elems_per_sample > 0 && elems_per_sample % 4 == 0, CSWIN_ERR_SHAPE,
                  "dropout: n and elems_per_sample must be positive multiples of 4");
    CSWIN_REQUIRE(p >= 0.f && p < 1.f, CSWIN_ERR_SHAPE, "dropout: p = %f outside [0, 1)", p);
    CSWIN_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)residual)) & 15) == 0, CSWIN_ERR_ALIGN, "dropout: 16-B alignment required");
    long b = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((int)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream, x, residual, row_scale, y, n,
                       elems_per_sample, p, seed, seed_epoch);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}


int cswin_nchw_to_tokens(const float* x, float* y, int B, int C, int H, int W, int Cpad, void* stream) {
    CSWIN_REQUIRE(x && y && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, CSWIN_ERR_SHAPE, "nchw_to_tokens: bad arguments");
    hipLaunchKernelGGL(nchw_to_tok_kernel, dim3(grid1d((long)B * H * W * Cpad)), dim3(256), 0, (hipStream_t)stream, x, y, B, C, (long)H * W, Cpad);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

int cswin_tokens_to_nchw(const float* x, float* y, int B, int C, int H, int W, int Cpad, void* stream) {
    CSWIN_REQUIRE(x && y && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, CSWIN_ERR_SHAPE, "tokens_to_nchw: bad arguments");
    long tiles = (long)B * (((long)H * W + 63) / 64);
    hipLaunchKernelGGL(tok_to_nchw_kernel, dim3((int)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, (hipStream_t)stream, x, y, B, C, (long)H * W, Cpad);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// w [Cout][Cin][ks][ks] -> w_perm [Cout][ks*ks][Cpad] (may be NULL) and w_permT [ks*ks][Cout][Cpad] (may be NULL)
int cswin_conv_weight_permute(const float* w, float* w_perm, float* w_permT, int Cout, int Cin, int ks, int Cpad, void* stream) {
    CSWIN_REQUIRE(w && (w_perm || w_permT) && Cout > 0 && Cin > 0 && ks > 0 && Cpad >= Cin, CSWIN_ERR_SHAPE, "conv_weight_permute: bad arguments");
    hipLaunchKernelGGL(conv_w_permute_kernel, dim3(grid1d((long)Cout * ks * ks * Cpad)), dim3(256), 0, (hipStream_t)stream, w, w_perm, w_permT, Cout, Cin, ks * ks, Cpad);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

int cswin_conv_weight_flipT(const float* w, float* wf, int Cout, int Cin, int ks, void* stream) {
    CSWIN_REQUIRE(w && wf && Cout > 0 && Cin > 0 && ks > 0, CSWIN_ERR_SHAPE, "conv_weight_flipT: bad arguments");
    hipLaunchKernelGGL(conv_w_flipT_kernel, dim3(grid1d((long)Cout * Cin * ks * ks)), dim3(256), 0, (hipStream_t)stream, w, wf, Cout, Cin, ks * ks);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

int cswin_conv_weight_unpermute(const float* dw_perm, float* dw, int Cout, int Cin, int ks, int Cpad, void* stream) {
    CSWIN_REQUIRE(dw_perm && dw && Cout > 0 && Cin > 0 && ks > 0 && Cpad >= Cin, CSWIN_ERR_SHAPE, "conv_weight_unpermute: bad arguments");
    hipLaunchKernelGGL(conv_w_unpermute_kernel, dim3(grid1d((long)Cout * Cin * ks * ks)), dim3(256), 0, (hipStream_t)stream, dw_perm, dw, Cout, Cin, ks * ks, Cpad);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

}  // extern "C"
