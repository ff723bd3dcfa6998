// Optimiser side of the step: SGD with momentum and weight decay (trainer.py:42, torch.optim.SGD
// semantics) over ONE flat parameter buffer, plus the multi-tensor gather that packs the per-tensor
// gradients into the flat gradient buffer that RCCL all-reduces.  One launch each instead of 463.
//   g' = g * grad_scale + wd * p ;  m = mu * m + g' ;  p -= lr * m      (m starts at 0, so step 1 gives m = g')
// lr is read from DEVICE memory so a captured hipGraph can be replayed under the poly schedule.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sgd_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, long n, const float* __restrict__ lr_dev,
                                                        float momentum, float wd, float grad_scale, __bf16* __restrict__ shadow) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const float lr = lr_dev[0];
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        mv = momentum * mv + (gv * grad_scale + wd * pv);
        pv -= lr * mv;
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        if (shadow) reinterpret_cast<bf16x4*>(shadow)[i] = __builtin_convertvector(pv, bf16x4);     // the GEMMs' bf16 copy of the weights
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float mv = momentum * m[i] + (g[i] * grad_scale + wd * p[i]);
        m[i] = mv;
        const float pn = p[i] - lr * mv;
        p[i] = pn;
        if (shadow) shadow[i] = (__bf16)pn;
    }
}

struct CopyChunk { const float* src; float* dst; long n; };

// table[i] = {src, dst, n}: one workgroup per chunk (chunks are <= 16384 floats)
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyChunk* __restrict__ table) {
    const CopyChunk c = table[blockIdx.x];
    const bool vec = (((uintptr_t)c.src | (uintptr_t)c.dst) & 15) == 0;
    if (vec) {
        const long n4 = c.n / 4;
        for (long i = threadIdx.x; i < n4; i += 256) reinterpret_cast<f32x4*>(c.dst)[i] = reinterpret_cast<const f32x4*>(c.src)[i];
        for (long i = n4 * 4 + threadIdx.x; i < c.n; i += 256) c.dst[i] = c.src[i];
    } else {
        for (long i = threadIdx.x; i < c.n; i += 256) c.dst[i] = c.src[i];
    }
}

// bf16 gradient wire (BASELINE configs[2]: "RCCL all-reduce bf16"): the fp32 flat gradient bucket is rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32: NaN stays NaN) into a wire buffer that the collective sums, then widened back in place of the bucket.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n, float scale) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        reinterpret_cast<bf16x4_t*>(dst)[i] = __builtin_convertvector(reinterpret_cast<const f32x4*>(src)[i] * scale, bf16x4_t);
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (__bf16)(src[i] * scale);
}

__global__ __launch_bounds__(256) void unpack_bf16_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        reinterpret_cast<f32x4*>(dst)[i] = __builtin_convertvector(reinterpret_cast<const bf16x4_t*>(src)[i], f32x4);
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (float)src[i];
}

}  // namespace

extern "C" {

// fp32 -> bf16 (round to nearest even) / bf16 -> fp32 over n elements: the gradient-bucket wire format
int cswin_pack_bf16_scaled(const float* src, void* dst, long n, float scale, void* stream) {
    CSWIN_REQUIRE(src && dst && n > 0, CSWIN_ERR_SHAPE, "pack_bf16: bad arguments");
    CSWIN_REQUIRE((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0, CSWIN_ERR_ALIGN, "pack_bf16: src 16-B / dst 8-B alignment required");
    long b = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((int)(b < 1 ? 1 : (b > 4096 ? 4096 : b))), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n, scale);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

int cswin_pack_bf16(const float* src, void* dst, long n, void* stream) { return cswin_pack_bf16_scaled(src, dst, n, 1.0f, stream); }

int cswin_unpack_bf16(const void* src, float* dst, long n, void* stream) {
    CSWIN_REQUIRE(src && dst && n > 0, CSWIN_ERR_SHAPE, "unpack_bf16: bad arguments");
    CSWIN_REQUIRE((((uintptr_t)dst) & 15) == 0 && (((uintptr_t)src) & 7) == 0, CSWIN_ERR_ALIGN, "unpack_bf16: dst 16-B / src 8-B alignment required");
    long b = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(unpack_bf16_kernel, dim3((int)(b < 1 ? 1 : (b > 4096 ? 4096 : b))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, dst, n);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

int cswin_sgd_flat(float* p, const float* g, float* m, long n, const float* lr_dev, float momentum, float weight_decay,
                   float grad_scale, void* shadow_bf16, void* stream) {
    CSWIN_REQUIRE(p && g && m && lr_dev && n > 0, CSWIN_ERR_SHAPE, "sgd_flat: bad arguments");
    CSWIN_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m)) & 15) == 0, CSWIN_ERR_ALIGN, "sgd_flat: buffers must be 16-B aligned");
    CSWIN_REQUIRE(!shadow_bf16 || (((uintptr_t)shadow_bf16) & 7) == 0, CSWIN_ERR_ALIGN, "sgd_flat: the bf16 shadow must be 8-B aligned");
    long b = (n / 4 + 255) / 256;
    int grid = (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
    hipLaunchKernelGGL(sgd_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, n, lr_dev, momentum, weight_decay, grad_scale, (__bf16*)shadow_bf16);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

// table: device array of nchunks {const float* src; float* dst; int64 n} records (24 bytes each)
int cswin_multi_copy(const void* table, int nchunks, void* stream) {
    CSWIN_REQUIRE(table && nchunks > 0, CSWIN_ERR_SHAPE, "multi_copy: bad arguments");
    hipLaunchKernelGGL(multi_copy_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const CopyChunk*)table);
    CSWIN_LAUNCH_CHECK();
    return CSWIN_OK;
}

}  // extern "C"
