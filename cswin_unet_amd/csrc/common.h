// Shared device/host helpers for libcswin_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CSWIN_OK 0
#define CSWIN_ERR_SHAPE (-1)
#define CSWIN_ERR_ALIGN (-2)
#define CSWIN_ERR_WORKSPACE (-3)
#define CSWIN_ERR_HIP (-4)
#define CSWIN_ERR_UNSUPPORTED (-5)

// thread-local message buffer behind cswin_last_error()
void cswin_set_error(const char* fmt, ...);

#define CSWIN_REQUIRE(cond, code, ...)          \
    do {                                        \
        if (!(cond)) {                          \
            cswin_set_error(__VA_ARGS__);       \
            return (code);                      \
        }                                       \
    } while (0)

#define CSWIN_LAUNCH_CHECK()                                                   \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            cswin_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,  \
                            hipGetErrorString(e__));                           \
            return CSWIN_ERR_HIP;                                              \
        }                                                                      \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU, matching torch.nn.GELU() default, and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// out[i] = sum_r part[r * stride + i], i < n  (deterministic order).  32 columns x G row-groups per workgroup so that
// partial-slab reductions (split-K weight gradients, LayerNorm dgamma/dbeta, loss sums, bias sums) are parallel over
// the slabs instead of one serial chain per column (G = 8 for few slabs, 32 for many; two loads in flight per thread).
// Columns >= n_first go to out2[i - n_first] when out2 != NULL.
template <int G>
static __global__ __launch_bounds__(32 * G) void rows_sum_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                  float* __restrict__ out2, long n_first, long n, int rows,
                                                                  long stride) {
    __shared__ float red[G][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int r = g;
        for (; r + G < rows; r += 2 * G) {
            s0 += part[(long)r * stride + i];
            s1 += part[(long)(r + G) * stride + i];
        }
        if (r < rows) s0 += part[(long)r * stride + i];
    }
    red[g][c] = s0 + s1;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) t += red[k][c];
        if (out2 && i >= n_first) out2[i - n_first] = t;
        else out[i] = t;
    }
}

static inline void launch_rows_sum(const float* part, float* out, float* out2, long n_first, long n, int rows, long stride,
                                   hipStream_t st) {
    const unsigned blocks = (unsigned)((n + 31) / 32);
    if (rows > 48)
        hipLaunchKernelGGL(rows_sum_kernel<32>, dim3(blocks), dim3(1024), 0, st, part, out, out2, n_first, n, rows, stride);
    else
        hipLaunchKernelGGL(rows_sum_kernel<8>, dim3(blocks), dim3(256), 0, st, part, out, out2, n_first, n, rows, stride);
}

// ---- deferred slab reductions: several producers' partial slabs reduced by ONE launch ----------------------------
// (a CSWinBlock backward has six: four split-K weight gradients and two LayerNorm dgamma/dbeta; as separate launches
// each costs ~5 us of pure launch latency)
extern "C" {
typedef struct cswin_reduce_job {
    const float* part;       // [rows][stride] partial slabs
    float* out;              // columns [0, n_first)
    float* out2;             // columns [n_first, n) (may be NULL: then everything goes to out)
    long long n_first, n, stride;
    int rows, reserved;
} cswin_reduce_job;
}

constexpr int CSWIN_MAX_REDUCE_JOBS = 8;
struct ReduceJobs {
    cswin_reduce_job j[CSWIN_MAX_REDUCE_JOBS];
    int first_block[CSWIN_MAX_REDUCE_JOBS + 1];
    int njobs;
};

static __global__ __launch_bounds__(512) void rows_sum_multi_kernel(ReduceJobs J) {
    constexpr int G = 16;
    __shared__ float red[G][33];
    int k = 0;
    while (k + 1 < J.njobs && (int)blockIdx.x >= J.first_block[k + 1]) ++k;
    const cswin_reduce_job job = J.j[k];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long i = (long)(blockIdx.x - J.first_block[k]) * 32 + c;
    float s0 = 0.f, s1 = 0.f;
    if (i < job.n) {
        int r = g;
        for (; r + G < job.rows; r += 2 * G) {
            s0 += job.part[(long)r * job.stride + i];
            s1 += job.part[(long)(r + G) * job.stride + i];
        }
        if (r < job.rows) s0 += job.part[(long)r * job.stride + i];
    }
    red[g][c] = s0 + s1;
    __syncthreads();
    if (g == 0 && i < job.n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < G; ++q) t += red[q][c];
        if (job.out2 && i >= job.n_first) job.out2[i - job.n_first] = t;
        else job.out[i] = t;
    }
}

// run `job` now, or hand it to the caller (deferred != NULL) to be batched by cswin_rows_sum_multi
static inline void reduce_now_or_defer(const cswin_reduce_job& job, cswin_reduce_job* deferred, hipStream_t st) {
    if (deferred) *deferred = job;
    else launch_rows_sum(job.part, job.out, job.out2, (long)job.n_first, (long)job.n, job.rows, (long)job.stride, st);
}
