// Shared device/host helpers for libcswin_hip (gfx950 / CDNA4 only).
#pragma once
#include "tuning.h"
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
// raw bf16 pairs travel as INTEGER vectors: hipcc 7.2 (gfx950, -O3) miscompiles __builtin_bit_cast of the elements of a 2-float
// vector (element 1 reads element 0; tools/micro/f32x2_bitcast_probe.hip shows it in the ISA), so there is no f32x2 typedef here
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bumped whenever a prototype of include/cswin_hip.h changes (the same constant is defined there; tests compare the two)
#define CSWIN_ABI_VERSION 4
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

// erf-based GELU (torch.nn.GELU() default, cswin_unet.py:24) and its derivative.  erf by Abramowitz & Stegun 7.1.26:
//   erf(u) = 1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) exp(-u^2),  t = 1 / (1 + p u),  u >= 0,  |error| <= 1.5e-7
// i.e. at the rounding level of fp32 for (1 + erf): one v_exp, one v_rcp and seven FMAs instead of libm's erff, and the
// derivative reuses the same exponential (exp(-u^2) = exp(-x^2 / 2) with u = x / sqrt(2)).  The GEMM epilogues that apply
// these are VALU-bound once the MFMAs are bf16 (fc1 forward 21.8 vs 12.2 us, fc2 data gradient 27.8 vs 16.7 us with erff).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
    e = __expf(-u * u);
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    // Phi(-|x|) = (1 - erf(|x| / sqrt 2)) / 2 = poly e / 2 is formed directly, so the negative tail keeps its relative accuracy
    // (1 - (1 - tail) cancels at fp32 epsilon).  Explicit fmaf: under -ffp-contract=fast the compiler would otherwise contract
    // 1 - tail in some instantiations and not in others, and the storage variants of one epilogue must agree bit for bit.
    const float pe = poly * e;
    cdf = x < 0.f ? 0.5f * pe : fmaf(-0.5f, pe, 1.0f);
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return fmaf(x * 0.3989422804014327f, e, cdf);
}

// ---- partial-slab reductions ------------------------------------------------------------------------------------
// out[i] = sum_r part[r * stride + i], i < n, in a fixed order (deterministic).  Used for split-K weight gradients,
// LayerNorm dgamma/dbeta, loss sums and bias sums.  Columns >= n_first go to out2[i - n_first] when out2 != NULL.
// Several producers' slabs can be reduced by ONE launch (a CSWinBlock backward has six: four split-K weight gradients
// and two LayerNorm dgamma/dbeta; as separate launches each costs ~5 us of pure launch latency).
extern "C" {
typedef struct cswin_reduce_job {
    const float* part;       // [rows][stride] partial slabs
    float* out;              // columns [0, n_first)
    float* out2;             // columns [n_first, n) (may be NULL: then everything goes to out)
    long long n_first, n, stride;
    int rows, reserved;      // reserved: set by the library (bit 0 = 16-B loads are legal)
    int conv_kk, conv_cin;   // != 0: columns [0, n_first) are a conv weight gradient in the implicit-GEMM order [Cout][k*k][Cin],
                             // stored to `out` in the nn.Conv2d order [Cout][Cin][k][k] (conv_kk = k*k, conv_cin = Cin)
} cswin_reduce_job;
}

static inline int reduce_job_vec_ok(const cswin_reduce_job& j) {
    return ((uintptr_t)j.part % 16 == 0) && j.stride % 4 == 0 && j.n % 4 == 0 && (!j.out2 || j.n_first % 4 == 0);
}

// One workgroup (256 threads) = 64 columns x 16 row groups: a wave reads 4 slab rows x 256 contiguous bytes per
// instruction with up to four 16-B loads in flight per lane; the 16 row-group sums meet in LDS.
constexpr int RS_COLS = 64, RS_G = 16;
__device__ __forceinline__ void rows_sum_block(const cswin_reduce_job& job, long blk, float (*red)[RS_COLS + 1]) {
    const int c4 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const long i0 = blk * RS_COLS + 4 * c4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (i0 < job.n) {
        const float* base = job.part + i0;
        if (job.reserved & 1) {
            int r = g;
            for (; r + 3 * RS_G < job.rows; r += 4 * RS_G) {
                s0 += *reinterpret_cast<const f32x4*>(base + (long)r * job.stride);
                s1 += *reinterpret_cast<const f32x4*>(base + (long)(r + RS_G) * job.stride);
                s2 += *reinterpret_cast<const f32x4*>(base + (long)(r + 2 * RS_G) * job.stride);
                s3 += *reinterpret_cast<const f32x4*>(base + (long)(r + 3 * RS_G) * job.stride);
            }
            for (; r < job.rows; r += RS_G) s0 += *reinterpret_cast<const f32x4*>(base + (long)r * job.stride);
        } else {
            for (int r = g; r < job.rows; r += RS_G)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + e < job.n) s0[e] += base[(long)r * job.stride + e];
        }
    }
    s0 = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[g][4 * c4 + e] = s0[e];
    __syncthreads();
    const long i = blk * RS_COLS + threadIdx.x;
    if (threadIdx.x < RS_COLS && i < job.n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RS_G; ++k) t += red[k][threadIdx.x];
        if (job.out2 && i >= job.n_first) job.out2[i - job.n_first] = t;
        else job.out[i] = t;
    }
}

// Same reduction for a convolution weight gradient whose slab columns are [Cout][k*k][Cin] (the implicit-GEMM order) while
// the parameter is [Cout][Cin][k][k]: slabs are read along their columns (coalesced), the nn.Conv2d layout is produced by the
// 4-B stores of the (small) result.  Columns >= n_first are the bias gradient as usual.
__device__ __forceinline__ void rows_sum_conv_block(const cswin_reduce_job& j, long blk, float (*red)[RS_COLS + 1]) {
    const int kk = j.conv_kk, Cin = j.conv_cin;
    float* out = j.out;
    // reduce into LDS exactly like rows_sum_block, then remap the store
    const int c4 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const long i0 = blk * RS_COLS + 4 * c4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (i0 < j.n) {
        const float* base = j.part + i0;
        if (j.reserved & 1) {
            int r = g;
            for (; r + RS_G < j.rows; r += 2 * RS_G) {
                s0 += *reinterpret_cast<const f32x4*>(base + (long)r * j.stride);
                s1 += *reinterpret_cast<const f32x4*>(base + (long)(r + RS_G) * j.stride);
            }
            for (; r < j.rows; r += RS_G) s0 += *reinterpret_cast<const f32x4*>(base + (long)r * j.stride);
        } else {
            for (int r = g; r < j.rows; r += RS_G)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + e < j.n) s0[e] += base[(long)r * j.stride + e];
        }
    }
    s0 += s1;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[g][4 * c4 + e] = s0[e];
    __syncthreads();
    const long i = blk * RS_COLS + threadIdx.x;
    if (threadIdx.x < RS_COLS && i < j.n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RS_G; ++k) t += red[k][threadIdx.x];
        if (j.out2 && i >= j.n_first) j.out2[i - j.n_first] = t;
        else if (i < j.n_first) {
            const long co = i / ((long)kk * Cin);
            const int rem = (int)(i - co * kk * Cin), tap = rem / Cin, ci = rem - tap * Cin;
            out[(co * Cin + ci) * kk + tap] = t;
        }
    }
}

extern "C" {
typedef struct cswin_wgrad_desc {
    const float* dy;         // (M, N)
    const float* x;          // (M, K)
    const float* row_scale;  // per-sample multiplier of dy rows, or NULL
    float* dw;               // (N, K)
    float* dbias;            // (N) or NULL
    void* workspace;         // cswin_linear_bwd_weight_workspace(M, N, K) bytes, 16-B aligned
    size_t ws_bytes;
    int rows_per_sample, M, N, K;
    int precision;           /* 0 = exact fp32 MFMA, 1 = bf16 operands (all problems of one launch agree) */
    int io_bf16;             /* precision 1 only: bit 0 = dy is stored as bf16, bit 1 = x is stored as bf16 */
} cswin_wgrad_desc;
}

constexpr int CSWIN_MAX_REDUCE_JOBS = 48;          // one launch's kernel arguments: 48 x 64 B + 49 x 4 B < 4 KB
struct ReduceJobs {
    cswin_reduce_job j[CSWIN_MAX_REDUCE_JOBS];
    int first_block[CSWIN_MAX_REDUCE_JOBS + 1];
    int njobs;
};

// Few slab rows (split-K weight gradients: 4 - 16 slabs of up to 1 M columns): the 16 row groups of rows_sum_block would leave
// three quarters of the threads without a row.  Here a thread owns four columns and walks all rows (every load of the
// workgroup is a full 4-KB line set, up to eight in flight per thread); one workgroup covers 1024 columns.
constexpr int RS_FEW_ROWS = 16, RS_FEW_COLS = 1024;
__device__ __forceinline__ void rows_sum_few(const cswin_reduce_job& job, long blk) {
    const long i0 = blk * RS_FEW_COLS + 4 * threadIdx.x;
    if (i0 >= job.n) return;
    const float* base = job.part + i0;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int r = 0;
    for (; r + 1 < job.rows; r += 2) {
        s0 += *reinterpret_cast<const f32x4*>(base + (long)r * job.stride);
        s1 += *reinterpret_cast<const f32x4*>(base + (long)(r + 1) * job.stride);
    }
    if (r < job.rows) s0 += *reinterpret_cast<const f32x4*>(base + (long)r * job.stride);
    s0 += s1;
    if (job.out2 && i0 >= job.n_first) *reinterpret_cast<f32x4*>(job.out2 + (i0 - job.n_first)) = s0;
    else *reinterpret_cast<f32x4*>(job.out + i0) = s0;
}
// reserved bit 1 (set by cswin_rows_sum_multi): this job runs in the few-rows mode (needs bit 0 and 16-B aligned outputs)
static inline int reduce_job_few_ok(const cswin_reduce_job& j) {
    return !j.conv_kk && reduce_job_vec_ok(j) && j.rows <= RS_FEW_ROWS && j.n >= 4 * RS_FEW_COLS && ((uintptr_t)j.out % 16 == 0) &&
           (!j.out2 || ((uintptr_t)j.out2 % 16 == 0));
}

static __global__ __launch_bounds__(256) void rows_sum_kernel(cswin_reduce_job job) {
    __shared__ float red[RS_G][RS_COLS + 1];
    if (job.conv_kk) rows_sum_conv_block(job, blockIdx.x, red);
    else if (job.reserved & 2) rows_sum_few(job, blockIdx.x);
    else rows_sum_block(job, blockIdx.x, red);
}

static inline void launch_rows_sum(const float* part, float* out, float* out2, long n_first, long n, int rows, long stride,
                                   hipStream_t st) {
    cswin_reduce_job job = {part, out, out2, n_first, n, stride, rows, 0, 0, 0};
    const int few = reduce_job_few_ok(job);
    job.reserved = reduce_job_vec_ok(job) | (few ? 2 : 0);
    const long cols = few ? RS_FEW_COLS : RS_COLS;
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)((n + cols - 1) / cols)), dim3(256), 0, st, job);
}

// workgroup `blk` of a job table (256 threads; `red` = 16 x 65 floats of LDS): the job is the last k with first_block[k] <= blk
__device__ __forceinline__ void rows_sum_dispatch(const cswin_reduce_job* j, const int* first_block, int njobs, int blk,
                                                  float (*red)[RS_COLS + 1]) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk >= first_block[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int k = lo;
    if (j[k].conv_kk) rows_sum_conv_block(j[k], blk - first_block[k], red);
    else if (j[k].reserved & 2) rows_sum_few(j[k], blk - first_block[k]);
    else rows_sum_block(j[k], blk - first_block[k], red);
}

static __global__ __launch_bounds__(256) void rows_sum_multi_kernel(ReduceJobs J) {
    __shared__ float red[RS_G][RS_COLS + 1];
    rows_sum_dispatch(J.j, J.first_block, J.njobs, (int)blockIdx.x, red);
}

// Reductions that ride at the end of another kernel's grid (gemm_block_tail_kernel): memory-bound workgroups of a few hundred
// cycles each beside matrix-pipe-bound ones, instead of a launch of their own
constexpr int CSWIN_TAIL_RIDER_JOBS = 16;
struct ReduceRiders {
    cswin_reduce_job j[CSWIN_TAIL_RIDER_JOBS];
    int first_block[CSWIN_TAIL_RIDER_JOBS + 1];
    int njobs;
};
// validate + flag the jobs of a table and give every job its workgroup range; returns the number of workgroups or -1
static inline int fill_reduce_table(const cswin_reduce_job* jobs, int njobs, cswin_reduce_job* out, int* first_block) {
    int blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        if (!(jobs[i].part && jobs[i].out && jobs[i].n > 0 && jobs[i].rows > 0)) return -1;
        if ((jobs[i].conv_kk == 0) != (jobs[i].conv_cin == 0) || jobs[i].conv_kk < 0) return -1;
        out[i] = jobs[i];
        const int few = reduce_job_few_ok(jobs[i]);
        out[i].reserved = reduce_job_vec_ok(jobs[i]) | (few ? 2 : 0);
        first_block[i] = blocks;
        blocks += few ? (int)((jobs[i].n + RS_FEW_COLS - 1) / RS_FEW_COLS) : (int)((jobs[i].n + RS_COLS - 1) / RS_COLS);
    }
    first_block[njobs] = blocks;
    return blocks;
}

// run `job` now, or hand it to the caller (deferred != NULL) to be batched by cswin_rows_sum_multi
static inline void reduce_now_or_defer(const cswin_reduce_job& job, cswin_reduce_job* deferred, hipStream_t st) {
    if (deferred) *deferred = job;
    else launch_rows_sum(job.part, job.out, job.out2, (long)job.n_first, (long)job.n, job.rows, (long)job.stride, st);
}
