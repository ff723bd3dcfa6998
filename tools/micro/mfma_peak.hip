// Raw MFMA issue-rate probe (no memory traffic): how many fp32 TFLOP/s can the chip sustain with W waves per SIMD and
// C independent accumulator chains per wave?   hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool BIG>
__global__ void k(float* out, int iters) {
    f32x16 acc[CHAINS];
    f32x4 acc4[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { for (int e = 0; e < 16; ++e) acc[c][e] = 0.f; acc4[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (BIG) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            else acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) { for (int e = 0; e < 16; ++e) s += acc[c][e]; for (int e = 0; e < 4; ++e) s += acc4[c][e]; }
    if (s == 123.456f) out[0] = s;
}

template <int CHAINS, bool BIG>
void run(int waves_per_simd) {
    float* out; hipMalloc(&out, 4);
    const int iters = 4000, blocks = 256 * waves_per_simd, threads = 256;      // 4 waves per block = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS, BIG><<<blocks, threads>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<CHAINS, BIG><<<blocks, threads>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * CHAINS * (BIG ? 4096.0 : 2048.0);
    printf("%s chains %d waves/SIMD %d: %.1f TFLOP/s (%.3f ms)\n", BIG ? "32x32x2 " : "16x16x4 ", CHAINS, waves_per_simd, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) { run<1, true>(w); run<2, true>(w); run<4, true>(w); }
    for (int w : {1, 2, 4}) { run<1, false>(w); run<2, false>(w); run<4, false>(w); }
    return 0;
}
