// How fast does the chip start workgroups?  Empty kernels with the GEMM's launch shape (256 threads, 18 KB LDS) and a
// short spin of `work` cycles per workgroup.   hipcc -O3 --offload-arch=gfx950 dispatch_rate.hip -o dispatch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k(float* out, int work) {
    __shared__ float lds[LDS_BYTES / 4];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < work) {}
    if (lds[(threadIdx.x + 1) & 255] == -1.f) out[0] = 1.f;
}
template <int LDS_BYTES>
void run(int blocks, int work) {
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<LDS_BYTES><<<blocks, 256>>>(out, work); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<LDS_BYTES><<<blocks, 256>>>(out, work);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %6d B, %5d workgroups of 256 threads, %6d cycles of work each: %.1f us per launch (%.0f workgroups/us)\n", LDS_BYTES, blocks, work, ms * 100, blocks / (ms * 100));
    (void)hipFree(out);
}
int main() {
    for (int work : {0, 2000, 14000}) { run<18432>(3528, work); run<18432>(1184, work); run<36864>(3528, work); run<1024>(3528, work); }
    return 0;
}
