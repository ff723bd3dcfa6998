// hipcc 7.2 / gfx950 / -O3: k<true> loads ONE dword per value and returns {e0, e1, e0, e1}: the bit casts of raw[1] read raw[0].
// With "typedef unsigned f32x2" (an integer vector) the code is correct.  hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool Q16> struct QRaw { typedef f32x4 type; };
template <> struct QRaw<true> { typedef f32x2 type; };
struct P { const float* qkv; int n; };
template <bool Q16> __device__ __forceinline__ typename QRaw<Q16>::type ldq_raw(const P& p, const float* elem_ptr) {
    if constexpr (Q16) return *reinterpret_cast<const f32x2*>(reinterpret_cast<const __bf16*>(p.qkv) + (elem_ptr - p.qkv));
    else return *reinterpret_cast<const f32x4*>(elem_ptr);
}
__device__ __forceinline__ f32x4 qcv(f32x4 raw) { return raw; }
__device__ __forceinline__ f32x4 qcv(f32x2 raw) {
    const unsigned a = __builtin_bit_cast(unsigned, raw[0]), b = __builtin_bit_cast(unsigned, raw[1]);
    return f32x4{__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xffff0000u),
                 __builtin_bit_cast(float, b << 16), __builtin_bit_cast(float, b & 0xffff0000u)};
}
template <bool Q16> __global__ void k(P p, f32x4* out) {
    const float* src = p.qkv + 4 * threadIdx.x;
    typename QRaw<Q16>::type a = {}, b = {};
    if ((int)threadIdx.x < p.n) { a = ldq_raw<Q16>(p, src); b = ldq_raw<Q16>(p, src + 1024); }
    out[threadIdx.x] = qcv(a) + qcv(b);
}
template __global__ void k<true>(P, f32x4*);
