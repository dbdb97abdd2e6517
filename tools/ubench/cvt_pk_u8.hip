// Does v_cvt_pk_u8_f32 round to nearest even by itself (then a v_rndne_f32 in front of it is redundant)?  EXHAUSTIVE over every
// float bit pattern (all 2^32: NaNs, infinities, negatives included): cvt_pk_u8(x) against cvt_pk_u8(rint(x)) and against the
// arithmetic definition saturate_u8(rint(x)) with NaN -> 0.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* diff_rint, unsigned long long* diff_def, unsigned* first)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float x = __uint_as_float((unsigned)u);
        const unsigned a = __builtin_amdgcn_cvt_pk_u8_f32(x, 0u, 0u);
        const unsigned b = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(x), 0u, 0u);
        const float r = __builtin_rintf(x);
        const unsigned d = (x != x) ? 0u : (r <= 0.f ? 0u : (r >= 255.f ? 255u : (unsigned)r));
        if (a != b) { if (atomicAdd(diff_rint, 1ull) == 0) *first = (unsigned)u; }
        if (a != d) atomicAdd(diff_def, 1ull);
    }
}
int main()
{
    unsigned long long *d, h[2] = {0, 0}; unsigned *f, hf = 0;
    hipMalloc(&d, 16); hipMalloc(&f, 4); hipMemset(d, 0, 16); hipMemset(f, 0, 4);
    hipLaunchKernelGGL(k, dim3(256 * 32), dim3(256), 0, 0, d, d + 1, f);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
    printf("all 2^32 floats: cvt_pk_u8(x) != cvt_pk_u8(rint(x)) for %llu (first bits 0x%08x); != saturate_u8(rint(x)), NaN -> 0, for %llu\n", h[0], hf, h[1]);
    return 0;
}
