// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the stitching kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define CH 8
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    uint32_t a[CH];
    float f[CH];
    unsigned long long q[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + threadIdx.x * 7 + i; f[i] = (float)(a[i] & 1023) + 0.5f; q[i] = ((unsigned long long)a[i] << 20) + i; }
    uint32_t b = seed * 3 + 1, c = seed ^ 0x55;
    float fb = 1.0001f, fc = 0.5f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) a[i] = a[i] + b;                                   // v_add_u32
            if (OP == 1) a[i] = __umul24(a[i], b) + c;                      // v_mad_u32_u24
            if (OP == 2) a[i] = a[i] * b;                                   // v_mul_lo_u32
            if (OP == 3) f[i] = (float)(int)f[i] + fc;                      // cvt_i32_f32 + cvt_f32_i32 + add
            if (OP == 4) a[i] = (uint32_t)min(max((int)a[i], -32768), 32767) + b;  // v_med3_i32 + add
            if (OP == 5) q[i] = q[i] + ((unsigned long long)b << 2);        // v_lshl_add_u64
            if (OP == 6) q[i] = (unsigned long long)(uint32_t)q[i] * b + c; // v_mad_u64_u32
            if (OP == 7) a[i] = ((a[i] >> 8) & 255u) + b;                   // v_bfe_u32 + add
            if (OP == 8) f[i] = __fmaf_rn(f[i], fb, fc);                    // v_fma_f32
            if (OP == 9) f[i] = __fmul_rn(f[i], fb);                        // v_mul_f32
            if (OP == 10) f[i] = __fdiv_rn(f[i], fb);                       // IEEE div sequence
            if (OP == 11) a[i] = __builtin_amdgcn_alignbyte(a[i], b, c & 3);// v_alignbyte
            if (OP == 12) f[i] = __builtin_amdgcn_rcpf(f[i]) + fc;          // v_rcp_f32 + add
            if (OP == 13) a[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a[i] & 252), (int)b) + a[i]; // bpermute + add
            if (OP == 14) f[i] = (float)((a[i] >> 16) & 255u) * fb + f[i];  // cvt_f32_ubyte2 + mul + add
            if (OP == 15) a[i] = (uint32_t)((int)(short)(a[i] & 0xffff)) + b; // v_bfe_i32 / sext + add
            if (OP == 16) f[i] = rintf(f[i]) + fc;                          // v_rndne + add
            if (OP == 17) a[i] = (a[i] > b) ? c : a[i] + 1;                 // cmp + cndmask + add
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CH; i++) r += a[i] + (uint32_t)f[i] + (uint32_t)q[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> void run(const char* name, int ops_per_iter, uint32_t* d)
{
    const int blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * ITERS * CH * ops_per_iter;  // wave-instructions (approx)
    double per_simd = winstr / 1024.0;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-op per SIMD (%.2f cycles @2.4GHz, ops/iter=%d)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, ops_per_iter);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", 1, d); run<1>("v_mad_u32_u24", 1, d); run<2>("v_mul_lo_u32", 1, d); run<3>("cvt_i32_f32+cvt_f32_i32+add", 3, d);
    run<4>("med3_i32+add", 2, d); run<5>("v_lshl_add_u64", 1, d); run<6>("v_mad_u64_u32", 1, d); run<7>("bfe_u32+add", 2, d);
    run<8>("v_fma_f32", 1, d); run<9>("v_mul_f32", 1, d); run<10>("fdiv_rn (seq ~11)", 11, d); run<11>("v_alignbyte", 1, d);
    run<12>("rcp+add", 2, d); run<13>("ds_bpermute+add", 2, d); run<14>("cvt_ubyte+mul+add", 3, d); run<15>("sext16+add", 2, d);
    run<16>("rndne+add", 2, d); run<17>("cmp+cndmask+add", 3, d);
    return 0;
}
