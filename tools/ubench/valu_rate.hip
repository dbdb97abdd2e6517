// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the stitching kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define CH 8
typedef unsigned short v2h __attribute__((ext_vector_type(2)));
typedef short v2s __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
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
            if (OP == 18) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2h, a[i]) + __builtin_bit_cast(v2h, b));  // v_pk_add_u16
            if (OP == 19) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2h, a[i]) * __builtin_bit_cast(v2h, b) + __builtin_bit_cast(v2h, c));  // v_pk_mad_u16
            if (OP == 20) a[i] = __builtin_amdgcn_perm(a[i], b, c);         // v_perm_b32
            if (OP == 21) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2h, a[i]), __builtin_bit_cast(v2h, b), c, false);  // v_dot2_u32_u16
            if (OP == 22) { v2f t = {f[i], f[(i + 1) % CH]}; v2f u = {fb, fb}, w = {fc, fc}; t = __builtin_elementwise_fma(t, u, w); f[i] = t.x; f[(i + 1) % CH] = t.y; }  // v_pk_fma_f32
            if (OP == 23) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2h, a[i]) >> __builtin_bit_cast(v2h, 0x00020002u)) + b;  // v_pk_lshrrev_b16 + add
            if (OP == 24) f[i] = (float)(short)(a[i] >> 16) + f[i];          // cvt_f32_i32 sdwa + add
            if (OP == 25) a[i] = __builtin_amdgcn_sdot4((int)a[i], (int)b, (int)c, false);  // v_dot4_i32_i8
            if (OP == 26) a[i] = __builtin_amdgcn_alignbit(a[i], b, 16);    // v_alignbit
            if (OP == 27) a[i] = (a[i] << 16) | b;                           // v_lshl_or_b32
            if (OP == 28) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(v2s, a[i]), __builtin_bit_cast(v2s, b)));  // v_pk_add_i16 clamp
            if (OP == 29) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2s, a[i]), __builtin_bit_cast(v2s, b)));  // v_pk_max_i16
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
    run<18>("v_pk_add_u16", 1, d); run<19>("v_pk_mad_u16", 1, d); run<20>("v_perm_b32", 1, d); run<21>("v_dot2_u32_u16", 1, d);
    run<22>("v_pk_fma_f32 (0.5/elem)", 1, d); run<23>("pk_lshr16+add", 2, d); run<24>("cvt_f32_i16(sdwa)+add", 2, d); run<25>("v_dot4_i32_i8", 1, d);
    run<26>("v_alignbit", 1, d); run<27>("v_lshl_or", 1, d); run<28>("v_pk_add_i16 clamp", 1, d); run<29>("v_pk_max_i16", 1, d);
    return 0;
}
