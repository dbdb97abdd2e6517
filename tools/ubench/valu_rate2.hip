// Micro-benchmark 2: issue cost (cycles per wave64 instruction per SIMD) of single VALU / LDS instructions, written as inline asm so
// that the compiler can neither fold a chain nor pick another opcode.  8 independent chains per lane, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 1024
#define REP8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i * 977;
    uint32_t b = seed * 3 + 1, c = seed ^ 0x55;
    unsigned long long q[4];
    for (int i = 0; i < 4; i++) q[i] = ((unsigned long long)a[i] << 32) | a[i + 4];
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = b; lds[threadIdx.x + 512] = c; lds[threadIdx.x + 768] = seed;
    __syncthreads();
    for (int it = 0; it < ITERS; it++) {
#define V3(op) asm volatile(op " %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c)); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c)); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(a[4]) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" : "+v"(a[5]) : "v"(b), "v"(c)); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(a[6]) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" : "+v"(a[7]) : "v"(b), "v"(c));
#define V2(op) asm volatile(op " %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a[1]) : "v"(b)); \
               asm volatile(op " %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a[3]) : "v"(b)); \
               asm volatile(op " %0, %0, %1" : "+v"(a[4]) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a[5]) : "v"(b)); \
               asm volatile(op " %0, %0, %1" : "+v"(a[6]) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a[7]) : "v"(b));
#define V1(op) asm volatile(op " %0, %0" : "+v"(a[0])); asm volatile(op " %0, %0" : "+v"(a[1])); asm volatile(op " %0, %0" : "+v"(a[2])); \
               asm volatile(op " %0, %0" : "+v"(a[3])); asm volatile(op " %0, %0" : "+v"(a[4])); asm volatile(op " %0, %0" : "+v"(a[5])); \
               asm volatile(op " %0, %0" : "+v"(a[6])); asm volatile(op " %0, %0" : "+v"(a[7]));
#define P3(op) asm volatile(op " %0, %0, %1, %2" : "+v"(q[0]) : "v"(q[3]), "v"(q[3])); asm volatile(op " %0, %0, %1, %2" : "+v"(q[1]) : "v"(q[3]), "v"(q[3])); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(q[2]) : "v"(q[3]), "v"(q[3])); asm volatile(op " %0, %0, %1, %2" : "+v"(q[0]) : "v"(q[3]), "v"(q[3])); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(q[1]) : "v"(q[3]), "v"(q[3])); asm volatile(op " %0, %0, %1, %2" : "+v"(q[2]) : "v"(q[3]), "v"(q[3])); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(q[0]) : "v"(q[3]), "v"(q[3])); asm volatile(op " %0, %0, %1, %2" : "+v"(q[1]) : "v"(q[3]), "v"(q[3]));
#define P2(op) asm volatile(op " %0, %0, %1" : "+v"(q[0]) : "v"(q[3])); asm volatile(op " %0, %0, %1" : "+v"(q[1]) : "v"(q[3])); \
               asm volatile(op " %0, %0, %1" : "+v"(q[2]) : "v"(q[3])); asm volatile(op " %0, %0, %1" : "+v"(q[0]) : "v"(q[3])); \
               asm volatile(op " %0, %0, %1" : "+v"(q[1]) : "v"(q[3])); asm volatile(op " %0, %0, %1" : "+v"(q[2]) : "v"(q[3])); \
               asm volatile(op " %0, %0, %1" : "+v"(q[0]) : "v"(q[3])); asm volatile(op " %0, %0, %1" : "+v"(q[1]) : "v"(q[3]));
        if (OP == 0) { V2("v_add_u32") }
        if (OP == 1) { V2("v_mul_u32_u24") }
        if (OP == 2) { V3("v_mad_u32_u24") }
        if (OP == 3) { V2("v_mul_lo_u32") }
        if (OP == 4) { V3("v_lshl_add_u32") }
        if (OP == 5) { V3("v_lshl_or_b32") }
        if (OP == 6) { V3("v_mad_u32_u16") }
        if (OP == 7) { V2("v_pack_b32_f16") }
        if (OP == 8) { V2("v_and_b32") }
        if (OP == 9) { V3("v_bfe_u32") }
        if (OP == 10) { V3("v_perm_b32") }
        if (OP == 11) { V3("v_alignbyte_b32") }
        if (OP == 12) { V3("v_dot2_u32_u16") }
        if (OP == 13) { V2("v_pk_mul_lo_u16") }
        if (OP == 14) { V3("v_pk_mad_u16") }
        if (OP == 15) { V3("v_fma_f32") }
        if (OP == 16) { P3("v_pk_fma_f32") }
        if (OP == 17) { P2("v_pk_mul_f32") }
        if (OP == 18) { P2("v_pk_add_f32") }
        if (OP == 19) { V1("v_rcp_f32") }
        if (OP == 20) { V1("v_cvt_f32_ubyte1") }
        if (OP == 21) { V3("v_min3_u32") }
        if (OP == 22) { V2("v_min_u32") }
        if (OP == 23) { V3("v_mad_i32_i24") }
        if (OP == 24) { V2("v_lshlrev_b32") }
        if (OP == 25) { V3("v_add3_u32") }
        if (OP == 26) { V3("v_and_or_b32") }
        if (OP == 27) { V3("v_bfi_b32") }
        if (OP == 28) { V2("v_mul_f32") }
        if (OP == 29) { V3("v_xad_u32") }
        if (OP == 30) { V1("v_cvt_f32_u32") }
        if (OP == 31) { V1("v_cvt_u32_f32") }
        if (OP == 32) { V3("v_med3_i32") }
        if (OP == 33) { V3("v_dot4_u32_u8") }
        if (OP == 34) { V2("v_pk_add_u16") }
        if (OP == 35) { V3("v_mad_u16") }
        if (OP == 36) { V2("v_mul_lo_u16") }
        if (OP == 37) { V3("v_sad_u32") }
        if (OP == 38) { V3("v_lerp_u8") }
        if (OP == 39) { V2("v_mul_hi_u32") }
        if (OP == 40) {  // ds_write_b8 x 8
            uint32_t ad = (threadIdx.x * 3) & 1023;
            asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %1 offset:1\n ds_write_b8 %0, %1 offset:2\n ds_write_b8 %0, %1 offset:192\n"
                         "ds_write_b8 %0, %1 offset:193\n ds_write_b8 %0, %1 offset:194\n ds_write_b8 %0, %1 offset:384\n ds_write_b8 %0, %1 offset:385\n s_waitcnt lgkmcnt(0)" :: "v"(ad), "v"(a[0]) : "memory");
        }
        if (OP == 41) {  // ds_write_b32 x 8
            uint32_t ad = (threadIdx.x * 4) & 1023;
            asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:2048\n ds_write_b32 %0, %1 offset:3072\n"
                         "ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:2048\n ds_write_b32 %0, %1 offset:3072\n s_waitcnt lgkmcnt(0)" :: "v"(ad), "v"(a[0]) : "memory");
        }
        if (OP == 42) {  // ds_read_b32 x 8 (table lookups by a 5-bit index)
            uint32_t ad = (a[0] & 31) * 4;
            uint32_t r0, r1, r2, r3;
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:128\n ds_read_b32 %2, %4 offset:256\n ds_read_b32 %3, %4 offset:384\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(ad) : "memory");
            asm volatile("ds_read_b32 %0, %4 offset:512\n ds_read_b32 %1, %4 offset:640\n ds_read_b32 %2, %4 offset:768\n ds_read_b32 %3, %4 offset:896\n s_waitcnt lgkmcnt(0)"
                         : "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]) : "v"(ad) : "memory");
            a[0] += r0 + r1 + r2 + r3;
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r += a[i];
    for (int i = 0; i < 4; i++) r += (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = r + lds[(threadIdx.x * 5) & 1023];
}
template <int OP> void run(const char* name, uint32_t* d)
{
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * ITERS * 8 / 1024.0;
    printf("%-22s %8.3f ms -> %6.2f cycles per wave-instruction per SIMD @2.4 GHz\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_mul_u32_u24", d); run<2>("v_mad_u32_u24", d); run<3>("v_mul_lo_u32", d); run<4>("v_lshl_add_u32", d);
    run<5>("v_lshl_or_b32", d); run<6>("v_mad_u32_u16", d); run<7>("v_pack_b32_f16", d); run<8>("v_and_b32", d); run<9>("v_bfe_u32", d);
    run<10>("v_perm_b32", d); run<11>("v_alignbyte_b32", d); run<12>("v_dot2_u32_u16", d); run<13>("v_pk_mul_lo_u16", d); run<14>("v_pk_mad_u16", d);
    run<15>("v_fma_f32", d); run<16>("v_pk_fma_f32", d); run<17>("v_pk_mul_f32", d); run<18>("v_pk_add_f32", d); run<19>("v_rcp_f32", d);
    run<20>("v_cvt_f32_ubyte1", d); run<21>("v_min3_u32", d); run<22>("v_min_u32", d); run<23>("v_mad_i32_i24", d); run<24>("v_lshlrev_b32", d);
    run<25>("v_add3_u32", d); run<26>("v_and_or_b32", d); run<27>("v_bfi_b32", d); run<28>("v_mul_f32", d); run<29>("v_xad_u32", d);
    run<30>("v_cvt_f32_u32", d); run<31>("v_cvt_u32_f32", d); run<32>("v_med3_i32", d); run<33>("v_dot4_u32_u8", d); run<34>("v_pk_add_u16", d);
    run<35>("v_mad_u16", d); run<36>("v_mul_lo_u16", d); run<37>("v_sad_u32", d); run<38>("v_lerp_u8", d); run<39>("v_mul_hi_u32", d);
    run<40>("ds_write_b8 (x8)", d); run<41>("ds_write_b32 (x8)", d); run<42>("ds_read_b32 (x8)", d);
    return 0;
}
