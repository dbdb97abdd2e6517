// Cost of the vector-memory access patterns the stitching kernels use (wave64, every lane loads a window at a fixed
// byte stride from its neighbour): aligned vs dword-aligned wide loads, narrow loads, gathers of 12 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef v4u __attribute__((aligned(4))) v4u_a4;
typedef v2u __attribute__((aligned(4))) v2u_a4;
struct __attribute__((aligned(4))) U3 { uint32_t x, y, z; };
#define ROWS 64
template <int PAT>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ base, uint32_t* __restrict__ out, size_t row_bytes, int shift)
{
    // each wave walks ROWS rows; lane i reads its window of row r
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    uint32_t acc = 0;
    for (int r = 0; r < ROWS; r++) {
        const uint8_t* row = base + ((size_t)wave * ROWS + r) * row_bytes + shift;
        if (PAT == 0) { v4u v = *(const v4u*)(row + lane * 16); acc += v.x ^ v.y ^ v.z ^ v.w; }                       // 16 B aligned, stride 16
        if (PAT == 1) { v4u a = *(const v4u_a4*)(row + lane * 24), b = *(const v4u_a4*)(row + lane * 24 + 16); acc += a.x ^ a.w ^ b.x ^ b.y; }  // 2 x16 B at stride 24
        if (PAT == 2) { const uint32_t* q = (const uint32_t*)(row + lane * 24); acc += q[0] ^ q[1] ^ q[2] ^ q[3] ^ q[4] ^ q[5]; } // 6 dwords (merged by the compiler?)
        if (PAT == 3) { const v2u_a4* q = (const v2u_a4*)(row + lane * 24); v2u a = q[0], b = q[1], c = q[2]; acc += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y; } // 3 x 8 B
        if (PAT == 4) { v2u a = *(const v2u*)(row + lane * 8 + 8); unsigned short l = *(const unsigned short*)(row + lane * 8 + 6), h = *(const unsigned short*)(row + lane * 8 + 16); acc += a.x ^ a.y ^ l ^ h; } // 8 B + 2 shorts, stride 8
        if (PAT == 5) { v4u a = *(const v4u_a4*)(row + lane * 8 + 4); acc += a.x ^ a.y ^ a.z ^ a.w; }                   // 16 B at 4-byte alignment, stride 8
        if (PAT == 6) { U3 a = *(const U3*)(row + lane * 12); acc += a.x ^ a.y ^ a.z; }                                 // 12 B stride 12 (warp gathers)
        if (PAT == 7) { v4u a = *(const v4u*)(row + lane * 32), b = *(const v4u*)(row + lane * 32 + 16); acc += a.x ^ a.w ^ b.x ^ b.y; }  // 2 x 16 B aligned, stride 32
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int PAT> void run(const char* name, const uint8_t* a, uint32_t* o, size_t row_bytes, int shift, double useful_per_lane)
{
    const int blocks = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, a, o, row_bytes, shift);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, a, o, row_bytes, shift);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lanes = (double)blocks * 256 * ROWS;
    printf("%-44s shift %d: %7.3f ms  %6.2f ns per wave-row  %7.1f GB/s useful\n", name, shift, ms, ms * 1e6 / (lanes / 64), lanes * useful_per_lane / ms / 1e6);
}
int main()
{
    const size_t row_bytes = 4096, total = (size_t)4096 * 4 * ROWS * row_bytes + 65536;
    uint8_t* a; uint32_t* o; hipMalloc(&a, total); hipMalloc(&o, 4096 * 256 * 4); hipMemset(a, 1, total);
    for (int shift = 0; shift <= 4; shift += 4) {
        run<0>("16 B aligned, stride 16", a, o, row_bytes, 0, 16);
        run<1>("2 x 16 B dword-aligned, stride 24 (img rows)", a, o, row_bytes, shift, 24);
        run<2>("6 dwords, stride 24", a, o, row_bytes, shift, 24);
        run<3>("3 x 8 B dword-aligned, stride 24", a, o, row_bytes, shift, 24);
        run<4>("8 B + 2 shorts, stride 8 (pyrUp taps)", a, o, row_bytes, 0, 12);
        run<5>("16 B at 4-byte alignment, stride 8", a, o, row_bytes, 0, 12);
        run<6>("12 B, stride 12 (warp gathers)", a, o, row_bytes, shift, 12);
        run<7>("2 x 16 B aligned, stride 32", a, o, row_bytes, 0, 32);
    }
    return 0;
}
