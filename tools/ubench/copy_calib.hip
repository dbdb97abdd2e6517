// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on this machine: kernels with a known byte count in the access
// widths the stitching kernels use (16 B/lane streaming, 12 B/lane gathers, 4 B/lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
__global__ void copy4(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
struct __attribute__((aligned(4))) U3 { uint32_t x, y, z; };
__global__ void read12(const uint8_t* __restrict__ a, uint32_t* __restrict__ b, size_t n)  // 12 B per lane at stride 12, 4 B out
{ size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { U3 v = *reinterpret_cast<const U3*>(a + i * 12); b[i] = v.x ^ v.y ^ v.z; } }
int main()
{
    const size_t bytes = 768ull << 20;
    void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    for (int rep = 0; rep < 2; rep++) {
        size_t n16 = bytes / 16; hipLaunchKernelGGL(copy16, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16);
        size_t n4 = bytes / 4; hipLaunchKernelGGL(copy4, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, n4);
        size_t n12 = bytes / 12; hipLaunchKernelGGL(read12, dim3((n12 + 255) / 256), dim3(256), 0, 0, (const uint8_t*)a, (uint32_t*)b, n12);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: copy16 read %zu write %zu; copy4 same; read12 read %zu write %zu\n", bytes, bytes, bytes, bytes / 3);
    return 0;
}
