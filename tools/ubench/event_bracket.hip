// How much longer than the kernel is a hipEventRecord / kernel / hipEventRecord bracket?  The same kernel timed (a) by events recorded
// before and after the launch, (b) by events attached to the launch (hipExtLaunchKernelGGL: the dispatch's own begin / end stamps),
// alone on its stream and right behind another kernel (as inside a panorama).   hipcc --offload-arch=gfx950 -O2 event_bracket.hip -o event_bracket
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(float* p, int work)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < work; i++) v = v * 1.0000001f + 1e-7f;
    p[threadIdx.x] = v;
}

int main()
{
    float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int work = 10000;  // ~170 us
    for (int behind = 0; behind < 2; behind++)
        for (int mode = 0; mode < 2; mode++) {
            std::vector<float> t;
            for (int rep = 0; rep < 60; rep++) {
                if (behind) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s, d, work);
                else CK(hipStreamSynchronize(s));
                if (mode == 0) {
                    CK(hipEventRecord(a, s));
                    hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s, d, work);
                    CK(hipEventRecord(b, s));
                } else {
                    hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s, a, b, 0, d, work);
                }
                CK(hipStreamSynchronize(s));
                float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
                t.push_back(ms * 1e3f);
            }
            std::sort(t.begin(), t.end());
            printf("%-22s %-28s median %7.2f us  min %7.2f  max %7.2f\n", behind ? "behind another kernel" : "on an idle stream", mode ? "events attached to launch" : "record / launch / record", t[t.size() / 2], t[0], t.back());
        }
    return 0;
}
