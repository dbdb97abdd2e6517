// What does a cross-stream dependency cost on this stack?  A chain of K dependent kernels (each `work` iterations long) run
//   (a) on one stream;
//   (b) alternating between two streams, joined by hipEventRecord + hipStreamWaitEvent;
//   (c) alternating, joined by hipStreamWriteValue32 + hipStreamWaitValue32 (signal memory);
// wall time per kernel from launch of the first to completion of the last.   hipcc --offload-arch=gfx950 -O2 stream_edge.hip -o stream_edge
#include <hip/hip_runtime.h>
#include <chrono>
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
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    const int K = 40;
    std::vector<hipEvent_t> ev(K);
    for (int k = 0; k < K; k++) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    uint32_t* sig = nullptr; bool have_sig = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) == hipSuccess;
    uint32_t seq = 0;
    if (have_sig) { CK(hipStreamWriteValue32(s[0], sig, 0, 0)); CK(hipStreamSynchronize(s[0])); }
    for (int work : {100, 20000, 200000}) {
        for (int mode = 0; mode < (have_sig ? 3 : 2); mode++) {
            double best = 1e30;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < K; k++) {
                    hipStream_t st = mode == 0 ? s[0] : s[k & 1];
                    if (mode == 1 && k > 0) CK(hipStreamWaitEvent(st, ev[k - 1], 0));
                    if (mode == 2 && k > 0) CK(hipStreamWaitValue32(st, sig, seq, hipStreamWaitValueGte, 0xffffffffu));
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, work);
                    if (mode == 1) CK(hipEventRecord(ev[k], st));
                    if (mode == 2) CK(hipStreamWriteValue32(st, sig, ++seq, 0));
                }
                CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (us < best) best = us;
            }
            printf("work %6d  %-28s %8.1f us per kernel (chain of %d)\n", work, mode == 0 ? "one stream" : (mode == 1 ? "two streams, events" : "two streams, write/wait value"), best / K, K);
        }
    }
    return 0;
}
