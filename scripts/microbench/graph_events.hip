// Can HIP events recorded INSIDE a captured stream time the kernels of a replayed hipGraph on this ROCm?
// ROCm 7.2 / MI355X (round 3): no -- the captured graph holds the two kernel nodes only and hipEventElapsedTime returns
// "invalid resource handle"; bench.py therefore times kernels on the launch-by-launch path and the graph replay separately.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_events scripts/microbench/graph_events.hip && /tmp/graph_events
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void spin(float *p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
int main()
{
    float *d; CK(hipMalloc(&d, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e[3]; for (auto &x : e) CK(hipEventCreate(&x));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(e[0], st));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 200000);
    CK(hipEventRecord(e[1], st));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 400000);
    CK(hipEventRecord(e[2], st));
    CK(hipStreamEndCapture(st, &g));
    size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n)); printf("nodes %zu\n", n);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        float a = -1, b = -1;
        hipError_t r1 = hipEventElapsedTime(&a, e[0], e[1]), r2 = hipEventElapsedTime(&b, e[1], e[2]);
        printf("rep %d: %s %.3f ms, %s %.3f ms\n", rep, hipGetErrorString(r1), a, hipGetErrorString(r2), b);
    }
    return 0;
}
