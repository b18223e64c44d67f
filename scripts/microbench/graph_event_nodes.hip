// Do explicit event-record NODES (hipGraphAddEventRecordNode) time a kernel inside a replayed hipGraph on this ROCm?
// (events recorded during stream capture are dropped: graph_events.hip)
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -o /tmp/gen scripts/microbench/graph_event_nodes.hip && /tmp/gen
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void spin(float *p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
int main()
{
    float *d; CK(hipMalloc(&d, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 200000);
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 400000);
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 100000);
    CK(hipStreamEndCapture(st, &g));
    size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n); CK(hipGraphGetNodes(g, nodes.data(), &n));
    // find the chain order: node with no dependencies first
    std::vector<hipGraphNode_t> chain;
    {
        size_t nr = 0; CK(hipGraphGetRootNodes(g, nullptr, &nr)); std::vector<hipGraphNode_t> r(nr); CK(hipGraphGetRootNodes(g, r.data(), &nr));
        hipGraphNode_t cur = r[0];
        for (;;) { chain.push_back(cur); size_t nd = 0; CK(hipGraphNodeGetDependentNodes(cur, nullptr, &nd)); if (!nd) break;
                   std::vector<hipGraphNode_t> dn(nd); CK(hipGraphNodeGetDependentNodes(cur, dn.data(), &nd)); cur = dn[0]; }
    }
    printf("nodes %zu, chain %zu\n", n, chain.size());
    // wrap the middle kernel: K0 -> A(e0) -> K1 -> B(e1) -> K2
    hipGraphNode_t A, B;
    CK(hipGraphRemoveDependencies(g, &chain[0], &chain[1], 1));
    CK(hipGraphRemoveDependencies(g, &chain[1], &chain[2], 1));
    CK(hipGraphAddEventRecordNode(&A, g, &chain[0], 1, e0));
    CK(hipGraphAddDependencies(g, &A, &chain[1], 1));
    CK(hipGraphAddEventRecordNode(&B, g, &chain[1], 1, e1));
    CK(hipGraphAddDependencies(g, &B, &chain[2], 1));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        float a = -1; hipError_t r1 = hipEventElapsedTime(&a, e0, e1);
        printf("rep %d: %s %.3f ms (middle kernel)\n", rep, hipGetErrorString(r1), a);
    }
    // reference: the same three kernels on the stream with events
    hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 200000);
    CK(hipEventRecord(f0, st));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 400000);
    CK(hipEventRecord(f1, st));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 100000);
    CK(hipStreamSynchronize(st));
    float b = -1; CK(hipEventElapsedTime(&b, f0, f1)); printf("stream reference: %.3f ms\n", b);
    return 0;
}
