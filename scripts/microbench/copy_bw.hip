// copy_bw.hip -- what HBM rate do row-shaped streaming kernels reach with 8-byte vs 16-byte accesses?
// (same grid shape as k_fwd_row4096: one 32 KB row per 256-thread workgroup, 16 accesses per thread)
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip ; run: ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void copy_f2(const float2 *in, float2 *out)
{
    const float2 *r = in + (size_t)blockIdx.x * 4096;
    float2 *w = out + (size_t)blockIdx.x * 4096;
    float2 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = r[threadIdx.x + 256 * k];
#pragma unroll
    for (int k = 0; k < 16; k++) w[threadIdx.x + 256 * k] = make_float2(v[k].x + 1.0f, v[k].y);
}

__global__ __launch_bounds__(256) void copy_f4(const float4 *in, float4 *out)
{
    const float4 *r = in + (size_t)blockIdx.x * 2048;
    float4 *w = out + (size_t)blockIdx.x * 2048;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = r[threadIdx.x + 256 * k];
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k].x += 1.0f; w[threadIdx.x + 256 * k] = v[k]; }
}

__global__ __launch_bounds__(256) void read_f2(const float2 *in, float *out)
{
    const float2 *r = in + (size_t)blockIdx.x * 4096;
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { float2 v = r[threadIdx.x + 256 * k]; acc += v.x + v.y; }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void read_f4(const float4 *in, float *out)
{
    const float4 *r = in + (size_t)blockIdx.x * 2048;
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { float4 v = r[threadIdx.x + 256 * k]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

// write only, and the pair kernel's mix (two rows read per row written), same row shape
__global__ __launch_bounds__(256) void write_f2(float2 *out, float seed)
{
    float2 *w = out + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int k = 0; k < 16; k++) w[threadIdx.x + 256 * k] = make_float2(seed + k, seed - threadIdx.x);
}

__global__ __launch_bounds__(256) void read2_write1_f2(const float2 *in, float2 *out, size_t half_rows)
{
    const float2 *r0 = in + (size_t)blockIdx.x * 4096, *r1 = in + ((size_t)blockIdx.x + half_rows) * 4096;
    float2 *w = out + (size_t)blockIdx.x * 4096;
    float2 v[16], u[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k] = r0[threadIdx.x + 256 * k]; u[k] = r1[threadIdx.x + 256 * k]; }
#pragma unroll
    for (int k = 0; k < 16; k++) w[threadIdx.x + 256 * k] = make_float2(v[k].x + u[k].y, v[k].y - u[k].x);
}

int main()
{
    const size_t rows = 76032;                    // 297 station-windows x 256 rows
    const size_t bytes = rows * 4096 * sizeof(float2);
    void *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch, double moved) {
        for (int i = 0; i < 2; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s %7.1f us  %6.2f TB/s\n", name, ms * 100, moved / (ms / 10 * 1e-3) / 1e12);
    };
    time("copy f2", [&] { copy_f2<<<rows, 256>>>((float2 *)a, (float2 *)b); }, 2.0 * bytes);
    time("copy f4", [&] { copy_f4<<<rows, 256>>>((float4 *)a, (float4 *)b); }, 2.0 * bytes);
    time("read f2", [&] { read_f2<<<rows, 256>>>((float2 *)a, (float *)b); }, 1.0 * bytes);
    time("read f4", [&] { read_f4<<<rows, 256>>>((float4 *)a, (float *)b); }, 1.0 * bytes);
    time("write f2", [&] { write_f2<<<rows, 256>>>((float2 *)b, 1.0f); }, 1.0 * bytes);
    // 2 reads : 1 write over half the rows each (the pair kernel's mix): 1.5 x bytes moved in all
    time("2R:1W f2", [&] { read2_write1_f2<<<rows / 2, 256>>>((float2 *)a, (float2 *)b, rows / 2); }, 1.5 * bytes);
    return 0;
}
