// lds_dma_probe.hip -- how fast does ONE wave per CU get rows into LDS, and what bounds it?
// One 64-thread workgroup per CU (64 KB of LDS each, so no two share a CU) streams 1 KB pieces (the staged column walk's
// LDS-DMA: global_load_lds_dwordx4, lanes 0..31 and 32..63 two 512-byte runs) with at most INFLIGHT instructions outstanding,
// (a) every workgroup its own region of a 4 GB buffer (first touch: memory latency), (b) all workgroups the same 2 MB
// (L2 hits), and the same with plain global_load_dwordx4 into registers.  Prints cycles per 1 KB instruction and the
// aggregate rate.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip ; run: ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int INFLIGHT>
__device__ __forceinline__ void wait_vm()
{
    if constexpr (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (INFLIGHT == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (INFLIGHT == 32) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(59)" ::: "memory");
}

// n pieces of 1 KB; piece i of workgroup b comes from src + (b * wg_stride + i * 1024) % span (two 512-byte halves 64 KB apart
// would be closer to the walk; one contiguous KB keeps the probe simple)
template <int INFLIGHT, bool DMA>
__global__ __launch_bounds__(64) void probe(const char *src, unsigned long long wg_stride, unsigned long long span, int n,
                                            unsigned long long *cycles, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // 64 KB
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
    const unsigned int lds0 = (unsigned int)(uintptr_t)((lds_ptr)ring);
    const unsigned int lane = threadIdx.x;
    const unsigned long long base = ((unsigned long long)blockIdx.x * wg_stride) & (span - 1);      // span: a power of two
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        const unsigned long long off = (base + (unsigned long long)i * 1024ull) & (span - 1);
        const char *p = src + off;
        if constexpr (DMA) {
            const unsigned int dst = lds0 + (unsigned int)(i & 15) * 1024u;      // (16 KB of the ring: the LDS size only sets the occupancy)
            const unsigned int voff = 16u * lane;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(p), "s"(dst) : "memory");
            wait_vm<INFLIGHT>();
        } else {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f v;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(16u * lane), "s"(p) : "memory");
            wait_vm<INFLIGHT>();      // (the value is only looked at after the last wait)
            if (i + 1 == n) acc = make_float4(v.x, v.y, v.z, v.w);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    if (sink && acc.x == 123.456f) sink[0] = acc.x + ring[lane];
}

static int g_lds_bytes = 65536;      // dynamic LDS per workgroup: 64 KB = one workgroup per CU at a time... 16 KB = up to eight

// the loader next to busy waves: wave 15 of a 1024-thread workgroup streams as above while waves 0..14 issue packed multiply-adds
// (mode 1) or multiply-adds and LDS reads in the walk's proportion -- four ds_read_b64 per 42 vector instructions (mode 2)
template <int MODE>
__global__ __launch_bounds__(1024) void probe_busy(const char *src, unsigned long long wg_stride, unsigned long long span, int n, int spin,
                                                   unsigned long long *cycles, float *sink, int n_loaders, int drain)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // 128 KB
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
    const unsigned int lds0 = (unsigned int)(uintptr_t)((lds_ptr)ring);
    const unsigned int lane = threadIdx.x & 63, wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= 16 - (unsigned int)n_loaders) {
        const unsigned long long base = ((unsigned long long)blockIdx.x * wg_stride + (unsigned long long)(wave & 3) * (span >> 2)) & (span - 1);
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; i++) {
            const unsigned long long off = (base + (unsigned long long)i * 1024ull) & (span - 1);
            const char *p = src + off;
            const unsigned int dst = lds0 + (unsigned int)((i & 31) + 32 * (wave & 3)) * 1024u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(16u * lane), "s"(p), "s"(dst) : "memory");
            wait_vm<32>();
            if (drain && (i + 1) % drain == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a phase of the staged walk: everything lands, then the next
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && wave == 15) cycles[blockIdx.x] = __builtin_readcyclecounter() - t0;
        return;
    }
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f a[8];
    for (int k = 0; k < 8; k++) a[k] = v2f{(float)lane, (float)k};
    const v2f m = {1.0001f, 0.9999f};
    const float2 *rd = reinterpret_cast<const float2 *>(ring) + wave * 512 + lane;
    float2 acc = make_float2(0.f, 0.f);
    for (int it = 0; it < spin; it++) {
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float2 v = rd[64 * q + 8192 * (it & 1)];
                acc.x += v.x;
                acc.y += v.y;
            }
        }
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(m));
    }
    float sum = acc.x + acc.y;
    for (int k = 0; k < 8; k++) sum += a[k].x + a[k].y;
    if (sink && sum == 123.456f) sink[0] = sum;
}

template <int MODE>
int run_busy(const char *src, unsigned long long wg_stride, unsigned long long span, const char *what, unsigned long long *d_cyc, int n, int spin, int n_loaders = 1, int drain = 0)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_busy<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL((probe_busy<MODE>), dim3(256), dim3(1024), 131072, 0, src, wg_stride, span, n, spin, d_cyc, (float *)nullptr, n_loaders, drain);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_busy<MODE>), dim3(256), dim3(1024), 131072, 0, src, wg_stride, span, n, spin, d_cyc, (float *)nullptr, n_loaders, drain);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(256);
    CHECK(hipMemcpy(c.data(), d_cyc, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : c) sum += (double)v;
    if (drain) printf("(all in flight land every %2d) ", drain);
    printf("lds-dma %-22s %d loader(s) next to waves %-28s: %8.1f counter ticks per KB and loader   kernel %7.3f ms (the loaders stream %.2f TB/s)\n", what,
           n_loaders, MODE == 0 ? "that exit at once" : MODE == 1 ? "of packed multiply-adds" : "of multiply-adds + LDS reads", sum / 256 / n, ms,
           256.0 * n_loaders * n * 1024.0 / (sum / 256 / 2.3e9) / 1e12);
    return 0;
}

template <int INFLIGHT, bool DMA>
int run(const char *src, unsigned long long wg_stride, unsigned long long span, const char *what, unsigned long long *d_cyc, int wgs, int n)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<INFLIGHT, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL((probe<INFLIGHT, DMA>), dim3(wgs), dim3(64), g_lds_bytes, 0, src, wg_stride, span, n, d_cyc, (float *)nullptr);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<INFLIGHT, DMA>), dim3(wgs), dim3(64), g_lds_bytes, 0, src, wg_stride, span, n, d_cyc, (float *)nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(wgs);
    CHECK(hipMemcpy(c.data(), d_cyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : c) sum += (double)v;
    const double per = sum / wgs / n;
    printf("%-7s %-22s in flight %2d: %8.1f counter ticks per KB (counter ~%.0f MHz)  %7.3f ms   %7.2f TB/s aggregate\n", DMA ? "lds-dma" : "vgpr",
           what, INFLIGHT, per, sum / wgs / (ms * 1e3), ms, (double)wgs * n * 1024.0 / ms / 1e9);
    return 0;
}

template <bool DMA>
int sweep(const char *src, unsigned long long wg_stride, unsigned long long span, const char *what, unsigned long long *d_cyc, int wgs, int n)
{
    if (run<1, DMA>(src, wg_stride, span, what, d_cyc, wgs, n)) return 1;
    if (run<4, DMA>(src, wg_stride, span, what, d_cyc, wgs, n)) return 1;
    if (run<8, DMA>(src, wg_stride, span, what, d_cyc, wgs, n)) return 1;
    if (run<16, DMA>(src, wg_stride, span, what, d_cyc, wgs, n)) return 1;
    if (run<32, DMA>(src, wg_stride, span, what, d_cyc, wgs, n)) return 1;
    if (run<60, DMA>(src, wg_stride, span, what, d_cyc, wgs, n)) return 1;
    return 0;
}

int main()
{
    const unsigned long long big = 4ull << 30;
    char *src;
    unsigned long long *d_cyc;
    const int wgs = 256, n = 8192;      // 8 MB per workgroup, 2 GB per launch
    CHECK(hipMalloc(&src, big));
    CHECK(hipMemset(src, 1, big));
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 4096));
    // (the counter of __builtin_readcyclecounter: s_memtime -- its rate is printed by timing a known launch below)
    if (sweep<true>(src, 16ull << 20, big, "own 8 MB (memory)", d_cyc, wgs, n)) return 1;
    if (sweep<true>(src, 0, 2ull << 20, "shared 2 MB (L2)", d_cyc, wgs, n)) return 1;
    if (sweep<false>(src, 16ull << 20, big, "own 8 MB (memory)", d_cyc, wgs, n)) return 1;
    if (sweep<false>(src, 0, 2ull << 20, "shared 2 MB (L2)", d_cyc, wgs, n)) return 1;
    for (int shared = 0; shared < 2; shared++) {
        const unsigned long long stride = shared ? 0 : 16ull << 20, span = shared ? 2ull << 20 : big;
        const char *what = shared ? "shared 2 MB (L2)" : "own 8 MB (memory)";
        if (run_busy<0>(src, stride, span, what, d_cyc, n, 0)) return 1;
        if (run_busy<1>(src, stride, span, what, d_cyc, n, 6000)) return 1;
        if (run_busy<2>(src, stride, span, what, d_cyc, n, 6000)) return 1;
        if (run_busy<0>(src, stride, span, what, d_cyc, n, 0, 2)) return 1;
        if (run_busy<0>(src, stride, span, what, d_cyc, n, 0, 4)) return 1;
        if (run_busy<2>(src, stride, span, what, d_cyc, n, 12000, 2)) return 1;
        for (int drain : {64, 32, 16})
            for (int nl : {1, 2})
                if (run_busy<2>(src, stride, span, what, d_cyc, n / nl, 12000, nl, drain / nl)) return 1;
    }
    // several waves per CU: 1024 workgroups of 64 KB (two per CU at a time), of 32 KB (four), of 16 KB (eight) -- ticks are per
    // wave, the aggregate rate tells what a CU takes
    for (int lds : {65536, 32768, 16384}) {
        g_lds_bytes = lds;
        printf("---- %d KB of LDS per workgroup, 2048 workgroups\n", lds / 1024);
        if (run<16, true>(src, 2ull << 20, big, "own 2 MB (memory)", d_cyc, 2048, n / 4)) return 1;
        if (run<16, true>(src, 0, 2ull << 20, "shared 2 MB (L2)", d_cyc, 2048, n / 4)) return 1;
        if (run<16, false>(src, 2ull << 20, big, "own 2 MB (memory)", d_cyc, 2048, n / 4)) return 1;
        if (run<16, false>(src, 0, 2ull << 20, "shared 2 MB (L2)", d_cyc, 2048, n / 4)) return 1;
    }
    return 0;
}
