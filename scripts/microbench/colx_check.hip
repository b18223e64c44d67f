// Stand-alone check of the forward column kernels against a direct f64 DFT of a few outputs.
// build + run on the GPU box:  hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/colx scripts/microbench/colx_check.hip && /tmp/colx
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <complex>
#include "../../tdoa-geolocation_amd/csrc/fft_radix16.hpp"
using namespace tdoa;

template <int F>
int check()
{
    FftPlan pl{};
    pl.N1 = 4096; pl.N2 = 256 * F; pl.logN1 = 12; pl.logN2 = 8 + (F == 2 ? 1 : 2); pl.C = 8; pl.logC = 3;
    pl.Nc = (long long)pl.N1 * pl.N2;
    const long long len = 2 * pl.Nc - 12345;
    std::vector<short> codes(2 * pl.Nc);
    unsigned s = 12345;
    for (auto &c : codes) { s = s * 1664525u + 1013904223u; c = (short)((s >> 16) % 2001 - 1000); }
    for (long long i = len; i < 2 * pl.Nc; i++) codes[i] = 0;
    short *d_codes; float2 *d_t; SWDesc *d_sw; FmStats *d_st;
    hipMalloc(&d_codes, codes.size() * 2); hipMalloc(&d_t, pl.Nc * 8); hipMalloc(&d_sw, sizeof(SWDesc)); hipMalloc(&d_st, sizeof(FmStats));
    hipMemcpy(d_codes, codes.data(), codes.size() * 2, hipMemcpyHostToDevice);
    SWDesc sw{nullptr, (int32_t)len, 0};
    FmStats st{0, 0, 0, 0.0f, 1.0f};
    hipMemcpy(d_sw, &sw, sizeof(sw), hipMemcpyHostToDevice);
    hipMemcpy(d_st, &st, sizeof(st), hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd_colx_c16<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k_fwd_colx_c16<F>, dim3(pl.N1 * F / 32, 1), dim3(512), 65536, 0, d_sw, d_codes, 2 * pl.Nc, d_st, d_t, pl);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 1; }
    std::vector<float2> t(pl.Nc);
    hipMemcpy(t.data(), d_t, pl.Nc * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    int bad = 0;
    const int cols[] = {0, 1, 7, 8, 9, 2047, 4095};
    for (int n1 : cols)
        for (int k2 = 0; k2 < pl.N2; k2++) {
            std::complex<double> acc = 0;
            for (int n2 = 0; n2 < pl.N2; n2++) {
                const long long m = (long long)n2 * pl.N1 + n1;
                std::complex<double> z(codes[2 * m], codes[2 * m + 1]);
                acc += z * std::polar(1.0, -2.0 * M_PI * (double)n2 * k2 / pl.N2);
            }
            acc *= std::polar(1.0, -2.0 * M_PI * (double)((long long)n1 * k2 % pl.Nc) / (double)pl.Nc);
            const float2 g = t[(size_t)k2 * pl.N1 + n1];
            const double e = std::abs(acc - std::complex<double>(g.x, g.y)) / (1000.0 * std::sqrt((double)pl.N2));
            if (e > 1e-4 && bad < 12) { std::printf("F=%d n1=%d k2=%d (j=%d k=%d q=%d) err %.3g\n", F, n1, k2, k2 & 15, (k2 >> 4) & 15, k2 >> 8, e); bad++; }
            worst = std::max(worst, e);
        }
    std::printf("F=%d worst relative error %.3g\n", F, worst);
    return worst > 1e-4;
}

int main() { int r = check<2>(); r |= check<4>(); return r; }
