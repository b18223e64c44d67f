// window_quality.hpp -- capture-quality statistics of raw IQ windows.
//
// Replaces (reference file:line): the byte statistics loop of fastAnalyzeSamples
// (fast_analyzer.go:117-136: sums of I, Q, I^2, Q^2 and the extremes) and the block power check of
// validateDataFile (collector.go:219-224: mean of (I-127.5)^2 + (Q-127.5)^2), for every
// (station, window) in one streaming pass.  The reference adds float64(byte) terms in sample
// order; every partial sum is an integer below 2^53, so exact integer accumulation in any
// order gives the same float64 bits.  The few f64 operations after the sums (fast_analyzer.go:139-155)
// are done on the host, in the reference's expression order.
#pragma once

#include "device_common.hpp"
#include "k1_discriminator.hpp"

namespace tdoa {

struct QualAcc {
    unsigned long long si, sq, sii, sqq;     // sums of b_I, b_Q, b_I^2, b_Q^2
    unsigned int imin, imax, qmin, qmax;
};

__global__ void k_quality_init(QualAcc *acc, int n)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) acc[id] = QualAcc{0, 0, 0, 0, 255u, 0u, 255u, 0u};
}

constexpr int kQualChunk = 16384;   // samples per workgroup: 256 threads x 8 trips x 8 samples

// grid (ceil(max_len / kQualChunk), n_sw), 256 threads
__global__ __launch_bounds__(256) void k_window_quality(const SWDesc *sw, QualAcc *acc)
{
    const SWDesc d = sw[blockIdx.y];
    const int len = d.len;
    const int start = blockIdx.x * kQualChunk;
    if (start >= len) return;
    const gptr16 p = k1_global(d.base);
    unsigned int si = 0, sq = 0, sii = 0, sqq = 0, imin = 255u, imax = 0u, qmin = 255u, qmax = 0u;   // <= 64 samples: no overflow
    for (int trip = 0; trip < kQualChunk / 2048; trip++) {
        const int i0 = start + trip * 2048 + threadIdx.x * 8;
        if (i0 >= len) break;
        unsigned int s[9];
        if (i0 + 8 <= len) {
            k1_load8(p, i0, s);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k + 1] = i0 + k < len ? (unsigned int)p[i0 + k] : 0xffffffffu;
        }
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (s[k] != 0xffffffffu) {
                const unsigned int bi = s[k] & 0xffu, bq = (s[k] >> 8) & 0xffu;
                si += bi; sq += bq; sii += bi * bi; sqq += bq * bq;
                imin = min(imin, bi); imax = max(imax, bi); qmin = min(qmin, bq); qmax = max(qmax, bq);
            }
        }
    }
    unsigned long long a = si, b = sq, c = sii, e = sqq;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, kWave); b += __shfl_xor(b, off, kWave);
        c += __shfl_xor(c, off, kWave); e += __shfl_xor(e, off, kWave);
        imin = min(imin, (unsigned int)__shfl_xor((int)imin, off, kWave));
        imax = max(imax, (unsigned int)__shfl_xor((int)imax, off, kWave));
        qmin = min(qmin, (unsigned int)__shfl_xor((int)qmin, off, kWave));
        qmax = max(qmax, (unsigned int)__shfl_xor((int)qmax, off, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        QualAcc *o = acc + blockIdx.y;
        atomicAdd(&o->si, a); atomicAdd(&o->sq, b); atomicAdd(&o->sii, c); atomicAdd(&o->sqq, e);
        atomicMin(&o->imin, imin); atomicMax(&o->imax, imax); atomicMin(&o->qmin, qmin); atomicMax(&o->qmax, qmax);
    }
}

}  // namespace tdoa
