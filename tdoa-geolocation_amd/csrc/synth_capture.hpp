// synth_capture.hpp -- synthetic .dat captures generated directly in HBM, modelled on the
// reference's simulator.go (pure tones + uniform noise, delay as a carrier phase only,
// [ref | target | ref] blocks, byte(x*127.5+127.5) quantisation; simulator.go:67-161).
// Used by bench.py so that 1.2 GB of input never crosses PCIe; the reference seeds
// math/rand from the clock, so there is no bit pattern to reproduce -- only the model.
#pragma once

#include "device_common.hpp"

namespace tdoa {

__device__ __forceinline__ unsigned long long synth_mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// counter-based uniform [0,1), 53 bits
__device__ __forceinline__ double synth_uniform(unsigned long long seed, unsigned long long counter)
{
    unsigned long long z = synth_mix64(seed + 0x9E3779B97F4A7C15ull * (counter + 1));
    z = synth_mix64(z ^ 0xD6E8FEB86659FD93ull);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ unsigned int synth_quantise(float v)
{
    float q = v * 127.5f + 127.5f;
    q = q < 0.0f ? 0.0f : q;
    q = q > 255.0f ? 255.0f : q;
    return (unsigned int)q;   // truncation toward zero, like Go's byte()
}

struct SynthBlock {
    double omega_over_fs;   // 2*pi*f / fs
    double phase;
    double amp;
    double noise;
    unsigned long long seed;
    unsigned long long block_id;
};

// one block of `n` samples; each thread writes one IQ pair
__global__ void k_synth_tone_block(uint8_t *out, long long n, SynthBlock b)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double arg = b.omega_over_fs * (double)i + b.phase;
    double s, c;
    sincos(arg, &s, &c);
    float re = (float)(b.amp * c), im = (float)(b.amp * s);
    unsigned long long ctr = (b.block_id << 40) + (unsigned long long)i;
    re += (float)(b.noise * (2.0 * synth_uniform(b.seed, 2 * ctr) - 1.0));
    im += (float)(b.noise * (2.0 * synth_uniform(b.seed, 2 * ctr + 1) - 1.0));
    reinterpret_cast<uint16_t *>(out)[i] = (uint16_t)(synth_quantise(re) | (synth_quantise(im) << 8));
}

}  // namespace tdoa
