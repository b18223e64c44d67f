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

// ---- weak_signal_simulator.go (BASELINE config 3: weak reference, strong target) ----------------------------
// standard normal: Box-Muller on two counter-based uniforms (the reference calls rand.NormFloat64, time-seeded:
// only the distribution is reproducible); same counters as the CPU restatement in oracle/tdoa_oracle.c
__device__ __forceinline__ double synth_normal(unsigned long long seed, unsigned long long counter)
{
    double u1 = synth_uniform(seed, 2 * counter);
    const double u2 = synth_uniform(seed, 2 * counter + 1);
    u1 = u1 < 1e-300 ? 1e-300 : u1;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

struct SynthWeakBlock {
    double omega_over_fs;   // 2*pi*f / fs
    double phase;
    double amp;
    double gaussian;        // sigma of the Gaussian noise (weak block: 0.8 amp; strong block: 0.001 absolute)
    double impulse_p;       // weak block only: probability, level, phase drift per sample, DC offset
    double impulse_level;
    double drift_per_sample;
    double dc;
    unsigned long long seed;
    unsigned long long block_id;
    int weak;               // 1: generateWeakSignal (weak_signal_simulator.go:89-126), 0: generateStrongSignal (:129-148)
};

// one block of `n` samples; each thread writes one IQ pair.  Phase drift in closed form ((i + 1) drift / fs instead of
// the reference's running sum) so that samples are independent, as in the CPU restatement.
__global__ void k_synth_weak_block(uint8_t *out, long long n, SynthWeakBlock b)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double arg = b.omega_over_fs * (double)i + b.phase;
    if (b.weak) arg += (double)(i + 1) * b.drift_per_sample;
    double s, c;
    sincos(arg, &s, &c);
    double re = b.amp * c, im = b.amp * s;
    const unsigned long long ctr = ((b.block_id << 40) + (unsigned long long)i) * 8;
    if (b.weak) {
        re += b.dc;
        im += b.dc;
        if (b.gaussian > 0) {
            re += b.gaussian * synth_normal(b.seed, ctr);
            im += b.gaussian * synth_normal(b.seed, ctr + 1);
        }
        if (synth_uniform(b.seed, 2 * (ctr + 2)) < b.impulse_p) {
            re += b.impulse_level * (2.0 * synth_uniform(b.seed, 2 * (ctr + 3)) - 1.0);
            im += b.impulse_level * (2.0 * synth_uniform(b.seed, 2 * (ctr + 4)) - 1.0);
        }
    } else {
        re += b.gaussian * synth_normal(b.seed, ctr);
        im += b.gaussian * synth_normal(b.seed, ctr + 1);
    }
    reinterpret_cast<uint16_t *>(out)[i] = (uint16_t)(synth_quantise((float)re) | (synth_quantise((float)im) << 8));
}

}  // namespace tdoa
