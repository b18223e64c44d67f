/* Plain C99 caller of the C ABI -- what cgo sees.  Correlates two raw IQ buffers (the second is the first
 * delayed by 7 samples) and prints the peak; exits 0 when the lag is 7.
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/pair_from_c.c -Ltdoa-geolocation_amd -ltdoa_mi355x \
 *       -Wl,-rpath,$PWD/tdoa-geolocation_amd -o pair_from_c */
#include <stdio.h>
#include <stdlib.h>

#include "tdoa_mi355x.h"

int main(void)
{
    enum { N = 20000, DELAY = 7 };
    uint8_t *a = malloc(2 * N), *b = malloc(2 * N);
    uint32_t s = 12345u, phase = 0;
    tdoa_params prm;
    tdoa_ctx *ctx = NULL;
    tdoa_peak peak;
    tdoa_fine_peak fine;
    int rc, i;
    if (!a || !b) return 2;
    for (i = 0; i < N + DELAY; i++) {            /* a random-walk phase: an FM-like signal */
        uint8_t ib, qb;
        s = s * 1664525u + 1013904223u;
        phase += (s >> 20) - 2048u + 300u;       /* carrier offset + noise-like modulation, units of 2 pi / 65536 */
        {
            /* 8-entry octagon approximation of (cos, sin): good enough for a demo signal */
            static const int c8[8] = {100, 71, 0, -71, -100, -71, 0, 71}, s8[8] = {0, 71, 100, 71, 0, -71, -100, -71};
            const unsigned o = (phase >> 13) & 7u;
            ib = (uint8_t)(128 + c8[o]);
            qb = (uint8_t)(128 + s8[o]);
        }
        if (i < N) { b[2 * i] = ib; b[2 * i + 1] = qb; }                 /* b sees the stream from sample 0 */
        if (i >= DELAY) { a[2 * (i - DELAY)] = ib; a[2 * (i - DELAY) + 1] = qb; }   /* a started DELAY samples later */
    }
    tdoa_default_params(&prm);
    prm.max_lag = 100;
    prm.window_len = N;
    rc = tdoa_create(&prm, &ctx);
    if (rc != TDOA_OK) {
        fprintf(stderr, "tdoa_create: %s\n", tdoa_strerror(rc));
        return rc == TDOA_ERR_NO_DEVICE ? 77 : 2;
    }
    rc = tdoa_fm_xcorr_fine_u8(ctx, a, N, b, N, prm.max_lag, 120.0, &peak, &fine);
    if (rc != TDOA_OK) {
        fprintf(stderr, "tdoa_fm_xcorr_fine_u8: %s (%s)\n", tdoa_strerror(rc), tdoa_last_error(ctx));
        tdoa_destroy(ctx);
        return 2;
    }
    printf("ABI %d: lag %d samples, corr %.3f, refined delay %.3f, plausible %d\n", tdoa_abi_version(), (int)peak.lag,
           peak.corr, fine.delay, (int)fine.plausible);
    tdoa_destroy(ctx);
    free(a);
    free(b);
    return peak.lag == DELAY ? 0 : 1;
}
