/*
 * wav_dither.c -- CPU restatement of the reference's WAV sample conversion with TPDF dither.
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h).
 *
 * Follows wav.d:474-527 (WAVEncoder.writeSamples: ditherInput, then the per-format integer conversion) and
 * wav.d:674-701 (TPDFDither.process).  The reference draws from libc rand(); here the generator is a callback so
 * that a test can feed the product and this restatement the same sequence (NULL = libc rand() / RAND_MAX).
 */
#include "afg_oracle.h"

#include <math.h>
#include <stdlib.h>

static int libc_draw(void *user) { (void)user; return rand(); }

void afgo_tpdf_dither(double *inout, int frames, double scaleFactor, afgo_rand_fn rng, void *user, double rand_max)   /* wav.d:680-699 */
{
    if (!rng) { rng = libc_draw; rand_max = (double)RAND_MAX; }
    for (int n = 0; n < frames; ++n) {
        double x = inout[n];
        x *= scaleFactor;
        const double TUNE0 = 0.25;
        const double TUNE1 = TUNE0 * 0.5;
        x += (0.5 - 0.5 * (TUNE0 + TUNE1));
        x += TUNE0 * (rng(user) / rand_max);
        x += TUNE1 * (rng(user) / rand_max);
        x = floor(x);
        x /= scaleFactor;
        if (x < -1.0) x = -1.0;
        if (x > 1.0) x = 1.0;
        inout[n] = x;
    }
}

/* writeSamples for the integer formats (bits = 8, 16 or 24): returns the sample values the reference writes
 * (s8 as the byte value 0..255 reinterpreted through cast(byte), others as signed ints). */
int afgo_wav_pcm(const float *in, int samples, int bits, int enable_dither, afgo_rand_fn rng, void *user, double rand_max,
                 int32_t *out)
{
    double *buf = (double *)malloc(sizeof(double) * (size_t)(samples > 0 ? samples : 1));
    if (!buf) return -1;
    for (int n = 0; n < samples; ++n) buf[n] = in[n];                              /* ditherInput, wav.d:624-636 */
    const double scale = bits == 8 ? 127.0f : bits == 16 ? 32767.0f : 8388607.0f;
    if (enable_dither) afgo_tpdf_dither(buf, samples, scale, rng, user, rand_max);
    for (int n = 0; n < samples; ++n) {
        const double x = buf[n];
        if (bits == 8) {
            const int b = (int)(128.5 + x * 127.0);                                /* :486-487 */
            out[n] = (int32_t)(int8_t)b;
        } else if (bits == 16) {
            int s = (int)(32768.5 + x * 32767.0);                                  /* :501-502 */
            s -= 32768;
            out[n] = s;
        } else {
            int s = (int)(8388608.5 + x * 8388607.0);                              /* :517-518 */
            s -= 8388608;
            out[n] = s;
        }
    }
    free(buf);
    return 0;
}
