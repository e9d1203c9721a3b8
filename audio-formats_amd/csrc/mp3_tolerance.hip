// mp3_tolerance.hip -- the MP3 transform stage in AFG_NUMERIC_TOLERANCE (round 4): mp3_kernel.h compiled with fused
// multiply-adds (Makefile: -ffp-contract=fast) and the polyphase window (minimp3.d:1371-1405) accumulating in fma chains.
// The kernel was bound by vector issue (profiles/r03_pmc_mp3_transform_kernel.json: 1393 VALU instructions per granule
// pair, the 512-tap window 576 of them); this form issues 20 % fewer: C2 9.6 -> 8.6 ms at 1.3e-6 RMS from the oracle.
#define AFG_MP3_FMA 1
#define AFG_MP3_KERNEL mp3_tolerance_kernel
#include "mp3_kernel.h"

void afg::mp3_launch_tolerance(uint32_t n_segs, const void *d_segs, const void *d_streams, const float *d_coef,
                               const uint32_t *d_flags, float *d_pcm, float *d_state, hipStream_t stream)
{
    hipLaunchKernelGGL(mp3_tolerance_kernel, dim3(n_segs), dim3(64), 0, stream, (const Mp3Seg *)d_segs,
                       (const Mp3Stream *)d_streams, d_coef, d_flags, d_pcm, d_state);
}
