/*
 * afg_oracle.h -- CPU restatement of the audio-formats transform stage.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it, and there only as the checker / reported baseline.
 *
 * PARITY UNPINNED by the reference's own tests: the reference
 * (AuburnSounds/audio-formats, D) ships no golden vectors, known-answer
 * tests or fixtures for any decoder, and no D compiler exists in this image
 * or on the GPU box, so the reference cannot be run (oracle/_ref is
 * unbuildable).  What pins this restatement instead is listed in DESIGN.md 7 (in full: HISTORY.md 4):
 * float64 textbook definitions of every transform (tests/test_oracle_*.py),
 * lossless FLAC/QOA encode->decode round trips, frozen golden vectors under
 * tests/golden/, and -- the one check against code written elsewhere -- the
 * whole-file decodes of the fixture files by Chromium's FFmpeg / libopus
 * (tests/test_oracle_independent.py: MP3 8.9e-6 RMS, Vorbis 3e-8 RMS, FLAC
 * exact, Opus 0.5 % / 6 % RMS).  That is not the reference either: the
 * restatement stays unpinned in the sense above.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic -- operation order, operand widths, constants -- it follows.
 * Build with -ffp-contract=off: the D reference has no fused multiply-adds.
 */
#ifndef AFG_ORACLE_H
#define AFG_ORACLE_H

#include <stddef.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ MP3 -- */

/* Per gr-ch flag word (same packing as include/afg.h AFG_MP3_FLAGS):
 *   bits 0..1  block_type (0 normal, 1 start, 2 short, 3 stop)
 *   bits 8..15 n_long_bands (0, 2 or 4)       minimp3.d:1218
 *   bits 16..23 aa_bands + 1 (0..32)           minimp3.d:1217,1222
 */
#define AFGO_MP3_FLAGS(block_type, n_long_bands, aa_bands) \
    ((uint32_t)(block_type) | ((uint32_t)(n_long_bands) << 8) | ((uint32_t)((aa_bands) + 1) << 16))

/* Persistent per-stream transform state, minimp3.d:38-46 (mdct_overlap, qmf_state). */
typedef struct afgo_mp3_state {
    float mdct_overlap[2][9 * 32];
    float qmf_state[15 * 2 * 32];
} afgo_mp3_state;

/* Single pieces, exposed so tests can pin each against its float64 definition. */
void afgo_mp3_antialias(float *grbuf, int nbands);                       /* minimp3.d:1002-1020 */
void afgo_mp3_imdct_gr(float *grbuf, float *overlap, unsigned block_type,
                       unsigned n_long_bands);                           /* minimp3.d:1152-1168 */
void afgo_mp3_change_sign(float *grbuf);                                 /* minimp3.d:1144-1150 */
void afgo_mp3_dct2(float *grbuf, int n);                                 /* minimp3.d:1232-1298 */
void afgo_mp3_synth_granule(float *qmf_state, float *grbuf, int nbands, int nch,
                            float *pcm, float *lins);                    /* minimp3.d:1408-1434 */

/* One granule of one stream: the transform tail of L3_decode (minimp3.d:1215-1229)
 * for every channel followed by mp3d_synth_granule (minimp3.d:1553).
 * coef: nch*576 floats ([ch][576], post-dequant/stereo/reorder), modified in place.
 * flags: nch words.  pcm: 576*nch floats, interleaved. */
void afgo_mp3_granule(afgo_mp3_state *st, float *coef, const uint32_t *flags, int nch, float *pcm);

/* Whole batch.  Stream s owns blocks [blk_base, blk_base + ngr[s]*nch[s]) of 576
 * floats each in `coef` (order [granule][ch]) where blk_base is the running sum
 * over earlier streams; `pcm` has the same float count, 576*nch per granule.
 * Each stream starts from zero state (minimp3.d:1509).  coef is NOT modified.
 * If states != NULL it receives the final state of every stream. */
void afgo_mp3_transform(uint32_t n_streams, const uint32_t *ngr, const uint8_t *nch,
                        const float *coef, const uint32_t *flags, float *pcm,
                        afgo_mp3_state *states);

/* MP3 front-end + file drive (mp3_frontend.c): everything minimp3 / minimp3_ex do ahead of the transform
 * seam, for whole files in memory.  The transform-stage records of every decoded granule are kept so
 * that the product's parser can be compared record by record. */
typedef struct afgo_mp3_file {
    int channels, hz;
    int vbr_tag_found;
    int start_delay;                /* samples (channels included) skipped at the start, minimp3_ex.d:594 */
    uint64_t detected_samples;      /* 0 = read to the end of the data, minimp3_ex.d:600 */
    uint64_t samples;               /* mp3dec_ex_t.samples: what AudioStream divides by the channel count */
    uint32_t n_streams;             /* runs of granules with continuous decoder state (a resync resets it) */
    uint32_t *stream_granules;      /* [n_streams] */
    uint64_t n_blocks;              /* gr-ch blocks in coef / flags, order [granule][channel] */
    float *coef;
    uint32_t *flags;
    uint64_t pcm_samples;           /* what mp3dec_ex_read delivers until it returns 0 */
    float *pcm;
    int layer;                      /* 3, or 1 / 2: the blocks are then 12-slot synthesis granules flagged 0x40000000 (subband
                                       samples, index band * 18 + slot, slots 12..17 unused), minimp3.d:1557-1578 */
} afgo_mp3_file;

int afgo_mp3_decode_file(const uint8_t *data, size_t size, afgo_mp3_file *out);   /* 0 ok, -1 no MPEG audio stream, -2 memory */
void afgo_mp3_file_free(afgo_mp3_file *f);

/* --------------------------------------------------------------- Vorbis -- */

/* Tables of one blocksize, stb_vorbis2.d:851-898. */
typedef struct afgo_vorbis_tables {
    int n;
    float *A;            /* n/2 */
    float *B;            /* n/2 */
    float *C;            /* n/4 */
    float *window;       /* n/2 */
    uint16_t *bitrev;    /* n/8 */
} afgo_vorbis_tables;

int  afgo_vorbis_tables_init(afgo_vorbis_tables *t, int n);  /* stb_vorbis2.d:883-898 */
/* Which legal D evaluation of the table expressions to use (0 = default, see vorbis_transform.c); tests only. */
void afgo_vorbis_set_table_mode(int mode);
void afgo_vorbis_tables_free(afgo_vorbis_tables *t);

/* In-place n/2 spectrum -> n time samples, stb_vorbis2.d:1941-2242.
 * buffer holds n floats, scratch n/2 floats. */
void afgo_vorbis_inverse_mdct(float *buffer, int n, const afgo_vorbis_tables *t, float *scratch);

/* Packet flag byte (same packing as include/afg.h):
 *   bit 0 blockflag (long block), bit 1 prev-window flag, bit 2 next-window flag
 *   (stb_vorbis2.d:2324-2331). */
#define AFGO_VORBIS_LONG 1u
#define AFGO_VORBIS_PREV 2u
#define AFGO_VORBIS_NEXT 4u

/* Window bounds of one packet, stb_vorbis2.d:2333-2349. */
void afgo_vorbis_window_bounds(int blocksize0, int blocksize1, unsigned pflags,
                               int *n, int *left_start, int *left_end,
                               int *right_start, int *right_end);

/* Whole batch: per stream, inverse_mdct per channel (stb_vorbis2.d:2526-2527)
 * then vorbis_finish_frame (stb_vorbis2.d:2606-2657) for each packet in order,
 * starting with previous_length = 0, and the interleave of
 * stb_vorbis_get_samples_float_interleaved (stb_vorbis2.d:3927-3952).
 *
 * Streams are concatenated: stream s has npkt[s] packets and nch[s] channels.
 * Packet p of a stream reads nch*(n/2) floats from `spec` at spec_off[p]
 * ([ch][n/2]) and writes (right_start-left_start)*nch interleaved floats at
 * out_off[p] (in floats); the first packet of a stream writes nothing.
 * spec_off/out_off are absolute (batch-wide) and have total_packets entries.
 * Returns 0, or -1 on a window mismatch (get_window() == NULL, :2621). */
int afgo_vorbis_transform(uint32_t n_streams, const uint32_t *npkt, const uint8_t *nch,
                          const uint16_t *blocksize0, const uint16_t *blocksize1,
                          const uint8_t *pflags, const uint64_t *spec_off,
                          const uint64_t *out_off, const float *spec, float *out);

/* Output frames (per channel) of each packet and the running total; helper that
 * mirrors the return value of vorbis_finish_frame for a flag sequence. */
uint64_t afgo_vorbis_layout(uint32_t npkt, int nch, int blocksize0, int blocksize1,
                            const uint8_t *pflags, uint64_t spec_base, uint64_t out_base,
                            uint64_t *spec_off, uint64_t *out_off, uint64_t *spec_total);

/* Ogg Vorbis front-end (vorbis_frontend.c): Ogg pages, the three header packets, and every audio packet up to
 * the transform seam, for a whole file in memory; `take_*` is what the pull API delivers of each packet's
 * (right_start - left_start) output frames. */
typedef struct afgo_vorbis_file {
    int channels;
    unsigned sample_rate;
    int blocksize0, blocksize1;
    uint32_t total_samples;          /* stb_vorbis_stream_length_in_samples (0 = unknown) */
    uint32_t n_packets;              /* decoded audio packets, the discarded first one included */
    uint8_t *pflags;                 /* AFG_VORBIS_LONG | PREV | NEXT */
    uint64_t spec_floats;
    float *spec;                     /* per packet [channel][n/2], after floor multiply and inverse coupling */
    int32_t *take_from, *take_count; /* delivered frames of packet p: [take_from, take_from + take_count) */
    uint64_t pcm_frames;             /* sum of take_count */
} afgo_vorbis_file;

int afgo_vorbis_decode_file(const uint8_t *data, size_t size, afgo_vorbis_file *out);   /* 0 ok, -1 not Ogg Vorbis */
int afgo_vorbis_decode_file_ex(const uint8_t *data, size_t size, afgo_vorbis_file *out, int seek_clears_eof);
void afgo_vorbis_file_free(afgo_vorbis_file *f);

/* The tail of the packet decode on records (same layouts as include/afg.h afg_vorbis_floor_packet / _curve): inverse
 * coupling (stb_vorbis2.d:2493-2514), silent channels and do_floor (:2516-2523, :2255-2284) with draw_line (:1534-1563),
 * in place on residue vectors.  Checker of afg_vorbis_floor_hip. */
typedef struct afgo_vorbis_floor_packet {
    uint64_t spec_off;
    uint32_t n2, channels, curve_index, step_off, n_steps, pad;
} afgo_vorbis_floor_packet;
typedef struct afgo_vorbis_floor_curve { uint32_t point_off, n_points; } afgo_vorbis_floor_curve;
void afgo_vorbis_floor(uint64_t n_packets, const afgo_vorbis_floor_packet *packets, const afgo_vorbis_floor_curve *curves,
                       const int32_t *points, const uint8_t *steps, float *spec);

/* ----------------------------------------------------------------- FLAC -- */

/* drflac.d:1060-1140 */
int32_t afgo_flac_prediction_32(unsigned order, int shift, const int16_t *coef, const int32_t *p);
int32_t afgo_flac_prediction_64(unsigned order, int shift, const int16_t *coef, const int32_t *p);

/* Subframe record (same layout as include/afg.h afg_flac_subframe). */
typedef struct afgo_flac_subframe {
    int16_t coef[32];
    uint8_t order;      /* 0..32; warm-up samples sit in res[0..order) */
    uint8_t shift;      /* lpcShift, 0..31 (negative shifts are rejected by the host) */
    uint8_t wasted;     /* wastedBitsPerSample, drflac.d:1561-1566 */
    uint8_t use64;      /* subframe bitsPerSample > 16, drflac.d:1308 */
} afgo_flac_subframe;

/* Frame record (same layout as include/afg.h afg_flac_frame). */
typedef struct afgo_flac_frame {
    uint64_t in_off;      /* int32 index of channel 0's residual plane; channel c at in_off + c*block_size */
    uint64_t out_off;     /* int32 index of the interleaved output */
    uint32_t block_size;
    uint32_t sf_index;    /* index of channel 0's subframe record; channel c at sf_index + c */
    uint8_t  channels;    /* 1..8 */
    uint8_t  assignment;  /* 0 independent, 8 left/side, 9 right/side, 10 mid/side (drflac.d channelAssignment) */
    uint8_t  bps;         /* STREAMINFO bitsPerSample (pFlac.bitsPerSample) */
    uint8_t  res16;       /* 1: the frame's residual rows are int16 (in_off an int16 index, rows padded to 8: include/afg.h) */
    uint8_t  pad[4];
} afgo_flac_frame;

#define AFGO_FLAC_INDEPENDENT 0
#define AFGO_FLAC_LEFT_SIDE   8
#define AFGO_FLAC_RIGHT_SIDE  9
#define AFGO_FLAC_MID_SIDE   10

/* LPC restore of one subframe in place (drflac.d:1235, :1264-1269). */
void afgo_flac_restore_subframe(const afgo_flac_subframe *sf, int32_t *samples, uint32_t block_size);

/* Whole batch: restore every subframe, then the decorrelate + shift + interleave
 * of drflac_read_s32 (drflac.d:2885-2941).  subframes are indexed
 * frames[f].sf_index + channel.
 * out_i32 receives drflac_read_s32-identical samples; if out_f32 != NULL it
 * also receives stream.d:505-511's float conversion.  res is NOT modified. */
void afgo_flac_transform(uint64_t n_frames, const afgo_flac_frame *frames,
                         const afgo_flac_subframe *subframes, const int32_t *res,
                         int32_t *out_i32, float *out_f32);

/* The FLAC front-end (oracle/flac_frontend.c): native FLAC bytes -> what drflac_read_s32 delivers, read to the end of the
 * stream the way AudioStream reads it (stream.d:492-515), with the reference's 32-bit cache bit reader
 * (drflac.d:680-1043), frame / subframe parse (:1444-1695) and fused Rice + prediction loop (:1143-1328). */
#define AFGO_FLAC_F_IGNORED_FAILURE 1u   /* a subframe's sample decode failed and was delivered as the buffer stood (drflac.d:1591-1594) */
#define AFGO_FLAC_F_UNINITIALISED   2u   /* ... and part of what was delivered had never been written: malloc'ed memory in the reference */
#define AFGO_FLAC_F_UNDEFINED       4u   /* the next frame needs an operation D does not define (or overruns the decode buffer): stream ended there */
typedef struct afgo_flac_file {
    uint32_t channels, sample_rate, bps, max_block;    /* STREAMINFO */
    uint64_t total_samples;                            /* STREAMINFO's count times channels (pFlac.totalSampleCount) */
    uint64_t n_samples;                                /* interleaved int32 values delivered */
    int32_t *pcm;
    uint32_t n_frames;
    uint32_t flags;                                    /* AFGO_FLAC_F_* */
    uint64_t first_flag_sample;                        /* values delivered before the first flagged frame (UINT64_MAX: none) */
} afgo_flac_file;
int  afgo_flac_decode_file(const uint8_t *data, size_t size, afgo_flac_file *out);   /* 0 ok, -1 not native FLAC, -2 memory */
void afgo_flac_file_free(afgo_flac_file *file);

/* The QOA stream layer (oracle/qoa_lms.c): qoa_decode_header (qoa.d:413-453) + the frame loop of QOADecoder.readSamples
 * (qoa.d:803-851) to the end of the stream: interleaved int16, and the float the reader delivers (* 1.0f / 32767). */
typedef struct afgo_qoa_file {
    uint32_t channels, samplerate, samples;            /* file header / first frame header */
    uint64_t n_frames_pcm;                             /* sample frames delivered */
    uint32_t n_qoa_frames;
    int16_t *pcm;
} afgo_qoa_file;
int  afgo_qoa_decode_file(const uint8_t *data, size_t size, afgo_qoa_file *out);      /* 0 ok, -1 not QOA, -2 memory */
void afgo_qoa_file_free(afgo_qoa_file *file);

/* ------------------------------------------------------------------ QOA -- */

/* Frame record (same layout as include/afg.h afg_qoa_frame). */
typedef struct afgo_qoa_frame {
    uint64_t byte_off;    /* offset of the 8-byte frame header in the byte plane */
    uint64_t out_off;     /* index of the frame's first output value (frames * channels so far) */
    uint16_t samples;     /* samples per channel in this frame (<= 5120), qoa.d:476 */
    uint8_t  channels;    /* 1..8 */
    uint8_t  pad[5];
} afgo_qoa_frame;

uint32_t afgo_qoa_decode_frame(const uint8_t *frame, size_t avail, int expect_channels,
                               int16_t *sample_data);                        /* qoa.d:455-534 */
void afgo_qoa_transform(uint64_t n_frames, const afgo_qoa_frame *frames, const uint8_t *bytes,
                        int16_t *out_i16, float *out_f32);
size_t afgo_qoa_encode(const int16_t *pcm, uint32_t samples, int channels, uint32_t samplerate,
                       uint8_t *out, int16_t *recon);                        /* qoa.d:295-399 */

/* ----------------------------------------------------------------- CELT -- */

/* Per-channel carry state (the transform-stage part of CeltFrame, dopus.d:1645-1663). */
typedef struct afgo_celt_state {
    float   buf[2048];
    int32_t pf_period;
    float   pf_gains[3];
    int32_t pf_period_old;
    float   pf_gains_old[3];
    float   deemph_coeff;
    int32_t pad[7];
} afgo_celt_state;          /* 8256 bytes; same layout as include/afg.h afg_celt_state */

/* One CELT frame of one output channel (same layout as include/afg.h afg_celt_frame). */
typedef struct afgo_celt_frame {
    uint64_t coef_off;       /* float index of coeffs[ch][0] (frame_size floats, blocks interleaved, dopus.d:3688) */
    uint64_t out_off;        /* float index of output sample 0 */
    uint32_t out_stride;     /* distance between consecutive output samples (1 = planar) */
    uint16_t frame_size;     /* 120, 240, 480 or 960 */
    uint8_t  blocks;         /* 1, or 1 << duration when transient (dopus.d:3630) */
    uint8_t  pad;
    int32_t  pf_period_new;  /* dopus.d:3407 */
    float    pf_gains_new[3];
    float    imdct_scale;    /* 1.0, or 0.5 for the stereo->mono downmix (dopus.d:3665) */
    uint32_t pad2;
} afgo_celt_frame;           /* 48 bytes */

/* 0 = tables in x87 real (default), 1 = in double (targets where real == double); tests only. */
void afgo_celt_set_table_mode(int mode);
void afgo_celt_imdct_half(int N, float *dst, const float *src, int stride, float scale);   /* dopus.d:1611-1637 */
void afgo_celt_frame_channel(afgo_celt_state *f, const afgo_celt_frame *fr, const float *coeffs,
                             float *out, int out_stride);                                   /* dopus.d:3680-3702 */
void afgo_celt_transform(uint32_t n_chan, const uint64_t *rec_base, const afgo_celt_frame *recs,
                         const float *coeffs, float *out, afgo_celt_state *states);
void afgo_opus_output(uint64_t n, const float *in, int16_t *out_i16, float *out_f32);   /* dopus.d:7923-7926, stream.d:480 */

/* Ogg Opus front-end, CELT-only (opus_frontend.c).  frames[i] is channel 0's transform record of frame i (out_off /
 * out_stride address interleaved PCM); channel c of the same frame: coef_off + c * frame_size, out_off + c. */
typedef struct afgo_opus_file {
    int32_t  channels, preskip, gain_i, error;
    float    gain;                   /* what opus_decode_packet multiplies the floats by when gain_i != 0 (dopus.d:6690) */
    int32_t  pad;
    int64_t  declared_frames;        /* last page's granule position - preskip (dopus.d:8159): AudioStream's length */
    uint64_t pcm_frames;             /* frames the packets decode to */
    uint64_t n_frames;
    afgo_celt_frame *frames;
    uint64_t n_coeffs;
    float   *coeffs;
} afgo_opus_file;
int  afgo_opus_decode_file(const uint8_t *data, size_t size, afgo_opus_file *out);
void afgo_opus_file_free(afgo_opus_file *f);

/* ------------------------------------------------------------- WAV out -- */
typedef int (*afgo_rand_fn)(void *user);
void afgo_tpdf_dither(double *inout, int frames, double scaleFactor, afgo_rand_fn rng, void *user, double rand_max);  /* wav.d:674-701 */
int  afgo_wav_pcm(const float *in, int samples, int bits, int enable_dither, afgo_rand_fn rng, void *user, double rand_max,
                  int32_t *out);                                                                                     /* wav.d:474-527 */

#ifdef __cplusplus
}
#endif
#endif
