/*
 * oracle/qoa_lms.c -- CPU restatement of the QOA frame decoder (and of the encoder, which is the
 * fixture generator: the reference has no QOA sample files).  TEST INFRASTRUCTURE ONLY (see
 * afg_oracle.h).  PARITY UNPINNED by reference vectors; pinned by tests/test_oracle_qoa.py
 * (encode -> decode returns exactly the encoder's own reconstruction).
 *
 * Follows source/audioformats/qoa.d of the reference:
 *   tables                :150-215      lms predict / update   :231-254
 *   qoa_div, clamps       :263-286      encode_frame           :295-399
 *   decode_frame          :455-534      float conversion       :831-838
 *   encoder start weights :578-582      file header            :413-453
 */
#include "afg_oracle.h"
#include <string.h>

#define QOA_SLICE_LEN 20
#define QOA_SLICES_PER_FRAME 256
#define QOA_FRAME_LEN (QOA_SLICES_PER_FRAME * QOA_SLICE_LEN)
#define QOA_LMS_LEN 4
#define QOA_MAGIC 0x716f6166u

typedef struct { int history[QOA_LMS_LEN]; int weights[QOA_LMS_LEN]; } lms_t;

static const int k_quant_tab[17] = { 7, 7, 7, 5, 5, 3, 3, 1, 0, 0, 2, 2, 4, 4, 6, 6, 6 };
static const int k_reciprocal_tab[16] = { 65536, 9363, 3121, 1457, 781, 475, 311, 216, 156, 117, 90, 71, 57, 47, 39, 32 };
static const int k_dequant_tab[16][8] = {
    {    1,    -1,    3,    -3,    5,    -5,     7,     -7 }, {    5,    -5,   18,   -18,   32,   -32,    49,    -49 },
    {   16,   -16,   53,   -53,   95,   -95,   147,   -147 }, {   34,   -34,  113,  -113,  203,  -203,   315,   -315 },
    {   63,   -63,  210,  -210,  378,  -378,   588,   -588 }, {  104,  -104,  345,  -345,  621,  -621,   966,   -966 },
    {  158,  -158,  528,  -528,  950,  -950,  1477,  -1477 }, {  228,  -228,  760,  -760, 1368, -1368,  2128,  -2128 },
    {  316,  -316, 1053, -1053, 1895, -1895,  2947,  -2947 }, {  422,  -422, 1405, -1405, 2529, -2529,  3934,  -3934 },
    {  548,  -548, 1828, -1828, 3290, -3290,  5117,  -5117 }, {  696,  -696, 2320, -2320, 4176, -4176,  6496,  -6496 },
    {  868,  -868, 2893, -2893, 5207, -5207,  8099,  -8099 }, { 1064, -1064, 3548, -3548, 6386, -6386,  9933,  -9933 },
    { 1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005 }, { 1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336 } };

static int lms_predict(const lms_t *l)                        /* :231-239, int arithmetic wraps */
{
    unsigned p = 0;
    for (int i = 0; i < QOA_LMS_LEN; i++) p += (unsigned)l->weights[i] * (unsigned)l->history[i];
    return (int)p >> 13;
}

static void lms_update(lms_t *l, int sample, int residual)     /* :241-254 */
{
    int delta = residual >> 4;
    for (int i = 0; i < QOA_LMS_LEN; i++)
        l->weights[i] = (int)((unsigned)l->weights[i] + (unsigned)(l->history[i] < 0 ? -delta : delta));
    for (int i = 0; i < QOA_LMS_LEN - 1; i++) l->history[i] = l->history[i + 1];
    l->history[QOA_LMS_LEN - 1] = sample;
}

static int clamp_s16(int v)                                    /* :278-286 */
{
    if (v < -32768) return -32768;
    if (v > 32767) return 32767;
    return v;
}

static int qoa_div(int v, int scalefactor)                     /* :263-269 */
{
    int reciprocal = k_reciprocal_tab[scalefactor];
    int n = (int)(((unsigned)v * (unsigned)reciprocal + (1u << 15))) >> 16;
    n = n + ((v > 0) - (v < 0)) - ((n > 0) - (n < 0));
    return n;
}

static uint64_t rd64be(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}

static void wr64be(uint8_t *p, uint64_t v)
{
    for (int i = 7; i >= 0; i--) { p[i] = (uint8_t)v; v >>= 8; }
}

/* Decode one frame (:455-534).  `frame` points at the 8-byte frame header.  Returns the number of
 * samples per channel written to sample_data (interleaved), 0 if the header is inconsistent. */
uint32_t afgo_qoa_decode_frame(const uint8_t *frame, size_t avail, int expect_channels, int16_t *sample_data)
{
    if (avail < 8) return 0;
    uint64_t fh = rd64be(frame);
    int channels = (int)((fh >> 56) & 0xff);
    int samples = (int)((fh >> 16) & 0xffff);
    int frame_size = (int)(fh & 0xffff);
    int data_size = frame_size - 8 - QOA_LMS_LEN * 4 * channels;
    int num_slices = data_size / 8;
    if ((size_t)frame_size > avail || channels != expect_channels || samples * channels > num_slices * QOA_SLICE_LEN)
        return 0;
    lms_t lms[8];
    const uint8_t *p = frame + 8;
    for (int c = 0; c < channels; c++) {
        uint64_t history = rd64be(p); p += 8;
        uint64_t weights = rd64be(p); p += 8;
        for (int i = 0; i < QOA_LMS_LEN; i++) {
            lms[c].history[i] = (int16_t)(history >> 48);
            history <<= 16;
            lms[c].weights[i] = (int16_t)(weights >> 48);
            weights <<= 16;
        }
    }
    for (int sample_index = 0; sample_index < samples; sample_index += QOA_SLICE_LEN) {
        for (int c = 0; c < channels; c++) {
            uint64_t slice = rd64be(p); p += 8;
            int scalefactor = (int)((slice >> 60) & 0xf);
            int slice_start = sample_index * channels + c;
            int end = sample_index + QOA_SLICE_LEN; if (end > samples) end = samples;
            int slice_end = end * channels + c;
            for (int si = slice_start; si < slice_end; si += channels) {
                int predicted = lms_predict(&lms[c]);
                int quantized = (int)((slice >> 57) & 0x7);
                int dequantized = k_dequant_tab[scalefactor][quantized];
                int reconstructed = clamp_s16(predicted + dequantized);
                sample_data[si] = (int16_t)reconstructed;
                slice <<= 3;
                lms_update(&lms[c], reconstructed, dequantized);
            }
        }
    }
    return (uint32_t)samples;
}

/* Batch: frame f starts at bytes + frames[f].byte_off and writes samples*channels values at
 * out_off.  out_f32 = value * (1.0f / 32767) (:831-838); either output may be NULL. */
void afgo_qoa_transform(uint64_t n_frames, const afgo_qoa_frame *frames, const uint8_t *bytes,
                        int16_t *out_i16, float *out_f32)
{
    const float F = 1.0f / 32767;                              /* enum float F = 1.0f / short.max */
    int16_t buf[QOA_FRAME_LEN * 8];
    for (uint64_t f = 0; f < n_frames; f++) {
        const afgo_qoa_frame *fr = &frames[f];
        memset(buf, 0, sizeof(buf));
        uint32_t n = afgo_qoa_decode_frame(bytes + fr->byte_off, (size_t)1 << 20, fr->channels, buf);
        size_t cnt = (size_t)n * fr->channels;
        if (out_i16) memcpy(out_i16 + fr->out_off, buf, cnt * sizeof(int16_t));
        if (out_f32)
            for (size_t i = 0; i < cnt; i++) out_f32[fr->out_off + i] = buf[i] * F;
    }
}

/* Whole-file encoder (:295-399, :538-700): returns the number of bytes written to `out`
 * (capacity must be >= 8 + frames * QOA_FRAME_SIZE).  If recon != NULL it receives the encoder's own
 * reconstruction, which a correct decoder must reproduce exactly. */
size_t afgo_qoa_encode(const int16_t *pcm, uint32_t samples, int channels, uint32_t samplerate,
                       uint8_t *out, int16_t *recon)
{
    lms_t lms[8];
    for (int c = 0; c < channels; c++) {                       /* :578-589 */
        lms[c].weights[0] = 0; lms[c].weights[1] = 0;
        lms[c].weights[2] = -(1 << 13); lms[c].weights[3] = (1 << 14);
        for (int i = 0; i < QOA_LMS_LEN; i++) lms[c].history[i] = 0;
    }
    uint8_t *p = out;
    wr64be(p, ((uint64_t)QOA_MAGIC << 32) | samples); p += 8;  /* :413-453 */
    for (uint32_t s0 = 0; s0 < samples; s0 += QOA_FRAME_LEN) {
        uint32_t frame_len = samples - s0 < QOA_FRAME_LEN ? samples - s0 : QOA_FRAME_LEN;
        const int16_t *sample_data = pcm + (size_t)s0 * channels;
        uint32_t slices = (frame_len + QOA_SLICE_LEN - 1) / QOA_SLICE_LEN;
        uint32_t frame_size = 8 + QOA_LMS_LEN * 4 * channels + 8 * slices * channels;
        wr64be(p, (uint64_t)channels << 56 | (uint64_t)samplerate << 32 | (uint64_t)frame_len << 16 | frame_size); p += 8;
        for (int c = 0; c < channels; c++) {
            uint64_t weights = 0, history = 0;
            for (int i = 0; i < QOA_LMS_LEN; i++) {
                history = (history << 16) | (uint64_t)(lms[c].history[i] & 0xffff);
                weights = (weights << 16) | (uint64_t)(lms[c].weights[i] & 0xffff);
            }
            wr64be(p, history); p += 8;
            wr64be(p, weights); p += 8;
            /* like the reference, the encoder keeps its int state: a decoder restarts every frame from
             * the 16-bit values above, so the two agree as long as the weights fit 16 bits */
        }
        for (uint32_t sample_index = 0; sample_index < frame_len; sample_index += QOA_SLICE_LEN) {
            for (int c = 0; c < channels; c++) {
                int slice_len = (int)(frame_len - sample_index); if (slice_len > QOA_SLICE_LEN) slice_len = QOA_SLICE_LEN;
                int slice_start = (int)sample_index * channels + c;
                int slice_end = ((int)sample_index + slice_len) * channels + c;
                uint64_t best_error = ~(uint64_t)0, best_slice = 0;
                lms_t best_lms = lms[c];
                int16_t best_rec[QOA_SLICE_LEN];
                memset(best_rec, 0, sizeof(best_rec));
                for (int scalefactor = 0; scalefactor < 16; scalefactor++) {
                    lms_t l = lms[c];
                    uint64_t slice = (uint64_t)scalefactor, current_error = 0;
                    int16_t rec[QOA_SLICE_LEN];
                    int k = 0;
                    for (int si = slice_start; si < slice_end; si += channels, k++) {
                        int sample = sample_data[si];
                        int predicted = lms_predict(&l);
                        int residual = sample - predicted;
                        int scaled = qoa_div(residual, scalefactor);
                        int clamped = scaled < -8 ? -8 : (scaled > 8 ? 8 : scaled);
                        int quantized = k_quant_tab[clamped + 8];
                        int dequantized = k_dequant_tab[scalefactor][quantized];
                        int reconstructed = clamp_s16(predicted + dequantized);
                        long long error = sample - reconstructed;
                        current_error += (uint64_t)(error * error);
                        if (current_error > best_error) break;
                        lms_update(&l, reconstructed, dequantized);
                        slice = (slice << 3) | (uint64_t)quantized;
                        rec[k] = (int16_t)reconstructed;
                    }
                    if (current_error < best_error) {
                        best_error = current_error;
                        best_slice = slice;
                        best_lms = l;
                        memcpy(best_rec, rec, sizeof(rec));
                    }
                }
                lms[c] = best_lms;
                best_slice <<= (QOA_SLICE_LEN - slice_len) * 3;
                wr64be(p, best_slice); p += 8;
                if (recon)
                    for (int k = 0; k < slice_len; k++)
                        recon[((size_t)s0 + sample_index + (size_t)k) * channels + c] = best_rec[k];
            }
        }
    }
    return (size_t)(p - out);
}

/* ------------------------------------------------------------------------------------------------------------------
 * The stream layer: qoa_decode_header (qoa.d:413-453), QOADecoder.initialize (:770-791) and the frame loop of
 * QOADecoder.readSamples (:803-851) over a memory stream, to the end of the stream.  qoa_decode_frame (:455-534) is
 * restated again here AS A READER: it consumes the frame header, the LMS state and ceil(samples / 20) slices per
 * channel from the cursor and leaves the cursor there -- the frame-size field is only checked against the bytes left
 * (:477) and against the sample count (:481-486), it never positions the next frame.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { const uint8_t *d; size_t n, pos; } qoa_io;

static int io_u64(qoa_io *io, uint64_t *v)                    /* read_ulong_BE, io.d:164-190: all eight bytes or an error */
{
    if (io->n - io->pos < 8) { io->pos = io->n; return 0; }
    *v = rd64be(io->d + io->pos);
    io->pos += 8;
    return 1;
}

static uint32_t qoa_stream_frame(qoa_io *io, uint32_t want_channels, uint32_t want_rate, int16_t *sample_data, int *overrun)
{
    if ((long long)(io->n - io->pos) < 8 + QOA_LMS_LEN * 4 * (long long)want_channels) return 0;        /* :460 */
    uint64_t fh;
    if (!io_u64(io, &fh)) return 0;
    int channels = (int)((fh >> 56) & 0xff);
    int samplerate = (int)((fh >> 32) & 0xffffff);
    int samples = (int)((fh >> 16) & 0xffff);
    int frame_size = (int)(fh & 0xffff);
    int data_size = frame_size - 8 - QOA_LMS_LEN * 4 * channels;
    int num_slices = data_size / 8;
    int max_total_samples = num_slices * QOA_SLICE_LEN;
    if ((long long)(io->n - io->pos) < frame_size - 8) return 0;                                         /* :477 */
    if (channels != (int)want_channels || samplerate != (int)want_rate || samples * channels > max_total_samples) return 0;
    if (samples > QOA_FRAME_LEN) { *overrun = 1; return 0; }      /* the reference writes past its QOA_FRAME_LEN buffer (:786): undefined */
    lms_t lms[8];
    for (int c = 0; c < channels; c++) {
        uint64_t history, weights;
        if (!io_u64(io, &history)) return 0;
        if (!io_u64(io, &weights)) return 0;
        for (int i = 0; i < QOA_LMS_LEN; i++) {
            lms[c].history[i] = (int16_t)(history >> 48);
            history <<= 16;
            lms[c].weights[i] = (int16_t)(weights >> 48);
            weights <<= 16;
        }
    }
    for (int sample_index = 0; sample_index < samples; sample_index += QOA_SLICE_LEN) {
        for (int c = 0; c < channels; c++) {
            uint64_t slice;
            if (!io_u64(io, &slice)) return 0;
            int scalefactor = (int)((slice >> 60) & 0xf);
            int slice_start = sample_index * channels + c;
            int end = sample_index + QOA_SLICE_LEN; if (end > samples) end = samples;
            int slice_end = end * channels + c;
            for (int si = slice_start; si < slice_end; si += channels) {
                int predicted = lms_predict(&lms[c]);
                int quantized = (int)((slice >> 57) & 0x7);
                int dequantized = k_dequant_tab[scalefactor][quantized];
                int reconstructed = clamp_s16(predicted + dequantized);
                sample_data[si] = (int16_t)reconstructed;
                slice <<= 3;
                lms_update(&lms[c], reconstructed, dequantized);
            }
        }
    }
    return (uint32_t)samples;
}

#include <stdlib.h>

int afgo_qoa_decode_file(const uint8_t *data, size_t size, afgo_qoa_file *out)
{
    memset(out, 0, sizeof(*out));
    qoa_io io = { data, size, 0 };
    if (size < 16) return -1;                                      /* QOA_MIN_FILESIZE, :416 */
    uint64_t fh, first;
    if (!io_u64(&io, &fh) || (fh >> 32) != QOA_MAGIC) return -1;
    out->samples = (uint32_t)(fh & 0xffffffffu);
    if (!out->samples) return -1;
    if (!io_u64(&io, &first)) return -1;                           /* peek into the first frame header, :441-451 */
    out->channels = (uint32_t)((first >> 56) & 0xff);
    out->samplerate = (uint32_t)((first >> 32) & 0xffffff);
    if (out->channels == 0 || out->samplerate == 0) return -1;
    if (out->channels > 8) return -1;                              /* qoa_desc.lms has eight entries (:221): more is an overrun in the reference */
    io.pos = 8;                                                    /* QOADecoder.initialize seeks back to the first frame, :781 */
    size_t cap = (size_t)QOA_FRAME_LEN * out->channels * 4, n = 0;
    int16_t *pcm = (int16_t *)malloc(cap * sizeof(int16_t));
    int16_t *buffer = (int16_t *)calloc((size_t)QOA_FRAME_LEN * out->channels, sizeof(int16_t));
    if (!pcm || !buffer) { free(pcm); free(buffer); return -2; }
    for (;;) {
        int overrun = 0;
        uint32_t frame_len = qoa_stream_frame(&io, out->channels, out->samplerate, buffer, &overrun);
        if (frame_len == 0) break;                                 /* readSamples returns what it has, :813-814 */
        size_t cnt = (size_t)frame_len * out->channels;
        if (n + cnt > cap) {
            while (n + cnt > cap) cap *= 2;
            int16_t *np = (int16_t *)realloc(pcm, cap * sizeof(int16_t));
            if (!np) { free(pcm); free(buffer); return -2; }
            pcm = np;
        }
        memcpy(pcm + n, buffer, cnt * sizeof(int16_t));
        n += cnt;
        out->n_qoa_frames++;
    }
    free(buffer);
    out->pcm = pcm;
    out->n_frames_pcm = n / out->channels;
    return 0;
}

void afgo_qoa_file_free(afgo_qoa_file *file)
{
    free(file->pcm);
    memset(file, 0, sizeof(*file));
}
