/*
 * afg.h -- C ABI of the MI355X-native batched audio-decode transform path.
 *
 * Drop-in boundary for the transform stage of AuburnSounds/audio-formats'
 * decoders (reference paths are relative to /root/reference).  The reference
 * is pure D and has no FFI layer; the seams below are the D function calls a
 * maintainer would redirect to this library (binding stub: INTEGRATION.md,
 * bindings/d/afgpu.d).  Bitstream / entropy decoding stays on the host; only
 * the per-frame transform stage runs on the device.
 *
 * Conventions (reference: stream.d:31-33, :105, internals.d:16-23):
 *   - no exceptions, nothing aborts; every entry returns an afg_status (0 = ok)
 *   - a handle is not thread-safe; distinct handles share no mutable state
 *   - d_* pointers are device (HIP) pointers, everything else is host memory
 *   - hip_stream is a hipStream_t passed as void* (NULL = default stream);
 *     *_hip entries only enqueue work, they never synchronise
 *   - the persistent kernels (Vorbis, CELT) draw their work from per-launch counters kept in rings: 32 launches of
 *     one Vorbis plan, 64 CELT launches per device may be in flight at once on different streams (launches on ONE
 *     stream run in order and cannot collide)
 *   - the library fails loudly (AFG_ERR_NO_DEVICE) when no gfx950 device or
 *     no device code is available: there is no CPU fallback in the product.
 */
#ifndef AFG_H
#define AFG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): afg_dev_option replaced the per-call AFG_* environment knobs of version 1 (round 5 added the call without
 * counting); afg_host_pool_trim also releases pooled device planes. */
#define AFG_ABI_VERSION 2

typedef enum afg_status {
    AFG_OK              =  0,
    AFG_ERR_INVALID     = -1,   /* bad argument / inconsistent batch description */
    AFG_ERR_NO_DEVICE   = -2,   /* no HIP device, or device code not loadable    */
    AFG_ERR_HIP         = -3,   /* a HIP runtime call failed (see afg_last_error) */
    AFG_ERR_OOM         = -4,
    AFG_ERR_UNSUPPORTED = -5
} afg_status;

int         afg_abi_version(void);
const char *afg_status_string(int status);
const char *afg_last_error(void);          /* thread-local detail of the last failure, never NULL */
int         afg_device_count(void);        /* number of HIP devices visible, <0 on error */

/* Numeric mode of the float transform stages, process-wide.
 *   AFG_NUMERIC_EXACT      every float32 result is produced by the reference's own expression tree (no fused
 *                          multiply-adds, recurrences in the reference's order): bit-identical to the D decoders' arithmetic.
 *   AFG_NUMERIC_TOLERANCE  (default) results within the 1e-5 RMS of the decoders' float output that drop-in use asks
 *                          for; lets the Opus/CELT stage re-associate the de-emphasis recurrence (dopus.d:3695-3701) into a
 *                          prefix sum inside the frame walk, fuse multiply-adds there, and cut a stream into independently
 *                          walked segments wherever the post-filter is provably idle (dopus.d:3294-3296, :3333); lets the
 *                          Vorbis stage compute inverse_mdct (stb_vorbis2.d:1941-2242) of 1024-, 2048- and 4096-sample long
 *                          blocks as ONE complex FFT of n/4 points with fused multiply-adds instead of scheduling the reference's
 *                          8-step algorithm, and write window + overlap (:2606-2657) on the transform's DCT-IV (csrc/vorbis_walk.hip);
 *                          lets the MP3 stage fuse multiply-adds and sum the polyphase window's sixteen products per output
 *                          (minimp3.d:1371-1405) in one chain (csrc/mp3_tolerance.hip).
 * FLAC and QOA (integer work) compute the same bits in both modes.  The environment variable AFG_NUMERIC=exact|tolerance
 * decides until afg_set_numeric_mode is called, and again after afg_set_numeric_mode(AFG_NUMERIC_FROM_ENV).  Returns the
 * mode that was in effect before, AFG_ERR_INVALID for an unknown one. */
#define AFG_NUMERIC_FROM_ENV  (-1)
#define AFG_NUMERIC_EXACT     0
#define AFG_NUMERIC_TOLERANCE 1
int         afg_set_numeric_mode(int mode);

/* Test hooks, not part of the reference's surface: alternative code paths that deliver the same samples as the default
 * ones, which the test-suite runs against each other.  The library reads no environment variable for them (rounds 1-4
 * did, on every call); value < 0 returns an option to "not set".  Names: "celt_path" (1 stream walk, 2 split kernels --
 * the two bit-exact paths --, 3 the tolerance-mode walk), "celt_de_seq" (2 .. 32 sequences per de-emphasis wavefront),
 * "celt_de_duo" (0 / 1), "celt_seg_recs", "celt_whole_frames", "vorbis_single" (1: one channel per wavefront),
 * "mp3_chunks", "mp3_float_upload" (1: float spectra instead of quantised values cross the bus), "vorbis_host_floor"
 * (1: floor curves on the host), "flac_host_res32" (1: int32 residual rows only), "vorbis_seg_packets" (packets per walk
 * item of plans created with seg_packets = 0, instead of the library's choice), "batch_groups" (groups of files a batch call
 * pipelines: 1 = none; default 4 for batches of FLAC / Ogg Vorbis files from 512 files up, else 1).  AFG_ERR_INVALID: no such name. */
int         afg_dev_option(const char *name, int value);
int         afg_get_numeric_mode(void);
int         afg_device_name(int device, char *buf, size_t buflen);

/* ========================================================================== *
 *  MP3 Layer III transform stage
 *  replaces, for every granule of every stream of a batch:
 *    minimp3.d:1226-1228  L3_antialias -> L3_imdct_gr -> L3_change_sign
 *    minimp3.d:1553       mp3d_synth_granule (mp3d_DCT_II + 9 x mp3d_synth)
 *  Input is the dequantised, stereo-processed, reordered spectrum that
 *  L3_decode holds in scratch.grbuf right before minimp3.d:1226.
 * ========================================================================== */

/* per gr-ch flag word */
#define AFG_MP3_FLAGS(block_type, n_long_bands, aa_bands) \
    ((uint32_t)(block_type) | ((uint32_t)(n_long_bands) << 8) | ((uint32_t)((aa_bands) + 1) << 16))
/*   block_type    gr_info.block_type (0 normal, 1 start, 2 short, 3 stop)
 *   n_long_bands  minimp3.d:1218 (0, 2 or 4)
 *   aa_bands      minimp3.d:1217/1222 (31, or n_long_bands-1 for short blocks; -1 = none) */

/* Optional, OR-ed into the flag word: the first `bands` subbands (0..32) are the only ones that may hold nonzero
 * lines; the lines of the others must be +0.0 (what L3_huffman's zero fill leaves above the last coded line,
 * minimp3.d:868-883) and are not read by the device.  Absent (field 0): all 576 lines are read. */
#define AFG_MP3_NZ_BANDS(bands) ((uint32_t)((bands) + 1) << 24)

/* Optional, OR-ed into the flag word: the block holds subband samples, not a spectrum -- what the Layer I / II decoder
 * hands to mp3d_synth_granule (minimp3.d:1563-1566): index band * 18 + time slot.  Alias reduction, IMDCT and frequency
 * inversion are skipped, the 18 slots go to the synthesis (mp3d_DCT_II + mp3d_synth) as they are.  The reference runs
 * that synthesis 12 slots at a time; it is a filterbank over slot pairs with a 15-slot history and no notion of a
 * granule, so a front-end packs the slots of a run of frames into 18-slot blocks (three 12-slot granules = two blocks). */
#define AFG_MP3_SUBBAND 0x80000000u

#define AFG_MP3_STATE_FLOATS 1536   /* opaque per-stream carry state (same size as mdct_overlap+qmf_state, minimp3.d:40-41) */

typedef struct afg_mp3_plan afg_mp3_plan;

/* Describe a batch: stream s has granules[s] granules of channels[s] (1|2)
 * channels.  Blocks of 576 floats are laid out stream after stream, inside a
 * stream as [granule][channel]; PCM uses the same float offsets, 576*nch
 * interleaved floats per granule (minimp3.d:1549 `pcm += 576*channels`).
 * seg_granules = granules a wavefront walks sequentially (0 = default). */
int      afg_mp3_plan_create(afg_mp3_plan **plan, uint32_t n_streams, const uint32_t *granules,
                             const uint8_t *channels, uint32_t seg_granules);
void     afg_mp3_plan_destroy(afg_mp3_plan *plan);
uint64_t afg_mp3_plan_blocks(const afg_mp3_plan *plan);     /* total gr-ch blocks = floats/576 */
uint32_t afg_mp3_plan_segments(const afg_mp3_plan *plan);   /* workgroups one launch uses */

/* d_coef   : blocks*576 floats      d_flags : blocks words (AFG_MP3_FLAGS)
 * d_pcm    : blocks*576 floats, scaled by 1/32768, not clipped (minimp3.d:1300)
 * d_state  : NULL (every stream starts from zero state, minimp3.d:1509) or
 *            n_streams*AFG_MP3_STATE_FLOATS floats read at the first granule of
 *            each stream and rewritten after its last (chunked decoding). */
int afg_mp3_transform_hip(const afg_mp3_plan *plan, const float *d_coef, const uint32_t *d_flags,
                          float *d_pcm, float *d_state, void *hip_stream);

/* -------------------------------------------------------------------------- *
 *  MP3 requantisation on the device (ship the Huffman values, not floats)
 *  replaces, between the entropy decoder and the transform stage above:
 *    minimp3.d:722-746   L3_pow_43 and the `* sf` of L3_huffman (:835-858, :868-879)
 *    minimp3.d:885-982   L3_midside_stereo / L3_intensity_stereo / L3_stereo_process
 *    minimp3.d:984-1000  L3_reorder
 *  The host keeps Huffman decoding and the scalefactor arithmetic (L3_decode_scalefactors, :616-719: 39 floats per
 *  granule-channel) and decides the stereo plan (which band is mid/side or intensity coded, with which factors); the
 *  device turns int16 values into exactly the floats L3_decode holds in grbuf at :1226 -- 2 bytes per line cross the
 *  bus instead of 4.  Two cases are not covered: MPEG-2.5 8 kHz mixed blocks (their reorder walks outside the channel,
 *  :1218-1223) and a MONO frame whose header carries the intensity-stereo bit (the reference runs L3_intensity_stereo over
 *  the one channel and the scratch row behind it, :100, :1207-1210: damaged files).  afg_mp3_parse_q reports
 *  AFG_ERR_UNSUPPORTED for such a file and the float path takes it.
 * -------------------------------------------------------------------------- */
#define AFG_MP3_NO_SDESC 0xffffffffu

typedef struct afg_mp3_qgranule {
    uint64_t q_off;          /* int16 index of channel 0's 576 values; channel c at q_off + 576*c */
    uint64_t coef_off;       /* float index of channel 0's block in the coefficient plane; channel c at + 576*c */
    uint32_t sdesc;          /* stereo == 2: index of the granule's afg_mp3_sdesc, else AFG_MP3_NO_SDESC */
    uint8_t  nch;            /* 1 | 2; 0: unused record slot, skipped */
    uint8_t  stereo;         /* 0 none, 1 mid/side on every line (:1203), 2 per band (intensity frames, :1201) */
    uint8_t  table[2];       /* per channel: scalefactor-band table (kind*8 + rate row; kind 0 long, 1 short, 2 mixed),
                                bit 7: the short part is reordered (block_type 2) */
    float    scale[2][40];   /* band scales of each channel (scf[] of L3_decode_scalefactors) */
} afg_mp3_qgranule;          /* 344 bytes */

typedef struct afg_mp3_sdesc {
    uint8_t type[40];        /* per band of channel 0's table: 0 leave, 1 mid/side, 2 intensity */
    float   fl[40], fr[40];  /* intensity: right = left * fr, then left = left * fl (:929-936) */
} afg_mp3_sdesc;             /* 360 bytes */

/* d_q: quantised lines (sign included, |v| <= 8206); d_coef receives 576 floats per granule-channel. */
int afg_mp3_requant_hip(uint64_t n_granules, const afg_mp3_qgranule *d_granules, const int16_t *d_q,
                        const afg_mp3_sdesc *d_sdesc, float *d_coef, void *hip_stream);

/* ========================================================================== *
 *  Vorbis transform stage
 *  replaces stb_vorbis2.d:2526-2527 (inverse_mdct per channel) and
 *  stb_vorbis2.d:2606-2657 (vorbis_finish_frame: window + overlap-add),
 *  plus the interleave of stb_vorbis2.d:3927-3952.  Tables are those of
 *  stb_vorbis2.d:851-898, built on the host at plan creation.
 * ========================================================================== */

#define AFG_VORBIS_LONG 1u   /* mode blockflag          (stb_vorbis2.d:2324) */
#define AFG_VORBIS_PREV 2u   /* previous-window flag    (stb_vorbis2.d:2326) */
#define AFG_VORBIS_NEXT 4u   /* next-window flag        (stb_vorbis2.d:2327) */
/* Optional, bits 4..7 of a LONG packet's flag byte: only the first e eighths of every channel's n/2 spectral values may be
 * different from +0.0 (e = 0 .. 8) -- what the host decoder knows from the residue's `end` (stb_vorbis2.d:1586-1600: bins past
 * it are never written, inverse coupling and the floor multiply keep +0.0).  The device then need not fetch the rest (the
 * tolerance-mode walk does not; the bit-exact kernels ignore the declaration).  0 in these bits: nothing declared. */
#define AFG_VORBIS_NZ_EIGHTHS(e) ((((unsigned)(e)) + 1u) << 4)

typedef struct afg_vorbis_plan afg_vorbis_plan;

/* Stream s: packets[s] audio packets, channels[s] channels, block sizes
 * blocksize0/1[s] (powers of two, 256..8192; 64/128 are rejected).
 * pflags: one byte per packet, streams concatenated.
 * Packet p reads channels*(n/2) floats ([ch][n/2]) at spec offset p and writes
 * (right_start-left_start)*channels interleaved floats at out offset p; the
 * first packet of a stream produces no output (stb_vorbis2.d:2645-2649). */
int      afg_vorbis_plan_create(afg_vorbis_plan **plan, uint32_t n_streams, const uint32_t *packets,
                                const uint8_t *channels, const uint16_t *blocksize0,
                                const uint16_t *blocksize1, const uint8_t *pflags, uint32_t seg_packets);
void     afg_vorbis_plan_destroy(afg_vorbis_plan *plan);
uint64_t afg_vorbis_plan_packets(const afg_vorbis_plan *plan);
uint64_t afg_vorbis_plan_spec_floats(const afg_vorbis_plan *plan);
uint64_t afg_vorbis_plan_out_floats(const afg_vorbis_plan *plan);
/* copies the per-packet float offsets (total packets entries each; either may be NULL) */
int      afg_vorbis_plan_offsets(const afg_vorbis_plan *plan, uint64_t *spec_off, uint64_t *out_off);

int afg_vorbis_transform_hip(const afg_vorbis_plan *plan, const float *d_spec, float *d_out, void *hip_stream);

/* Vorbis inverse coupling and floor curve on the device: replaces stb_vorbis2.d:2493-2514 (inverse coupling),
 * :2516-2523 / :2255-2284 (do_floor, silent channels) and :1534-1563 (draw_line) between the host's residue decode and the
 * transform above.  d_spec holds the decoded *residue* vectors in the transform's layout ([channel][n2] per packet) and is
 * rewritten in place with the spectra.
 * A curve is the list of floor-1 points that survive step 2 (finalY >= 0), in sorted_order, as int32 pairs
 * (x = Xlist[j], y = finalY[j] * floor1_multiplier), the first one at x = 0.  n_points == 0: really_zero_channel.
 * Coupling steps are byte pairs (magnitude channel, angle channel) in the order they are applied (coupling_steps-1 .. 0). */
typedef struct afg_vorbis_floor_packet {
    uint64_t spec_off;     /* float index of channel 0's n2 values; channel c at spec_off + c*n2 */
    uint32_t n2;           /* blocksize / 2, a multiple of 4 */
    uint32_t channels;
    uint32_t curve_index;  /* channel 0's afg_vorbis_floor_curve; channel c at curve_index + c */
    uint32_t step_off;     /* first coupling step: bytes d_steps[2*step_off], d_steps[2*step_off + 1] */
    uint32_t n_steps;
    uint32_t pad;
} afg_vorbis_floor_packet; /* 32 bytes */

typedef struct afg_vorbis_floor_curve {
    uint32_t point_off;    /* first point: d_points[2*point_off] = x, d_points[2*point_off + 1] = y */
    uint32_t n_points;
} afg_vorbis_floor_curve;  /* 8 bytes */

int afg_vorbis_floor_hip(uint64_t n_packets, const afg_vorbis_floor_packet *d_packets, const afg_vorbis_floor_curve *d_curves,
                         const int32_t *d_points, const uint8_t *d_steps, float *d_spec, void *hip_stream);

/* ========================================================================== *
 *  FLAC sample restore
 *  replaces drflac.d:1235 (residual + drflac__calculate_prediction_32/_64,
 *  drflac.d:1060-1140) for whole subframes, and the decorrelate / shift /
 *  interleave of drflac_read_s32 (drflac.d:2885-2941); optionally also the
 *  int32 -> float conversion of stream.d:505-511.  Bit-exact int32.
 * ========================================================================== */

#define AFG_FLAC_INDEPENDENT 0
#define AFG_FLAC_LEFT_SIDE   8    /* DRFLAC_CHANNEL_ASSIGNMENT_LEFT_SIDE  */
#define AFG_FLAC_RIGHT_SIDE  9    /* DRFLAC_CHANNEL_ASSIGNMENT_RIGHT_SIDE */
#define AFG_FLAC_MID_SIDE   10    /* DRFLAC_CHANNEL_ASSIGNMENT_MID_SIDE   */

typedef struct afg_flac_subframe {
    int16_t coef[32];   /* LPC coefficients (fixed predictors: drflac.d:1397-1403 table, shift 0) */
    uint8_t order;      /* 0..32; warm-up samples occupy res[0..order) (constant/verbatim: order 0) */
    uint8_t shift;      /* lpcShift 0..31 */
    uint8_t wasted;     /* wastedBitsPerSample */
    uint8_t use64;      /* subframe bitsPerSample > 16 -> 64-bit accumulator (drflac.d:1308) */
} afg_flac_subframe;     /* 68 bytes */

/* Residual rows of a frame whose residuals and warm-up samples all fit 16 bits may be stored as int16 (half
 * the bytes on the bus and in HBM): rows padded to 16 bytes. */
#define AFG_FLAC_ROW16(block_size) (((uint64_t)(block_size) + 7u) & ~(uint64_t)7u)

typedef struct afg_flac_frame {
    uint64_t in_off;      /* res16 == 0: int32 index of channel 0's residual row; channel c at in_off + c*block_size.
                             res16 == 1: int16 index (d_res viewed as int16_t), a multiple of 8; channel c at
                             in_off + c*AFG_FLAC_ROW16(block_size); the row padding is read, never used */
    uint64_t out_off;     /* int32 index of the interleaved output (block_size*channels samples) */
    uint32_t block_size;
    uint32_t sf_index;    /* index of channel 0's afg_flac_subframe; channel c at sf_index + c */
    uint8_t  channels;    /* 1..8 */
    uint8_t  assignment;  /* AFG_FLAC_* */
    uint8_t  bps;         /* STREAMINFO bitsPerSample */
    uint8_t  res16;       /* 0: int32 residual rows, 1: int16 rows */
    uint8_t  pad[4];
} afg_flac_frame;         /* 32 bytes */

/* d_out_i32 and/or d_out_f32 may be NULL (at least one must be given). */
int afg_flac_transform_hip(uint64_t n_frames, const afg_flac_frame *d_frames,
                           const afg_flac_subframe *d_subframes, const int32_t *d_res,
                           int32_t *d_out_i32, float *d_out_f32, void *hip_stream);

/* The restore kernel exists in 8 instantiations (LPC-order bucket <= 4 / 8 / 12 / 32 x 64-bit accumulator, so that each gets
 * the registers it needs and no more) plus 4 of a kernel for groups of frames with one channel count above two (round 6:
 * order <= 12 / 32 x accumulator); a wavefront of 32 consecutive frames -- a lane per subframe -- runs in the one its
 * largest order and widest subframe select.  afg_flac_transform_hip is stream-ordered -- the records are read by the
 * device when hip_stream gets there, never by the host at the call -- so it launches all 12; the ones nobody selects exit
 * at once.  A caller that still holds the FINAL records in host memory can say which are populated:
 * afg_flac_variants (host pointers, pure host code) returns the set as a bit mask, and afg_flac_transform_variants_hip
 * launches only those -- two or more of them side by side on the caller's stream and an internal one, joined before
 * the call returns to the stream's order.  Frames of an instantiation missing from `variants` are NOT decoded.  The mask
 * is only meaningful within the process that computed it. */
uint32_t afg_flac_variants(uint64_t n_frames, const afg_flac_frame *frames, const afg_flac_subframe *subframes);
int afg_flac_transform_variants_hip(uint64_t n_frames, const afg_flac_frame *d_frames,
                                    const afg_flac_subframe *d_subframes, const int32_t *d_res,
                                    int32_t *d_out_i32, float *d_out_f32, uint32_t variants, void *hip_stream);

/* ========================================================================== *
 *  QOA frame decode (LMS predict / dequantise / clamp / update)
 *  replaces the slice loop of qoa_decode_frame (qoa.d:489-530, qoa_lms_predict /
 *  qoa_lms_update :231-254) and the float conversion of QOADecoder.readSamples
 *  (qoa.d:831-838).  Input is the raw file bytes: the host only locates frames
 *  (qoa.d:465-486) -- the LMS state and the 64-bit slices are read on the device.
 * ========================================================================== */

typedef struct afg_qoa_frame {
    uint64_t byte_off;    /* offset of the 8-byte frame header in the byte plane (multiple of 8) */
    uint64_t out_off;     /* index of the frame's first output value (interleaved) */
    uint16_t samples;     /* samples per channel in this frame, <= 5120 (header field, qoa.d:476) */
    uint8_t  channels;    /* 1..8 (header field) */
    uint8_t  pad[5];
} afg_qoa_frame;           /* 24 bytes */

/* d_out_i16 (qoa_decode_frame's sample_data) and/or d_out_f32 (value * (1.0f/32767)) may be NULL. */
int afg_qoa_transform_hip(uint64_t n_frames, const afg_qoa_frame *d_frames, const uint8_t *d_bytes,
                          int16_t *d_out_i16, float *d_out_f32, void *hip_stream);

/* ========================================================================== *
 *  Opus / CELT transform stage
 *  replaces the per-channel tail of ff_celt_decode_frame (dopus.d:3680-3702):
 *  imdct15_half (dopus.d:1611-1637) + vector_fmul_window (dopus.d:230-243) per
 *  block, celt_postfilter (dopus.d:3281-3378) and de-emphasis / output scaling
 *  (dopus.d:3695-3701).  Input is coeffs[ch] after celt_denormalize / downmix
 *  (dopus.d:3653-3668); output is ff_celt_decode_frame's float output[ch][].
 *  The post-filter and de-emphasis are recursive over the whole stream, so the
 *  unit of parallelism is a channel sequence: all frames of one output channel
 *  of one stream, processed in order.
 * ========================================================================== */

typedef struct afg_celt_frame {
    uint64_t coef_off;       /* float index of coeffs[ch][0]: frame_size floats, short blocks interleaved */
    uint64_t out_off;        /* float index of output sample 0 */
    uint32_t out_stride;     /* distance between consecutive output samples (1 = planar) */
    uint16_t frame_size;     /* 120, 240, 480 or 960 */
    uint8_t  blocks;         /* 1, or 1 << duration when transient (dopus.d:3630) */
    uint8_t  pad;
    int32_t  pf_period_new;  /* dopus.d:3407 (>= 15) */
    float    pf_gains_new[3];
    float    imdct_scale;    /* 1.0, or 0.5 for the stereo -> mono downmix (dopus.d:3665) */
    uint32_t pad2;
} afg_celt_frame;            /* 48 bytes */

#define AFG_CELT_STATE_FLOATS 2064   /* afg_celt_state: buf[2048] + post-filter + de-emphasis memory */

/* Channel sequence k owns records [rec_base[k], rec_base[k+1]) (n_chan + 1 entries).
 * d_states: NULL (zero state: a fresh decoder) or n_chan * AFG_CELT_STATE_FLOATS words read
 * before the first and rewritten after the last frame of each sequence (chunked decoding).
 * Sequences 2p and 2p + 1 are walked by one wavefront, half each, when they are the two channels of a stereo stream
 * (equally long, records of equal geometry, out_stride 2, out_off even and out_off + 1): put a stereo stream's channels on
 * an even and the following odd index -- an empty sequence (rec_base[k] == rec_base[k+1]) after an odd number of mono
 * streams does it, as afg_batch_decode does.  Any layout is decoded correctly; in AFG_NUMERIC_TOLERANCE the two forms
 * of the walk round differently (both within the tolerance), in AFG_NUMERIC_EXACT they are the same bits. */
int afg_celt_transform_hip(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs,
                           const float *d_coeffs, float *d_out, float *d_states, void *hip_stream);

/* The same with the sequential part on a second stream.  The post-filter and the de-emphasis are serial chains per
 * channel sequence: with few, long sequences (a file-sharded mixed corpus holds ~1600 of up to 1.4 M samples per wave)
 * they occupy a fraction of the device for the length of the longest sequence.  Here the record-parallel transform is
 * queued on hip_stream and the per-sequence passes on hip_tail_stream behind an event, so that they run beside whatever
 * the caller queues on hip_stream next; the caller joins the two streams (an event on hip_tail_stream) before it reads
 * d_out -- and before it frees or overwrites d_coeffs, d_recs or d_rec_base: in AFG_NUMERIC_TOLERANCE the whole walk,
 * input reads included, runs on hip_tail_stream.  hip_tail_stream NULL or equal to hip_stream: afg_celt_transform_hip. */
int afg_celt_transform_streams_hip(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs,
                                   const float *d_coeffs, float *d_out, float *d_states, void *hip_stream,
                                   void *hip_tail_stream);

/* What OpusFile.readFrame and AudioStream.readSamplesFloat do to the decoder's floats (dopus.d:7923-7926,
 * :8098-8105; stream.d:480): Float2IntScaled (x * 32768 rounded to nearest even by a magic-number add, saturated
 * to int16), then int16 / 32767.0f.  Element-wise; d_out_f32 may alias d_in; either output may be NULL. */
int afg_opus_output_hip(uint64_t n_samples, const float *d_in, int16_t *d_out_i16, float *d_out_f32, void *hip_stream);
/* The same behind opus_decode_packet's output gain (dopus.d:6688-6691: every float times OpusContext.gain, applied when the
 * header gain + R128_TRACK_GAIN is not zero): x * gain rounded to float, then the conversion above. */
int afg_opus_output_gain_hip(uint64_t n_samples, const float *d_in, float gain, int16_t *d_out_i16, float *d_out_f32,
                             void *hip_stream);

/* ========================================================================== *
 *  Outer surface: the AudioStream subset (stream.d:102-637) over the host front-ends
 *  -- FLAC (native container, drflac.d:680-1695, :1887-2153), QOA (qoa.d:413-486, :703-851), MP3 Layer I / II / III
 *  (minimp3.d, minimp3_ex.d), Ogg Vorbis (stb_vorbis2.d) and Ogg Opus with CELT-only packets (dopus.d; a file that holds
 *  SILK / hybrid packets is refused at open with this library's own message).  WAV, MOD and XM report "unrecognized
 *  encoding".  Like the reference the stream decodes as the caller pulls: afg_open_from_memory parses the container only,
 *  a read that finds the FIFO empty decodes the next chunk (64 MP3 frames / Vorbis or Opus packets, 16 FLAC or QOA frames)
 *  on the device; afg_batch_decode parses whole files into transform-stage records and decodes them in one pass.
 * ========================================================================== */

typedef enum afg_format {          /* AudioFileFormat, stream.d:36-47 */
    AFG_FORMAT_WAV = 0, AFG_FORMAT_MP3 = 1, AFG_FORMAT_FLAC = 2, AFG_FORMAT_OGG = 3, AFG_FORMAT_OPUS = 4,
    AFG_FORMAT_QOA = 5, AFG_FORMAT_MOD = 6, AFG_FORMAT_XM = 7, AFG_FORMAT_UNKNOWN = 8
} afg_format;

#define AFG_UNKNOWN_LENGTH (-1)    /* audiostreamUnknownLength, stream.d:90 */

typedef struct afg_stream afg_stream;

/* openFromMemory (stream.d:150-170): copies the bytes, as the reference does (stream.d:2031-2041) -- `data` may be freed
 * when the call returns; never throws; returns NULL only when out of memory.  On failure the stream is in error state with the
 * reference's message (internals.d:16-23). */
afg_stream *afg_open_from_memory(const uint8_t *data, size_t length);
int         afg_is_error(const afg_stream *s);                 /* stream.d:295-301 */
const char *afg_error_message(const afg_stream *s);            /* NULL when valid, stream.d:310-316 */
int         afg_get_format(const afg_stream *s);               /* afg_format */
int         afg_get_num_channels(const afg_stream *s);
int64_t     afg_get_length_in_frames(const afg_stream *s);     /* AFG_UNKNOWN_LENGTH if unknown */
float       afg_get_samplerate(const afg_stream *s);
/* readSamplesFloat (stream.d:429-637): interleaved, returns frames read (< frames: end or error). */
int         afg_read_samples_float(afg_stream *s, float *out, int frames);
/* canSeek / seekPosition / tellPosition (stream.d:352-369, :1095-1189, :1208-1261): positions are frames; a seek outside
 * [0, length] fails and leaves the position alone (the invariants of examples/transcode's additionalTests). */
int         afg_can_seek(const afg_stream *s);
int         afg_seek_position(afg_stream *s, int frame);     /* 1 = done, 0 = refused */
int         afg_tell_position(const afg_stream *s);          /* -1 on an invalid stream */
void        afg_close(afg_stream *s);

/* Host front-ends on their own (no device needed): what the stream and batch entry points run
 * before the device stage.  afg_flac_parse walks the native FLAC container (drflac.d:1901-2118),
 * every frame and subframe header (drflac.d:1444-1569) and the Rice residuals (drflac.d:1279-1328)
 * and returns transform-stage records; it stops at the first frame that does not parse, like the
 * reference's read loop (drflac.d:2860).  afg_qoa_parse validates the headers (qoa.d:413-486). */
typedef struct afg_flac_parsed {
    uint32_t sample_rate, channels, bps, max_block;
    uint64_t total_samples;        /* per channel, from STREAMINFO; 0 = unknown */
    uint64_t n_frames, n_subframes, n_res, out_samples;
    afg_flac_frame    *frames;
    afg_flac_subframe *subframes;
    int32_t           *res;
    void              *owner;      /* internal */
} afg_flac_parsed;

int  afg_flac_parse(const uint8_t *data, size_t length, afg_flac_parsed *out);  /* AFG_ERR_UNSUPPORTED: not FLAC */
void afg_flac_parsed_free(afg_flac_parsed *parsed);
/* frames may be NULL (count only); at most frame_cap records are written, *n_frames gets the total. */
int  afg_qoa_parse(const uint8_t *data, size_t length, uint32_t *channels, uint32_t *samplerate,
                   uint32_t *samples, afg_qoa_frame *frames, size_t frame_cap, size_t *n_frames);

/* MP3 (MPEG-1/2/2.5 Layer III) front-end on its own: frame sync, side info, scalefactors, Huffman +
 * requantisation, stereo processing, reorder and bit reservoir (minimp3.d:487-1000, :1170-1230, :1436-1581)
 * driven the way minimp3_ex does it (ID3/APE skipping, Xing/Info tag, delay/padding: minimp3_ex.d:93-190,
 * :566-639, :787-888).  Result: the records afg_mp3_transform_hip consumes -- one plan stream per run of
 * continuous decoder state -- and the copy plan that turns its PCM plane into what mp3dec_ex_read delivers. */
typedef struct afg_mp3_copy { uint64_t src_float, count; } afg_mp3_copy;
typedef struct afg_mp3_parsed {
    int32_t  channels, hz, tagged, start_delay;
    uint64_t detected_samples;     /* 0: delivery runs to the end of the data */
    uint64_t declared_samples;     /* mp3dec_ex_t.samples; AudioStream length = this / channels (stream.d:1737) */
    uint64_t pcm_samples;          /* floats the copy plan delivers */
    uint64_t n_runs, n_blocks, n_copies;
    uint32_t *run_granules;        /* [n_runs] granules per plan stream, all with `channels` channels */
    float    *coef;                /* n_blocks * 576 */
    uint32_t *flags;               /* n_blocks, AFG_MP3_FLAGS */
    afg_mp3_copy *copies;
    void     *owner;               /* internal */
} afg_mp3_parsed;

int  afg_mp3_parse(const uint8_t *data, size_t length, afg_mp3_parsed *out);   /* AFG_ERR_UNSUPPORTED: no Layer III stream */
void afg_mp3_parsed_free(afg_mp3_parsed *parsed);

/* The same front-end in quantised mode: `coef` stays NULL, the records of afg_mp3_requant_hip come back instead
 * (coef_off = q_off = 576 * first block of the granule). */
typedef struct afg_mp3_parsed_q {
    afg_mp3_parsed    base;          /* coef == NULL */
    uint64_t          n_granules, n_sdesc;
    int16_t          *q;             /* n_blocks * 576 */
    afg_mp3_qgranule *granules;
    afg_mp3_sdesc    *sdesc;
} afg_mp3_parsed_q;

int  afg_mp3_parse_q(const uint8_t *data, size_t length, afg_mp3_parsed_q *out);
void afg_mp3_parsed_q_free(afg_mp3_parsed_q *parsed);
/* The requantiser's tables, as the device holds them (24 scalefactor-band tables = 3 kinds x 8 rate rows): band of every
 * line, destination of every line under L3_reorder, and g_pow43 (minimp3.d:722-735).  Any pointer may be NULL. */
void afg_mp3_qtables(uint8_t band_of_line[24][576], uint16_t dst_of_src[24][576], float pow43[145]);

/* Ogg Vorbis I front-end on its own: Ogg pages and lacing, the three header packets (code books, floor 1,
 * residues 0/1/2, mappings, modes: stb_vorbis2.d:2669-3266) and every audio packet up to the transform seam
 * (floor decode, residue decode, inverse coupling, floor curve: :2354-2523), plus what the pull API delivers of
 * each packet's output (first frame primed only, last-page truncation: :2531-2596, :2606-2657) and the stream
 * length (:3797-3868).  Result: the inputs of afg_vorbis_plan_create / afg_vorbis_transform_hip for one stream. */
typedef struct afg_vorbis_parsed {
    int32_t  channels, blocksize0, blocksize1;
    uint32_t sample_rate;
    uint32_t total_samples;        /* 0 = unknown */
    uint64_t n_packets, spec_floats, pcm_frames;
    uint8_t *pflags;               /* [n_packets] */
    float   *spec;                 /* per packet [channel][n/2] */
    int32_t *take_from, *take_count;   /* frames [take_from, take_from + take_count) of packet p's output are delivered */
    void    *owner;                /* internal */
} afg_vorbis_parsed;

int  afg_vorbis_parse(const uint8_t *data, size_t length, afg_vorbis_parsed *out);   /* AFG_ERR_UNSUPPORTED: not Ogg Vorbis */
void afg_vorbis_parsed_free(afg_vorbis_parsed *parsed);
/* The same with the tail of the packet decode left to the device: base.spec holds the residue vectors and the records
 * below are the inputs of afg_vorbis_floor_hip (spec_off counted from base.spec; steps per mapping, shared by its packets). */
typedef struct afg_vorbis_parsed_r {
    afg_vorbis_parsed base;
    uint64_t n_curves, n_points, n_steps;
    afg_vorbis_floor_packet *packets;  /* [base.n_packets] */
    afg_vorbis_floor_curve  *curves;   /* [n_curves] = one per packet-channel */
    int32_t *points;                   /* [2 * n_points] */
    uint8_t *steps;                    /* [2 * n_steps] */
} afg_vorbis_parsed_r;

int  afg_vorbis_parse_r(const uint8_t *data, size_t length, afg_vorbis_parsed_r *out);
void afg_vorbis_parsed_r_free(afg_vorbis_parsed_r *parsed);

/* Ogg Opus front-end on its own, CELT-only packets: Ogg pages, OpusHead / OpusTags (dopus.d:7791-7829, :8120-8193; output
 * gain :1311-1316 with R128_TRACK_GAIN :8011-8059), packet framing (ff_opus_parse_packet, :1081-1258), the range decoder
 * (:809-1034) and the CELT frame decoder up to the transform seam (:2128-3678).  Result: the inputs of
 * afg_celt_transform_hip for one stream.  frames[i] is channel 0's record of frame i, addressed for interleaved output
 * (out_off = first sample * channels, out_stride = channels); channel c of the same frame reads coef_off + c * frame_size
 * and writes out_off + c.  AFG_ERR_UNSUPPORTED: not an Ogg Opus stream the reference opens (no fields set), or one that
 * holds SILK / hybrid packets, which this front-end does not decode (channels != 0 then; afg_last_error says which). */
typedef struct afg_opus_parsed {
    int32_t  channels, preskip;
    int32_t  gain_i;               /* header gain + R128_TRACK_GAIN, Q7.8 dB; 0: the decoder does not scale */
    int32_t  error;                /* 1: a packet failed to frame; the records end there and the reference's read reports an error */
    float    gain;                 /* 10^(gain_i / 5120) as a float: afg_opus_output_gain_hip's factor */
    int32_t  pad;
    int64_t  declared_frames;      /* last page's granule position - preskip (dopus.d:8159): AudioStream's length */
    uint64_t pcm_frames;           /* frames the records decode to (the reference delivers min(pcm_frames, declared_frames)) */
    uint64_t n_frames, n_coeffs;
    afg_celt_frame *frames;
    float   *coeffs;
    void    *owner;                /* internal */
} afg_opus_parsed;

int  afg_opus_parse(const uint8_t *data, size_t length, afg_opus_parsed *out);
void afg_opus_parsed_free(afg_opus_parsed *parsed);

/* Batch decode (no reference counterpart: the throughput path).  Files are parsed by n_threads
 * pooled host threads (0 = one per physical core: half the logical CPUs of an SMT host) straight
 * into page-locked staging, restored on the current device chunk by chunk with upload, kernel and
 * download overlapped, and returned as interleaved float PCM owned by the result. */
typedef struct afg_batch_item {
    int         status;        /* afg_status of this file: a bad file never poisons the batch */
    const char *message;       /* static string, NULL when ok */
    int         format;        /* afg_format */
    int         channels;
    float       samplerate;
    int64_t     frames;
    float      *pcm;           /* frames * channels floats, NULL on error */
} afg_batch_item;

typedef struct afg_batch_result {
    int             n_files;
    afg_batch_item *items;
    void           *owner;     /* internal */
} afg_batch_result;

int  afg_batch_decode(const uint8_t *const *data, const size_t *length, int n_files, int n_threads,
                      afg_batch_result *out);
void afg_batch_free(afg_batch_result *result);

/* Device selection (files are independent -- stream.d:1363-1434 is all per-instance -- so a batch shards
 * by file across the GPUs of a node, with no exchange between devices).
 *   afg_set_device   makes `device` current for the calling host thread (HIP's current device is per thread) and
 *                    checks that it is a gfx950; plans, *_hip entries, afg_open_from_memory and afg_batch_decode
 *                    all work on the calling thread's current device.
 *   afg_batch_decode_ex  the batch entry with options: n_devices = 0 runs on the current device (what
 *                    afg_batch_decode does), n_devices = -1 on every visible device, n_devices = k > 0 on
 *                    devices[0..k) (NULL: 0..k-1; a device may be named more than once).  Files are assigned
 *                    longest-first to the least loaded device; every device gets its own host thread, helper
 *                    threads (n_threads in total, 0 = one per physical core) and stream set.  Per-file results do
 *                    not depend on the number of devices. */
typedef struct afg_batch_opts {
    uint32_t   struct_size;    /* sizeof(afg_batch_opts) */
    int        n_threads;
    int        n_devices;
    const int *devices;
} afg_batch_opts;

int  afg_set_device(int device);
int  afg_get_device(void);                 /* current device of the calling thread, < 0: afg_status */
int  afg_batch_decode_ex(const uint8_t *const *data, const size_t *length, int n_files, const afg_batch_opts *opts,
                         afg_batch_result *out);
/* Page-locked staging buffers and the device planes of the batch path are pooled between batch calls (pinning costs about
 * as much as the transfer; freed device memory is wiped by the copy engines the next call's transfers need): this
 * releases every pooled buffer -- host and device -- that is not in use and returns the bytes freed. */
uint64_t afg_host_pool_trim(void);

/* ========================================================================== *
 *  Utilities used by the host mirror, the tests and bench.py
 * ========================================================================== */
int afg_device_malloc(void **d_ptr, size_t bytes);
int afg_device_free(void *d_ptr);
int afg_memcpy_h2d(void *d_dst, const void *src, size_t bytes, void *hip_stream);
int afg_memcpy_d2h(void *dst, const void *d_src, size_t bytes, void *hip_stream);
int afg_stream_synchronize(void *hip_stream);
/* ========================================================================== *
 *  Output side: what `transcode` needs after the decode.
 *
 *  QOA encoder: replaces qoa_encode_frame (qoa.d:295-399) and the framing of QOAEncoder (qoa.d:538-700).
 *  The LMS state of the encoder runs through a whole stream, so a stream's channels are serial; the brute-force
 *  search over the 16 scalefactors of every slice runs on 16 lanes.  Output bytes are the reference's.
 * ========================================================================== */

typedef struct afg_qoa_enc_stream {
    uint64_t pcm_off;      /* index of the stream's first sample in the interleaved input plane (int16 or float) */
    uint64_t out_off;      /* byte offset of the stream's file in the output plane (multiple of 8) */
    uint32_t samples;      /* frames (samples per channel) */
    uint32_t samplerate;   /* 1 .. 0xffffff (QOAEncoder.initialize rejects others, qoa.d:592) */
    uint8_t  channels;     /* 1 .. 8 */
    uint8_t  pad[7];
} afg_qoa_enc_stream;      /* 32 bytes */

/* Size in bytes of the QOA file for `samples` frames of `channels` channels (file header + frames). */
uint64_t afg_qoa_encoded_size(uint32_t samples, uint32_t channels);

/* Exactly one of d_pcm_i16 / d_pcm_f32 is given; floats are converted as QOAEncoder.writeSamples does
 * (qoa.d:632-636: (int)(32768.5 + x * 32767.0) - 32768, |x| <= 1).  Descriptors must respect the ranges above. */
int afg_qoa_encode_hip(uint32_t n_streams, const afg_qoa_enc_stream *d_streams, const int16_t *d_pcm_i16,
                       const float *d_pcm_f32, uint8_t *d_out, void *hip_stream);

/* WAV writer (wav.d:365-701, host only): 44-byte RIFF/WAVE header ('fmt ' of 16 bytes, tag 1 for PCM, 3 for IEEE
 * float) followed by the samples; PCM conversions are the reference's (wav.d:482-527).  afg_wav_encode is
 * EncodingOptions.enableDither = false; afg_wav_encode_dithered applies TPDFDither.process (wav.d:674-701) to the
 * integer formats first, as WAVEncoder.writeSamples does by default (stream.d:66): per sample, in order, two draws
 * rng(user) / rng_max.  rng = NULL is the reference's generator, libc rand() / RAND_MAX (rng_max ignored). */
#define AFG_WAV_S8     0
#define AFG_WAV_S16LE  1
#define AFG_WAV_S24LE  2
#define AFG_WAV_FP32LE 3
#define AFG_WAV_FP64LE 4
uint64_t afg_wav_encoded_size(uint64_t frames, uint32_t channels, int format);          /* 0: bad arguments */
/* Writes the whole file into `out` (capacity `cap`); returns the bytes written, 0 on bad arguments / short buffer. */
uint64_t afg_wav_encode(const float *samples, uint64_t frames, uint32_t channels, uint32_t samplerate, int format,
                        uint8_t *out, uint64_t cap);
typedef int (*afg_rand_fn)(void *user);        /* a draw in [0, rng_max] */
uint64_t afg_wav_encode_dithered(const float *samples, uint64_t frames, uint32_t channels, uint32_t samplerate, int format,
                                 afg_rand_fn rng, void *rng_user, uint32_t rng_max, uint8_t *out, uint64_t cap);

/* Streaming device-to-device copy (16-byte aligned) used by bench.py to measure the copy rate this device
 * actually sustains, the practical ceiling the HBM-bound kernels are compared with next to the 8 TB/s spec. */
int afg_copy_probe_hip(void *d_dst, const void *d_src, size_t bytes, void *hip_stream);

/* Test aid: fills the LDS of every compute unit with `word` (workgroups of 160 KiB, many times the CU count).  LDS is not
 * cleared between kernels, so a kernel that reads a location it never wrote normally finds zeros or old finite data and
 * passes; with 0x7fc00000 (NaN) behind it such a read shows.  The GPU tests run every case behind this. */
int afg_lds_fill_probe_hip(uint32_t word, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* AFG_H */
