/// afgpu.d -- D binding of the MI355X transform-stage library (include/afg.h, ABI version 1).
///
/// Source only: no D compiler exists in the build image, so this file is exercised through the
/// same C symbols from C/Python harnesses instead.  Every prototype carries `nothrow @nogc`, which is
/// what makes it callable from audio-formats' `nothrow @nogc` decoder modules
/// (source/audioformats/minimp3.d:16-17, stb_vorbis2.d:110-111, drflac.d:21).
module audioformats.afgpu;

nothrow @nogc extern (C):

enum AFG_ABI_VERSION = 2;

enum afg_status : int
{
    AFG_OK = 0,
    AFG_ERR_INVALID = -1,
    AFG_ERR_NO_DEVICE = -2,
    AFG_ERR_HIP = -3,
    AFG_ERR_OOM = -4,
    AFG_ERR_UNSUPPORTED = -5,
}

int afg_abi_version();
const(char)* afg_status_string(int status);
const(char)* afg_last_error();
int afg_device_count();

/// Numeric mode of the float transform stages (afg.h): exact = the reference's expression trees bit for bit; tolerance
/// (default) = within 1e-5 RMS, which lets the Opus/CELT stage re-associate, the Vorbis stage use one radix-8 FFT per block
/// and the MP3 stage fuse its multiply-adds (round 4).  AFG_NUMERIC_FROM_ENV hands the choice back to
/// the environment variable AFG_NUMERIC.  Returns the mode in effect before.
enum AFG_NUMERIC_FROM_ENV = -1, AFG_NUMERIC_EXACT = 0, AFG_NUMERIC_TOLERANCE = 1;
int afg_set_numeric_mode(int mode);
int afg_dev_option(const(char)* name, int value);
int afg_get_numeric_mode();
int afg_device_name(int device, char* buf, size_t buflen);

// ---- MP3 (replaces minimp3.d:1226-1228 + :1553) ------------------------------------------
struct afg_mp3_plan;
enum AFG_MP3_STATE_FLOATS = 1536;

uint AFG_MP3_FLAGS(uint block_type, uint n_long_bands, int aa_bands) pure
{
    return block_type | (n_long_bands << 8) | (cast(uint)(aa_bands + 1) << 16);
}

// optional, OR-ed into the flag word: only the first `bands` subbands (0..32) may hold lines that are not +0.0
uint AFG_MP3_NZ_BANDS(uint bands) pure { return (bands + 1) << 24; }
enum uint AFG_MP3_SUBBAND = 0x80000000;   // the block holds Layer I / II subband samples (band * 18 + slot): synthesis only

int afg_mp3_plan_create(afg_mp3_plan** plan, uint n_streams, const(uint)* granules,
                        const(ubyte)* channels, uint seg_granules);
void afg_mp3_plan_destroy(afg_mp3_plan* plan);
ulong afg_mp3_plan_blocks(const(afg_mp3_plan)* plan);
uint afg_mp3_plan_segments(const(afg_mp3_plan)* plan);
int afg_mp3_transform_hip(const(afg_mp3_plan)* plan, const(float)* d_coef, const(uint)* d_flags,
                          float* d_pcm, float* d_state, void* hip_stream);

// ---- Vorbis (replaces stb_vorbis2.d:2526-2527 + :2606-2657 + :3927-3952) ---------------------
struct afg_vorbis_plan;
enum AFG_VORBIS_LONG = 1u, AFG_VORBIS_PREV = 2u, AFG_VORBIS_NEXT = 4u;
/// optional, long packets: only the first e eighths of the spectrum can be nonzero (residue end, stb_vorbis2.d:1586-1600)
uint AFG_VORBIS_NZ_EIGHTHS(uint e) { return (e + 1u) << 4; }

int afg_vorbis_plan_create(afg_vorbis_plan** plan, uint n_streams, const(uint)* packets,
                           const(ubyte)* channels, const(ushort)* blocksize0,
                           const(ushort)* blocksize1, const(ubyte)* pflags, uint seg_packets);
void afg_vorbis_plan_destroy(afg_vorbis_plan* plan);
ulong afg_vorbis_plan_packets(const(afg_vorbis_plan)* plan);
ulong afg_vorbis_plan_spec_floats(const(afg_vorbis_plan)* plan);
ulong afg_vorbis_plan_out_floats(const(afg_vorbis_plan)* plan);
int afg_vorbis_plan_offsets(const(afg_vorbis_plan)* plan, ulong* spec_off, ulong* out_off);
int afg_vorbis_transform_hip(const(afg_vorbis_plan)* plan, const(float)* d_spec, float* d_out,
                             void* hip_stream);

// inverse coupling + floor curves on the device (replaces stb_vorbis2.d:2493-2523 with do_floor :2255-2284 / draw_line :1534-1563)
struct afg_vorbis_floor_packet
{
    ulong spec_off;
    uint n2, channels, curve_index, step_off, n_steps, pad;
}
static assert(afg_vorbis_floor_packet.sizeof == 32);
struct afg_vorbis_floor_curve { uint point_off, n_points; }
int afg_vorbis_floor_hip(ulong n_packets, const(afg_vorbis_floor_packet)* d_packets, const(afg_vorbis_floor_curve)* d_curves,
                         const(int)* d_points, const(ubyte)* d_steps, float* d_spec, void* hip_stream);

// ---- FLAC (replaces drflac.d:1235 prediction half + :2885-2941, optionally stream.d:505-511) --
enum AFG_FLAC_INDEPENDENT = 0, AFG_FLAC_LEFT_SIDE = 8, AFG_FLAC_RIGHT_SIDE = 9, AFG_FLAC_MID_SIDE = 10;

struct afg_flac_subframe
{
    short[32] coef;
    ubyte order;
    ubyte shift;
    ubyte wasted;
    ubyte use64;
}
static assert(afg_flac_subframe.sizeof == 68);

struct afg_flac_frame
{
    ulong in_off;
    ulong out_off;
    uint block_size;
    uint sf_index;
    ubyte channels;
    ubyte assignment;
    ubyte bps;
    ubyte res16;          // 1: the frame's residual rows are int16 (in_off counts int16 elements, rows of AFG_FLAC_ROW16)
    ubyte[4] pad;
}
static assert(afg_flac_frame.sizeof == 32);
ulong AFG_FLAC_ROW16(ulong block_size) { return (block_size + 7) & ~7UL; }

int afg_flac_transform_hip(ulong n_frames, const(afg_flac_frame)* d_frames,
                           const(afg_flac_subframe)* d_subframes, const(int)* d_res,
                           int* d_out_i32, float* d_out_f32, void* hip_stream);
uint afg_flac_variants(ulong n_frames, const(afg_flac_frame)* frames, const(afg_flac_subframe)* subframes);   // host records
int afg_flac_transform_variants_hip(ulong n_frames, const(afg_flac_frame)* d_frames,
                                    const(afg_flac_subframe)* d_subframes, const(int)* d_res,
                                    int* d_out_i32, float* d_out_f32, uint variants, void* hip_stream);

// ---- QOA (replaces the slice loop of qoa_decode_frame, qoa.d:489-530, and qoa.d:831-838) --------
struct afg_qoa_frame
{
    ulong byte_off;
    ulong out_off;
    ushort samples;
    ubyte channels;
    ubyte[5] pad;
}
static assert(afg_qoa_frame.sizeof == 24);

int afg_qoa_transform_hip(ulong n_frames, const(afg_qoa_frame)* d_frames, const(ubyte)* d_bytes,
                          short* d_out_i16, float* d_out_f32, void* hip_stream);

// OpusFile.readFrame's Float2IntScaled + saturation (dopus.d:7923-7926, :8098-8105) and stream.d:480
int afg_opus_output_hip(ulong n_samples, const(float)* d_in, short* d_out_i16, float* d_out_f32, void* hip_stream);
int afg_opus_output_gain_hip(ulong n_samples, const(float)* d_in, float gain, short* d_out_i16, float* d_out_f32, void* hip_stream);

// ---- output side: QOA encoder (replaces qoa_encode_frame, qoa.d:295-399, and QOAEncoder's framing, :538-700) and
//      the WAV writer (WAVEncoder, wav.d:365-701; host only) ----
struct afg_qoa_enc_stream
{
    ulong pcm_off;
    ulong out_off;
    uint samples;
    uint samplerate;
    ubyte channels;
    ubyte[7] pad;
}
static assert(afg_qoa_enc_stream.sizeof == 32);

ulong afg_qoa_encoded_size(uint samples, uint channels);
int afg_qoa_encode_hip(uint n_streams, const(afg_qoa_enc_stream)* d_streams, const(short)* d_pcm_i16,
                       const(float)* d_pcm_f32, ubyte* d_out, void* hip_stream);

enum { AFG_WAV_S8 = 0, AFG_WAV_S16LE = 1, AFG_WAV_S24LE = 2, AFG_WAV_FP32LE = 3, AFG_WAV_FP64LE = 4 }
ulong afg_wav_encoded_size(ulong frames, uint channels, int format);
ulong afg_wav_encode(const(float)* samples, ulong frames, uint channels, uint samplerate, int format,
                     ubyte* out_, ulong cap);

// ---- Opus/CELT (replaces the per-channel tail of ff_celt_decode_frame, dopus.d:3680-3702) -------
struct afg_celt_frame
{
    ulong coef_off;
    ulong out_off;
    uint out_stride;
    ushort frame_size;
    ubyte blocks;
    ubyte pad;
    int pf_period_new;
    float[3] pf_gains_new;
    float imdct_scale;
    uint pad2;
}
static assert(afg_celt_frame.sizeof == 48);
enum AFG_CELT_STATE_FLOATS = 2064;

int afg_celt_transform_hip(uint n_chan, const(ulong)* d_rec_base, const(afg_celt_frame)* d_recs,
                           const(float)* d_coeffs, float* d_out, float* d_states, void* hip_stream);

// ---- utilities ----------------------------------------------------------------------------------
int afg_device_malloc(void** d_ptr, size_t bytes);
int afg_device_free(void* d_ptr);
int afg_memcpy_h2d(void* d_dst, const(void)* src, size_t bytes, void* hip_stream);
int afg_memcpy_d2h(void* dst, const(void)* d_src, size_t bytes, void* hip_stream);
int afg_stream_synchronize(void* hip_stream);
int afg_copy_probe_hip(void* d_dst, const(void)* d_src, size_t bytes, void* hip_stream);

// ---- outer surface: the reading half of AudioStream (stream.d:102-637) ----------------------------
enum afg_format : int   // AudioFileFormat, stream.d:36-47 (same order)
{
    wav = 0, mp3 = 1, flac = 2, ogg = 3, opus = 4, qoa = 5, mod = 6, xm = 7, unknown = 8,
}
enum long AFG_UNKNOWN_LENGTH = -1;   // audiostreamUnknownLength, stream.d:90

struct afg_stream;
afg_stream* afg_open_from_memory(const(ubyte)* data, size_t length);
int afg_is_error(const(afg_stream)* s);
const(char)* afg_error_message(const(afg_stream)* s);
int afg_get_format(const(afg_stream)* s);
int afg_get_num_channels(const(afg_stream)* s);
long afg_get_length_in_frames(const(afg_stream)* s);
float afg_get_samplerate(const(afg_stream)* s);
int afg_read_samples_float(afg_stream* s, float* outData, int frames);
int afg_can_seek(const(afg_stream)* s);
int afg_seek_position(afg_stream* s, int frame);
int afg_tell_position(const(afg_stream)* s);
void afg_close(afg_stream* s);

struct afg_flac_parsed
{
    uint sample_rate, channels, bps, max_block;
    ulong total_samples;
    ulong n_frames, n_subframes, n_res, out_samples;
    afg_flac_frame* frames;
    afg_flac_subframe* subframes;
    int* res;
    void* owner;
}
int afg_flac_parse(const(ubyte)* data, size_t length, afg_flac_parsed* parsed);
void afg_flac_parsed_free(afg_flac_parsed* parsed);
int afg_qoa_parse(const(ubyte)* data, size_t length, uint* channels, uint* samplerate, uint* samples,
                  afg_qoa_frame* frames, size_t frame_cap, size_t* n_frames);

struct afg_mp3_copy { ulong src_float, count; }
struct afg_mp3_parsed
{
    int channels, hz, tagged, start_delay;
    ulong detected_samples, declared_samples, pcm_samples;
    ulong n_runs, n_blocks, n_copies;
    uint* run_granules;
    float* coef;
    uint* flags;
    afg_mp3_copy* copies;
    void* owner;
}
int afg_mp3_parse(const(ubyte)* data, size_t length, afg_mp3_parsed* parsed);
void afg_mp3_parsed_free(afg_mp3_parsed* parsed);

struct afg_vorbis_parsed
{
    int channels, blocksize0, blocksize1;
    uint sample_rate, total_samples;
    ulong n_packets, spec_floats, pcm_frames;
    ubyte* pflags;
    float* spec;
    int* take_from;
    int* take_count;
    void* owner;
}
int afg_vorbis_parse(const(ubyte)* data, size_t length, afg_vorbis_parsed* parsed);
void afg_vorbis_parsed_free(afg_vorbis_parsed* parsed);
struct afg_vorbis_parsed_r
{
    afg_vorbis_parsed base;          // base.spec: residue vectors
    ulong n_curves, n_points, n_steps;
    afg_vorbis_floor_packet* packets;
    afg_vorbis_floor_curve* curves;
    int* points;
    ubyte* steps;
}
int afg_vorbis_parse_r(const(ubyte)* data, size_t length, afg_vorbis_parsed_r* parsed);
void afg_vorbis_parsed_r_free(afg_vorbis_parsed_r* parsed);

struct afg_opus_parsed
{
    int channels, preskip, gain_i, error;
    float gain;
    int pad;
    long declared_frames;
    ulong pcm_frames, n_frames, n_coeffs;
    afg_celt_frame* frames;
    float* coeffs;
    void* owner;
}
int afg_opus_parse(const(ubyte)* data, size_t length, afg_opus_parsed* parsed);
void afg_opus_parsed_free(afg_opus_parsed* parsed);

struct afg_batch_item
{
    int status;
    const(char)* message;
    int format;
    int channels;
    float samplerate;
    long frames;
    float* pcm;
}
struct afg_batch_result
{
    int n_files;
    afg_batch_item* items;
    void* owner;
}
int afg_batch_decode(const(ubyte*)* data, const(size_t)* length, int n_files, int n_threads, afg_batch_result* result);

// ---- round 2: device selection, multi-device batches, MP3 quantised upload, CELT on two streams, dithered WAV ----
struct afg_batch_opts { uint struct_size; int n_threads; int n_devices; const(int)* devices; }
int afg_set_device(int device);
int afg_get_device();
int afg_batch_decode_ex(const(ubyte*)* data, const(size_t)* length, int n_files, const(afg_batch_opts)* opts, afg_batch_result* result);
ulong afg_host_pool_trim();

enum uint AFG_MP3_NO_SDESC = 0xffffffffu;
struct afg_mp3_qgranule { ulong q_off; ulong coef_off; uint sdesc; ubyte nch; ubyte stereo; ubyte[2] table; float[40][2] scale; }
struct afg_mp3_sdesc { ubyte[40] type; float[40] fl; float[40] fr; }
static assert(afg_mp3_qgranule.sizeof == 344 && afg_mp3_sdesc.sizeof == 360);
int afg_mp3_requant_hip(ulong n_granules, const(afg_mp3_qgranule)* d_granules, const(short)* d_q,
                        const(afg_mp3_sdesc)* d_sdesc, float* d_coef, void* hip_stream);

struct afg_mp3_parsed_q
{
    afg_mp3_parsed base;             // base.coef is null
    ulong n_granules, n_sdesc;
    short* q;                        // n_blocks * 576 quantised lines
    afg_mp3_qgranule* granules;
    afg_mp3_sdesc* sdesc;
}
int afg_mp3_parse_q(const(ubyte)* data, size_t length, afg_mp3_parsed_q* parsed);
void afg_mp3_parsed_q_free(afg_mp3_parsed_q* parsed);
void afg_mp3_qtables(ubyte* band_of_line /* [24][576] */, ushort* dst_of_src /* [24][576] */, float* pow43 /* [145] */);
int afg_lds_fill_probe_hip(uint word, void* hip_stream);   // test aid: fills the LDS of every CU with `word`
int afg_celt_transform_streams_hip(uint n_chan, const(ulong)* d_rec_base, const(afg_celt_frame)* d_recs, const(float)* d_coeffs,
                                   float* d_out, float* d_states, void* hip_stream, void* hip_tail_stream);
alias afg_rand_fn = extern(C) int function(void* user) nothrow @nogc;
ulong afg_wav_encode_dithered(const(float)* samples, ulong frames, uint channels, uint samplerate, int format,
                              afg_rand_fn rng, void* rng_user, uint rng_max, ubyte* outData, ulong cap);
void afg_batch_free(afg_batch_result* result);

/// Drop-in for the decoding use of `AudioStream` (stream.d:102): same member names and error-state contract
/// (never throws, `isError` + `errorMessage`), backed by the device library.  Not thread-safe per instance,
/// like the original (stream.d:31-33).
struct GpuAudioStream
{
nothrow @nogc:
    private afg_stream* _h;

    @disable this(this);
    ~this() { cleanUp(); }

    void openFromMemory(const(ubyte)[] inputData)
    {
        cleanUp();
        _h = afg_open_from_memory(inputData.ptr, inputData.length);
    }
    void cleanUp() { if (_h !is null) { afg_close(_h); _h = null; } }

    bool isError() { return afg_is_error(_h) != 0; }
    bool isValid() { return !isError(); }
    const(char)[] errorMessage()
    {
        import core.stdc.string : strlen;
        const(char)* m = afg_error_message(_h);
        return m is null ? null : m[0 .. strlen(m)];
    }
    afg_format getFormat() { return cast(afg_format) afg_get_format(_h); }
    int getNumChannels() { return afg_get_num_channels(_h); }
    long getLengthInFrames() { return afg_get_length_in_frames(_h); }
    float getSamplerate() { return afg_get_samplerate(_h); }
    bool canSeek() { return afg_can_seek(_h) != 0; }
    bool seekPosition(int frame) { return afg_seek_position(_h, frame) != 0; }
    int tellPosition() { return afg_tell_position(_h); }
    int readSamplesFloat(float* outData, int frames) { return afg_read_samples_float(_h, outData, frames); }
    int readSamplesFloat(float[] outData)
    {
        const int ch = getNumChannels();
        return ch > 0 ? readSamplesFloat(outData.ptr, cast(int)(outData.length / ch)) : 0;
    }
}
