// afg_vorbis_front.h -- host front-end for Ogg Vorbis I files (see afg_vorbis_front.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/afg.h"

namespace afg_vorbis {

struct File {
    int channels = 0, blocksize0 = 0, blocksize1 = 0;
    uint32_t sample_rate = 0;
    uint32_t total_samples = 0;            // stb_vorbis_stream_length_in_samples: granule of the last page, 0 = unknown
    std::vector<uint8_t> pflags;           // one per decoded audio packet (AFG_VORBIS_LONG | PREV | NEXT)
    std::vector<float> spec;               // per packet [channel][n/2]: floor-multiplied, uncoupled spectra
    // Optional external destination (the batch path parses straight into a page-locked staging buffer): when set,
    // `spec` stays empty; `overflow` reports that `ext_cap` floats were not enough.
    float *ext_spec = nullptr;
    size_t ext_cap = 0, n_spec = 0;        // n_spec: floats recorded (either destination)
    bool overflow = false;
    const float *spectra() const { return ext_spec ? ext_spec : spec.data(); }
    std::vector<int32_t> take_from, take_count;   // frames of packet p's (right_start - left_start) output that are delivered
    uint64_t pcm_frames = 0;               // sum of take_count
    // device_floor (set before parsing / opening): the packet decode stops after the residues -- `spec` holds residue
    // vectors, and inverse coupling, silent channels and the floor curves are left to afg_vorbis_floor_hip, described
    // by the records below (spec_off counted from the file's first float; steps per mapping, filled when the stream opens).
    bool device_floor = false;
    std::vector<afg_vorbis_floor_packet> fl_packets;   // one per recorded packet
    std::vector<afg_vorbis_floor_curve> fl_curves;     // one per packet-channel
    std::vector<int32_t> fl_points;                    // (x, y) pairs
    std::vector<uint8_t> fl_steps;                     // (magnitude, angle) pairs in the order applied, mapping after mapping
};

bool parse_file(const uint8_t *data, size_t size, File &out, bool device_floor = false);   // false: not an Ogg Vorbis I stream the reference accepts
bool parse_file_into(const uint8_t *data, size_t size, File &out, float *spec_dst, size_t cap, bool device_floor = false);
size_t max_spec_floats(const uint8_t *data, size_t size);          // 0: parse_file would return false

// Chunked reading for the AudioStream surface (stream.d:429-637: the reference decodes as the caller pulls): `open`
// parses the three headers and the stream length, `more` decodes the next audio packets into fresh records.  From the
// second chunk on the records start with the previous chunk's last packet (nothing to take from it): the transform
// stage re-derives the overlap from its spectrum, so no device state is carried between chunks.
class Reader {
public:
    Reader();
    ~Reader();
    Reader(const Reader &) = delete;
    Reader &operator=(const Reader &) = delete;
    bool open(const uint8_t *data, size_t size, File &meta, bool device_floor = false);
    bool more(File &out, int max_packets);     // false: the stream has ended (out holds nothing to deliver)
private:
    struct Impl;
    Impl *p;
};

}  // namespace afg_vorbis
