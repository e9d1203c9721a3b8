// vorbis_walk.h -- entry points of the tolerance-mode Vorbis walk (vorbis_walk.hip) used by the plan code in
// vorbis_transform.hip.
#pragma once
#include "vorbis_core.h"

namespace afg_vorbis {

size_t walk_table_floats();
// dst: walk_table_floats() floats; window2048: the n = 2048 window as stb_vorbis2.d:866-873 builds it (1024 floats)
void walk_build_tables(float *dst, const float *window2048);
uint32_t walk_waves_per_group();
// Stereo segments of streams with blocksize_1 = 2048 and blocksize_0 <= 512; `out` 16-byte aligned; *counter zeroed on `stream`.
int walk_launch(const VorbisSeg *segs, uint32_t n_segs, const VorbisStream *streams, const uint8_t *pflags,
                const uint64_t *spec_off, const uint64_t *out_off, const float *tables, const float *walk_tables,
                const float *spec, float *out, uint32_t *counter, uint32_t groups, hipStream_t stream);

}  // namespace afg_vorbis
