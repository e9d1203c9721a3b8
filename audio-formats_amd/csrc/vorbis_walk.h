// vorbis_walk.h -- entry points of the tolerance-mode Vorbis walk (vorbis_walk.hip) used by the plan code in
// vorbis_transform.hip.
#pragma once
#include "vorbis_core.h"

namespace afg_vorbis {

constexpr int kWalkShapes = 12;
// The walk's kernel for a stream with blocksize_1 in {1024, 2048, 4096} and blocksize_0 <= 512 (or = blocksize_1); with
// size = log2(blocksize_1) - 10:  2 size + (channels - 1) for mono / stereo streams (one wavefront walks every channel of
// a segment); 6 + size for 3, 5 or 7 channels (a workgroup per segment, one wavefront per channel), 9 + size for an even
// number up to 16 (a workgroup per segment, one wavefront per pair of channels); -1 for every other stream.
int walk_shape(int channels, int blocksize0, int blocksize1);
int walk_shape_channels(int shape);          // channels one wavefront of the shape walks
int walk_shape_blocksize(int shape);
size_t walk_table_floats(int shape);
// dst: walk_table_floats(shape) floats; window: the window of the shape's blocksize_1 as stb_vorbis2.d:866-873 builds it
void walk_build_tables(int shape, float *dst, const float *window);
// Segments of streams of one shape -- for the shapes with more than two channels: of ONE channel count, `nch` (else 0) --;
// `out` 16-byte aligned; *counter zeroed on `stream`.
int walk_launch(int shape, const VorbisSeg *segs, uint32_t n_segs, int nch, const VorbisStream *streams, const uint8_t *pflags,
                const uint64_t *spec_off, const uint64_t *out_off, const float *tables, const float *walk_tables,
                const float *spec, float *out, uint32_t *counter, hipStream_t stream);

}  // namespace afg_vorbis
