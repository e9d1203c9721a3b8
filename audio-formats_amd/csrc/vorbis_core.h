// vorbis_core.h -- device code shared by the Vorbis transform kernels (vorbis_transform.hip: the bit-exact paths;
// vorbis_walk.hip: the tolerance-mode walk): segment / stream records, window bounds (stb_vorbis2.d:2333-2349) and the
// reference-ordered inverse MDCT over LDS (stb_vorbis2.d:1941-2242) that both use for short blocks.
#pragma once
#include "afg_common.h"

namespace afg_vorbis {

struct VorbisSeg {
    uint32_t stream;
    uint32_t p0;       // first packet (stream-relative)
    uint32_t count;
    uint32_t pad;      // wave path: channel walked by this wavefront
};

struct VorbisStream {
    uint64_t pkt_base;     // index of the stream's first packet in the batch-wide arrays
    uint32_t npkt;
    uint32_t nch;
    uint32_t bs[2];
    uint32_t tab[2];       // float offset of each blocksize's table set: A[n/2] B[n/2] C[n/4] W[n/2]
};

__device__ __forceinline__ void bfly(float *p, float *q, float c0, float c1)
{
    float d0 = p[0] - q[0];
    float d1 = p[-1] - q[-1];
    p[0] = p[0] + q[0];
    p[-1] = p[-1] + q[-1];
    q[0] = d0 * c0 - d1 * c1;
    q[-1] = d1 * c0 + d0 * c1;
}

// stb_vorbis2.d:1866-1896
__device__ __forceinline__ void iter_54(float *z)
{
    float k00 = z[0] - z[-4];
    float y0 = z[0] + z[-4];
    float y2 = z[-2] + z[-6];
    float k22 = z[-2] - z[-6];
    z[0] = y0 + y2;
    z[-2] = y0 - y2;
    float k33 = z[-3] - z[-7];
    z[-4] = k00 + k33;
    z[-6] = k00 - k33;
    float k11 = z[-1] - z[-5];
    float y1 = z[-1] + z[-5];
    float y3 = z[-3] + z[-7];
    z[-1] = y1 + y3;
    z[-3] = y1 - y3;
    z[-5] = k11 - k22;
    z[-7] = k11 + k22;
}

// In-place inverse MDCT of one channel held in LDS; buffer[0..n/2) spectrum in,
// buffer[0..n) samples out; buf2 = n/2 floats of scratch.  stb_vorbis2.d:1941-2242.
// A workgroup of one wavefront -- or wavefronts that never share data -- only needs its own LDS
// accesses ordered, which the hardware does in program order: a compiler-level ordering point is
// enough, and unlike __syncthreads() it does not drain the spectrum loads that are in flight.
// That point has to be a fence: wave_barrier alone is declared to touch no memory, so the compiler may move or forward LDS
// accesses across it.  A wavefront-scope fence on the LDS address space emits no instruction and no wait.
template <int kThreads>
__device__ __forceinline__ void pass_sync()
{
    if (kThreads > 64) __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int kThreads>
__device__ void inverse_mdct_lds(float *buffer, float *buf2, int n, int ld,
                                 const float *__restrict__ A, const float *__restrict__ B,
                                 const float *__restrict__ C)
{
    const int tid = threadIdx.x & (kThreads - 1);
    const int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
    float *u = buffer, *v = buf2;

    // copy-and-reflect + step 0, :1972-1994 (n/4 items)
    for (int it = tid; it < n4; it += kThreads) {
        if (it < n8) {
            const float *e = buffer + 4 * it;
            float *d = buf2 + n2 - 2 - 2 * it;
            const float *AA = A + 2 * it;
            d[1] = (e[0] * AA[0] - e[2] * AA[1]);
            d[0] = (e[0] * AA[1] + e[2] * AA[0]);
        } else {
            const int q = it - n8;
            const float *e = buffer + n2 - 3 - 4 * q;
            float *d = buf2 + n4 - 2 - 2 * q;
            const float *AA = A + n4 + 2 * q;
            d[1] = (-e[2] * AA[0] - -e[0] * AA[1]);
            d[0] = (-e[2] * AA[1] + -e[0] * AA[0]);
        }
    }
    pass_sync<kThreads>();

    // step 2, :2006-2040 (n/8 half-iterations)
    for (int it = tid; it < n8; it += kThreads) {
        const int q = it >> 1, h = it & 1;
        const float *AA = A + n2 - 8 - 8 * q;
        const float *e0 = v + n4 + 4 * q, *e1 = v + 4 * q;
        float *d0 = u + n4 + 4 * q, *d1 = u + 4 * q;
        if (h == 0) {
            float v41_21 = e0[1] - e1[1];
            float v40_20 = e0[0] - e1[0];
            d0[1] = e0[1] + e1[1];
            d0[0] = e0[0] + e1[0];
            d1[1] = v41_21 * AA[4] - v40_20 * AA[5];
            d1[0] = v40_20 * AA[4] + v41_21 * AA[5];
        } else {
            float v41_21 = e0[3] - e1[3];
            float v40_20 = e0[2] - e1[2];
            d0[3] = e0[3] + e1[3];
            d0[2] = e0[2] + e1[2];
            d1[3] = v41_21 * AA[0] - v40_20 * AA[1];
            d1[2] = v40_20 * AA[0] + v41_21 * AA[1];
        }
    }
    pass_sync<kThreads>();

    // step 3 stages l = 0 .. ld-7, :2053-2083 (n/8 butterflies each)
    for (int l = 0; l <= ld - 7; l++) {
        const int k0 = n >> (l + 2);
        const int nb_log = ld - (l + 4);          // butterflies per group = n >> (l+4)
        const int nb_mask = (1 << nb_log) - 1;
        for (int it = tid; it < n8; it += kThreads) {
            const int i = it >> nb_log, b = it & nb_mask;
            float *p = u + n2 - 1 - k0 * i - 2 * b;
            const float *a = A + (b << (l + 3));
            bfly(p, p - (k0 >> 1), a[0], a[1]);
        }
        pass_sync<kThreads>();
    }

    // last three stages fused, :1898-1939 (n/32 blocks of 16 floats)
    {
        const float A2 = A[n >> 3];
        for (int it = tid; it < (n >> 5); it += kThreads) {
            float *z = u + n2 - 1 - 16 * it;
            float k00, k11, l00, l11;
            k00 = z[0] - z[-8];
            k11 = z[-1] - z[-9];
            l00 = z[-2] - z[-10];
            l11 = z[-3] - z[-11];
            z[0] = z[0] + z[-8];
            z[-1] = z[-1] + z[-9];
            z[-2] = z[-2] + z[-10];
            z[-3] = z[-3] + z[-11];
            z[-8] = k00;
            z[-9] = k11;
            z[-10] = (l00 + l11) * A2;
            z[-11] = (l11 - l00) * A2;

            k00 = z[-4] - z[-12];
            k11 = z[-5] - z[-13];
            l00 = z[-6] - z[-14];
            l11 = z[-7] - z[-15];
            z[-4] = z[-4] + z[-12];
            z[-5] = z[-5] + z[-13];
            z[-6] = z[-6] + z[-14];
            z[-7] = z[-7] + z[-15];
            z[-12] = k11;
            z[-13] = -k00;
            z[-14] = (l11 - l00) * A2;
            z[-15] = (l00 + l11) * -A2;

            iter_54(z);
            iter_54(z - 8);
        }
    }
    pass_sync<kThreads>();

    // steps 4-6: bit-reversed gather u -> v, :2096-2124 (n/8 entries; table of :875-881 computed inline)
    for (int e = tid; e < n8; e += kThreads) {
        const int k4 = (int)((__brev((unsigned)e) >> (32 - ld + 3)) << 2);
        const int q = e >> 1;
        float *d0 = v + n4 - 4 - 4 * q;
        float *d1 = v + n2 - 4 - 4 * q;
        if ((e & 1) == 0) {
            d1[3] = u[k4 + 0];
            d1[2] = u[k4 + 1];
            d0[3] = u[k4 + 2];
            d0[2] = u[k4 + 3];
        } else {
            d1[1] = u[k4 + 0];
            d1[0] = u[k4 + 1];
            d0[1] = u[k4 + 2];
            d0[0] = u[k4 + 3];
        }
    }
    pass_sync<kThreads>();

    // step 7, :2133-2175 (n/8 half-iterations)
    for (int it = tid; it < n8; it += kThreads) {
        const int q = it >> 1, h = it & 1;
        float *d = v + 4 * q;
        float *e = v + n2 - 4 - 4 * q;
        const float *CC = C + 4 * q;
        if (h == 0) {
            float a02 = d[0] - e[2];
            float a11 = d[1] + e[3];
            float b0 = CC[1] * a02 + CC[0] * a11;
            float b1 = CC[1] * a11 - CC[0] * a02;
            float b2 = d[0] + e[2];
            float b3 = d[1] - e[3];
            d[0] = b2 + b0;
            d[1] = b3 + b1;
            e[2] = b2 - b0;
            e[3] = b1 - b3;
        } else {
            float a02 = d[2] - e[0];
            float a11 = d[3] + e[1];
            float b0 = CC[3] * a02 + CC[2] * a11;
            float b1 = CC[3] * a11 - CC[2] * a02;
            float b2 = d[2] + e[0];
            float b3 = d[3] - e[1];
            d[2] = b2 + b0;
            d[3] = b3 + b1;
            e[0] = b2 - b0;
            e[1] = b1 - b3;
        }
    }
    pass_sync<kThreads>();

    // step 8 + decode, :2187-2238 (n/4 items)
    for (int it = tid; it < n4; it += kThreads) {
        const int q = it >> 2, m = it & 3;
        const float *BB = B + n2 - 8 - 8 * q;
        const float *e = buf2 + n2 - 8 - 8 * q;
        const float ea = e[6 - 2 * m], eb = e[7 - 2 * m];
        const float ba = BB[6 - 2 * m], bb = BB[7 - 2 * m];
        const float pa = ea * bb - eb * ba;
        const float pb = -ea * ba - eb * bb;
        buffer[4 * q + m] = pa;
        buffer[n2 - 4 - 4 * q + 3 - m] = -pa;
        buffer[n2 + 4 * q + m] = pb;
        buffer[n - 4 - 4 * q + 3 - m] = pb;
    }
    pass_sync<kThreads>();
}

// stb_vorbis2.d:2333-2349
__device__ __forceinline__ void window_bounds(int bs0, int bs1, unsigned fl, int &n, int &left_start,
                                              int &right_start, int &right_end)
{
    const bool lng = (fl & AFG_VORBIS_LONG) != 0;
    const bool prev = lng && (fl & AFG_VORBIS_PREV);
    const bool next = lng && (fl & AFG_VORBIS_NEXT);
    n = lng ? bs1 : bs0;
    left_start = (lng && !prev) ? ((n - bs0) >> 2) : 0;
    if (lng && !next) {
        right_start = (n * 3 - bs0) >> 2;
        right_end = (n * 3 + bs0) >> 2;
    } else {
        right_start = n >> 1;
        right_end = n;
    }
}

}  // namespace afg_vorbis
