#include "afg_opus_front.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
// Mutation fuzzing of the Ogg Opus front-end under ASan / UBSan.  Pages carry checksums, so most byte flips only end the
// stream at that page; mode 1 therefore repairs every page checksum after mutating (the damage then reaches the packet
// framing, the range decoder and the CELT layer).
static uint32_t crc_tab[256];
static uint32_t ogg_crc(const uint8_t *p, size_t n)
{
    uint32_t c = 0;
    for (size_t i = 0; i < n; i++) c = (c << 8) ^ crc_tab[((c >> 24) ^ ((i - 22 < 4) ? 0 : p[i])) & 0xff];
    return c;
}
static void repair(std::vector<uint8_t> &v)
{
    size_t at = 0;
    while (at + 27 <= v.size() && !memcmp(v.data() + at, "OggS", 4)) {
        const size_t nseg = v[at + 26];
        if (at + 27 + nseg > v.size()) break;
        size_t len = 27 + nseg;
        for (size_t i = 0; i < nseg; i++) len += v[at + 27 + i];
        if (at + len > v.size()) break;
        const uint32_t c = ogg_crc(v.data() + at, len);
        v[at + 22] = (uint8_t)c; v[at + 23] = (uint8_t)(c >> 8); v[at + 24] = (uint8_t)(c >> 16); v[at + 25] = (uint8_t)(c >> 24);
        at += len;
    }
}
int main(int argc, char **argv)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t r = i << 24;
        for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
        crc_tab[i] = r;
    }
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> base(1 << 22);
    base.resize(fread(base.data(), 1, base.size(), f));
    fclose(f);
    std::mt19937 rng(atoi(argv[2]));
    size_t total = 0, opened = 0;
    for (int it = 0; it < atoi(argv[3]); it++) {
        std::vector<uint8_t> v = base;
        int n = 1 + rng() % 8;
        const bool keep_lengths = rng() % 2;
        for (int k = 0; k < n; k++) {
            size_t pos = rng() % v.size();
            switch (keep_lengths ? rng() % 2 : rng() % 4) {
            case 0: v[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
            case 1: v[pos] = (uint8_t)rng(); break;
            case 2: v.erase(v.begin() + pos, v.begin() + std::min(v.size(), pos + rng() % 400)); break;
            default: v.insert(v.begin() + pos, rng() % 100, (uint8_t)rng()); break;
            }
            if (v.empty()) v.push_back(0);
        }
        if (rng() % 4) repair(v);
        uint8_t *p = (uint8_t *)malloc(v.size());
        memcpy(p, v.data(), v.size());
        afg_opus::File out;
        if (afg_opus::parse_file(p, v.size(), out) == afg_opus::kOpened) { opened++; total += out.frames.size(); }
        // the chunked reader over the same bytes
        afg_opus::Reader r;
        afg_opus::File meta, chunk;
        if (r.open(p, v.size(), meta) == afg_opus::kOpened)
            while (r.more(chunk, 7)) total += chunk.error;
        free(p);
    }
    printf("ok opened=%zu frames=%zu\n", opened, total);
}
