#include "afg_flac_front.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> base(1 << 22);
    base.resize(fread(base.data(), 1, base.size(), f));
    fclose(f);
    std::mt19937 rng(atoi(argv[2]));
    size_t total = 0;
    for (int it = 0; it < atoi(argv[3]); it++) {
        std::vector<uint8_t> v = base;
        int n = 1 + rng() % 6;
        for (int k = 0; k < n; k++) {
            size_t pos = (rng() % 3 == 0) ? rng() % std::min<size_t>(v.size(), 80) : rng() % v.size();
            switch (rng() % 4) {
            case 0: v[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
            case 1: v[pos] = (uint8_t)rng(); break;
            case 2: v.erase(v.begin() + pos, v.begin() + std::min(v.size(), pos + rng() % 400)); break;
            default: v.insert(v.begin() + pos, rng() % 100, (uint8_t)rng()); break;
            }
            if (v.empty()) v.push_back(0);
        }
        uint8_t *p = (uint8_t *)malloc(v.size());
        memcpy(p, v.data(), v.size());
        afg_front::FlacInfo fi; afg_front::FlacRecords rec;
        if (afg_front::flac_parse(p, v.size(), fi, rec)) total += rec.frames.size();
        afg_front::QoaInfo qi; std::vector<afg_qoa_frame> q;
        afg_front::qoa_parse(p, v.size(), qi, q);
        free(p);
    }
    printf("ok frames=%zu\n", total);
}
