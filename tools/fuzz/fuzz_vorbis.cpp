#include "afg_vorbis_front.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> base(1 << 20);
    base.resize(fread(base.data(), 1, base.size(), f));
    fclose(f);
    std::mt19937 rng(atoi(argv[2]));
    size_t total = 0;
    for (int it = 0; it < atoi(argv[3]); it++) {
        std::vector<uint8_t> v = base;
        int n = 1 + rng() % 8;
        for (int k = 0; k < n; k++) {
            size_t pos = (rng() % 3) ? rng() % v.size() : 58 + rng() % 4000;   // the setup header gets its share
            if (pos >= v.size()) pos = v.size() - 1;
            switch (rng() % 4) {
            case 0: v[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
            case 1: v[pos] = (uint8_t)rng(); break;
            case 2: v.erase(v.begin() + pos, v.begin() + std::min(v.size(), pos + rng() % 400)); break;
            default: v.insert(v.begin() + pos, rng() % 100, (uint8_t)rng()); break;
            }
            if (v.empty()) v.push_back(0);
        }
        // exact-size heap copy so that ASAN sees reads past the end
        uint8_t *p = (uint8_t *)malloc(v.size());
        memcpy(p, v.data(), v.size());
        afg_vorbis::File out;
        if (afg_vorbis::parse_file(p, v.size(), out)) total += out.pflags.size();
        free(p);
    }
    printf("ok blocks=%zu\n", total);
}
