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
        // the batch path: bound from the headers, then decode into an exact-size external buffer (ASAN guards its end)
        const size_t bound = afg_vorbis::max_spec_floats(p, v.size());
        if ((bound != 0) != (out.channels != 0)) { printf("probe/parse disagree\n"); return 1; }
        if (bound) {
            float *dst = (float *)malloc(bound * sizeof(float));
            afg_vorbis::File o2;
            const bool ok = afg_vorbis::parse_file_into(p, v.size(), o2, dst, bound);
            if (!ok || o2.overflow || o2.n_spec != out.spec.size() || o2.pflags != out.pflags ||
                (o2.n_spec && memcmp(dst, out.spec.data(), o2.n_spec * sizeof(float)))) { printf("staged parse differs\n"); return 1; }
            free(dst);
        }
        free(p);
    }
    printf("ok blocks=%zu\n", total);
}
