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
        const bool is_flac = afg_front::flac_parse(p, v.size(), fi, rec);
        if (is_flac) total += rec.frames.size();
        // the batch path: bound from STREAMINFO, then parse into an exact-size external plane (ASAN guards its end)
        const size_t bound = afg_front::flac_res_bound(p, v.size());
        if (bound && !is_flac) { printf("bound without a FLAC stream\n"); return 1; }
        if (bound && bound < ((size_t)1 << 26)) {
            int32_t *dst = (int32_t *)malloc(bound * sizeof(int32_t));
            afg_front::FlacInfo f2; afg_front::FlacRecords r2;
            const bool ok = afg_front::flac_parse_into(p, v.size(), f2, r2, dst, bound);
            if (!ok) { printf("staged parse rejected a FLAC stream\n"); return 1; }
            if (!r2.overflow && (r2.n_res != rec.res.size() || r2.frames.size() != rec.frames.size() ||
                                 (r2.n_res && memcmp(dst, rec.res.data(), r2.n_res * sizeof(int32_t))))) { printf("staged parse differs\n"); return 1; }
            free(dst);
        }
        // int16 residual rows: the packed parse must describe the same residuals (rows widened back), inside the same words
        if (is_flac) {
            afg_front::FlacInfo f3; afg_front::FlacRecords r3;
            r3.pack16 = true;
            if (!afg_front::flac_parse(p, v.size(), f3, r3) || r3.frames.size() != rec.frames.size() || r3.res.size() != rec.res.size()) {
                printf("packed parse differs in shape\n"); return 1;
            }
            for (size_t i = 0; i < rec.frames.size(); i++) {
                const afg_flac_frame &a = rec.frames[i], &b = r3.frames[i];
                if (a.block_size != b.block_size || a.channels != b.channels || a.res16) { printf("packed parse: frame differs\n"); return 1; }
                for (unsigned c = 0; c < a.channels; c++) {
                    const int32_t *want = rec.res.data() + a.in_off + (size_t)c * a.block_size;
                    for (uint32_t k = 0; k < a.block_size; k++) {
                        int32_t got;
                        if (b.res16) {
                            const size_t at = b.in_off + (size_t)c * AFG_FLAC_ROW16(b.block_size) + k;      // int16 index
                            if ((b.in_off & 7) || at / 2 >= r3.res.size()) { printf("packed row out of range\n"); return 1; }
                            got = ((const int16_t *)r3.res.data())[at];
                        } else {
                            got = r3.res[b.in_off + (size_t)c * b.block_size + k];
                        }
                        if (got != want[k]) { printf("packed parse: residual differs\n"); return 1; }
                    }
                }
            }
        }
        afg_front::QoaInfo qi; std::vector<afg_qoa_frame> q;
        afg_front::qoa_parse(p, v.size(), qi, q);
        free(p);
    }
    printf("ok frames=%zu\n", total);
}
