#include "afg_vorbis_front.h"
#include "vorbis_front_tables.h"
#include <algorithm>
#include <vector>
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
        // the device-floor form: every record must stay inside what was recorded (the kernel trusts them), and applying
        // them the way csrc/vorbis_floor.hip does (draw_line in closed form) must give the host path's spectra
        if (out.channels) {
            afg_vorbis::File r;
            if (!afg_vorbis::parse_file(p, v.size(), r, true) || r.pflags != out.pflags || r.n_spec != out.spec.size() ||
                r.fl_packets.size() != r.pflags.size()) { printf("device-floor parse differs\n"); return 1; }
            std::vector<float> sp = r.spec;
            for (const afg_vorbis_floor_packet &k : r.fl_packets) {
                const size_t n2 = k.n2, C = k.channels;
                if (k.spec_off + C * n2 > sp.size() || (size_t)k.curve_index + C > r.fl_curves.size() ||
                    2 * ((size_t)k.step_off + k.n_steps) > r.fl_steps.size() || (n2 & 3)) { printf("floor record out of range\n"); return 1; }
                float *base = sp.data() + k.spec_off;
                for (uint32_t s = 0; s < k.n_steps; s++) {
                    const unsigned m = r.fl_steps[2 * (k.step_off + s)], a = r.fl_steps[2 * (k.step_off + s) + 1];
                    if (m >= C || a >= C || m == a) { printf("coupling step out of range\n"); return 1; }
                    float *mp = base + m * n2, *ap = base + a * n2;
                    for (size_t j = 0; j < n2; j++) {
                        const float mv = mp[j], av = ap[j];
                        float m2, a2;
                        if (mv > 0) { if (av > 0) { m2 = mv; a2 = mv - av; } else { a2 = mv; m2 = mv + av; } }
                        else { if (av > 0) { m2 = mv; a2 = mv + av; } else { a2 = mv; m2 = mv - av; } }
                        mp[j] = m2; ap[j] = a2;
                    }
                }
                for (size_t c = 0; c < C; c++) {
                    const afg_vorbis_floor_curve cv = r.fl_curves[k.curve_index + c];
                    float *t = base + c * n2;
                    if (cv.n_points == 0) { std::fill(t, t + n2, 0.0f); continue; }
                    if (2 * ((size_t)cv.point_off + cv.n_points) > r.fl_points.size()) { printf("curve out of range\n"); return 1; }
                    const int32_t *pts = r.fl_points.data() + 2 * (size_t)cv.point_off;
                    const int np = (int)cv.n_points;
                    if (pts[0] != 0) { printf("curve does not start at 0\n"); return 1; }
                    for (int q = 1; q < np; q++) if (pts[2 * q] < pts[2 * q - 2]) { printf("curve x not ascending\n"); return 1; }
                    int sgm = 0;
                    for (int j = 0; j < (int)n2; j++) {
                        while (sgm + 1 < np && pts[2 * (sgm + 1)] <= j) sgm++;
                        const int x0 = pts[2 * sgm], y0 = pts[2 * sgm + 1];
                        int y = y0;
                        if (sgm + 1 < np) {
                            const int x1 = pts[2 * sgm + 2], y1 = pts[2 * sgm + 3];
                            const int dy = y1 - y0, adx = x1 - x0, bs = dy / adx, sy = dy < 0 ? bs - 1 : bs + 1;
                            const int ady = (dy < 0 ? -dy : dy) - (bs < 0 ? -bs : bs) * adx, kk = j - x0;
                            y = y0 + kk * bs + (sy - bs) * (int)(((int64_t)kk * ady) / adx);
                        }
                        uint32_t bits = k_inverse_db_bits[y & 255];
                        float f;
                        memcpy(&f, &bits, 4);
                        t[j] *= f;
                    }
                }
            }
            if (sp.size() && memcmp(sp.data(), out.spec.data(), sp.size() * sizeof(float))) {
                // (NaN payloads may differ in principle; compare as values then)
                for (size_t i = 0; i < sp.size(); i++)
                    if (memcmp(&sp[i], &out.spec[i], 4) && !(sp[i] != sp[i] && out.spec[i] != out.spec[i])) { printf("device-floor spectra differ at %zu\n", i); return 1; }
            }
        }
        free(p);
    }
    printf("ok blocks=%zu\n", total);
}
