// Exercises the C++ host mirrors (cube_slam_amd/host/*.hpp) end to end on one gray frame read from a raw file; prints what the Python
// test compares with the Python mirrors' results: keypoint / descriptor / KeyLine checksums.
#include <cstdio>
#include <cstring>
#include <vector>

#include "cube_slam_amd/host/orb_slam_mirrors.hpp"

static unsigned long long fnv(const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    std::vector<uint8_t> img((size_t)W * H);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
    fclose(f);
    try {
        cubeslam::Context ctx(0);
        cubeslam::ORBextractor orb(ctx, 500, 1.2f, 8, 20, 7, W, H);
        std::vector<cs_keypoint> kps; std::vector<uint8_t> desc;
        orb(img.data(), W, kps, desc);
        printf("orb %zu %llx %llx\n", kps.size(), fnv(kps.data(), kps.size() * sizeof(cs_keypoint)), fnv(desc.data(), desc.size()));
        printf("levels %d sf1 %.9g\n", orb.GetLevels(), (double)orb.GetScaleFactors()[1]);
        cubeslam::line_lbd_detect ld(ctx, W, H);
        std::vector<cs_keyline> kl; std::vector<float> lm; std::vector<uint8_t> ldesc;
        ld.detect_raw_lines(img.data(), W, kl);
        ld.detect_filter_lines(img.data(), W, lm);
        ld.get_line_descriptors(img.data(), W, kl, ldesc);
        printf("lines %zu %llx filtered %zu %llx lbd %llx\n", kl.size(), fnv(kl.data(), kl.size() * sizeof(cs_keyline)), lm.size() / 4, fnv(lm.data(), lm.size() * 4), fnv(ldesc.data(), ldesc.size()));
    } catch (const std::exception &e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
