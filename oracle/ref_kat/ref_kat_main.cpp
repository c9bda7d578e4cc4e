// ref_kat_main.cpp -- known-answer generator built from the REFERENCE's own, unmodified,
// Eigen-free headers (compiled where they lie under /root/reference; see oracle/Makefile `ref`).
// It is the only part of the reference path that builds in this image (everything else needs
// Eigen).  Its stdout is committed as tests/golden/ref_kat.json by tools/gen_golden.py and pins
// rows a8 (DistVoxel/ColorVoxel), a11 (truncators, weighter), a16 (ColorImage::At; Interpolate.h) of
// the oracle restatement and the HIP kernels must reproduce it bit for bit.
// Floats are printed as their IEEE-754 bit patterns (hex) so the fixture is exact.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include <open_chisel/ColorVoxel.h>
#include <open_chisel/DistVoxel.h>
#include <open_chisel/camera/ColorImage.h>
#include <open_chisel/geometry/Interpolate.h>
#include <open_chisel/truncation/ConstantTruncator.h>
#include <open_chisel/truncation/InverseTruncator.h>
#include <open_chisel/truncation/QuadraticTruncator.h>
#include <open_chisel/weighting/ConstantWeighter.h>

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fromBits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t lcg_state = 20260102u;
static uint32_t lcg() { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state; }
static float urand(float lo, float hi) { return lo + (hi - lo) * ((lcg() >> 8) * (1.0f / 16777216.0f)); }

int main() {
    printf("{\n");
    printf("\"sizeof_DistVoxel\": %zu,\n\"sizeof_ColorVoxel\": %zu,\n", sizeof(chisel::DistVoxel), sizeof(chisel::ColorVoxel));

    // ---- DistVoxel: sequences of Integrate / Carve ------------------------------------------
    printf("\"dist_sequences\": [\n");
    const int nSeq = 24, seqLen = 40;
    for (int s = 0; s < nSeq; s++) {
        chisel::DistVoxel v;
        printf(" {\"init\": [\"%08x\", \"%08x\"], \"steps\": [", bits(v.GetSDF()), bits(v.GetWeight()));
        for (int k = 0; k < seqLen; k++) {
            int op = (lcg() >> 24) % 16;  // 0: carve, else integrate
            float d = urand(-0.3f, 0.3f);
            float wu = (s % 3 == 0) ? 1.0f : urand(0.05f, 6.0f);
            if (s == 1 && k == 7) wu = std::numeric_limits<float>::infinity();  // weighter at truncation 0
            if (s == 2 && k == 5) d = std::numeric_limits<float>::quiet_NaN();
            if (op == 0) {
                v.Carve();
                printf("%s[\"c\", \"0\", \"0\", \"%08x\", \"%08x\"]", k ? ", " : "", bits(v.GetSDF()), bits(v.GetWeight()));
            } else {
                v.Integrate(d, wu);
                printf("%s[\"i\", \"%08x\", \"%08x\", \"%08x\", \"%08x\"]", k ? ", " : "", bits(d), bits(wu), bits(v.GetSDF()), bits(v.GetWeight()));
            }
        }
        printf("]}%s\n", s + 1 < nSeq ? "," : "");
    }
    printf("],\n");

    // ---- ColorVoxel ---------------------------------------------------------------------------
    printf("\"color_sequences\": [\n");
    for (int s = 0; s < 16; s++) {
        chisel::ColorVoxel v;
        printf(" [");
        int len = (s == 0) ? 300 : 24;  // s==0 runs into the weight >= 255 - wu saturation branch
        for (int k = 0; k < len; k++) {
            uint8_t r = lcg() >> 24, g = lcg() >> 24, b = lcg() >> 24;
            uint8_t wu = (s < 8) ? 1 : (uint8_t)(1 + (lcg() >> 24) % 40);
            v.Integrate(r, g, b, wu);
            printf("%s[%u,%u,%u,%u,%u,%u,%u,%u]", k ? "," : "", r, g, b, wu, v.GetRed(), v.GetGreen(), v.GetBlue(), v.GetWeight());
        }
        printf("]%s\n", s + 1 < 16 ? "," : "");
    }
    printf("],\n");

    // ---- truncators and weighter ---------------------------------------------------------------
    std::vector<float> depths;
    const float special[] = {0.0f, -0.0f, 1e-30f, -1e-30f, 1e-10f, 0.001f, -0.5f, -3.0f, 0.05f, 0.1f, 0.3f, 1.0f, 2.0f, 2.5f, 5.0f,
                             20.0f, 49.9f, 50.0f, 50.1f, 99.9f, 100.0f, 100.5f, 1e4f, 1e20f, 3e38f,
                             std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(),
                             std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::denorm_min()};
    for (float f : special) depths.push_back(f);
    for (int i = 0; i < 400; i++) depths.push_back(urand(0.05f, 20.0f));
    for (int i = 0; i < 100; i++) depths.push_back(fromBits(lcg()));  // arbitrary bit patterns
    const float params[] = {0.5f, 1.0f, 2.0f, 8.0f, 0.04f};
    printf("\"depths\": [");
    for (size_t i = 0; i < depths.size(); i++) printf("%s\"%08x\"", i ? "," : "", bits(depths[i]));
    printf("],\n\"truncation\": {\n");
    for (int p = 0; p < 5; p++) {
        chisel::ConstantTruncator ct(params[p]);
        chisel::InverseTruncator it(params[p]);
        chisel::QuadraticTruncator qt(params[p]);
        chisel::ConstantWeighter w1(1.0f), w2(params[p]);
        const chisel::Truncator *ts[3] = {&ct, &it, &qt};
        const char *names[3] = {"constant", "inverse", "quadratic"};
        for (int k = 0; k < 3; k++) {
            printf(" \"%s_%08x\": [", names[k], bits(params[p]));
            for (size_t i = 0; i < depths.size(); i++) printf("%s\"%08x\"", i ? "," : "", bits(ts[k]->GetTruncationDistance(depths[i])));
            printf("],\n");
        }
        // weights from the inverse truncation (what ChiselServer wires together)
        printf(" \"weight1_inverse_%08x\": [", bits(params[p]));
        for (size_t i = 0; i < depths.size(); i++) printf("%s\"%08x\"", i ? "," : "", bits(w1.GetWeight(0.0f, it.GetTruncationDistance(depths[i]))));
        printf("],\n \"weightp_inverse_%08x\": [", bits(params[p]));
        for (size_t i = 0; i < depths.size(); i++) printf("%s\"%08x\"", i ? "," : "", bits(w2.GetWeight(0.0f, it.GetTruncationDistance(depths[i]))));
        printf("]%s\n", p + 1 < 5 ? "," : "");
    }
    printf("},\n");

    // ---- ColorImage::At channel decode ---------------------------------------------------------
    printf("\"color_at\": [\n");
    for (int ch = 1; ch <= 4; ch++) {
        const int W = 7, H = 5;
        chisel::ColorImage<uint8_t> img(W, H, ch);
        printf(" {\"channels\": %d, \"width\": %d, \"height\": %d, \"data\": [", ch, W, H);
        for (int i = 0; i < W * H * ch; i++) {
            img.GetMutableData()[i] = lcg() >> 24;
            printf("%s%u", i ? "," : "", img.GetData()[i]);
        }
        printf("], \"rgba\": [");
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                chisel::Color<uint8_t> col;
                img.At(r, c, &col);
                printf("%s[%u,%u,%u,%u]", (r || c) ? "," : "", col.red, col.green, col.blue, col.alpha);
            }
        printf("]}%s\n", ch < 4 ? "," : "");
    }
    printf("],\n");

    // ---- geometry/Interpolate.h -----------------------------------------------------------------
    // Off the live path (its only caller, DepthImage::BilinearInterpolateDepth, is called from
    // nowhere: ProjectionIntegrator.h:72,131 are commented out); pinned so that a maintainer who
    // re-enables it has the vectors.  Own seed: the sections above keep their values.
    lcg_state = 777u;
    printf("\"bilinear\": [");
    for (int k = 0; k < 64; k++) {
        float c[4], tx = urand(0.0f, 1.0f), ty = urand(0.0f, 1.0f);
        for (int i = 0; i < 4; i++) c[i] = urand(0.2f, 8.0f);
        if (k == 3) c[1] = std::numeric_limits<float>::quiet_NaN();
        if (k == 4) { tx = 0.0f; ty = 1.0f; }
        float v = chisel::BilinearInterpolate(c[0], c[1], c[2], c[3], tx, ty);
        printf("%s[\"%08x\",\"%08x\",\"%08x\",\"%08x\",\"%08x\",\"%08x\",\"%08x\"]", k ? "," : "",
               bits(c[0]), bits(c[1]), bits(c[2]), bits(c[3]), bits(tx), bits(ty), bits(v));
    }
    printf("]\n}\n");
    return 0;
}
