#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../project3-cuda-path-tracer_amd/csrc/pt_bvh.hpp"
#include "../../project3-cuda-path-tracer_amd/csrc/pt_cull.hpp"
int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-5, 5), S(0.001f, 2.0f);
    for (int trial = 0; trial < 60; ++trial) {
        int n = (trial < 8) ? trial : (int)(rng() % 20000);
        std::vector<float> v((size_t)n * 9);
        for (int i = 0; i < n; ++i) {
            float c[3] = {U(rng), U(rng), U(rng)}; float s = S(rng);
            for (int k = 0; k < 9; ++k) v[(size_t)i * 9 + k] = c[k % 3] + s * U(rng) * 0.2f;
            if (trial % 7 == 3 && i % 3 == 0) for (int k = 0; k < 9; ++k) v[(size_t)i * 9 + k] = 1.0f;   // coincident
            if (trial % 11 == 5 && i % 50 == 0) v[(size_t)i * 9] = NAN;
            if (trial % 13 == 6 && i % 70 == 0) v[(size_t)i * 9 + 4] = INFINITY;
        }
        ptbvh::Tree t;
        ptbvh::build(v.data(), n, t);
        // every triangle once
        std::vector<int> seen((size_t)n, 0);
        for (int x : t.order) seen[(size_t)x]++;
        for (int i = 0; i < n; ++i) if (seen[(size_t)i] != 1) { printf("trial %d: triangle %d seen %d times\n", trial, i, seen[(size_t)i]); return 1; }
        // links in range, leaves tile the slots
        long covered = 0;
        for (int k = 0; k < t.num_nodes(); ++k) {
            const float *r = &t.nodes[(size_t)k * ptbvh::NODE_WORDS];
            for (int c = 0; c < 2; ++c) {
                uint32_t w; memcpy(&w, &r[6 + c], 4);
                uint32_t link = w & 0xffffffu, info = w >> 24;
                if (info & 8) { covered += info & 7; if ((long)link + (info & 7) > n) { printf("leaf out of range\n"); return 1; } }
                else if ((int)link <= k || (int)link >= t.num_nodes()) { printf("bad child link\n"); return 1; }
            }
            for (int o = 0; o < 8; ++o) { int32_t m; memcpy(&m, &r[8 + o], 4); if (m != -1 && (m < 0 || m == k || m >= t.num_nodes())) { printf("bad miss link %d at %d\n", m, k); return 1; } }
        }
        if (covered != n) { printf("trial %d: leaves cover %ld of %d\n", trial, covered, n); return 1; }
    }
    // cull boxes: random affine maps incl. singular, non-finite, huge and tiny ones
    for (int trial = 0; trial < 200; ++trial) {
        const int n = 1 + (int)(rng() % 40);
        std::vector<float> m((size_t)n * 16);
        std::vector<const float *> ptr((size_t)n);
        std::vector<char> sph((size_t)n), skip((size_t)n);
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 16; ++k) m[(size_t)i * 16 + k] = (k % 5 == 0 ? 1.0f : 0.0f) * S(rng) * 50.0f + ((rng() % 3 == 0) ? U(rng) : 0.0f);
            if (rng() % 9 == 0) m[(size_t)i * 16 + (rng() % 16)] = (rng() % 2) ? NAN : INFINITY;
            if (rng() % 9 == 1) for (int k = 0; k < 4; ++k) m[(size_t)i * 16 + 4 + k] = 0.0f;      // singular
            if (rng() % 9 == 2) for (int k = 0; k < 16; ++k) m[(size_t)i * 16 + k] *= 1e30f;
            ptr[(size_t)i] = &m[(size_t)i * 16]; sph[(size_t)i] = rng() % 2; skip[(size_t)i] = rng() % 7 == 0;
        }
        const double eye[3] = {U(rng), U(rng), trial % 5 == 0 ? (double)INFINITY : U(rng)};
        std::vector<ptcull::Box> bx;
        const float R = ptcull::make_boxes(ptr.data(), reinterpret_cast<const bool *>(sph.data()), reinterpret_cast<const bool *>(skip.data()), n, eye, 1, bx);
        if (!(R >= 1.0f) || (int)bx.size() != n) { printf("cull: bad bound %g\n", R); return 1; }
        for (int i = 0; i < n; ++i) {
            float row[4];
            const int ax = ptcull::reject_row(ptr[(size_t)i], row);
            if (ax < 0 || ax > 4) { printf("cull: bad reject mode\n"); return 1; }
            for (int k = 0; k < 3; ++k) if (!(bx[(size_t)i].lo[k] <= bx[(size_t)i].hi[k])) { printf("cull: inverted / NaN box\n"); return 1; }
        }
    }
    printf("bvh + cull harness ok\n");
    return 0;
}
