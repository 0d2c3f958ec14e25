// Plain C++ caller of the drop-in boundary (no Python, no torch): dam-break lattice, a few frames of
// SphFluidSimulation::Update, invariants of VelPos.compute:150-154 checked on the read-back.
//   g++ -O2 -std=c++17 -Iinclude host/sph_host_demo.cpp -Lsphfluidsimulation_amd/lib -lsph_hip -o host/sph_host_demo
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "SphFluidSimulation.hpp"

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 262144;
    const int R = argc > 2 ? atoi(argv[2]) : 47;
    const int frames = argc > 3 ? atoi(argv[3]) : 5;
    SphFluidSimulation sim;
    sim.particleNumber = n;
    sim.bucketResolution = R;
    sim.damFillRate = 0.8f;
    // Preset1 lattice, Assets/Resources/InitParticles.compute:19-35 (no jitter)
    const int ppd = (int)std::ceil(std::cbrt((float)n / 0.8f));
    const int xs = (int)std::ceil(ppd * 0.8f), ys = ppd, zs = ppd;
    const float cs = 1.0f / ppd;
    std::vector<float> pos(4 * (size_t)n), vel;
    for (int i = 0; i < n; i++) {
        pos[4 * i + 0] = cs / 2 + (float)(i / (zs * ys)) * 0.8f / xs;
        pos[4 * i + 1] = cs / 2 + (float)((i / zs) % ys) * 0.9f / ys;
        pos[4 * i + 2] = cs / 2 + (float)(i % zs) * 1.0f / zs;
        pos[4 * i + 3] = 1.0f;
    }
    try {
        sim.Start(pos.data(), nullptr);
        const float dt_frame = 0.7666667f * sim._effectiveRadius;
        for (int f = 0; f < frames; f++) sim.Update(dt_frame);
        std::vector<float> p, v, rho;
        sim.Download(p, v, rho);
        int bad = 0;
        for (unsigned i = 0; i < sim.count(); i++) {
            if (p[4 * i + 3] != 1.0f || v[4 * i + 3] != 5.0f * frames) bad++;   // pos.w == 1, vel.w counts sub-steps
            for (int a = 0; a < 3; a++)
                if (!(p[4 * i + a] >= 0.0f && p[4 * i + a] <= 1.0f)) bad++;      // clamp, VelPos.compute:154
        }
        // row f3: render feed from the same state; identity _SimTRS -> mat = translate(pos - 0.5) * scale(radius)
        sim.UpdateMeshProperties();
        std::vector<float> props(20 * (size_t)sim.count());
        sim.DownloadMeshProperties(props.data());
        for (unsigned i = 0; i < sim.count(); i++) {
            const float *r = &props[20 * (size_t)i];
            for (int a = 0; a < 3; a++)
                if (r[12 + a] != p[4 * i + a] * 1.0f - 0.5f || r[5 * a] != sim.particleRadius) bad++;
            if (r[15] != 1.0f || r[19] != 1.0f || r[17] != 0.0f || r[18] != 1.0f - r[16]) bad++;
        }
        SphStats st = sim.Stats();
        printf("host demo: N=%u frames=%d max_cell_occupancy=%u nan_resets=%llu violations=%d\n", sim.count(), frames,
               st.max_cell_occupancy, (unsigned long long)st.nan_resets, bad);
        return bad ? 1 : 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}
