// demo.cpp -- C++ host driving the library through the FluidSolver mirror: a few scripted frames,
// prints a checksum (used by tests/test_gpu_host_cpp.py, which checks it against the Python host).
//   g++ -std=c++17 -I include -I host host/demo.cpp -L fluidsolver_amd -lfluidsolver_hip -Wl,-rpath,$PWD/fluidsolver_amd
#include <cstdio>
#include <cstdlib>

#include "FluidSolver.hpp"

int main(int argc, char **argv)
{
    using namespace fluidsolver;
    const int n = argc > 1 ? std::atoi(argv[1]) : 128;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 5;
    try {
        FluidSolver solver(n, n);
        solver.applyPreset(presets()[2]);  // "Turbulent"
        solver.displayMode = FluidSolver::DisplayMode::density;
        const float2 window = {640.0f, 640.0f};
        for (int f = 0; f < frames; ++f) {
            solver.addForce({200.0f + 20.0f * f, 300.0f + 7.0f * f}, window, (float)(f % 360) / 360.0f);
            solver.update();
        }
        solver.sync();
        std::vector<float> vel = solver.readField(FS_FIELD_VELOCITY, 2);
        std::vector<float> tex = solver.texture();
        double sv = 0, st = 0;
        for (float x : vel) sv += (double)x * x;
        for (float x : tex) st += (double)x;
        std::printf("%d %d %.17g %.17g\n", n, frames, sv, st);
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
