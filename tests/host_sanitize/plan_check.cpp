// plan_check.cpp -- the HOST side of the library (frame planner of slab.hip, context bookkeeping) under ASan + UBSan:
// dumps the frame plans of 1/2/3/8 ranks for both slab plans through fs_slab_plan_dump (no GPU is touched).  Built by
// tests/test_slab_plan.py with hipcc, sanitizers on the host code only (-Xarch_host -fsanitize=...).
#include <cstdio>
#include <cstring>
#include <vector>
#include "fluidsolver.h"
int main()
{
    std::vector<char> buf(1 << 20);
    int total = 0;
    const int worlds[] = {1, 2, 3, 8};
    for (int overlap = 0; overlap < 2; ++overlap)
        for (int w : worlds)
            for (int rank = 0; rank < w; ++rank) {
                fs_params p;
                fs_default_params(&p);
                p.vorticityStrength = 10.0f; p.solverIterations = 40; p.displayMode = FS_DISPLAY_NONE;
                int halos[3] = {2, 9, 30};
                int rc = fs_slab_plan_dump(16384, 16384, w, rank, -1, overlap, &p, 3, halos, buf.data(), buf.size());
                if (rc) { std::fprintf(stderr, "plan dump failed: %s\n", fs_last_error(nullptr)); return 1; }
                total += (int)std::strlen(buf.data());
                p.diffusion = 0.00001f; p.solverIterations = 20;
                rc = fs_slab_plan_dump(300, 260, w, rank, 12, overlap, &p, 3, halos, buf.data(), buf.size());
                if (rc && w * 4 <= 260) { std::fprintf(stderr, "plan dump failed: %s\n", fs_last_error(nullptr)); return 1; }
            }
    std::printf("plan_check ok (%d bytes of plans)\n", total);
    return 0;
}
