/* sanitize_check.c -- the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host code only): a few frames
 * of update() with every stage on at sizes that are not multiples of anything, splats that straddle the domain
 * edge, the slab form of the kernels on row ranges of heap buffers sized exactly, and the tracer functions.
 * Test infrastructure: `make -C oracle sanitize` builds and runs it; exit code 0 and no sanitizer report = pass. */
#include <stdio.h>
#include <stdlib.h>

#include "fluid_oracle.h"

static float *plane(int rows, int W, unsigned seed)
{
    float *p = (float *)malloc((size_t)rows * W * sizeof(float));
    fo_fill_uniform(p, (long)rows * W, -2.0f, 2.0f, seed);
    return p;
}

int main(void)
{
    /* whole solver, dye diffusion and vorticity on */
    const int W = 37, H = 29;
    fo_state *s = fo_create(W, H);
    fo_params p = {0.00015f, 0.00001f, 0.5f, 10.0f, 0.002f, 7};
    for (int f = 0; f < 5; ++f) {
        fo_add_force(s, 0.5f + 9.0f * f, 1.0f + 6.0f * f, 30.0f, -12.0f, 100.0f, 50.0f, 25.0f, 5.0f, p.deltaTime);
        fo_add_force(s, (float)W - 1.0f, (float)H - 0.5f, -300.0f, 250.0f, 1.0f, 2.0f, 3.0f, 9.0f, p.deltaTime);
        fo_update(s, &p);
    }
    double sum = 0.0;
    for (int i = 0; i < W * H; ++i) sum += fo_field(s, FO_VELOCITY, 0)[i] + fo_field(s, FO_DENSITY, 2)[i];
    fo_destroy(s);

    /* slab form: a buffer of rows [5, 20) of a 29-row domain, computing rows [7, 18) */
    fo_grid g = {W, H, 5, 15, 7, 18};
    float *u = plane(15, W, 1), *v = plane(15, W, 2), *a = plane(15, W, 3), *b = plane(15, W, 4), *c = plane(15, W, 5);
    fo_calculateVorticity(&g, u, v, a);
    fo_applyVorticity(&g, a, u, v, 10.0f, 0.5f);
    fo_diffuse(&g, u, b, 0.00015f, 0.5f);
    fo_divergence(&g, u, v, c);
    fo_pressureSolve(&g, c, a, b);
    fo_subtractGradient(&g, b, u, v);
    fo_fadeDensity(&g, a, 0.002f);
    fo_clear(&g, c);
    free(a); free(b); free(c);

    /* tracers on the full grid, some far outside the window */
    fo_grid full = {W, 15, 0, 15, 0, 15};
    const long n = 1000;
    float *arr[7];
    for (int i = 0; i < 7; ++i) arr[i] = (float *)malloc(n * sizeof(float));
    fo_fill_uniform(arr[0], n, -80.0f, 120.0f, 11);
    fo_fill_uniform(arr[1], n, -80.0f, 95.0f, 12);
    fo_fill_uniform(arr[2], n, -3.0f, 3.0f, 13);
    fo_fill_uniform(arr[3], n, -3.0f, 3.0f, 14);
    fo_fill_uniform(arr[4], n, 0.0f, 1.0f, 15);
    fo_fill_uniform(arr[5], n, 0.1f, 1.0f, 16);
    fo_fill_uniform(arr[6], n, 0.0f, 2.0f, 17);
    fo_particle_params pp = {0.5f, 0.6f, 3.0f, 1, 0.00015f, 37.0f, 15.0f};
    for (int k = 0; k < 3; ++k) fo_particles_update(&full, u, v, &pp, n, arr[0], arr[1], arr[2], arr[3], arr[4], arr[5], arr[6]);
    float *xy = (float *)malloc(4 * n * sizeof(float)), *al = (float *)malloc(n * sizeof(float));
    fo_particle_lines(n, arr[0], arr[1], arr[2], arr[3], arr[4], 37.0f, 15.0f, 1.5f, xy, al);
    for (long i = 0; i < n; ++i) sum += al[i];
    for (int i = 0; i < 7; ++i) free(arr[i]);
    free(xy); free(al); free(u); free(v);
    printf("sanitize_check ok (%g)\n", sum);
    return sum == sum ? 0 : 1;
}
