/*
 * fluid_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See fluid_oracle.h for scope, citations legend and the "parity unpinned" note.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * No FMA is ever formed, every operation is a single correctly rounded fp32
 * IEEE-754 operation, evaluated left to right exactly as the Metal source is
 * written (MSL has no double; all literals are float).
 *
 * Math-library sites and how they are restated:
 *   length(v)  -> sqrtf(x*x + y*y)            (correctly rounded sqrt)
 *   exp(x)     -> (float)exp((double)x)       (correctly rounded fp32 exp in
 *                 all but ~1e-9 of cases; the reference itself runs with
 *                 MTL_FAST_MATH = YES, project.pbxproj:256, i.e. an approximate
 *                 exp, so this is the strict-IEEE reading of the same formula)
 *   a / b      -> IEEE fp32 division
 *   mix(x,y,a) -> x + (y - x)*a               (MSL definition)
 *   clamp(x,lo,hi) -> fminf(fmaxf(x,lo),hi)
 */
#include "fluid_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define AT(buf, g, x, y) (buf)[(size_t)((y) - (g)->y0) * (size_t)(g)->W + (size_t)(x)]

/* "Boundary" predicate shared by the stencil kernels, e.g. Shaders:91.
 * width/height are uint in MSL, so width-1 wraps for width==0; sizes here are >= 1. */
static inline int is_boundary(const fo_grid *g, int x, int y)
{
    return x == 0 || x >= g->W - 1 || y == 0 || y >= g->H - 1;
}

int fo_omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Shaders:43-69 addForces.  No boundary rule: the splat also writes boundary cells. */
void fo_addForces(const fo_grid *g, float *u, float *v, float *r, float *gch, float *b,
                  float px, float py, float fx, float fy,
                  float cr, float cg, float cb, float radius, float dt)
{
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            float dx = (float)x - px;             /* :52-53 */
            float dy = (float)y - py;
            float dist = sqrtf(dx * dx + dy * dy); /* :54 length(diff) */
            if (dist < radius) {                   /* :56 */
                /* :57 exp(-dist * dist / (2.0 * radius * radius)) */
                float num = (-dist) * dist;
                float den = (2.0f * radius) * radius;
                float infl = (float)exp((double)(num / den));
                /* :61 vel.xy += force * influence * deltaTime */
                AT(u, g, x, y) = AT(u, g, x, y) + (fx * infl) * dt;
                AT(v, g, x, y) = AT(v, g, x, y) + (fy * infl) * dt;
                /* :66 dens += color * influence * deltaTime */
                AT(r, g, x, y) = AT(r, g, x, y) + (cr * infl) * dt;
                AT(gch, g, x, y) = AT(gch, g, x, y) + (cg * infl) * dt;
                AT(b, g, x, y) = AT(b, g, x, y) + (cb * infl) * dt;
            }
        }
    }
}

/* Shaders:82-105 diffuse (one plane).  RHS is the current iterate, not the
 * original field -- reproduced as written, not "fixed" to Stam's Jacobi. */
void fo_diffuse(const fo_grid *g, const float *in, float *out, float viscosity, float dt)
{
    /* :96 a = deltaTime * viscosity * width * height (uint -> float conversions) */
    const float a = ((dt * viscosity) * (float)(unsigned)g->W) * (float)(unsigned)g->H;
    const float den = 1.0f + 4.0f * a;                       /* :103 */
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            if (is_boundary(g, x, y)) {                      /* :91-94 copy */
                AT(out, g, x, y) = AT(in, g, x, y);
                continue;
            }
            float c = AT(in, g, x, y);
            float l = AT(in, g, x - 1, y);
            float rr = AT(in, g, x + 1, y);
            float t = AT(in, g, x, y - 1);
            float bo = AT(in, g, x, y + 1);
            /* :103 (center + a * (left + right + top + bottom)) / (1.0 + 4.0 * a) */
            AT(out, g, x, y) = (c + a * (((l + rr) + t) + bo)) / den;
        }
    }
}

/* Shaders:118-151 advect (one plane of `in`, traced through velocity u,v). */
long fo_advect(const fo_grid *g, const float *u, const float *v,
               const float *in, float *out, float dt)
{
    long outside = 0;
    const float hx = (float)(unsigned)g->W - 1.5f;           /* :136 width - 1.5f */
    const float hy = (float)(unsigned)g->H - 1.5f;           /* :137 */
#pragma omp parallel for schedule(static) reduction(+ : outside)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            if (is_boundary(g, x, y)) {                      /* :127-130 copy */
                AT(out, g, x, y) = AT(in, g, x, y);
                continue;
            }
            float posx = (float)x - AT(u, g, x, y) * dt;     /* :132-133 */
            float posy = (float)y - AT(v, g, x, y) * dt;
            posx = fminf(fmaxf(posx, 0.5f), hx);             /* :136 */
            posy = fminf(fmaxf(posy, 0.5f), hy);             /* :137 */
            float flx = floorf(posx), fly = floorf(posy);    /* :140 */
            int i0x = (int)flx, i0y = (int)fly;
            int i1x = i0x + 1, i1y = i0y + 1;                /* :141 */
            float sx = posx - (float)i0x;                    /* :142 */
            float sy = posy - (float)i0y;
            if (i0y < g->y0 || i1y >= g->y0 + g->rows) {     /* slab halo too small */
                outside++;
                AT(out, g, x, y) = NAN;
                continue;
            }
            float q00 = AT(in, g, i0x, i0y);                 /* :144-147 */
            float q10 = AT(in, g, i1x, i0y);
            float q01 = AT(in, g, i0x, i1y);
            float q11 = AT(in, g, i1x, i1y);
            float m0 = q00 + (q10 - q00) * sx;               /* :149 mix(q00,q10,s.x) */
            float m1 = q01 + (q11 - q01) * sx;               /*      mix(q01,q11,s.x) */
            AT(out, g, x, y) = m0 + (m1 - m0) * sy;          /*      mix(.,.,s.y)     */
        }
    }
    return outside;
}

/* Shaders:162-181 divergence. */
void fo_divergence(const fo_grid *g, const float *u, const float *v, float *div)
{
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            if (is_boundary(g, x, y)) { AT(div, g, x, y) = 0.0f; continue; } /* :169-172 */
            float lx = AT(u, g, x - 1, y), rx = AT(u, g, x + 1, y);
            float ty = AT(v, g, x, y - 1), by = AT(v, g, x, y + 1);
            AT(div, g, x, y) = -0.5f * ((rx - lx) + (by - ty));              /* :179 */
        }
    }
}

/* Shaders:193-214 pressureSolve.  Boundary is Dirichlet p = 0 (:201-204). */
void fo_pressureSolve(const fo_grid *g, const float *div, const float *p, float *pout)
{
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            if (is_boundary(g, x, y)) { AT(pout, g, x, y) = 0.0f; continue; }
            float d = AT(div, g, x, y);
            float l = AT(p, g, x - 1, y), rr = AT(p, g, x + 1, y);
            float t = AT(p, g, x, y - 1), bo = AT(p, g, x, y + 1);
            AT(pout, g, x, y) = ((((d + l) + rr) + t) + bo) / 4.0f;          /* :212 */
        }
    }
}

/* Shaders:225-245 subtractGradient (in place; boundary untouched :232-234). */
void fo_subtractGradient(const fo_grid *g, const float *p, float *u, float *v)
{
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            if (is_boundary(g, x, y)) continue;
            float l = AT(p, g, x - 1, y), rr = AT(p, g, x + 1, y);
            float t = AT(p, g, x, y - 1), bo = AT(p, g, x, y + 1);
            AT(u, g, x, y) = AT(u, g, x, y) - 0.5f * (rr - l);               /* :242 */
            AT(v, g, x, y) = AT(v, g, x, y) - 0.5f * (bo - t);               /* :243 */
        }
    }
}

/* Shaders:256-275 calculateVorticity.  curl = du/dy - dv/dx (code, not README). */
void fo_calculateVorticity(const fo_grid *g, const float *u, const float *v, float *w)
{
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            if (is_boundary(g, x, y)) { AT(w, g, x, y) = 0.0f; continue; }   /* :263-266 */
            float ly = AT(v, g, x - 1, y), ry = AT(v, g, x + 1, y);
            float tx = AT(u, g, x, y - 1), bx = AT(u, g, x, y + 1);
            AT(w, g, x, y) = (bx - tx) - (ry - ly);                          /* :273 */
        }
    }
}

/* Shaders:288-316 applyVorticity (in place; boundary untouched :297-299). */
void fo_applyVorticity(const fo_grid *g, const float *w, float *u, float *v,
                       float strength, float dt)
{
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y) {
        for (int x = 0; x < g->W; ++x) {
            if (is_boundary(g, x, y)) continue;
            float l = fabsf(AT(w, g, x - 1, y));                             /* :301-304 */
            float rr = fabsf(AT(w, g, x + 1, y));
            float t = fabsf(AT(w, g, x, y - 1));
            float bo = fabsf(AT(w, g, x, y + 1));
            float c = AT(w, g, x, y);                                        /* :305 */
            float gx = rr - l, gy = bo - t;                                  /* :307 */
            float glen = sqrtf(gx * gx + gy * gy) + 0.00001f;                /* :308 */
            gx = gx / glen;                                                  /* :309 */
            gy = gy / glen;
            float sc = strength * c;                                         /* :311 */
            float fx = sc * gy;
            float fy = sc * (-gx);
            AT(u, g, x, y) = AT(u, g, x, y) + fx * dt;                       /* :314 */
            AT(v, g, x, y) = AT(v, g, x, y) + fy * dt;
        }
    }
}

/* Shaders:326-334 fadeDensity (one plane, all cells). */
void fo_fadeDensity(const fo_grid *g, float *d, float fadeRate)
{
    const float k = 1.0f - fadeRate;                                         /* :331 */
#pragma omp parallel for schedule(static)
    for (int y = g->yb; y < g->ye; ++y)
        for (int x = 0; x < g->W; ++x)
            AT(d, g, x, y) = fmaxf(AT(d, g, x, y) * k, 0.0f);                /* :331-332 */
}

/* Shaders:343-346 clearTexture. */
void fo_clear(const fo_grid *g, float *t)
{
    for (int y = g->yb; y < g->ye; ++y)
        for (int x = 0; x < g->W; ++x) AT(t, g, x, y) = 0.0f;
}

static inline size_t plane_sz(const fo_grid *g) { return (size_t)g->rows * (size_t)g->W; }

/* Shaders:357-366 renderDensity. */
void fo_renderDensity(const fo_grid *g, const float *r, const float *gch, const float *b,
                      float *out4, float brightness)
{
    size_t n = plane_sz(g);
    for (int y = g->yb; y < g->ye; ++y)
        for (int x = 0; x < g->W; ++x) {
            AT(out4, g, x, y) = AT(r, g, x, y) * brightness;                 /* :363 */
            AT(out4 + n, g, x, y) = AT(gch, g, x, y) * brightness;
            AT(out4 + 2 * n, g, x, y) = AT(b, g, x, y) * brightness;
            AT(out4 + 3 * n, g, x, y) = 1.0f;                                /* :364 */
        }
}

/* Shaders:378-394 renderVelocity. */
void fo_renderVelocity(const fo_grid *g, const float *u, const float *v, float *out4, float scale)
{
    size_t n = plane_sz(g);
    for (int y = g->yb; y < g->ye; ++y)
        for (int x = 0; x < g->W; ++x) {
            float vx = AT(u, g, x, y), vy = AT(v, g, x, y);
            float speed = sqrtf(vx * vx + vy * vy);                          /* :384 */
            AT(out4, g, x, y) = vx * scale + 0.5f;                           /* :389 */
            AT(out4 + n, g, x, y) = vy * scale + 0.5f;                       /* :390 */
            AT(out4 + 2 * n, g, x, y) = speed * scale;                       /* :391 */
            AT(out4 + 3 * n, g, x, y) = 1.0f;
        }
}

/* Shaders:405-414 renderSpeed. */
void fo_renderSpeed(const fo_grid *g, const float *u, const float *v, float *out4, float scale)
{
    size_t n = plane_sz(g);
    for (int y = g->yb; y < g->ye; ++y)
        for (int x = 0; x < g->W; ++x) {
            float vx = AT(u, g, x, y), vy = AT(v, g, x, y);
            float s = sqrtf(vx * vx + vy * vy) * scale;                      /* :411 */
            AT(out4, g, x, y) = s;
            AT(out4 + n, g, x, y) = s;
            AT(out4 + 2 * n, g, x, y) = s;
            AT(out4 + 3 * n, g, x, y) = 1.0f;
        }
}

/* ------------------------------------------------------------------ */
/* Whole-solver restatement (Solver:322-374, :444-470, :477-591).      */
/* ------------------------------------------------------------------ */
struct fo_state {
    int W, H;
    fo_grid g;
    float *u, *v, *ut, *vt;          /* velocityTexture / velocityTempTexture (xy) */
    float *d[3], *dt_[3];            /* densityTexture / densityTempTexture (rgb)  */
    float *p, *pt, *div, *vort;      /* pressure / pressureTemp / divergence / vorticity */
};

static float *plane_alloc(int W, int H) { return (float *)calloc((size_t)W * (size_t)H, sizeof(float)); }

fo_state *fo_create(int W, int H)
{
    fo_state *s = (fo_state *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->W = W; s->H = H;
    s->g = (fo_grid){W, H, 0, H, 0, H};
    s->u = plane_alloc(W, H); s->v = plane_alloc(W, H);
    s->ut = plane_alloc(W, H); s->vt = plane_alloc(W, H);
    for (int c = 0; c < 3; ++c) { s->d[c] = plane_alloc(W, H); s->dt_[c] = plane_alloc(W, H); }
    s->p = plane_alloc(W, H); s->pt = plane_alloc(W, H);
    s->div = plane_alloc(W, H); s->vort = plane_alloc(W, H);
    return s;
}

void fo_destroy(fo_state *s)
{
    if (!s) return;
    free(s->u); free(s->v); free(s->ut); free(s->vt);
    for (int c = 0; c < 3; ++c) { free(s->d[c]); free(s->dt_[c]); }
    free(s->p); free(s->pt); free(s->div); free(s->vort);
    free(s);
}

/* Solver:348-366 clearTextures (all eight simulation textures) via reset() :371-374. */
void fo_reset(fo_state *s)
{
    size_t n = (size_t)s->W * (size_t)s->H * sizeof(float);
    memset(s->u, 0, n); memset(s->v, 0, n); memset(s->ut, 0, n); memset(s->vt, 0, n);
    for (int c = 0; c < 3; ++c) { memset(s->d[c], 0, n); memset(s->dt_[c], 0, n); }
    memset(s->p, 0, n); memset(s->pt, 0, n); memset(s->div, 0, n); memset(s->vort, 0, n);
}

/* Solver:444-470 applyForce: one full-grid addForces dispatch. */
void fo_add_force(fo_state *s, float px, float py, float fx, float fy,
                  float cr, float cg, float cb, float radius, float dt)
{
    fo_addForces(&s->g, s->u, s->v, s->d[0], s->d[1], s->d[2], px, py, fx, fy, cr, cg, cb, radius, dt);
}

#define SWAPP(a, b) do { float *t_ = (a); (a) = (b); (b) = t_; } while (0)

/* Solver:477-591 update(), simulation part (the render dispatch :594-623 is separate). */
void fo_update(fo_state *s, const fo_params *pr)
{
    const fo_grid *g = &s->g;
    const float dt = pr->deltaTime;
    const int N = pr->solverIterations;

    if (pr->vorticityStrength > 0.0f) {                                    /* :486 */
        fo_calculateVorticity(g, s->u, s->v, s->vort);                     /* :488-491 */
        fo_applyVorticity(g, s->vort, s->u, s->v, pr->vorticityStrength, dt); /* :494-501 */
    }
    if (pr->viscosity > 0.0f) {                                            /* :505 */
        for (int i = 0; i < N; ++i) {                                      /* :506 */
            fo_diffuse(g, s->u, s->ut, pr->viscosity, dt);                 /* :507-514 */
            fo_diffuse(g, s->v, s->vt, pr->viscosity, dt);
            SWAPP(s->u, s->ut); SWAPP(s->v, s->vt);                        /* :516 */
        }
    }
    fo_divergence(g, s->u, s->v, s->div);                                  /* :522-525 */
    fo_clear(g, s->p);                                                     /* :528-530 */
    for (int i = 0; i < N; ++i) {                                          /* :533 */
        fo_pressureSolve(g, s->div, s->p, s->pt);                          /* :534-538 */
        SWAPP(s->p, s->pt);                                                /* :540 */
    }
    fo_subtractGradient(g, s->p, s->u, s->v);                              /* :544-547 */
    fo_advect(g, s->u, s->v, s->u, s->ut, dt);                             /* :550-556 */
    fo_advect(g, s->u, s->v, s->v, s->vt, dt);
    SWAPP(s->u, s->ut); SWAPP(s->v, s->vt);                                /* :558 */
    if (pr->diffusion > 0.0f) {                                            /* :562 */
        for (int i = 0; i < N; ++i) {                                      /* :563 */
            for (int c = 0; c < 3; ++c) fo_diffuse(g, s->d[c], s->dt_[c], pr->diffusion, dt);
            for (int c = 0; c < 3; ++c) SWAPP(s->d[c], s->dt_[c]);         /* :572 */
        }
    }
    for (int c = 0; c < 3; ++c) fo_advect(g, s->u, s->v, s->d[c], s->dt_[c], dt); /* :577-582 */
    for (int c = 0; c < 3; ++c) SWAPP(s->d[c], s->dt_[c]);                 /* :584 */
    for (int c = 0; c < 3; ++c) fo_fadeDensity(g, s->d[c], pr->fadeSpeed); /* :587-591 */
}

float *fo_field(fo_state *s, int field, int c)
{
    switch (field) {
    case FO_VELOCITY: return c == 0 ? s->u : (c == 1 ? s->v : NULL);
    case FO_DENSITY: return (c >= 0 && c < 3) ? s->d[c] : NULL;
    case FO_PRESSURE: return c == 0 ? s->p : NULL;
    case FO_DIVERGENCE: return c == 0 ? s->div : NULL;
    case FO_VORTICITY: return c == 0 ? s->vort : NULL;
    }
    return NULL;
}

/* ------------------------------------------------------------------ */
/* Tracers: Part:242-304.  The velocity query is the nearest-cell rule */
/* of Solver:648-657 applied to the real field (the reference's cache  */
/* is never filled, Solver:639,669-672; SURVEY.md 8(a) a15/a16 records */
/* the build's definition: fluidVel = vel_cells / (W,H), real grid     */
/* size).  Dead particles (life == 0) stay in place and are skipped;   */
/* a particle that dies this update keeps its updated pos/vel/age and  */
/* gets life = 0 (the reference removes it, Part:296).                 */
/* ------------------------------------------------------------------ */
void fo_particles_update(const fo_grid *g, const float *u, const float *v,
                         const fo_particle_params *pp, long n,
                         float *px, float *py, float *vx, float *vy,
                         float *life, float *mass, float *age)
{
    const float invW = 1.0f / pp->windowW, invH = 1.0f / pp->windowH;       /* Part:245-248 */
    const float gw = (float)g->W, gh = (float)g->H;
    /* Part:262-268 */
    const float eff = pp->linkToViscosity ? 0.1f + (pp->viscosity / 0.01f) * 0.9f : pp->fluidForce;
    const float margin = 50.0f;                                              /* Part:282 */
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        if (life[i] == 0.0f) continue;
        float nx = px[i] * invW, ny = py[i] * invH;                          /* Part:255 */
        /* Solver:650-651 */
        int cx = (int)fmaxf(0.0f, fminf(gw - 1.0f, nx * gw));
        int cy = (int)fmaxf(0.0f, fminf(gh - 1.0f, ny * gh));
        float fvx = AT(u, g, cx, cy) / gw;   /* build's normalisation, see header */
        float fvy = AT(v, g, cx, cy) / gh;
        float ageDecay = fmaxf(0.0f, 1.0f - age[i] * pp->fluidDecayRate);    /* Part:259 */
        float k = (mass[i] * eff) * ageDecay;                                /* Part:271 */
        float fcx = (fvx * k) * pp->windowW;
        float fcy = (fvy * k) * pp->windowH;
        float em = pp->momentum + (1.0f - pp->momentum) * fminf(1.0f, age[i] * 2.0f); /* Part:272 */
        float nvx = fcx + vx[i] * em;                                        /* Part:273-275 */
        float nvy = fcy + vy[i] * em;
        float npx = px[i] + nvx;                                             /* Part:276 */
        float npy = py[i] + nvy;
        float nage = age[i] + 1.0f / 60.0f;                                  /* Part:279 */
        float l = life[i];
        if (npx < -margin || npx > pp->windowW + margin ||
            npy < -margin || npy > pp->windowH + margin)                     /* Part:283-288 */
            l = 0.0f;
        l = l * 0.999f;                                                      /* Part:291 */
        if (l < 0.01f) l = 0.0f;                                             /* Part:294-295 */
        vx[i] = nvx; vy[i] = nvy; px[i] = npx; py[i] = npy; age[i] = nage; life[i] = l;
    }
}

/* ------------------------------------------------------------------ */
/* Tracer line vertices: vertex shader particleVertex, Shaders:446-476. */
/* Two vertices per particle, start = pos - vel * particleScale (:462)  */
/* and end = pos (:464), each mapped to NDC (pos / viewport) * 2 - 1    */
/* with y flipped (:468-469); colour = white with alpha = life (:472).  */
/* out_xy = n x (x0, y0, x1, y1); out_alpha = n.                        */
/* ------------------------------------------------------------------ */
void fo_particle_lines(long n, const float *px, const float *py, const float *vx, const float *vy,
                       const float *life, float viewW, float viewH, float particleScale,
                       float *out_xy, float *out_alpha)
{
    for (long i = 0; i < n; ++i) {
        float sx = px[i] - vx[i] * particleScale, sy = py[i] - vy[i] * particleScale;   /* :462 */
        out_xy[4 * i + 0] = (sx / viewW) * 2.0f - 1.0f;                                  /* :468 */
        out_xy[4 * i + 1] = -((sy / viewH) * 2.0f - 1.0f);                               /* :469 */
        out_xy[4 * i + 2] = (px[i] / viewW) * 2.0f - 1.0f;                               /* :464, :468 */
        out_xy[4 * i + 3] = -((py[i] / viewH) * 2.0f - 1.0f);
        out_alpha[i] = life[i];                                                          /* :472 */
    }
}

/* ------------------------------------------------------------------ */
unsigned fo_xorshift32(unsigned *state)
{
    unsigned x = *state;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    *state = x;
    return x;
}

/* U[lo,hi): 24 random bits -> [0,1) exactly representable, then affine in fp32. */
static float spawn_draw(unsigned *st, float lo, float hi)
{
    float r01 = (float)(fo_xorshift32(st) >> 8) * (1.0f / 16777216.0f);
    return lo + (hi - lo) * r01;
}

/* Part:186-231, see fluid_oracle.h */
void fo_particles_spawn(long cap, long *n, float *px, float *py, float *vx, float *vy, float *life, float *mass, float *age,
                        float x, float y, long count, float radius, unsigned seed)
{
    float *a[7] = {px, py, vx, vy, life, mass, age};
    long m = 0;
    for (long i = 0; i < *n; ++i) {            /* the reference's array holds no dead tracers (Part:294-296) */
        if (life[i] == 0.0f) continue;
        for (int k = 0; k < 7; ++k) a[k][m] = a[k][i];
        ++m;
    }
    for (long j = 0; j < count; ++j) {         /* Part:195 */
        if (cap <= 0) break;
        if (m >= cap) {                        /* Part:196-199 removeFirst() */
            for (int k = 0; k < 7; ++k)
                for (long i = 1; i < m; ++i) a[k][i - 1] = a[k][i];
            --m;
        }
        unsigned st = seed ^ ((unsigned)(j + 1) * 0x9E3779B9u);
        if (!st) st = 1u;
        const float ox = spawn_draw(&st, -radius, radius);       /* Part:202-205 */
        const float oy = spawn_draw(&st, -radius, radius);
        const float alpha = spawn_draw(&st, 0.3f, 1.0f);         /* Part:211 */
        const float ms = spawn_draw(&st, 0.1f, 1.0f);            /* Part:212 */
        px[m] = x + ox; py[m] = y + oy;                          /* Part:218 */
        vx[m] = 0.0f; vy[m] = 0.0f;                              /* Part:208 */
        life[m] = alpha; mass[m] = ms; age[m] = 0.0f;            /* Part:221-223 */
        ++m;
    }
    *n = m;
}

void fo_fill_uniform(float *dst, long n, float lo, float hi, unsigned seed)
{
    unsigned st = seed ? seed : 1u;
    for (long i = 0; i < n; ++i) {
        float r01 = (float)(fo_xorshift32(&st) >> 8) * (1.0f / 16777216.0f);
        dst[i] = lo + (hi - lo) * r01;
    }
}
