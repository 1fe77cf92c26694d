/*
 * fluid_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A scalar, strict-IEEE fp32 restatement of the reference's per-frame
 * stable-fluids step.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path
 * (fluidsolver_amd/, libfluidsolver_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference (an Apple-only SwiftUI + Metal app) ships no
 * tests, golden vectors or fixtures (its Xcode project has a single application
 * target, fluidSolver.xcodeproj/project.pbxproj:118-136) and cannot be built or
 * run here (Metal/Swift toolchains absent).  The oracle is therefore pinned
 * only by the analytic known-answer tests in tests/test_oracle_kat.py, which
 * are derived line by line from the shader arithmetic cited below.
 *
 * Citations are into /root/reference/:
 *   Shaders = fluidSolver/Shaders/FluidShaders.metal
 *   Solver  = fluidSolver/Models/FluidSolver.swift
 *   Part    = fluidSolver/Models/ParticleSystem.swift
 *
 * Data layout (the build's choice, SURVEY.md 2.2 notes): planar fp32, one
 * dense row-major plane per channel (velocity = u,v; dye = r,g,b; scalars p,
 * div, vort).  The reference's rgba32Float texels carry dead channels
 * (velocity .zw == 0 always, dye alpha never observable: Shaders:364) which
 * are dropped.
 *
 * Slab form: every kernel takes an fo_grid describing which global rows the
 * buffers hold and which rows to compute, so the same code checks the
 * multi-GPU row-slab path.  The full grid is {W,H,0,H,0,H}.
 */
#ifndef FLUID_ORACLE_H
#define FLUID_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fo_grid {
    int W, H;     /* global domain size in cells                         */
    int y0, rows; /* buffers hold global rows [y0, y0+rows), pitch == W  */
    int yb, ye;   /* compute output rows [yb, ye) (global), others untouched */
} fo_grid;

/* Solver parameters: the @Published scalars of Solver:139-202 that update() reads. */
typedef struct fo_params {
    float viscosity;         /* Solver:139, default 0.00015 */
    float diffusion;         /* Solver:146, default 0.0     */
    float deltaTime;         /* Solver:153, default 0.5     */
    float vorticityStrength; /* Solver:160, default 0.0     */
    float fadeSpeed;         /* Solver:167, default 0.002   */
    int   solverIterations;  /* Solver:174, default 10      */
} fo_params;

/* Tracer parameters: Part:63-111 plus the solver viscosity read at Part:265. */
typedef struct fo_particle_params {
    float momentum;        /* Part:63,  default 0.5 */
    float fluidForce;      /* Part:70,  default 0.6 */
    float fluidDecayRate;  /* Part:98,  default 3.0 */
    int   linkToViscosity; /* Part:104, default true */
    float viscosity;       /* FluidSolver.viscosity, read at Part:265 */
    float windowW, windowH;/* windowSize argument of Part:242 */
} fo_particle_params;

/* ---- per-kernel restatements (one per reference kernel name) ---- */
void fo_addForces(const fo_grid *g, float *u, float *v, float *r, float *gch, float *b,
                  float px, float py, float fx, float fy,
                  float cr, float cg, float cb, float radius, float dt);
void fo_diffuse(const fo_grid *g, const float *in, float *out, float viscosity, float dt);
/* returns the number of taps that fell outside the buffered rows (0 == halo sufficient) */
long fo_advect(const fo_grid *g, const float *u, const float *v,
               const float *in, float *out, float dt);
void fo_divergence(const fo_grid *g, const float *u, const float *v, float *div);
void fo_pressureSolve(const fo_grid *g, const float *div, const float *p, float *pout);
void fo_subtractGradient(const fo_grid *g, const float *p, float *u, float *v);
void fo_calculateVorticity(const fo_grid *g, const float *u, const float *v, float *w);
void fo_applyVorticity(const fo_grid *g, const float *w, float *u, float *v,
                       float strength, float dt);
void fo_fadeDensity(const fo_grid *g, float *d, float fadeRate);
void fo_clear(const fo_grid *g, float *t);
/* render kernels: out is 4 planes R,G,B,A of the same geometry */
void fo_renderDensity(const fo_grid *g, const float *r, const float *gch, const float *b,
                      float *out4, float brightness);
void fo_renderVelocity(const fo_grid *g, const float *u, const float *v, float *out4, float scale);
void fo_renderSpeed(const fo_grid *g, const float *u, const float *v, float *out4, float scale);

/* ---- whole-solver restatement of FluidSolver (full grid) ---- */
typedef struct fo_state fo_state;
enum { FO_VELOCITY = 0, FO_DENSITY = 1, FO_PRESSURE = 2, FO_DIVERGENCE = 3, FO_VORTICITY = 4 };

fo_state *fo_create(int W, int H);
void fo_destroy(fo_state *s);
void fo_reset(fo_state *s);                               /* Solver:348-374 */
void fo_add_force(fo_state *s, float px, float py, float fx, float fy,
                  float cr, float cg, float cb, float radius, float dt); /* Solver:444-470 */
void fo_update(fo_state *s, const fo_params *p);          /* Solver:477-591 */
/* plane c of field id (velocity 2 planes, density 3, others 1); points at current data */
float *fo_field(fo_state *s, int field, int c);

/* ---- tracers (SoA), Part:242-304 with the lookup of Solver:648-663 ---- */
void fo_particles_update(const fo_grid *g, const float *u, const float *v,
                         const fo_particle_params *pp, long n,
                         float *px, float *py, float *vx, float *vy,
                         float *life, float *mass, float *age);

/* Shaders:446-476 particleVertex: line vertices (pos - vel*scale, pos) in NDC and alpha = life */
void fo_particle_lines(long n, const float *px, const float *py, const float *vx, const float *vy,
                       const float *life, float viewW, float viewH, float particleScale,
                       float *out_xy, float *out_alpha);

/* ParticleSystem.addParticles(at:count:windowSize:), Part:186-231, on SoA arrays that may hold dead slots (life == 0; the
 * reference removes a tracer when it dies, Part:296, so its array is the live tracers in age order).  *n slots are in
 * use, `cap` = maxParticles (Part:53).  Literally: drop the dead slots (order kept), then for each of `count` new
 * tracers: if the array is full remove its FIRST element (:196-199), append {pos + U(-r, r)^2, vel 0, life = alpha ~
 * U(0.3, 1), mass ~ U(0.1, 1), age 0} (:202-226).  The reference draws from the system RNG; here new tracer j draws its
 * four numbers, in the order offset x, offset y, alpha, mass, from an xorshift32 stream of its own seeded with
 * seed ^ (j + 1) * 0x9E3779B9 (0 -> 1), each as lo + (hi - lo) * ((x >> 8) * 2^-24) in fp32.  Returns the new *n. */
void fo_particles_spawn(long cap, long *n, float *px, float *py, float *vx, float *vy, float *life, float *mass, float *age,
                        float x, float y, long count, float radius, unsigned seed);

/* xorshift32 harness RNG (SURVEY 8(d) config 3 / Jacobi micro-benchmark). */
unsigned fo_xorshift32(unsigned *state);
void fo_fill_uniform(float *dst, long n, float lo, float hi, unsigned seed);

int fo_omp_threads(void);

#ifdef __cplusplus
}
#endif
#endif
