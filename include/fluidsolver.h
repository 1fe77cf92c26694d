/*
 * fluidsolver.h -- C ABI of libfluidsolver_hip.so
 *
 * MI355X (gfx950) implementation of the per-frame stable-fluids step of
 * aidancornelius/fluidSolver: FluidSolver.update() + applyForce() + the tracer
 * update.  The reference has no FFI; its boundary is the Swift -> Metal call
 * pattern (13 compute kernels looked up BY NAME, FluidSolver.swift:288-300, then
 * setTexture/setBytes/dispatchThreads per stage, e.g. :534-538).  This header is
 * what a host binds instead of those Metal dispatches.  Citations are into
 * /root/reference/:
 *     Shaders = fluidSolver/Shaders/FluidShaders.metal
 *     Solver  = fluidSolver/Models/FluidSolver.swift
 *     Part    = fluidSolver/Models/ParticleSystem.swift
 *
 * Two layers:
 *   1. fs_k_*   one entry per reference kernel name, on caller-owned DEVICE
 *               buffers (plain pointers + an fs_grid), asynchronous on a HIP
 *               stream.  A host can swap a single stage.  The *_n forms run n
 *               Jacobi sweeps temporally blocked (same per-sweep arithmetic).
 *   2. fs_*     an opaque solver context that owns all device memory and runs
 *               the whole update() as a captured hipGraph.
 *
 * Conventions
 *   - Every function returns 0 (FS_OK) or a negative FS_ERR_* code and never
 *     aborts the process (the reference fatalError()s on set-up failure,
 *     Solver:270,277,302, and silently returns on encode failure, :349,:445,:478).
 *   - All arithmetic is strict IEEE fp32 (no FMA contraction, correctly rounded
 *     divide/sqrt; the splat's exp is evaluated in fp64 and rounded once).
 *   - Fields are PLANAR fp32: velocity = 2 planes (u,v), dye = 3 planes (r,g,b),
 *     pressure / divergence / vorticity = 1 plane.  The reference's rgba32Float
 *     texels (Solver:323-330) carry dead channels (velocity .zw, dye alpha) which
 *     are dropped.
 *   - Neighbour naming follows the shaders: left=(x-1,y) right=(x+1,y)
 *     top=(x,y-1) bottom=(x,y+1).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Calls are asynchronous and execute in call order on their stream, like the
 *     reference's single MTLCommandQueue (Solver:273).  A context is not
 *     thread-safe; callers serialise (the reference lives on the main thread).
 */
#ifndef FLUIDSOLVER_H
#define FLUIDSOLVER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_OK            0
#define FS_ERR_INVALID  -1   /* bad argument / shape */
#define FS_ERR_HIP      -2   /* a HIP runtime call failed (see fs_last_error) */
#define FS_ERR_NOMEM    -3
#define FS_ERR_HALO     -4   /* advect backtrace left the buffered rows (slab halo too small) */
#define FS_ERR_STATE    -5

/* Geometry of one plane as it lies in device memory.
 * A single GPU holds the whole grid: {W, H, pitch, 0, H, 0, H}.
 * A row slab of a multi-GPU run holds global rows [y0, y0+rows) (owned rows plus
 * ghost rows) and computes output rows [yb, ye); boundary tests always use the
 * GLOBAL width/height, exactly like gid vs get_width()/get_height() in the
 * shaders (e.g. Shaders:91). */
typedef struct fs_grid {
    int width, height; /* global domain, cells */
    int pitch;         /* floats per buffer row; multiple of 64 (256-byte rows) and >= width */
    int y0, rows;      /* buffer row 0 is global row y0; the buffer holds `rows` rows */
    int yb, ye;        /* rows to compute, global, y0 <= yb <= ye <= y0+rows */
} fs_grid;

/* smallest legal pitch for a width */
int fs_pitch_for_width(int width);

/* Process-wide DEFAULT of the Jacobi tuning knobs (keys: FS_OPT_BLOCK_DEPTH = sweeps fused per Jacobi launch, 1..48;
 * FS_OPT_BLOCK_DEPTH_PRESSURE / _DIFFUSE; FS_OPT_TILE_VARIANT / _DIFFUSE; FS_OPT_STAGGER / _MODE): what the per-kernel entry
 * points fs_k_*_n use and what a context created afterwards starts from.  A context keeps a copy of its own, changed
 * with fs_set_option, so two contexts in one process never change each other's tuning.  No reference counterpart. */
int fs_config_set(int key, int value);
int fs_config_get(int key);

/* -------------------------------------------------------------------------
 * Layer 1: per-kernel entry points (device pointers).
 * Multi-plane arguments take the pointer of plane 0 and `plane_stride` floats
 * between planes.
 * ---------------------------------------------------------------------- */

/* replaces kernel addForces, Shaders:43-69 (buffers 0-4: position, force, color,
 * radius, deltaTime; Solver:444-470).  Launched on the splat's bounding box only. */
int fs_k_addForces(const fs_grid *g, float *u, float *v, float *r, float *gch, float *b,
                   float pos_x, float pos_y, float force_x, float force_y,
                   float color_r, float color_g, float color_b,
                   float radius, float deltaTime, void *stream);

/* replaces kernel diffuse, Shaders:82-105 (buffer 0 viscosity, 1 deltaTime), one sweep */
int fs_k_diffuse(const fs_grid *g, const float *in, float *out, int nplanes, size_t plane_stride,
                 float viscosity, float deltaTime, void *stream);

/* n sweeps of diffuse, temporally blocked (the loop of Solver:506-517 / :563-573).
 * Ping-pongs between a (holds the input) and b; *result_in_b says where the result is.
 * Needs input rows [yb-n, ye+n) (clipped to the domain) present in the buffer. */
int fs_k_diffuse_n(const fs_grid *g, float *a, float *b, int nplanes, size_t plane_stride,
                   float viscosity, float deltaTime, int n, int *result_in_b, void *stream);

/* replaces kernel advect, Shaders:118-151 (buffer 0 deltaTime).  `in` may alias u/v
 * planes (velocity self-advection, Solver:550-556).  err_flag (device int, may be
 * NULL) is set non-zero if a tap needed a row outside the buffer (slab halo too small;
 * the access is clamped, never out of bounds). */
int fs_k_advect(const fs_grid *g, const float *u, const float *v, const float *in, float *out,
                int nplanes, size_t plane_stride, float deltaTime, int *err_flag, void *stream);

/* advect followed by fadeDensity on the advected field, in one pass: replaces the two dispatches
 * Solver:577-584 + :587-591 (kernels advect, Shaders:118-151, and fadeDensity, Shaders:326-334).
 * Same per-cell operations in the same order, so the result equals fs_k_advect + fs_k_fadeDensity. */
int fs_k_advect_fade(const fs_grid *g, const float *u, const float *v, const float *in, float *out,
                     int nplanes, size_t plane_stride, float deltaTime, float fadeRate,
                     int *err_flag, void *stream);

/* Both advections of update() in one pass: the velocity by itself (Solver:550-558) into u_out/v_out, then the
 * three dye planes through that NEW velocity (Solver:577-584), optionally followed by fadeDensity (:587-591).  Same
 * bits as fs_k_advect(velocity) + fs_k_advect[_fade](dye); the advected velocity is not read back. */
int fs_k_advect_both(const fs_grid *g, const float *u, const float *v, float *u_out, float *v_out,
                     const float *dye, float *dye_out, size_t dye_stride, float deltaTime,
                     int fade, float fadeRate, int *err_flag, void *stream);

/* replaces kernel divergence, Shaders:162-181 */
int fs_k_divergence(const fs_grid *g, const float *u, const float *v, float *div, void *stream);

/* replaces kernel pressureSolve, Shaders:193-214, one sweep */
int fs_k_pressureSolve(const fs_grid *g, const float *div, const float *p, float *pOut, void *stream);

/* n sweeps of pressureSolve, temporally blocked (the loop of Solver:533-541).
 * If p_is_zero != 0 the contents of pa are ignored and taken as 0, which folds the
 * clearTexture dispatch of Solver:528-530 into the first sweep.  Ping-pongs pa/pb. */
int fs_k_pressureSolve_n(const fs_grid *g, const float *div, float *pa, float *pb, int n,
                         int p_is_zero, int *result_in_b, void *stream);

/* fs_k_pressureSolve_n followed by fs_k_subtractGradient (Solver:533-547) with the gradient subtracted by the
 * LAST launch of the loop from the final iterate while it is in registers: same bits in p, u, v as the two
 * calls, one pass over p, u, v less.  n >= 1; rows [yb-n-1, ye+n+1) of div must be valid (clipped to the domain). */
int fs_k_pressureSolve_n_gradient(const fs_grid *g, const float *div, float *pa, float *pb, int n,
                                  int p_is_zero, float *u, float *v, int *result_in_b, void *stream);

/* The whole projection of update(), Solver:522-547: divergence + clearTexture + n pressureSolve sweeps +
 * subtractGradient on rows [yb,ye), with the divergence computed by the FIRST launch of the loop (and stored to div
 * for the others) and the gradient subtracted by the LAST one.  Same bits in p, u, v on [yb,ye) as the four calls
 * (div is stored on the rows the later launches read).  n >= 1; with a single launch (n <= block depth) the
 * divergence stays a separate pass.  On a row range (slab) the buffer must hold n + 1 rows with valid velocity
 * beyond [yb,ye) (clipped to the domain).  _cfl: also fs_k_max_abs of the new v plane over [yb,ye) into *cfl_dev
 * (device float holding a lower bound, normally 0) -- the slab runtime's CFL bound, no counterpart in the reference. */
int fs_k_project_n(const fs_grid *g, float *u, float *v, float *div, float *pa, float *pb, int n,
                   int *result_in_b, void *stream);
int fs_k_project_n_cfl(const fs_grid *g, float *u, float *v, float *div, float *pa, float *pb, int n,
                       float *cfl_dev, int *result_in_b, void *stream);

/* replaces kernel subtractGradient, Shaders:225-245 (velocity in place) */
int fs_k_subtractGradient(const fs_grid *g, const float *p, float *u, float *v, void *stream);

/* replaces kernel calculateVorticity, Shaders:256-275 */
int fs_k_calculateVorticity(const fs_grid *g, const float *u, const float *v, float *vort, void *stream);

/* replaces kernel applyVorticity, Shaders:288-316 (buffer 0 strength, 1 deltaTime) */
int fs_k_applyVorticity(const fs_grid *g, const float *vort, float *u, float *v,
                        float strength, float deltaTime, void *stream);

/* calculateVorticity + applyVorticity (Solver:486-502) in one pass on rows [yb,ye), OUT OF PLACE (u_out, v_out must
 * not alias u, v); the vorticity plane is neither written nor read.  Same bits in the velocity as the two kernels.
 * Needs two rows of velocity beyond [yb,ye) in the buffer (clipped to the domain). */
int fs_k_vorticity_apply(const fs_grid *g, const float *u, const float *v, float *u_out, float *v_out,
                         float strength, float deltaTime, void *stream);

/* replaces kernel fadeDensity, Shaders:326-334 (buffer 0 fadeRate) */
int fs_k_fadeDensity(const fs_grid *g, float *d, int nplanes, size_t plane_stride,
                     float fadeRate, void *stream);

/* replaces kernel clearTexture, Shaders:343-346 */
int fs_k_clear(const fs_grid *g, float *t, int nplanes, size_t plane_stride, void *stream);

/* replace kernels renderDensity / renderVelocity / renderSpeed, Shaders:357-414.
 * out4 = 4 planes (R,G,B,A) with stride out_stride. */
int fs_k_renderDensity(const fs_grid *g, const float *d, size_t d_stride, int d_planes,
                       float *out4, size_t out_stride, float brightness, void *stream);
int fs_k_renderVelocity(const fs_grid *g, const float *u, const float *v,
                        float *out4, size_t out_stride, float scale, void *stream);
int fs_k_renderSpeed(const fs_grid *g, const float *u, const float *v,
                     float *out4, size_t out_stride, float scale, void *stream);

/* Tracer parameters: the @Published scalars of Part:63-111 that update() reads,
 * the solver viscosity it reads at Part:265 and its windowSize argument (Part:242). */
typedef struct fs_particle_params {
    float momentum;         /* Part:63,  default 0.5 */
    float fluidForce;       /* Part:70,  default 0.6 */
    float fluidDecayRate;   /* Part:98,  default 3.0 */
    int   linkToViscosity;  /* Part:104, default 1   */
    float viscosity;        /* FluidSolver.viscosity, Part:265 */
    float windowW, windowH; /* windowSize of Part:242; tracer positions are in window units */
} fs_particle_params;

/* replaces the CPU loop ParticleSystem.update(), Part:242-304, with the velocity
 * query of FluidSolver.getVelocityAt, Solver:648-663, done for real on the device
 * field (nearest cell, real grid size; fluidVel = cell velocity / (W,H)).  SoA, in
 * place.  Slots with life == 0 are dead and skipped; a particle that dies keeps its
 * updated position and gets life = 0 (the reference removes it, Part:296). */
int fs_k_particles_update(const fs_grid *g, const float *u, const float *v,
                          const fs_particle_params *pp, long n,
                          float *px, float *py, float *vx, float *vy,
                          float *life, const float *mass, float *age, void *stream);

/* replaces the vertex shader particleVertex, Shaders:446-476 (buffer 1 viewportSize, buffer 2 particleScale), as a
 * compute pass: per tracer the two line vertices in NDC, out_xy4 = n x (x0, y0, x1, y1) with
 * (x0, y0) = ndc(pos - vel * particleScale), (x1, y1) = ndc(pos), ndc(p) = (p / viewport) * 2 - 1 with y flipped,
 * and out_alpha = life (the colour is constant white, :472).  Device pointers; out_xy4 16-byte aligned. */
int fs_k_particle_lines(long n, const float *px, const float *py, const float *vx, const float *vy, const float *life,
                        float viewportW, float viewportH, float particleScale, float *out_xy4, float *out_alpha,
                        void *stream);

/* Tracers on row slabs; no counterpart in the reference.  A tracer lives on the rank that
 * owns the grid row its velocity lookup reads (Solver:651); dead slots never move.  Arrays are the seven SoA
 * planes in the order px, py, vx, vy, life, mass, age (device pointers).
 *   fs_k_tracer_count   counts_dev[d] += tracers of `in` whose home is rank d (counts_dev: `world` ints)
 *   fs_k_tracer_scatter copies every tracer to keep[...] (home == rank) or outbox[...] (home == d) at the slot
 *                       taken from cursors_dev[d]++ -- preset cursors_dev[d] to rank d's outbox offset and
 *                       cursors_dev[rank] to 0.  The order inside a destination is not deterministic. */
int fs_k_tracer_count(const fs_grid *g, const fs_particle_params *pp, int world, int rank, long n,
                      const float *py, const float *life, int *counts_dev, void *stream);
int fs_k_tracer_scatter(const fs_grid *g, const fs_particle_params *pp, int world, int rank, long n,
                        float *const in[7], float *const keep[7], float *const outbox[7], int *cursors_dev,
                        void *stream);

/* No counterpart in the reference (it has one GPU and no CFL logic): max |plane| over
 * rows [yb,ye), written as a float into *out_dev (device; must hold 0.0f beforehand).
 * Used to size the advection halo of a row slab: ceil(max|v_y| * dt) + 1 rows. */
int fs_k_max_abs(const fs_grid *g, const float *plane, float *out_dev, void *stream);

/* subtractGradient (Shaders:225-245) and, in the same pass over the velocity, fs_k_max_abs of the
 * new v plane over rows [yb,ye) into *cfl_dev (device float; must hold a lower bound, normally 0.0f,
 * beforehand).  Velocity bits equal fs_k_subtractGradient's, the maximum equals fs_k_max_abs's.  The
 * CFL part has no counterpart in the reference; the slab runtime uses it to save a pass. */
int fs_k_subtractGradient_cfl(const fs_grid *g, const float *p, float *u, float *v, float *cfl_dev, void *stream);

/* -------------------------------------------------------------------------
 * Layer 2: solver context (mirrors class FluidSolver, Solver:240-632, and the
 * update half of class ParticleSystem, Part:242-304).
 * ---------------------------------------------------------------------- */
typedef struct fs_ctx fs_ctx;

/* The @Published parameters update() reads (Solver:139-202, :227). */
typedef struct fs_params {
    float viscosity;         /* Solver:139 default 0.00015; diffuse runs iff > 0 (:505) */
    float diffusion;         /* Solver:146 default 0.0;     dye diffuse runs iff > 0 (:562) */
    float deltaTime;         /* Solver:153 default 0.5  */
    float vorticityStrength; /* Solver:160 default 0.0;     confinement runs iff > 0 (:486) */
    float fadeSpeed;         /* Solver:167 default 0.002 */
    int   solverIterations;  /* Solver:174 default 10   */
    int   displayMode;       /* Solver:214-225, FS_DISPLAY_*; FS_DISPLAY_NONE skips the render */
    float brightness;        /* Solver:195 default 0.06 */
} fs_params;

enum { FS_DISPLAY_NONE = -1, FS_DISPLAY_DENSITY = 0, FS_DISPLAY_VELOCITY = 1, FS_DISPLAY_SPEED = 2,
       FS_DISPLAY_VORTICITY = 3, FS_DISPLAY_PARTICLES_ONLY = 4 };

/* fs_read_field(FS_FIELD_VORTICITY) after a frame that computed curl and confinement in one pass (the default on grids at
 * least 512 wide) returns FS_ERR_STATE instead of an old plane.
 * Intermediate planes are only as current as the frame plan keeps them: with FS_OPT_FUSE != 0 fs_step computes the
 * vorticity in registers on grids at least 512 wide (FS_FIELD_VORTICITY is then not updated, except in the vorticity
 * display mode), and the divergence of the domain's outermost ring may be skipped by the fused projection.  Velocity,
 * density, pressure and display always hold the reference's values.  FS_OPT_FUSE = 0 or fs_stage keep every plane. */
enum { FS_FIELD_VELOCITY = 0, FS_FIELD_DENSITY = 1, FS_FIELD_PRESSURE = 2,
       FS_FIELD_DIVERGENCE = 3, FS_FIELD_VORTICITY = 4, FS_FIELD_DISPLAY = 5 };

/* options for fs_set_option */
enum { FS_OPT_USE_GRAPH = 0,      /* 1 (default): replay update() as a hipGraph */
       FS_OPT_BLOCK_DEPTH = 1,    /* sweeps fused per Jacobi launch (temporal blocking), both loops */
       FS_OPT_WITH_PARTICLES = 2, /* 1: fs_step also runs the tracer update (default 0) */
       FS_OPT_FUSE = 4,           /* fs_step fusions, same bits either way: 1 (default) = fadeDensity into the dye advection AND
                                     the projection in one loop where each pays (size rules in DESIGN.md), both advections in one pass and
                                     curl + confinement in one pass for out-of-cache planes; 2 = fade only; 3 = projection loop only, at any size; 4 = every fusion at any
                                     size (tests); 0 = none */
       FS_OPT_BLOCK_DEPTH_PRESSURE = 5, /* depth of the pressure loop, default 14 */
       FS_OPT_BLOCK_DEPTH_DIFFUSE = 6,  /* depth of the diffuse loops, default 8 */
       FS_OPT_STAGGER = 7,        /* start delay (10 ns ticks) of every other resident
                                     Jacobi workgroup, so one workgroup of a CU loads while the other sweeps */
       FS_OPT_STAGGER_MODE = 8,   /* which workgroups are held back, 0..3 (tuning) */
       FS_OPT_BAND_DELAY = 14,    /* two-band projection: head start of the upper band, in 10 ns ticks (default 100; tuning) */
       FS_OPT_SPLIT_ROWS = 13,    /* 1 (default): when the projection runs as one fused loop of at least two launches on a grid (or
                                     slab) of >= 768 tiles whose planes are below 512 MB, the loop runs as two row bands on two streams, each computing the
                                     rows beside the split redundantly (no exchange); the holes one chain of launches leaves
                                     (partly filled last round, workgroups loading) are filled by the other; same bits;
                                     0 = one chain */
       FS_OPT_SPLIT_PLANES = 12,  /* 1 (default): the independent planes of a diffuse loop run as two concurrent chains of
                                     launches (a second stream, fork/join by event; parallel branches of the frame's graph),
                                     so the partly filled last round of one launch is filled by the other chain, and the dye diffusion loop
                                     of a single-GPU frame runs beside the velocity chain (third stream); 0 = one chain */
       FS_OPT_SLAB_PROFILE = 11,  /* slabs only: 1 = time every wait of the kernels for halo rows (fs_slab_waits) */
       FS_OPT_TILE_VARIANT_DIFFUSE = 10, /* tile shape of the diffuse loops alone (default 8 = by size: 7, 5 rows per lane x 16 waves, while a
                                     plane's two sides fit the Infinity Cache, 6 beyond) */
       FS_OPT_SLAB_DEFER = 15,    /* slabs only, same value on every rank: 1 (default) = a frame that speculates on its advection halo
                                     (FS_OPT_SLAB_OVERLAP) returns without waiting for the all-reduced CFL bound; the check -- and, when
                                     the guess was too small, the repair: advection planned again with the true depth, the splats
                                     applied since repeated -- runs at the start of the next call on the context that is not
                                     fs_add_force (fs_step, fs_sync, a read, fs_slab_info ...), where FS_ERR_HALO is then reported.
                                     0 = fs_step waits for the bound itself (one device round trip per frame) */
       FS_OPT_SPLAT_IN_GRAPH = 19, /* one GPU: 1 = while frames are replayed, fs_add_force does not launch: up to 8 splats wait and
                                     become the first nodes of the next fs_step's graph, their arguments set per replay; any other call
                                     that reads or orders work behind the fields (fs_sync, reads, writes, fs_stage, fs_stream,
                                     fs_timer_*, fs_particles_update) launches them first, so the order of effects is the call order.
                                     Default 0: on ROCm 7.2 a node in front of the frame's graph costs more than the launch it
                                     replaces (459 -> 475 us per frame at 3072^2, profiles/r03_o_splat_in_graph.log) */
       FS_OPT_TRACER_SORT = 18,   /* one GPU: frames between two re-orderings of the tracer arrays by the cache line their velocity lookup
                                     reads (default 32; 0 = never).  Invisible outside: every call that exposes slot order restores it */
       FS_OPT_SLAB_SERIALIZE = 17, /* local groups only, a measurement aid (set it on EVERY slab of the group): 1 = the K slabs, which share
                                     one GPU there, run their launches one slab after the other instead of side by side, so that the
                                     group's frame takes K x what one slab's frame takes on a GPU of its own (scripts/time_slabs.py) */
       FS_OPT_SEGMENT_GRAPH_AFTER = 16, /* slabs only: frames the parameters must have stayed put before the exchange-free runs of
                                     launches of a slab's frame are recorded as hipGraphs and replayed (default 2; FS_OPT_USE_GRAPH = 0
                                     keeps every launch eager) */
       FS_OPT_SLAB_OVERLAP = 9,   /* slabs only, same value on every rank: 1 (default) hides the halo traffic
                                     behind the kernels (second stream, split advection, speculated halo
                                     depth); 0 = exchange, wait, compute */
       FS_OPT_TILE_VARIANT = 3    /* Jacobi tile shape of both loops (rows per lane x waves): 0 = 8x8, 1 = 16x8, 2 = 8x16, 3 = 16x4,
                                     4 = 4x16, 5 = 12x8, 6 = 6x16, 7 = 5x16; default 8 = by size (pressure: 4 while the loop's three
                                     planes fit the Infinity Cache, 7 beyond) */ };

/* FluidSolver.init() + setResolution (Solver:240-255, :322-343): allocates the fields
 * of a width x height grid on HIP device `device`, all zero (clearTextures :348-366). */
int fs_create(int width, int height, int device, long max_particles, fs_ctx **out);
int fs_destroy(fs_ctx *ctx);
int fs_reset(fs_ctx *ctx);                                   /* reset(), Solver:371-374 */
int fs_default_params(fs_params *p);                         /* defaults of Solver:139-202 */
int fs_set_params(fs_ctx *ctx, const fs_params *p);
int fs_get_params(fs_ctx *ctx, fs_params *p);
int fs_set_option(fs_ctx *ctx, int option, int value);   /* the tuning keys of fs_config_set change THIS context's copy only */
int fs_get_option(fs_ctx *ctx, int option, int *value);  /* what THIS context runs with (fs_config_get reads the process default) */

/* applyForce(position:force:color:), Solver:444-470, in GRID units (the 5 addForces
 * arguments; color alpha is accepted for signature parity and ignored). */
int fs_add_force(fs_ctx *ctx, float pos_x, float pos_y, float force_x, float force_y,
                 const float color_rgba[4], float radius, float deltaTime);

int fs_step(fs_ctx *ctx);   /* update(), Solver:477-627 */
int fs_sync(fs_ctx *ctx);   /* wait for all queued work; reports a pending FS_ERR_HALO etc. */

/* Host <-> device copies of a whole field, dense planar [plane][height][width] fp32.
 * These double as checkpoint/restore and as the parity tap.  Blocking. */
int fs_read_field(fs_ctx *ctx, int field, float *host, size_t bytes);
int fs_write_field(fs_ctx *ctx, int field, const float *host, size_t bytes);

/* Zero-copy view for hosts that already live on the device (benchmarks, torch):
 * device pointer of plane `plane` of the field's CURRENT ping-pong side.  The pointer is good until the next fs_step /
 * fs_stage / fs_write_field on the context (the sides swap, Solver:516,:540,:558,:584) and its contents are complete
 * once fs_sync has returned: for the pressure in particular the rows a two-band projection (FS_OPT_SPLIT_ROWS) left in
 * the band's private plane are copied home by THIS call, asynchronously on the context's stream -- a host that keeps the
 * pointer across frames, or reads it from a stream of its own without fs_sync, sees the lower rows of an older frame. */
int fs_field_device_ptr(fs_ctx *ctx, int field, int plane, void **dev_ptr, int *pitch);
void *fs_stream(fs_ctx *ctx);

/* HIP-event stopwatch on the context's stream (benchmarks): begin records an event, end records a
 * second one, waits for it and returns the elapsed milliseconds of the work queued in between. */
int fs_timer_begin(fs_ctx *ctx);
int fs_timer_end(fs_ctx *ctx, float *ms);

/* run one stage of update() on the context's own fields (n = sweeps for the two loops) */
enum { FS_STAGE_VORTICITY = 0, FS_STAGE_APPLY_VORTICITY, FS_STAGE_DIFFUSE_VELOCITY, FS_STAGE_DIVERGENCE,
       FS_STAGE_CLEAR_PRESSURE, FS_STAGE_PRESSURE, FS_STAGE_SUBTRACT_GRADIENT, FS_STAGE_ADVECT_VELOCITY,
       FS_STAGE_DIFFUSE_DENSITY, FS_STAGE_ADVECT_DENSITY, FS_STAGE_FADE, FS_STAGE_RENDER,
       FS_STAGE_PRESSURE_CLEARED, /* clearTexture + the n sweeps exactly as update() runs them (clear folded in) */
       FS_STAGE_PROJECT /* Solver:522-547 as update() runs them on one GPU: divergence, clear, n sweeps, subtractGradient in one
                           loop of launches (fs_k_project_n), as two row bands when FS_OPT_SPLIT_ROWS applies; n >= 1 */ };
int fs_stage(fs_ctx *ctx, int stage, int n);

/* Tracers (SoA on the device, owned by the context). */
int fs_particles_set(fs_ctx *ctx, long n, const float *px, const float *py, const float *vx,
                     const float *vy, const float *life, const float *mass, const float *age);
int fs_particles_get(fs_ctx *ctx, long n, float *px, float *py, float *vx, float *vy,
                     float *life, float *mass, float *age);
/* The same with the reference's own array-of-structures layout, for a host that holds [Particle] (Part:18-31; MSL twin
 * Shaders:419-426): position float2 @0, velocity float2 @8, color float4 @16, life @32, mass @36, age @40, `stride` bytes
 * per tracer (48 in Swift and MSL).  `color` is dead state on this path -- constant white (Part:215), never read by
 * update(), the line's alpha is `life` (Shaders:472) -- so set ignores it and get leaves the caller's bytes untouched. */
int fs_particles_set_aos(fs_ctx *ctx, long n, const void *particles, size_t stride);
int fs_particles_get_aos(fs_ctx *ctx, long n, void *particles, size_t stride);
int fs_default_particle_params(fs_particle_params *pp, float windowW, float windowH);
int fs_particles_update(fs_ctx *ctx, const fs_particle_params *pp); /* Part:242-304 */
long fs_particles_count(fs_ctx *ctx);   /* Part:229 particleCount; blocks when a spawn has put its compaction off (below) */
/* ParticleSystem.addParticles(at:count:windowSize:), Part:186-231, on the device: `count` new tracers at (x, y) + U(-radius,
 * radius)^2 in window units (:202-205, spawnRadius :91), zero velocity (:208), life = alpha ~ U(0.3, 1) (:211, :221), mass ~
 * U(0.1, 1) (:212), age 0.  The reference's array is the live tracers in age order, appended to with removeFirst() at
 * maxParticles (:196-199): afterwards the context holds the last `max_particles` of (live tracers + new ones), in that order,
 * and no dead slots.  The reference draws from the system RNG; here tracer j of the call draws offset x, offset y, alpha, mass
 * from an xorshift32 stream seeded seed ^ (j + 1) * 0x9E3779B9, so a host that wants fresh tracers passes a fresh seed per
 * call.  While the new tracers fit behind the slots in use the call neither waits for the device nor touches
 * the recorded frame graphs: the slots dead at that moment are only MARKED, and dropped by the next call that shows the
 * arrays or their count to the host (fs_particles_get/_count/_set, fs_particle_lines*), which see exactly the state described
 * above.  Once the slots run out the call compacts at once, which blocks (the live count comes back to the host).
 * On a slab the call is COLLECTIVE (same arguments on every rank): rank r appends tracers [count r / k, count (r+1) / k) of the
 * call to its own arrays, wherever they lie, and the next fs_step re-homes them like every other tracer; the live tracers of
 * all ranks together are the single-GPU result as a multiset.  A slab keeps no age order, so nothing is pushed out there: dead slots are
 * dropped when room runs out, and FS_ERR_NOMEM says that the rank's capacity (fs_particles_reserve) is full of live tracers. */
int fs_particles_spawn(fs_ctx *ctx, float x, float y, long count, float radius, unsigned seed);
/* (Re)allocate room for `capacity` tracers on this context; existing tracers are dropped.  On a slab this is
 * the only way to get tracer storage (capacity is per rank: leave headroom for tracers that migrate in). */
/* Line vertices of the context's tracers (fs_k_particle_lines) into host arrays of 4 n and n floats. Blocking. */
int fs_particle_lines(fs_ctx *ctx, float viewportW, float viewportH, float particleScale, float *host_xy4, float *host_alpha);
/* fs_particle_lines for callers that cannot allocate (code under HIP graph capture rules, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same outputs, bit for bit, but the
 * vertices are staged in a file-scope device array of the library (5 x FS_LINES_STATIC_MAX floats = 5 MiB per GPU,
 * resolved once per device by fs_create) instead of a per-call hipMalloc: no hipMalloc / hipFree is called.
 * More than FS_LINES_STATIC_MAX tracers -> FS_ERR_INVALID.  All calls on one GPU share that array: calls that
 * overlap on one GPU (threads, contexts) race; serialise them. */
#define FS_LINES_STATIC_MAX 262144
int fs_particle_lines_static(fs_ctx *ctx, float viewportW, float viewportH, float particleScale, float *host_xy4, float *host_alpha);
int fs_particles_reserve(fs_ctx *ctx, long capacity);
int fs_set_particle_params(fs_ctx *ctx, const fs_particle_params *pp);

/* -------------------------------------------------------------------------
 * Layer 3: row slabs across GPUs.  No counterpart in the reference (one MTLDevice, Solver:269);
 * the contract is SURVEY.md 8(e).  GPU g owns rows [g*H/k, (g+1)*H/k) and all columns; halo rows
 * move with ncclSend/ncclRecv between slab neighbours over xGMI, two batched rounds per frame
 * (deep ghost zones, recomputed redundantly), plus one 4-byte all-reduce(max) that sizes the
 * advection halo from the CFL bound of the backtrace.  The transfers run on a second stream under
 * the kernels (FS_OPT_SLAB_OVERLAP).  Every stage is the same kernel as on one GPU, so k slabs
 * equal one GPU bit for bit.
 *
 * One process per GPU:
 *     rank 0: fs_comm_unique_id(id); broadcast id to all ranks by any means (torch.distributed, MPI, a file)
 *     all:    fs_comm_create_rccl(world, rank, device, id, &comm);
 *             fs_create_slab(W, H, device, comm, rank, -1, &ctx);
 *             per frame: fs_add_force(ctx, ...) with GLOBAL coordinates on EVERY rank; fs_step(ctx);
 *     fs_read_field / fs_write_field move the rows the slab owns ([plane][r1-r0][W]).
 * COLLECTIVE on slabs -- every rank of the communicator must make the same call, in the same order, with the same
 * scalar arguments, or the frame plans of the ranks (exchange depths, launch counts, ghost validity) diverge and the
 * next fs_step deadlocks in a send/recv or computes on stale ghost rows:
 *     fs_step, fs_add_force (global coordinates), fs_set_params, fs_set_option, fs_reset, fs_write_field (it marks the
 *     ghost rows of that field stale on THIS rank; a neighbour that did not call it skips the refill), fs_particles_set,
 *     fs_set_particle_params, fs_particles_spawn, fs_comm_selftest, fs_destroy/fs_comm_destroy (order only).
 * Rank-local (any rank, any time): fs_read_field, fs_field_device_ptr, fs_particles_get/_count, fs_particle_lines,
 * fs_slab_info/_stats/_waits, fs_sync, fs_last_error.  Single-GPU only: fs_stage.
 * RCCL is loaded at run time (FS_RCCL_LIBRARY if set, else a librccl already in the process, else the
 * default search path), so single-GPU hosts need none.
 * ---------------------------------------------------------------------- */
typedef struct fs_comm fs_comm;
#define FS_COMM_ID_BYTES 128
int fs_comm_unique_id(void *id128);   /* ncclGetUniqueId */
int fs_comm_create_rccl(int world, int rank, int device, const void *id128, fs_comm **out);
/* In-process group: all `world` slabs live in this process (on one GPU or several) and exchange by
 * device memcpy.  It exists to test the slab logic on a 1-GPU box; step it with fs_group_step. */
int fs_comm_create_local(int world, fs_comm **out);
/* The same, but the copies are asynchronous on each slab's exchange stream and ordered only by the
 * events the RCCL path uses: the K slabs run concurrently on the GPU, as K ranks would. */
int fs_comm_create_local_async(int world, fs_comm **out);
/* Measurement aid: rank `rank` of a `world`-rank run whose peers do not exist.  A slab created on it is stepped with
 * fs_step like a rank of an RCCL run and does everything that rank would -- plan, launches, transfers of the same sizes,
 * CFL reduction, host work -- except that the halo rows it "receives" are copies of its own rows and the CFL bound is
 * its own: the fields mean nothing, the TIMING is that of one rank alone on its GPU with a free wire (scripts/time_slabs.py). */
int fs_comm_create_loopback(int world, int rank, fs_comm **out);
int fs_comm_destroy(fs_comm *comm);   /* after the contexts that use it */
/* Round trip through an RCCL communicator (collective: every rank calls it): `count` floats to the
 * next rank / from the previous one in one send/recv group, and a 1-float all-reduce(max). */
int fs_comm_selftest(fs_comm *comm, int count);

/* ghost_rows = capacity per interior side; < 0 picks one whole slab.  It bounds the advection halo
 * (FS_ERR_HALO from fs_step when ceil(max|v_y| dt) + 2 exceeds it); values above the slab height
 * are legal (rows then come from several ranks).  fs_stage is single-GPU only.
 * Tracers on slabs: fs_particles_reserve(ctx, capacity per rank); fs_particles_set with the SAME global arrays on
 * every rank (rank r keeps the index block [r n / world, (r+1) n / world)); FS_OPT_WITH_PARTICLES = 1 makes
 * fs_step / fs_group_step re-home the tracers (each moves to the rank whose rows its lookup reads: counts by
 * all-reduce, payload by send/recv) and then update them; fs_particles_get returns the rank's own tracers
 * (fs_particles_count of them), in no particular order. */
int fs_create_slab(int width, int height, int device, fs_comm *comm, int rank, int ghost_rows, fs_ctx **out);
int fs_slab_info(fs_ctx *ctx, int info[6]);   /* r0, r1 (owned rows), b0, b1 (buffered rows), ghost capacity, last advection halo */
int fs_slab_stats(fs_ctx *ctx, long stats[5]); /* exchange rounds, halo rows sent, frames, advection re-plans (halo guessed too small), tracers received */
/* With FS_OPT_SLAB_PROFILE = 1: milliseconds this slab's kernels spent waiting for halo rows since the last call
 * -- [0] velocity halo, [1] dye halo, [2] blocking exchanges -- and how many waits were timed.  Blocks like fs_sync. */
int fs_slab_waits(fs_ctx *ctx, double wait_ms[3], long joins[3]);
/* Host time since the last call: us[0] = inside fs_step / fs_group_step on this context (planning, enqueueing, waiting),
 * us[1] = the part of it blocked on the CFL value that sizes the advection halo (a device round trip + all-reduce per frame);
 * us[0] - us[1] is what the host spends enqueueing.  frames = calls counted.  (A local group counts on its first slab.) */
int fs_slab_host_times(fs_ctx *ctx, double us[2], long *frames);
/* Segments of this slab's frames (exchange-free runs of launches): graphs held, replays, recordings, runs launched eagerly. */
int fs_slab_graph_stats(fs_ctx *ctx, long stats[4]);
/* Where the host time of this slab's frames went since the last call, us: [0] planning, [1] segment replays (hipGraphLaunch),
 * [2] segment recordings, [3] launches made eagerly, [4] exchanges, [5] CFL reduction, [6] joins, [7] tracer steps.  (The wait
 * for the CFL value is fs_slab_host_times' us[1].) */
int fs_slab_host_breakdown(fs_ctx *ctx, double us[8]);
int fs_group_step(fs_ctx **ctxs, int n);      /* update() on all n slabs of a local group, in lock step */
/* Host only, no GPU: the frame plans (stages with their row ranges, exchanges with their depths) of
 * `frames` consecutive update() calls of one rank, as text; halos[f] = advection halo of frame f;
 * overlap = the FS_OPT_SLAB_OVERLAP setting to plan for. */
int fs_slab_plan_dump(int width, int height, int world, int rank, int ghost_rows, int overlap, const fs_params *p,
                      int frames, const int *halos, char *buf, size_t cap);

/* The frame graphs of a single-GPU context: [0] graphs held (one per ping-pong state), [1] argument refreshes in place
 * (hipGraphExecUpdate, after fs_set_params changed values only), [2] instantiations.  No reference counterpart
 * (Metal encodes every frame anew, Solver:478-627). */
int fs_graph_stats(fs_ctx *ctx, long stats[3]);

const char *fs_last_error(fs_ctx *ctx); /* ctx may be NULL: last error of the calling thread */
const char *fs_version(void);

/* Self-test (GPU): counts the fp32 bit patterns x (all 2^32) for which the constant-divisor division
 * used by the diffuse kernel differs from the IEEE division x / den.  Must return 0 mismatches. */
int fs_selftest_divide(float den, unsigned long long *mismatches_host);

/* Harness helper (host code, no GPU): xorshift32 stream -> U[lo,hi) fp32, the
 * generator SURVEY.md 8(d) names for synthetic inputs. */
int fs_harness_fill_uniform(float *host_dst, long n, float lo, float hi, unsigned seed);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDSOLVER_H */
