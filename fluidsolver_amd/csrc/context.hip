// context.hip -- solver context: owns the device fields of one grid and runs
// FluidSolver.update() (Solver:477-627) as a captured hipGraph on one in-order stream.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <new>
#include <unordered_map>
#include <vector>

namespace fs {

int config_set_block_depth(int op, int v);
int config_get_block_depth(int op);
int config_set_tile_variant(int v);
int config_get_tile_variant();

static thread_local char tl_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tl_error, sizeof(tl_error), fmt, ap);
    va_end(ap);
}
const char *get_error() { return tl_error; }

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return e == hipErrorOutOfMemory ? FS_ERR_NOMEM : FS_ERR_HIP;
}

int check_grid(const fs_grid *g)
{
    if (!g) { set_error("grid is NULL"); return FS_ERR_INVALID; }
    if (g->width < 1 || g->height < 1) { set_error("grid %dx%d is empty", g->width, g->height); return FS_ERR_INVALID; }
    if (g->pitch < g->width || (g->pitch & 63)) {
        set_error("pitch %d must be a multiple of 64 and >= width %d", g->pitch, g->width);
        return FS_ERR_INVALID;
    }
    if (g->rows < 1 || g->y0 < 0 || (long)g->y0 + g->rows > g->height) {
        set_error("buffer rows [%d,%d+%d) leave the domain height %d", g->y0, g->y0, g->rows, g->height);
        return FS_ERR_INVALID;
    }
    if (g->yb < g->y0 || g->ye > g->y0 + g->rows || g->ye < g->yb) {
        set_error("compute rows [%d,%d) are not inside buffer rows [%d,%d)", g->yb, g->ye, g->y0, g->y0 + g->rows);
        return FS_ERR_INVALID;
    }
    return FS_OK;
}

}  // namespace fs

using namespace fs;

// ---------------------------------------------------------------------------
struct fs_ctx {
    int W = 0, H = 0, pitch = 0, device = 0;
    size_t plane = 0;            // floats per plane
    fs_grid grid{};
    hipStream_t stream = nullptr;
    // ping-pong pairs: index [cur] is the field, [cur^1] the temp (the reference swaps
    // texture references, Solver:516,:540,:558,:572,:584)
    float *vel[2] = {nullptr, nullptr};   // 2 planes each
    float *den[2] = {nullptr, nullptr};   // 3 planes each
    float *prs[2] = {nullptr, nullptr};
    float *div = nullptr, *vort = nullptr;
    float *display = nullptr;             // 4 planes
    int vcur = 0, dcur = 0, pcur = 0;
    int *err_flag = nullptr;              // device
    fs_params params{};
    unsigned params_version = 0;
    int use_graph = 1;
    int with_particles = 0;
    int fuse = 1;
    std::unordered_map<unsigned long long, hipGraphExec_t> graphs;
    std::unordered_map<unsigned long long, unsigned long long> graphs_after;  // ping-pong state a replay leaves
    // tracers
    long max_particles = 0, n_particles = 0;
    float *part[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    fs_particle_params pparams{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // fs_timer_*
    char error[512] = "";
};

static int ctx_fail(fs_ctx *c, int rc)
{
    if (c) {
        strncpy(c->error, fs::get_error(), sizeof(c->error) - 1);
        c->error[sizeof(c->error) - 1] = 0;
    }
    return rc;
}
#define CTX(c)                                             \
    do {                                                   \
        if (!(c)) { fs::set_error("ctx is NULL"); return FS_ERR_INVALID; } \
        hipError_t e_ = hipSetDevice((c)->device);         \
        if (e_ != hipSuccess) return ctx_fail(c, fs::hip_fail(e_, "hipSetDevice")); \
    } while (0)
#define TRY(c, call)                        \
    do {                                    \
        int rc_ = (call);                   \
        if (rc_) return ctx_fail(c, rc_);   \
    } while (0)
#define TRYHIP(c, call)                                            \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return ctx_fail(c, fs::hip_fail(e_, #call)); \
    } while (0)

static void drop_graphs(fs_ctx *c)
{
    for (auto &kv : c->graphs) (void)hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    c->graphs_after.clear();
}

static int nplanes_of(int field)
{
    switch (field) {
    case FS_FIELD_VELOCITY: return 2;
    case FS_FIELD_DENSITY: return 3;
    case FS_FIELD_PRESSURE:
    case FS_FIELD_DIVERGENCE:
    case FS_FIELD_VORTICITY: return 1;
    case FS_FIELD_DISPLAY: return 4;
    }
    return 0;
}
static float *field_ptr(fs_ctx *c, int field)
{
    switch (field) {
    case FS_FIELD_VELOCITY: return c->vel[c->vcur];
    case FS_FIELD_DENSITY: return c->den[c->dcur];
    case FS_FIELD_PRESSURE: return c->prs[c->pcur];
    case FS_FIELD_DIVERGENCE: return c->div;
    case FS_FIELD_VORTICITY: return c->vort;
    case FS_FIELD_DISPLAY: return c->display;
    }
    return nullptr;
}

// ---- the stages of update(), on the context's own buffers -------------------
static int st_vorticity(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    return fs_k_calculateVorticity(&c->grid, u, v, c->vort, c->stream);               // Solver:488-491
}
static int st_apply_vorticity(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    return fs_k_applyVorticity(&c->grid, c->vort, u, v, c->params.vorticityStrength,  // Solver:494-501
                               c->params.deltaTime, c->stream);
}
static int st_diffuse_velocity(fs_ctx *c, int n)
{
    int in_b = 0;
    int rc = fs_k_diffuse_n(&c->grid, c->vel[c->vcur], c->vel[c->vcur ^ 1], 2, c->plane,  // Solver:506-517
                            c->params.viscosity, c->params.deltaTime, n, &in_b, c->stream);
    if (!rc && in_b) c->vcur ^= 1;
    return rc;
}
static int st_divergence(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    return fs_k_divergence(&c->grid, u, v, c->div, c->stream);                        // Solver:522-525
}
static int st_clear_pressure(fs_ctx *c)
{
    return fs_k_clear(&c->grid, c->prs[c->pcur], 1, 0, c->stream);                    // Solver:528-530
}
static int st_pressure(fs_ctx *c, int n, bool p_is_zero)
{
    int in_b = 0;
    int rc = fs_k_pressureSolve_n(&c->grid, c->div, c->prs[c->pcur], c->prs[c->pcur ^ 1], n, p_is_zero ? 1 : 0,
                                  &in_b, c->stream);                                  // Solver:533-541
    if (!rc && in_b) c->pcur ^= 1;
    return rc;
}
static int st_subtract_gradient(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    return fs_k_subtractGradient(&c->grid, c->prs[c->pcur], u, v, c->stream);         // Solver:544-547
}
static int st_advect_velocity(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    int rc = fs_k_advect(&c->grid, u, v, u, c->vel[c->vcur ^ 1], 2, c->plane, c->params.deltaTime, c->err_flag,
                         c->stream);                                                  // Solver:550-556
    if (!rc) c->vcur ^= 1;                                                            // :558
    return rc;
}
static int st_diffuse_density(fs_ctx *c, int n)
{
    int in_b = 0;
    int rc = fs_k_diffuse_n(&c->grid, c->den[c->dcur], c->den[c->dcur ^ 1], 3, c->plane,  // Solver:563-573
                            c->params.diffusion, c->params.deltaTime, n, &in_b, c->stream);
    if (!rc && in_b) c->dcur ^= 1;
    return rc;
}
static int st_advect_density(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    int rc = fs_k_advect(&c->grid, u, v, c->den[c->dcur], c->den[c->dcur ^ 1], 3, c->plane, c->params.deltaTime,
                         c->err_flag, c->stream);                                     // Solver:577-582
    if (!rc) c->dcur ^= 1;                                                            // :584
    return rc;
}
static int st_advect_density_fade(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    int rc = fs_k_advect_fade(&c->grid, u, v, c->den[c->dcur], c->den[c->dcur ^ 1], 3, c->plane,
                              c->params.deltaTime, c->params.fadeSpeed, c->err_flag, c->stream);  // Solver:577-591
    if (!rc) c->dcur ^= 1;
    return rc;
}
static int st_fade(fs_ctx *c)
{
    return fs_k_fadeDensity(&c->grid, c->den[c->dcur], 3, c->plane, c->params.fadeSpeed, c->stream);  // Solver:587-591
}
static int st_render(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    switch (c->params.displayMode) {                                                  // Solver:594-623
    case FS_DISPLAY_DENSITY:
        return fs_k_renderDensity(&c->grid, c->den[c->dcur], c->plane, 3, c->display, c->plane, c->params.brightness,
                                  c->stream);
    case FS_DISPLAY_VELOCITY: return fs_k_renderVelocity(&c->grid, u, v, c->display, c->plane, 1.0f, c->stream);
    case FS_DISPLAY_SPEED: return fs_k_renderSpeed(&c->grid, u, v, c->display, c->plane, 25.0f, c->stream);
    case FS_DISPLAY_VORTICITY:
        return fs_k_renderDensity(&c->grid, c->vort, c->plane, 1, c->display, c->plane, 10.0f, c->stream);
    case FS_DISPLAY_PARTICLES_ONLY: return fs_k_clear(&c->grid, c->display, 4, c->plane, c->stream);
    default: return FS_OK;  // FS_DISPLAY_NONE: headless
    }
}
static int st_particles(fs_ctx *c)
{
    float *u = c->vel[c->vcur], *v = u + c->plane;
    return fs_k_particles_update(&c->grid, u, v, &c->pparams, c->n_particles, c->part[0], c->part[1], c->part[2],
                                 c->part[3], c->part[4], c->part[5], c->part[6], c->stream);
}

// the whole of update(): Solver:477-627
static int enqueue_update(fs_ctx *c)
{
    const fs_params &p = c->params;
    const int N = p.solverIterations;
    int rc;
    if (p.vorticityStrength > 0.0f) {                            // :486
        if ((rc = st_vorticity(c))) return rc;
        if ((rc = st_apply_vorticity(c))) return rc;
    }
    if (p.viscosity > 0.0f) {                                    // :505
        if ((rc = st_diffuse_velocity(c, N))) return rc;
    }
    if ((rc = st_divergence(c))) return rc;                      // :522
    // clear (:528-530) is folded into the first pressure sweep (p_is_zero); with N == 0
    // fs_k_pressureSolve_n clears explicitly.
    if ((rc = st_pressure(c, N, true))) return rc;               // :533-541
    if ((rc = st_subtract_gradient(c))) return rc;               // :544
    if ((rc = st_advect_velocity(c))) return rc;                 // :550-558
    if (p.diffusion > 0.0f) {                                    // :562
        if ((rc = st_diffuse_density(c, N))) return rc;
    }
    if (c->fuse) {
        if ((rc = st_advect_density_fade(c))) return rc;         // :577-591 in one pass
    } else {
        if ((rc = st_advect_density(c))) return rc;              // :577-584
        if ((rc = st_fade(c))) return rc;                        // :587
    }
    if ((rc = st_render(c))) return rc;                          // :594-623
    if (c->with_particles && c->n_particles > 0) {
        if ((rc = st_particles(c))) return rc;                   // MetalView.swift:155 -> Part:242
    }
    return FS_OK;
}

extern "C" {

int fs_pitch_for_width(int width) { return width < 1 ? 0 : (width + 63) & ~63; }

const char *fs_version(void) { return "fluidsolver_amd 0.1 (gfx950)"; }

const char *fs_last_error(fs_ctx *ctx) { return ctx ? ctx->error : fs::get_error(); }

int fs_default_params(fs_params *p)
{
    if (!p) return FS_ERR_INVALID;
    p->viscosity = 0.00015f;       // Solver:139
    p->diffusion = 0.0f;           // :146
    p->deltaTime = 0.5f;           // :153
    p->vorticityStrength = 0.0f;   // :160
    p->fadeSpeed = 0.002f;         // :167
    p->solverIterations = 10;      // :174
    p->displayMode = FS_DISPLAY_DENSITY;  // :227
    p->brightness = 0.06f;         // :195
    return FS_OK;
}

int fs_default_particle_params(fs_particle_params *pp, float windowW, float windowH)
{
    if (!pp) return FS_ERR_INVALID;
    pp->momentum = 0.5f;           // Part:63
    pp->fluidForce = 0.6f;         // Part:70
    pp->fluidDecayRate = 3.0f;     // Part:98
    pp->linkToViscosity = 1;       // Part:104
    pp->viscosity = 0.00015f;      // Solver:139
    pp->windowW = windowW;
    pp->windowH = windowH;
    return FS_OK;
}

int fs_create(int width, int height, int device, long max_particles, fs_ctx **out)
{
    if (!out) { fs::set_error("fs_create: out is NULL"); return FS_ERR_INVALID; }
    *out = nullptr;
    if (width < 1 || height < 1 || width > (1 << 22) || height > (1 << 22) || max_particles < 0) {
        fs::set_error("fs_create: bad size %dx%d / %ld particles", width, height, max_particles);
        return FS_ERR_INVALID;
    }
    FS_HIP(hipSetDevice(device));
    fs_ctx *c = new (std::nothrow) fs_ctx();
    if (!c) return FS_ERR_NOMEM;
    c->W = width; c->H = height; c->device = device;
    c->pitch = fs_pitch_for_width(width);
    c->plane = (size_t)c->pitch * (size_t)height;
    c->grid = fs_grid{width, height, c->pitch, 0, height, 0, height};
    fs_default_params(&c->params);
    fs_default_particle_params(&c->pparams, (float)width, (float)height);
    c->max_particles = max_particles;
    int rc = FS_OK;
    auto alloc = [&](float **p, size_t nplanes) {
        if (rc) return;
        hipError_t e = hipMalloc((void **)p, nplanes * c->plane * sizeof(float));
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMalloc(field)");
    };
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) rc = fs::hip_fail(e, "hipStreamCreate");
    for (int i = 0; i < 2; ++i) { alloc(&c->vel[i], 2); alloc(&c->den[i], 3); alloc(&c->prs[i], 1); }
    alloc(&c->div, 1); alloc(&c->vort, 1); alloc(&c->display, 4);
    if (!rc) {
        e = hipMalloc((void **)&c->err_flag, sizeof(int));
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMalloc(err_flag)");
    }
    for (int i = 0; i < 7 && !rc && max_particles > 0; ++i) {
        e = hipMalloc((void **)&c->part[i], (size_t)max_particles * sizeof(float));
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMalloc(particles)");
    }
    if (rc) { fs_destroy(c); return rc; }
    rc = fs_reset(c);
    if (rc) { fs_destroy(c); return rc; }
    *out = c;
    return FS_OK;
}

int fs_destroy(fs_ctx *c)
{
    if (!c) return FS_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graphs(c);
    for (int i = 0; i < 2; ++i) { (void)hipFree(c->vel[i]); (void)hipFree(c->den[i]); (void)hipFree(c->prs[i]); }
    (void)hipFree(c->div); (void)hipFree(c->vort); (void)hipFree(c->display); (void)hipFree(c->err_flag);
    for (int i = 0; i < 7; ++i) (void)hipFree(c->part[i]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FS_OK;
}

// reset(), Solver:371-374 -> clearTextures(), :348-366: all eight simulation textures
// (and here the display too) to zero.
int fs_reset(fs_ctx *c)
{
    CTX(c);
    const size_t pb = c->plane * sizeof(float);
    for (int i = 0; i < 2; ++i) {
        TRYHIP(c, hipMemsetAsync(c->vel[i], 0, 2 * pb, c->stream));
        TRYHIP(c, hipMemsetAsync(c->den[i], 0, 3 * pb, c->stream));
        TRYHIP(c, hipMemsetAsync(c->prs[i], 0, pb, c->stream));
    }
    TRYHIP(c, hipMemsetAsync(c->div, 0, pb, c->stream));
    TRYHIP(c, hipMemsetAsync(c->vort, 0, pb, c->stream));
    TRYHIP(c, hipMemsetAsync(c->display, 0, 4 * pb, c->stream));
    TRYHIP(c, hipMemsetAsync(c->err_flag, 0, sizeof(int), c->stream));
    return FS_OK;
}

int fs_set_params(fs_ctx *c, const fs_params *p)
{
    CTX(c);
    if (!p) { fs::set_error("params is NULL"); return ctx_fail(c, FS_ERR_INVALID); }
    if (p->solverIterations < 0 || p->solverIterations > 100000) {
        fs::set_error("solverIterations %d out of range", p->solverIterations);
        return ctx_fail(c, FS_ERR_INVALID);
    }
    if (memcmp(&c->params, p, sizeof(*p)) != 0) {
        c->params = *p;
        c->pparams.viscosity = p->viscosity;
        drop_graphs(c);  // scalars are baked into the captured kernel nodes
    }
    return FS_OK;
}

int fs_get_params(fs_ctx *c, fs_params *p)
{
    if (!c || !p) return FS_ERR_INVALID;
    *p = c->params;
    return FS_OK;
}

int fs_set_option(fs_ctx *c, int option, int value)
{
    CTX(c);
    switch (option) {
    case FS_OPT_USE_GRAPH: c->use_graph = value ? 1 : 0; break;
    case FS_OPT_BLOCK_DEPTH:
        if (fs::config_set_block_depth(-1, value)) { fs::set_error("block depth %d out of range 1..48", value); return ctx_fail(c, FS_ERR_INVALID); }
        break;
    case FS_OPT_WITH_PARTICLES: c->with_particles = value ? 1 : 0; break;
    case FS_OPT_FUSE: c->fuse = value ? 1 : 0; break;
    case FS_OPT_TILE_VARIANT:
        if (fs::config_set_tile_variant(value)) { fs::set_error("tile variant %d out of range 0..5", value); return ctx_fail(c, FS_ERR_INVALID); }
        break;
    default: fs::set_error("unknown option %d", option); return ctx_fail(c, FS_ERR_INVALID);
    }
    drop_graphs(c);
    return FS_OK;
}

int fs_add_force(fs_ctx *c, float pos_x, float pos_y, float force_x, float force_y, const float color_rgba[4],
                 float radius, float deltaTime)
{
    CTX(c);
    if (!color_rgba) { fs::set_error("color is NULL"); return ctx_fail(c, FS_ERR_INVALID); }
    float *u = c->vel[c->vcur], *v = u + c->plane;
    float *d = c->den[c->dcur];
    TRY(c, fs_k_addForces(&c->grid, u, v, d, d + c->plane, d + 2 * c->plane, pos_x, pos_y, force_x, force_y,
                          color_rgba[0], color_rgba[1], color_rgba[2], radius, deltaTime, c->stream));
    return FS_OK;
}

int fs_step(fs_ctx *c)
{
    CTX(c);
    if (!c->use_graph) {
        TRY(c, enqueue_update(c));
        return FS_OK;
    }
    // Buffer roles alternate between frames when a swap count is odd, so a captured graph
    // is keyed by the ping-pong state it was recorded from.
    const unsigned long long key = (unsigned long long)c->vcur | ((unsigned long long)c->dcur << 1) |
                                   ((unsigned long long)c->pcur << 2) |
                                   ((unsigned long long)(c->with_particles ? c->n_particles : 0) << 3);
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        const int v0 = c->vcur, d0 = c->dcur, p0 = c->pcur;
        hipGraph_t graph = nullptr;
        TRYHIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
        int rc = enqueue_update(c);
        hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (rc || e != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            c->vcur = v0; c->dcur = d0; c->pcur = p0;
            if (rc) return ctx_fail(c, rc);
            return ctx_fail(c, fs::hip_fail(e, "hipStreamEndCapture"));
        }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
            c->vcur = v0; c->dcur = d0; c->pcur = p0;
            return ctx_fail(c, fs::hip_fail(e, "hipGraphInstantiate"));
        }
        // remember where the roles end up so replays can update them without re-recording
        const unsigned long long after = (unsigned long long)c->vcur | ((unsigned long long)c->dcur << 1) |
                                         ((unsigned long long)c->pcur << 2);
        c->graphs[key] = exec;
        c->graphs_after[key] = after;
        TRYHIP(c, hipGraphLaunch(exec, c->stream));
        return FS_OK;
    }
    TRYHIP(c, hipGraphLaunch(it->second, c->stream));
    const unsigned long long after = c->graphs_after[key];
    c->vcur = (int)(after & 1); c->dcur = (int)((after >> 1) & 1); c->pcur = (int)((after >> 2) & 1);
    return FS_OK;
}

int fs_sync(fs_ctx *c)
{
    CTX(c);
    TRYHIP(c, hipStreamSynchronize(c->stream));
    int flag = 0;
    TRYHIP(c, hipMemcpy(&flag, c->err_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
        (void)hipMemset(c->err_flag, 0, sizeof(int));
        fs::set_error("advect backtrace left the buffered rows (halo too small)");
        return ctx_fail(c, FS_ERR_HALO);
    }
    return FS_OK;
}

int fs_read_field(fs_ctx *c, int field, float *host, size_t bytes)
{
    CTX(c);
    const int np = nplanes_of(field);
    if (!np || !host) { fs::set_error("fs_read_field: bad field %d or NULL host pointer", field); return ctx_fail(c, FS_ERR_INVALID); }
    const size_t need = (size_t)np * c->W * c->H * sizeof(float);
    if (bytes != need) { fs::set_error("fs_read_field: expected %zu bytes, got %zu", need, bytes); return ctx_fail(c, FS_ERR_INVALID); }
    TRYHIP(c, hipStreamSynchronize(c->stream));
    const float *src = field_ptr(c, field);
    for (int p = 0; p < np; ++p)
        TRYHIP(c, hipMemcpy2D(host + (size_t)p * c->W * c->H, (size_t)c->W * sizeof(float), src + p * c->plane,
                              (size_t)c->pitch * sizeof(float), (size_t)c->W * sizeof(float), c->H,
                              hipMemcpyDeviceToHost));
    return FS_OK;
}

int fs_write_field(fs_ctx *c, int field, const float *host, size_t bytes)
{
    CTX(c);
    const int np = nplanes_of(field);
    if (!np || !host) { fs::set_error("fs_write_field: bad field %d or NULL host pointer", field); return ctx_fail(c, FS_ERR_INVALID); }
    const size_t need = (size_t)np * c->W * c->H * sizeof(float);
    if (bytes != need) { fs::set_error("fs_write_field: expected %zu bytes, got %zu", need, bytes); return ctx_fail(c, FS_ERR_INVALID); }
    TRYHIP(c, hipStreamSynchronize(c->stream));
    float *dst = field_ptr(c, field);
    for (int p = 0; p < np; ++p)
        TRYHIP(c, hipMemcpy2D(dst + p * c->plane, (size_t)c->pitch * sizeof(float), host + (size_t)p * c->W * c->H,
                              (size_t)c->W * sizeof(float), (size_t)c->W * sizeof(float), c->H,
                              hipMemcpyHostToDevice));
    return FS_OK;
}

int fs_field_device_ptr(fs_ctx *c, int field, int plane, void **dev_ptr, int *pitch)
{
    if (!c || !dev_ptr) return FS_ERR_INVALID;
    const int np = nplanes_of(field);
    if (plane < 0 || plane >= np) { fs::set_error("bad field/plane %d/%d", field, plane); return ctx_fail(c, FS_ERR_INVALID); }
    *dev_ptr = field_ptr(c, field) + (size_t)plane * c->plane;
    if (pitch) *pitch = c->pitch;
    return FS_OK;
}

void *fs_stream(fs_ctx *c) { return c ? (void *)c->stream : nullptr; }

int fs_timer_begin(fs_ctx *c)
{
    CTX(c);
    if (!c->ev0) { TRYHIP(c, hipEventCreate(&c->ev0)); TRYHIP(c, hipEventCreate(&c->ev1)); }
    TRYHIP(c, hipEventRecord(c->ev0, c->stream));
    return FS_OK;
}

int fs_timer_end(fs_ctx *c, float *ms)
{
    CTX(c);
    if (!c->ev0 || !ms) { fs::set_error("fs_timer_end without fs_timer_begin"); return ctx_fail(c, FS_ERR_STATE); }
    TRYHIP(c, hipEventRecord(c->ev1, c->stream));
    TRYHIP(c, hipEventSynchronize(c->ev1));
    TRYHIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return FS_OK;
}

int fs_stage(fs_ctx *c, int stage, int n)
{
    CTX(c);
    if (n < 0) { fs::set_error("negative sweep count"); return ctx_fail(c, FS_ERR_INVALID); }
    int rc;
    switch (stage) {
    case FS_STAGE_VORTICITY: rc = st_vorticity(c); break;
    case FS_STAGE_APPLY_VORTICITY: rc = st_apply_vorticity(c); break;
    case FS_STAGE_DIFFUSE_VELOCITY: rc = st_diffuse_velocity(c, n); break;
    case FS_STAGE_DIVERGENCE: rc = st_divergence(c); break;
    case FS_STAGE_CLEAR_PRESSURE: rc = st_clear_pressure(c); break;
    case FS_STAGE_PRESSURE: rc = st_pressure(c, n, false); break;
    case FS_STAGE_SUBTRACT_GRADIENT: rc = st_subtract_gradient(c); break;
    case FS_STAGE_ADVECT_VELOCITY: rc = st_advect_velocity(c); break;
    case FS_STAGE_DIFFUSE_DENSITY: rc = st_diffuse_density(c, n); break;
    case FS_STAGE_ADVECT_DENSITY: rc = st_advect_density(c); break;
    case FS_STAGE_FADE: rc = st_fade(c); break;
    case FS_STAGE_RENDER: rc = st_render(c); break;
    default: fs::set_error("unknown stage %d", stage); rc = FS_ERR_INVALID;
    }
    return rc ? ctx_fail(c, rc) : FS_OK;
}

int fs_particles_set(fs_ctx *c, long n, const float *px, const float *py, const float *vx, const float *vy,
                     const float *life, const float *mass, const float *age)
{
    CTX(c);
    if (n < 0 || n > c->max_particles) { fs::set_error("particle count %ld exceeds capacity %ld", n, c->max_particles); return ctx_fail(c, FS_ERR_INVALID); }
    const float *src[7] = {px, py, vx, vy, life, mass, age};
    TRYHIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 7 && n > 0; ++i) {
        if (!src[i]) { fs::set_error("NULL particle array %d", i); return ctx_fail(c, FS_ERR_INVALID); }
        TRYHIP(c, hipMemcpy(c->part[i], src[i], (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    }
    if (n != c->n_particles) drop_graphs(c);
    c->n_particles = n;
    return FS_OK;
}

int fs_particles_get(fs_ctx *c, long n, float *px, float *py, float *vx, float *vy, float *life, float *mass,
                     float *age)
{
    CTX(c);
    if (n < 0 || n > c->n_particles) { fs::set_error("asked for %ld of %ld particles", n, c->n_particles); return ctx_fail(c, FS_ERR_INVALID); }
    float *dst[7] = {px, py, vx, vy, life, mass, age};
    TRYHIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 7 && n > 0; ++i)
        if (dst[i]) TRYHIP(c, hipMemcpy(dst[i], c->part[i], (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return FS_OK;
}

long fs_particles_count(fs_ctx *c) { return c ? c->n_particles : 0; }

int fs_particles_update(fs_ctx *c, const fs_particle_params *pp)
{
    CTX(c);
    if (pp) {
        if (memcmp(&c->pparams, pp, sizeof(*pp)) != 0) { c->pparams = *pp; if (c->with_particles) drop_graphs(c); }
    }
    if (c->n_particles == 0) return FS_OK;   // Part:243 guard
    TRY(c, st_particles(c));
    return FS_OK;
}

int fs_harness_fill_uniform(float *dst, long n, float lo, float hi, unsigned seed)
{
    if (!dst || n < 0) return FS_ERR_INVALID;
    unsigned x = seed ? seed : 1u;
    for (long i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        float r01 = (float)(x >> 8) * (1.0f / 16777216.0f);
        dst[i] = lo + (hi - lo) * r01;
    }
    return FS_OK;
}

int fs_config_set(int key, int value)
{
    if (key == FS_OPT_BLOCK_DEPTH) return fs::config_set_block_depth(-1, value);
    if (key == FS_OPT_BLOCK_DEPTH_PRESSURE) return fs::config_set_block_depth(0, value);
    if (key == FS_OPT_BLOCK_DEPTH_DIFFUSE) return fs::config_set_block_depth(1, value);
    if (key == FS_OPT_TILE_VARIANT) return fs::config_set_tile_variant(value);
    return FS_ERR_INVALID;
}
int fs_config_get(int key)
{
    if (key == FS_OPT_BLOCK_DEPTH || key == FS_OPT_BLOCK_DEPTH_PRESSURE) return fs::config_get_block_depth(0);
    if (key == FS_OPT_BLOCK_DEPTH_DIFFUSE) return fs::config_get_block_depth(1);
    if (key == FS_OPT_TILE_VARIANT) return fs::config_get_tile_variant();
    return FS_ERR_INVALID;
}

}  // extern "C"
