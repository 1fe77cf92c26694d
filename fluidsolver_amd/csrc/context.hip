// context.hip -- solver context: owns the device fields of one grid (or one row slab of it) and runs
// FluidSolver.update() (Solver:477-627) on one in-order stream; on a single GPU as a captured hipGraph.
#include "ctx.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <mutex>
#include <new>

namespace fs {


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

int ctx_fail(fs_ctx *c, int rc)
{
    if (c) {
        strncpy(c->error, fs::get_error(), sizeof(c->error) - 1);
        c->error[sizeof(c->error) - 1] = 0;
    }
    return rc;
}

}  // namespace fs

namespace fs {
// GK::nt, the size rule of the streaming stores (common.h st4s): bit 0 = the single passes, bit 1 = the Jacobi launches.  A plane
// that is far from fitting the 256 MB Infinity Cache beside the other planes of its pass is gone from every cache before anything
// reads it back, and a store that says so leaves the lines that ARE about to be re-read (stencil neighbours, gather taps) alone.
// Frames, streaming stores off / passes / Jacobi / both (profiles/r03_z_streaming_stores.log): 16384^2 (1 GB planes) 11832 /
// 11684 / 11672 / 11559 us; 16384 x 2048 (128 MB) 1585 / 1575 / 1545 / 1546; 6144^2 (151 MB) 1788 / 1791 / 1749 / 1752; 4096^2
// (67 MB) 826 / 817 / 826 / 825; 3072^2 (37 MB) 464 / -- / -- / 478.  So: both from 96 MB per plane on, none below.
int stream_stores(const fs_grid *g)
{
    return (double)g->pitch * (double)g->rows * sizeof(float) >= 96.0 * 1024 * 1024 ? 3 : 0;
}
}  // namespace fs

using namespace fs;

static void drop_graphs(fs_ctx *c)
{
    c->stable_frames = 0;
    fs::drop_segment_graphs(c);
    // (a replay may still be running, and not every runtime puts the destruction of an executing graph off; drops are rare)
    if (!c->graphs.empty() && c->stream && !c->dry) (void)hipStreamSynchronize(c->stream);
    for (auto &kv : c->graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto &kv : c->graphs_src) (void)hipGraphDestroy(kv.second);
    c->graphs_src.clear();
    c->graphs_splat_nodes.clear();
    c->graphs.clear();
    c->graphs_after.clear();
    c->graphs_epoch.clear();
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
static int field_index(int field)
{
    switch (field) {
    case FS_FIELD_VELOCITY: return F_VEL;
    case FS_FIELD_DENSITY: return F_DYE;
    case FS_FIELD_PRESSURE: return F_P;
    case FS_FIELD_DIVERGENCE: return F_DIV;
    case FS_FIELD_VORTICITY: return F_VORT;
    }
    return -1;
}
static float *field_ptr(fs_ctx *c, int field)
{
    if (field == FS_FIELD_DISPLAY) return c->display;
    const int i = field_index(field);
    return i < 0 ? nullptr : c->f[i].p();
}

#define SINGLE(c, what)                                                                              \
    do {                                                                                             \
        if ((c)->world > 1) { fs::set_error(what " is not available on a slab of a multi-GPU run"); \
                              return fs::ctx_fail(c, FS_ERR_STATE); }                                \
    } while (0)

// ---- the stages of update(), on the context's own buffers (single GPU) ----------------------
static int st_sweeps(fs_ctx *c, int fi, int n, bool pressure, bool p_is_zero)
{
    Field &F = c->f[fi];
    int in_b = 0, rc;
    if (pressure)
        rc = fs::pressure_n(c->tuning, &c->grid, c->f[F_DIV].p(), F.p(), F.t(), n, p_is_zero ? 1 : 0, &in_b, c->stream);  // Solver:533-541
    else
        rc = fs::diffuse_n(c->tuning, &c->grid, F.p(), F.t(), F.np, c->plane, fi == F_VEL ? c->params.viscosity : c->params.diffusion,
                            c->params.deltaTime, n, &in_b, c->stream);                 // Solver:506-517, :563-573
    if (!rc && in_b) F.flip();
    return rc;
}
// Solver:522-547 the way a single-GPU frame runs them (FS_STAGE_PROJECT)
static int st_project(fs_ctx *c, int n)
{
    if (n < 1) { fs::set_error("FS_STAGE_PROJECT needs at least one sweep"); return FS_ERR_INVALID; }
    Field &P = c->f[F_P];
    float *u = c->f[F_VEL].p(), *v = u + c->plane, *div = c->f[F_DIV].p();
    int in_b = 0, rc;
    fs::ensure_band_planes(c, n);
    if (fs::bands_wanted(c, n) && c->p_band[0]) {
        const int split = (c->H / 2 + 3) & ~3;
        rc = fs::project_n_bands(c->tuning, &c->grid, split, u, v, div, P.p(), P.t(), c->p_band[0], c->p_band[1], n, &in_b, c->stream,
                                 c->stream2, c->ev_fork, c->ev_join, c->ev_band[0], c->ev_band[1], nullptr, 0);
        if (!rc) { c->p_band_dirty = true; c->p_band_row = split; c->p_band_final = in_b; }
    } else
        rc = fs::project_n_cfl(c->tuning, &c->grid, u, v, div, P.p(), P.t(), n, nullptr, &in_b, c->stream, 0);
    if (!rc && in_b) P.flip();
    return rc;
}
static int st_advect(fs_ctx *c, int fi)
{
    Field &F = c->f[fi];
    float *u = c->f[F_VEL].p(), *v = u + c->plane;
    int rc = fs_k_advect(&c->grid, u, v, F.p(), F.t(), F.np, c->plane, c->params.deltaTime, c->err_flag, c->stream);   // Solver:550-556, :577-582
    if (!rc) F.flip();                                                               // :558, :584
    return rc;
}
static int st_render(fs_ctx *c)
{
    float *u = c->f[F_VEL].p(), *v = u + c->plane;
    switch (c->params.displayMode) {                                                  // Solver:594-623
    case FS_DISPLAY_DENSITY:
        return fs_k_renderDensity(&c->grid, c->f[F_DYE].p(), c->plane, 3, c->display, c->plane, c->params.brightness, c->stream);
    case FS_DISPLAY_VELOCITY: return fs_k_renderVelocity(&c->grid, u, v, c->display, c->plane, 1.0f, c->stream);
    case FS_DISPLAY_SPEED: return fs_k_renderSpeed(&c->grid, u, v, c->display, c->plane, 25.0f, c->stream);
    case FS_DISPLAY_VORTICITY:
        return fs_k_renderDensity(&c->grid, c->f[F_VORT].p(), c->plane, 1, c->display, c->plane, 10.0f, c->stream);
    case FS_DISPLAY_PARTICLES_ONLY: return fs_k_clear(&c->grid, c->display, 4, c->plane, c->stream);
    default: return FS_OK;  // FS_DISPLAY_NONE: headless
    }
}
static int st_particles(fs_ctx *c)
{
    float *u = c->f[F_VEL].p(), *v = u + c->plane;
    return fs_k_particles_update(&c->grid, u, v, &c->pparams, c->n_particles, c->part[0], c->part[1], c->part[2],
                                 c->part[3], c->part[4], c->part[5], c->part[6], c->stream);
}

// ---- tracer order (one GPU): see passes.hip "Tracer order" ------------------------------------------------------
static int ensure_part_t(fs_ctx *c)
{
    bool all = true;
    for (int i = 0; i < 7; ++i) all = all && c->part_t[i];
    if (all) return FS_OK;
    for (int i = 0; i < 7; ++i) {
        if (c->part_t[i]) continue;
        hipError_t e = hipMalloc((void **)&c->part_t[i], (size_t)c->max_particles * sizeof(float));
        if (e != hipSuccess) {   // all or nothing: a later call must not find half of the planes
            for (int k = 0; k < 7; ++k) { (void)hipFree(c->part_t[k]); c->part_t[k] = nullptr; }
            return fs::hip_fail(e, "hipMalloc(tracer partner arrays)");
        }
    }
    FS_HIP(hipMemsetAsync(c->part_t[4], 0, (size_t)c->max_particles * sizeof(float), c->stream));
    return FS_OK;
}
// Invariant of a single-GPU context: in the CURRENT arrays every slot from n_particles up to the capacity has life 0, so
// the tracer node of a frame's graph can be launched over a rounded-up slot count (tracer_bound) and spawning a few
// tracers does not change the node.  Whoever fills the partner arrays with m tracers clears the rest of their life plane.
static int clear_life_tail(fs_ctx *c, float *life, long m)
{
    if (!life || m >= c->max_particles) return FS_OK;
    FS_HIP(hipMemsetAsync(life + m, 0, (size_t)(c->max_particles - m) * sizeof(float), c->stream));
    return FS_OK;
}
static long tracer_bound(const fs_ctx *c)
{
    if (c->world > 1) return c->n_particles;
    return std::min(c->max_particles, (c->n_particles + 16383L) & ~16383L);
}
static int ensure_spawn_sums(fs_ctx *c)
{
    const long nb = fs::tracer_blocks(c->max_particles);
    if (c->spawn_sums_cap >= nb + 1) return FS_OK;
    (void)hipFree(c->spawn_sums);
    c->spawn_sums = nullptr; c->spawn_sums_cap = 0;   // (a failed allocation below leaves "none", not a stale size)
    FS_HIP(hipMalloc((void **)&c->spawn_sums, (size_t)(nb + 1) * sizeof(int)));
    c->spawn_sums_cap = nb + 1;
    return FS_OK;
}

// back to slot order: every call that shows the arrays to the host, or relies on their age order, comes through here
static int tracer_restore_order(fs_ctx *c)
{
    if (!c->part_sorted) return FS_OK;
    c->part_sorted = false;
    c->frames_since_sort = 1 << 20;   // sorted again by the next update
    if (c->n_particles > 0) {
        int rc = fs::tracer_unsort(c->n_particles, c->part, c->part_t, c->part_perm, c->stream);
        if (!rc) rc = clear_life_tail(c, c->part_t[4], c->n_particles);
        if (rc) return rc;
        for (int i = 0; i < 7; ++i) std::swap(c->part[i], c->part_t[i]);
        c->part_flip ^= 1;
    }
    return FS_OK;
}
// The compaction a fast-path spawn put off: drop the slots it marked removed (slot order first).  Blocking: the count
// comes back to the host.  Allocates nothing (the fast path made sure the partner arrays and the scan buffer exist).
static int tracer_settle_removed(fs_ctx *c)
{
    if (!c->removed_pending) return FS_OK;
    int rc = tracer_restore_order(c);
    if (rc) return rc;
    c->removed_pending = false;
    const long n = c->n_particles;
    if (n == 0) return FS_OK;
    rc = fs::tracer_count_live(n, c->part[4], c->spawn_sums, true, c->stream);
    if (rc) return rc;
    int total = 0;
    FS_HIP(hipMemcpyAsync(&total, c->spawn_sums + fs::tracer_blocks(n), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    if ((rc = fs::tracer_compact(n, c->part, c->part_t, c->spawn_sums, 0, true, c->stream))) return rc;
    if ((rc = clear_life_tail(c, c->part_t[4], total))) return rc;
    for (int i = 0; i < 7; ++i) std::swap(c->part[i], c->part_t[i]);
    c->part_flip ^= 1;
    c->n_particles = total;
    return FS_OK;
}
// called once per tracer update, before it is enqueued (never under stream capture: it allocates on first use)
static int tracer_maybe_sort(fs_ctx *c)
{
    if (c->world > 1 || c->sort_every <= 0 || c->n_particles < 16384) return FS_OK;
    if (++c->frames_since_sort < c->sort_every) return FS_OK;
    int nbx, shift;
    const long nbins = (fs::tracer_bins(&c->grid, &nbx, &shift) + 3) & ~3;   // (the scan reads and writes quads)
    int rc = ensure_part_t(c);
    if (rc) return rc;
    if (c->sort_cap < c->max_particles) {
        (void)hipFree(c->part_perm); (void)hipFree(c->part_perm_t); (void)hipFree(c->sort_keys);
        c->part_perm = c->part_perm_t = c->sort_keys = nullptr;
        c->sort_cap = 0;
        c->part_sorted = false;   // (only reached with the arrays in slot order: the capacity changes through fs_particles_reserve)
        FS_HIP(hipMalloc((void **)&c->part_perm, (size_t)c->max_particles * sizeof(int)));
        FS_HIP(hipMalloc((void **)&c->part_perm_t, (size_t)c->max_particles * sizeof(int)));
        FS_HIP(hipMalloc((void **)&c->sort_keys, (size_t)c->max_particles * sizeof(int)));
        c->sort_cap = c->max_particles;
    }
    if (c->sort_hist_cap < nbins) {
        (void)hipFree(c->sort_hist);
        c->sort_hist = nullptr; c->sort_hist_cap = 0;
        FS_HIP(hipMalloc((void **)&c->sort_hist, (size_t)nbins * sizeof(int)));
        c->sort_hist_cap = nbins;
    }
    rc = fs::tracer_sort(&c->grid, &c->pparams, c->n_particles, c->part, c->part_t, c->sort_keys, c->sort_hist,
                         c->part_sorted ? c->part_perm : nullptr, c->part_perm_t, c->stream);
    if (!rc) rc = clear_life_tail(c, c->part_t[4], c->n_particles);
    if (rc) return rc;
    for (int i = 0; i < 7; ++i) std::swap(c->part[i], c->part_t[i]);
    std::swap(c->part_perm, c->part_perm_t);
    c->part_flip ^= 1;
    c->part_sorted = true;
    c->frames_since_sort = 0;
    c->stat_sorts += 1;
    return FS_OK;
}

// ---- staging area of fs_particle_lines_static (the entry point that allocates nothing) ----------------------
// A file-scope device array: every GPU the library runs on holds its own copy, at its own address, until the
// process ends (5 MiB).  In host code the array's NAME is only a stand-in for the symbol, never a device
// address: the address on a device comes from hipGetSymbolAddress with that device current, once per device
// (scratch_for_device, called by fs_create), and only that pointer is ever handed to kernels and copies.
static __device__ __align__(256) float scratch[5 * FS_LINES_STATIC_MAX];
constexpr int kMaxDevices = 64;
static float *g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mu;

static int scratch_for_device(int device, float **out)   // `device` must be the current device
{
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) { fs::set_error("device %d out of range", device); return FS_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        float *ptr = nullptr;
        FS_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) { fs::set_error("hipGetSymbolAddress(scratch) returned NULL on device %d", device); return FS_ERR_HIP; }
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return FS_OK;
}

// Allocates one slab (world 1: the whole grid).  dry = plan-only context without device memory.
int create_common(int width, int height, int device, fs_comm *comm, int world, int rank, int ghost_rows,
                  long max_particles, bool dry, fs_ctx **out)
{
    if (!out) { fs::set_error("fs_create: out is NULL"); return FS_ERR_INVALID; }
    *out = nullptr;
    // height: the streaming passes put ceil(rows / 4), padded to a multiple of 64, into gridDim.y (at most 65535)
    if (width < 1 || height < 1 || width > (1 << 22) || height > 261888 || max_particles < 0) {
        fs::set_error("fs_create: bad size %dx%d / %ld particles (width <= 4194304, height <= 261888)", width, height, max_particles);
        return FS_ERR_INVALID;
    }
    const int own_min = height / world;
    if (world > 1 && own_min < 4) { fs::set_error("slabs of %d rows are too thin for %d ranks", own_min, world); return FS_ERR_INVALID; }
    int G = 0;
    if (world > 1) {
        G = ghost_rows < 0 ? own_min : std::min(ghost_rows, height);   // > own_min: multi-hop exchange
        if (G < 2) { fs::set_error("ghost capacity must be >= 2 rows"); return FS_ERR_INVALID; }
    }
    if (!dry) FS_HIP(hipSetDevice(device));
    fs_ctx *c = new (std::nothrow) fs_ctx();
    if (!c) return FS_ERR_NOMEM;
    c->W = width; c->H = height; c->device = device;
    c->comm = comm; c->world = world; c->rank = rank; c->G = G; c->dry = dry;
    fs::slab_rows(height, world, rank, &c->r0, &c->r1);
    c->b0 = std::max(0, c->r0 - G);
    c->b1 = std::min(height, c->r1 + G);
    c->pitch = fs_pitch_for_width(width);
    c->plane = (size_t)c->pitch * (size_t)(c->b1 - c->b0);
    c->grid = fs::grid_of(c, 0);
    c->f[F_VEL].np = 2; c->f[F_DYE].np = 3;
    c->tuning = fs::default_tuning();
    fs_default_params(&c->params);
    fs_default_particle_params(&c->pparams, (float)width, (float)height);
    c->max_particles = max_particles;
    if (dry) { *out = c; return FS_OK; }
    int rc = FS_OK;
    auto alloc = [&](float **p, size_t nplanes) {
        if (rc) return;
        hipError_t e = hipMalloc((void **)p, nplanes * c->plane * sizeof(float));
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMalloc(field)");
    };
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) rc = fs::hip_fail(e, "hipStreamCreate");
    for (int i = 0; i < 2; ++i) { alloc(&c->f[F_VEL].buf[i], 2); alloc(&c->f[F_DYE].buf[i], 3); alloc(&c->f[F_P].buf[i], 1); }
    alloc(&c->f[F_DIV].buf[0], 1); alloc(&c->f[F_VORT].buf[0], 1); alloc(&c->display, 4);
    if (world > 1) alloc(&c->f[F_VEL].buf[2], 2);   // the spare of a slab's velocity (ctx.h)
    if (!rc) {
        e = hipMalloc((void **)&c->err_flag, sizeof(int));
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMalloc(err_flag)");
    }
    if (!rc) {
        e = hipMalloc((void **)&c->cfl_dev, 2 * sizeof(float));
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMalloc(cfl)");
    }
    if (!rc) {
        e = hipHostMalloc((void **)&c->cfl_host, 2 * (size_t)world * sizeof(float), hipHostMallocDefault);
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipHostMalloc(cfl)");
    }
    if (!rc) {
        e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork3, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_band[0], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_band[1], hipEventDisableTiming);
        if (e != hipSuccess) rc = fs::hip_fail(e, "second stream / events");
    }
    if (!rc && world > 1) {
        // exchange stream and its events (slab.hip)
        // (a high-priority stream was tried for it: with hipStreamCreateWithPriority the cross-stream event waits of
        // the in-process groups and of a 2-rank run never complete on this runtime -- the tests hang -- so plain)
        e = hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_cfl[0], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_cfl[1], hipEventDisableTiming);
        for (int i = 0; i < SLOT_COUNT && e == hipSuccess; ++i) {
            e = hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_slot[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) rc = fs::hip_fail(e, "exchange stream / events");
    }
    for (int i = 0; i < 7 && !rc && max_particles > 0; ++i) {
        e = hipMalloc((void **)&c->part[i], (size_t)max_particles * sizeof(float));
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMalloc(particles)");
    }
    if (!rc && max_particles > 0) {   // (life 0 = an empty slot, see tracer_bound)
        e = hipMemsetAsync(c->part[4], 0, (size_t)max_particles * sizeof(float), c->stream);
        if (e != hipSuccess) rc = fs::hip_fail(e, "hipMemsetAsync(life)");
    }
    // (the staging area only serves fs_particle_lines_static, which reports its absence itself: not a reason to fail a context)
    if (!rc && scratch_for_device(device, &c->lines_scratch)) { c->lines_scratch = nullptr; (void)hipGetLastError(); }
    if (rc) { fs_destroy(c); return rc; }
    rc = fs_reset(c);
    if (rc) { fs_destroy(c); return rc; }
    *out = c;
    return FS_OK;
}

static int launch_splat(fs_ctx *c, const fs_ctx::Splat &sp)
{
    const fs_grid g = fs::grid_of(c, c->G);
    float *u = c->f[F_VEL].p(), *v = u + c->plane;
    float *d = c->f[F_DYE].p();
    return fs_k_addForces(&g, u, v, d, d + c->plane, d + 2 * c->plane, sp.px, sp.py, sp.fx, sp.fy, sp.r, sp.g, sp.b, sp.radius, sp.dt,
                          c->stream);
}
// the splats that were waiting to become nodes of the next frame's graph, launched now, in order
static int flush_splats(fs_ctx *c)
{
    if (c->pending_splats.empty()) return FS_OK;
    std::vector<fs_ctx::Splat> todo;
    todo.swap(c->pending_splats);
    for (const fs_ctx::Splat &sp : todo) {
        const int rc = launch_splat(c, sp);
        if (rc) return rc;
    }
    return FS_OK;
}

extern "C" {

int fs_pitch_for_width(int width) { return width < 1 ? 0 : (width + 63) & ~63; }

const char *fs_version(void) { return "fluidsolver_amd 0.2 (gfx950)"; }

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
    return create_common(width, height, device, nullptr, 1, 0, 0, max_particles, false, out);
}

int fs_destroy(fs_ctx *c)
{
    if (!c) return FS_OK;
    if (c->dry) { delete c; return FS_OK; }
    (void)hipSetDevice(c->device);
    (void)fs::settle(c);   // a halo check still owed: repair the last frame while the group is whole (no-op otherwise)
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    drop_graphs(c);
    if (c->comm && !c->comm->solo() && c->rank < (int)c->comm->members.size() && c->comm->members[c->rank] == c)
        c->comm->members[c->rank] = nullptr;
    for (int i = 0; i < F_COUNT; ++i) { (void)hipFree(c->f[i].buf[0]); (void)hipFree(c->f[i].buf[1]); (void)hipFree(c->f[i].buf[2]); }
    (void)hipFree(c->display); (void)hipFree(c->err_flag); (void)hipFree(c->cfl_dev);
    if (c->cfl_host) (void)hipHostFree(c->cfl_host);
    for (int i = 0; i < 7; ++i) { (void)hipFree(c->part[i]); (void)hipFree(c->part_t[i]); (void)hipFree(c->outbox[i]); }
    (void)hipFree(c->tr_counts_dev); (void)hipFree(c->tr_matrix_dev); (void)hipFree(c->spawn_sums);
    (void)hipFree(c->part_perm); (void)hipFree(c->part_perm_t); (void)hipFree(c->sort_keys); (void)hipFree(c->sort_hist);
    if (c->tr_counts_host) (void)hipHostFree(c->tr_counts_host);
    if (c->tr_matrix_host) (void)hipHostFree(c->tr_matrix_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto *list : {&c->probes_live, &c->probes_free})
        for (const fs_ctx::WaitProbe &pr : *list) { (void)hipEventDestroy(pr.a); (void)hipEventDestroy(pr.b); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_fork3) (void)hipEventDestroy(c->ev_fork3);
    if (c->ev_join3) (void)hipEventDestroy(c->ev_join3);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_band[i]) (void)hipEventDestroy(c->ev_band[i]);
        (void)hipFree(c->p_band[i]);
    }
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    for (int i = 0; i < 2; ++i)
        if (c->ev_cfl[i]) (void)hipEventDestroy(c->ev_cfl[i]);
    if (c->ev_serial) {
        if (c->comm && c->comm->last_serial == c->ev_serial) c->comm->last_serial = nullptr;
        (void)hipEventDestroy(c->ev_serial);
    }
    for (int i = 0; i < SLOT_COUNT; ++i) {
        if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_slot[i]) (void)hipEventDestroy(c->ev_slot[i]);
    }
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FS_OK;
}

// reset(), Solver:371-374 -> clearTextures(), :348-366: all eight simulation textures
// (and here the display too) to zero.
int fs_reset(fs_ctx *c)
{
    CTX(c);
    TRY(c, fs::settle(c));
    c->pending_splats.clear();   // (splats not yet launched would be wiped by the clears below anyway)
    const size_t pb = c->plane * sizeof(float);
    for (int i = 0; i < F_COUNT; ++i) {
        for (int k = 0; k < 3; ++k)
            if (c->f[i].buf[k]) TRYHIP(c, hipMemsetAsync(c->f[i].buf[k], 0, c->f[i].np * pb, c->stream));
        c->f[i].valid = kInf;   // an all-zero field is valid at any ghost depth
    }
    TRYHIP(c, hipMemsetAsync(c->display, 0, 4 * pb, c->stream));
    TRYHIP(c, hipMemsetAsync(c->err_flag, 0, sizeof(int), c->stream));
    for (int &h : c->halo_hist) h = -1;
    c->p_band_dirty = false;
    c->vort_stale = false;
    return FS_OK;
}

int fs_set_params(fs_ctx *c, const fs_params *p)
{
    CTX(c);
    TRY(c, fs::settle(c));
    if (!p) { fs::set_error("params is NULL"); return ctx_fail(c, FS_ERR_INVALID); }
    if (p->solverIterations < 0 || p->solverIterations > 100000) {
        fs::set_error("solverIterations %d out of range", p->solverIterations);
        return ctx_fail(c, FS_ERR_INVALID);
    }
    if (memcmp(&c->params, p, sizeof(*p)) != 0) {
        // Scalars are baked into the recorded kernel nodes.  A change that can alter the PLAN (sweep counts, a stage
        // switched on or off, the display mode) drops the graphs; a change of values only marks them stale, and the
        // next fs_step refreshes the arguments of the instantiated graph in place (hipGraphExecUpdate).
        const fs_params &o = c->params;
        const bool same_plan = o.solverIterations == p->solverIterations && o.displayMode == p->displayMode &&
                               (o.viscosity > 0.0f) == (p->viscosity > 0.0f) && (o.diffusion > 0.0f) == (p->diffusion > 0.0f) &&
                               (o.vorticityStrength > 0.0f) == (p->vorticityStrength > 0.0f) && (o.fadeSpeed > 0.0f) == (p->fadeSpeed > 0.0f);
        c->params = *p;
        c->pparams.viscosity = p->viscosity;
        if (same_plan && !c->graphs.empty()) c->param_epoch += 1;
        else drop_graphs(c);
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
    TRY(c, fs::settle(c));
    TRY(c, flush_splats(c));
    auto set = [](int &field, int v) { const bool changed = field != v; field = v; return changed; };
    bool changed = true;
    switch (option) {
    case FS_OPT_USE_GRAPH: changed = set(c->use_graph, value ? 1 : 0); break;
    case FS_OPT_BLOCK_DEPTH: case FS_OPT_BLOCK_DEPTH_PRESSURE: case FS_OPT_BLOCK_DEPTH_DIFFUSE:
    case FS_OPT_TILE_VARIANT: case FS_OPT_TILE_VARIANT_DIFFUSE: case FS_OPT_STAGGER: case FS_OPT_STAGGER_MODE: case FS_OPT_BAND_DELAY:
        // this context's own copy: another context in the process keeps its tuning (fs_config_set changes the default)
        if (fs::tuning_set(c->tuning, option, value)) { fs::set_error("option %d: value %d out of range", option, value); return ctx_fail(c, FS_ERR_INVALID); }
        break;
    case FS_OPT_WITH_PARTICLES: changed = set(c->with_particles, value ? 1 : 0); break;
    case FS_OPT_FUSE: changed = set(c->fuse, (value >= 0 && value <= 4) ? value : 1); break;
    case FS_OPT_SLAB_OVERLAP: changed = set(c->overlap, value ? 1 : 0); break;
    case FS_OPT_SLAB_PROFILE: c->profile = value ? 1 : 0; changed = false; break;        // (joins are never part of a recorded segment)
    case FS_OPT_SLAB_DEFER: c->defer_check = value ? 1 : 0; changed = false; break;
    case FS_OPT_SEGMENT_GRAPH_AFTER: c->seg_min_stable = value < 0 ? 0 : value; changed = false; break;
    case FS_OPT_SLAB_SERIALIZE:
        if (!c->comm || c->comm->solo()) { fs::set_error("FS_OPT_SLAB_SERIALIZE is a measurement aid of local groups"); return ctx_fail(c, FS_ERR_STATE); }
        if (value && !c->ev_serial) TRYHIP(c, hipEventCreateWithFlags(&c->ev_serial, hipEventDisableTiming));
        c->comm->serialize = value ? 1 : 0;
        c->comm->last_serial = nullptr;
        changed = false;
        break;
    case FS_OPT_SPLIT_PLANES: changed = set(c->split_planes, value ? 1 : 0); break;
    case FS_OPT_SPLIT_ROWS: changed = set(c->split_rows, value ? 1 : 0); break;
    case FS_OPT_SPLAT_IN_GRAPH: c->defer_splats = value ? 1 : 0; changed = false; break;
    case FS_OPT_TRACER_SORT:
        if (value < 0) { fs::set_error("FS_OPT_TRACER_SORT: negative interval"); return ctx_fail(c, FS_ERR_INVALID); }
        c->sort_every = value;
        c->frames_since_sort = 1 << 20;
        changed = false;
        break;
    default: fs::set_error("unknown option %d", option); return ctx_fail(c, FS_ERR_INVALID);
    }
    if (changed) drop_graphs(c);   // (a value that is already in force keeps the recorded frames)
    return FS_OK;
}

int fs_get_option(fs_ctx *c, int option, int *value)
{
    if (!c || !value) { fs::set_error("fs_get_option: NULL argument"); return FS_ERR_INVALID; }
    switch (option) {
    case FS_OPT_USE_GRAPH: *value = c->use_graph; break;
    case FS_OPT_BLOCK_DEPTH: case FS_OPT_BLOCK_DEPTH_PRESSURE: case FS_OPT_BLOCK_DEPTH_DIFFUSE:
    case FS_OPT_TILE_VARIANT: case FS_OPT_TILE_VARIANT_DIFFUSE: case FS_OPT_STAGGER: case FS_OPT_STAGGER_MODE: case FS_OPT_BAND_DELAY:
        *value = fs::tuning_get(c->tuning, option);
        break;
    case FS_OPT_WITH_PARTICLES: *value = c->with_particles; break;
    case FS_OPT_FUSE: *value = c->fuse; break;
    case FS_OPT_SLAB_OVERLAP: *value = c->overlap; break;
    case FS_OPT_SLAB_PROFILE: *value = c->profile; break;
    case FS_OPT_SLAB_DEFER: *value = c->defer_check; break;
    case FS_OPT_SEGMENT_GRAPH_AFTER: *value = c->seg_min_stable; break;
    case FS_OPT_SLAB_SERIALIZE: *value = c->comm ? c->comm->serialize : 0; break;
    case FS_OPT_SPLIT_PLANES: *value = c->split_planes; break;
    case FS_OPT_SPLIT_ROWS: *value = c->split_rows; break;
    case FS_OPT_TRACER_SORT: *value = c->sort_every; break;
    case FS_OPT_SPLAT_IN_GRAPH: *value = c->defer_splats; break;
    default: fs::set_error("unknown option %d", option); return ctx_fail(c, FS_ERR_INVALID);
    }
    return FS_OK;
}

// applyForce, Solver:444-470.  A slab applies the splat to every row it holds (owned + ghost), so
// ghost depths survive a splat.
int fs_add_force(fs_ctx *c, float pos_x, float pos_y, float force_x, float force_y, const float color_rgba[4],
                 float radius, float deltaTime)
{
    CTX(c);
    if (!color_rgba) { fs::set_error("color is NULL"); return ctx_fail(c, FS_ERR_INVALID); }
    // a slab whose last frame still owes its halo check: remember the splat, the repair has to repeat it (slab.hip)
    const fs_ctx::Splat sp{pos_x, pos_y, force_x, force_y, color_rgba[0], color_rgba[1], color_rgba[2], radius, deltaTime};
    if (c->chk_pending) c->splat_log.push_back(sp);
    // one GPU, frames replayed: the splat becomes a node of the next frame's graph (fs_step) instead of a launch of its own
    if (c->world == 1 && c->defer_splats && c->use_graph && c->stable_frames >= 2 && !c->dry &&
        (int)c->pending_splats.size() < fs_ctx::kMaxPendingSplats && std::isfinite(pos_x) && std::isfinite(pos_y) && std::isfinite(radius)) {
        c->pending_splats.push_back(sp);
        return FS_OK;
    }
    TRY(c, flush_splats(c));
    TRY(c, launch_splat(c, sp));
    return FS_OK;
}

static unsigned long long pingpong_key(const fs_ctx *c)
{
    return (unsigned long long)c->f[F_VEL].cur | ((unsigned long long)c->f[F_DYE].cur << 1) |
           ((unsigned long long)c->f[F_P].cur << 2);
}
static void set_pingpong(fs_ctx *c, unsigned long long k)
{
    const int cur[3] = {(int)(k & 1), (int)((k >> 1) & 1), (int)((k >> 2) & 1)};
    fs::Field *fl[3] = {&c->f[F_VEL], &c->f[F_DYE], &c->f[F_P]};
    for (int i = 0; i < 3; ++i) { fl[i]->cur = cur[i]; fl[i]->oth = cur[i] ^ 1; }   // (one GPU: the roles never rotate)
}

int fs_step(fs_ctx *c)
{
    CTX(c);
    if (c->world > 1) {
        if (!c->comm || !c->comm->solo()) {
            fs::set_error("fs_step: a slab of an in-process group is stepped by fs_group_step");
            return ctx_fail(c, FS_ERR_STATE);
        }
        TRY(c, fs::run_frame(c));
        return FS_OK;
    }
    fs::ensure_band_planes(c, c->params.solverIterations);
    if (c->with_particles && c->n_particles > 0) TRY(c, tracer_maybe_sort(c));
    struct Mark { fs_ctx *c; ~Mark() { c->p_band_dirty = c->p_band_active; } } mark{c};   // after the frame, however it ran
    // Scalars are baked into the recorded kernel nodes, so a change of parameters invalidates the graphs.  A host
    // that drags a slider changes them every frame: frames run eagerly (as fast as a replay on large grids, 40 us of
    // host time) until the parameters have stayed put for two frames, and only then is a graph recorded again.
    if (!c->use_graph || c->stable_frames < 2) {
        TRY(c, flush_splats(c));
        TRY(c, fs::run_frame(c));
        c->stable_frames += 1;
        return FS_OK;
    }
    // (from here on a change of values costs one host-side recording per ping-pong state and no instantiation)
    // Buffer roles alternate between frames when a swap count is odd, so a captured graph
    // is keyed by the ping-pong state it was recorded from -- and by the number of splats that lead it.
    const size_t k = c->pending_splats.size();
    const unsigned long long key = pingpong_key(c) | ((unsigned long long)(c->with_particles ? tracer_bound(c) : 0) << 3) |
                                   ((unsigned long long)k << 56) |
                                   ((unsigned long long)(c->with_particles ? c->part_flip : 0) << 62);   // (which of the two sets of tracer arrays)
    // argument blocks of the k splat nodes for THIS frame (Solver:444-470: position, force, colour, radius, dt; the fields'
    // current sides)
    c->splat_scratch.resize(k);
    {
        const fs_grid g = fs::grid_of(c, c->G);
        float *u = c->f[F_VEL].p(), *v = u + c->plane, *d = c->f[F_DYE].p();
        for (size_t i = 0; i < k; ++i) {
            const fs_ctx::Splat &sp = c->pending_splats[i];
            const float f[9] = {sp.px, sp.py, sp.fx, sp.fy, sp.r, sp.g, sp.b, sp.radius, sp.dt};
            fs::splat_node_params(c->splat_scratch[i], &g, u, v, d, d + c->plane, d + 2 * c->plane, f);
        }
    }
    auto it = c->graphs.find(key);
    if (it != c->graphs.end() && c->graphs_epoch[key] == c->param_epoch) {
        const std::vector<hipGraphNode_t> &nodes = c->graphs_splat_nodes[key];
        for (size_t i = 0; i < k; ++i) TRYHIP(c, hipGraphExecKernelNodeSetParams(it->second, nodes[i], &c->splat_scratch[i].np));
        c->pending_splats.clear();
        TRYHIP(c, hipGraphLaunch(it->second, c->stream));
        set_pingpong(c, c->graphs_after[key]);
        c->stat_frames += 1;
        return FS_OK;
    }
    // record the frame: a new graph for this ping-pong state, or new kernel arguments for the one that exists
    const unsigned long long before = pingpong_key(c);
    hipGraph_t graph = nullptr;
    TRYHIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    int rc = fs::run_frame(c);
    hipError_t e = hipStreamEndCapture(c->stream, &graph);
    std::vector<hipGraphNode_t> splat_nodes(k);
    if (!rc && e == hipSuccess && k > 0) {
        // the splats in front of everything the frame recorded: a chain of k kernel nodes, every root of the frame behind it
        size_t nroots = 0;
        e = hipGraphGetRootNodes(graph, nullptr, &nroots);
        std::vector<hipGraphNode_t> roots(nroots);
        if (e == hipSuccess && nroots) e = hipGraphGetRootNodes(graph, roots.data(), &nroots);
        for (size_t i = 0; i < k && e == hipSuccess; ++i)
            e = hipGraphAddKernelNode(&splat_nodes[i], graph, i ? &splat_nodes[i - 1] : nullptr, i ? 1 : 0, &c->splat_scratch[i].np);
        for (size_t r = 0; r < nroots && e == hipSuccess; ++r) e = hipGraphAddDependencies(graph, &splat_nodes[k - 1], &roots[r], 1);
    }
    if (rc || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        set_pingpong(c, before);
        if (rc) return ctx_fail(c, rc);
        return ctx_fail(c, fs::hip_fail(e, "recording the frame's graph"));
    }
    hipGraphExec_t exec = it != c->graphs.end() ? it->second : nullptr;
    bool keep_graph = false;
    if (exec) {
        hipGraphNode_t bad = nullptr;
        hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
        if (hipGraphExecUpdate(exec, graph, &bad, &res) == hipSuccess && res == hipGraphExecUpdateSuccess)
            c->stat_graph_updates += 1;   // (the exec still answers to the nodes of the graph it was instantiated from)
        else {   // the plan changed after all (or the runtime cannot update this node): instantiate afresh
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(exec);
            c->graphs.erase(key);
            auto src = c->graphs_src.find(key);
            if (src != c->graphs_src.end()) { (void)hipGraphDestroy(src->second); c->graphs_src.erase(src); }
            exec = nullptr;
        }
    }
    if (!exec) {
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            (void)hipGraphDestroy(graph);
            set_pingpong(c, before);
            return ctx_fail(c, fs::hip_fail(e, "hipGraphInstantiate"));
        }
        c->stat_graph_instantiations += 1;
        c->graphs_src[key] = graph;
        c->graphs_splat_nodes[key] = splat_nodes;
        keep_graph = true;
    }
    if (!keep_graph) (void)hipGraphDestroy(graph);
    // remember where the roles end up so replays can update them without re-recording
    c->graphs[key] = exec;
    c->graphs_after[key] = pingpong_key(c);
    c->graphs_epoch[key] = c->param_epoch;
    c->pending_splats.clear();
    TRYHIP(c, hipGraphLaunch(exec, c->stream));
    return FS_OK;
}

int fs_graph_stats(fs_ctx *c, long stats[3])
{
    if (!c || !stats) return FS_ERR_INVALID;
    stats[0] = (long)c->graphs.size();
    stats[1] = c->stat_graph_updates;
    stats[2] = c->stat_graph_instantiations;
    return FS_OK;
}

int fs_sync(fs_ctx *c)
{
    CTX(c);
    TRY(c, fs::settle(c));
    TRY(c, flush_splats(c));
    if (c->xstream) TRYHIP(c, hipStreamSynchronize(c->xstream));
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

// After a frame projected in two row bands the lower band's pressure still sits in the band's private plane: bring
// it home before anything looks at the pressure field (nothing in update() does: p is cleared every frame).
static int settle_pressure(fs_ctx *c)
{
    if (!c->p_band_dirty) return FS_OK;
    c->p_band_dirty = false;
    const size_t o = (size_t)(c->p_band_row - c->b0) * c->pitch, n = (size_t)(c->r1 - c->p_band_row) * c->pitch;
    TRYHIP(c, hipMemcpyAsync(c->f[F_P].p() + o, c->p_band[c->p_band_final] + o, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return FS_OK;
}

// Host copies move the rows this context OWNS: dense planar [plane][r1-r0][width] (the whole field on one GPU).
static int copy_field(fs_ctx *c, int field, float *host, size_t bytes, bool to_host, const char *who)
{
    const int np = nplanes_of(field);
    if (!np || !host) { fs::set_error("%s: bad field %d or NULL host pointer", who, field); return ctx_fail(c, FS_ERR_INVALID); }
    const size_t rows = (size_t)(c->r1 - c->r0);
    const size_t need = (size_t)np * c->W * rows * sizeof(float);
    if (bytes != need) { fs::set_error("%s: expected %zu bytes, got %zu", who, need, bytes); return ctx_fail(c, FS_ERR_INVALID); }
    if (field == FS_FIELD_PRESSURE) TRY(c, settle_pressure(c));
    if (field == FS_FIELD_VORTICITY && to_host && c->vort_stale) {
        // (the reference's vorticity texture holds the curl of the velocity update() started from: not something that can
        // be recomputed from the fields a finished frame leaves)
        fs::set_error("%s: the last frame computed curl and confinement in one pass and did not write the vorticity plane; "
                      "FS_OPT_FUSE 0, the vorticity display mode or fs_stage keep it", who);
        return ctx_fail(c, FS_ERR_STATE);
    }
    if (field == FS_FIELD_VORTICITY && !to_host) c->vort_stale = false;
    TRYHIP(c, hipStreamSynchronize(c->stream));
    float *dev = field_ptr(c, field) + (size_t)(c->r0 - c->b0) * c->pitch;
    const size_t hp = (size_t)c->W * sizeof(float), dp = (size_t)c->pitch * sizeof(float);
    for (int p = 0; p < np; ++p) {
        float *h = host + (size_t)p * c->W * rows;
        float *d = dev + p * c->plane;
        if (to_host) TRYHIP(c, hipMemcpy2D(h, hp, d, dp, hp, rows, hipMemcpyDeviceToHost));
        else TRYHIP(c, hipMemcpy2D(d, dp, h, hp, hp, rows, hipMemcpyHostToDevice));
    }
    if (!to_host && c->world > 1) {
        const int i = field_index(field);
        if (i >= 0) c->f[i].valid = 0;   // the ghost rows no longer mirror the neighbours
    }
    return FS_OK;
}

int fs_read_field(fs_ctx *c, int field, float *host, size_t bytes)
{
    CTX(c);
    TRY(c, fs::settle(c));
    TRY(c, flush_splats(c));
    return copy_field(c, field, host, bytes, true, "fs_read_field");
}

int fs_write_field(fs_ctx *c, int field, const float *host, size_t bytes)
{
    CTX(c);
    TRY(c, fs::settle(c));
    TRY(c, flush_splats(c));
    return copy_field(c, field, const_cast<float *>(host), bytes, false, "fs_write_field");
}

int fs_field_device_ptr(fs_ctx *c, int field, int plane, void **dev_ptr, int *pitch)
{
    if (!c || !dev_ptr) return FS_ERR_INVALID;
    TRY(c, fs::settle(c));
    TRY(c, flush_splats(c));
    const int np = nplanes_of(field);
    if (plane < 0 || plane >= np) { fs::set_error("bad field/plane %d/%d", field, plane); return ctx_fail(c, FS_ERR_INVALID); }
    if (field == FS_FIELD_PRESSURE) TRY(c, settle_pressure(c));
    *dev_ptr = field_ptr(c, field) + (size_t)plane * c->plane;
    if (pitch) *pitch = c->pitch;
    return FS_OK;
}

void *fs_stream(fs_ctx *c)
{
    if (c && !c->pending_splats.empty()) { (void)hipSetDevice(c->device); (void)flush_splats(c); }   // the caller is about to order its own work behind ours
    return c ? (void *)c->stream : nullptr;
}

int fs_timer_begin(fs_ctx *c)
{
    CTX(c);
    TRY(c, flush_splats(c));
    if (!c->ev0) { TRYHIP(c, hipEventCreate(&c->ev0)); TRYHIP(c, hipEventCreate(&c->ev1)); }
    TRYHIP(c, hipEventRecord(c->ev0, c->stream));
    return FS_OK;
}

int fs_timer_end(fs_ctx *c, float *ms)
{
    CTX(c);
    if (!c->ev0 || !ms) { fs::set_error("fs_timer_end without fs_timer_begin"); return ctx_fail(c, FS_ERR_STATE); }
    TRY(c, flush_splats(c));
    TRYHIP(c, hipEventRecord(c->ev1, c->stream));
    TRYHIP(c, hipEventSynchronize(c->ev1));
    TRYHIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return FS_OK;
}

int fs_stage(fs_ctx *c, int stage, int n)
{
    CTX(c);
    TRY(c, fs::settle(c));
    TRY(c, flush_splats(c));
    SINGLE(c, "fs_stage");
    if (n < 0) { fs::set_error("negative sweep count"); return ctx_fail(c, FS_ERR_INVALID); }
    TRY(c, settle_pressure(c));
    float *u = c->f[F_VEL].p(), *v = u + c->plane;
    int rc;
    switch (stage) {
    case FS_STAGE_VORTICITY: rc = fs_k_calculateVorticity(&c->grid, u, v, c->f[F_VORT].p(), c->stream); c->vort_stale = false; break;   // Solver:488-491
    case FS_STAGE_APPLY_VORTICITY:
        rc = fs_k_applyVorticity(&c->grid, c->f[F_VORT].p(), u, v, c->params.vorticityStrength, c->params.deltaTime, c->stream);   // :494-501
        break;
    case FS_STAGE_DIFFUSE_VELOCITY: rc = st_sweeps(c, F_VEL, n, false, false); break;
    case FS_STAGE_DIVERGENCE: rc = fs_k_divergence(&c->grid, u, v, c->f[F_DIV].p(), c->stream); break;              // :522-525
    case FS_STAGE_CLEAR_PRESSURE: rc = fs_k_clear(&c->grid, c->f[F_P].p(), 1, 0, c->stream); break;                 // :528-530
    case FS_STAGE_PRESSURE: rc = st_sweeps(c, F_P, n, true, false); break;
    case FS_STAGE_SUBTRACT_GRADIENT: rc = fs_k_subtractGradient(&c->grid, c->f[F_P].p(), u, v, c->stream); break;   // :544-547
    case FS_STAGE_ADVECT_VELOCITY: rc = st_advect(c, F_VEL); break;
    case FS_STAGE_DIFFUSE_DENSITY: rc = st_sweeps(c, F_DYE, n, false, false); break;
    case FS_STAGE_ADVECT_DENSITY: rc = st_advect(c, F_DYE); break;
    case FS_STAGE_FADE: rc = fs_k_fadeDensity(&c->grid, c->f[F_DYE].p(), 3, c->plane, c->params.fadeSpeed, c->stream); break;   // :587-591
    case FS_STAGE_RENDER: rc = st_render(c); break;
    case FS_STAGE_PRESSURE_CLEARED: rc = st_sweeps(c, F_P, n, true, true); break;                                  // :528-541
    case FS_STAGE_PROJECT: rc = st_project(c, n); break;                                                             // :522-547
    default: fs::set_error("unknown stage %d", stage); rc = FS_ERR_INVALID;
    }
    return rc ? ctx_fail(c, rc) : FS_OK;
}

int fs_particles_reserve(fs_ctx *c, long capacity)
{
    CTX(c);
    TRY(c, fs::settle(c));
    if (capacity < 0) { fs::set_error("negative tracer capacity"); return ctx_fail(c, FS_ERR_INVALID); }
    TRYHIP(c, hipStreamSynchronize(c->stream));
    const bool slab = c->world > 1;
    for (int i = 0; i < 7; ++i) {
        (void)hipFree(c->part[i]); (void)hipFree(c->part_t[i]); (void)hipFree(c->outbox[i]);
        c->part[i] = c->part_t[i] = c->outbox[i] = nullptr;
    }
    c->n_particles = 0;
    c->max_particles = 0;
    c->part_sorted = false;
    c->removed_pending = false;
    (void)hipFree(c->part_perm); (void)hipFree(c->part_perm_t); (void)hipFree(c->sort_keys);
    c->part_perm = c->part_perm_t = c->sort_keys = nullptr;
    c->sort_cap = 0;
    drop_graphs(c);
    for (int i = 0; i < 7 && capacity > 0; ++i) {
        TRYHIP(c, hipMalloc((void **)&c->part[i], (size_t)capacity * sizeof(float)));
        if (slab) {
            TRYHIP(c, hipMalloc((void **)&c->part_t[i], (size_t)capacity * sizeof(float)));
            TRYHIP(c, hipMalloc((void **)&c->outbox[i], (size_t)capacity * sizeof(float)));
        }
    }
    if (slab && capacity > 0 && !c->tr_counts_dev) {
        const size_t w = (size_t)c->world;
        TRYHIP(c, hipMalloc((void **)&c->tr_counts_dev, w * sizeof(int)));
        TRYHIP(c, hipHostMalloc((void **)&c->tr_counts_host, w * sizeof(int), hipHostMallocDefault));
        TRYHIP(c, hipMalloc((void **)&c->tr_matrix_dev, w * w * sizeof(int)));
        TRYHIP(c, hipHostMalloc((void **)&c->tr_matrix_host, w * w * sizeof(int), hipHostMallocDefault));
    }
    c->max_particles = capacity;
    if (capacity > 0) TRYHIP(c, hipMemsetAsync(c->part[4], 0, (size_t)capacity * sizeof(float), c->stream));   // (life 0 = an empty slot)
    return FS_OK;
}

int fs_set_particle_params(fs_ctx *c, const fs_particle_params *pp)
{
    CTX(c);
    TRY(c, fs::settle(c));
    if (!pp) { fs::set_error("particle params is NULL"); return ctx_fail(c, FS_ERR_INVALID); }
    if (memcmp(&c->pparams, pp, sizeof(*pp)) != 0) { c->pparams = *pp; if (c->with_particles) drop_graphs(c); }
    return FS_OK;
}

// One GPU: the n tracers.  A slab: every rank is handed the SAME n tracers and keeps the index block
// [rank n / world, (rank+1) n / world); the next tracer step moves each to the rank that owns its row.
int fs_particles_set(fs_ctx *c, long n, const float *px, const float *py, const float *vx, const float *vy,
                     const float *life, const float *mass, const float *age)
{
    CTX(c);
    TRY(c, fs::settle(c));
    if (n < 0) { fs::set_error("negative particle count"); return ctx_fail(c, FS_ERR_INVALID); }
    const long lo = c->world > 1 ? (long)((__int128)n * c->rank / c->world) : 0;
    const long hi = c->world > 1 ? (long)((__int128)n * (c->rank + 1) / c->world) : n;
    const long mine = hi - lo;
    if (mine > c->max_particles) { fs::set_error("particle count %ld exceeds capacity %ld", mine, c->max_particles); return ctx_fail(c, FS_ERR_INVALID); }
    const float *src[7] = {px, py, vx, vy, life, mass, age};
    TRY(c, tracer_settle_removed(c));
    TRY(c, tracer_restore_order(c));
    TRYHIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 7 && mine > 0; ++i) {
        if (!src[i]) { fs::set_error("NULL particle array %d", i); return ctx_fail(c, FS_ERR_INVALID); }
        TRYHIP(c, hipMemcpy(c->part[i], src[i] + lo, (size_t)mine * sizeof(float), hipMemcpyHostToDevice));
    }
    if (mine != c->n_particles) drop_graphs(c);
    c->n_particles = mine;
    if (c->world == 1) TRY(c, clear_life_tail(c, c->part[4], mine));
    return FS_OK;
}

int fs_particles_get(fs_ctx *c, long n, float *px, float *py, float *vx, float *vy, float *life, float *mass,
                     float *age)
{
    CTX(c);
    TRY(c, fs::settle(c));
    TRY(c, tracer_settle_removed(c));
    if (n < 0 || n > c->n_particles) { fs::set_error("asked for %ld of %ld particles", n, c->n_particles); return ctx_fail(c, FS_ERR_INVALID); }
    float *dst[7] = {px, py, vx, vy, life, mass, age};
    TRY(c, tracer_restore_order(c));
    TRYHIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 7 && n > 0; ++i)
        if (dst[i]) TRYHIP(c, hipMemcpy(dst[i], c->part[i], (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return FS_OK;
}

// (Blocking when a spawn has put its compaction off: the count is the reference's particleCount, live slots as of the last spawn.)
long fs_particles_count(fs_ctx *c)
{
    if (!c) return 0;
    if (c->removed_pending && hipSetDevice(c->device) == hipSuccess) (void)tracer_settle_removed(c);
    return c->n_particles;
}

// The reference's own array layout: [Particle] as Swift and MSL lay it out (Part:18-31, Shaders:419-426): position float2 @0,
// velocity float2 @8, color float4 @16, life @32, mass @36, age @40, stride 48 (any stride >= 44 is accepted).  The update
// never reads `color` (constant white, Part:215) and the line vertices take their alpha from `life` (Shaders:472), so it
// is not carried on the device: set ignores it, get leaves the caller's bytes alone.
namespace {
constexpr size_t kAosOff[7] = {0, 4, 8, 12, 32, 36, 40};   // px py vx vy life mass age
}
int fs_particles_set_aos(fs_ctx *c, long n, const void *particles, size_t stride)
{
    if (!c) { fs::set_error("ctx is NULL"); return FS_ERR_INVALID; }
    if (n < 0 || (n > 0 && !particles) || stride < 44 || (stride & 3)) { fs::set_error("fs_particles_set_aos: bad arguments (stride %zu)", stride); return ctx_fail(c, FS_ERR_INVALID); }
    std::vector<float> soa((size_t)n * 7);
    const char *src = static_cast<const char *>(particles);
    for (long i = 0; i < n; ++i)
        for (int f = 0; f < 7; ++f) memcpy(&soa[(size_t)f * n + i], src + (size_t)i * stride + kAosOff[f], sizeof(float));
    const float *p = soa.data();
    return fs_particles_set(c, n, p, p + n, p + 2 * n, p + 3 * n, p + 4 * n, p + 5 * n, p + 6 * n);
}
int fs_particles_get_aos(fs_ctx *c, long n, void *particles, size_t stride)
{
    if (!c) { fs::set_error("ctx is NULL"); return FS_ERR_INVALID; }
    if (n < 0 || (n > 0 && !particles) || stride < 44 || (stride & 3)) { fs::set_error("fs_particles_get_aos: bad arguments (stride %zu)", stride); return ctx_fail(c, FS_ERR_INVALID); }
    std::vector<float> soa((size_t)n * 7);
    float *p = soa.data();
    int rc = fs_particles_get(c, n, p, p + n, p + 2 * n, p + 3 * n, p + 4 * n, p + 5 * n, p + 6 * n);
    if (rc) return rc;
    char *dst = static_cast<char *>(particles);
    for (long i = 0; i < n; ++i)
        for (int f = 0; f < 7; ++f) memcpy(dst + (size_t)i * stride + kAosOff[f], &soa[(size_t)f * n + i], sizeof(float));
    return FS_OK;
}

// ParticleSystem.addParticles(at:count:windowSize:), Part:186-231, on the device.  The reference's array holds the live
// tracers in age order; `count` appends with removeFirst() at the cap (:196-199) leave the last `cap` of (live + new).
// While there is room for the new tracers behind the slots in use: the slots that are dead now are marked removed, the
// new tracers are appended, nothing comes back to the host and the frame graphs stay (the compaction is owed to the next
// call that shows the arrays or the count: tracer_settle_removed).  Otherwise: count the live slots, drop the dead ones and the
// oldest that do not fit (order kept), write the new tracers behind them -- blocking, the live count comes back to the host.
int fs_particles_spawn(fs_ctx *c, float x, float y, long count, float radius, unsigned seed)
{
    CTX(c);
    if (count < 0) { fs::set_error("negative spawn count"); return ctx_fail(c, FS_ERR_INVALID); }
    const long cap = c->max_particles;
    if (cap == 0) return FS_OK;   // (count == 0 still drops the dead slots: the reference's array never holds any)
    TRY(c, ensure_part_t(c));
    TRY(c, ensure_spawn_sums(c));
    if (c->world > 1) {
        // a slab: the same call on every rank, each appends ITS index block of the new tracers (tracer j's draws depend on
        // (seed, j) alone), wherever they lie -- the next fs_step re-homes them like every other tracer.  A slab's arrays
        // are in no order (the re-homing scatters), so there is no "oldest" to push out: dead slots are dropped when the
        // room runs out, and a slab that is full of live tracers refuses.
        TRY(c, fs::settle(c));
        const long lo = (long)((__int128)count * c->rank / c->world), hi = (long)((__int128)count * (c->rank + 1) / c->world);
        const long mine = hi - lo;
        long n = c->n_particles;
        if (n + mine > cap && n > 0) {
            TRY(c, fs::tracer_count_live(n, c->part[4], c->spawn_sums, false, c->stream));
            int live = 0;
            TRYHIP(c, hipMemcpyAsync(&live, c->spawn_sums + fs::tracer_blocks(n), sizeof(int), hipMemcpyDeviceToHost, c->stream));
            TRYHIP(c, hipStreamSynchronize(c->stream));
            if ((long)live + mine <= cap) {
                TRY(c, fs::tracer_compact(n, c->part, c->part_t, c->spawn_sums, 0, false, c->stream));
                for (int i = 0; i < 7; ++i) std::swap(c->part[i], c->part_t[i]);
                c->n_particles = n = live;
            }
        }
        if (n + mine > cap) {
            fs::set_error("rank %d would hold %ld tracers after the spawn, its capacity is %ld (fs_particles_reserve)", c->rank, n + mine, cap);
            return ctx_fail(c, FS_ERR_NOMEM);
        }
        TRY(c, fs::tracer_spawn(n, lo, hi, c->part, x, y, radius, seed, nullptr, c->stream));
        c->n_particles = n + mine;
        return FS_OK;
    }
    if (c->n_particles + count <= cap) {
        const long n = c->n_particles;
        TRY(c, fs::tracer_mark_removed(n, c->part[4], c->stream));
        // (sorted arrays: the new tracers go behind them, their slots are their positions)
        TRY(c, fs::tracer_spawn(n, 0, count, c->part, x, y, radius, seed, c->part_sorted ? c->part_perm : nullptr, c->stream));
        c->removed_pending = true;
        const long bound = tracer_bound(c);
        c->n_particles = n + count;
        if (c->with_particles && tracer_bound(c) != bound) drop_graphs(c);   // (the tracer node's slot count is rounded up: rarely)
        return FS_OK;
    }
    TRY(c, tracer_settle_removed(c));
    TRY(c, tracer_restore_order(c));   // the slots are in age order, which decides who is dropped at the cap
    const long n = c->n_particles;
    long live = 0;
    if (n > 0) {
        TRY(c, fs::tracer_count_live(n, c->part[4], c->spawn_sums, false, c->stream));
        int total = 0;
        TRYHIP(c, hipMemcpyAsync(&total, c->spawn_sums + fs::tracer_blocks(n), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        TRYHIP(c, hipStreamSynchronize(c->stream));
        live = total;
    }
    const long after = std::min(cap, live + count);                   // Part:229 particleCount
    const long drop = std::max(0L, live + count - cap);               // oldest tracers pushed out at the cap, :196-199
    const long kept = live - std::min(drop, live);
    if (n > 0) TRY(c, fs::tracer_compact(n, c->part, c->part_t, c->spawn_sums, drop, false, c->stream));
    const long j0 = std::max(0L, count - cap);                        // more new tracers than slots: the first ones are pushed out too
    TRY(c, fs::tracer_spawn(kept, j0, count, c->part_t, x, y, radius, seed, nullptr, c->stream));
    TRY(c, clear_life_tail(c, c->part_t[4], after));
    for (int i = 0; i < 7; ++i) std::swap(c->part[i], c->part_t[i]);
    c->part_flip ^= 1;
    c->n_particles = after;
    if (c->with_particles) drop_graphs(c);   // array pointers and the count are baked into the tracer node
    return FS_OK;
}

int fs_particle_lines(fs_ctx *c, float viewportW, float viewportH, float particleScale, float *host_xy4, float *host_alpha)
{
    CTX(c);
    TRY(c, fs::settle(c));
    TRY(c, tracer_settle_removed(c));
    const long n = c->n_particles;
    if (n == 0) return FS_OK;
    if (!host_xy4 || !host_alpha) { fs::set_error("fs_particle_lines: NULL output"); return ctx_fail(c, FS_ERR_INVALID); }
    TRY(c, tracer_restore_order(c));
    float *xy = nullptr, *al = nullptr;
    TRYHIP(c, hipMalloc((void **)&xy, (size_t)n * 5 * sizeof(float)));
    al = xy + 4 * n;
    int rc = fs_k_particle_lines(n, c->part[0], c->part[1], c->part[2], c->part[3], c->part[4], viewportW, viewportH,
                                 particleScale, xy, al, c->stream);
    hipError_t e = rc ? hipSuccess : hipStreamSynchronize(c->stream);
    if (!rc && e == hipSuccess) e = hipMemcpy(host_xy4, xy, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess) e = hipMemcpy(host_alpha, al, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(xy);
    if (rc) return ctx_fail(c, rc);
    if (e != hipSuccess) return ctx_fail(c, fs::hip_fail(e, "fs_particle_lines"));
    return FS_OK;
}

// fs_particle_lines without any allocation: the vertices are staged in the device's copy of `scratch`
// (resolved by fs_create).  Every call on a GPU shares that one array: the kernel overwrites the 5 n floats it
// uses, so nothing has to be cleared, but two calls that overlap on one GPU (two threads, two contexts) race.
int fs_particle_lines_static(fs_ctx *c, float viewportW, float viewportH, float particleScale, float *host_xy4, float *host_alpha)
{
    CTX(c);
    TRY(c, fs::settle(c));
    TRY(c, tracer_settle_removed(c));   // (allocates nothing either)
    const long n = c->n_particles;
    if (n == 0) return FS_OK;
    if (!host_xy4 || !host_alpha) { fs::set_error("fs_particle_lines_static: NULL output"); return ctx_fail(c, FS_ERR_INVALID); }
    if (n > FS_LINES_STATIC_MAX) {
        fs::set_error("fs_particle_lines_static: %ld tracers exceed the static staging area (FS_LINES_STATIC_MAX = %d); "
                      "use fs_particle_lines", n, FS_LINES_STATIC_MAX);
        return ctx_fail(c, FS_ERR_INVALID);
    }
    TRY(c, tracer_restore_order(c));   // (allocates nothing: the partner arrays exist whenever the order is not the slots')
    float *xy = c->lines_scratch;
    if (!xy) { fs::set_error("fs_particle_lines_static: no staging area on device %d", c->device); return ctx_fail(c, FS_ERR_STATE); }
    float *al = xy + 4 * n;
    TRY(c, fs_k_particle_lines(n, c->part[0], c->part[1], c->part[2], c->part[3], c->part[4], viewportW, viewportH,
                               particleScale, xy, al, c->stream));
    TRYHIP(c, hipStreamSynchronize(c->stream));
    TRYHIP(c, hipMemcpy(host_xy4, xy, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    TRYHIP(c, hipMemcpy(host_alpha, al, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return FS_OK;
}

int fs_particles_update(fs_ctx *c, const fs_particle_params *pp)
{
    CTX(c);
    SINGLE(c, "fs_particles_update (on a slab the tracer step is part of fs_step, FS_OPT_WITH_PARTICLES)");
    if (pp) {
        if (memcmp(&c->pparams, pp, sizeof(*pp)) != 0) { c->pparams = *pp; if (c->with_particles) drop_graphs(c); }
    }
    if (c->n_particles == 0) return FS_OK;   // Part:243 guard
    TRY(c, flush_splats(c));   // (the lookup reads the velocity the splats write)
    TRY(c, tracer_maybe_sort(c));
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

int fs_config_set(int key, int value) { return fs::tuning_set(fs::default_tuning(), key, value); }
int fs_config_get(int key) { return fs::tuning_get(fs::default_tuning(), key); }

}  // extern "C"
