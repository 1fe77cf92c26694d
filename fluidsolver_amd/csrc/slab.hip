// slab.hip -- frame planner of update() (Solver:477-627) for a row slab, and the halo transports.
//
// The reference has one GPU and no communication (Solver:269); this layer is new work whose contract
// is SURVEY.md 8(e):
//   * GPU g owns the contiguous rows [r0, r1) and all columns; its buffers also hold up to G ghost
//     rows on each interior side, so halo rows are contiguous in memory.
//   * Every stage is the SAME fs_k_* kernel the single-GPU path runs, launched on a row range, so a
//     k-slab run equals the 1-GPU run bit for bit.
//   * Ghost rows are not refreshed after every stencil.  Each field carries a valid depth; a radius-1
//     stencil computed on the owned rows +- d consumes one row of depth.  Exchanges are planned lazily:
//     one round of velocity at frame start (2N+3 rows for vorticity -> N diffuse sweeps -> divergence
//     -> N pressure sweeps -> gradient) and one of velocity + dye before advection.  xGMI halo traffic
//     is latency-bound (a 16384-column row is 64 KiB ~ 0.4 us of wire time), so two rounds per frame
//     beat one per sweep, and the redundant ghost-row compute is cheap.
//   * The semi-Lagrangian backtrace is clamped only to the domain (Shaders:133-137), so the advection
//     halo is sized per frame from the CFL bound: h = ceil(max|v_y| dt) + 2, max over all ranks (one
//     4-byte all-reduce per frame).
//   * Overlap (FS_OPT_SLAB_OVERLAP, default on).  Measured on the scripted workloads the advection halo
//     is 15-50 rows at 3072^2 and 70-170 rows at 16384^2 (profiles/r01_probe_cfl.log): tens of MB per
//     neighbour and frame, too much to leave on the critical path.  So (1) every transfer runs on a
//     second stream and the kernels wait for it by event only where they need the rows; (2) the
//     velocity advection also computes the 2N+3 ghost rows the NEXT frame's stencil chain needs (same
//     kernel, same inputs as the owner => same bits), so the frame-start exchange disappears into the
//     pre-advection one (depth h + 2N+3); (3) advection is split into interior rows, whose backtrace
//     cannot leave the owned rows, and two rims: the interior of the velocity and of the dye run while
//     the halo is in flight; (4) the halo depth is SPECULATED from the last four frames (1.25 max + 4), so
//     the dye halo is fetched at frame start, under the whole stencil chain, and the host never waits
//     for the CFL value before enqueueing; advection writes to the ping-pong partner, so when the
//     all-reduced bound turns out larger than the guess the advection part of the frame is simply
//     planned again with the true depth (stat_redo counts it).
//
// A frame is first PLANNED (pure host bookkeeping of ghost depths and ping-pong roles; identical on
// every rank) into a list of steps -- kernel launches, exchanges, the CFL reduction -- then executed.
// On one GPU the plan has no exchange and the whole list is captured as a hipGraph (context.hip).
// With RCCL a rank walks its list; an in-process group (tests) walks K lists in lock step.
// fluidsolver_amd/slabs.py is the same planner in Python (executable specification, tested on the CPU
// under gloo against the unsplit oracle); tests compare the two plans line by line.
#include "ctx.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

namespace fs {


void slab_rows(int H, int world, int rank, int *r0, int *r1)
{
    const int base = H / world, rem = H % world;   // the first H % world ranks own one extra row
    *r0 = rank * base + std::min(rank, rem);
    *r1 = *r0 + base + (rank < rem ? 1 : 0);
}

// fs_grid computing the owned rows +- depth (clipped to the buffer, hence to the domain)
fs_grid grid_of(const fs_ctx *c, int depth)
{
    depth = std::min(depth, c->G);
    const int yb = std::max(c->b0, c->r0 - depth), ye = std::min(c->b1, c->r1 + depth);
    return fs_grid{c->W, c->H, c->pitch, c->b0, c->b1 - c->b0, yb, ye};
}

// Two bands pay when one launch of the loop leaves holes worth filling: a single GPU, the fused projection (the loop's
// planes in the Infinity Cache), at least two launches, and enough tiles that each band still fills the chip once.
bool bands_wanted(const fs_ctx *c, int N)
{
    if (!c->split_rows || !c->stream2 || c->dry) return false;
    // (launched eagerly the fork / join events cost more than the bands gain: frames that are never replayed keep one chain;
    // FS_STAGE_PROJECT, a test and measurement hook, runs the bands either way)
    if (jacobi_launch_count(c->tuning, 0, N) < 2) return false;
    // a slab: its own rows (the nominal height, so that every rank decides alike) plus the ghost rows the loop reads
    const int rows = c->world > 1 ? c->H / c->world : c->H;
    const double loop_rows = c->world > 1 ? (double)(rows + 2 * (N + 1)) : (double)rows;
    if (!(c->fuse >= 3 || c->fuse == 1)) return false;
    // (planes of half a GB and more: one chain, the bands cost 0.5 % at 16384^2; below, -1.5 % at 8192^2 .. -4.5 % at 6144^2,
    // profiles/r03_z_projection_beyond_cache.log)
    if ((double)c->pitch * loop_rows * sizeof(float) >= 512.0 * 1024 * 1024) return false;
    if (c->world > 1 && c->G < N + 1) return false;   // the fused projection needs N + 1 ghost rows of velocity
    const int T = std::min(c->tuning.depth[0], N), oh = std::max(8, 64 - 2 * T);   // (the default 256 x 64 tile)
    const long tiles = (long)((c->W + 231) / 232) * (long)((rows + oh - 1) / oh);
    return tiles >= 768 && rows >= 8 * (N + 2);
}

// band B of a two-band projection iterates in two private planes (allocated outside any capture)
void ensure_band_planes(fs_ctx *c, int N)
{
    if (c->p_band[0] || !bands_wanted(c, N)) return;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipMalloc((void **)&c->p_band[i], c->plane * sizeof(float));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(c->p_band[0]);
            c->p_band[0] = c->p_band[1] = nullptr;   // the loop then stays one chain
            break;
        }
    }
}

namespace {

struct Planner {
    fs_ctx *c;
    const bool multi;
    const int G;
    hipStream_t lane = nullptr;   // != null: the Jacobi loops planned now run as a side chain on this stream
    explicit Planner(fs_ctx *ctx) : c(ctx), multi(ctx->world > 1), G(ctx->G) {}

    // One GPU, dye diffusion on (Solver:562-574): that loop touches nothing but the dye, so it runs beside the whole
    // velocity chain (third stream, forked at the start of the frame, joined where update() has it) and fills the
    // idle issue slots of the pressure loop and the streaming passes.
    bool dye_diffusion_hoisted() const
    {
        return !multi && c->split_planes && c->params.diffusion > 0.0f && c->params.solverIterations > 0;
    }
    void plumbing(const char *what, std::function<int()> fn)
    {
        Step s;
        s.kind = STEP_COMPUTE;
        s.what = what;
        s.hidden = true;
        s.run = std::move(fn);
        c->steps.push_back(std::move(s));
    }

    int eff(int f) const { return std::min(c->f[f].valid, G); }   // usable depth: valid AND present in the buffer

    void compute(const char *what, const fs_grid &g, int n, std::function<int()> fn)
    {
        Step s;
        s.kind = STEP_COMPUTE;
        s.what = what;
        s.yb = g.yb; s.ye = g.ye; s.n = n;
        s.run = std::move(fn);
        c->steps.push_back(std::move(s));
    }

    // refresh, in ONE batched round, every listed field whose valid depth is short (capped at G);
    // async: the main stream does not wait here, a join(slot) is planned where the rows are needed
    bool ensure(std::initializer_list<std::pair<int, int>> wants, int slot = SLOT_SYNC, bool async = false)
    {
        if (!multi) return false;
        Step s;
        s.kind = STEP_EXCHANGE;
        s.what = async ? "exchange_async" : "exchange";
        s.slot = slot;
        s.async = async;
        for (const auto &w : wants) {
            const int d = std::min(w.second, G);
            if (c->f[w.first].valid < d) s.xf.push_back(Xfer{w.first, c->f[w.first].cur, d});
        }
        if (s.xf.empty()) return false;
        for (const auto &x : s.xf) c->f[x.field].valid = x.depth;
        c->steps.push_back(std::move(s));
        return true;
    }
    void need(int f, int depth)
    {
        if (multi && c->f[f].valid < std::min(depth, G)) ensure({{f, depth}});
    }
    void join(int slot)
    {
        Step s;
        s.kind = STEP_JOIN;
        s.what = slot == SLOT_VEL ? "join vel" : "join dye";
        s.slot = slot;
        c->steps.push_back(std::move(s));
    }
    // rows [yb, ye) clipped to the buffer (possibly empty)
    fs_grid grid_rows(int yb, int ye) const
    {
        yb = std::max(yb, c->b0);
        ye = std::min(ye, c->b1);
        if (ye < yb) ye = yb;
        return fs_grid{c->W, c->H, c->pitch, c->b0, c->b1 - c->b0, yb, ye};
    }
    // ghost depth of velocity the stencil chain of update() consumes, backwards from subtractGradient
    // on the owned rows: p(1) <- N sweeps <- div(N) <- vel(N+1) <- N diffuse sweeps <- vel(2N+1)
    // <- applyVorticity <- vort(2N+2) <- vel(2N+3)
    struct Needs { int p, div, v2, v1, v0; };
    Needs chain_needs() const
    {
        const fs_params &p = c->params;
        const int N = p.solverIterations;
        Needs n;
        n.p = 1;
        n.div = n.p + std::max(N - 1, 0);
        n.v2 = n.div + 1;
        n.v1 = n.v2 + (p.viscosity > 0.0f ? N : 0);
        n.v0 = n.v1 + (p.vorticityStrength > 0.0f ? 2 : 0);
        return n;
    }

    // N Jacobi sweeps on field fi (ping-pong).  One sweep computed on the owned rows +- d reads the
    // previous iterate on +- (d+1), i.e. consumes one row of ghost depth.  The sweeps run in as few
    // temporally blocked chunks as the available depth allows; ghosts are re-exchanged only when the
    // depth is used up.
    void sweeps(bool pressure, int fi, int N, int need_out, bool with_gradient = false)
    {
        Field &F = c->f[fi];
        int remaining = N;
        while (remaining > 0) {
            int n, out_depth;
            if (!multi) {
                n = remaining;
                out_depth = 0;
            } else {
                const int want = need_out + remaining;   // depth that finishes the loop with no further exchange
                if (F.valid < 1) ensure({{fi, want}});
                const int a = std::min(F.valid, G);
                if (a >= want) { n = remaining; out_depth = need_out; }
                else { n = std::min(remaining, a); out_depth = a - n; }
                if (pressure) {
                    // the sweep at depth d reads div at depth d; the deepest sweep of the chunk is out_depth + n - 1
                    Field &D = c->f[F_DIV];
                    if (D.valid < out_depth + n - 1) ensure({{F_DIV, need_out + remaining - 1}});
                    const int vd = std::min(D.valid, G);
                    if (vd < out_depth + n - 1) {
                        n = std::min(n, vd + 1);
                        out_depth = std::min(out_depth, vd - (n - 1));
                    }
                }
            }
            const fs_grid g = grid_of(c, out_depth);
            float *src = F.p(), *dst = F.t();
            hipStream_t st = lane ? lane : c->stream;
            const JacobiTuning tn = c->tuning;
            if (pressure) {
                const int pz = c->p_zero ? 1 : 0;
                c->p_zero = false;
                const float *div = c->f[F_DIV].p();
                float *u = c->f[F_VEL].p(), *v = off(u, c->plane);
                c->p_band_active = false;
                if (with_gradient && c->use_graph && bands_wanted(c, n) && c->p_band[0] && !lane) {
                    float *dw = c->f[F_DIV].p(), *qa = c->p_band[0], *qb = c->p_band[1];
                    hipStream_t st2 = c->stream2;
                    hipEvent_t ef = c->ev_fork, ej = c->ev_join, ea = c->ev_band[0], eb = c->ev_band[1];
                    const int split = (c->H / 2 + 3) & ~3;
                    c->p_band_active = true;
                    c->p_band_row = split;
                    c->p_band_final = jacobi_launch_count(tn, 0, n) & 1;
                    compute("project_n (2 bands)", g, n, [=]() {                      // Solver:522-547
                        int ib = 0;
                        return project_n_bands(tn, &g, split, u, v, dw, src, dst, qa, qb, n, &ib, st, st2, ef, ej, ea, eb, nullptr, 0);
                    });
                }
                else if (with_gradient) {
                    float *dw = c->f[F_DIV].p();
                    compute("project_n", g, n, [=]() {                                // Solver:522-547
                        int ib = 0;
                        return project_n_cfl(tn, &g, u, v, dw, src, dst, n, nullptr, &ib, st, 0);
                    });
                }
                else
                    compute("pressure_n", g, n, [=]() {                               // Solver:533-541
                        int ib = 0;
                        return pressure_n(tn, &g, div, src, dst, n, pz, &ib, st);
                    });
            } else {
                const float coeff = fi == F_VEL ? c->params.viscosity : c->params.diffusion;
                const float dt = c->params.deltaTime;
                const int np = F.np;
                const size_t plane = c->plane;
                // The planes of a diffuse loop do not depend on each other.  As ONE launch per chunk of sweeps, every
                // launch ends in a partly filled last round of workgroups (3072^2: 1664 workgroups on 512 slots) before
                // the next may start; as two chains on two streams (fork / join by event, captured into the frame's
                // graph as parallel branches) the tail of one chain is filled by the other.  Same kernels, same bits.
                const bool split = c->split_planes && np > 1 && c->stream2 && !lane;
                hipStream_t st2 = c->stream2;
                hipEvent_t ef = c->ev_fork, ej = c->ev_join;
                compute("diffuse_n", g, n, [=]() {                                    // Solver:506-517, :563-573
                    int ib = 0;
                    if (!split) return diffuse_n(tn, &g, src, dst, np, plane, coeff, dt, n, &ib, st);
                    FS_HIP(hipEventRecord(ef, st));
                    FS_HIP(hipStreamWaitEvent(st2, ef, 0));
                    const int rc1 = diffuse_n(tn, &g, src, dst, 1, plane, coeff, dt, n, &ib, st);
                    const int rc2 = diffuse_n(tn, &g, src + plane, dst + plane, np - 1, plane, coeff, dt, n, &ib, st2);
                    FS_HIP(hipEventRecord(ej, st2));           // joined even after an error: a capture must end whole
                    FS_HIP(hipStreamWaitEvent(st, ej, 0));
                    return rc1 ? rc1 : rc2;
                });
            }
            if (jacobi_launch_count(c->tuning, pressure ? 0 : 1, n) & 1) F.flip();
            if (multi) F.valid = out_depth;
            remaining -= n;
        }
    }

    // The advected dye is read next by the NEXT frame's dye advection, a frame of other traffic away.  When the planes are of
    // the size at which the just-advected velocity and the dye read here still fit the 256 MB Infinity Cache together
    // (2 + 3 planes) but not with the dye written as well (+ 3), the result leaves through streaming stores and the velocity
    // is still there for the next frame's curl and diffuse launches: 463.5 -> 457.2 us per frame at 3072^2 (37.7 MB planes);
    // at 4096^2 (67 MB) nothing of that fits either way and the hint costs 0.6 % (profiles/r03_z_streaming_stores.log).  Not
    // when the display pass reads the dye right behind.  (From 96 MB on every result streams, common.h st4s.)
    bool dye_result_streams(int fi) const
    {
        if (fi != F_DYE || c->fuse != 1 || c->dry) return false;
        if (c->params.displayMode == FS_DISPLAY_DENSITY) return false;
        const double P = (double)c->pitch * (double)(c->b1 - c->b0) * sizeof(float);
        return 8.0 * P > 256.0 * 1024 * 1024 && 5.0 * P <= 256.0 * 1024 * 1024;
    }
    // advect field fi on rows [yb, ye) (no role swap: the caller swaps once all parts are planned)
    void advect_rows(int fi, bool fade, int yb, int ye)
    {
        Field &F = c->f[fi];
        const fs_grid g = grid_rows(yb, ye);
        const bool vel_done = fi != F_VEL;   // the dye is traced through the advected velocity (Solver:577-582)
        const float *u = vel_done ? c->f[F_VEL].p() : F.p(), *v = off(u, c->plane);
        const float *src = F.p();
        float *dst = F.t();
        const int np = F.np;
        const size_t plane = c->plane;
        const float dt = c->params.deltaTime, fr = c->params.fadeSpeed;
        int *err = c->err_flag;
        hipStream_t st = c->stream;
        if (dye_result_streams(fi))
            compute(fade ? "advect_fade" : "advect", g, np,
                    [=]() { return advect_around(&g, g.yb, g.yb, u, v, src, dst, np, plane, dt, fade ? 1 : 0, fr, err, st, 1); });
        else if (fade)
            compute("advect_fade", g, np, [=]() { return fs_k_advect_fade(&g, u, v, src, dst, np, plane, dt, fr, err, st); });
        else
            compute("advect", g, np, [=]() { return fs_k_advect(&g, u, v, src, dst, np, plane, dt, err, st); });
    }
    // the two rims [ya, yb) and [yc, yd) of a split advection.  Planned as two steps (the plan reads as it always did); when both
    // have rows the first step launches them together (advect_around) and the second has nothing left to do: one launch
    // and one dependency gap less per field for a slab with two neighbours
    void advect_rims(int fi, bool fade, int ya, int yb, int yc, int yd)
    {
        const fs_grid ga = grid_rows(ya, yb), gb = grid_rows(yc, yd);
        if (ga.ye <= ga.yb || gb.ye <= gb.yb || gb.yb < ga.ye) {
            advect_rows(fi, fade, ya, yb);
            advect_rows(fi, fade, yc, yd);
            return;
        }
        Field &F = c->f[fi];
        fs_grid g = ga;
        g.ye = gb.ye;
        const int h0 = ga.ye, h1 = gb.yb;
        const bool vel_done = fi != F_VEL;
        const float *u = vel_done ? c->f[F_VEL].p() : F.p(), *v = off(u, c->plane);
        const float *src = F.p();
        float *dst = F.t();
        const int np = F.np;
        const size_t plane = c->plane;
        const float dt = c->params.deltaTime, fr = c->params.fadeSpeed;
        int *err = c->err_flag;
        hipStream_t st = c->stream;
        const char *what = fade ? "advect_fade" : "advect";
        const int streams = dye_result_streams(fi) ? 1 : 0;
        compute(what, ga, np, [=]() { return advect_around(&g, h0, h1, u, v, src, dst, np, plane, dt, fade ? 1 : 0, fr, err, st, streams); });
        compute(what, gb, np, []() { return (int)FS_OK; });
    }
    // both advections on rows [yb, ye) in one pass (FS_OPT_FUSE 1); no role swaps
    void advect_both_rows(int yb, int ye)
    {
        Field &V = c->f[F_VEL], &D = c->f[F_DYE];
        const fs_grid g = grid_rows(yb, ye);
        const float *u = V.p(), *v = off(u, c->plane), *dye = D.p();
        float *uo = V.t(), *vo = off(uo, c->plane), *dyo = D.t();
        const size_t plane = c->plane;
        const float dt = c->params.deltaTime, fr = c->params.fadeSpeed;
        int *err = c->err_flag;
        hipStream_t st = c->stream;
        compute("advect_both", g, 5, [=]() { return fs_k_advect_both(&g, u, v, uo, vo, dye, dyo, plane, dt, 1, fr, err, st); });
    }
    void advect(int fi, bool fade, int depth = 0)
    {
        advect_rows(fi, fade, c->r0 - depth, c->r1 + depth);
        c->f[fi].flip();                                                            // Solver:558, :584
        if (multi) c->f[fi].valid = depth;
    }

    // update() up to subtractGradient, Solver:486-547.  h_spec >= 0: the advection halo this frame is
    // expected to need; the dye halo of that depth is fetched now, under the whole stencil chain.
    void plan_a(int h_spec)
    {
        const fs_params &p = c->params;
        const int N = p.solverIterations;
        const float dt = p.deltaTime;
        const bool vort_on = p.vorticityStrength > 0.0f, visc_on = p.viscosity > 0.0f;
        hipStream_t st = c->stream;
        const size_t plane = c->plane;
        const Needs nd = chain_needs();
        const int need_p = nd.p, need_div = nd.div, need_v2 = nd.v2, need_v1 = nd.v1, need_v0 = nd.v0;
        Field &V = c->f[F_VEL];
        c->p_band_active = false;

        if (multi && h_spec >= 0) ensure({{F_DYE, h_spec}}, SLOT_DYE, true);
        if (dye_diffusion_hoisted()) {                                                // :562-574, early and aside
            hipStream_t s3 = c->stream3;
            hipEvent_t ef = c->ev_fork3, ej = c->ev_join3;
            plumbing("fork dye", [=]() { FS_HIP(hipEventRecord(ef, st)); FS_HIP(hipStreamWaitEvent(s3, ef, 0)); return (int)FS_OK; });
            lane = s3;
            sweeps(false, F_DYE, N, 0);
            lane = nullptr;
            plumbing("dye chain end", [=]() { FS_HIP(hipEventRecord(ej, s3)); return (int)FS_OK; });
        }
        if (vort_on) {                                                                // :486-502
            need(F_VEL, need_v0);
            const int dw = multi ? std::min(need_v1 + 1, eff(F_VEL) - 1) : 0;
            const int dv = multi ? std::min(std::min(need_v1, dw - 1), eff(F_VEL)) : 0;
            // out-of-cache planes: both passes in one, the vorticity plane never touches memory (it is then not
            // available to read back or display, so not in the vorticity display mode)
            const bool big = (double)c->pitch * (double)(c->H / c->world) * sizeof(float) >= 256.0 * 1024 * 1024;
            // (the marching form of the fused pass wins at every size it applies to: grids at least 512 wide)
            if ((c->fuse == 4 || (c->fuse == 1 && (big || c->W >= 512))) && p.displayMode != FS_DISPLAY_VORTICITY) {
                const fs_grid g = grid_of(c, dv);
                const float *u = V.p(), *v = off(u, plane);
                float *uo = V.t(), *vo = off(uo, plane);
                const float strength = p.vorticityStrength;
                compute("vorticity_apply", g, 0, [=]() { return fs_k_vorticity_apply(&g, u, v, uo, vo, strength, dt, st); });   // :486-502
                V.flip();
                c->vort_stale = true;
            } else {
                c->vort_stale = false;
                {
                    const fs_grid g = grid_of(c, dw);
                    float *u = V.p(), *v = off(u, plane), *w = c->f[F_VORT].p();
                    compute("vorticity", g, 0, [=]() { return fs_k_calculateVorticity(&g, u, v, w, st); });   // :488-491
                }
                {
                    const fs_grid g = grid_of(c, dv);
                    float *u = V.p(), *v = off(u, plane), *w = c->f[F_VORT].p();
                    const float strength = p.vorticityStrength;
                    compute("applyVorticity", g, 0, [=]() { return fs_k_applyVorticity(&g, w, u, v, strength, dt, st); });   // :494-501
                }
            }
            if (multi) { c->f[F_VORT].valid = dw; V.valid = dv; }
        }
        if (visc_on && N > 0) {                                                       // :505-518
            need(F_VEL, need_v1);
            sweeps(false, F_VEL, N, need_v2);
        }
        need(F_VEL, need_v2);
        // The projection in one loop (fs_k_project_n).  A slab needs N + 1 ghost rows of velocity for it (it has them whenever
        // the capacity allows: need_v2 = N + 1).  Rounds 1 and 2 kept the separate divergence and gradient passes beyond the
        // Infinity Cache (+0.3 % for the loop there); with the streaming stores of round 3 the loop wins at every size
        // (-2.9 % per frame at 16384^2, profiles/r03_z_projection_beyond_cache.log)
        const bool project_fused = (c->fuse >= 3 || c->fuse == 1) && N > 0 && (!multi || G >= N + 1);
        const int dd = multi ? std::min(need_div, eff(F_VEL) - 1) : 0;
        if (!project_fused) {
            const fs_grid g = grid_of(c, dd);
            float *u = V.p(), *v = off(u, plane), *d = c->f[F_DIV].p();
            compute("divergence", g, 0, [=]() { return fs_k_divergence(&g, u, v, d, st); });              // :522-525
        }
        if (multi) c->f[F_DIV].valid = dd;
        c->p_zero = true;                                                             // clear folded in, :528-530
        c->f[F_P].valid = kInf;
        // the divergence is computed by the first pressure launch and the gradient is subtracted by the last pressure launch,
        // from registers (FS_OPT_FUSE): -10 us of 597 at 3072^2 (profiles/r01_h_ab_fuse*.log), -2.9 % at 16384^2 once the result
        // planes leave through streaming stores (see project_fused above)
        const bool grad_fused = project_fused;
        if (project_fused && multi) {
            const fs_grid g = grid_of(c, 0);
            Field &P = c->f[F_P];
            float *u = V.p(), *v = off(u, plane), *dw = c->f[F_DIV].p(), *src = P.p(), *dst = P.t(), *cfl = off(c->cfl_dev, c->cfl_slot);
            const JacobiTuning tn = c->tuning;
            // as two row bands out of step, like the single-GPU frame (project_n_bands), when the launches are replayed
            const int split = ((c->r0 + c->r1) / 2 + 3) & ~3;
            if (c->use_graph && bands_wanted(c, N) && c->p_band[0] && split > g.yb && split < g.ye) {
                float *qa = c->p_band[0], *qb = c->p_band[1];
                hipStream_t st2 = c->stream2;
                hipEvent_t ef = c->ev_fork, ej = c->ev_join, ea = c->ev_band[0], eb = c->ev_band[1];
                c->p_band_active = true;
                c->p_band_row = split;
                c->p_band_final = jacobi_launch_count(tn, 0, N) & 1;
                compute("project_n_cfl", g, N, [=]() {                                // :522-547 + the CFL bound
                    int ib = 0;
                    return project_n_bands(tn, &g, split, u, v, dw, src, dst, qa, qb, N, &ib, st, st2, ef, ej, ea, eb, cfl, 1);
                });
            } else
            compute("project_n_cfl", g, N, [=]() {                                    // :522-547 + the CFL bound
                int ib = 0;
                return project_n_cfl(tn, &g, u, v, dw, src, dst, N, cfl, &ib, st, 1);
            });
            if (jacobi_launch_count(c->tuning, 0, N) & 1) P.flip();
            P.valid = 0;
            c->f[F_DIV].valid = 0;
            c->p_zero = false;
        } else if (N > 0) {
            sweeps(true, F_P, N, need_p, grad_fused);                                 // :533-541 (+ :544-547)
        } else {
            const fs_grid g = grid_of(c, G);
            float *pp = c->f[F_P].p();
            compute("clear", g, 0, [=]() { return fs_k_clear(&g, pp, 1, 0, st); });
            c->p_zero = false;
        }
        if (!grad_fused) need(F_P, 1);
        {
            const fs_grid g = grid_of(c, 0);
            float *u = V.p(), *v = off(u, plane), *pp = c->f[F_P].p();
            if (grad_fused) {
                // done
            } else if (!multi) {
                compute("subtractGradient", g, 0, [=]() { return fs_k_subtractGradient(&g, pp, u, v, st); }); // :544-547
            } else {
                // ... and, in the same pass, the CFL bound of the backtrace: max |v_y| over the owned rows
                float *cfl = off(c->cfl_dev, c->cfl_slot);
                compute("subtractGradient_cfl", g, 0, [=]() {
                    hipError_t e = hipMemsetAsync(cfl, 0, sizeof(float), st);
                    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(cfl)");
                    return fs_k_subtractGradient_cfl(&g, pp, u, v, cfl, st);
                });
            }
        }
        if (multi) {
            V.valid = 0;
            Step s;                                                                   // ... then max over ranks
            s.kind = STEP_CFL_REDUCE;
            s.what = "cfl_allreduce";
            c->steps.push_back(std::move(s));
        }
    }

    // advection .. tracers, Solver:550-623, with advection halo h (0 on one GPU)
    void plan_b(int h)
    {
        const fs_params &p = c->params;
        const int N = p.solverIterations;
        const int ddn = (p.diffusion > 0.0f && N > 0) ? N : 0;
        hipStream_t st = c->stream;
        const size_t plane = c->plane;
        const bool fade_fused = c->fuse == 1 || c->fuse == 2 || c->fuse == 4;
        // both advections in one pass: no gain while a plane fits the Infinity Cache (572 vs 570 us per frame at
        // 3072^2: the advected velocity is read back from the cache anyway), -0.5 % at 16384^2
        // (profiles/r01_q_ab_fuse_*.log); so only when a plane of the slab is at least as big as the cache
        const bool both_fused = c->fuse == 4 || (c->fuse == 1 && (double)c->pitch * (double)(c->H / c->world) * sizeof(float) >= 256.0 * 1024 * 1024);
        // overlap: the advected velocity is also computed on the ghost rows the next frame's stencil
        // chain consumes (as far as the ghost capacity allows), from a velocity halo of h + adv rows
        const int adv = (multi && c->overlap) ? std::max(0, std::min(chain_needs().v0, G - h)) : 0;
        const bool split = multi && c->overlap && ddn == 0 && c->H / c->world > 2 * h + 8;
        if (split) {
            ensure({{F_VEL, h + adv}, {F_DYE, h}}, SLOT_VEL, true);                   // dye: normally prefetched by plan_a
            const int i0 = c->r0 + (c->r0 > 0 ? h : 0), i1 = c->r1 - (c->r1 < c->H ? h : 0);
            // interior rows: no tap of their backtrace leaves the owned rows            :550-558, :577-591
            if (both_fused) {
                advect_both_rows(i0, i1);
            } else {
                advect_rows(F_VEL, false, i0, i1);
                c->f[F_VEL].flip();
                advect_rows(F_DYE, fade_fused, i0, i1);
                c->f[F_VEL].flip();
            }
            join(SLOT_VEL);
            advect_rims(F_VEL, false, c->r0 - adv, i0, i1, c->r1 + adv);
            c->f[F_VEL].flip();                                                     // :558
            c->f[F_VEL].valid = adv;
            join(SLOT_DYE);
            advect_rims(F_DYE, fade_fused, c->r0, i0, i1, c->r1);
            c->f[F_DYE].flip();                                                     // :584
        } else {
            if (multi) {
                ensure({{F_VEL, h + adv}, {F_DYE, h + ddn}});                         // one batched round
                join(SLOT_DYE);                                                       // a prefetch of plan_a, if any
            }
            if (!multi && !ddn && both_fused) {
                advect_both_rows(c->r0, c->r1);                                       // :550-558 + :577-591 in one pass
                c->f[F_VEL].flip();
                c->f[F_DYE].flip();
            } else {
                advect(F_VEL, false, adv);                                            // :550-558
                if (ddn && dye_diffusion_hoisted()) {                                 // planned by plan_a: wait for it here
                    hipEvent_t ej = c->ev_join3;
                    plumbing("join dye", [=]() { FS_HIP(hipStreamWaitEvent(st, ej, 0)); return (int)FS_OK; });
                } else if (ddn) {                                                     // :562-574
                    sweeps(false, F_DYE, N, h);
                    need(F_DYE, h);
                }
                advect(F_DYE, fade_fused);                                            // :577-584 (+ :587-591 when fused)
            }
        }
        if (!fade_fused) {
            const fs_grid g = grid_of(c, 0);
            float *d = c->f[F_DYE].p();
            const float fr = p.fadeSpeed;
            compute("fade", g, 0, [=]() { return fs_k_fadeDensity(&g, d, 3, plane, fr, st); });           // :587-591
        }
        if (multi) c->f[F_DYE].valid = 0;
        // render, :594-623
        {
            const fs_grid g = grid_of(c, 0);
            float *u = c->f[F_VEL].p(), *v = off(u, plane), *d = c->f[F_DYE].p(), *w = c->f[F_VORT].p(), *disp = c->display;
            const float br = p.brightness;
            switch (p.displayMode) {
            case FS_DISPLAY_DENSITY:
                compute("render", g, 0, [=]() { return fs_k_renderDensity(&g, d, plane, 3, disp, plane, br, st); });
                break;
            case FS_DISPLAY_VELOCITY:
                compute("render", g, 0, [=]() { return fs_k_renderVelocity(&g, u, v, disp, plane, 1.0f, st); });
                break;
            case FS_DISPLAY_SPEED:
                compute("render", g, 0, [=]() { return fs_k_renderSpeed(&g, u, v, disp, plane, 25.0f, st); });
                break;
            case FS_DISPLAY_VORTICITY:
                compute("render", g, 0, [=]() { return fs_k_renderDensity(&g, w, plane, 1, disp, plane, 10.0f, st); });
                break;
            case FS_DISPLAY_PARTICLES_ONLY:
                compute("render", g, 0, [=]() { return fs_k_clear(&g, disp, 4, plane, st); });
                break;
            default: break;   // FS_DISPLAY_NONE: headless
            }
        }
        if (c->with_particles && c->n_particles > 0 && !multi) {                      // MetalView.swift:155 -> Part:242
            const fs_grid g = grid_of(c, 0);
            float *u = c->f[F_VEL].p(), *v = off(u, plane);
            const fs_particle_params pp = c->pparams;
            // (slots rounded up to 16384, the empty ones hold life 0: a few tracers spawned do not change the node, context.hip)
            const long n = std::min(c->max_particles, (c->n_particles + 16383L) & ~16383L);
            float *a0 = c->part[0], *a1 = c->part[1], *a2 = c->part[2], *a3 = c->part[3], *a4 = c->part[4],
                  *a5 = c->part[5], *a6 = c->part[6];
            compute("particles", g, 0, [=]() { return fs_k_particles_update(&g, u, v, &pp, n, a0, a1, a2, a3, a4, a5, a6, st); });
        }
        if (c->with_particles && multi) {                                             // collective: every rank plans it
            Step s;
            s.kind = STEP_TRACERS;
            s.what = "tracers";
            c->steps.push_back(std::move(s));
        }
    }
};

// ---- RCCL, resolved at run time so that single-GPU hosts need no librccl ------------------------
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return FS_OK;
    void *h = nullptr;
    // FS_RCCL_LIBRARY wins; else a copy already in the process (e.g. the one PyTorch ships), so that one
    // RCCL serves the process; else the default search path
    const char *env = getenv("FS_RCCL_LIBRARY");
    if (env && *env) {
        h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        if (!h) { set_error("cannot load FS_RCCL_LIBRARY=%s (%s)", env, dlerror()); return FS_ERR_STATE; }
    }
    const char *loaded[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : loaded)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char *paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : paths)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_error("cannot load librccl (%s); set FS_RCCL_LIBRARY", dlerror()); return FS_ERR_STATE; }
#define SYM(field, name)                                                                  \
    do {                                                                                  \
        *(void **)(&g_rccl.field) = dlsym(h, name);                                       \
        if (!g_rccl.field) { set_error("librccl lacks %s", name); return FS_ERR_STATE; }  \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.handle = h;
    return FS_OK;
}

int rccl_fail(ncclResult_t r, const char *what)
{
    set_error("RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
    return FS_ERR_HIP;
}
#define FS_NCCL(call)                                       \
    do {                                                    \
        ncclResult_t r_ = (call);                           \
        if (r_ != ncclSuccess) return rccl_fail(r_, #call); \
    } while (0)

// One contiguous block of rows of one plane that moves between two ranks.
struct Seg {
    int peer, field, bufidx, plane, y0, y1;
    bool send;
};

// Rows a rank with owned rows [q0,q1) needs for ghost depth d: [max(0,q0-d), q0) and [q1, min(H,q1+d)).
// Every other rank that owns part of them sends that part (single hop when d <= its slab height,
// multi hop otherwise).  Sends to a peer are generated in the order the peer generates its receives.
void build_segments(const fs_ctx *c, const std::vector<Xfer> &xf, std::vector<Seg> &out)
{
    for (const Xfer &x : xf) {
        const int d = x.depth, np = c->f[x.field].np;
        for (int q = 0; q < c->world; ++q) {
            if (q == c->rank) continue;
            int q0, q1;
            slab_rows(c->H, c->world, q, &q0, &q1);
            const int mine[2][2] = {{std::max(0, c->r0 - d), c->r0}, {c->r1, std::min(c->H, c->r1 + d)}};
            const int theirs[2][2] = {{std::max(0, q0 - d), q0}, {q1, std::min(c->H, q1 + d)}};
            for (int k = 0; k < 2; ++k) {   // what I receive from q
                const int a = std::max(mine[k][0], q0), b = std::min(mine[k][1], q1);
                for (int pl = 0; a < b && pl < np; ++pl) out.push_back(Seg{q, x.field, x.bufidx, pl, a, b, false});
            }
            for (int k = 0; k < 2; ++k) {   // what I send to q
                const int a = std::max(theirs[k][0], c->r0), b = std::min(theirs[k][1], c->r1);
                for (int pl = 0; a < b && pl < np; ++pl) out.push_back(Seg{q, x.field, x.bufidx, pl, a, b, true});
            }
        }
    }
}

float *seg_ptr(const fs_ctx *c, const Seg &s)
{
    return c->f[s.field].buf[s.bufidx] + (size_t)s.plane * c->plane + (size_t)(s.y0 - c->b0) * c->pitch;
}

int sync_members(const fs_comm *cm)
{
    for (fs_ctx *m : cm->members) {
        if (!m) { set_error("local group: a rank has no context yet"); return FS_ERR_STATE; }
        FS_HIP(hipSetDevice(m->device));
        FS_HIP(hipStreamSynchronize(m->stream));
    }
    return FS_OK;
}

}  // namespace

void plan_update_a(fs_ctx *c, int h_spec) { Planner(c).plan_a(h_spec); }
void plan_update_b(fs_ctx *c, int h) { Planner(c).plan_b(h); }

namespace {

// "everything enqueued on the main stream so far": what a transfer or the CFL reduction on the exchange stream waits for.
// Two of those back to back (the reduction and the velocity exchange after the projection) share one marker.
int mark_main(fs_ctx *c)
{
    if (c->ev_main_fresh) return FS_OK;
    FS_HIP(hipEventRecord(c->ev_main, c->stream));
    c->ev_main_fresh = true;
    return FS_OK;
}

int join_slot(fs_ctx *c, int slot)
{
    if (!c->pending[slot]) return FS_OK;
    fs_ctx::WaitProbe pr{nullptr, nullptr, slot};
    const bool probe = c->profile && c->probes_live.size() < 8192;
    if (probe) {
        if (!c->probes_free.empty()) { pr = c->probes_free.back(); c->probes_free.pop_back(); pr.slot = slot; }
        else { FS_HIP(hipEventCreate(&pr.a)); FS_HIP(hipEventCreate(&pr.b)); }
        FS_HIP(hipEventRecord(pr.a, c->stream));
    }
    FS_HIP(hipStreamWaitEvent(c->stream, c->ev_slot[slot], 0));
    if (probe) {
        FS_HIP(hipEventRecord(pr.b, c->stream));
        c->probes_live.push_back(pr);
    }
    c->pending[slot] = false;
    return FS_OK;
}

// One halo exchange, in three phases so that an in-process group can walk them in lock step (phase p
// of every rank before phase p+1 of any); a rank of an RCCL run just runs them back to back.
//   0  main stream: "my rows are final" (ev_main)
//   1  exchange stream: wait for that (and, kind 2, for the peers' rows), move the rows
//   2  kind 2: my sends are complete once the peers have copied; the main stream joins unless async
int exchange_phase(fs_ctx *c, const Step &st, int phase)
{
    fs_comm *cm = c->comm;
    if (!cm) { set_error("exchange planned on a context without a communicator"); return FS_ERR_STATE; }
    std::vector<Seg> segs;
    build_segments(c, st.xf, segs);
    if (phase == 0) {
        c->stat_exchanges += 1;
        for (const Seg &s : segs)
            if (s.send) c->stat_rows += s.y1 - s.y0;
        if (cm->kind != 0) return mark_main(c);
        return FS_OK;
    }
    if (phase == 1) {
        if (cm->kind == 1) {
            FS_HIP(hipStreamWaitEvent(c->xstream, c->ev_main, 0));
            FS_NCCL(g_rccl.GroupStart());
            for (const Seg &s : segs) {
                const size_t count = (size_t)(s.y1 - s.y0) * c->pitch;
                ncclResult_t r = s.send ? g_rccl.Send(seg_ptr(c, s), count, ncclFloat, s.peer, (ncclComm_t)cm->nccl, c->xstream)
                                        : g_rccl.Recv(seg_ptr(c, s), count, ncclFloat, s.peer, (ncclComm_t)cm->nccl, c->xstream);
                if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail(r, s.send ? "ncclSend" : "ncclRecv"); }
            }
            FS_NCCL(g_rccl.GroupEnd());
            FS_HIP(hipEventRecord(c->ev_slot[st.slot], c->xstream));
            return FS_OK;
        }
        if (cm->kind == 3) {
            // loopback: rows of the same count out of my own slab (timing only, see fs_comm)
            FS_HIP(hipStreamWaitEvent(c->xstream, c->ev_main, 0));
            const int own = c->r1 - c->r0;
            for (const Seg &s : segs) {
                if (s.send) continue;
                for (int a = s.y0; a < s.y1; a += own) {
                    const int cnt = std::min(own, s.y1 - a);
                    Seg dst = s, src = s;
                    dst.y0 = a;
                    src.y0 = a < c->r0 ? c->r0 : c->r1 - cnt;   // rows above me <- my first rows, rows below me <- my last rows
                    FS_HIP(hipMemcpyAsync(seg_ptr(c, dst), seg_ptr(c, src), (size_t)cnt * c->pitch * sizeof(float), hipMemcpyDeviceToDevice,
                                          c->xstream));
                }
            }
            FS_HIP(hipEventRecord(c->ev_slot[st.slot], c->xstream));
            return FS_OK;
        }
        if (cm->kind == 2) {
            FS_HIP(hipStreamWaitEvent(c->xstream, c->ev_main, 0));
            int last = -1;
            for (const Seg &s : segs) {
                if (s.send) continue;
                const fs_ctx *peer = cm->members[s.peer];
                if (s.peer != last) FS_HIP(hipStreamWaitEvent(c->xstream, peer->ev_main, 0));
                last = s.peer;
                FS_HIP(hipMemcpyAsync(seg_ptr(c, s), seg_ptr(peer, s), (size_t)(s.y1 - s.y0) * c->pitch * sizeof(float),
                                      hipMemcpyDeviceToDevice, c->xstream));
            }
            FS_HIP(hipEventRecord(c->ev_copied[st.slot], c->xstream));
            return FS_OK;
        }
        // kind 0: every rank's owned rows are final when its turn comes (lock step), ghost rows are
        // only written here; plain blocking copies between full synchronisations
        int rc = sync_members(cm);
        if (rc) return rc;
        FS_HIP(hipSetDevice(c->device));
        for (const Seg &s : segs) {
            if (s.send) continue;
            const fs_ctx *peer = cm->members[s.peer];
            FS_HIP(hipMemcpy(seg_ptr(c, s), seg_ptr(peer, s), (size_t)(s.y1 - s.y0) * c->pitch * sizeof(float),
                             hipMemcpyDeviceToDevice));
        }
        FS_HIP(hipDeviceSynchronize());
        return FS_OK;
    }
    if (cm->kind == 0) return FS_OK;
    if (cm->kind == 2) {
        int last = -1;
        for (const Seg &s : segs) {
            if (!s.send || s.peer == last) continue;
            last = s.peer;
            FS_HIP(hipStreamWaitEvent(c->xstream, cm->members[s.peer]->ev_copied[st.slot], 0));
        }
        FS_HIP(hipStreamWaitEvent(c->xstream, c->ev_copied[st.slot], 0));
        FS_HIP(hipEventRecord(c->ev_slot[st.slot], c->xstream));
    }
    c->pending[st.slot] = true;
    return st.async ? FS_OK : join_slot(c, st.slot);
}

// max over ranks of the per-slab CFL bound of the frame being enqueued (slot c->cfl_slot); the value lands in pinned host
// memory behind ev_cfl[slot]
int cfl_phase(fs_ctx *c, int phase)
{
    fs_comm *cm = c->comm;
    if (!cm) { set_error("CFL reduction planned on a context without a communicator"); return FS_ERR_STATE; }
    if (phase == 0) {
        if (cm->kind != 0) return mark_main(c);
        return FS_OK;
    }
    if (phase != 1) return FS_OK;
    const int slot = c->cfl_slot;
    float *dev = c->cfl_dev + slot, *host = c->cfl_host + (size_t)slot * c->world;
    if (cm->kind == 1) {
        FS_HIP(hipStreamWaitEvent(c->xstream, c->ev_main, 0));
        FS_NCCL(g_rccl.AllReduce(dev, dev, 1, ncclFloat, ncclMax, (ncclComm_t)cm->nccl, c->xstream));
        FS_HIP(hipMemcpyAsync(host, dev, sizeof(float), hipMemcpyDeviceToHost, c->xstream));
        FS_HIP(hipEventRecord(c->ev_cfl[slot], c->xstream));
        return FS_OK;
    }
    if (cm->kind == 3) {   // loopback: my own bound
        FS_HIP(hipStreamWaitEvent(c->xstream, c->ev_main, 0));
        FS_HIP(hipMemcpyAsync(host, dev, sizeof(float), hipMemcpyDeviceToHost, c->xstream));
        FS_HIP(hipEventRecord(c->ev_cfl[slot], c->xstream));
        return FS_OK;
    }
    if (cm->kind == 0) {
        int rc = sync_members(cm);
        if (rc) return rc;
        FS_HIP(hipSetDevice(c->device));
    }
    for (fs_ctx *x : cm->members) {
        if (cm->kind == 2) {
            FS_HIP(hipStreamWaitEvent(c->xstream, x->ev_main, 0));
            FS_HIP(hipMemcpyAsync(&host[x->rank], x->cfl_dev + slot, sizeof(float), hipMemcpyDeviceToHost, c->xstream));
        } else {
            FS_HIP(hipMemcpy(&host[x->rank], x->cfl_dev + slot, sizeof(float), hipMemcpyDeviceToHost));
        }
    }
    if (cm->kind == 2) FS_HIP(hipEventRecord(c->ev_cfl[slot], c->xstream));
    return FS_OK;
}

int wait_cfl(fs_ctx *c, int slot)
{
    fs_comm *cm = c->comm;
    if (cm->kind != 0) {
        const auto t0 = std::chrono::steady_clock::now();
        FS_HIP(hipEventSynchronize(c->ev_cfl[slot]));
        c->host_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    const int k = cm->solo() ? 1 : cm->world;
    const float *host = c->cfl_host + (size_t)slot * c->world;
    float m = 0.0f;
    for (int i = 0; i < k; ++i)
        if (!(host[i] <= m)) m = host[i];   // NaN propagates
    c->cfl_max = m;
    return FS_OK;
}

// Tracers of a slab: re-home (count per destination -> counts of everybody -> scatter into keep set and
// outboxes -> move the outboxes), then ParticleSystem.update() on the tracers now living here.  Host-synchronous
// (three small waits per frame): a tracer step is cheap next to a frame and no config times it on slabs.
//   0  count my tracers per home rank; RCCL: sum the world x world count matrix over the ranks
//   1  everybody's counts are known: capacity check, outbox offsets, scatter
//   2  move the outboxes (RCCL send/recv, or copies out of the peers' outboxes), swap, update
int tracer_phase(fs_ctx *c, int phase)
{
    fs_comm *cm = c->comm;
    if (!cm) { set_error("tracer step planned on a context without a communicator"); return FS_ERR_STATE; }
    if (cm->kind == 3) { set_error("a loopback slab (measurement aid) has no tracer step"); return FS_ERR_STATE; }
    if (c->max_particles <= 0 || !c->tr_counts_dev) { set_error("tracers on a slab need fs_particles_reserve first"); return FS_ERR_STATE; }
    const int W = c->world, me = c->rank;
    const fs_grid g = grid_of(c, 0);
    const long n = c->n_particles;
    c->ev_main_fresh = false;
    if (phase == 0) {
        FS_HIP(hipMemsetAsync(c->tr_counts_dev, 0, (size_t)W * sizeof(int), c->stream));
        int rc = fs_k_tracer_count(&g, &c->pparams, W, me, n, c->part[1], c->part[4], c->tr_counts_dev, c->stream);
        if (rc) return rc;
        FS_HIP(hipMemcpyAsync(c->tr_counts_host, c->tr_counts_dev, (size_t)W * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        FS_HIP(hipStreamSynchronize(c->stream));
        if (cm->kind == 1) {
            for (int i = 0; i < W * W; ++i) c->tr_matrix_host[i] = 0;
            for (int d = 0; d < W; ++d) c->tr_matrix_host[me * W + d] = c->tr_counts_host[d];
            FS_HIP(hipMemcpyAsync(c->tr_matrix_dev, c->tr_matrix_host, (size_t)W * W * sizeof(int), hipMemcpyHostToDevice, c->stream));
            // integer counts: as floats they would round above 2^24 per pair and the grouped send / recv sizes would stop matching
            FS_NCCL(g_rccl.AllReduce(c->tr_matrix_dev, c->tr_matrix_dev, (size_t)W * W, ncclInt32, ncclSum, (ncclComm_t)cm->nccl, c->stream));
            FS_HIP(hipMemcpyAsync(c->tr_matrix_host, c->tr_matrix_dev, (size_t)W * W * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            FS_HIP(hipStreamSynchronize(c->stream));
        }
        return FS_OK;
    }
    if (phase == 1) {
        c->tr_recv_cnt.assign(W, 0);
        c->tr_send_off.assign(W, 0);
        long total = c->tr_counts_host[me], off = 0;
        for (int q = 0; q < W; ++q) {
            if (q == me) continue;
            c->tr_recv_cnt[q] = cm->kind == 1 ? (long)c->tr_matrix_host[q * W + me] : (long)cm->members[q]->tr_counts_host[me];
            total += c->tr_recv_cnt[q];
            c->tr_send_off[q] = off;
            off += c->tr_counts_host[q];
        }
        if (total > c->max_particles) {
            set_error("rank %d would hold %ld tracers, its capacity is %ld (fs_particles_reserve)", me, total, c->max_particles);
            return FS_ERR_NOMEM;
        }
        std::vector<int> cursors(W);
        for (int q = 0; q < W; ++q) cursors[q] = q == me ? 0 : (int)c->tr_send_off[q];
        FS_HIP(hipMemcpy(c->tr_counts_dev, cursors.data(), (size_t)W * sizeof(int), hipMemcpyHostToDevice));
        return fs_k_tracer_scatter(&g, &c->pparams, W, me, n, c->part, c->part_t, c->outbox, c->tr_counts_dev, c->stream);
    }
    const long keep = c->tr_counts_host[me];
    long at = keep, moved = 0;
    if (cm->kind == 1) {
        FS_NCCL(g_rccl.GroupStart());
        for (int q = 0; q < W; ++q) {
            if (q == me) continue;
            for (int f = 0; f < 7; ++f) {
                ncclResult_t r = ncclSuccess;
                if (c->tr_counts_host[q] > 0)
                    r = g_rccl.Send(c->outbox[f] + c->tr_send_off[q], (size_t)c->tr_counts_host[q], ncclFloat, q, (ncclComm_t)cm->nccl, c->stream);
                if (r == ncclSuccess && c->tr_recv_cnt[q] > 0)
                    r = g_rccl.Recv(c->part_t[f] + at, (size_t)c->tr_recv_cnt[q], ncclFloat, q, (ncclComm_t)cm->nccl, c->stream);
                if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail(r, "tracer send/recv"); }
            }
            at += c->tr_recv_cnt[q];
            moved += c->tr_recv_cnt[q];
        }
        FS_NCCL(g_rccl.GroupEnd());
    } else {
        int rc = sync_members(cm);   // every rank's scatter is complete
        if (rc) return rc;
        FS_HIP(hipSetDevice(c->device));
        for (int q = 0; q < W; ++q) {
            if (q == me || c->tr_recv_cnt[q] == 0) continue;
            const fs_ctx *peer = cm->members[q];
            for (int f = 0; f < 7; ++f)
                FS_HIP(hipMemcpy(c->part_t[f] + at, peer->outbox[f] + peer->tr_send_off[me], (size_t)c->tr_recv_cnt[q] * sizeof(float),
                                 hipMemcpyDeviceToDevice));
            at += c->tr_recv_cnt[q];
            moved += c->tr_recv_cnt[q];
        }
        FS_HIP(hipDeviceSynchronize());   // device-to-device copies may return before they are done
    }
    for (int f = 0; f < 7; ++f) std::swap(c->part[f], c->part_t[f]);
    c->n_particles = at;
    c->stat_migrated += moved;
    const float *u = c->f[F_VEL].p(), *v = u + c->plane;
    return fs_k_particles_update(&g, u, v, &c->pparams, c->n_particles, c->part[0], c->part[1], c->part[2], c->part[3],
                                 c->part[4], c->part[5], c->part[6], c->stream);   // MetalView.swift:155 -> Part:242
}

int exec_phase(fs_ctx *c, const Step &s, int phase)
{
    switch (s.kind) {
    case STEP_COMPUTE: return phase == 0 ? s.run() : FS_OK;
    case STEP_EXCHANGE: return exchange_phase(c, s, phase);
    case STEP_CFL_REDUCE: return cfl_phase(c, phase);
    case STEP_JOIN: return phase == 0 ? join_slot(c, s.slot) : FS_OK;
    case STEP_TRACERS: return tracer_phase(c, phase);
    }
    return FS_ERR_STATE;
}

// bilinear taps lie within ceil(|v_y| dt) + 1 rows; +1 more because the dye is traced through the
// ADVECTED velocity, whose fp32 mix() can exceed the old maximum by rounding only
int halo_from_cfl(fs_ctx *c, int *h)
{
    const double reach = (double)c->cfl_max * std::fabs((double)c->params.deltaTime);
    if (!(reach < 1.0e9)) {
        set_error("velocity is not finite (max |v_y| = %g): cannot size the advection halo", (double)c->cfl_max);
        return FS_ERR_HALO;
    }
    *h = (int)std::ceil(reach) + 2;
    c->last_halo = *h;
    if (*h > c->G) {
        set_error("advection halo of %d rows (max |v_y| dt = %.1f) exceeds the ghost capacity of %d rows; create the "
                  "slabs with more ghost rows", *h, reach, c->G);
        return FS_ERR_HALO;
    }
    return FS_OK;
}

struct HostTimer {
    double &acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit HostTimer(double &a) : acc(a) {}
    ~HostTimer() { acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};

// ---- segments: the exchange-free runs of kernel launches of a slab's plan, replayed as hipGraphs ----------------
// What decides the launches of a segment: the ping-pong roles the plan started from and which plan it is (plan_salt),
// where the segment sits in the plan, every step's stage / row range / sweep count, and the scalars baked into the
// kernel arguments.  Options and tuning are not hashed: changing one drops the graphs (fs_set_option).
unsigned long long segment_key(const fs_ctx *c, size_t i, size_t j)
{
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t k = 0; k < n; ++k) { h ^= b[k]; h *= 1099511628211ull; }
    };
    mix(&c->plan_salt, sizeof(c->plan_salt));
    mix(&i, sizeof(i));
    // (field by field: the struct's padding bytes are not ours to hash)
    const fs_params &p = c->params;
    mix(&p.viscosity, sizeof(float)); mix(&p.diffusion, sizeof(float)); mix(&p.deltaTime, sizeof(float));
    mix(&p.vorticityStrength, sizeof(float)); mix(&p.fadeSpeed, sizeof(float)); mix(&p.brightness, sizeof(float));
    mix(&p.solverIterations, sizeof(int)); mix(&p.displayMode, sizeof(int));
    for (size_t k = i; k < j; ++k) {
        const Step &s = c->steps[k];
        mix(s.what, strlen(s.what) + 1);
        const int v[3] = {s.yb, s.ye, s.n};
        mix(v, sizeof(v));
    }
    return h;
}

int run_steps_eager(fs_ctx *c, size_t i, size_t j)
{
    for (size_t k = i; k < j; ++k) {
        const int rc = c->steps[k].run();
        if (rc) return rc;
    }
    return FS_OK;
}

// steps [i, j) of c's plan are all kernel launches on the context's own streams.  Runs of one or two steps (the split
// advection: interior, rims) are launched as they are: a replay would save nothing, and their row ranges follow the
// advection halo, which changes from frame to frame.
int run_segment(fs_ctx *c, size_t i, size_t j)
{
    c->ev_main_fresh = false;
    const bool graph = c->world > 1 && !c->dry && c->use_graph && c->stable_frames >= c->seg_min_stable && j - i >= 3;
    if (!graph) {
        HostTimer t(c->host_part_us[3]);
        if (c->world > 1) c->stat_seg_eager += 1;
        return run_steps_eager(c, i, j);
    }
    const unsigned long long key = segment_key(c, i, j);
    auto it = c->seg_graphs.find(key);
    if (it != c->seg_graphs.end()) {
        HostTimer t(c->host_part_us[1]);
        FS_HIP(hipGraphLaunch(it->second, c->stream));
        c->stat_seg_replays += 1;
        return FS_OK;
    }
    HostTimer t(c->host_part_us[2]);
    if (c->seg_graphs.size() >= 96) drop_segment_graphs(c);   // (halo depths that keep changing: start over rather than grow)
    hipGraph_t g = nullptr;
    FS_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    const int rc = run_steps_eager(c, i, j);
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (rc || e != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        if (rc) return rc;
        return hip_fail(e, "hipStreamEndCapture(segment)");
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) {   // nothing has run yet: this runtime cannot replay the segment, launch it eagerly from now on
        (void)hipGetLastError();
        c->use_graph = 0;
        c->stat_seg_eager += 1;
        return run_steps_eager(c, i, j);
    }
    c->seg_graphs[key] = exec;
    c->stat_seg_captures += 1;
    FS_HIP(hipGraphLaunch(exec, c->stream));
    return FS_OK;
}

// the planned steps of all n slabs, in lock step (n = 1: a plain walk)
int exec_all(fs_ctx **cs, int n)
{
    const size_t count = cs[0]->steps.size();
    for (int r = 1; r < n; ++r)
        if (cs[r]->steps.size() != count) {
            set_error("rank %d planned %zu steps, rank 0 %zu", r, cs[r]->steps.size(), count);
            return ctx_fail(cs[r], FS_ERR_STATE);
        }
    for (size_t i = 0; i < count;) {
        const int kind = cs[0]->steps[i].kind;
        if (kind == STEP_COMPUTE) {
            // a run of kernel launches: nothing in it involves another rank (every rank plans the same kinds)
            size_t j = i;
            while (j < count && cs[0]->steps[j].kind == STEP_COMPUTE) ++j;
            for (int r = 0; r < n; ++r) {
                fs_ctx *c = cs[r];
                if (n > 1) CTX(c);
                for (size_t k = i; k < j; ++k)
                    if (c->steps[k].kind != STEP_COMPUTE) { set_error("rank %d: step %zu is not a launch", r, k); return ctx_fail(c, FS_ERR_STATE); }
                fs_comm *cm = c->comm;
                const bool serial = n > 1 && cm && cm->serialize && c->ev_serial;
                if (serial && cm->last_serial) TRYHIP(c, hipStreamWaitEvent(c->stream, cm->last_serial, 0));
                TRY(c, run_segment(c, i, j));
                if (serial) { TRYHIP(c, hipEventRecord(c->ev_serial, c->stream)); cm->last_serial = c->ev_serial; }
            }
            i = j;
            continue;
        }
        const int phases = kind == STEP_EXCHANGE || kind == STEP_CFL_REDUCE || kind == STEP_TRACERS ? 3 : 1;
        for (int phase = 0; phase < phases; ++phase)
            for (int r = 0; r < n; ++r) {
                fs_ctx *c = cs[r];
                if (n > 1) CTX(c);
                HostTimer t(c->host_part_us[kind == STEP_EXCHANGE ? 4 : kind == STEP_CFL_REDUCE ? 5 : kind == STEP_JOIN ? 6 : 7]);
                TRY(c, exec_phase(c, c->steps[i], phase));
            }
        ++i;
    }
    return FS_OK;
}

}  // namespace

// One update() on n slabs (n = 1: this rank of an RCCL run, or a single GPU -- possibly under stream
// capture, where nothing but kernel launches happens).
static int run_frames_impl(fs_ctx **cs, int n);
int run_frames(fs_ctx **cs, int n)
{
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = run_frames_impl(cs, n);
    cs[0]->host_frame_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    cs[0]->host_frames += 1;
    return rc;
}
static unsigned long long roles_of(const fs_ctx *c)
{
    return (unsigned long long)c->f[F_VEL].cur | ((unsigned long long)c->f[F_VEL].oth << 2) | ((unsigned long long)c->f[F_DYE].cur << 4) |
           ((unsigned long long)c->f[F_P].cur << 5) | ((unsigned long long)c->cfl_slot << 6);
}
// plan B of one frame on every slab: advection .. render.  hold_tracers: without the tracer step, which is collective
// and host-synchronous and must run once, after the advection is known to be right (tracer_part)
static int advect_part(fs_ctx **cs, int n, int h, bool hold_tracers)
{
    for (int r = 0; r < n; ++r) {
        fs_ctx *c = cs[r];
        HostTimer t(c->host_part_us[0]);
        c->steps.clear();
        c->plan_salt = 0x200ull | roles_of(c);
        plan_update_b(c, h);
        if (hold_tracers && !c->steps.empty() && c->steps.back().kind == STEP_TRACERS) c->steps.pop_back();
    }
    return exec_all(cs, n);
}
static int tracer_part(fs_ctx **cs, int n)
{
    for (int r = 0; r < n; ++r) {
        Step s;
        s.kind = STEP_TRACERS;
        s.what = "tracers";
        cs[r]->steps.clear();
        cs[r]->steps.push_back(std::move(s));
    }
    return exec_all(cs, n);
}
static void push_halo(fs_ctx *c, int h_true)
{
    int *hist = c->halo_hist;
    hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = h_true;
}

// The check a speculating frame owes: was the guessed advection halo deep enough?  (All ranks hold the same
// all-reduced bound and the same guess, so all take the same decision.)  Returns the frame's true halo; *miss: the
// guess was too small, some backtraces of that frame's advection read stale ghost rows.
static int check_pending(fs_ctx **cs, int n, int *h_true, bool *miss)
{
    for (int r = 0; r < n; ++r) cs[r]->chk_pending = false;
    for (int r = 0; r < n; ++r) {
        if (n > 1) CTX(cs[r]);
        TRY(cs[r], wait_cfl(cs[r], cs[r]->chk_slot));
        TRY(cs[r], halo_from_cfl(cs[r], h_true));
    }
    *miss = *h_true > cs[0]->chk_h;
    return FS_OK;
}
// Advection wrote only the ping-pong partners and its inputs are intact (the velocity it read is in the spare role or
// still the partner: nothing has been allowed to overwrite it): plan it again with the true depth from the roles it
// started from, then repeat the splats that have been applied to its output since.
static int repair_advection(fs_ctx **cs, int n, int h_true)
{
    for (int r = 0; r < n; ++r) {
        fs_ctx *c = cs[r];
        if (n > 1) CTX(c);
        for (int slot = 0; slot < SLOT_COUNT; ++slot) TRY(c, join_slot(c, slot));   // transfers a discarded chain had in flight
        std::copy(c->chk_saved, c->chk_saved + F_COUNT, c->f);
        c->stat_redo += 1;
    }
    int rc = advect_part(cs, n, h_true, true);   // (a tracer step, if any, follows the check: run_frames_impl)
    if (rc) return rc;
    for (int r = 0; r < n; ++r) {
        fs_ctx *c = cs[r];
        if (n > 1) CTX(c);
        const fs_grid g = grid_of(c, c->G);
        float *u = c->f[F_VEL].p(), *v = u + c->plane, *d = c->f[F_DYE].p();
        c->ev_main_fresh = false;
        for (const fs_ctx::Splat &sp : c->splat_log)
            TRY(c, fs_k_addForces(&g, u, v, d, d + c->plane, d + 2 * c->plane, sp.px, sp.py, sp.fx, sp.fy, sp.r, sp.g, sp.b,
                                  sp.radius, sp.dt, c->stream));
        c->steps.clear();
    }
    return FS_OK;
}

// the check at a call that has not enqueued anything on top of the frame (fs_sync, a read, a frame that cannot run ahead)
static int settle_all(fs_ctx **cs, int n)
{
    if (!cs[0]->chk_pending) return FS_OK;
    int h_true = 0;
    bool miss = false;
    int rc = check_pending(cs, n, &h_true, &miss);
    if (rc) return rc;
    if (miss && (rc = repair_advection(cs, n, h_true))) return rc;
    for (int r = 0; r < n; ++r) {
        cs[r]->splat_log.clear();
        push_halo(cs[r], h_true);
    }
    return FS_OK;
}

int settle(fs_ctx *c)
{
    if (!c || !c->chk_pending) return FS_OK;
    fs_comm *cm = c->comm;
    if (!cm || cm->solo()) return settle_all(&c, 1);
    for (fs_ctx *m : cm->members)
        if (!m) { c->chk_pending = false; c->splat_log.clear(); return FS_OK; }   // the group is being torn down
    std::vector<fs_ctx *> cs(cm->members);
    const int rc = settle_all(cs.data(), (int)cs.size());
    (void)hipSetDevice(c->device);
    return rc;
}

void drop_segment_graphs(fs_ctx *c)
{
    if (c->seg_graphs.empty()) return;
    // (a replay may still be running, and not every runtime puts the destruction of an executing graph off; drops are rare)
    if (c->stream && !c->dry) (void)hipStreamSynchronize(c->stream);
    for (auto &kv : c->seg_graphs) (void)hipGraphExecDestroy(kv.second);
    c->seg_graphs.clear();
}

static int guess_halo(const fs_ctx *c)
{
    // the bound is set by the newest splat's jet and jumps by 2x from frame to frame (profiles/r01_probe_halo_series.log):
    // the envelope of the last four frames known, + 25 % + 4 rows
    const int envelope = *std::max_element(c->halo_hist, c->halo_hist + 4);
    return std::min(c->G, (envelope * 5 + 3) / 4 + 4);
}

// update() up to the CFL reduction on every slab (plan A); h_dye >= 0: prefetch the dye halo of that depth
static int chain_part(fs_ctx **cs, int n, int h_dye)
{
    for (int r = 0; r < n; ++r) {
        HostTimer t(cs[r]->host_part_us[0]);
        cs[r]->steps.clear();
        cs[r]->plan_salt = 0x100ull | roles_of(cs[r]);
        plan_update_a(cs[r], h_dye);
    }
    return exec_all(cs, n);
}

// One frame on n slabs.  A slab frame that may speculate runs one frame ahead of its checks:
//   frame f:      chain (A) | advection with a GUESSED halo (B), roles from before B remembered      -> returns, no wait
//   frame f + 1:  velocity roles rotated, so that this chain writes into the spare and B(f)'s input survives | chain (A)
//                 | only now the host looks at frame f's all-reduced bound -- by then long there, while the GPU has B(f)
//                 and this chain queued, so it never runs dry | guess too small (rare): B(f) again with the true depth
//                 from the remembered roles, the splats since repeated, the chain again | B(f + 1) ...
// Any other call on the context (fs_sync, a read ...) runs the owed check first (settle).
static int run_frames_impl(fs_ctx **cs, int n)
{
    fs_ctx *c0 = cs[0];
    const bool multi = c0->world > 1;
    const fs_params &p = c0->params;
    const bool ddn = p.diffusion > 0.0f && p.solverIterations > 0;
    const bool tracers = multi && c0->with_particles;
    const bool may_spec = multi && c0->overlap && !ddn;
    // check the last frame AFTER this frame's chain is enqueued?  Needs the spare velocity buffer and a frame that
    // speculates itself (its chain then ends without the host reading anything back)
    for (int r = 0; r < n; ++r) cs[r]->ev_main_fresh = false;   // (splats, writes, stages ... since the last frame)
    const bool late = c0->chk_pending && may_spec && c0->defer_check && !tracers && c0->f[F_VEL].buf[2] != nullptr;
    int rc;
    if (!late && (rc = settle_all(cs, n))) return rc;
    if (multi)
        for (int r = 0; r < n; ++r) {
            if (n > 1) CTX(cs[r]);
            ensure_band_planes(cs[r], p.solverIterations);
            cs[r]->cfl_slot ^= 1;
        }
    const bool spec = may_spec && c0->halo_hist[0] >= 0;
    int h = 0;
    if (late) {
        for (int r = 0; r < n; ++r) cs[r]->f[F_VEL].rotate();
        // (the dye prefetch is sized without the last frame's bound; B below is sized with it and tops the dye halo up if need be)
        if ((rc = chain_part(cs, n, guess_halo(c0)))) return rc;
        int h_last = 0;
        bool miss = false;
        if ((rc = check_pending(cs, n, &h_last, &miss))) return rc;
        if (miss) {
            // this chain ran on a wrong velocity: put B(f) right, then enqueue the chain again (same CFL slot: nobody
            // has waited for the discarded reduction)
            if ((rc = repair_advection(cs, n, h_last))) return rc;
            for (int r = 0; r < n; ++r) { push_halo(cs[r], h_last); cs[r]->splat_log.clear(); cs[r]->f[F_VEL].rotate(); }
            if ((rc = chain_part(cs, n, guess_halo(c0)))) return rc;
        } else {
            for (int r = 0; r < n; ++r) { push_halo(cs[r], h_last); cs[r]->splat_log.clear(); }
        }
        h = guess_halo(c0);
    } else {
        if (spec) h = guess_halo(c0);
        if ((rc = chain_part(cs, n, spec ? h : -1))) return rc;
    }
    int h_true = 0;
    if (multi && !spec) {
        for (int r = 0; r < n; ++r) {
            TRY(cs[r], wait_cfl(cs[r], cs[r]->cfl_slot));
            TRY(cs[r], halo_from_cfl(cs[r], &h_true));
        }
        h = h_true;
    }
    if (spec)
        for (int r = 0; r < n; ++r) {
            fs_ctx *c = cs[r];
            std::copy(c->f, c->f + F_COUNT, c->chk_saved);
            c->chk_h = h;
            c->chk_slot = c->cfl_slot;
            c->splat_log.clear();
        }
    if ((rc = advect_part(cs, n, h, spec && tracers))) return rc;
    if (spec) {
        for (int r = 0; r < n; ++r) cs[r]->chk_pending = true;
        // the check is put off to the next call unless something host-synchronous follows in this frame anyway
        if (!c0->defer_check || tracers) {
            if ((rc = settle_all(cs, n))) return rc;
        }
    } else if (multi) {
        for (int r = 0; r < n; ++r) push_halo(cs[r], h_true);
    }
    if (spec && tracers && (rc = tracer_part(cs, n))) return rc;
    for (int r = 0; r < n; ++r) {
        cs[r]->steps.clear();
        cs[r]->stat_frames += 1;
        if (multi) { cs[r]->stable_frames += 1; cs[r]->p_band_dirty = cs[r]->p_band_active; }
    }
    return FS_OK;
}

int run_frame(fs_ctx *c) { return run_frames(&c, 1); }

static void append_plan(const fs_ctx *c, std::string &out)
{
    static const char *names[F_COUNT] = {"vel", "dye", "p", "div", "vort"};
    char line[160];
    for (const Step &s : c->steps) {
        if (s.hidden) continue;
        if (s.kind == STEP_COMPUTE) {
            snprintf(line, sizeof(line), "%s %d %d %d\n", s.what, s.yb, s.ye, s.n);
            out += line;
        } else if (s.kind == STEP_JOIN || s.kind == STEP_TRACERS) {
            out += s.what;
            out += "\n";
        } else if (s.kind == STEP_EXCHANGE) {
            out += s.what;
            for (const Xfer &x : s.xf) {
                snprintf(line, sizeof(line), " %s:%d", names[x.field], x.depth);
                out += line;
            }
            out += "\n";
        } else {
            out += "cfl_allreduce\n";
        }
    }
}

}  // namespace fs

using namespace fs;

int create_common(int width, int height, int device, fs_comm *comm, int world, int rank, int ghost_rows,
                  long max_particles, bool dry, fs_ctx **out);   // context.hip

extern "C" {

int fs_comm_unique_id(void *id128)
{
    if (!id128) { set_error("fs_comm_unique_id: NULL buffer"); return FS_ERR_INVALID; }
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == FS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    FS_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return FS_OK;
}

int fs_comm_create_rccl(int world, int rank, int device, const void *id128, fs_comm **out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("fs_comm_create_rccl: bad arguments"); return FS_ERR_INVALID; }
    *out = nullptr;
    int rc = load_rccl();
    if (rc) return rc;
    FS_HIP(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t nc = nullptr;
    FS_NCCL(g_rccl.CommInitRank(&nc, world, id, rank));
    fs_comm *cm = new (std::nothrow) fs_comm();
    if (!cm) { (void)g_rccl.CommDestroy(nc); return FS_ERR_NOMEM; }
    cm->kind = 1; cm->world = world; cm->rank = rank; cm->device = device; cm->nccl = nc;
    *out = cm;
    return FS_OK;
}

static int create_local(int world, int kind, fs_comm **out)
{
    if (!out || world < 1 || world > 1024) { set_error("fs_comm_create_local: bad arguments"); return FS_ERR_INVALID; }
    fs_comm *cm = new (std::nothrow) fs_comm();
    if (!cm) return FS_ERR_NOMEM;
    cm->kind = kind; cm->world = world;
    cm->members.assign(world, nullptr);
    *out = cm;
    return FS_OK;
}

int fs_comm_create_local(int world, fs_comm **out) { return create_local(world, 0, out); }

int fs_comm_create_loopback(int world, int rank, fs_comm **out)
{
    if (rank < 0 || rank >= world) { set_error("fs_comm_create_loopback: rank %d of %d", rank, world); return FS_ERR_INVALID; }
    int rc = create_local(world, 3, out);
    if (!rc) { (*out)->rank = rank; (*out)->members.clear(); }
    return rc;
}

int fs_comm_create_local_async(int world, fs_comm **out) { return create_local(world, 2, out); }

int fs_comm_destroy(fs_comm *cm)
{
    if (!cm) return FS_OK;
    if (cm->kind == 1 && cm->nccl && g_rccl.CommDestroy) {
        (void)hipSetDevice(cm->device);
        (void)g_rccl.CommDestroy((ncclComm_t)cm->nccl);
    }
    delete cm;
    return FS_OK;
}

// Round trip through the transport: a grouped ncclSend/ncclRecv of `count` floats to the next rank
// and from the previous one (to itself when world == 1) and a 1-float all-reduce(max).
int fs_comm_selftest(fs_comm *cm, int count)
{
    if (!cm || cm->kind != 1 || count < 1) { set_error("fs_comm_selftest: needs an RCCL communicator"); return FS_ERR_INVALID; }
    FS_HIP(hipSetDevice(cm->device));
    const int next = (cm->rank + 1) % cm->world, prev = (cm->rank + cm->world - 1) % cm->world;
    std::vector<float> h(count), back(count);
    for (int i = 0; i < count; ++i) h[i] = (float)(cm->rank * 1000003 + i);
    float *src = nullptr, *dst = nullptr, *red = nullptr;
    hipStream_t st = nullptr;
    int rc = FS_OK;
    hipError_t e;
    float mine = (float)(cm->rank + 1), got = 0.0f;
#define ST_HIP(call) do { if (!rc && (e = (call)) != hipSuccess) rc = hip_fail(e, #call); } while (0)
#define ST_NCCL(call) do { ncclResult_t r_; if (!rc && (r_ = (call)) != ncclSuccess) rc = rccl_fail(r_, #call); } while (0)
    ST_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    ST_HIP(hipMalloc((void **)&src, count * sizeof(float)));
    ST_HIP(hipMalloc((void **)&dst, count * sizeof(float)));
    ST_HIP(hipMalloc((void **)&red, sizeof(float)));
    ST_HIP(hipMemcpy(src, h.data(), count * sizeof(float), hipMemcpyHostToDevice));
    ST_HIP(hipMemset(dst, 0, count * sizeof(float)));
    ST_HIP(hipMemcpy(red, &mine, sizeof(float), hipMemcpyHostToDevice));
    ST_NCCL(g_rccl.GroupStart());
    ST_NCCL(g_rccl.Send(src, count, ncclFloat, next, (ncclComm_t)cm->nccl, st));
    ST_NCCL(g_rccl.Recv(dst, count, ncclFloat, prev, (ncclComm_t)cm->nccl, st));
    ST_NCCL(g_rccl.GroupEnd());
    ST_NCCL(g_rccl.AllReduce(red, red, 1, ncclFloat, ncclMax, (ncclComm_t)cm->nccl, st));
    ST_HIP(hipStreamSynchronize(st));
    ST_HIP(hipMemcpy(back.data(), dst, count * sizeof(float), hipMemcpyDeviceToHost));
    ST_HIP(hipMemcpy(&got, red, sizeof(float), hipMemcpyDeviceToHost));
#undef ST_HIP
#undef ST_NCCL
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(red);
    if (st) (void)hipStreamDestroy(st);
    if (rc) return rc;
    for (int i = 0; i < count; ++i)
        if (back[i] != (float)(prev * 1000003 + i)) { set_error("fs_comm_selftest: element %d arrived as %g", i, (double)back[i]); return FS_ERR_STATE; }
    if (got != (float)cm->world) { set_error("fs_comm_selftest: all-reduce(max) gave %g, expected %d", (double)got, cm->world); return FS_ERR_STATE; }
    return FS_OK;
}

int fs_create_slab(int width, int height, int device, fs_comm *comm, int rank, int ghost_rows, fs_ctx **out)
{
    if (!comm) { set_error("fs_create_slab: communicator is NULL"); return FS_ERR_INVALID; }
    if (comm->solo()) rank = comm->rank;
    if (rank < 0 || rank >= comm->world) { set_error("fs_create_slab: rank %d of %d", rank, comm->world); return FS_ERR_INVALID; }
    if (comm->kind == 1 && device != comm->device) { set_error("fs_create_slab: the communicator lives on device %d", comm->device); return FS_ERR_INVALID; }
    if (!comm->solo() && comm->members[rank]) { set_error("fs_create_slab: rank %d already exists in this group", rank); return FS_ERR_INVALID; }
    int rc = create_common(width, height, device, comm, comm->world, rank, ghost_rows, 0, false, out);
    if (!rc && !comm->solo()) comm->members[rank] = *out;
    return rc;
}

int fs_slab_info(fs_ctx *c, int info[6])
{
    if (!c || !info) return FS_ERR_INVALID;
    TRY(c, fs::settle(c));   // (the last advection halo is known once the frame's check has run)
    info[0] = c->r0; info[1] = c->r1; info[2] = c->b0; info[3] = c->b1; info[4] = c->G; info[5] = c->last_halo;
    return FS_OK;
}

// Milliseconds the kernels of this slab waited at joins since the last call, per slot (velocity halo, dye halo,
// blocking exchanges), and the number of joins measured; needs FS_OPT_SLAB_PROFILE = 1.  Blocks like fs_sync.
int fs_slab_waits(fs_ctx *c, double wait_ms[3], long joins[3])
{
    CTX(c);
    TRY(c, fs::settle(c));
    if (!wait_ms || !joins) { set_error("fs_slab_waits: NULL output"); return ctx_fail(c, FS_ERR_INVALID); }
    if (c->xstream) TRYHIP(c, hipStreamSynchronize(c->xstream));
    TRYHIP(c, hipStreamSynchronize(c->stream));
    for (const fs_ctx::WaitProbe &pr : c->probes_live) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) { c->wait_ms[pr.slot] += ms; c->wait_count[pr.slot] += 1; }
        c->probes_free.push_back(pr);
    }
    c->probes_live.clear();
    for (int i = 0; i < SLOT_COUNT; ++i) {
        wait_ms[i] = c->wait_ms[i]; joins[i] = c->wait_count[i];
        c->wait_ms[i] = 0.0; c->wait_count[i] = 0;
    }
    return FS_OK;
}

int fs_slab_host_times(fs_ctx *c, double us[2], long *frames)
{
    if (!c || !us || !frames) return FS_ERR_INVALID;
    us[0] = c->host_frame_us; us[1] = c->host_wait_us; *frames = c->host_frames;
    c->host_frame_us = c->host_wait_us = 0.0; c->host_frames = 0;
    return FS_OK;
}

int fs_slab_host_breakdown(fs_ctx *c, double us[8])
{
    if (!c || !us) return FS_ERR_INVALID;
    for (int i = 0; i < 8; ++i) { us[i] = c->host_part_us[i]; c->host_part_us[i] = 0.0; }
    return FS_OK;
}

int fs_slab_graph_stats(fs_ctx *c, long stats[4])
{
    if (!c || !stats) return FS_ERR_INVALID;
    stats[0] = (long)c->seg_graphs.size(); stats[1] = c->stat_seg_replays; stats[2] = c->stat_seg_captures; stats[3] = c->stat_seg_eager;
    return FS_OK;
}

int fs_slab_stats(fs_ctx *c, long stats[5])
{
    if (!c || !stats) return FS_ERR_INVALID;
    TRY(c, fs::settle(c));
    stats[0] = c->stat_exchanges; stats[1] = c->stat_rows; stats[2] = c->stat_frames; stats[3] = c->stat_redo;
    stats[4] = c->stat_migrated;
    return FS_OK;
}

// One update() on every slab of an in-process group, in lock step: step i of every rank runs before
// step i+1 of any, which is the ordering the exchanges of a real multi-process run enforce.
int fs_group_step(fs_ctx **ctxs, int n)
{
    if (!ctxs || n < 1 || !ctxs[0] || !ctxs[0]->comm || ctxs[0]->comm->solo() || ctxs[0]->comm->world != n) {
        set_error("fs_group_step: needs all %d contexts of one local group, in rank order", n);
        return FS_ERR_INVALID;
    }
    for (int r = 0; r < n; ++r)
        if (!ctxs[r] || ctxs[r]->comm != ctxs[0]->comm || ctxs[r]->rank != r) {
            set_error("fs_group_step: context %d is not rank %d of the group", r, r);
            return FS_ERR_INVALID;
        }
    return run_frames(ctxs, n);
}

// Host-only: the frame plans of `frames` consecutive update() calls of one rank, as text (one step per
// line: "<stage> <yb> <ye> <n>", "exchange <field>:<depth> ...", "cfl_allreduce"), with the advection
// halo of frame f given by halos[f] (and guessed exactly right by the frames that speculate).  No GPU is touched.  Tests compare it with the Python planner.
int fs_slab_plan_dump(int width, int height, int world, int rank, int ghost_rows, int overlap, const fs_params *p,
                      int frames, const int *halos, char *buf, size_t cap)
{
    if (!p || !buf || cap < 1 || frames < 0 || (frames > 0 && world > 1 && !halos)) { set_error("fs_slab_plan_dump: bad arguments"); return FS_ERR_INVALID; }
    fs_ctx *c = nullptr;
    int rc = create_common(width, height, 0, nullptr, world, rank, ghost_rows, 0, true, &c);
    if (rc) return rc;
    c->params = *p;
    std::string out;
    c->overlap = overlap ? 1 : 0;
    const bool ddn = p->diffusion > 0.0f && p->solverIterations > 0;
    for (int f = 0; f < frames; ++f) {
        // a frame after the first speculates on its halo (here: exactly right) unless the dye diffuses
        const bool spec = world > 1 && c->overlap && f > 0 && !ddn;
        c->steps.clear();
        plan_update_a(c, spec ? halos[f] : -1);
        append_plan(c, out);
        c->steps.clear();
        plan_update_b(c, world > 1 ? halos[f] : 0);
        append_plan(c, out);
    }
    c->steps.clear();
    delete c;
    if (out.size() + 1 > cap) { set_error("fs_slab_plan_dump: buffer of %zu bytes is too small for %zu", cap, out.size() + 1); return FS_ERR_INVALID; }
    memcpy(buf, out.c_str(), out.size() + 1);
    return FS_OK;
}

}  // extern "C"
