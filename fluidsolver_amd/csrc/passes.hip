// passes.hip -- single-pass kernels of the stable-fluids step for gfx950.
//
// Every kernel here is a streaming pass bounded by HBM bandwidth.  Layout: planar
// fp32, 256-byte-aligned rows (pitch % 64 == 0).  One thread owns 4 consecutive cells
// of one row, so every center/top/bottom access is one aligned 16-byte load per lane
// (1 KiB per wave instruction, the widest coalesced form on CDNA4); the two cells
// beside the quad come from one extra dword load each (L1/L2 hits).  A workgroup is
// 64 x 4 threads = 256 columns x 4 rows, so the rows it re-reads as top/bottom
// neighbours are shared through the vector L1.
//
// Arithmetic follows the reference shaders operation by operation (cited per kernel,
// Shaders = fluidSolver/Shaders/FluidShaders.metal); the library is compiled with
// -ffp-contract=off so no FMA is formed and results are bit-identical to the oracle.
#include "common.h"
#include <cstdlib>
#include <math.h>
#include <stdint.h>

namespace fs {

constexpr int BX = 64;  // lanes along x (each 4 cells)
constexpr int BY = 4;   // rows per workgroup
constexpr int XCD_BAND = 8;   // tile rows per XCD band (32 grid rows), see xcd_tile

struct Quad {
    int x, y, z;   // first cell of the quad, row, plane
    bool ok;
};

__device__ __forceinline__ Quad quad_of_thread(const GK &g, int yb, int ye)
{
    // thin launches (the rims of a slab's split advection: a few dozen rows) are dispatched unpadded and in plain order:
    // padded to the 64 tile rows of one round of XCD bands they were mostly workgroups that exit at once (quad_grid)
    const uint2 t = gridDim.y < 8u * XCD_BAND ? make_uint2(blockIdx.x, blockIdx.y) : xcd_tile<XCD_BAND>();
    Quad q;
    q.x = (t.x * BX + threadIdx.x) * 4;
    q.y = yb + t.y * BY + threadIdx.y;
    q.z = blockIdx.z;
    q.ok = q.x < g.pitch && q.y < ye;
    return q;
}

// center quad plus the cells left of .x and right of .w (0 outside the row)
struct RowN {
    float4 c;
    float l, r;
};
__device__ __forceinline__ RowN load_row_nb(const float *p, const GK &g, int x, int y)
{
    const float *row = p + row_off(g, y);
    RowN n;
    n.c = ld4(row + x);
    n.l = x > 0 ? row[x - 1] : 0.0f;
    n.r = x + 4 < g.pitch ? row[x + 4] : 0.0f;
    return n;
}
__device__ __forceinline__ float4 lefts(const RowN &n) { return make_float4(n.l, n.c.x, n.c.y, n.c.z); }
__device__ __forceinline__ float4 rights(const RowN &n) { return make_float4(n.c.y, n.c.z, n.c.w, n.r); }

#define FOR4(expr_x, expr_y, expr_z, expr_w) \
    do { expr_x; expr_y; expr_z; expr_w; } while (0)

// ---------------------------------------------------------------------------
// divergence, Shaders:162-181:  div = -0.5*((r.x - l.x) + (b.y - t.y)); boundary 0
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_divergence(GK g, int yb, int ye, const float *__restrict__ u,
                                                    const float *__restrict__ v, float *__restrict__ div)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    RowN un = load_row_nb(u, g, q.x, q.y);
    float4 vt = ld4(v + row_off(g, clamp_row(g, q.y - 1)) + q.x);
    float4 vb = ld4(v + row_off(g, clamp_row(g, q.y + 1)) + q.x);
    float4 ul = lefts(un), ur = rights(un);
    float4 o;
    o.x = is_boundary(g, q.x + 0, q.y) ? 0.0f : -0.5f * ((ur.x - ul.x) + (vb.x - vt.x));
    o.y = is_boundary(g, q.x + 1, q.y) ? 0.0f : -0.5f * ((ur.y - ul.y) + (vb.y - vt.y));
    o.z = is_boundary(g, q.x + 2, q.y) ? 0.0f : -0.5f * ((ur.z - ul.z) + (vb.z - vt.z));
    o.w = is_boundary(g, q.x + 3, q.y) ? 0.0f : -0.5f * ((ur.w - ul.w) + (vb.w - vt.w));
    st4s(div + row_off(g, q.y) + q.x, o, g);
}

// ---------------------------------------------------------------------------
// calculateVorticity, Shaders:256-275: curl = (b.x - t.x) - (r.y - l.y); boundary 0
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vorticity(GK g, int yb, int ye, const float *__restrict__ u,
                                                   const float *__restrict__ v, float *__restrict__ w)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    RowN vn = load_row_nb(v, g, q.x, q.y);
    float4 ut = ld4(u + row_off(g, clamp_row(g, q.y - 1)) + q.x);
    float4 ub = ld4(u + row_off(g, clamp_row(g, q.y + 1)) + q.x);
    float4 vl = lefts(vn), vr = rights(vn);
    float4 o;
    o.x = is_boundary(g, q.x + 0, q.y) ? 0.0f : (ub.x - ut.x) - (vr.x - vl.x);
    o.y = is_boundary(g, q.x + 1, q.y) ? 0.0f : (ub.y - ut.y) - (vr.y - vl.y);
    o.z = is_boundary(g, q.x + 2, q.y) ? 0.0f : (ub.z - ut.z) - (vr.z - vl.z);
    o.w = is_boundary(g, q.x + 3, q.y) ? 0.0f : (ub.w - ut.w) - (vr.w - vl.w);
    st4s(w + row_off(g, q.y) + q.x, o, g);
}

// ---------------------------------------------------------------------------
// subtractGradient, Shaders:225-245: vel.x -= 0.5*(r-l); vel.y -= 0.5*(b-t); boundary untouched
// ---------------------------------------------------------------------------
__device__ __forceinline__ float subtract_gradient_quad(const GK &g, const Quad &q, const float *__restrict__ p,
                                                        float *__restrict__ u, float *__restrict__ v)
{
    RowN pn = load_row_nb(p, g, q.x, q.y);
    float4 pt = ld4(p + row_off(g, clamp_row(g, q.y - 1)) + q.x);
    float4 pb = ld4(p + row_off(g, clamp_row(g, q.y + 1)) + q.x);
    float4 pl = lefts(pn), pr = rights(pn);
    size_t o = row_off(g, q.y) + q.x;
    float4 uu = ld4(u + o), vv = ld4(v + o);
    if (!is_boundary(g, q.x + 0, q.y)) { uu.x = uu.x - 0.5f * (pr.x - pl.x); vv.x = vv.x - 0.5f * (pb.x - pt.x); }
    if (!is_boundary(g, q.x + 1, q.y)) { uu.y = uu.y - 0.5f * (pr.y - pl.y); vv.y = vv.y - 0.5f * (pb.y - pt.y); }
    if (!is_boundary(g, q.x + 2, q.y)) { uu.z = uu.z - 0.5f * (pr.z - pl.z); vv.z = vv.z - 0.5f * (pb.z - pt.z); }
    if (!is_boundary(g, q.x + 3, q.y)) { uu.w = uu.w - 0.5f * (pr.w - pl.w); vv.w = vv.w - 0.5f * (pb.w - pt.w); }
    st4s(u + o, uu, g);
    st4s(v + o, vv, g);
    // |v| of the cells inside the domain (padding columns are don't-care)
    float m = fabsf(vv.x);
    if (q.x + 1 < g.W) m = fmaxf(m, fabsf(vv.y));
    if (q.x + 2 < g.W) m = fmaxf(m, fabsf(vv.z));
    if (q.x + 3 < g.W) m = fmaxf(m, fabsf(vv.w));
    return q.x < g.W ? m : 0.0f;
}

// CFL: also max |v| of the new v plane over the launch (in-domain columns only), one atomicMax per
// workgroup (non-negative floats order like their bit patterns).
template <bool CFL>
__global__ __launch_bounds__(256) void k_subtract_gradient(GK g, int yb, int ye, const float *__restrict__ p,
                                                           float *__restrict__ u, float *__restrict__ v,
                                                           unsigned *__restrict__ cfl)
{
    __shared__ float wave_max[4];
    Quad q = quad_of_thread(g, yb, ye);
    if (CFL) {
        // every thread of the workgroup reaches the barrier: out-of-range quads contribute 0
        float m = 0.0f;
        if (q.ok) m = subtract_gradient_quad(g, q, p, u, v);
        for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
        const int tid = threadIdx.y * BX + threadIdx.x;
        if ((tid & 63) == 0) wave_max[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
            if (m > 0.0f) atomic_max_if_larger(cfl, __float_as_uint(m));
        }
        return;
    }
    if (!q.ok) return;
    (void)subtract_gradient_quad(g, q, p, u, v);
}

// ---------------------------------------------------------------------------
// applyVorticity, Shaders:288-316
// ---------------------------------------------------------------------------
__device__ __forceinline__ void vort_force(float l, float r, float t, float b, float c, float strength, float dt,
                                           float &uu, float &vv)
{
    float gx = fabsf(r) - fabsf(l);              // :301-307
    float gy = fabsf(b) - fabsf(t);
    float glen = sqrtf(gx * gx + gy * gy) + 0.00001f;  // :308
    gx = gx / glen;                              // :309
    gy = gy / glen;
    float sc = strength * c;                     // :311
    uu = uu + (sc * gy) * dt;                    // :314
    vv = vv + (sc * (-gx)) * dt;
}

// calculateVorticity + applyVorticity in one pass, for planes that outgrow the Infinity Cache: every vorticity value
// a cell needs (its own and its four neighbours') is recomputed from u, v -- the re-read rows come out of L2 -- so
// the vorticity plane is neither written nor read back (32 -> 16 B/cell of HBM traffic).  The confinement force
// needs the OLD velocity of the neighbours, so the result goes to the ping-pong partner, not in place.  Same
// expression per value, hence the same bits as the two kernels.
__device__ __forceinline__ float curl_at(const GK &g, int x, int y, float u_b, float u_t, float v_r, float v_l)
{
    return is_boundary(g, x, y) ? 0.0f : (u_b - u_t) - (v_r - v_l);           // Shaders:262-273
}
__global__ __launch_bounds__(256) void k_vorticity_apply(GK g, int yb, int ye, const float *__restrict__ u,
                                                         const float *__restrict__ v, float *__restrict__ uo,
                                                         float *__restrict__ vo, float strength, float dt)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    const int x = q.x, y = q.y;
    const int ym1 = clamp_row(g, y - 1), ym2 = clamp_row(g, y - 2), yp1 = clamp_row(g, y + 1), yp2 = clamp_row(g, y + 2);
    const float4 u0 = ld4(u + row_off(g, y) + x), um2 = ld4(u + row_off(g, ym2) + x), up2 = ld4(u + row_off(g, yp2) + x);
    const RowN um1 = load_row_nb(u, g, x, ym1), up1 = load_row_nb(u, g, x, yp1);
    const RowN vm1 = load_row_nb(v, g, x, ym1), v0 = load_row_nb(v, g, x, y), vp1 = load_row_nb(v, g, x, yp1);
    const float *vrow = v + row_off(g, y);
    const float v_ll = x > 1 ? vrow[x - 2] : 0.0f, v_rr = x + 5 < g.pitch ? vrow[x + 5] : 0.0f;
    // vorticity of row y at columns x-1 .. x+4
    const float wl = curl_at(g, x - 1, y, up1.l, um1.l, v0.c.x, v_ll);
    const float w0 = curl_at(g, x + 0, y, up1.c.x, um1.c.x, v0.c.y, v0.l);
    const float w1 = curl_at(g, x + 1, y, up1.c.y, um1.c.y, v0.c.z, v0.c.x);
    const float w2 = curl_at(g, x + 2, y, up1.c.z, um1.c.z, v0.c.w, v0.c.y);
    const float w3 = curl_at(g, x + 3, y, up1.c.w, um1.c.w, v0.r, v0.c.z);
    const float wr = curl_at(g, x + 4, y, up1.r, um1.r, v_rr, v0.c.w);
    // vorticity of rows y-1 and y+1 at columns x .. x+3
    const float4 vl_t = lefts(vm1), vr_t = rights(vm1), vl_b = lefts(vp1), vr_b = rights(vp1);
    const float t0 = curl_at(g, x + 0, y - 1, u0.x, um2.x, vr_t.x, vl_t.x), t1 = curl_at(g, x + 1, y - 1, u0.y, um2.y, vr_t.y, vl_t.y);
    const float t2 = curl_at(g, x + 2, y - 1, u0.z, um2.z, vr_t.z, vl_t.z), t3 = curl_at(g, x + 3, y - 1, u0.w, um2.w, vr_t.w, vl_t.w);
    const float b0 = curl_at(g, x + 0, y + 1, up2.x, u0.x, vr_b.x, vl_b.x), b1 = curl_at(g, x + 1, y + 1, up2.y, u0.y, vr_b.y, vl_b.y);
    const float b2 = curl_at(g, x + 2, y + 1, up2.z, u0.z, vr_b.z, vl_b.z), b3 = curl_at(g, x + 3, y + 1, up2.w, u0.w, vr_b.w, vl_b.w);
    float4 uu = u0, vv = v0.c;
    if (!is_boundary(g, x + 0, y)) vort_force(wl, w1, t0, b0, w0, strength, dt, uu.x, vv.x);
    if (!is_boundary(g, x + 1, y)) vort_force(w0, w2, t1, b1, w1, strength, dt, uu.y, vv.y);
    if (!is_boundary(g, x + 2, y)) vort_force(w1, w3, t2, b2, w2, strength, dt, uu.z, vv.z);
    if (!is_boundary(g, x + 3, y)) vort_force(w2, wr, t3, b3, w3, strength, dt, uu.w, vv.w);
    const size_t o = row_off(g, y) + x;
    st4s(uo + o, uu, g);
    st4s(vo + o, vv, g);
}

// The same fusion for planes that FIT the cache, where re-reading rows buys nothing and the two streaming kernels used to
// win: a wave owns a 256-column strip (lane l: columns 4l..4l+3, the outermost lanes are halo) and MARCHES down K rows with
// a rolling window in registers -- three rows of u, two of v, three of vorticity -- so every row of u and v is loaded once
// per strip (+4 / +2 rows of run-in per K), the vorticity of a row is computed once, and its left / right neighbours
// come from the neighbouring lanes by DPP.  Same expression per value (curl_at, vort_force): the same bits.
template <int K, int XM>
__global__ __launch_bounds__(256) void k_vorticity_apply_march(GK g, int yb, int ye, const float *__restrict__ u,
                                                               const float *__restrict__ v, float *__restrict__ uo,
                                                               float *__restrict__ vo, float strength, float dt)
{
    constexpr int OW = 248;                               // columns a strip stores: lanes 1..62
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (XM > 0: bands of XM block rows dealt to the XCDs in turn, see xcd_tile; the launch pads gridDim.y)
    const uint2 t = XM > 0 ? xcd_tile<(XM > 0 ? XM : 1)>() : make_uint2(blockIdx.x, blockIdx.y);
    const int x = (int)t.x * OW - 4 + 4 * lane;           // this lane's first column (multiple of 4)
    const int y0 = yb + ((int)t.y * 4 + wave) * K;        // first row of this wave's chunk
    if (y0 >= ye) return;
    const bool in_buf = x >= 0 && x < g.pitch;
    const bool stores = lane >= 1 && lane <= 62 && in_buf;
    // Loads are unconditional (a lane outside the buffer reads the nearest quad inside it): no exec-mask branch between
    // them, so the rows of the next step stay in flight under this step's arithmetic.  What such a lane holds reaches
    // only boundary and padding cells, whose curl is 0 by definition and which get no force.
    const int xc = x < 0 ? 0 : x > g.pitch - 4 ? g.pitch - 4 : x;
    auto row = [&](const float *p, int r) { return ld4(p + row_off(g, clamp_row(g, r)) + xc); };
    // vorticity of row r at this lane's four columns, from u(r-1), u(r+1), v(r)            Shaders:262-273
    auto curl_row = [&](int r, const float4 &ut, const float4 &ub, const float4 &vc) {
        const float vl = from_lane_below(vc.w), vr = from_lane_above(vc.x);
        float4 w;
        w.x = curl_at(g, x + 0, r, ub.x, ut.x, vc.y, vl);
        w.y = curl_at(g, x + 1, r, ub.y, ut.y, vc.z, vc.x);
        w.z = curl_at(g, x + 2, r, ub.z, ut.z, vc.w, vc.y);
        w.w = curl_at(g, x + 3, r, ub.w, ut.w, vr, vc.z);
        return w;
    };
    float4 u_m1 = row(u, y0 - 1), u_c = row(u, y0), u_p1 = row(u, y0 + 1);
    float4 v_c = row(v, y0);
    float4 wt, wc;
    {
        const float4 u_m2 = row(u, y0 - 2), v_m1 = row(v, y0 - 1);
        wt = curl_row(y0 - 1, u_m2, u_c, v_m1);
        wc = curl_row(y0, u_m1, u_p1, v_c);
    }
    // (rolled: unrolled with the next step's rows prefetched it needs 72 VGPRs and is slower, 36.5 vs 32 us at 3072^2)
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const int y = y0 + k;
        if (y >= ye) break;
        const float4 u_p2 = row(u, y + 2), v_p1 = row(v, y + 1);
        const float4 wb = curl_row(y + 1, u_c, u_p2, v_p1);
        const float wl = from_lane_below(wc.w), wr = from_lane_above(wc.x);
        float4 uu = u_c, vv = v_c;
        if (!is_boundary(g, x + 0, y)) vort_force(wl, wc.y, wt.x, wb.x, wc.x, strength, dt, uu.x, vv.x);
        if (!is_boundary(g, x + 1, y)) vort_force(wc.x, wc.z, wt.y, wb.y, wc.y, strength, dt, uu.y, vv.y);
        if (!is_boundary(g, x + 2, y)) vort_force(wc.y, wc.w, wt.z, wb.z, wc.z, strength, dt, uu.z, vv.z);
        if (!is_boundary(g, x + 3, y)) vort_force(wc.z, wr, wt.w, wb.w, wc.w, strength, dt, uu.w, vv.w);
        if (stores) {
            const size_t o = row_off(g, y) + x;
            st4s(uo + o, uu, g);
            st4s(vo + o, vv, g);
        }
        wt = wc; wc = wb;
        u_c = u_p1; u_p1 = u_p2;
        v_c = v_p1;
    }
}

__global__ __launch_bounds__(256) void k_apply_vorticity(GK g, int yb, int ye, const float *__restrict__ w,
                                                         float *__restrict__ u, float *__restrict__ v,
                                                         float strength, float dt)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    RowN wn = load_row_nb(w, g, q.x, q.y);
    float4 wt = ld4(w + row_off(g, clamp_row(g, q.y - 1)) + q.x);
    float4 wb = ld4(w + row_off(g, clamp_row(g, q.y + 1)) + q.x);
    float4 wl = lefts(wn), wr = rights(wn);
    size_t o = row_off(g, q.y) + q.x;
    float4 uu = ld4(u + o), vv = ld4(v + o);
    if (!is_boundary(g, q.x + 0, q.y)) vort_force(wl.x, wr.x, wt.x, wb.x, wn.c.x, strength, dt, uu.x, vv.x);
    if (!is_boundary(g, q.x + 1, q.y)) vort_force(wl.y, wr.y, wt.y, wb.y, wn.c.y, strength, dt, uu.y, vv.y);
    if (!is_boundary(g, q.x + 2, q.y)) vort_force(wl.z, wr.z, wt.z, wb.z, wn.c.z, strength, dt, uu.z, vv.z);
    if (!is_boundary(g, q.x + 3, q.y)) vort_force(wl.w, wr.w, wt.w, wb.w, wn.c.w, strength, dt, uu.w, vv.w);
    st4s(u + o, uu, g);
    st4s(v + o, vv, g);
}

// ---------------------------------------------------------------------------
// fadeDensity, Shaders:326-334: d = max(d*(1-fade), 0), every cell; planes on blockIdx.z
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fade(GK g, int yb, int ye, float *__restrict__ d, size_t stride, float k)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    float *p = d + (size_t)q.z * stride + row_off(g, q.y) + q.x;
    float4 x = ld4(p);
    x.x = fmaxf(x.x * k, 0.0f);
    x.y = fmaxf(x.y * k, 0.0f);
    x.z = fmaxf(x.z * k, 0.0f);
    x.w = fmaxf(x.w * k, 0.0f);
    st4(p, x);
}

// clearTexture, Shaders:343-346
__global__ __launch_bounds__(256) void k_clear(GK g, int yb, int ye, float *__restrict__ t, size_t stride)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    st4(t + (size_t)q.z * stride + row_off(g, q.y) + q.x, make_float4(0.f, 0.f, 0.f, 0.f));
}

// ---------------------------------------------------------------------------
// render kernels, Shaders:357-414 (display = 4 planes R,G,B,A)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_render_density(GK g, int yb, int ye, const float *__restrict__ d,
                                                        size_t dstride, int dplanes, float *__restrict__ out,
                                                        size_t ostride, float brightness)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    size_t o = row_off(g, q.y) + q.x;
    // vorticity display mode binds a 1-channel field whose .gba are 0 (Solver:613-618)
    for (int c = 0; c < 3; ++c) {
        float4 x = c < dplanes ? ld4(d + c * dstride + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        x.x *= brightness; x.y *= brightness; x.z *= brightness; x.w *= brightness;   // :363
        st4(out + c * ostride + o, x);
    }
    st4(out + 3 * ostride + o, make_float4(1.f, 1.f, 1.f, 1.f));                       // :364
}

template <bool SPEED_ONLY>
__global__ __launch_bounds__(256) void k_render_velocity(GK g, int yb, int ye, const float *__restrict__ u,
                                                         const float *__restrict__ v, float *__restrict__ out,
                                                         size_t ostride, float scale)
{
    Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    size_t o = row_off(g, q.y) + q.x;
    float4 a = ld4(u + o), b = ld4(v + o);
    float4 sp;
    sp.x = sqrtf(a.x * a.x + b.x * b.x);   // :384 / :411 length(vel)
    sp.y = sqrtf(a.y * a.y + b.y * b.y);
    sp.z = sqrtf(a.z * a.z + b.z * b.z);
    sp.w = sqrtf(a.w * a.w + b.w * b.w);
    sp.x *= scale; sp.y *= scale; sp.z *= scale; sp.w *= scale;
    if (SPEED_ONLY) {                       // :411-413
        st4(out + o, sp);
        st4(out + ostride + o, sp);
    } else {                                // :389-391
        st4(out + o, make_float4(a.x * scale + 0.5f, a.y * scale + 0.5f, a.z * scale + 0.5f, a.w * scale + 0.5f));
        st4(out + ostride + o,
            make_float4(b.x * scale + 0.5f, b.y * scale + 0.5f, b.z * scale + 0.5f, b.w * scale + 0.5f));
    }
    st4(out + 2 * ostride + o, sp);
    st4(out + 3 * ostride + o, make_float4(1.f, 1.f, 1.f, 1.f));
}

// ---------------------------------------------------------------------------
// advect, Shaders:118-151.  One thread per 4 cells of a row: the velocity read and the
// output write are coalesced 16-byte accesses; the bilinear taps are a data-dependent
// gather served by L1/L2 (neighbouring cells trace to neighbouring taps), two 8-byte
// loads per cell and plane.
// ---------------------------------------------------------------------------
// FADE fuses the fadeDensity pass that follows the dye advection in update() (Solver:577-591): the value
// a cell would be given by advect is passed through max(x*(1-fade), 0) before the one store, which is
// exactly what the separate in-place kernel computes from the stored value (Shaders:331-332).
// the two taps (x, x+1) of one row in one 8-byte load; the address is only 4-byte aligned, which global
// loads accept (halves the number of gather instructions the texture-address unit has to expand)
struct __attribute__((packed, aligned(4))) TapPair {
    float a, b;
};
__device__ __forceinline__ TapPair ld_pair(const float *p) { return *reinterpret_cast<const TapPair *>(p); }

template <int NP, bool FADE>
__global__ __launch_bounds__(256) void k_advect(GK g, int yb, int ye, int hole_y, int hole_n, const float *__restrict__ u,
                                                const float *__restrict__ v, const float *in,
                                                float *__restrict__ out, size_t stride, float dt, int *err,
                                                float fade_k)
{
    // rows [yb, ye) less the hole_n rows from hole_y on: the two rims of a slab's split advection in one launch
    Quad q = quad_of_thread(g, yb, ye - hole_n);
    if (!q.ok) return;
    if (q.y >= hole_y) q.y += hole_n;
    const size_t o = row_off(g, q.y) + q.x;
    const float4 uu = ld4(u + o), vv = ld4(v + o);
    const float ux[4] = {uu.x, uu.y, uu.z, uu.w}, vy[4] = {vv.x, vv.y, vv.z, vv.w};
    const bool rowb = q.y == 0 || q.y >= g.H - 1;
    bool bnd[4], any_bnd = false, short_halo = false;
    size_t o0[4], o1[4];
    float sx[4], sy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = q.x + j;
        bnd[j] = rowb || x == 0 || x >= g.W - 1;                  // :127-130 (padding columns count as boundary)
        any_bnd |= bnd[j];
        float posx = (float)x - ux[j] * dt;                       // :132-133
        float posy = (float)q.y - vy[j] * dt;
        posx = fminf(fmaxf(posx, 0.5f), (float)(unsigned)g.W - 1.5f);   // :136
        posy = fminf(fmaxf(posy, 0.5f), (float)(unsigned)g.H - 1.5f);   // :137
        // (W = 1: the clamp range of :136 is empty and posx lands at -0.5; every cell is a boundary cell then and
        // copies its input, the max() only keeps the unused taps inside the buffer)
        const int i0x = max((int)floorf(posx), 0);                // :140
        int i0y = (int)floorf(posy);
        sx[j] = posx - (float)i0x;                                // :142
        sy[j] = posy - (float)i0y;
        int i1y = i0y + 1;
        if (i0y < g.y0 || i1y >= g.y0 + g.rows) {                 // slab halo too small: flag, stay in bounds
            short_halo |= !bnd[j];
            i0y = clamp_row(g, i0y);
            i1y = clamp_row(g, i1y);
        }
        o0[j] = row_off(g, i0y) + i0x;
        o1[j] = row_off(g, i1y) + i0x;
    }
    if (short_halo && err) atomicOr(err, 1);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const float *p = in + c * stride;
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const TapPair r0 = ld_pair(p + o0[j]), r1 = ld_pair(p + o1[j]);   // :144-147
            const float m0 = r0.a + (r0.b - r0.a) * sx[j];        // :149 mix(mix(),mix(),s.y)
            const float m1 = r1.a + (r1.b - r1.a) * sx[j];
            r[j] = m0 + (m1 - m0) * sy[j];
        }
        if (any_bnd) {                                            // boundary cells copy the input, :127-130
            const float4 cen = ld4(p + o);
            if (bnd[0]) r[0] = cen.x;
            if (bnd[1]) r[1] = cen.y;
            if (bnd[2]) r[2] = cen.z;
            if (bnd[3]) r[3] = cen.w;
        }
        if (FADE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = fmaxf(r[j] * fade_k, 0.0f);
        }
        st4s(out + c * stride + o, make_float4(r[0], r[1], r[2], r[3]), g);
    }
}

// ---------------------------------------------------------------------------
// Both advections of update() in one pass (Solver:550-558 velocity by itself, :577-591 dye by the NEW velocity,
// then fade): the dye is traced through the velocity this very thread has just produced for the same cell, so
// the advected velocity is not read back (8 B/cell less) and one launch disappears.  Same operations per cell,
// in the same order, as k_advect<2,false> followed by k_advect<3,true>.
// ---------------------------------------------------------------------------
struct Trace {
    size_t o0[4], o1[4];
    float sx[4], sy[4];
};
__device__ __forceinline__ bool backtrace4(const GK &g, const Quad &q, const float (&ux)[4], const float (&vy)[4], float dt,
                                           const bool (&bnd)[4], Trace &t)
{
    bool short_halo = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = q.x + j;
        float posx = (float)x - ux[j] * dt;                       // :132-133
        float posy = (float)q.y - vy[j] * dt;
        posx = fminf(fmaxf(posx, 0.5f), (float)(unsigned)g.W - 1.5f);   // :136
        posy = fminf(fmaxf(posy, 0.5f), (float)(unsigned)g.H - 1.5f);   // :137
        // (W = 1: the clamp range of :136 is empty and posx lands at -0.5; every cell is a boundary cell then and
        // copies its input, the max() only keeps the unused taps inside the buffer)
        const int i0x = max((int)floorf(posx), 0);                // :140
        int i0y = (int)floorf(posy);
        t.sx[j] = posx - (float)i0x;                              // :142
        t.sy[j] = posy - (float)i0y;
        int i1y = i0y + 1;
        if (i0y < g.y0 || i1y >= g.y0 + g.rows) {                 // slab halo too small: flag, stay in bounds
            short_halo |= !bnd[j];
            i0y = clamp_row(g, i0y);
            i1y = clamp_row(g, i1y);
        }
        t.o0[j] = row_off(g, i0y) + i0x;
        t.o1[j] = row_off(g, i1y) + i0x;
    }
    return short_halo;
}
__device__ __forceinline__ void gather4(const float *p, const Trace &t, float (&r)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const TapPair r0 = ld_pair(p + t.o0[j]), r1 = ld_pair(p + t.o1[j]);   // :144-147
        const float m0 = r0.a + (r0.b - r0.a) * t.sx[j];          // :149 mix(mix(),mix(),s.y)
        const float m1 = r1.a + (r1.b - r1.a) * t.sx[j];
        r[j] = m0 + (m1 - m0) * t.sy[j];
    }
}
template <bool FADE>
__global__ __launch_bounds__(256, 8) void k_advect_both(GK g, int yb, int ye, const float *__restrict__ u,
                                                     const float *__restrict__ v, float *__restrict__ uo,
                                                     float *__restrict__ vo, const float *__restrict__ dye,
                                                     float *__restrict__ dye_out, size_t stride, float dt, int *err,
                                                     float fade_k)
{
    const Quad q = quad_of_thread(g, yb, ye);
    if (!q.ok) return;
    const size_t o = row_off(g, q.y) + q.x;
    const float4 uu = ld4(u + o), vv = ld4(v + o);
    const float ux[4] = {uu.x, uu.y, uu.z, uu.w}, vy[4] = {vv.x, vv.y, vv.z, vv.w};
    const bool rowb = q.y == 0 || q.y >= g.H - 1;
    bool bnd[4], any_bnd = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = q.x + j;
        bnd[j] = rowb || x == 0 || x >= g.W - 1;                  // :127-130 (padding columns count as boundary)
        any_bnd |= bnd[j];
    }
    Trace t;
    bool short_halo = backtrace4(g, q, ux, vy, dt, bnd, t);
    float nu[4], nv[4];
    gather4(u, t, nu);
    gather4(v, t, nv);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (bnd[j]) { nu[j] = ux[j]; nv[j] = vy[j]; }             // boundary cells copy the input, :127-130
    st4s(uo + o, make_float4(nu[0], nu[1], nu[2], nu[3]), g);
    st4s(vo + o, make_float4(nv[0], nv[1], nv[2], nv[3]), g);
    // the dye, through the velocity just computed (Solver:577-582 binds the swapped velocity texture)
    short_halo |= backtrace4(g, q, nu, nv, dt, bnd, t);
    if (short_halo && err) atomicOr(err, 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *p = dye + c * stride;
        float r[4];
        gather4(p, t, r);
        if (any_bnd) {
            const float4 cen = ld4(p + o);
            if (bnd[0]) r[0] = cen.x;
            if (bnd[1]) r[1] = cen.y;
            if (bnd[2]) r[2] = cen.z;
            if (bnd[3]) r[3] = cen.w;
        }
        if (FADE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = fmaxf(r[j] * fade_k, 0.0f);
        }
        st4s(dye_out + c * stride + o, make_float4(r[0], r[1], r[2], r[3]), g);
    }
}

// ---------------------------------------------------------------------------
// addForces, Shaders:43-69, on the splat's bounding box only (the reference launches
// the full grid for <= ~7.9k touched cells).  exp is evaluated in fp64 and rounded
// once, i.e. a correctly rounded fp32 exp.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_forces(GK g, int bx0, int by0, int bx1, int by1, float *__restrict__ u,
                                                    float *__restrict__ v, float *__restrict__ r,
                                                    float *__restrict__ gch, float *__restrict__ b, float px,
                                                    float py, float fx, float fy, float cr, float cg, float cb,
                                                    float radius, float dt)
{
    int x = bx0 + blockIdx.x * 64 + threadIdx.x;
    int y = by0 + blockIdx.y * 4 + threadIdx.y;
    if (x > bx1 || y > by1) return;
    float dx = (float)x - px, dy = (float)y - py;             // :52-53
    float dist = sqrtf(dx * dx + dy * dy);                    // :54
    if (dist < radius) {                                      // :56
        float arg = ((-dist) * dist) / ((2.0f * radius) * radius);   // :57
        float infl = (float)exp((double)arg);
        size_t o = row_off(g, y) + x;
        u[o] = u[o] + (fx * infl) * dt;                       // :61
        v[o] = v[o] + (fy * infl) * dt;
        r[o] = r[o] + (cr * infl) * dt;                       // :66
        gch[o] = gch[o] + (cg * infl) * dt;
        b[o] = b[o] + (cb * infl) * dt;
    }
}

// ---------------------------------------------------------------------------
// tracers: ParticleSystem.update(), Part:242-304 + lookup of Solver:648-657. SoA.
// ---------------------------------------------------------------------------
struct TracerState {
    float x, y, vx, vy, life, age;
};
// one tracer, Part:251-297; returns false for a dead slot (nothing to write)
__device__ __forceinline__ bool tracer_step(const GK &g, const float *__restrict__ u, const float *__restrict__ v,
                                            const fs_particle_params &pp, float eff, float mass, TracerState &t)
{
    float l = t.life;
    if (l == 0.0f) return false;                              // dead slot
    const float invW = 1.0f / pp.windowW, invH = 1.0f / pp.windowH;    // Part:245-248
    const float gw = (float)g.W, gh = (float)g.H;
    const float x = t.x, y = t.y, a = t.age;
    float nx = x * invW, ny = y * invH;                       // Part:255
    int cx = (int)fmaxf(0.0f, fminf(gw - 1.0f, nx * gw));     // Solver:650
    int cy = (int)fmaxf(0.0f, fminf(gh - 1.0f, ny * gh));     // Solver:651
    cy = clamp_row(g, cy);
    size_t o = row_off(g, cy) + cx;
    float fvx = u[o] / gw, fvy = v[o] / gh;                   // build's normalisation (fluidsolver.h)
    float ageDecay = fmaxf(0.0f, 1.0f - a * pp.fluidDecayRate);        // Part:259
    float k = (mass * eff) * ageDecay;                        // Part:271
    float fcx = (fvx * k) * pp.windowW, fcy = (fvy * k) * pp.windowH;
    float em = pp.momentum + (1.0f - pp.momentum) * fminf(1.0f, a * 2.0f);   // Part:272
    float nvx = fcx + t.vx * em, nvy = fcy + t.vy * em;       // Part:273-275
    float npx = x + nvx, npy = y + nvy;                       // Part:276
    const float margin = 50.0f;                               // Part:282
    if (npx < -margin || npx > pp.windowW + margin || npy < -margin || npy > pp.windowH + margin) l = 0.0f;
    l = l * 0.999f;                                           // Part:291
    if (l < 0.01f) l = 0.0f;                                  // Part:294-295
    t.vx = nvx; t.vy = nvy; t.x = npx; t.y = npy;
    t.age = a + 1.0f / 60.0f;                                 // Part:279
    t.life = l;
    return true;
}
__global__ __launch_bounds__(256) void k_particles(GK g, const float *__restrict__ u, const float *__restrict__ v,
                                                   fs_particle_params pp, float eff, long n, float *__restrict__ px,
                                                   float *__restrict__ py, float *__restrict__ vx,
                                                   float *__restrict__ vy, float *__restrict__ life,
                                                   const float *__restrict__ mass, float *__restrict__ age)
{
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    TracerState t{px[i], py[i], vx[i], vy[i], life[i], age[i]};
    if (!tracer_step(g, u, v, pp, eff, mass[i], t)) return;
    vx[i] = t.vx; vy[i] = t.vy; px[i] = t.x; py[i] = t.y;
    age[i] = t.age;
    life[i] = t.life;
}
// (Four tracers per lane with 16-byte accesses was tried and is slower, 27 vs 21 us for 1,048,576 sorted tracers: the j-th
// lookups of a wave's lanes are then four tracers apart and touch four times the lines per instruction.)

// tracer line vertices, vertex shader particleVertex Shaders:446-476: (pos - vel*scale, pos) in NDC, alpha = life
__global__ __launch_bounds__(256) void k_particle_lines(long n, const float *__restrict__ px, const float *__restrict__ py,
                                                        const float *__restrict__ vx, const float *__restrict__ vy,
                                                        const float *__restrict__ life, float vw, float vh, float scale,
                                                        float4 *__restrict__ out_xy, float *__restrict__ out_alpha)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = px[i], y = py[i];
    const float sx = x - vx[i] * scale, sy = y - vy[i] * scale;            // :462
    float4 o;
    o.x = (sx / vw) * 2.0f - 1.0f;                                          // :468
    o.y = -((sy / vh) * 2.0f - 1.0f);                                       // :469
    o.z = (x / vw) * 2.0f - 1.0f;                                           // :464
    o.w = -((y / vh) * 2.0f - 1.0f);
    out_xy[i] = o;
    out_alpha[i] = life[i];                                                 // :472
}

// ---------------------------------------------------------------------------
// Tracers on row slabs (no counterpart in the reference): a tracer lives on the rank
// that owns the grid row its velocity lookup reads (Solver:651).  Before each update the tracers are
// re-homed: count per destination, then scatter into a "keep" set (partner arrays) and per-destination
// outboxes.  One atomic per destination and WAVE, not per tracer.
// ---------------------------------------------------------------------------
struct SlabSplit {
    int world, rank, base, rem;   // rows per rank = base (+1 for the first rem ranks), see slab_rows()
};
__device__ __forceinline__ int tracer_dest(const fs_particle_params &pp, const GK &g, const SlabSplit sp, float y, float l)
{
    if (l == 0.0f) return sp.rank;                            // dead slots never move
    const float invH = 1.0f / pp.windowH;                     // the same operations as k_particles
    const float gh = (float)g.H;
    const float ny = y * invH;
    const int cy = (int)fmaxf(0.0f, fminf(gh - 1.0f, ny * gh));   // Solver:651
    const int big = sp.rem * (sp.base + 1);
    return cy < big ? cy / (sp.base + 1) : sp.rem + (cy - big) / sp.base;
}
// slot for this lane among the lanes of the wave with the same destination: base from one atomicAdd
// by the first such lane (counting only when cursors is a counter array: same code)
__device__ __forceinline__ int wave_slot(int *cursor, int dest, bool valid)
{
    const int lane = threadIdx.x & 63;
    int slot = -1;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int d = __shfl(dest, leader);
        const unsigned long long same = __ballot(valid && dest == d);
        int base = 0;
        if (lane == leader) base = atomicAdd(&cursor[d], __popcll(same));
        base = __shfl(base, leader);
        if (valid && dest == d) slot = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}
__global__ __launch_bounds__(256) void k_tracer_count(GK g, fs_particle_params pp, SlabSplit sp, long n,
                                                      const float *__restrict__ py, const float *__restrict__ life,
                                                      int *__restrict__ counts)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const int dest = valid ? tracer_dest(pp, g, sp, py[i], life[i]) : 0;
    (void)wave_slot(counts, dest, valid);
}
struct TracerArrays {
    float *a[7];
};
__global__ __launch_bounds__(256) void k_tracer_scatter(GK g, fs_particle_params pp, SlabSplit sp, long n,
                                                        TracerArrays in, TracerArrays keep, TracerArrays outbox,
                                                        int *__restrict__ cursors)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const int dest = valid ? tracer_dest(pp, g, sp, in.a[1][i], in.a[4][i]) : 0;
    const int slot = wave_slot(cursors, dest, valid);   // cursors start at the destination's outbox offset (0 for me)
    if (!valid) return;
    const TracerArrays &to = dest == sp.rank ? keep : outbox;
#pragma unroll
    for (int f = 0; f < 7; ++f) to.a[f][slot] = in.a[f][i];
}

// ---------------------------------------------------------------------------
// Spawning, ParticleSystem.addParticles Part:186-231.  The reference's array holds the live tracers in age order and
// appends with removeFirst() at the cap, i.e. it keeps the last `cap` of (live + new).  Here the arrays keep dead
// slots until the next spawn, which (1) counts the live tracers per block of 256, (2) scans the counts, (3) copies the
// live tracers, in order, minus the `drop` oldest, to the partner arrays and (4) writes the new ones behind them.
// ---------------------------------------------------------------------------
// A slot is dead when its life is zero.  fs_particles_spawn's fast path cannot compact (that needs the live count on the
// host), so it marks the slots that are dead AT THAT MOMENT as removed -- life = -0.0f, still "== 0" to every kernel -- and
// the compaction that is owed later drops exactly those (removed_only), not the tracers that have died since.
__device__ __forceinline__ bool tracer_kept(float life, bool removed_only)
{
    return removed_only ? __float_as_uint(life) != 0x80000000u : life != 0.0f;
}
__global__ __launch_bounds__(256) void k_tracer_mark_removed(long n, float *__restrict__ life)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && life[i] == 0.0f) life[i] = -0.0f;
}
__global__ __launch_bounds__(256) void k_tracer_live_count(long n, const float *__restrict__ life, int *__restrict__ block_sums,
                                                           bool removed_only)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n && tracer_kept(life[i], removed_only);
    const int cnt = __popcll(__ballot(live));
    __shared__ int w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
// one wave: exclusive scan of sums[0..nb) in place, total to sums[nb]
__global__ __launch_bounds__(64) void k_tracer_scan(int nb, int *sums)
{
    const int lane = threadIdx.x;
    const int seg = (nb + 63) / 64;
    const int lo = min(lane * seg, nb), hi = min(lo + seg, nb);
    int mine = 0;
    for (int i = lo; i < hi; ++i) mine += sums[i];
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    int run = incl - mine;
    for (int i = lo; i < hi; ++i) { const int t = sums[i]; sums[i] = run; run += t; }
    if (lane == 63) sums[nb] = incl;
}
__global__ __launch_bounds__(256) void k_tracer_compact(long n, TracerArrays in, TracerArrays out, const int *__restrict__ block_off,
                                                        long drop, bool removed_only)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n && tracer_kept(in.a[4][i], removed_only);
    const unsigned long long b = __ballot(live);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ int w[4];
    if (lane == 0) w[wv] = __popcll(b);
    __syncthreads();
    int base = block_off[blockIdx.x];
    for (int k = 0; k < wv; ++k) base += w[k];
    const long rank = (long)base + __popcll(b & ((1ull << lane) - 1ull));   // position among the live tracers, age order kept
    if (live && rank >= drop) {
#pragma unroll
        for (int f = 0; f < 7; ++f) out.a[f][rank - drop] = in.a[f][i];
    }
}
__device__ __forceinline__ float spawn_draw(unsigned &x, float lo, float hi)
{
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const float r01 = (float)(x >> 8) * (1.0f / 16777216.0f);
    return lo + (hi - lo) * r01;
}
// new tracer j (j0 <= j < count) -> slot first + (j - j0); its four draws from an xorshift32 stream of its own
__global__ __launch_bounds__(256) void k_tracer_spawn(long first, long j0, long count, TracerArrays out, float x, float y,
                                                      float radius, unsigned seed, int *__restrict__ perm)
{
    const long j = j0 + (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= count) return;
    unsigned st = seed ^ ((unsigned)(j + 1) * 0x9E3779B9u);
    if (!st) st = 1u;
    const float ox = spawn_draw(st, -radius, radius);      // Part:202-205
    const float oy = spawn_draw(st, -radius, radius);
    const float alpha = spawn_draw(st, 0.3f, 1.0f);        // Part:211
    const float ms = spawn_draw(st, 0.1f, 1.0f);           // Part:212
    const long s = first + (j - j0);
    out.a[0][s] = x + ox; out.a[1][s] = y + oy;            // Part:218
    out.a[2][s] = 0.0f; out.a[3][s] = 0.0f;                // Part:208
    out.a[4][s] = alpha; out.a[5][s] = ms; out.a[6][s] = 0.0f;   // Part:221-223
    if (perm) perm[s] = (int)s;   // appended behind sorted arrays: the new tracer's slot is its position
}

// ---------------------------------------------------------------------------
// Tracer order (no counterpart in the reference, whose CPU loop has no memory system to please).  k_particles gathers
// u and v at each tracer's cell: with the tracers in arrival order every lane of a wave pulls its own 128-byte line,
// twice (1,048,576 tracers at 3072^2: ~270 MB of lines for 8 MB of values, 46 us).  Every few dozen frames the arrays
// are therefore counting-sorted by where their lookup reads -- key = the block of 8 grid rows x 32 cells (one cache line
// wide) that holds the cell of Solver:650-651 -- so that a wave's 64 lookups share a handful of lines (at one tracer per
// nine cells about 18 per plane whichever way they are grouped; finer bins only make the scan longer).  Tracers drift slowly (their coupling
// to the fluid decays within a third of a second, Part:259), so the order keeps for many frames.  `perm` remembers the
// slot every tracer came from; every call that exposes slot order (get, spawn, line vertices) restores it first, so
// the order is invisible outside, and the per-tracer arithmetic is untouched: same bits.
// ---------------------------------------------------------------------------
struct TracerBins {
    int nbx, shift, nbins;   // column blocks per row, log2(cells per block), row blocks x nbx (+ 1 bin for dead slots)
};
__device__ __forceinline__ int tracer_bin(const GK &g, const fs_particle_params &pp, const TracerBins tb, float x, float y, float l)
{
    if (l == 0.0f) return tb.nbins - 1;                              // dead slots: out of the way, at the end
    const float invW = 1.0f / pp.windowW, invH = 1.0f / pp.windowH;  // the same operations as k_particles
    const float gw = (float)g.W, gh = (float)g.H;
    const int cx = (int)fmaxf(0.0f, fminf(gw - 1.0f, (x * invW) * gw));
    const int cy = (int)fmaxf(0.0f, fminf(gh - 1.0f, (y * invH) * gh));
    return (cy >> 3) * tb.nbx + (cx >> tb.shift);
}
__global__ __launch_bounds__(256) void k_tracer_bin_count(GK g, fs_particle_params pp, TracerBins tb, long n,
                                                          const float *__restrict__ px, const float *__restrict__ py,
                                                          const float *__restrict__ life, int *__restrict__ keys,
                                                          int *__restrict__ hist)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = tracer_bin(g, pp, tb, px[i], py[i], life[i]);
    keys[i] = b;
    atomicAdd(&hist[b], 1);
}
// exclusive scan of hist[0..nb) in place by ONE workgroup of 1024 threads, a coalesced tile of 4096 counters per trip
// (nb is a few ten thousand at 3072^2, a million at 16384^2; the sort runs every few dozen frames).  hist holds nb
// rounded up to a multiple of 4 ints.
__global__ __launch_bounds__(1024) void k_tracer_bin_scan(int nb, int *hist)
{
    __shared__ int wave_sum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 4096) {
        const int i = base + 4 * t;
        int4 v = make_int4(0, 0, 0, 0);
        if (i < nb) v = *reinterpret_cast<const int4 *>(hist + i);   // (the tail beyond nb is zero-filled padding)
        const int mine = v.x + v.y + v.z + v.w;
        int incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < wv; ++k) before += wave_sum[k];
        int run = before + incl - mine;
        if (i < nb) {
            int4 o;
            o.x = run; run += v.x;
            o.y = run; run += v.y;
            o.z = run; run += v.z;
            o.w = run;
            *reinterpret_cast<int4 *>(hist + i) = o;
        }
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
}
// perm_in == nullptr: the arrays are in slot order
__global__ __launch_bounds__(256) void k_tracer_bin_scatter(long n, TracerArrays in, TracerArrays out, const int *__restrict__ keys,
                                                            int *__restrict__ cursor, const int *__restrict__ perm_in,
                                                            int *__restrict__ perm_out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int slot = atomicAdd(&cursor[keys[i]], 1);
#pragma unroll
    for (int f = 0; f < 7; ++f) out.a[f][slot] = in.a[f][i];
    perm_out[slot] = perm_in ? perm_in[i] : (int)i;
}
__global__ __launch_bounds__(256) void k_tracer_unsort(long n, TracerArrays in, TracerArrays out, const int *__restrict__ perm)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int slot = perm[i];
#pragma unroll
    for (int f = 0; f < 7; ++f) out.a[f][slot] = in.a[f][i];
}

// max |plane| over rows [yb,ye): wave shuffle reduce, one atomicMax per wave (non-negative
// floats order like their bit patterns).
__global__ __launch_bounds__(256) void k_max_abs(GK g, int yb, int ye, const float *__restrict__ p, unsigned *out)
{
    float m = 0.0f;
    int nq = g.W / 4;  // whole quads; tail handled below
    for (int y = yb + blockIdx.x; y < ye; y += gridDim.x) {
        const float *row = p + row_off(g, y);
        for (int qx = threadIdx.x; qx < nq; qx += 256) {
            float4 a = ld4(row + 4 * qx);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
        for (int x = nq * 4 + threadIdx.x; x < g.W; x += 256) m = fmaxf(m, fabsf(row[x]));
    }
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomic_max_if_larger(out, __float_as_uint(m));
}

static inline dim3 quad_grid(const fs_grid *g, int nplanes)
{
    int qx = g->pitch / 4;
    const unsigned tile_rows = (unsigned)((g->ye - g->yb + BY - 1) / BY);
    return dim3((qx + BX - 1) / BX, tile_rows < 8u * XCD_BAND ? tile_rows : xcd_pad_rows(tile_rows, XCD_BAND), nplanes);
}
static const dim3 kBlock(BX, BY, 1);

long tracer_blocks(long n);
// tracer order helpers for the context (context.hip)
int tracer_bins(const fs_grid *g, int *nbx, int *shift)
{
    // blocks of 8 rows x 32 cells (one 128-byte line wide) unless that makes more than ~4M bins
    int sh = 5;
    const long rb = (g->height + 7) / 8;
    while (rb * (long)((g->width + (1 << sh) - 1) >> sh) > (4L << 20)) ++sh;
    *shift = sh;
    *nbx = (g->width + (1 << sh) - 1) >> sh;
    return (int)(rb * *nbx + 1);
}
int tracer_sort(const fs_grid *g, const fs_particle_params *pp, long n, float *const in[7], float *const out[7], int *keys, int *hist,
                const int *perm_in, int *perm_out, hipStream_t st)
{
    TracerBins tb;
    tb.nbins = tracer_bins(g, &tb.nbx, &tb.shift);
    TracerArrays a, b;
    for (int f = 0; f < 7; ++f) { a.a[f] = in[f]; b.a[f] = out[f]; }
    FS_HIP(hipMemsetAsync(hist, 0, (size_t)((tb.nbins + 3) & ~3) * sizeof(int), st));
    const unsigned blocks = (unsigned)tracer_blocks(n);
    hipLaunchKernelGGL(k_tracer_bin_count, dim3(blocks), dim3(256), 0, st, make_gk(g), *pp, tb, n, in[0], in[1], in[4], keys, hist);
    hipLaunchKernelGGL(k_tracer_bin_scan, dim3(1), dim3(1024), 0, st, tb.nbins, hist);
    hipLaunchKernelGGL(k_tracer_bin_scatter, dim3(blocks), dim3(256), 0, st, n, a, b, keys, hist, perm_in, perm_out);
    FS_LAUNCH_CHECK("tracer sort");
    return FS_OK;
}
int tracer_unsort(long n, float *const in[7], float *const out[7], const int *perm, hipStream_t st)
{
    TracerArrays a, b;
    for (int f = 0; f < 7; ++f) { a.a[f] = in[f]; b.a[f] = out[f]; }
    hipLaunchKernelGGL(k_tracer_unsort, dim3((unsigned)tracer_blocks(n)), dim3(256), 0, st, n, a, b, perm);
    FS_LAUNCH_CHECK("k_tracer_unsort");
    return FS_OK;
}

// spawn helpers for the context (context.hip); sums holds blocks + 1 ints
long tracer_blocks(long n) { return (n + 255) / 256; }
int tracer_mark_removed(long n, float *life, hipStream_t st)
{
    if (n <= 0) return FS_OK;
    hipLaunchKernelGGL(k_tracer_mark_removed, dim3((unsigned)tracer_blocks(n)), dim3(256), 0, st, n, life);
    FS_LAUNCH_CHECK("k_tracer_mark_removed");
    return FS_OK;
}
int tracer_count_live(long n, const float *life, int *sums, bool removed_only, hipStream_t st)
{
    const long nb = tracer_blocks(n);
    hipLaunchKernelGGL(k_tracer_live_count, dim3((unsigned)nb), dim3(256), 0, st, n, life, sums, removed_only);
    hipLaunchKernelGGL(k_tracer_scan, dim3(1), dim3(64), 0, st, (int)nb, sums);
    FS_LAUNCH_CHECK("k_tracer_live_count / k_tracer_scan");
    return FS_OK;
}
int tracer_compact(long n, float *const in[7], float *const out[7], const int *sums, long drop, bool removed_only, hipStream_t st)
{
    TracerArrays a, b;
    for (int f = 0; f < 7; ++f) { a.a[f] = in[f]; b.a[f] = out[f]; }
    hipLaunchKernelGGL(k_tracer_compact, dim3((unsigned)tracer_blocks(n)), dim3(256), 0, st, n, a, b, sums, drop, removed_only);
    FS_LAUNCH_CHECK("k_tracer_compact");
    return FS_OK;
}
int tracer_spawn(long first, long j0, long count, float *const out[7], float x, float y, float radius, unsigned seed, int *perm, hipStream_t st)
{
    if (count <= j0) return FS_OK;
    TracerArrays b;
    for (int f = 0; f < 7; ++f) b.a[f] = out[f];
    hipLaunchKernelGGL(k_tracer_spawn, dim3((unsigned)((count - j0 + 255) / 256)), dim3(256), 0, st, first, j0, count, b, x, y, radius, seed, perm);
    FS_LAUNCH_CHECK("k_tracer_spawn");
    return FS_OK;
}

}  // namespace fs

using namespace fs;

#define STREAM ((hipStream_t)stream)
#define REQUIRE(cond, msg)                      \
    do {                                        \
        if (!(cond)) { fs::set_error("%s: %s", __func__, msg); return FS_ERR_INVALID; } \
    } while (0)
#define CHECK_GRID(g)                 \
    do {                              \
        int rc_ = fs::check_grid(g);  \
        if (rc_) return rc_;          \
    } while (0)
#define EMPTY_OK(g) \
    if ((g)->ye <= (g)->yb) return FS_OK

extern "C" {

int fs_k_divergence(const fs_grid *g, const float *u, const float *v, float *div, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && div, "null pointer");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_divergence, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, u, v, div);
    FS_LAUNCH_CHECK("k_divergence");
    return FS_OK;
}

int fs_k_calculateVorticity(const fs_grid *g, const float *u, const float *v, float *vort, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && vort, "null pointer");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_vorticity, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, u, v, vort);
    FS_LAUNCH_CHECK("k_vorticity");
    return FS_OK;
}

int fs_k_subtractGradient(const fs_grid *g, const float *p, float *u, float *v, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(p && u && v, "null pointer");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_subtract_gradient<false>, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, p, u, v,
                       (unsigned *)nullptr);
    FS_LAUNCH_CHECK("k_subtract_gradient");
    return FS_OK;
}

int fs_k_subtractGradient_cfl(const fs_grid *g, const float *p, float *u, float *v, float *cfl_dev, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(p && u && v && cfl_dev, "null pointer");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_subtract_gradient<true>, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, p, u, v,
                       reinterpret_cast<unsigned *>(cfl_dev));
    FS_LAUNCH_CHECK("k_subtract_gradient_cfl");
    return FS_OK;
}

int fs_k_applyVorticity(const fs_grid *g, const float *vort, float *u, float *v, float strength, float deltaTime,
                        void *stream)
{
    CHECK_GRID(g);
    REQUIRE(vort && u && v, "null pointer");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_apply_vorticity, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, vort, u, v,
                       strength, deltaTime);
    FS_LAUNCH_CHECK("k_apply_vorticity");
    return FS_OK;
}

int fs_k_vorticity_apply(const fs_grid *g, const float *u, const float *v, float *u_out, float *v_out, float strength,
                         float deltaTime, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && u_out && v_out, "null pointer");
    REQUIRE(u != u_out && v != v_out, "the confinement force needs the old velocity of the neighbours: not in place");
    EMPTY_OK(g);
    if (g->width >= 512) {   // wide enough for 256-column strips: the marching form (32.1 vs 50.0 us at 3072^2, 918 vs 1234 us at 16384^2)
        constexpr int K = 8;   // 5, 6, 8 rows per wave tie; 4 and 16 lose (run-in rows / too few waves)
        unsigned gy = (g->ye - g->yb + 4 * K - 1) / (4 * K);
        const int xm = gy >= 512u ? 8 : gy >= 32u ? 4 : 0;   // XCD bands of 8 / 4 block rows (profiles/r02_f_vorticity_march.log)
        if (xm) gy = xcd_pad_rows(gy, xm);
        const dim3 grid((g->width + 247) / 248, gy);
        if (xm == 8) hipLaunchKernelGGL((k_vorticity_apply_march<K, 8>), grid, dim3(256), 0, STREAM, make_gk(g), g->yb, g->ye, u, v, u_out, v_out, strength, deltaTime);
        else if (xm == 4) hipLaunchKernelGGL((k_vorticity_apply_march<K, 4>), grid, dim3(256), 0, STREAM, make_gk(g), g->yb, g->ye, u, v, u_out, v_out, strength, deltaTime);
        else hipLaunchKernelGGL((k_vorticity_apply_march<K, 0>), grid, dim3(256), 0, STREAM, make_gk(g), g->yb, g->ye, u, v, u_out, v_out, strength, deltaTime);
        FS_LAUNCH_CHECK("k_vorticity_apply_march");
        return FS_OK;
    }
    hipLaunchKernelGGL(k_vorticity_apply, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, u, v, u_out, v_out,
                       strength, deltaTime);
    FS_LAUNCH_CHECK("k_vorticity_apply");
    return FS_OK;
}

int fs_k_fadeDensity(const fs_grid *g, float *d, int nplanes, size_t plane_stride, float fadeRate, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(d && nplanes >= 1, "bad arguments");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_fade, quad_grid(g, nplanes), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, d, plane_stride,
                       1.0f - fadeRate);  // Shaders:331
    FS_LAUNCH_CHECK("k_fade");
    return FS_OK;
}

int fs_k_clear(const fs_grid *g, float *t, int nplanes, size_t plane_stride, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(t && nplanes >= 1, "bad arguments");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_clear, quad_grid(g, nplanes), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, t, plane_stride);
    FS_LAUNCH_CHECK("k_clear");
    return FS_OK;
}

int fs_k_renderDensity(const fs_grid *g, const float *d, size_t d_stride, int d_planes, float *out4,
                       size_t out_stride, float brightness, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(d && out4 && d_planes >= 1 && d_planes <= 3, "bad arguments");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_render_density, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, d, d_stride,
                       d_planes, out4, out_stride, brightness);
    FS_LAUNCH_CHECK("k_render_density");
    return FS_OK;
}

int fs_k_renderVelocity(const fs_grid *g, const float *u, const float *v, float *out4, size_t out_stride,
                        float scale, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && out4, "null pointer");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_render_velocity<false>, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, u, v,
                       out4, out_stride, scale);
    FS_LAUNCH_CHECK("k_render_velocity");
    return FS_OK;
}

int fs_k_renderSpeed(const fs_grid *g, const float *u, const float *v, float *out4, size_t out_stride, float scale,
                     void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && out4, "null pointer");
    EMPTY_OK(g);
    hipLaunchKernelGGL(k_render_velocity<true>, quad_grid(g, 1), kBlock, 0, STREAM, make_gk(g), g->yb, g->ye, u, v,
                       out4, out_stride, scale);
    FS_LAUNCH_CHECK("k_render_speed");
    return FS_OK;
}

static int launch_advect(const fs_grid *g, const float *u, const float *v, const float *in, float *out, int nplanes,
                         size_t plane_stride, float deltaTime, int *err_flag, bool fade, float fadeRate, void *stream,
                         int hole_y = 0, int hole_n = 0, int stream_result = 0)
{
    CHECK_GRID(g);
    REQUIRE(u && v && in && out, "null pointer");
    REQUIRE(nplanes >= 1 && nplanes <= 3, "nplanes must be 1..3");
    REQUIRE(in != out, "advect cannot run in place");
    EMPTY_OK(g);
    REQUIRE(hole_n >= 0 && (hole_n == 0 || (hole_y >= g->yb && hole_y + hole_n <= g->ye)), "hole outside the rows of the launch");
    fs_grid launched = *g;
    launched.ye = g->ye - hole_n;
    EMPTY_OK(&launched);
    dim3 grid = quad_grid(&launched, 1);
    GK gk = make_gk(g);
    if (stream_result) gk.nt |= 1;   // (the caller knows that nothing reads the result soon: slab.hip dye_result_streams)
    const float k = 1.0f - fadeRate;  // Shaders:331
#define ADV(NP, FD) hipLaunchKernelGGL((k_advect<NP, FD>), grid, kBlock, 0, STREAM, gk, g->yb, g->ye, hole_y, hole_n, u, v, in, out, plane_stride, deltaTime, err_flag, k)
    if (fade) {
        switch (nplanes) { case 1: ADV(1, true); break; case 2: ADV(2, true); break; default: ADV(3, true); break; }
    } else {
        switch (nplanes) { case 1: ADV(1, false); break; case 2: ADV(2, false); break; default: ADV(3, false); break; }
    }
#undef ADV
    FS_LAUNCH_CHECK("k_advect");
    return FS_OK;
}

int fs_k_advect(const fs_grid *g, const float *u, const float *v, const float *in, float *out, int nplanes,
                size_t plane_stride, float deltaTime, int *err_flag, void *stream)
{
    return launch_advect(g, u, v, in, out, nplanes, plane_stride, deltaTime, err_flag, false, 0.0f, stream);
}

int fs_k_advect_fade(const fs_grid *g, const float *u, const float *v, const float *in, float *out, int nplanes,
                     size_t plane_stride, float deltaTime, float fadeRate, int *err_flag, void *stream)
{
    return launch_advect(g, u, v, in, out, nplanes, plane_stride, deltaTime, err_flag, true, fadeRate, stream);
}

int fs_k_advect_both(const fs_grid *g, const float *u, const float *v, float *u_out, float *v_out, const float *dye,
                     float *dye_out, size_t dye_stride, float deltaTime, int fade, float fadeRate, int *err_flag,
                     void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && u_out && v_out && dye && dye_out, "null pointer");
    REQUIRE(u != u_out && v != v_out && dye != dye_out, "advect cannot run in place");
    EMPTY_OK(g);
    const dim3 grid = quad_grid(g, 1);
    const GK gk = make_gk(g);
    const float k = 1.0f - fadeRate;  // Shaders:331
    if (fade)
        hipLaunchKernelGGL(k_advect_both<true>, grid, kBlock, 0, STREAM, gk, g->yb, g->ye, u, v, u_out, v_out, dye, dye_out,
                           dye_stride, deltaTime, err_flag, k);
    else
        hipLaunchKernelGGL(k_advect_both<false>, grid, kBlock, 0, STREAM, gk, g->yb, g->ye, u, v, u_out, v_out, dye, dye_out,
                           dye_stride, deltaTime, err_flag, k);
    FS_LAUNCH_CHECK("k_advect_both");
    return FS_OK;
}

// bounding box of the cells with dist < radius, clipped to the rows the launch may touch; false: no such cell
static bool splat_box(const fs_grid *g, float pos_x, float pos_y, float radius, int box[4])
{
    if (!(radius > 0.0f) || g->ye <= g->yb) return false;  // dist < radius can never hold
    // every cell with dist < radius lies in [pos-radius, pos+radius]
    double lo_x = floor((double)pos_x - radius), hi_x = ceil((double)pos_x + radius);
    double lo_y = floor((double)pos_y - radius), hi_y = ceil((double)pos_y + radius);
    if (hi_x < 0 || hi_y < g->yb || lo_x > g->width - 1 || lo_y > g->ye - 1) return false;
    box[0] = (int)fmax(lo_x, 0.0); box[2] = (int)fmin(hi_x, (double)g->width - 1);
    box[1] = (int)fmax(lo_y, (double)g->yb); box[3] = (int)fmin(hi_y, (double)g->ye - 1);
    return box[2] >= box[0] && box[3] >= box[1];
}

int fs_k_addForces(const fs_grid *g, float *u, float *v, float *r, float *gch, float *b, float pos_x, float pos_y,
                   float force_x, float force_y, float color_r, float color_g, float color_b, float radius,
                   float deltaTime, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && r && gch && b, "null pointer");
    REQUIRE(isfinite(pos_x) && isfinite(pos_y) && isfinite(radius), "non-finite splat geometry");
    EMPTY_OK(g);
    int box[4];
    if (!splat_box(g, pos_x, pos_y, radius, box)) return FS_OK;
    dim3 grid((box[2] - box[0] + 64) / 64, (box[3] - box[1] + 4) / 4, 1);
    hipLaunchKernelGGL(k_add_forces, grid, kBlock, 0, STREAM, make_gk(g), box[0], box[1], box[2], box[3], u, v, r, gch, b, pos_x,
                       pos_y, force_x, force_y, color_r, color_g, color_b, radius, deltaTime);
    FS_LAUNCH_CHECK("k_add_forces");
    return FS_OK;
}

}  // extern "C"

// rows [g->yb, hole_y0) and [hole_y1, g->ye) in one launch (slab.hip: top and bottom rim of the split advection; hole_y0 ==
// hole_y1: all rows); stream_result: the result leaves through streaming stores whatever the size rule says
namespace fs {
int advect_around(const fs_grid *g, int hole_y0, int hole_y1, const float *u, const float *v, const float *in, float *out, int nplanes,
                  size_t plane_stride, float deltaTime, int fade, float fadeRate, int *err_flag, void *stream, int stream_result)
{
    return launch_advect(g, u, v, in, out, nplanes, plane_stride, deltaTime, err_flag, fade != 0, fadeRate, stream, hole_y0,
                         hole_y1 - hole_y0, stream_result);
}
}  // namespace fs


// The splat as a NODE of a frame's graph (context.hip): its argument block and launch geometry, set afresh before every
// replay.  A splat that touches no cell keeps its node as one workgroup over an empty box.
namespace fs {
void splat_node_params(SplatNode &sn, const fs_grid *g, float *u, float *v, float *r, float *gch, float *b, const float s[9])
{
    sn.g = make_gk(g);
    sn.u = u; sn.v = v; sn.r = r; sn.gch = gch; sn.b = b;
    for (int i = 0; i < 9; ++i) sn.f[i] = s[i];   // px py fx fy cr cg cb radius dt
    unsigned gx = 1, gy = 1;
    if (isfinite(s[0]) && isfinite(s[1]) && isfinite(s[7]) && splat_box(g, s[0], s[1], s[7], sn.box)) {
        gx = (unsigned)(sn.box[2] - sn.box[0] + 64) / 64;
        gy = (unsigned)(sn.box[3] - sn.box[1] + 4) / 4;
    } else {
        sn.box[0] = sn.box[1] = 0; sn.box[2] = sn.box[3] = -1;   // every thread leaves at once
    }
    void *args[19] = {&sn.g, &sn.box[0], &sn.box[1], &sn.box[2], &sn.box[3], &sn.u, &sn.v, &sn.r, &sn.gch, &sn.b,
                      &sn.f[0], &sn.f[1], &sn.f[2], &sn.f[3], &sn.f[4], &sn.f[5], &sn.f[6], &sn.f[7], &sn.f[8]};
    for (int i = 0; i < 19; ++i) sn.args[i] = args[i];
    sn.np = hipKernelNodeParams{};
    sn.np.func = reinterpret_cast<void *>(&k_add_forces);
    sn.np.gridDim = dim3(gx, gy, 1);
    sn.np.blockDim = kBlock;
    sn.np.sharedMemBytes = 0;
    sn.np.kernelParams = sn.args;
    sn.np.extra = nullptr;
}
}  // namespace fs

extern "C" {

int fs_k_particles_update(const fs_grid *g, const float *u, const float *v, const fs_particle_params *pp, long n,
                          float *px, float *py, float *vx, float *vy, float *life, const float *mass, float *age,
                          void *stream)
{
    CHECK_GRID(g);
    REQUIRE(u && v && pp, "null pointer");
    REQUIRE(n >= 0, "negative particle count");
    if (n == 0) return FS_OK;
    REQUIRE(px && py && vx && vy && life && mass && age, "null particle array");
    // a slab passes only tracers whose lookup row it owns (fs_tracer_rehome); a lookup outside the buffer is
    // clamped into it, never out of bounds
    // Part:262-268
    float eff = pp->linkToViscosity ? 0.1f + (pp->viscosity / 0.01f) * 0.9f : pp->fluidForce;
    unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_particles, dim3(blocks), dim3(256), 0, STREAM, make_gk(g), u, v, *pp, eff, n, px, py, vx,
                       vy, life, mass, age);
    FS_LAUNCH_CHECK("k_particles");
    return FS_OK;
}

int fs_k_particle_lines(long n, const float *px, const float *py, const float *vx, const float *vy, const float *life,
                        float viewportW, float viewportH, float particleScale, float *out_xy4, float *out_alpha,
                        void *stream)
{
    REQUIRE(n >= 0, "negative particle count");
    if (n == 0) return FS_OK;
    REQUIRE(px && py && vx && vy && life && out_xy4 && out_alpha, "null pointer");
    REQUIRE(((uintptr_t)out_xy4 & 15) == 0, "out_xy4 must be 16-byte aligned");
    hipLaunchKernelGGL(k_particle_lines, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, n, px, py, vx, vy, life,
                       viewportW, viewportH, particleScale, reinterpret_cast<float4 *>(out_xy4), out_alpha);
    FS_LAUNCH_CHECK("k_particle_lines");
    return FS_OK;
}

static fs::SlabSplit split_of(const fs_grid *g, int world, int rank)
{
    return fs::SlabSplit{world, rank, g->height / world, g->height % world};
}
static fs::TracerArrays arrays_of(float *const a[7])
{
    fs::TracerArrays t;
    for (int i = 0; i < 7; ++i) t.a[i] = a[i];
    return t;
}

int fs_k_tracer_count(const fs_grid *g, const fs_particle_params *pp, int world, int rank, long n, const float *py,
                      const float *life, int *counts_dev, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(pp && counts_dev && world >= 1 && rank >= 0 && rank < world && g->height / world >= 1, "bad arguments");
    REQUIRE(n >= 0, "negative particle count");
    if (n == 0) return FS_OK;
    REQUIRE(py && life, "null particle array");
    hipLaunchKernelGGL(k_tracer_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, make_gk(g), *pp,
                       split_of(g, world, rank), n, py, life, counts_dev);
    FS_LAUNCH_CHECK("k_tracer_count");
    return FS_OK;
}

int fs_k_tracer_scatter(const fs_grid *g, const fs_particle_params *pp, int world, int rank, long n,
                        float *const in[7], float *const keep[7], float *const outbox[7], int *cursors_dev,
                        void *stream)
{
    CHECK_GRID(g);
    REQUIRE(pp && cursors_dev && in && keep && outbox && world >= 1 && rank >= 0 && rank < world, "bad arguments");
    REQUIRE(n >= 0, "negative particle count");
    if (n == 0) return FS_OK;
    for (int i = 0; i < 7; ++i) REQUIRE(in[i] && keep[i] && outbox[i], "null particle array");
    hipLaunchKernelGGL(k_tracer_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, make_gk(g), *pp,
                       split_of(g, world, rank), n, arrays_of(in), arrays_of(keep), arrays_of(outbox), cursors_dev);
    FS_LAUNCH_CHECK("k_tracer_scatter");
    return FS_OK;
}

int fs_k_max_abs(const fs_grid *g, const float *plane, float *out_dev, void *stream)
{
    CHECK_GRID(g);
    REQUIRE(plane && out_dev, "null pointer");
    EMPTY_OK(g);
    int blocks = g->ye - g->yb;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_max_abs, dim3(blocks), dim3(256), 0, STREAM, make_gk(g), g->yb, g->ye, plane,
                       reinterpret_cast<unsigned *>(out_dev));
    FS_LAUNCH_CHECK("k_max_abs");
    return FS_OK;
}

}  // extern "C"
