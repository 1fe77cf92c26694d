// jacobi.hip -- temporally blocked Jacobi sweeps (pressureSolve / diffuse) for gfx950.
//
// The two hot loops of update() (Solver:506-517 diffuse, :533-541 pressureSolve) are N
// back-to-back 5-point sweeps.  One sweep is pure HBM streaming (12 B/cell pressure,
// 8 B/cell/plane diffuse); T sweeps fused in one launch move the same bytes once, so
// the kernel turns from HBM-bound into VALU-bound.  The per-sweep arithmetic is
// unchanged: every cell value is produced by exactly the reference's expression,
// evaluated in the same order, so T fused sweeps equal T single sweeps bit for bit.
//
// CDNA4 mapping (register-resident tile, no LDS ping-pong):
//   * a wave (64 lanes) owns a 256-column strip: lane l holds columns 4l..4l+3 of R
//     rows in VGPRs (float4 per row) -> global loads/stores are 16 B/lane, 1 KiB per
//     wave instruction, rows 256-byte aligned;
//   * left/right neighbours inside a lane are registers; across lanes they are a DPP
//     operand (wave_shr:1 / wave_shl:1 folded into the v_add), no LDS, no shuffle;
//   * top/bottom neighbours inside a lane's R rows are registers; the two edge rows of
//     each wave are exchanged with the waves above/below through a small
//     double-buffered LDS array (2 x 16 B per lane per sweep), one barrier per sweep;
//   * NW waves are stacked vertically: tile = 256 x (NW*R) cells; after T sweeps the
//     inner (256-2*HX) x (NW*R-2*T) cells are exact and are stored, the rim is the
//     redundantly computed halo (HX = T rounded up to 4 keeps stores 16-byte aligned).
// Boundary cells (and cells outside the domain) do not depend on their neighbours:
// pressure boundary = 0 (Shaders:201-204), diffuse boundary = copy (Shaders:91-94), so
// inside the tile they are overwritten with that value every sweep.
#include "common.h"

namespace fs {

enum { OP_PRESSURE = 0, OP_DIFFUSE = 1 };

#ifdef FS_PROBE
// Timeline of the workgroups of the LAST k_jacobi_block launch (probe build only, scripts/jacobi_timeline.py):
// per workgroup 8 words -- 100 MHz clock at entry / tile loaded / sweeps done / exit, HW_ID, XCC_ID.
#define FS_TL_MAX 8192
__device__ unsigned long long g_timeline[8 * FS_TL_MAX];
__device__ __forceinline__ void tl_stamp(int slot)
{
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == blockDim.x / 2 && lin < FS_TL_MAX) {   // a wave in the middle of the tile: it runs every sweep and stores
        g_timeline[8 * lin + slot] = wall_clock64();
        if (slot == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_timeline[8 * lin + 4] = hw;
            g_timeline[8 * lin + 5] = xcc;
        }
    }
}
#define FS_TL(slot) tl_stamp(slot)
#else
#define FS_TL(slot)
#endif

struct JacobiArgs {
    GK g;
    int yb, ye;        // global rows this launch must produce
    const float *in;   // plane 0 of the input iterate (nullptr => all zero)
    float *out;        // plane 0 of the output iterate
    const float *div;  // pressure only
    size_t stride;     // floats between planes (plane = blockIdx.z)
    int T;             // sweeps in this launch (>= 1)
    int TE;            // rim of the tile that is not stored: T, or T + 1 with the gradient epilogue
    int HX;            // horizontal halo, multiple of 4, >= TE
    float *gu, *gv;    // gradient epilogue (EPI): velocity planes updated in place from the final iterate
    const float *du, *dv;   // divergence prologue (PRO): velocity planes the divergence is computed from ...
    float *div_out;         // ... and the plane it is stored to (for the later launches of the loop)
    unsigned *cfl;          // EPI on a slab: max |v| of the updated v plane over the stored rows (atomicMax), or null
    unsigned *cfl_zero;     // PRO on a slab: word to clear (the bound the loop's LAST launch accumulates into), or null
    float a, den;      // diffuse: a = dt*visc*W*H, den = 1 + 4a
    double rden;       // RN64(1 / den)
    int xcd_chunked;   // tile order, see the kernel
    int stagger;       // start delay of every other resident workgroup, 10 ns ticks (0 = off)
    int stagger_mode;
};

// x / den for a launch-constant den, bit-identical to the IEEE fp32 division the reference formula
// asks for, at 3 instructions instead of the ~12 of v_div_scale/v_rcp/fma.../v_div_fixup:
//     q = RN32( RN64( (double)x * RN64(1/den) ) ).
// Proof sketch: the fp64 product is within 2^-52 (relative) of the exact quotient x/den, while the exact
// quotient of two fp32 numbers is never closer than 2^-49 (relative) to an fp32 rounding boundary
// (midpoint of adjacent floats) and never on one; in the fp32-subnormal range the same holds in
// absolute terms (2^-174 vs 2^-178).  So both round to the same fp32.  fp64 holds every fp32
// (sub)normal exactly, so there are no special cases.  Checked exhaustively over all 2^32 x for a set
// of den by fs_selftest_divide (tests/test_gpu_kernels.py).
__device__ __forceinline__ float div_const(float x, double rden) { return (float)((double)x * rden); }

// A lane's 4 cells of one row as two register pairs, so that every add/mul below that has no
// cross-lane operand is ONE packed instruction (v_pk_add_f32 / v_pk_mul_f32: two IEEE fp32 results
// per lane per issue; same rounding as the scalar forms).
typedef float v2f __attribute__((ext_vector_type(2)));
struct Row {
    v2f a, b;   // a = columns (4l, 4l+1), b = columns (4l+2, 4l+3)
};
__device__ __forceinline__ Row row_of(float4 v) { return Row{v2f{v.x, v.y}, v2f{v.z, v.w}}; }
__device__ __forceinline__ float4 f4_of(const Row &r) { return make_float4(r.a.x, r.a.y, r.b.x, r.b.y); }

// Keeps a scalar sum scalar: without it the compiler re-packs the two adds that take a DPP operand
// (v_add_f32_dpp: the neighbour lane's value folded into the add) into a v_pk_add_f32 whose operand
// pair it first has to assemble with two extra moves.
#define FS_SCALAR(x) asm("" : "+v"(x))

// One sweep of the lane's 4 cells of a row.  c = this row, up/dn = the rows above/below (previous
// iterate), d = divergence (pressure only).  Evaluation order is the reference's:
//   pressure  ((((div + l) + r) + t) + b) / 4                      Shaders:212 (x/4 == x*0.25 exactly)
//   diffuse   (c + a * (((l + r) + t) + b)) / (1 + 4a)             Shaders:103
// 13 VALU instructions per 4 cells for pressure (the scalar form took 20 + moves).
template <int OP>
__device__ __forceinline__ Row sweep_row(const Row &c, const Row &up, const Row &dn, const Row &d, float a, double rden)
{
    const float lx = from_lane_below(c.b.y);   // column 4l-1
    const float rw = from_lane_above(c.a.x);   // column 4l+4
    Row n;
    if (OP == OP_PRESSURE) {
        const v2f mid = {c.a.y, c.b.x};        // right neighbours of pair a == left neighbours of pair b
        float t0 = d.a.x + lx, t1 = d.a.y + c.a.x;
        FS_SCALAR(t0); FS_SCALAR(t1);
        const v2f sa = v2f{t0, t1} + mid;
        const v2f sb0 = d.b + mid;
        float t2 = sb0.x + c.b.y, t3 = sb0.y + rw;
        FS_SCALAR(t2); FS_SCALAR(t3);
        const v2f sb = {t2, t3};
        const v2f quarter = {0.25f, 0.25f};
        n.a = ((sa + up.a) + dn.a) * quarter;
        n.b = ((sb + up.b) + dn.b) * quarter;
    } else {
        // t1, t2 as two scalar adds straight into their halves of the pairs {t0,t1} / {t2,t3}: left to itself the
        // compiler forms ONE packed add (c.a + c.b) and then needs two moves to split its result over the pairs
        float t0 = lx + c.a.y, t3 = c.b.x + rw, t1, t2;
        asm("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(c.a.x), "v"(c.b.x));
        asm("v_add_f32 %0, %1, %2" : "=v"(t2) : "v"(c.a.y), "v"(c.b.y));
        FS_SCALAR(t0); FS_SCALAR(t3);
        const v2f av = {a, a};
        const v2f xa = c.a + av * ((v2f{t0, t1} + up.a) + dn.a);
        const v2f xb = c.b + av * ((v2f{t2, t3} + up.b) + dn.b);
#ifdef FS_PROBE_FASTDIV   // timing experiment only: unguarded q0 = x r; n = q0 den - x; q = q0 - n r
        const float rf = (float)rden, denf = (float)(1.0 / rden);
        const v2f rv = {rf, rf}, dv = {denf, denf};
        const v2f qa0 = xa * rv, qb0 = xb * rv;
        const v2f na = __builtin_elementwise_fma(qa0, dv, -xa), nb = __builtin_elementwise_fma(qb0, dv, -xb);
        n.a = __builtin_elementwise_fma(na, -rv, qa0);
        n.b = __builtin_elementwise_fma(nb, -rv, qb0);
#else
        n.a = v2f{div_const(xa.x, rden), div_const(xa.y, rden)};
        n.b = v2f{div_const(xb.x, rden), div_const(xb.y, rden)};
#endif
    }
    return n;
}

// One sweep src -> dst on the register tile (dst and src are different registers: no copies to keep the previous
// row alive).  EDGE = the tile touches a fixed (boundary / outside) row or column; interior tiles (the vast
// majority) run the lean instantiation with no selects at all.
template <int OP, int R, int NW, bool EDGE>
__device__ __forceinline__ void sweep_once(const Row (&src)[R], Row (&dst)[R], const Row (&d)[R], float4 (*xch)[NW][2][64],
                                           int buf, int lane, int wave, int wup, int wdn, unsigned rowfixed, bool fx0,
                                           bool fx1, bool fx2, bool fx3, float a, double rden)
{
    xch[buf][wave][0][lane] = f4_of(src[0]);
    xch[buf][wave][1][lane] = f4_of(src[R - 1]);
    __builtin_amdgcn_sched_barrier(0);   // (publish first: the scheduler would sink the two writes below the interior rows)
    // the interior rows need nothing from other waves: they are computed while the edge rows travel
    auto fix = [&](Row n, const Row &c, int i) {
        if (EDGE) {
            // fixed cells: pressureOut = 0 (Shaders:201-204), diffuse output = input (Shaders:91-94)
            const Row fixedv = OP == OP_PRESSURE ? Row{v2f{0.f, 0.f}, v2f{0.f, 0.f}} : c;
            if (fx0) n.a.x = fixedv.a.x;
            if (fx1) n.a.y = fixedv.a.y;
            if (fx2) n.b.x = fixedv.b.x;
            if (fx3) n.b.y = fixedv.b.y;
            if ((rowfixed >> i) & 1u) n = fixedv;
        }
        return n;
    };
#pragma unroll
    for (int i = 1; i < R - 1; ++i) dst[i] = fix(sweep_row<OP>(src[i], src[i - 1], src[i + 1], d[i], a, rden), src[i], i);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    const Row above = row_of(xch[buf][wup][1][lane]);    // row above this wave's strip
    const Row below = row_of(xch[buf][wdn][0][lane]);    // row below it
    dst[0] = fix(sweep_row<OP>(src[0], above, R > 1 ? src[1] : below, d[0], a, rden), src[0], 0);
    if (R > 1) dst[R - 1] = fix(sweep_row<OP>(src[R - 1], src[R - 2], below, d[R - 1], a, rden), src[R - 1], R - 1);
#ifdef FS_PROBE_GUARD   // timing experiment only: what a wave-level range guard of the fast division would cost per sweep
    if (OP == OP_DIFFUSE) {
        // smallest |value| and the OR of all bits of the lane's R x 4 results: one v_min3_f32 with |.| modifiers and one v_or3_b32
        // per two values; the wave takes the exact path when a lane holds a non-zero value below 2^-96 (or any zero beside
        // non-zeros, which the minimum cannot tell from a tiny value)
        float mn = 0x1p100f;
        unsigned orb = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(mn) : "v"(dst[i].a.x), "v"(dst[i].a.y));
            asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(mn) : "v"(dst[i].b.x), "v"(dst[i].b.y));
            asm("v_or3_b32 %0, %0, %1, %2" : "+v"(orb) : "v"(dst[i].a.x), "v"(dst[i].a.y));
            asm("v_or3_b32 %0, %0, %1, %2" : "+v"(orb) : "v"(dst[i].b.x), "v"(dst[i].b.y));
        }
        const bool ok = mn >= 0x1p-96f || (orb << 1) == 0u;
        if (__ballot(!ok) != 0ull && lane == 0) xch[buf][wave][0][0].x += 0.0f;   // (stands in for the branch to the exact path)
    }
#endif
}

// T sweeps on the register tile.  Diffuse: two per trip, ping-ponging between p and a second register tile q.
// After sweep s only the tile rows [s, TH - s) are still exact, and only they feed the cells that are stored.  A
// wave whose rows have all dropped out of that range (the outermost waves of the tile: 2 x 7 + 2 x 3 of the 160
// wave-sweeps of a depth-10 launch on the 4 x 16 tile) stops sweeping, leaves its edge rows of the last iterate in
// the exchange array and returns false: it stores nothing, and the caller ends it.  Its neighbour keeps reading
// those rows -- from then on stale values that only reach rows which are out of range themselves.  (s_barrier counts the waves of the workgroup that are still
// running, so the others do not wait for it.)
template <int OP, int R, int NW, bool EDGE>
__device__ __forceinline__ bool run_sweeps_tile(Row (&p)[R], const Row (&d)[R], float4 (*xch)[NW][2][64], int Tall,
                                                int lane, int wave, unsigned rowfixed, bool fx0, bool fx1, bool fx2,
                                                bool fx3, float a, double rden)
{
    const int wup = wave > 0 ? wave - 1 : 0;            // tile rim rows have no neighbour: any value, never used
    const int wdn = wave < NW - 1 ? wave + 1 : NW - 1;
    const int useful = R * (wave < NW - 1 - wave ? wave : NW - 1 - wave) + R - 1;   // sweeps after which no row of this wave is in range
    const int T = Tall < useful ? Tall : useful;
    int s = 0;
    if (OP == OP_DIFFUSE) {
        Row q[R];
        for (; s + 1 < T; s += 2) {
            sweep_once<OP, R, NW, EDGE>(p, q, d, xch, 0, lane, wave, wup, wdn, rowfixed, fx0, fx1, fx2, fx3, a, rden);
            sweep_once<OP, R, NW, EDGE>(q, p, d, xch, 1, lane, wave, wup, wdn, rowfixed, fx0, fx1, fx2, fx3, a, rden);
        }
    }
    // Pressure (p, q and the divergence tile do not fit 64 VGPRs) and the odd last sweep of a diffuse launch
    // (a third live tile for one sweep made the compiler spill): one tile, rows rotated in place.
    for (; s < T; ++s) {
        const int buf = s & 1;
        xch[buf][wave][0][lane] = f4_of(p[0]);
        xch[buf][wave][1][lane] = f4_of(p[R - 1]);
        __syncthreads();
        Row prev = row_of(xch[buf][wup][1][lane]);
        const Row below = row_of(xch[buf][wdn][0][lane]);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const Row c = p[i];
            const Row nx = (i < R - 1) ? p[i + 1] : below;
            Row n = sweep_row<OP>(c, prev, nx, d[i], a, rden);
            if (EDGE) {
                // fixed cells: pressureOut = 0 (Shaders:201-204), diffuse output = input (Shaders:91-94)
                const Row fixedv = OP == OP_PRESSURE ? Row{v2f{0.f, 0.f}, v2f{0.f, 0.f}} : c;
                if (fx0) n.a.x = fixedv.a.x;
                if (fx1) n.a.y = fixedv.a.y;
                if (fx2) n.b.x = fixedv.b.x;
                if (fx3) n.b.y = fixedv.b.y;
                if ((rowfixed >> i) & 1u) n = fixedv;
            }
            p[i] = n;
            prev = c;
        }
    }
    if (T == Tall) return true;
    // last words: the neighbours' next sweep still reads this wave's edge rows of the iterate just computed
    xch[T & 1][wave][0][lane] = f4_of(p[0]);
    xch[T & 1][wave][1][lane] = f4_of(p[R - 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the rows are in LDS before the wave ends
    return false;
}

// MINW = waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument):
// (8 rows, 8 waves) at MINW 4 keeps two 512-thread workgroups per CU, so one can load while the other sweeps.
// EPI (pressure only, last launch of the loop): subtractGradient (Shaders:225-245) from the final iterate while
// it is still in registers -- the sweeps are bound by VALU issue, so the 16 B/cell of velocity traffic ride along
// and the separate pass over p, u, v (Solver:544-547) disappears.  The gradient needs the final p of the four
// neighbours, i.e. one more ring of the tile: the stored region shrinks by TE = T + 1 instead of T.
// PRO (pressure only, first launch of the loop, p = 0): divergence (Shaders:162-181) computed from u, v for the
// tile instead of loaded, and stored once per cell for the later launches: the separate divergence pass
// (Solver:522-525) disappears.  The outermost ring of the tile gets a wrong divergence (its neighbours lie outside
// the tile); that ring is invalid after the first sweep anyway and is never stored.
// Everything a workgroup does with its tile: load (or compute the divergence), T sweeps, store (or subtract the gradient).
// EDGE = the tile touches a fixed (boundary / outside) row or column.  The two instantiations are separate bodies from
// the first load on: the interior one carries no masks, no selects and none of their live ranges.
template <int OP, int R, int NW, bool EPI, bool PRO, bool EDGE>
__device__ __forceinline__ void tile_body(const JacobiArgs &A, const GK &g, float4 (*xch)[NW][2][64], const uint2 tile, const int lane,
                                          const int wave, const int OW, const int OH)
{
    constexpr int TW = 256;
    constexpr int TH = NW * R;
    (void)OW; (void)OH;
    const int x0 = tile.x * OW - A.HX;                   // tile column 0 (global), multiple of 4
    const int ry0 = A.yb + tile.y * OH - A.TE;           // tile row 0 (global)
    const int gx = x0 + 4 * lane;                        // this lane's first column
    const int wy0 = ry0 + wave * R;                      // this wave's first row
    const size_t plane = (size_t)blockIdx.z * A.stride;
    const float *in = A.in ? A.in + plane : nullptr;
    float *out = A.out + plane;

    const bool col_in_buf = gx >= 0 && gx < g.pitch;
    // fixed (never updated) columns of this lane: domain boundary columns and everything outside
    // (a tile that touches no fixed row or column -- EDGE false, the vast majority -- has none of these)
    const bool fx0 = EDGE && ((gx + 0) <= 0 || (gx + 0) >= g.W - 1);
    const bool fx1 = EDGE && ((gx + 1) <= 0 || (gx + 1) >= g.W - 1);
    const bool fx2 = EDGE && ((gx + 2) <= 0 || (gx + 2) >= g.W - 1);
    const bool fx3 = EDGE && ((gx + 3) <= 0 || (gx + 3) >= g.W - 1);

#ifdef FS_PROBE
    {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (threadIdx.x == blockDim.x / 2 && lin < FS_TL_MAX) {
            g_timeline[8 * lin + 6] = EDGE ? 1 : 0;
            g_timeline[8 * lin + 7] = tile.x | (tile.y << 16);
        }
    }
#endif
    Row p[R];
    Row d[R];
    unsigned rowfixed = 0;  // bit i: row i is a boundary row or outside the domain (wave-uniform)
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int y = wy0 + i;
        const bool row_in_buf = y >= g.y0 && y < g.y0 + g.rows;
        const bool rf = EDGE && (y <= 0 || y >= g.H - 1);
        rowfixed |= (rf ? 1u : 0u) << i;
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef FS_PROBE
        const bool probe_no_loads = A.stagger_mode & 32;
#else
        const bool probe_no_loads = false;
#endif
        if (row_in_buf && col_in_buf && !probe_no_loads) {
            const size_t o = row_off(g, y) + gx;
            if (in) pv = ld4(in + o);
            if (OP == OP_PRESSURE && !PRO) dv = ld4(A.div + o);
        }
        p[i] = row_of(pv);
        d[i] = row_of(dv);
    }
    if (PRO) {
        // (a slab's frame: the CFL bound starts from 0 -- every launch with the epilogue, which accumulates into it, is
        // ordered behind every launch with the prologue, so no separate clear has to be enqueued)
        if (A.cfl_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *A.cfl_zero = 0u;
        // v of the rows above / below this wave's strip through the edge-row exchange (buffer 1: the sweeps start
        // with buffer 0), u's left / right neighbours by DPP
        const int wup = wave > 0 ? wave - 1 : 0, wdn = wave < NW - 1 ? wave + 1 : NW - 1;
        float4 vr[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int y = wy0 + i;
            const bool ok = y >= g.y0 && y < g.y0 + g.rows && col_in_buf;
            vr[i] = ok ? ld4(A.dv + row_off(g, y) + gx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        xch[1][wave][0][lane] = vr[0];
        xch[1][wave][1][lane] = vr[R - 1];
        __syncthreads();
        float4 vt = xch[1][wup][1][lane];
        const float4 vlast = xch[1][wdn][0][lane];
        const bool col_st = 4 * lane >= A.HX && 4 * lane < TW - A.HX && col_in_buf;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int y = wy0 + i;
            const int ty = wave * R + i;
            const bool ok = y >= g.y0 && y < g.y0 + g.rows && col_in_buf;
            const float4 uc = ok ? ld4(A.du + row_off(g, y) + gx) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 vb = (i < R - 1) ? vr[i + 1] : vlast;
            const float ul = from_lane_below(uc.w), ur = from_lane_above(uc.x);
            const bool rf = (rowfixed >> i) & 1u;
            float4 dv;   // -0.5 ((u_r - u_l) + (v_b - v_t)); boundary 0                  Shaders:172-180
            dv.x = (rf || fx0) ? 0.0f : -0.5f * ((uc.y - ul) + (vb.x - vt.x));
            dv.y = (rf || fx1) ? 0.0f : -0.5f * ((uc.z - uc.x) + (vb.y - vt.y));
            dv.z = (rf || fx2) ? 0.0f : -0.5f * ((uc.w - uc.y) + (vb.z - vt.z));
            dv.w = (rf || fx3) ? 0.0f : -0.5f * ((ur - uc.z) + (vb.w - vt.w));
            d[i] = row_of(dv);
            if (col_st && ty >= A.TE && ty < TH - A.TE && y >= A.yb && y < A.ye) st4s<2>(A.div_out + row_off(g, y) + gx, dv, g);
            vt = vr[i];
        }
    }

#ifdef FS_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (wave 0's own loads: the stamp is "tile loaded" for that wave)
    FS_TL(1);
#endif
    bool alive;
#ifdef FS_PROBE   // timing probes (scripts/jacobi_loops.py --probe): no sweeps / three times the sweeps
    const int Trun = (A.stagger_mode & 16) ? 0 : (A.stagger_mode & 64) ? 3 * A.T : A.T;
#else
    const int Trun = A.T;
#endif
    alive = run_sweeps_tile<OP, R, NW, EDGE>(p, d, xch, Trun, lane, wave, rowfixed, fx0, fx1, fx2, fx3, A.a, A.rden);
    if (!alive) return;   // (before the epilogue's barrier: a wave that ends is no longer waited for)
    FS_TL(2);

    // store the exact inner region
    const bool col_out = 4 * lane >= A.HX && 4 * lane < TW - A.HX && col_in_buf;
    if (EPI) {
        // final iterate of the rows above / below this wave's strip: one more round of the edge-row exchange
        const int buf = A.T & 1;
        const int wup = wave > 0 ? wave - 1 : 0, wdn = wave < NW - 1 ? wave + 1 : NW - 1;
        xch[buf][wave][0][lane] = f4_of(p[0]);
        xch[buf][wave][1][lane] = f4_of(p[R - 1]);
        // slabs: the workgroup's share of the CFL bound is gathered in LDS and leaves as ONE atomic (see below); every
        // wave still here writes the same start values (wave 0 may have left already)
        __shared__ unsigned wg_cfl[2];
        if (A.cfl && lane == 0) { wg_cfl[0] = 0u; wg_cfl[1] = (unsigned)NW; }
        __syncthreads();
        Row prev = row_of(xch[buf][wup][1][lane]);
        const Row below = row_of(xch[buf][wdn][0][lane]);
        float vmax = 0.0f;   // slabs: CFL bound of the backtrace, max |v| over the stored in-domain cells
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const Row c = p[i];
            const Row nx = (i < R - 1) ? p[i + 1] : below;
            const float lx = from_lane_below(c.b.y);   // column 4l-1
            const float rw = from_lane_above(c.a.x);   // column 4l+4
            const int ty = wave * R + i;
            const int y = wy0 + i;
            if (col_out && ty >= A.TE && ty < TH - A.TE && y >= A.yb && y < A.ye) {
                const size_t o = row_off(g, y) + gx;
                st4s<2>(out + o, f4_of(c), g);
                const bool rf = (rowfixed >> i) & 1u;      // boundary rows keep their velocity (Shaders:230-233)
                if (!rf || A.cfl) {
                    float4 uu = ld4(A.gu + o), vv = ld4(A.gv + o);
                    if (!rf) {
                        // vel.x -= 0.5 (p_r - p_l); vel.y -= 0.5 (p_b - p_t)            Shaders:240-243
                        if (!fx0) { uu.x = uu.x - 0.5f * (c.a.y - lx);    vv.x = vv.x - 0.5f * (nx.a.x - prev.a.x); }
                        if (!fx1) { uu.y = uu.y - 0.5f * (c.b.x - c.a.x); vv.y = vv.y - 0.5f * (nx.a.y - prev.a.y); }
                        if (!fx2) { uu.z = uu.z - 0.5f * (c.b.y - c.a.y); vv.z = vv.z - 0.5f * (nx.b.x - prev.b.x); }
                        if (!fx3) { uu.w = uu.w - 0.5f * (rw - c.b.x);    vv.w = vv.w - 0.5f * (nx.b.y - prev.b.y); }
                        st4s<2>(A.gu + o, uu, g);
                        st4s<2>(A.gv + o, vv, g);
                    }
                    if (A.cfl) {   // padding columns are don't-care
                        if (gx + 0 < g.W) vmax = fmaxf(vmax, fabsf(vv.x));
                        if (gx + 1 < g.W) vmax = fmaxf(vmax, fabsf(vv.y));
                        if (gx + 2 < g.W) vmax = fmaxf(vmax, fabsf(vv.z));
                        if (gx + 3 < g.W) vmax = fmaxf(vmax, fabsf(vv.w));
                    }
                }
            }
            prev = c;
        }
        if (A.cfl) {
            // non-negative floats order like their bit patterns.  Wave -> LDS -> one atomicMax per workgroup whose result
            // nobody waits for: an atomic (or a look at the word first) per wave made every wave of the launch end on a round
            // trip to one L2 channel, +17 us on the last launch of a 3072 x 1536 slab once the flow fills the grid
            for (int s = 32; s > 0; s >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, s));
            if (lane == 0) {
                atomicMax(&wg_cfl[0], __float_as_uint(vmax));
                atomicMin(&wg_cfl[1], (unsigned)wave);
            }
            __syncthreads();
            if (lane == 0 && (unsigned)wave == wg_cfl[1] && wg_cfl[0] != 0u) atomicMax(A.cfl, wg_cfl[0]);
        }
        FS_TL(3);
        return;
    }
    if (col_out) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int ty = wave * R + i;
            const int y = wy0 + i;
            if (ty >= A.T && ty < TH - A.T && y >= A.yb && y < A.ye) st4s<2>(out + row_off(g, y) + gx, f4_of(p[i]), g);
        }
    }
    FS_TL(3);
}

template <int OP, int R, int NW, int MINW, bool EPI, bool PRO>
__global__ __launch_bounds__(NW * 64, MINW) void k_jacobi_block(JacobiArgs A)
{
    constexpr int TW = 256;
    constexpr int TH = NW * R;
    __shared__ float4 xch[2][NW][2][64];

    const GK g = A.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    FS_TL(0);
    // De-phase the workgroups that share a CU.  All workgroups of the first dispatch wave start together
    // and would all load, then all sweep, then all store: HBM idles while the VALUs run and vice versa.
    // Holding back every other resident workgroup by about one load phase lets one workgroup of a CU
    // stream its tile while the other sweeps.  Timing only: results do not depend on it (stagger modes 0..3).
    if (A.stagger > 0) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        bool late;
        switch (A.stagger_mode & 15) {
        case 1: late = (lin & 1u) != 0 && lin < 512u; break;
        case 2: late = ((lin >> 3) & 1u) != 0 && lin < 512u; break;
        case 3: late = ((lin >> 8) & 1u) != 0; break;
        default: late = ((lin >> 8) & 1u) != 0 && lin < 512u; break;
        }
        if (late) {
            const long long t0 = wall_clock64();
            while (wall_clock64() - t0 < A.stagger) __builtin_amdgcn_s_sleep(8);
        }
    }
    const int OW = TW - 2 * A.HX;
    const int OH = TH - 2 * A.TE;
    // Tile order.  Plain blockIdx by default; for a pressure loop whose three planes fit the Infinity Cache
    // each XCD takes one contiguous chunk of the grid, so the overlapping tile rims hit in its L2 (3072^2:
    // 143 -> 128 us per 40 sweeps; at 16384^2, HBM-bound, the same remap costs 4 %: eight distant streams).
    const uint2 tile = A.xcd_chunked ? xcd_tile_chunked() : make_uint2(blockIdx.x, blockIdx.y);
    const int x0t = tile.x * OW - A.HX, ry0t = A.yb + tile.y * OH - A.TE;
    // block-uniform: does this tile touch a fixed column or row at all?
    bool edge_tile = x0t <= 0 || x0t + TW >= g.W - 1 || ry0t <= 0 || ry0t + TH >= g.H - 1;
#ifdef FS_PROBE
    if (A.stagger_mode & 128) edge_tile = true;   // probe: every tile through the edge path
#endif
    if (edge_tile) tile_body<OP, R, NW, EPI, PRO, true>(A, g, xch, tile, lane, wave, OW, OH);
    else tile_body<OP, R, NW, EPI, PRO, false>(A, g, xch, tile, lane, wave, OW, OH);
}

// tuning knobs (JacobiTuning, common.h: a process-wide default and a copy per context)
// Round 1 (profiles/r01_tune_jacobi*.log): depths 10 / 8 and the 4 x 16 tile (64 VGPRs, 8 waves per SIMD) for both loops.
// Round 2, after the kernel was split into an edge and an interior body (profiles/r02_h_retune.log, interleaved A/Bs in the
// frame): the pressure loop at depth 14 (three launches per 40 sweeps instead of four: -4 us per frame; the same per sweep
// at 16384^2) and the diffuse loops on the 5 x 16 tile (256 x 80 cells, 56 VGPRs: 64 of 80 rows stored at depth 8 instead of
// 48 of 64; -11 us per frame, -5 % per sweep at 16384^2).  The pressure loop stays on 4 x 16: with the divergence tile the
// 5-row form spills.
// Round 3 (profiles/r03_tune_tile_*.log, r03_ab_tile_frames.log): beyond the Infinity Cache the taller tiles win -- fewer rim rows
// re-read per stored row outweighs their spills and fewer waves: pressure on 5 x 16 (2816 -> 2597 us per 40 sweeps at 16384^2,
// 374 -> 341 us on a 16384 x 2048 slab), diffuse on 6 x 16 (5694 -> 5515 us); 2 % of those frames.  While the planes sit in
// the cache nothing moves (3072^2: 470 vs 471 us per frame).  Hence variant 8 = by size (launch_block), the default.
static JacobiTuning g_tuning = {{14, 8}, {8, 8}, 0, 0, 100};
JacobiTuning &default_tuning() { return g_tuning; }

template <int OP, int R, int NW, int MINW>
static int launch_variant(const JacobiArgs &A0, const fs_grid *g, int nplanes, hipStream_t stream)
{
    JacobiArgs A = A0;
    const int OW = 256 - 2 * A.HX, OH = NW * R - 2 * A.TE;
    if (OH < 1) { set_error("jacobi: block depth %d too deep for a %d-row tile", A.T, NW * R); return FS_ERR_INVALID; }
    dim3 grid((g->width + OW - 1) / OW, (A.ye - A.yb + OH - 1) / OH, nplanes);
    constexpr bool PR = OP == OP_PRESSURE;
    if (PR && A.gu && A.du) { set_error("jacobi: divergence prologue and gradient epilogue cannot share a launch"); return FS_ERR_INVALID; }
    if (PR && A.gu)
        hipLaunchKernelGGL((k_jacobi_block<OP, R, NW, MINW, PR, false>), grid, dim3(NW * 64), 0, stream, A);
    else if (PR && A.du)
        hipLaunchKernelGGL((k_jacobi_block<OP, R, NW, MINW, false, PR>), grid, dim3(NW * 64), 0, stream, A);
    else
        hipLaunchKernelGGL((k_jacobi_block<OP, R, NW, MINW, false, false>), grid, dim3(NW * 64), 0, stream, A);
    FS_LAUNCH_CHECK("k_jacobi_block");
    return FS_OK;
}

template <int OP>
static int launch_block(const JacobiTuning &tn, const fs_grid *g, int yb, int ye, const float *in, float *out, const float *div,
                        int nplanes, size_t stride, int T, float a, float den, float *gu, float *gv, hipStream_t stream,
                        const float *du = nullptr, const float *dv = nullptr, float *div_out = nullptr, float *cfl = nullptr,
                        float *cfl_zero = nullptr)
{
    JacobiArgs A;
    A.g = make_gk(g);
    A.yb = yb;
    A.ye = ye;
    A.in = in;
    A.out = out;
    A.div = div;
    A.stride = stride;
    A.T = T;
    A.gu = gu;
    A.gv = gv;
    A.du = du;
    A.dv = dv;
    A.div_out = div_out;
    A.cfl = reinterpret_cast<unsigned *>(cfl);
    A.cfl_zero = reinterpret_cast<unsigned *>(cfl_zero);
    A.TE = T + (gu ? 1 : 0);
    A.HX = (A.TE + 3) & ~3;
    A.a = a;
    A.den = den;
    A.rden = 1.0 / (double)den;
    A.stagger = tn.stagger;
    A.stagger_mode = tn.stagger_mode;
    // p in, p out and div of the rows in play, against ~3/4 of the 256 MiB Infinity Cache
    const double rows_bytes = (double)g->pitch * (double)(ye - yb) * sizeof(float);
    A.xcd_chunked = OP == OP_PRESSURE && 3.0 * rows_bytes < 192.0 * 1024 * 1024;
    int variant = tn.variant[OP];
    if (variant == 8)   // by size: in the cache 4 x 16 (pressure) / 5 x 16 (diffuse); beyond it one step taller each
        variant = OP == OP_PRESSURE ? (3.0 * rows_bytes < 192.0 * 1024 * 1024 ? 4 : 7) : (2.0 * rows_bytes < 192.0 * 1024 * 1024 ? 7 : 6);
    switch (variant) {
    case 1: return launch_variant<OP, 16, 8, 2>(A, g, nplanes, stream);   // 256 x 128 tile, 512 threads
    case 2: return launch_variant<OP, 8, 16, 4>(A, g, nplanes, stream);   // 256 x 128 tile, 1024 threads
    case 3: return launch_variant<OP, 16, 4, 2>(A, g, nplanes, stream);   // 256 x 64 tile, 256 threads
    case 0: return launch_variant<OP, 8, 8, 4>(A, g, nplanes, stream);    // 256 x 64 tile, 512 threads
    case 5: return launch_variant<OP, 12, 8, 3>(A, g, nplanes, stream);   // 256 x 96 tile, 512 threads
    case 6: return launch_variant<OP, 6, 16, 8>(A, g, nplanes, stream);   // 256 x 96 tile, 1024 threads, 8 waves/SIMD
    case 7: return launch_variant<OP, 5, 16, 8>(A, g, nplanes, stream);   // 256 x 80 tile, 1024 threads, 8 waves/SIMD
    default: return launch_variant<OP, 4, 16, 8>(A, g, nplanes, stream);  // 256 x 64 tile, 1024 threads
    }
}

// n sweeps ping-ponging a -> b -> a ...; launch j must leave rows [yb-rem, ye+rem) valid,
// rem = sweeps still to come, clipped to the domain.
// gu/gv != nullptr (pressure): the last launch also subtracts the gradient of the final iterate from them
// One launch per step(), so that two chains (row bands, project_n_bands) can be enqueued side by side.
template <int OP>
struct SweepChain {
    JacobiTuning tn;
    fs_grid g;
    const float *div;
    float *src, *dst;
    bool src_zero;
    int nplanes;
    size_t stride;
    float ca, cden;
    int rem, ex, launches = 0;
    float *gu, *gv, *div_out, *cfl;
    bool div_prologue, cfl_reset = false;   // cfl_reset: the first launch (which must then be a prologue launch) clears *cfl
    hipStream_t stream;

    SweepChain(const JacobiTuning &t, const fs_grid *grid, const float *div_, float *a, float *b, bool a_is_zero, int np, size_t str,
               float ca_, float cden_, int n, hipStream_t st, float *gu_ = nullptr, float *gv_ = nullptr, bool pro = false,
               float *div_out_ = nullptr, float *cfl_ = nullptr)
        : tn(t), g(*grid), div(div_), src(a), dst(b), src_zero(a_is_zero), nplanes(np), stride(str), ca(ca_), cden(cden_), rem(n),
          ex(gu_ ? 1 : 0), gu(gu_), gv(gv_), div_out(div_out_), cfl(cfl_), div_prologue(pro), stream(st) {}

    // the rows of the input iterate (and of u, v, div) the whole chain reads must be in the buffer
    int check(int n) const
    {
        const int need_lo = g.yb - n - ex > 0 ? g.yb - n - ex : 0;
        const int need_hi = g.ye + n + ex < g.height ? g.ye + n + ex : g.height;
        if (need_lo < g.y0 || need_hi > g.y0 + g.rows) {
            set_error("jacobi: %d sweeps on rows [%d,%d) need rows [%d,%d) but the buffer holds [%d,%d)", n, g.yb, g.ye, need_lo,
                      need_hi, g.y0, g.y0 + g.rows);
            return FS_ERR_INVALID;
        }
        return FS_OK;
    }
    bool done() const { return rem <= 0; }
    int step()
    {
        const int maxT = tn.depth[OP];
        const int T = rem < maxT ? rem : maxT;
        rem -= T;
        const int more = rem > 0 ? rem + ex : 0;
        int yb = g.yb - more, ye = g.ye + more;
        if (yb < 0) yb = 0;
        if (ye > g.height) ye = g.height;
        const bool last = rem == 0, first = launches == 0;
        // first launch of a projection: the divergence is computed from gu/gv (still untouched: the epilogue only
        // runs in the last launch) and stored to div_out for the launches that follow
        const bool pro = div_prologue && first && !last;
        int rc = launch_block<OP>(tn, &g, yb, ye, src_zero ? nullptr : src, dst, div, nplanes, stride, T, ca, cden,
                                  last ? gu : nullptr, last ? gv : nullptr, stream, pro ? gu : nullptr, pro ? gv : nullptr,
                                  pro ? div_out : nullptr, last ? cfl : nullptr, pro && cfl_reset ? cfl : nullptr);
        if (rc) return rc;
        float *t = src; src = dst; dst = t;
        src_zero = false;
        ++launches;
        return FS_OK;
    }
};

template <int OP>
static int run_sweeps(const JacobiTuning &tn, const fs_grid *g, const float *div, float *a, float *b, bool a_is_zero, int nplanes,
                      size_t stride, float ca, float cden, int n, int *result_in_b, hipStream_t stream,
                      float *gu = nullptr, float *gv = nullptr, bool div_prologue = false, float *div_out = nullptr,
                      float *cfl = nullptr, bool cfl_reset = false)
{
    SweepChain<OP> ch(tn, g, div, a, b, a_is_zero, nplanes, stride, ca, cden, n, stream, gu, gv, div_prologue, div_out, cfl);
    ch.cfl_reset = cfl_reset;
    int rc = ch.check(n);
    if (rc) return rc;
    while (!ch.done())
        if ((rc = ch.step())) return rc;
    if (result_in_b) *result_in_b = ch.launches & 1;
    return FS_OK;
}

// launches run_sweeps makes for n sweeps (the frame planner needs the ping-pong parity ahead of time)
int jacobi_launch_count(const JacobiTuning &t, int op, int n)
{
    const int maxT = t.depth[op == 1 ? 1 : 0];
    return n <= 0 ? 0 : (n + maxT - 1) / maxT;
}

int tuning_set(JacobiTuning &t, int key, int v)
{
    switch (key) {
    case FS_OPT_BLOCK_DEPTH:
    case FS_OPT_BLOCK_DEPTH_PRESSURE:
    case FS_OPT_BLOCK_DEPTH_DIFFUSE:
        if (v < 1 || v > 48) return FS_ERR_INVALID;
        if (key != FS_OPT_BLOCK_DEPTH_DIFFUSE) t.depth[0] = v;
        if (key != FS_OPT_BLOCK_DEPTH_PRESSURE) t.depth[1] = v;
        return FS_OK;
    case FS_OPT_TILE_VARIANT:
    case FS_OPT_TILE_VARIANT_DIFFUSE:
        if (v < 0 || v > 8) return FS_ERR_INVALID;   // 8: by size (launch_block)
        if (key == FS_OPT_TILE_VARIANT) t.variant[0] = v;
        t.variant[1] = v;
        return FS_OK;
    case FS_OPT_STAGGER:
        if (v < 0) return FS_ERR_INVALID;
        t.stagger = v;
        return FS_OK;
    case FS_OPT_BAND_DELAY:
        if (v < 0 || v > 100000) return FS_ERR_INVALID;   // at most 1 ms
        t.band_delay = v;
        return FS_OK;
    case FS_OPT_STAGGER_MODE:
#ifndef FS_PROBE
        if (v > 3) return FS_ERR_INVALID;   // modes 0..3 choose which workgroups start late; higher bits exist in probe builds only
#endif
        if (v < 0) return FS_ERR_INVALID;
        t.stagger_mode = v;
        return FS_OK;
    }
    return FS_ERR_INVALID;
}
int tuning_get(const JacobiTuning &t, int key)
{
    switch (key) {
    case FS_OPT_BLOCK_DEPTH:
    case FS_OPT_BLOCK_DEPTH_PRESSURE: return t.depth[0];
    case FS_OPT_BLOCK_DEPTH_DIFFUSE: return t.depth[1];
    case FS_OPT_TILE_VARIANT: return t.variant[0];
    case FS_OPT_TILE_VARIANT_DIFFUSE: return t.variant[1];
    case FS_OPT_STAGGER: return t.stagger;
    case FS_OPT_STAGGER_MODE: return t.stagger_mode;
    case FS_OPT_BAND_DELAY: return t.band_delay;
    }
    return FS_ERR_INVALID;
}

// exhaustive check of div_const against the IEEE division: all 2^32 bit patterns of x
__global__ __launch_bounds__(256) void k_selftest_divide(float den, double rden, unsigned long long *mismatches)
{
    unsigned long long bad = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const float q = x / den;
        const float f = div_const(x, rden);
        // NaNs compare by "both NaN"; everything else by bits (so the sign of zero is checked too)
        const bool same = (q != q && f != f) || __float_as_uint(q) == __float_as_uint(f);
        bad += same ? 0 : 1;
    }
    if (bad) atomicAdd(mismatches, bad);
}

#ifdef FS_PROBE
// Probe build only: how often does the three-instruction fp32 form  q0 = RN(x r), n = fma(q0, den, -x), q = fma(-n, r, q0)
// with r = RN(1 / den)  (Markstein; "division by a constant known in advance") differ from the IEEE quotient?  All 2^32
// bit patterns of x; counted apart for x inside [2^-100, 2^100] (no intermediate can underflow or overflow there) and outside.
__global__ __launch_bounds__(256) void k_selftest_divide_fast(float den, float r, unsigned long long *out)
{
    unsigned long long bad_in = 0, bad_out = 0;
    unsigned first = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const float q = x / den;
        const float q0 = x * r;
        const float n = __builtin_fmaf(q0, den, -x);
        const float f = __builtin_fmaf(-n, r, q0);
        const bool same = (q != q && f != f) || __float_as_uint(q) == __float_as_uint(f);
        if (!same) {
            const float ax = fabsf(x);
            if (ax >= 0x1p-100f && ax <= 0x1p100f) { bad_in += 1; if (!first) first = (unsigned)i; }
            else bad_out += 1;
        }
    }
    if (bad_in) atomicAdd(&out[0], bad_in);
    if (bad_out) atomicAdd(&out[1], bad_out);
    if (first) atomicMax(&out[2], (unsigned long long)first);
}
#endif

static void diffuse_coeffs(const fs_grid *g, float viscosity, float dt, float *a, float *den)
{
    // Shaders:96  a = deltaTime * viscosity * width * height  (uint -> float)
    *a = ((dt * viscosity) * (float)(unsigned)g->width) * (float)(unsigned)g->height;
    *den = 1.0f + 4.0f * *a;  // Shaders:103
}

#define REQUIRE_FS(who, cond, msg)              \
    do {                                        \
        if (!(cond)) { set_error("%s: %s", who, msg); return FS_ERR_INVALID; } \
    } while (0)

int pressure_n(const JacobiTuning &t, const fs_grid *g, const float *div, float *pa, float *pb, int n, int p_is_zero, int *result_in_b,
               void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    REQUIRE_FS("fs_k_pressureSolve_n", div && pa && pb && pa != pb, "bad pointers");
    REQUIRE_FS("fs_k_pressureSolve_n", n >= 0, "negative sweep count");
    if (result_in_b) *result_in_b = 0;
    if (g->ye <= g->yb) return FS_OK;
    if (n == 0) {
        if (p_is_zero) return fs_k_clear(g, pa, 1, 0, stream);
        return FS_OK;
    }
    return run_sweeps<OP_PRESSURE>(t, g, div, pa, pb, p_is_zero != 0, 1, 0, 0.f, 1.f, n, result_in_b, (hipStream_t)stream);
}

// cfl_reset: *cfl_dev starts from 0 (cleared by the loop's first launch; by a memset when the loop is a single launch)
int project_n_cfl(const JacobiTuning &t, const fs_grid *g, float *u, float *v, float *div, float *pa, float *pb, int n, float *cfl_dev,
                  int *result_in_b, void *stream, int cfl_reset)
{
    int rc = check_grid(g);
    if (rc) return rc;
    REQUIRE_FS("fs_k_project_n", u && v && div && pa && pb && pa != pb, "bad pointers");
    REQUIRE_FS("fs_k_project_n", n >= 1, "needs at least one sweep");
    if (result_in_b) *result_in_b = 0;
    if (cfl_reset && cfl_dev && (g->ye <= g->yb || jacobi_launch_count(t, OP_PRESSURE, n) < 2))
        FS_HIP(hipMemsetAsync(cfl_dev, 0, sizeof(float), (hipStream_t)stream));
    if (g->ye <= g->yb) return FS_OK;
    if (jacobi_launch_count(t, OP_PRESSURE, n) < 2) {
        // a single launch cannot hold both ends (its epilogue would write velocity another tile's prologue reads):
        // the divergence of the rows the sweeps read stays a separate pass
        fs_grid gd = *g;
        gd.yb = g->yb - n > g->y0 ? g->yb - n : g->y0;
        gd.ye = g->ye + n < g->y0 + g->rows ? g->ye + n : g->y0 + g->rows;
        rc = fs_k_divergence(&gd, u, v, div, stream);
        if (rc) return rc;
    }
    return run_sweeps<OP_PRESSURE>(t, g, div, pa, pb, true, 1, 0, 0.f, 1.f, n, result_in_b, (hipStream_t)stream, u, v, true, div,
                                   cfl_dev, cfl_reset != 0);
}

// One wave that sleeps: puts band B of a two-band projection out of step with band A (see project_n_bands).
__global__ __launch_bounds__(64) void k_delay(int ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// The projection of a whole grid as TWO row bands, [yb, split) and [split, ye), each a chain of its own on its own
// stream.  One chain leaves every launch with a partly filled last round of workgroups and a CU idles while its
// workgroups load; two chains out of step fill each other's holes (3072^2: 142 -> ~122 us per 40 sweeps).  Each band
// computes the rows beside the split redundantly, like a slab with deep ghost rows, so the bands never exchange:
//   - band A ping-pongs pa/pb (the context's pressure planes), band B two private planes qa/qb: an intermediate
//     iterate of one band is never overwritten under the other band's reads;
//   - div is written by both first launches -- the same values where the bands overlap;
//   - u, v: a band's LAST launch updates its own rows in place, the other band's FIRST launch reads rows on both
//     sides of the split for the divergence: ev_first[x] (recorded after the first launch of band x) is waited for
//     by the other band's last launch.
// Result: rows [yb, split) of the final iterate in pa/pb (*result_in_b), rows [split, ye) in qa/qb (the same side).
int project_n_bands(const JacobiTuning &t, const fs_grid *g, int split, float *u, float *v, float *div, float *pa, float *pb,
                    float *qa, float *qb, int n, int *result_in_b, void *stream_a, void *stream_b, void *ev_fork, void *ev_join,
                    void *ev_first_a, void *ev_first_b, float *cfl_dev, int cfl_reset)
{
    int rc = check_grid(g);
    if (rc) return rc;
    REQUIRE_FS("project_n_bands", u && v && div && pa && pb && qa && qb && pa != pb && qa != qb, "bad pointers");
    REQUIRE_FS("project_n_bands", split > g->yb && split < g->ye, "split row outside the rows to compute");
    REQUIRE_FS("project_n_bands", jacobi_launch_count(t, OP_PRESSURE, n) >= 2, "needs at least two launches");
    REQUIRE_FS("project_n_bands", stream_a != stream_b && ev_fork && ev_join && ev_first_a && ev_first_b, "needs two streams and four events");
    hipStream_t sa = (hipStream_t)stream_a, sb = (hipStream_t)stream_b;
    fs_grid ga = *g, gb = *g;
    ga.ye = split;
    gb.yb = split;
    // (cfl_dev, slabs: both bands' last launches fold max |v| of their rows into it; cfl_reset: band A's first launch
    // clears it -- both last launches wait for both first ones, ev_first_*)
    SweepChain<OP_PRESSURE> A(t, &ga, div, pa, pb, true, 1, 0, 0.f, 1.f, n, sa, u, v, true, div, cfl_dev);
    SweepChain<OP_PRESSURE> B(t, &gb, div, qa, qb, true, 1, 0, 0.f, 1.f, n, sb, u, v, true, div, cfl_dev);
    A.cfl_reset = cfl_reset != 0 && cfl_dev;
    if ((rc = A.check(n)) || (rc = B.check(n))) return rc;
    FS_HIP(hipEventRecord((hipEvent_t)ev_fork, sa));
    FS_HIP(hipStreamWaitEvent(sb, (hipEvent_t)ev_fork, 0));
    int rca = FS_OK, rcb = FS_OK;
    while (!A.done() && !rca && !rcb) {
        if (A.rem <= t.depth[OP_PRESSURE]) {   // the last launches write u, v
            FS_HIP(hipStreamWaitEvent(sa, (hipEvent_t)ev_first_b, 0));
            FS_HIP(hipStreamWaitEvent(sb, (hipEvent_t)ev_first_a, 0));
        }
        const bool first = A.launches == 0;
        // Two chains that start together stay in step -- every pair of launches ramps up and drains together, like
        // one launch -- so band B starts a fraction of a launch late: from then on one chain is in full swing
        // whenever the other drains and refills.
        if (first && t.band_delay > 0) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, sb, t.band_delay);
        rca = A.step();
        rcb = B.step();
        if (first) {
            FS_HIP(hipEventRecord((hipEvent_t)ev_first_a, sa));
            FS_HIP(hipEventRecord((hipEvent_t)ev_first_b, sb));
        }
    }
    FS_HIP(hipEventRecord((hipEvent_t)ev_join, sb));   // joined even after an error: a capture must end whole
    FS_HIP(hipStreamWaitEvent(sa, (hipEvent_t)ev_join, 0));
    if (result_in_b) *result_in_b = A.launches & 1;
    return rca ? rca : rcb;
}

int diffuse_n(const JacobiTuning &t, const fs_grid *g, float *a, float *b, int nplanes, size_t plane_stride, float viscosity,
              float deltaTime, int n, int *result_in_b, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    REQUIRE_FS("fs_k_diffuse_n", a && b && a != b, "bad pointers");
    REQUIRE_FS("fs_k_diffuse_n", nplanes >= 1 && nplanes <= 4, "nplanes must be 1..4");
    REQUIRE_FS("fs_k_diffuse_n", n >= 0, "negative sweep count");
    if (result_in_b) *result_in_b = 0;
    if (g->ye <= g->yb || n == 0) return FS_OK;
    float ca, cden;
    diffuse_coeffs(g, viscosity, deltaTime, &ca, &cden);
    return run_sweeps<OP_DIFFUSE>(t, g, nullptr, a, b, false, nplanes, plane_stride, ca, cden, n, result_in_b, (hipStream_t)stream);
}

}  // namespace fs

using namespace fs;

#define REQUIRE(cond, msg)                      \
    do {                                        \
        if (!(cond)) { fs::set_error("%s: %s", __func__, msg); return FS_ERR_INVALID; } \
    } while (0)

extern "C" {

#ifdef FS_PROBE
int fs_probe_timeline(unsigned long long *host, long nwords)   // probe build only: the last launch's g_timeline
{
    if (!host || nwords < 0 || nwords > 8L * FS_TL_MAX) return FS_ERR_INVALID;
    FS_HIP(hipDeviceSynchronize());
    FS_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(fs::g_timeline), (size_t)nwords * sizeof(unsigned long long)));
    return FS_OK;
}
#endif

#ifdef FS_PROBE
int fs_probe_divide_fast(float den, unsigned long long out_host[3])   // probe build only: see k_selftest_divide_fast
{
    if (!out_host || !(den > 0.0f)) return FS_ERR_INVALID;
    unsigned long long *d = nullptr;
    FS_HIP(hipMalloc((void **)&d, 3 * sizeof(*d)));
    FS_HIP(hipMemset(d, 0, 3 * sizeof(*d)));
    hipLaunchKernelGGL(k_selftest_divide_fast, dim3(256 * 16), dim3(256), 0, nullptr, den, 1.0f / den, d);
    hipError_t e = hipMemcpy(out_host, d, 3 * sizeof(*d), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fs::hip_fail(e, "fs_probe_divide_fast");
    return FS_OK;
}
#endif

int fs_selftest_divide(float den, unsigned long long *mismatches_host)
{
    if (!mismatches_host || !(den > 0.0f)) return FS_ERR_INVALID;
    unsigned long long *d = nullptr;
    FS_HIP(hipMalloc((void **)&d, sizeof(*d)));
    FS_HIP(hipMemset(d, 0, sizeof(*d)));
    hipLaunchKernelGGL(k_selftest_divide, dim3(256 * 16), dim3(256), 0, nullptr, den, 1.0 / (double)den, d);
    hipError_t e = hipMemcpy(mismatches_host, d, sizeof(*d), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fs::hip_fail(e, "fs_selftest_divide");
    return FS_OK;
}

int fs_k_pressureSolve_n(const fs_grid *g, const float *div, float *pa, float *pb, int n, int p_is_zero,
                         int *result_in_b, void *stream)
{
    return fs::pressure_n(fs::default_tuning(), g, div, pa, pb, n, p_is_zero, result_in_b, stream);
}

int fs_k_pressureSolve_n_gradient(const fs_grid *g, const float *div, float *pa, float *pb, int n, int p_is_zero,
                                  float *u, float *v, int *result_in_b, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    REQUIRE(div && pa && pb && pa != pb && u && v, "bad pointers");
    REQUIRE(n >= 1, "needs at least one sweep (use fs_k_subtractGradient after fs_k_clear otherwise)");
    if (result_in_b) *result_in_b = 0;
    if (g->ye <= g->yb) return FS_OK;
    // (the launches before the last leave one more row of p exact on each side: the gradient reads it)
    return run_sweeps<OP_PRESSURE>(fs::default_tuning(), g, div, pa, pb, p_is_zero != 0, 1, 0, 0.f, 1.f, n, result_in_b,
                                   (hipStream_t)stream, u, v);
}

int fs_k_project_n_cfl(const fs_grid *g, float *u, float *v, float *div, float *pa, float *pb, int n, float *cfl_dev,
                       int *result_in_b, void *stream)
{
    return fs::project_n_cfl(fs::default_tuning(), g, u, v, div, pa, pb, n, cfl_dev, result_in_b, stream, 0);
}

int fs_k_project_n(const fs_grid *g, float *u, float *v, float *div, float *pa, float *pb, int n, int *result_in_b,
                   void *stream)
{
    return fs::project_n_cfl(fs::default_tuning(), g, u, v, div, pa, pb, n, nullptr, result_in_b, stream, 0);
}

int fs_k_pressureSolve(const fs_grid *g, const float *div, const float *p, float *pOut, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    REQUIRE(div && p && pOut && p != pOut, "bad pointers");
    if (g->ye <= g->yb) return FS_OK;
    // one sweep: a is only read
    return run_sweeps<OP_PRESSURE>(fs::default_tuning(), g, div, const_cast<float *>(p), pOut, false, 1, 0, 0.f, 1.f, 1, nullptr,
                                   (hipStream_t)stream);
}

int fs_k_diffuse_n(const fs_grid *g, float *a, float *b, int nplanes, size_t plane_stride, float viscosity,
                   float deltaTime, int n, int *result_in_b, void *stream)
{
    return fs::diffuse_n(fs::default_tuning(), g, a, b, nplanes, plane_stride, viscosity, deltaTime, n, result_in_b, stream);
}

int fs_k_diffuse(const fs_grid *g, const float *in, float *out, int nplanes, size_t plane_stride, float viscosity,
                 float deltaTime, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    REQUIRE(in && out && in != out, "bad pointers");
    REQUIRE(nplanes >= 1 && nplanes <= 4, "nplanes must be 1..4");
    if (g->ye <= g->yb) return FS_OK;
    float ca, cden;
    diffuse_coeffs(g, viscosity, deltaTime, &ca, &cden);
    return run_sweeps<OP_DIFFUSE>(fs::default_tuning(), g, nullptr, const_cast<float *>(in), out, false, nplanes, plane_stride, ca, cden, 1,
                                  nullptr, (hipStream_t)stream);
}

}  // extern "C"
