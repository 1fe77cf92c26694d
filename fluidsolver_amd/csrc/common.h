// common.h -- shared device/host helpers of libfluidsolver_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "fluidsolver.h"

namespace fs {

// Plane geometry as the kernels see it (by value in the kernel argument block).
struct GK {
    int W, H;      // global domain
    int pitch;     // floats per row (multiple of 64)
    int y0, rows;  // buffer holds global rows [y0, y0+rows)
    int nt;        // planes too big to be found in a cache again: results leave through streaming stores (st4s); bit 0 passes, bit 1 Jacobi
};

int stream_stores(const fs_grid *g);   // context.hip: the size rule
inline GK make_gk(const fs_grid *g) { return GK{g->width, g->height, g->pitch, g->y0, g->rows, stream_stores(g)}; }

// Tuning of the two Jacobi loops.  One process-wide default (fs_config_set: what the per-kernel entry points fs_k_*_n use and
// what a new context starts from) and one copy per context (fs_set_option), so two contexts never change each other's.
struct JacobiTuning {
    int depth[2];      // sweeps fused per launch: [0] pressure, [1] diffuse
    int variant[2];    // register-tile shape, see launch_block
    int stagger, stagger_mode;
    int band_delay;    // two-band projection: head start of band A over band B, 10 ns ticks (FS_OPT_BAND_DELAY)
};
JacobiTuning &default_tuning();
int tuning_set(JacobiTuning &t, int key, int value);   // key = FS_OPT_*; FS_ERR_INVALID for an unknown key or a bad value
int tuning_get(const JacobiTuning &t, int key);
int jacobi_launch_count(const JacobiTuning &t, int op, int n);
// fs_k_pressureSolve_n / fs_k_project_n_cfl / fs_k_diffuse_n with an explicit tuning (the context's)
int pressure_n(const JacobiTuning &t, const fs_grid *g, const float *div, float *pa, float *pb, int n, int p_is_zero, int *result_in_b,
               void *stream);
int project_n_cfl(const JacobiTuning &t, const fs_grid *g, float *u, float *v, float *div, float *pa, float *pb, int n, float *cfl_dev,
                  int *result_in_b, void *stream, int cfl_reset);
int diffuse_n(const JacobiTuning &t, const fs_grid *g, float *a, float *b, int nplanes, size_t plane_stride, float viscosity,
              float deltaTime, int n, int *result_in_b, void *stream);
// the projection as two row bands on two streams (jacobi.hip); events are hipEvent_t
int project_n_bands(const JacobiTuning &t, const fs_grid *g, int split, float *u, float *v, float *div, float *pa, float *pb,
                    float *qa, float *qb, int n, int *result_in_b, void *stream_a, void *stream_b, void *ev_fork, void *ev_join,
                    void *ev_first_a, void *ev_first_b, float *cfl_dev, int cfl_reset);

// argument block + launch geometry of a k_add_forces graph node (passes.hip); must stay put while the node params are in use
struct SplatNode {
    GK g;
    int box[4];
    float *u, *v, *r, *gch, *b;
    float f[9];
    void *args[19];
    hipKernelNodeParams np;
};
void splat_node_params(SplatNode &sn, const fs_grid *g, float *u, float *v, float *r, float *gch, float *b, const float s[9]);

// k_advect on rows [g->yb, hole_y0) and [hole_y1, g->ye) in one launch (passes.hip); stream_result: streaming stores regardless of size
int advect_around(const fs_grid *g, int hole_y0, int hole_y1, const float *u, const float *v, const float *in, float *out, int nplanes,
                  size_t plane_stride, float deltaTime, int fade, float fadeRate, int *err_flag, void *stream, int stream_result);

void set_error(const char *fmt, ...);
const char *get_error();
int check_grid(const fs_grid *g);
int hip_fail(hipError_t e, const char *what);

#define FS_HIP(call)                                        \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return fs::hip_fail(e_, #call); \
    } while (0)

#define FS_LAUNCH_CHECK(name)                               \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return fs::hip_fail(e_, name); \
    } while (0)

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ size_t row_off(const GK &g, int y)
{
    return (size_t)(y - g.y0) * (size_t)g.pitch;
}
// Row index clamped into the buffer: a caller that under-sized a halo gets wrong
// values on that row (and FS_ERR_HALO from advect), never an out-of-bounds access.
__device__ __forceinline__ int clamp_row(const GK &g, int y)
{
    return min(max(y, g.y0), g.y0 + g.rows - 1);
}
__device__ __forceinline__ bool is_boundary(const GK &g, int x, int y)
{
    // gid.x == 0 || gid.x >= width-1 || gid.y == 0 || gid.y >= height-1  (e.g. Shaders:91)
    return x == 0 || x >= g.W - 1 || y == 0 || y >= g.H - 1;
}
// XCD-aware tile index.  Workgroups are dealt round-robin over the 8 XCDs in dispatch order (observed
// behaviour, not a contract: it is used here for L2 affinity only, never for correctness), so with the
// plain blockIdx mapping the tiles that share halo rows and columns sit on different L2s and each
// fetches them from the Infinity Cache / HBM again.  Here the tile rows are cut into bands of M tile
// rows and the bands are dealt to the XCDs in turn: the tiles of a band share one L2, while the eight
// XCDs still sweep the same 8*M tile rows of the grid together (giving every XCD one large contiguous
// chunk instead was measured slower at 16384^2: eight distant DRAM streams).
// The launch pads gridDim.y to a multiple of 8*M (xcd_pad_rows); tiles past the real grid exit.
template <int M>
__device__ __forceinline__ uint2 xcd_tile()
{
    const unsigned gx = gridDim.x;
    const unsigned lin = blockIdx.x + gx * blockIdx.y;
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned per_band = gx * M;
    const unsigned b = idx / per_band, w = idx - b * per_band;
    const unsigned wy = w / gx;
    return make_uint2(w - wy * gx, (b * 8u + xcd) * M + wy);
}
// The other extreme: every XCD gets ONE contiguous chunk of the (y, x) tile order (bijective for any grid).
// Pays only while the whole working set sits in the Infinity Cache (see jacobi.hip).
__device__ __forceinline__ uint2 xcd_tile_chunked()
{
    const unsigned gx = gridDim.x, total = gx * gridDim.y;
    const unsigned lin = blockIdx.x + gx * blockIdx.y;
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned t = xcd * q + (xcd < r ? xcd : r) + idx;
    const unsigned y = t / gx;
    return make_uint2(t - y * gx, y);
}
inline unsigned xcd_pad_rows(unsigned tile_rows, unsigned m) { return (tile_rows + 8u * m - 1u) / (8u * m) * (8u * m); }

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// a result plane's store: nontemporal when the plane is too big to be read back from a cache (GK::nt; BIT: which class of kernel asks)
template <int BIT = 1>
__device__ __forceinline__ void st4s(float *p, float4 v, const GK &g)
{
    if (g.nt & BIT) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v x = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(x, reinterpret_cast<f4v *>(p));
    } else {
        st4(p, v);
    }
}

// max into a device word that thousands of waves share (the CFL bound of a slab): the value only grows, so a wave that
// reads it first -- straight from L2; a stale, smaller reading merely costs the atomic it would have saved -- and finds
// it already at least as large has nothing to add.  Without this every wave of a launch queues on one address (the
// projection's last launch at 3072^2: 54 -> 88 us once the flow fills the grid).
__device__ __forceinline__ void atomic_max_if_larger(unsigned *addr, unsigned bits)
{
    if (bits > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(addr, bits);
}

// value of lane-1 / lane+1 of the wave (0 at the wave ends): one DPP operand, no LDS.
__device__ __forceinline__ float from_lane_below(float x)  // lane i gets lane i-1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float x)  // lane i gets lane i+1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
}

}  // namespace fs
