// valu_issue.hip -- issue-rate microbenchmark for the instructions the Jacobi sweeps are made of (gfx950).
//
// What the roofline of k_jacobi_block needs and no datasheet states per instruction: how many cycles one SIMD is
// occupied by ONE wave64 instruction of each kind, at 1, 2, 4 and 8 waves per SIMD.  Every kernel runs a loop of
// independent instructions of one kind (16 chains per lane, so no dependency stall) and reports
//   * cyc/inst/SIMD  = waves per SIMD x in-kernel cycles (s_memtime) of a wave / instructions it issued, median over
//                      waves: the cost of one wave-instruction in SIMD cycles;
//   * Tlane-ops/s    = lanes x instructions (x 2 for packed) / wall time (HIP events) over the whole chip.
// Build + run: hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_issue scripts/valu_issue.hip && /tmp/valu_issue
// Committed log: profiles/r02_valu_issue.log.  bench.py's roofline.valu_issue uses the cycle costs printed here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

enum { OP_ADD = 0, OP_PK_ADD, OP_ADD_DPP, OP_CVT_F64_F32, OP_MUL_F64, OP_CVT_F32_F64, OP_PK_MUL, OP_MOV_B64, OP_FMA, OP_PK_FMA,
       OP_DIVSEQ, OP_MUL, OP_ADD_E64, OP_FMAC, OP_FMA3, OP_PK_MOV, OP_MOV_B32, OP_CNDMASK, OP_COUNT };
static const char *names[OP_COUNT] = {"v_add_f32", "v_pk_add_f32", "v_add_f32_dpp wave_shr:1", "v_cvt_f64_f32", "v_mul_f64",
                                      "v_cvt_f32_f64", "v_pk_mul_f32", "v_mov_b64", "v_fma_f32", "v_pk_fma_f32",
                                      "cvt_f64+mul_f64+cvt_f32 (div_const)", "v_mul_f32", "v_add_f32_e64 (VOP3 encoding)",
                                      "v_fmac_f32_e32", "v_fma_f32, three distinct sources", "v_pk_mov_b32 op_sel:[1,0]", "v_mov_b32",
                                      "v_cndmask_b32_e32"};
static const int results_per_inst[OP_COUNT] = {1, 2, 1, 1, 1, 1, 2, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1, 1};
static const int insts_per_step[OP_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3, 1, 1, 1, 1, 1, 1, 1};

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(1024) void k_issue(int iters, unsigned long long *cycles, float *sink)
{
    constexpr bool USE_F = OP == OP_ADD || OP == OP_MUL || OP == OP_FMA || OP == OP_ADD_DPP || OP == OP_CVT_F64_F32 ||
                           OP == OP_CVT_F32_F64 || OP == OP_DIVSEQ || OP == OP_ADD_E64 || OP == OP_FMAC || OP == OP_FMA3 ||
                           OP == OP_MOV_B32 || OP == OP_CNDMASK;
    constexpr bool USE_P = OP == OP_PK_ADD || OP == OP_PK_MUL || OP == OP_PK_FMA || OP == OP_MOV_B64 || OP == OP_PK_MOV;
    constexpr bool USE_D = OP == OP_CVT_F64_F32 || OP == OP_MUL_F64 || OP == OP_CVT_F32_F64 || OP == OP_DIVSEQ;
    float f[USE_F ? 16 : 1];
    v2f p[USE_P ? 16 : 1];
    constexpr int ND = USE_D ? (USE_F ? 8 : 16) : 1;
    double d[ND];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float x = 1.0f + 0.001f * (float)(threadIdx.x + i);
        if (USE_F) f[i % (USE_F ? 16 : 1)] = x;
        if (USE_P) p[i % (USE_P ? 16 : 1)] = v2f{x, x + 0.5f};
        if (USE_D) d[i % ND] = (double)x;
    }
    const float c = 1.0000001f;
    float c2 = 0.9999999f;
    asm volatile("" : "+v"(c2));
    const v2f pc = {c, c};
    const double dc = 1.0000000001;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == OP_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_MOV_B64) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(p[i]) : "v"(pc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_ADD_DPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_CVT_F64_F32) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i % ND]) : "v"(f[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_MUL_F64) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_CVT_F32_F64) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i % ND]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_ADD_E64) {
#define X(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_FMAC) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_FMA3) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c), "v"(c2));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_PK_MOV) {
#define X(i) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(p[i]) : "v"(pc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_MOV_B32) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (OP == OP_DIVSEQ) {
#define X(i) asm volatile("v_cvt_f64_f32 %1, %0\n\tv_mul_f64 %1, %1, %2\n\tv_cvt_f32_f64 %0, %1" : "+v"(f[i]), "=&v"(d[i % ND]) : "v"(dc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (USE_F) acc += f[i % (USE_F ? 16 : 1)];
        if (USE_P) acc += p[i % (USE_P ? 16 : 1)].x + p[i % (USE_P ? 16 : 1)].y;
        if (USE_D) acc += (float)d[i % ND];
    }
    if (acc == 12345.678f) sink[0] = acc;   // keeps every chain alive
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(int waves_per_simd, float ghz_hint)
{
    // 4 SIMDs per CU: w waves per SIMD = 4 w waves per CU.  Blocks of up to 16 waves, 1 or 2 per CU.
    const int waves_per_cu = 4 * waves_per_simd;
    const int blocks_per_cu = waves_per_cu > 16 ? 2 : 1;
    const int block = 64 * waves_per_cu / blocks_per_cu;
    const int grid = 256 * blocks_per_cu;
    const int iters = 5000;
    const int nw = grid * (block / 64);
    unsigned long long *cyc;
    float *sink;
    hipMalloc(&cyc, nw * sizeof(*cyc));
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(block), 0, 0, 2000, cyc, sink);   // warm-up (clocks)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(block), 0, 0, iters, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(*cyc), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double insts = 64.0 * iters * insts_per_step[OP];
    // s_memtime counts shader cycles: the cycles one wave took / its instructions / the w waves that share its SIMD = SIMD cycles per instruction
    const double wave_ticks = (double)h[nw / 2];
    const double cyc_in_kernel = wave_ticks / insts / waves_per_simd;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_issue<OP>, block, 0);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_issue<OP>));
    if (occ < blocks_per_cu) printf("  (!) only %d of %d blocks per CU resident, %d VGPRs\n", occ, blocks_per_cu, fa.numRegs);
    const double lane_ops = (double)nw * 64.0 * insts * results_per_inst[OP];
    const double t = ms * 1e-3;
    // one SIMD issued (waves_per_simd * insts) instructions in t seconds
    const double ns_per_inst_simd = t * 1e9 / (waves_per_simd * insts);
    printf("%-38s %d waves/SIMD  %7.3f ms  %5.2f cyc/inst/SIMD in-kernel  (wall: %6.3f ns = %5.2f cyc @%.2f GHz)  %7.2f Tlane-ops/s\n",
           names[OP], waves_per_simd, ms, cyc_in_kernel, ns_per_inst_simd, ns_per_inst_simd * ghz_hint, ghz_hint, lane_ops / t / 1e12);
    hipFree(cyc);
    hipFree(sink);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

// clock the chip holds under an all-VALU load: a dependent chain of v_add_f32 (4 cycles latency is NOT assumed) --
// instead count cycles with s_memtime against s_memrealtime (100 MHz), as MI355X_MICROARCH.md (DVFS item 6) says.
__global__ void k_clock(int iters, unsigned long long *out)
{
    float a = (float)threadIdx.x;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#define X(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a));
        REP16(X)
#undef X
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (a == 12345.678f) out[2] = 1;
}

int main()
{
    unsigned long long *o, h[2];
    hipMalloc(&o, 32);
    hipLaunchKernelGGL(k_clock, dim3(2048), dim3(1024), 0, 0, 200000, o);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_clock, dim3(2048), dim3(1024), 0, 0, 200000, o);
    hipDeviceSynchronize();
    hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    const float ghz = (float)((double)h[0] / (double)h[1] * 0.1);
    printf("in-kernel clock under an all-VALU load: %.3f GHz (s_memtime %llu / s_memrealtime %llu x 100 MHz)\n", ghz, h[0], h[1]);
    for (int w : {1, 2, 4, 8}) {
        run<OP_ADD>(w, ghz);
        run<OP_MUL>(w, ghz);
        run<OP_FMA>(w, ghz);
        run<OP_PK_ADD>(w, ghz);
        run<OP_PK_MUL>(w, ghz);
        run<OP_PK_FMA>(w, ghz);
        run<OP_MOV_B64>(w, ghz);
        run<OP_ADD_DPP>(w, ghz);
        run<OP_CVT_F64_F32>(w, ghz);
        run<OP_MUL_F64>(w, ghz);
        run<OP_CVT_F32_F64>(w, ghz);
        run<OP_DIVSEQ>(w, ghz);
        run<OP_ADD_E64>(w, ghz);
        run<OP_FMAC>(w, ghz);
        run<OP_FMA3>(w, ghz);
        run<OP_PK_MOV>(w, ghz);
        run<OP_MOV_B32>(w, ghz);
        run<OP_CNDMASK>(w, ghz);
    }
    return 0;
}
