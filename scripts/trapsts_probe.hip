// Probe: do the sticky IEEE exception bits of TRAPSTS (hwreg 3, bits 0..6: invalid, input denormal, div0, overflow,
// underflow, inexact, int div0) accumulate for packed fp32 VALU ops on gfx950 with traps disabled, can a wave clear
// them with s_setreg, and how soon after the VALU op does s_getreg see them?
// build: hipcc -O2 --offload-arch=gfx950 -o scripts/trapsts_probe.bin scripts/trapsts_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k(const float *x, float den, float r, unsigned *flags, float *q, int n)
{
    for (int i = 0; i < n; ++i) {
        v2f xv = {x[2 * i], x[2 * i + 1]};
        unsigned before, after;
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 7), 0\n\ts_nop 2\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 7)" : "=s"(before)::"memory");
        v2f q0, nr, qq;
        const v2f dv = {den, den}, rv = {r, r};
        asm volatile("v_pk_mul_f32 %0, %3, %4\n\t"
                     "v_pk_fma_f32 %1, %0, %5, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                     "v_pk_fma_f32 %2, %1, %4, %0 neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
                     : "=&v"(q0), "=&v"(nr), "=&v"(qq) : "v"(xv), "v"(rv), "v"(dv));
        int dep = __builtin_amdgcn_readfirstlane(__float_as_int(qq.x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 7)" : "=s"(after) : "s"(dep) : "memory");
        if (threadIdx.x == 0) { flags[2 * i] = before; flags[2 * i + 1] = after; q[2 * i] = qq.x; q[2 * i + 1] = qq.y; }
    }
}

int main()
{
    const float den = 2832.1677f, r = 1.0f / den;
    float hx[] = {1.0f, 3.0f,  1e-30f, 2.0f,  1e-37f, 1.0f,  1e-40f, 1.0f,  0.0f, -0.0f,  __builtin_inff(), 1.0f,  3e38f, 1.0f, 2832.1677f * 4.0f, 2832.1677f, 1e-33f, -1e-35f};
    const int n = sizeof(hx) / sizeof(float) / 2;
    float *dx, *dq; unsigned *df;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dq, sizeof(hx)); hipMalloc(&df, sizeof(hx));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, den, r, df, dq, n);
    float hq[2 * n]; unsigned hf[2 * n];
    hipMemcpy(hq, dq, sizeof(hx), hipMemcpyDeviceToHost); hipMemcpy(hf, df, sizeof(hx), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        unsigned b0, b1, e0, e1; float e[2] = {hx[2 * i] / den, hx[2 * i + 1] / den};
        memcpy(&b0, &hq[2 * i], 4); memcpy(&b1, &hq[2 * i + 1], 4); memcpy(&e0, &e[0], 4); memcpy(&e1, &e[1], 4);
        printf("x = (%g, %g): flags before %02x after %02x   q = %08x %08x  IEEE %08x %08x %s\n", hx[2 * i], hx[2 * i + 1], hf[2 * i], hf[2 * i + 1], b0, b1, e0, e1,
               (b0 == e0 && b1 == e1) ? "same" : "DIFFERENT");
    }
    return 0;
}
