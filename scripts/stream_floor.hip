// Streaming floor of the dye advection at 3072^2: read 5 planes (u, v, three dye planes), write 3 other planes, 16 bytes per
// lane, nothing else -- the bytes of k_advect<3, true> without its taps.  Also the in-place variant (read + write 3 planes).
// build: hipcc -O2 --offload-arch=gfx950 -o scripts/stream_floor.bin scripts/stream_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ u, const float4 *__restrict__ v, const float4 *__restrict__ in,
                                              float4 *__restrict__ out, size_t plane4, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = u[i], b = v[i];
    const float s = (a.x + b.x) * 0.0f;   // (keeps the two reads alive; exact zero for finite data)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float4 x = in[c * plane4 + i];
        x.x += s; x.y += s; x.z += s; x.w += s;
        out[c * plane4 + i] = x;
    }
}
__global__ __launch_bounds__(256) void k_inplace(float4 *__restrict__ d, size_t plane4, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float4 x = d[c * plane4 + i];
        x.x *= 0.998f; x.y *= 0.998f; x.z *= 0.998f; x.w *= 0.998f;
        d[c * plane4 + i] = x;
    }
}
int main()
{
    const size_t W = 3072, H = 3072, n4 = W * H / 4, plane4 = n4;
    float4 *u, *v, *in, *out, *junk;
    hipMalloc(&u, n4 * 16); hipMalloc(&v, n4 * 16); hipMalloc(&in, 3 * n4 * 16); hipMalloc(&out, 3 * n4 * 16);
    const size_t junk4 = (size_t)512 * 1024 * 1024 / 16;   // 512 MB: flushes the Infinity Cache between launches
    hipMalloc(&junk, junk4 * 16);
    hipMemset(u, 0, n4 * 16); hipMemset(v, 0, n4 * 16); hipMemset(in, 0, 3 * n4 * 16); hipMemset(out, 0, 3 * n4 * 16); hipMemset(junk, 0, junk4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((n4 + 255) / 256);
    float ms;
    for (int flush = 0; flush < 2; ++flush) {
        float tot = 0.0f; const int reps = 20;
        for (int r = 0; r < reps + 3; ++r) {
            if (flush) hipMemsetAsync(junk, r, junk4 * 16, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_copy, grid, dim3(256), 0, 0, u, v, in, out, plane4, n4);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) tot += ms;
            float4 *t = in; in = out; out = t;   // the output of one launch is the input of the next, as in the frames
        }
        printf("copy  5 planes in, 3 planes out (301 MB)%s: %.1f us\n", flush ? ", 512 MB of other traffic between launches" : ", back to back", tot / reps * 1e3);
        tot = 0.0f;
        for (int r = 0; r < reps + 3; ++r) {
            if (flush) hipMemsetAsync(junk, r, junk4 * 16, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_inplace, grid, dim3(256), 0, 0, in, plane4, n4);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) tot += ms;
        }
        printf("in place, 3 planes (226 MB)%s: %.1f us\n", flush ? ", 512 MB of other traffic between launches" : ", back to back", tot / reps * 1e3);
    }
    return 0;
}
