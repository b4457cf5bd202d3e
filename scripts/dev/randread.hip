// Development microbenchmark (not part of the product): dependent random reads of 32/64/128-byte records, the access
// pattern of BVH traversal.  hipcc --offload-arch=gfx950 -O3 randread.hip -o randread && ./randread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int BYTES>
__global__ __launch_bounds__(64, 6) void chase(const float4* __restrict__ a, uint32_t n_rec, int iters, int active, uint32_t* out) {
    uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0;
    if ((int)threadIdx.x < active) {
        for (int i = 0; i < iters; i++) {
            const uint32_t r = (uint32_t)(((unsigned long long)x * n_rec) >> 32);
            const float4* p = a + (size_t)r * (BYTES / 16);
            float s = 0;
#pragma unroll
            for (int k = 0; k < BYTES / 16; k++) { const float4 v = p[k]; s += v.x + v.y + v.z + v.w; }
            acc += s;
            x = x * 1664525u + 1013904223u + (uint32_t)__float_as_uint(s);  // next index depends on the data
        }
    }
    if (acc == 123.456f) out[0] = x;
}
int main() {
    const size_t sizes[] = {(size_t)1 << 20, (size_t)3 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)1 << 30};
    uint32_t* out; hipMalloc(&out, 4);
    for (size_t bytes : sizes) {
        float4* a; hipMalloc(&a, bytes); hipMemset(a, 0, bytes);
        for (int active : {64}) {
            for (int rec : {32, 64, 128}) {
                const int blocks = 256 * 24, iters = 2000;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (rec == 32) hipLaunchKernelGGL(chase<32>, dim3(blocks), dim3(64), 0, 0, a, (uint32_t)(bytes / 32), iters, active, out);
                    if (rec == 64) hipLaunchKernelGGL(chase<64>, dim3(blocks), dim3(64), 0, 0, a, (uint32_t)(bytes / 64), iters, active, out);
                    if (rec == 128) hipLaunchKernelGGL(chase<128>, dim3(blocks), dim3(64), 0, 0, a, (uint32_t)(bytes / 128), iters, active, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double recs = (double)blocks * active * iters;
                printf("array %5zu MB  lanes %2d  record %3d B: %7.2f ms  %6.2f Grec/s  %6.2f TB/s useful  %.2f us/iteration\n", bytes >> 20, active, rec, ms,
                       recs / ms / 1e6, recs * rec / ms / 1e9, ms * 1e3 / iters);
            }
        }
        hipFree(a);
    }
    return 0;
}
