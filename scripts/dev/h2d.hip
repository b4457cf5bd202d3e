// development aid: what a commit's host->device traffic costs on this box.  hipMemcpy of N MB from pageable memory, from
// pinned memory, hipHostRegister of the pageable block, and a device-side copy for scale.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 160;
    const size_t n = mb << 20;
    char* d = nullptr; hipMalloc(&d, n);
    char* h = (char*)malloc(n); memset(h, 1, n);
    char* p = nullptr; hipHostMalloc(&p, n, 0); memset(p, 2, n);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t1 = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice); double t2 = now();
        printf("%zu MB: pageable %.1f ms (%.1f GB/s), pinned %.1f ms (%.1f GB/s)\n", mb, (t1 - t0) * 1e3, n / (t1 - t0) * 1e-9, (t2 - t1) * 1e3, n / (t2 - t1) * 1e-9);
    }
    double t0 = now(); hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault); double t1 = now();
    printf("hipHostRegister: %.1f ms (%s)\n", (t1 - t0) * 1e3, hipGetErrorString(e));
    if (e == hipSuccess) { t0 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); t1 = now(); printf("registered copy %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) * 1e-9); t0 = now(); hipHostUnregister(h); t1 = now(); printf("unregister %.1f ms\n", (t1 - t0) * 1e3); }
    // four threads' worth of parallel pageable copies (the commit's uploader could split the arrays)
    return 0;
}
