// dev tool: what do device / pinned allocations cost on this box?  (the CLI's end-to-end time has seconds in them)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t gb : {1, 4, 16, 64}) {
            void* p = nullptr; const size_t n = gb << 30;
            double t0 = now(); hipError_t e = hipMalloc(&p, n); double t1 = now();
            if (e != hipSuccess) { printf("hipMalloc %zu GB failed\n", gb); continue; }
            hipMemset(p, 1, n); hipDeviceSynchronize(); double t2 = now();
            hipMemset(p, 2, n); hipDeviceSynchronize(); double t3 = now();
            hipFree(p); double t4 = now();
            printf("rep %d: %3zu GB: hipMalloc %8.1f ms, first memset %8.1f ms, second memset %8.1f ms, hipFree %8.1f ms\n", rep, gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
        }
    { std::vector<void*> v; double t0 = now(); for (int i = 0; i < 32; ++i) { void* p; hipHostMalloc(&p, 32u << 20); v.push_back(p); } double t1 = now();
      printf("32 x hipHostMalloc(32 MB): %.1f ms\n", (t1 - t0) * 1e3); for (void* p : v) hipHostFree(p); }
    { void* p; double t0 = now(); hipMalloc(&p, (size_t)20 << 30); double t1 = now(); printf("hipMalloc 20 GB again: %.1f ms\n", (t1 - t0) * 1e3);
      void* q; t0 = now(); hipMalloc(&q, (size_t)8 << 30); t1 = now(); printf("hipMalloc 8 GB beside it: %.1f ms\n", (t1 - t0) * 1e3); }
    return 0;
}
