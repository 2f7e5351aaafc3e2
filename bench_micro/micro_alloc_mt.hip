// dev tool (round 6): does fresh device memory come faster from several host threads at once?  (the CLI's first job waits ~4 s for ~86 GB)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    for (int nt : {1, 2, 4, 8}) {
        const size_t total = (size_t)48 << 30, each = total / nt;
        std::vector<void*> p(nt, nullptr); std::vector<std::thread> th;
        const double t0 = now();
        for (int i = 0; i < nt; ++i) th.emplace_back([&, i] { hipSetDevice(0); if (hipMalloc(&p[i], each) != hipSuccess) p[i] = nullptr; });
        for (auto& t : th) t.join();
        const double t1 = now();
        printf("%d thread(s) x %zu GB: %.1f ms (%.1f ms per GB)\n", nt, each >> 30, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / 48.0);
        // (kept allocated: freed blocks come back from the runtime's cache in microseconds and would hide the cost of the next round)
    }
    return 0;
}
