// What a host load / store of device-mapped pinned memory costs (the resident worker's ticket ring and completion words live in
// hipHostMalloc(Mapped | Coherent) memory): hipcc tools/micro/probe_pinned_host.cpp -o /tmp/probe_pinned && /tmp/probe_pinned
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main() {
    using clk = std::chrono::steady_clock;
    const size_t n = 1 << 16;
    for (int kind = 0; kind < 3; ++kind) {
        unsigned long long* p = nullptr;
        const char* name = kind == 0 ? "malloc" : (kind == 1 ? "hipHostMalloc(Mapped|Coherent)" : "hipHostMalloc(Mapped)");
        if (kind == 0) p = (unsigned long long*)malloc(n * 8);
        else if (hipHostMalloc((void**)&p, n * 8, kind == 1 ? (hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocMapped) != hipSuccess) return 1;
        for (size_t i = 0; i < n; ++i) p[i] = i;
        volatile unsigned long long* v = p;
        auto t0 = clk::now();
        unsigned long long s = 0;
        for (int r = 0; r < 200; ++r)
            for (size_t i = 0; i < n; i += 8) s += __atomic_load_n(p + i, __ATOMIC_ACQUIRE);
        auto t1 = clk::now();
        for (int r = 0; r < 200; ++r)
            for (size_t i = 0; i < n; i += 8) __atomic_store_n(p + i, s + i, __ATOMIC_RELEASE);
        auto t2 = clk::now();
        const double nl = 200.0 * n / 8;
        printf("%-34s load %.1f ns  store %.1f ns  (one per 64-byte line, %zu KiB)\n", name,
               std::chrono::duration<double, std::nano>(t1 - t0).count() / nl, std::chrono::duration<double, std::nano>(t2 - t1).count() / nl, n * 8 / 1024);
        (void)v;
    }
    return 0;
}
