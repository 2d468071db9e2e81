// Single-wavefront probe of the float64 pieces of the zone averages on gfx950: dependent v_add_f64 chains, the scalar loop
// around a lone add, IEEE division, 64-bit integer multiply-adds.  Cycles per item, third run (instruction cache warm).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/probe_f64.hip -o tools/micro/probe_f64 && tools/micro/probe_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define TIMED(idx, per, body)                                        \
    for (int rep_ = 0; rep_ < 3; ++rep_) {                           \
        const unsigned long long t0_ = __builtin_readcyclecounter(); \
        body;                                                        \
        const unsigned long long t1_ = __builtin_readcyclecounter(); \
        if (threadIdx.x == 0) out[idx] = (t1_ - t0_) / (per);        \
    }

__global__ void probe(unsigned long long* out, double* sink, int n, double seed) {
    const int lane = threadIdx.x & 63;
    double acc = seed, v = seed * 0.5 + lane;
    // 64 dependent adds, straight line
    TIMED(0, 64, {
        _Pragma("unroll") for (int i = 0; i < 64; ++i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc) : "v"(v));
    })
    // 64 dependent adds, scalar loop with one add per trip (n = 64, not known to the compiler)
    TIMED(1, 64, {
        for (int i = n; i > 0; --i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc) : "v"(v));
    })
    // the same with the addend in an SGPR pair
    const double sv = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(v) >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(v)));
    TIMED(2, 64, {
        for (int i = n; i > 0; --i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc) : "s"(sv));
    })
    // independent adds (4 accumulators)
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    TIMED(3, 64, {
        _Pragma("unroll") for (int i = 0; i < 16; ++i)
            asm volatile("v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(v));
    })
    // IEEE divisions, dependent
    double q = seed + 3.0;
    TIMED(4, 16, {
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {
            q = (q + 1.5) / v;
            asm volatile("" : "+v"(q));
        }
    })
    // v_mad_u64_u32 chain
    unsigned long long m = (unsigned long long)lane;
    unsigned x = 12345u + lane;
    TIMED(5, 64, {
        _Pragma("unroll") for (int i = 0; i < 64; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(m) : "v"(x) : "vcc");
    })
    // dependent FMA f64
    TIMED(6, 64, {
        _Pragma("unroll") for (int i = 0; i < 64; ++i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(acc) : "v"(v));
    })
    // dependent v_add_f32 for comparison
    float f = (float)seed;
    TIMED(7, 64, {
        _Pragma("unroll") for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f) : "v"((float)v));
    })
    sink[threadIdx.x] = acc + a0 + a1 + a2 + a3 + q + (double)m + f;
}

int main() {
    unsigned long long* d_out;
    double* d_sink;
    (void)hipMalloc(&d_out, 128);
    (void)hipMalloc(&d_sink, 64 * 8);
    const char* names[8] = {"dependent v_add_f64, straight", "dependent v_add_f64, 1 per loop trip", "same, SGPR addend",
                            "independent v_add_f64 x4", "dependent IEEE f64 division", "dependent v_mad_u64_u32",
                            "dependent v_fma_f64", "dependent v_add_f32"};
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_sink, 64, 1.25);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%-40s %llu cycles\n", names[i], h[i]);
    return 0;
}
