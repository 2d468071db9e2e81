// Single-wavefront issue-rate / latency probe for gfx950 (what bounds a one-wave chain kernel).  Every test runs three
// times and reports the last run (instruction cache warm).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/probe_issue.hip -o tools/micro/probe_issue && tools/micro/probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define TIMED(idx, per, body)                                    \
    for (int rep_ = 0; rep_ < 3; ++rep_) {                       \
        const unsigned long long t0_ = __builtin_readcyclecounter(); \
        body;                                                    \
        const unsigned long long t1_ = __builtin_readcyclecounter(); \
        if (threadIdx.x == 0) out[idx] = (t1_ - t0_) / (per);    \
    }

__global__ void probe(unsigned long long* out, int* sink, int waves_active) {
    __shared__ int lds[4096];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 7 + 1) & 4095;
    __syncthreads();
    if (wave >= waves_active) return;
    int v = lane, s = 0, a = lane, b = lane + 1, c = lane + 2, d = lane + 3, p = lane, q = 5, r = lane;
    unsigned long long m = 0;
    TIMED(0, 1, for (int j = 0; j < 32; ++j) {
        asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n"
                     "v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n"
                     "v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n"
                     "v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(v));
    })
    TIMED(1, 1, for (int j = 0; j < 32; ++j) {
        asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                     "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                     "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                     "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    })
    TIMED(2, 1, for (int j = 0; j < 32; ++j) {
        asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                     "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                     "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                     "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(s));
    })
    TIMED(3, 64, for (int i = 0; i < 64; ++i) p = lds[p];)
    TIMED(4, 64, for (int i = 0; i < 64; ++i) q = __builtin_amdgcn_readfirstlane(lds[q]);)
    TIMED(5, 1, for (int j = 0; j < 32; ++j) {
        asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n"
                     "v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n"
                     "v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n"
                     "v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1" : "+v"(v), "+s"(s));
    })
    TIMED(6, 64, for (int i = 0; i < 64; ++i) {
        m = __ballot(r > i);
        r += __builtin_amdgcn_readlane(r, (int)(__ffsll(m) - 1) & 63) & 1;
    })
    TIMED(7, 1, ;)
    // v_cmp -> vcc -> s_cbranch round trip
    TIMED(8, 64, for (int i = 0; i < 64; ++i) {
        if (__ballot(r > i * 3) == 0) r += 1;
        asm volatile("" : "+v"(r));
    })
    sink[threadIdx.x] = v + a + b + c + d + s + p + q + r + (int)m;
}

int main() {
    unsigned long long* d_out;
    int* d_sink;
    (void)hipMalloc(&d_out, 128);
    (void)hipMalloc(&d_sink, 4096 * 4);
    const char* names[9] = {"dependent v_add x512", "independent v_add x512", "dependent s_add x512", "LDS pointer chase / hop",
                            "LDS uniform chase / hop", "alternating v/s x512", "ballot-ffs-readlane / iter", "s_memtime pair",
                            "ballot-branch / iter"};
    for (int block_waves : {1, 4, 16}) {
        for (int active : {1, block_waves}) {
            (void)hipMemset(d_out, 0, 128);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64 * block_waves), 0, 0, d_out, d_sink, active);
            (void)hipDeviceSynchronize();
            unsigned long long h[9];
            (void)hipMemcpy(h, d_out, 72, hipMemcpyDeviceToHost);
            printf("block of %d waves, %d running:\n", block_waves, active);
            for (int i = 0; i < 9; ++i) printf("  %-28s %llu cycles\n", names[i], h[i]);
            if (active == block_waves) break;
        }
    }
    return 0;
}
