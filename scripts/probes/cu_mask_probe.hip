// Probe: what a CU-masked HIP stream (hipExtStreamCreateWithCUMask) does on MI355X.
//   1. census: which (XCC, SE, CU) run the workgroups of a launch on a stream with a given mask;
//   2. partition: do two streams with disjoint masks run their kernels side by side, each at the speed of its own share?
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/cu_mask_probe scripts/probes/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void census(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (xcc & 15u) << 16 | ((hw >> 13) & 7u) << 8 | ((hw >> 12) & 1u) << 4 | ((hw >> 8) & 15u);
    }
    // stay resident a little so that the launch spreads over every CU it may use
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}

__global__ void spin(long long cycles, float* sink) {
    long long t0 = __builtin_amdgcn_s_memtime();
    float x = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) x = x * 1.0001f + 1.f;
    if (x == 123.f) sink[0] = x;
}

static void run_census(const char* name, hipStream_t st, unsigned* d_out, int blocks) {
    std::vector<unsigned> h(blocks);
    hipLaunchKernelGGL(census, dim3(blocks), dim3(64), 0, st, d_out);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(h.data(), d_out, blocks * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    for (unsigned v : h) cus.insert(v);
    for (unsigned v : cus) per_xcc[(v >> 16) & 15]++;
    printf("%-34s distinct (xcc,se,sh,cu) = %3zu   per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2d", per_xcc[x]);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, multiProcessorCount %d\n", prop.name, ncu);
    const int words = (ncu + 31) / 32;
    unsigned* d_out;
    float* d_sink;
    const int blocks = 8192;
    CHECK(hipMalloc(&d_out, blocks * 4));
    CHECK(hipMalloc(&d_sink, 4));
    hipStream_t plain;
    CHECK(hipStreamCreate(&plain));
    run_census("plain stream", plain, d_out, blocks);

    auto masked = [&](const std::vector<unsigned>& m) {
        hipStream_t s;
        CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
        return s;
    };
    auto range_mask = [&](int lo, int hi) {  // bits [lo, hi)
        std::vector<unsigned> m(words, 0u);
        for (int b = lo; b < hi; ++b) m[b / 32] |= 1u << (b % 32);
        return m;
    };
    auto stride_mask = [&](int phase, int stride) {  // bits b with b % stride == phase
        std::vector<unsigned> m(words, 0u);
        for (int b = phase; b < ncu; b += stride) m[b / 32] |= 1u << (b % 32);
        return m;
    };
    {
        hipStream_t s = masked(range_mask(0, 32));
        run_census("mask bits [0,32)", s, d_out, blocks);
        std::vector<unsigned> back(words, 0u);
        hipError_t e = hipExtStreamGetCUMask(s, words, back.data());
        printf("  hipExtStreamGetCUMask -> %s:", hipGetErrorString(e));
        for (int w = 0; w < words; ++w) printf(" %08x", back[w]);
        printf("\n");
        CHECK(hipStreamDestroy(s));
    }
    { hipStream_t s = masked(range_mask(32, 64)); run_census("mask bits [32,64)", s, d_out, blocks); CHECK(hipStreamDestroy(s)); }
    { hipStream_t s = masked(range_mask(0, ncu / 2)); run_census("mask bits [0,ncu/2)", s, d_out, blocks); CHECK(hipStreamDestroy(s)); }
    { hipStream_t s = masked(range_mask(ncu / 2, ncu)); run_census("mask bits [ncu/2,ncu)", s, d_out, blocks); CHECK(hipStreamDestroy(s)); }
    { hipStream_t s = masked(stride_mask(0, 8)); run_census("mask bits b%8==0", s, d_out, blocks); CHECK(hipStreamDestroy(s)); }
    { hipStream_t s = masked(stride_mask(3, 8)); run_census("mask bits b%8==3", s, d_out, blocks); CHECK(hipStreamDestroy(s)); }
    { hipStream_t s = masked(stride_mask(0, 2)); run_census("mask bits b%2==0", s, d_out, blocks); CHECK(hipStreamDestroy(s)); }

    // partition: 2048 workgroups x 256 threads spinning 20 us each, on one stream alone and on two masked streams side by side
    auto time_pair = [&](const char* name, hipStream_t a, hipStream_t b, int blocks_each) {
        hipEvent_t e0, e1, f0, f1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1));
        const long long cyc = 2000;  // s_memtime ticks at 100 MHz: 20 us
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, a));
        hipLaunchKernelGGL(spin, dim3(blocks_each), dim3(256), 0, a, cyc, d_sink);
        CHECK(hipEventRecord(e1, a));
        if (b) {
            CHECK(hipEventRecord(f0, b));
            hipLaunchKernelGGL(spin, dim3(blocks_each), dim3(256), 0, b, cyc, d_sink);
            CHECK(hipEventRecord(f1, b));
        }
        CHECK(hipDeviceSynchronize());
        float ta = 0, tb = 0;
        CHECK(hipEventElapsedTime(&ta, e0, e1));
        if (b) CHECK(hipEventElapsedTime(&tb, f0, f1));
        printf("%-44s stream A %.1f us   stream B %.1f us\n", name, ta * 1e3f, tb * 1e3f);
    };
    hipStream_t lo = masked(range_mask(0, ncu / 2)), hi = masked(range_mask(ncu / 2, ncu));
    hipStream_t ev = masked(stride_mask(0, 2)), od = masked(stride_mask(1, 2));
    hipStream_t plain2;
    CHECK(hipStreamCreate(&plain2));
    for (int rep = 0; rep < 2; ++rep) {
        time_pair("plain alone, 4096 wg", plain, nullptr, 4096);
        time_pair("plain + plain, 2048 wg each", plain, plain2, 2048);
        time_pair("half mask alone, 2048 wg", lo, nullptr, 2048);
        time_pair("lower half + upper half, 2048 wg each", lo, hi, 2048);
        time_pair("even bits + odd bits, 2048 wg each", ev, od, 2048);
    }
    return 0;
}
