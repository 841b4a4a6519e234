// Load-latency probe for gfx950 (development tool, not part of the library).
// Kernel W writes a buffer from every CU; kernel R (one wave per CU, launched right behind W on the same stream) times
// with s_memtime: (0) its first load of a line W wrote from ANOTHER workgroup, (1) a dependent load 8 MB further (fresh
// page, fresh line), (2) the same line again (vector L1 hit), (3) a line its own XCD neighbour (id + 8) just read
// (L2 hit), and (4) 32 independent loads 1 MB apart issued back to back (one batch), (5)/(6) the same 2 KB apart with 4- / 16-byte loads.  Prints medians in shader cycles.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void writer(float* x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = (float)(i & 1023);
}

__global__ void reader(const float* x, size_t n, long long* out, float* sink) {
    const int wg = blockIdx.x, lane = threadIdx.x;
    // a line written by a workgroup far away (the writer strides by the grid, so every line is foreign)
    size_t i0 = ((size_t)wg * 1048583u + 4099u * 64u) % (n / 2) + lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    float v0 = __builtin_nontemporal_load(x + i0) ;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    size_t i1 = (i0 + (size_t)(2u << 20) + ((int)v0 & 1)) % n;  // dependent on v0; +8 MB
    float v1 = x[i1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = __builtin_amdgcn_s_memtime();
    float v2 = x[i1 + ((int)v1 & 0)];  // same line again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t3 = __builtin_amdgcn_s_memtime();
    // line that workgroup wg+8 (same XCD) touched as its i1
    size_t j0 = ((size_t)(wg + 8) * 1048583u + 4099u * 64u) % (n / 2) + lane;
    size_t j1 = (j0 + (size_t)(2u << 20)) % n;
    float v3 = x[j1 + ((int)v2 & 0)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t4 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    float t[32];
    // (address arithmetic outside the timed region: a 64-bit modulo per load costs more than the load)
    const size_t b4 = (i0 + (size_t)(3u << 18) + 77u * 64u + ((int)v3 & 0)) % (n - ((size_t)33 << 18));
    long long t4b = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = x[b4 + (size_t)u * (1u << 18)];  // 1 MB apart
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t5 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += t[u];
    // (5) 32 independent loads 2 KB apart (one 64 KB block), fresh lines; (6) the same with 16-byte loads
    const size_t k0 = (i0 + (size_t)(40u << 18) + 13u * 64u + ((int)acc & 0)) % (n - (1u << 16));
    t5 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = x[k0 + (size_t)u * 512];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t6 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += t[u];
    const size_t m0 = ((i0 + (size_t)(50u << 18) + ((int)acc & 0)) % (n - (1u << 16))) / 4 * 4 - lane + (size_t)lane * 4;
    t6 = __builtin_amdgcn_s_memtime();
    float4 q[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) q[u] = *reinterpret_cast<const float4*>(x + m0 + (size_t)u * 512);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t7 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += q[u].x + q[u].w;
    if (lane == 0) {
        out[wg * 8 + 5] = t6 - t5; out[wg * 8 + 6] = t7 - t6;
        out[wg * 8 + 0] = t1 - t0; out[wg * 8 + 1] = t2 - t1; out[wg * 8 + 2] = t3 - t2; out[wg * 8 + 3] = t4 - t3;
        out[wg * 8 + 4] = t5 - t4b;
    }
    if (acc + v0 + v1 + v2 + v3 == -1.f) *sink = acc;
}

int main() {
    const size_t n = (size_t)64 << 20;  // 256 MB of floats
    float *x, *sink;
    long long* out;
    const int nwg = 256;
    hipMalloc(&x, n * 4); hipMalloc(&sink, 4); hipMalloc(&out, nwg * 8 * sizeof(long long));
    std::vector<long long> h(nwg * 8);
    const char* names[7] = {"first touch of a line written by the previous kernel", "dependent load, +8 MB (fresh page)",
                            "same line again (L1)", "line read by workgroup id+8 (same XCD: L2)", "32 independent loads 1 MB apart", "32 independent loads 2 KB apart (one 64 KB block)",
                            "32 independent 16-byte loads 2 KB apart"};
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, x, n);
        hipLaunchKernelGGL(reader, dim3(nwg), dim3(64), 0, 0, x, n, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        printf("rep %d\n", rep);
        for (int k = 0; k < 7; ++k) {
            std::vector<long long> v;
            for (int w = 0; w < nwg; ++w) v.push_back(h[w * 8 + k]);
            std::sort(v.begin(), v.end());
            printf("  %-58s p10 %6lld  median %6lld  p90 %6lld cycles\n", names[k], v[nwg / 10], v[nwg / 2], v[nwg * 9 / 10]);
        }
    }
    return 0;
}
