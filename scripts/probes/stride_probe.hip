// Translation-reach probe for gfx950 (development tool): one wave per CU issues 32 independent 4-byte-per-lane loads
// `stride` apart (fresh lines every time) and times the batch with s_memtime; a second pass repeats the SAME addresses
// (lines now cached in L2: what is left is translation + L2 latency).  Last: the cold batch again, launched right behind a
// kernel that has just written 0 / 16 / 64 / 256 MB somewhere else (its dirty lines drain to memory at the kernel boundary).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/stride_probe.hip -o /tmp/stride_probe && /tmp/stride_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void toucher(float* x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = 1.f;
}
__global__ void reader(const float* x, size_t n, size_t stride_f, size_t base_f, long long* out, float* sink) {
    const int wg = blockIdx.x, lane = threadIdx.x;
    const size_t i0 = (base_f + (size_t)wg * 64u * 3u) % (n - 33 * stride_f - 64) + lane;
    float t[32], acc = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = x[i0 + (size_t)u * stride_f];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += t[u];
    const size_t i1 = i0 + ((int)acc & 0);
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = __builtin_nontemporal_load(x + i1 + (size_t)u * stride_f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += t[u];
    if (lane == 0) { out[wg * 2] = t1 - t0; out[wg * 2 + 1] = t2 - t1; }
    if (acc == -1.f) *sink = acc;
}
int main() {
    const size_t n = (size_t)256 << 20;  // 1 GB of floats
    float *x, *sink; long long* out; const int nwg = 256;
    (void)hipMalloc(&x, n * 4); (void)hipMalloc(&sink, 4); (void)hipMalloc(&out, nwg * 2 * sizeof(long long));
    hipLaunchKernelGGL(toucher, dim3(4096), dim3(256), 0, 0, x, n);
    std::vector<long long> h(nwg * 2);
    const size_t strides[] = {256, 2048, 4096, 16384, 65536, 262144, 1048576, 2097152, 4194304};
    size_t base = 0;
    for (size_t sb : strides) {
        base += 40u << 20;  // a fresh 160 MB region per case
        hipLaunchKernelGGL(reader, dim3(nwg), dim3(64), 0, 0, x, n, sb / 4, base, out, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::vector<long long> a, b;
        for (int w = 0; w < nwg; ++w) { a.push_back(h[w * 2]); b.push_back(h[w * 2 + 1]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("stride %8zu B: 32 loads cold median %6lld cycles   repeated median %6lld cycles\n", sb, a[nwg / 2], b[nwg / 2]);
    }
    // the same batch of loads (clean data, 64 KB stride) right behind a kernel that dirtied `mb` MB elsewhere
    float* y;
    (void)hipMalloc(&y, (size_t)512 << 20);
    for (size_t mb : {0, 16, 64, 256}) {
        base += 40u << 20;
        if (mb) hipLaunchKernelGGL(toucher, dim3(4096), dim3(256), 0, 0, y, (mb << 20) / 4);
        hipLaunchKernelGGL(reader, dim3(nwg), dim3(64), 0, 0, x, n, (size_t)65536 / 4, base % ((size_t)200 << 20), out, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::vector<long long> a;
        for (int w = 0; w < nwg; ++w) a.push_back(h[w * 2]);
        std::sort(a.begin(), a.end());
        printf("behind a kernel that wrote %3zu MB: 32 cold loads median %6lld cycles (p90 %6lld)\n", mb, a[nwg / 2], a[nwg * 9 / 10]);
    }
    return 0;
}
