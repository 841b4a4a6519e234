// How long does an EMPTY kernel take as a function of its grid, workgroup size, dynamic LDS and register footprint?
// (The image-resident conv kernels ask for 40-140 KB of LDS per workgroup; measured inside the step, their empty
// launches take 3-5 us.)  Back-to-back launches on one stream, timed with events; prints us per launch.
//   hipcc --offload-arch=gfx950 -O3 -o launch_cost launch_cost.hip && ./launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

extern __shared__ char smem[];

template <int VG>
__global__ void empty_kernel(float* out, int never) {
    // VG live registers (kept alive by a store that never happens)
    float v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
    if (never) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VG; ++i) s += v[i] * (float)never;
        out[threadIdx.x] = s + smem[threadIdx.x];
    }
}

template <int VG>
static int run(const char* tag, int grid, int threads, int lds, float* out) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_kernel<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int n = 2000;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel<VG>, dim3(grid), dim3(threads), lds, 0, out, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel<VG>, dim3(grid), dim3(threads), lds, 0, out, 0);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&empty_kernel<VG>), threads, lds));
    printf("%-8s grid %5d x %3d threads  LDS %6d B  -> %d wg/CU : %6.2f us per launch\n", tag, grid, threads, lds, occ, ms * 1e3f / n);
    return 0;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 4096));
    const int ldss[] = {0, 16 * 1024, 37 * 1024, 64 * 1024, 78 * 1024, 110 * 1024, 141 * 1024};
    for (int lds : ldss)
        for (int grid : {256, 512, 1024, 2048}) {
            if (run<8>("vgpr8", grid, 256, lds, out)) return 1;
        }
    for (int lds : {37 * 1024, 78 * 1024, 141 * 1024}) {
        if (run<8>("vgpr8", 512, 512, lds, out)) return 1;
        if (run<120>("vgpr120", 512, 256, lds, out)) return 1;
        if (run<200>("vgpr200", 512, 256, lds, out)) return 1;
    }
    return 0;
}
