// What does a kernel boundary cost when it is replaced by a grid-wide barrier INSIDE one persistent kernel?
// (70 % of the step is the cost of its seventeen kernel boundaries; the alternative is one launch whose
// workgroups hand tiles from layer to layer.)  Two measurements on a cooperative launch (all workgroups co-resident):
//   1. barrier only: atomic arrival counter at agent scope + polling, per iteration;
//   2. barrier + exchange: every workgroup writes `bytes` of its own buffer, release fence, barrier, acquire fence, reads the
//      buffer of the workgroup half a grid away (another XCD: consecutive workgroups are dealt round robin over the 8 XCDs) and
//      checks it -- what a consumer layer would do with its producer's tile.
// Every poll loop has a bounded spin count and sets an error flag instead of hanging.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int MAX_SPINS = 4000000;

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > MAX_SPINS) {
                atomicExch(err, 1u);
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

// Two levels: the workgroups of one XCD (blockIdx % 8: workgroups are dealt round robin over the XCDs) count themselves on their own
// cache line, the last of each XCD counts on the global line and polls it (8 pollers), then releases its XCD's flag (<= gridDim/8 pollers
// per line).  state: [0..7] arrivals per XCD, [8] global, [9..16] release flags, one 128-byte line each.
constexpr int LINE = 32;  // uint32 per 128-byte line
__device__ __forceinline__ bool grid_barrier2(unsigned* state, unsigned epoch, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned xcd = blockIdx.x & 7, n_x = (gridDim.x + 7 - xcd) / 8;  // workgroups of this XCD
        unsigned* arrive = state + xcd * LINE;
        unsigned* global = state + 8 * LINE;
        unsigned* flag = state + (9 + xcd) * LINE;
        int spins = 0;
        const unsigned prev = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == epoch * n_x) {  // last of this XCD
            __hip_atomic_fetch_add(global, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned n_xcd = gridDim.x < 8 ? gridDim.x : 8;
            while (__hip_atomic_load(global, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * n_xcd) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > MAX_SPINS) { atomicExch(err, 1u); ok = false; break; }
            }
            __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > MAX_SPINS) { atomicExch(err, 1u); ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ void barrier2_only(unsigned* state, unsigned* err, int iters) {
    for (int it = 0; it < iters; ++it)
        if (!grid_barrier2(state, (unsigned)(it + 1), err)) return;
}

// the exchange without cache-wide fences: the tile is written and read with agent-scope (write-through / L2-bypassing) 8-byte accesses
__global__ void barrier2_exchange_uncached(unsigned* state, unsigned* err, int iters, uint64_t* buf, int qwords, unsigned* mismatches) {
    const unsigned nb = gridDim.x;
    const unsigned peer = (blockIdx.x + nb / 2) % nb;
    uint64_t* mine = buf + (size_t)blockIdx.x * qwords;
    uint64_t* theirs = buf + (size_t)peer * qwords;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t tag = (uint64_t)(it + 1) * 0x9E3779B97F4A7C15ull;
        for (int i = threadIdx.x; i < qwords; i += blockDim.x) __hip_atomic_store(mine + i, tag + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier2(state, (unsigned)(2 * it + 1), err)) return;
        uint64_t acc = 0;
        for (int i = threadIdx.x; i < qwords; i += blockDim.x) acc |= __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ (tag + i);
        if (acc) ++bad;
        if (!grid_barrier2(state, (unsigned)(2 * it + 2), err)) return;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Point to point: workgroup `a` and workgroup `b` of the grid bounce a flag (one 128-byte line each way), the rest exit at once.
// One round trip = two producer -> consumer handoffs.
__global__ void ping_pong(unsigned* state, unsigned* err, int iters, int a, int b) {
    if (threadIdx.x != 0 || ((int)blockIdx.x != a && (int)blockIdx.x != b)) return;
    unsigned* ping = state;
    unsigned* pong = state + LINE;
    const bool first = (int)blockIdx.x == a;
    for (int it = 1; it <= iters; ++it) {
        int spins = 0;
        if (first) {
            __hip_atomic_store(ping, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(pong, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it)
                if (++spins > MAX_SPINS) { atomicExch(err, 1u); return; }
        } else {
            while (__hip_atomic_load(ping, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it)
                if (++spins > MAX_SPINS) { atomicExch(err, 1u); return; }
            __hip_atomic_store(pong, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void barrier_only(unsigned* counter, unsigned* err, int iters) {
    const unsigned nb = gridDim.x;
    for (int it = 0; it < iters; ++it)
        if (!grid_barrier(counter, (unsigned)(it + 1) * nb, err)) return;
}

// buf: [grid][words] uint32; iteration `it` writes it * 2654435761 + index
__global__ void barrier_exchange(unsigned* counter, unsigned* err, int iters, uint32_t* buf, int words, unsigned* mismatches) {
    const unsigned nb = gridDim.x;
    const unsigned peer = (blockIdx.x + nb / 2) % nb;
    uint4* mine = reinterpret_cast<uint4*>(buf + (size_t)blockIdx.x * words);
    const uint4* theirs = reinterpret_cast<const uint4*>(buf + (size_t)peer * words);
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t tag = (uint32_t)(it + 1) * 2654435761u;
        for (int i = threadIdx.x; i < words / 4; i += blockDim.x) mine[i] = uint4{tag + 4 * i, tag + 4 * i + 1, tag + 4 * i + 2, tag + 4 * i + 3};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (!grid_barrier(counter, (unsigned)(2 * it + 1) * nb, err)) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        uint32_t acc = 0;
        for (int i = threadIdx.x; i < words / 4; i += blockDim.x) {
            const uint4 v = theirs[i];
            acc |= (v.x ^ (tag + 4 * i)) | (v.y ^ (tag + 4 * i + 1)) | (v.z ^ (tag + 4 * i + 2)) | (v.w ^ (tag + 4 * i + 3));
        }
        if (acc) ++bad;
        // second barrier: nobody overwrites its buffer before its reader is done
        if (!grid_barrier(counter, (unsigned)(2 * it + 2) * nb, err)) return;
    }
    if (bad) atomicAdd(mismatches, bad);
}

static int time_kernel(const void* fn, int grid, int threads, void** args, float* ms) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    CHECK(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(threads), args, 0, 0));
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(ms, a, b));
    return 0;
}

int main() {
    unsigned *counter, *err, *mism;
    CHECK(hipMalloc(&counter, 4));
    CHECK(hipMalloc(&err, 4));
    CHECK(hipMalloc(&mism, 4));
    unsigned* state;
    CHECK(hipMalloc(&state, 17 * 128));
    const int max_grid = 1024, max_words = 64 * 1024 / 4;
    uint32_t* buf;
    CHECK(hipMalloc(&buf, (size_t)max_grid * max_words * 4));
    CHECK(hipMemset(err, 0, 4));
    CHECK(hipMemset(mism, 0, 4));
    {
        const int pairs[][2] = {{0, 1}, {0, 4}, {0, 8}, {0, 16}};  // (workgroup b on another XCD: 1, 4; on the same XCD as workgroup 0: 8, 16)
        for (auto& pr : pairs) {
            float t0 = 0.f, t1 = 0.f;
            int a = pr[0], b = pr[1];
            for (int rep = 0; rep < 2; ++rep) {
                int iters = 0;
                void* ap[] = {&state, &err, &iters, &a, &b};
                CHECK(hipMemset(state, 0, 17 * 128));
                if (time_kernel(reinterpret_cast<const void*>(&ping_pong), 32, 64, ap, &t0)) return 1;
                iters = 2000;
                CHECK(hipMemset(state, 0, 17 * 128));
                if (time_kernel(reinterpret_cast<const void*>(&ping_pong), 32, 64, ap, &t1)) return 1;
            }
            printf("flag ping-pong between workgroups %d and %d: %.2f us per round trip (two handoffs)\n", a, b, (t1 - t0) * 1e3f / 2000);
        }
    }
    const int grids[] = {256, 512, 1024};
    for (int grid : grids) {
        const int threads = 256;
        float t0 = 0.f, t1 = 0.f;
        for (int rep = 0; rep < 2; ++rep) {  // (first repetition warms up)
            int iters = 0;
            void* a0[] = {&counter, &err, &iters};
            CHECK(hipMemset(counter, 0, 4));
            if (time_kernel(reinterpret_cast<const void*>(&barrier_only), grid, threads, a0, &t0)) return 1;
            iters = 2000;
            CHECK(hipMemset(counter, 0, 4));
            if (time_kernel(reinterpret_cast<const void*>(&barrier_only), grid, threads, a0, &t1)) return 1;
        }
        printf("grid %4d x %d threads: barrier only           %.2f us per barrier\n", grid, threads, (t1 - t0) * 1e3f / 2000);
        for (int rep = 0; rep < 2; ++rep) {
            int iters = 0;
            void* a2[] = {&state, &err, &iters};
            CHECK(hipMemset(state, 0, 17 * 128));
            if (time_kernel(reinterpret_cast<const void*>(&barrier2_only), grid, threads, a2, &t0)) return 1;
            iters = 2000;
            CHECK(hipMemset(state, 0, 17 * 128));
            if (time_kernel(reinterpret_cast<const void*>(&barrier2_only), grid, threads, a2, &t1)) return 1;
        }
        printf("grid %4d x %d threads: two-level barrier only %.2f us per barrier\n", grid, threads, (t1 - t0) * 1e3f / 2000);
        const int sizes[] = {4 * 1024, 32 * 1024, 64 * 1024};
        for (int bytes : sizes) {
            int qwords = bytes / 8;
            uint64_t* buf64 = reinterpret_cast<uint64_t*>(buf);
            for (int rep = 0; rep < 2; ++rep) {
                int iters = 0;
                void* a3[] = {&state, &err, &iters, &buf64, &qwords, &mism};
                CHECK(hipMemset(state, 0, 17 * 128));
                if (time_kernel(reinterpret_cast<const void*>(&barrier2_exchange_uncached), grid, threads, a3, &t0)) return 1;
                iters = 500;
                CHECK(hipMemset(state, 0, 17 * 128));
                if (time_kernel(reinterpret_cast<const void*>(&barrier2_exchange_uncached), grid, threads, a3, &t1)) return 1;
            }
            printf("grid %4d x %d threads: two-level, agent-scope stores / loads of %2d KB, no cache-wide fences   %.2f us per round\n", grid, threads,
                   bytes / 1024, (t1 - t0) * 1e3f / 500);
        }
        for (int bytes : sizes) {
            int words = bytes / 4;
            for (int rep = 0; rep < 2; ++rep) {
                int iters = 0;
                void* a1[] = {&counter, &err, &iters, &buf, &words, &mism};
                CHECK(hipMemset(counter, 0, 4));
                if (time_kernel(reinterpret_cast<const void*>(&barrier_exchange), grid, threads, a1, &t0)) return 1;
                iters = 500;
                CHECK(hipMemset(counter, 0, 4));
                if (time_kernel(reinterpret_cast<const void*>(&barrier_exchange), grid, threads, a1, &t1)) return 1;
            }
            printf("grid %4d x %d threads: write %2d KB, barrier, read the peer's, barrier   %.2f us per round (%.1f MB moved each way)\n", grid,
                   threads, bytes / 1024, (t1 - t0) * 1e3f / 500, (double)grid * bytes / 1e6);
        }
    }
    unsigned h_err = 0, h_m = 0;
    CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&h_m, mism, 4, hipMemcpyDeviceToHost));
    printf("spin limit reached: %u   stale reads seen by workgroups: %u\n", h_err, h_m);
    return (h_err || h_m) ? 2 : 0;
}
