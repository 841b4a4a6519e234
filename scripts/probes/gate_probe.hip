// Probe: what a cross-queue dependency costs inside a replayed hipGraph, as an event edge and as a device-side gate.
//   E  events:  A: K1, record e1, K2, record e2, K3, wait eJ        B: wait e1, K4, wait e2, K5, record eJ
//   G  gates:   A: K1 (counts into c1), K2 (c2), K3, gate(cJ)        B: gate(c1), K4, gate(c2), K5 (cJ)      -- no graph edge between A and B
//   N  nothing: the two chains without any dependency (lower bound; what the kernels alone take)
// A gate is one wave that polls a counter the producer's workgroups add to after a release fence, with a BOUNDED spin
// (timeout -> status word, never a hang).  Kernels are fixed-duration spins (wall clock), 512 x 256 threads, each workgroup
// also writes 16 KB so that the release has something to write back.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/gate_probe scripts/probes/gate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

constexpr int WGS = 512, THREADS = 256;

__global__ __launch_bounds__(THREADS) void work(float* buf, long long ticks, unsigned* done) {
    const long long t0 = wall_clock64();
    float x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1.0001f + 1.f;
    float4* dst = reinterpret_cast<float4*>(buf + (size_t)blockIdx.x * 4096) + threadIdx.x;  // 16 KB per workgroup
    *dst = float4{x, x, x, x};
    if (done != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();  // release (agent scope): this workgroup's stores are visible before the count
            __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// one wave; lane 0 polls.  status[0] counts timeouts.
__global__ __launch_bounds__(64) void gate(unsigned* cnt, unsigned expected, long long timeout_ticks, unsigned* status) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout_ticks) {
            atomicAdd(status, 1u);
            return;
        }
    }
    __hip_atomic_fetch_sub(cnt, expected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // consumed: ready for the next step
    __threadfence();
}

struct Ctx {
    hipStream_t A, B;
    float *bufA, *bufB;
    unsigned *c1, *c2, *cJ, *status;
    std::vector<hipEvent_t> ev;
    size_t next_ev = 0;
    hipEvent_t event() { return ev[next_ev++ % ev.size()]; }
};

static const long long T1 = 2000, T2 = 2500, T3 = 3500, T4 = 4000, T5 = 2500;  // 100 MHz ticks: 20 / 25 / 35 us on A, 40 / 25 on B
static const long long TIMEOUT = 200000;                                        // 2 ms

static void step_events(Ctx& c) {
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T1, (unsigned*)nullptr);
    hipEvent_t e1 = c.event();
    CHECK(hipEventRecord(e1, c.A));
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T2, (unsigned*)nullptr);
    CHECK(hipStreamWaitEvent(c.B, e1, 0));
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.B, c.bufB, T4, (unsigned*)nullptr);
    hipEvent_t e2 = c.event();
    CHECK(hipEventRecord(e2, c.A));
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T3, (unsigned*)nullptr);
    CHECK(hipStreamWaitEvent(c.B, e2, 0));
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.B, c.bufB, T5, (unsigned*)nullptr);
    hipEvent_t eJ = c.event();
    CHECK(hipEventRecord(eJ, c.B));
    CHECK(hipStreamWaitEvent(c.A, eJ, 0));
}

static void step_gates(Ctx& c, bool join_by_gate) {
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T1, c.c1);
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T2, c.c2);
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T3, (unsigned*)nullptr);
    hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, c.B, c.c1, (unsigned)WGS, TIMEOUT, c.status);
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.B, c.bufB, T4, (unsigned*)nullptr);
    hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, c.B, c.c2, (unsigned)WGS, TIMEOUT, c.status);
    if (join_by_gate) {
        hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.B, c.bufB, T5, c.cJ);
        hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, c.A, c.cJ, (unsigned)WGS, TIMEOUT, c.status);
    } else {
        hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.B, c.bufB, T5, (unsigned*)nullptr);
        hipEvent_t eJ = c.event();
        CHECK(hipEventRecord(eJ, c.B));
        CHECK(hipStreamWaitEvent(c.A, eJ, 0));
    }
}

static void step_nothing(Ctx& c) {
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T1, (unsigned*)nullptr);
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T2, (unsigned*)nullptr);
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, T3, (unsigned*)nullptr);
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.B, c.bufB, T4, (unsigned*)nullptr);
    hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.B, c.bufB, T5, (unsigned*)nullptr);
}

static void step_serial(Ctx& c) {
    const long long t[5] = {T1, T2, T3, T4, T5};
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(work, dim3(WGS), dim3(THREADS), 0, c.A, c.bufA, t[i], (unsigned*)nullptr);
}

template <class F>
static double run(const char* name, Ctx& c, int S, F step, bool uses_B) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(c.A, hipStreamCaptureModeThreadLocal));
    if (uses_B) {  // B joins the capture once, at the head
        hipEvent_t e0 = c.event();
        CHECK(hipEventRecord(e0, c.A));
        CHECK(hipStreamWaitEvent(c.B, e0, 0));
    }
    for (int s = 0; s < S; ++s) step(c);
    if (uses_B) {  // and is joined at the tail (a no-op edge when the step already joined)
        hipEvent_t eT = c.event();
        CHECK(hipEventRecord(eT, c.B));
        CHECK(hipStreamWaitEvent(c.A, eT, 0));
    }
    CHECK(hipStreamEndCapture(c.A, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ge, c.A));
    CHECK(hipStreamSynchronize(c.A));
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    const int R = 20;
    CHECK(hipEventRecord(t0, c.A));
    for (int i = 0; i < R; ++i) CHECK(hipGraphLaunch(ge, c.A));
    CHECK(hipEventRecord(t1, c.A));
    CHECK(hipStreamSynchronize(c.A));
    float ms;
    CHECK(hipEventElapsedTime(&ms, t0, t1));
    unsigned st = 0;
    CHECK(hipMemcpy(&st, c.status, 4, hipMemcpyDeviceToHost));
    const double us = ms * 1e3 / (R * S);
    printf("%-46s %8.2f us per step   gate timeouts %u\n", name, us, st);
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    return us;
}

int main() {
    Ctx c;
    CHECK(hipStreamCreateWithFlags(&c.A, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&c.B, hipStreamNonBlocking));
    CHECK(hipMalloc(&c.bufA, (size_t)WGS * 4096 * 4));
    CHECK(hipMalloc(&c.bufB, (size_t)WGS * 4096 * 4));
    unsigned* words;
    CHECK(hipMalloc(&words, 4 * 256));
    CHECK(hipMemset(words, 0, 4 * 256));
    c.c1 = words; c.c2 = words + 64; c.cJ = words + 128; c.status = words + 192;  // separate cache lines
    c.ev.resize(512);
    for (auto& e : c.ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int S = 20;
    printf("chains: A = 20 + 25 + 35 us, B = 40 (after A's first) + 25 (after A's second), join; dependency-bound critical path 85 us\n");
    run("S  one queue, five kernels in line", c, S, step_serial, false);
    run("N  two queues, no dependencies (wrong, bound)", c, S, step_nothing, true);
    run("E  event edges", c, S, step_events, true);
    run("G  gates, join by event", c, S, [](Ctx& c) { step_gates(c, false); }, true);
    run("G2 gates, join by gate (no edge at all)", c, S, [](Ctx& c) { step_gates(c, true); }, true);
    return 0;
}
