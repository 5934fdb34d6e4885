// What does a software grid barrier cost on an MI355X against a kernel boundary inside a replayed hipGraph?  (small grids.)
// K iterations of "touch a few cache lines" either as K launches of one graph or as ONE launch with a barrier per iteration
// (agent-scope release, atomic arrive, bounded spin on an agent-scope load, acquire - the barrier of the retired one-launch step).
// build: hipcc --offload-arch=gfx950 -O3 -o build_variants/barrier_probe scripts/experiments/barrier_probe.cpp ; run: build_variants/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void work(double* x, int n, int it)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = x[i] * 0.999 + it;
}
__global__ void k_one(double* x, int n, int it) { work(x, n, it); }
__global__ void k_loop(double* x, int n, int iters, unsigned int* bar, int* failed)
{
    __shared__ int ok;
    unsigned int gen = 0;
    for (int it = 0; it < iters; ++it) {
        work(x, n, it);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            gen += gridDim.x;
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int good = 1;
            const long long t0 = wall_clock64();
            while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gen) < 0) {
                if (wall_clock64() - t0 > 100000000LL) { good = 0; break; }        // 1 s at 100 MHz
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ok = good;
        }
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) *failed = 1; return; }
    }
}
int main()
{
    const int n = 40960, iters = 200;
    double* x; unsigned int* bar; int* failed;
    CHECK(hipMalloc(&x, n * sizeof(double))); CHECK(hipMemset(x, 0, n * sizeof(double)));
    CHECK(hipMalloc(&bar, 64)); CHECK(hipMalloc(&failed, 4));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int blocks : {16, 32, 64, 160}) {
        // (1) a graph of `iters` launches
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_one, dim3(blocks), dim3(256), 0, st, x, n, it);
        CHECK(hipStreamEndCapture(st, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best1 = 1e9f, best2 = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(a, st)); CHECK(hipGraphLaunch(ge, st)); CHECK(hipEventRecord(b, st)); CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best1) best1 = ms;
        }
        // (2) one launch, a barrier per iteration
        int h = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipMemsetAsync(bar, 0, 64, st)); CHECK(hipMemsetAsync(failed, 0, 4, st));
            CHECK(hipEventRecord(a, st));
            hipLaunchKernelGGL(k_loop, dim3(blocks), dim3(256), 0, st, x, n, iters, bar, failed);
            CHECK(hipEventRecord(b, st)); CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best2) best2 = ms;
            CHECK(hipMemcpy(&h, failed, 4, hipMemcpyDeviceToHost));
            if (h) break;
        }
        printf("blocks %4d: graph of %d launches %.2f us per launch; one launch with a grid barrier per iteration %.2f us per iteration%s\n",
               blocks, iters, best1 * 1e3 / iters, best2 * 1e3 / iters, h ? "  (BARRIER TIMED OUT)" : "");
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
