// feasibility probe: two processes, fine-grained device windows exchanged by hipIpc, device-side
// flag ping-pong with system-scope atomics (bounded spins).  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("[%d] %s failed: %s\n", getpid(), #x, hipGetErrorString(e)); exit(2); } } while (0)
struct Win { unsigned long long flag[2]; double payload[1024]; };
__global__ void pingpong(Win* mine, Win* peer, int rank, int iters, unsigned long long* result)
{
    unsigned long long fails = 0;
    for (int it = 1; it <= iters; ++it) {
        if ((it & 1) == rank) {   // my turn to send
            for (int k = threadIdx.x; k < 1024; k += blockDim.x) peer->payload[k] = it * 1000.0 + k;
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&peer->flag[0], (unsigned long long)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            if (threadIdx.x == 0) {
                long long t0 = wall_clock64();
                while (__hip_atomic_load(&mine->flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)it) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > 300000000LL) { fails = 1ull << 40; break; }   // 3 s at 100 MHz
                }
            }
            __syncthreads();
            for (int k = threadIdx.x; k < 1024; k += blockDim.x)
                if (__hip_atomic_load(&mine->payload[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != it * 1000.0 + k) atomicAdd(result + 1, 1ull);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) result[0] = fails;
}
int main(int argc, char** argv)
{
    int p2c[2], c2p[2];
    pipe(p2c); pipe(c2p);
    pid_t pid = fork();                       // fork BEFORE any HIP call
    int rank = pid == 0 ? 1 : 0;
    int rd = rank == 0 ? c2p[0] : p2c[0], wr = rank == 0 ? p2c[1] : c2p[1];
    CK(hipSetDevice(0));
    Win* mine = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&mine, sizeof(Win), hipDeviceMallocFinegrained);
    printf("[%d] finegrained alloc: %s\n", rank, hipGetErrorString(e));
    if (e != hipSuccess) CK(hipMalloc((void**)&mine, sizeof(Win)));
    CK(hipMemset(mine, 0, sizeof(Win)));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h, hp;
    CK(hipIpcGetMemHandle(&h, mine));
    write(wr, &h, sizeof(h)); read(rd, &hp, sizeof(hp));
    Win* peer = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
    unsigned long long* res; CK(hipMalloc((void**)&res, 16)); CK(hipMemset(res, 0, 16));
    char c = 1; write(wr, &c, 1); read(rd, &c, 1);            // both opened
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipEventRecord(a);
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(256), 0, 0, mine, peer, rank, iters, res);
    hipEventRecord(b);
    CK(hipDeviceSynchronize());
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    unsigned long long out[2]; CK(hipMemcpy(out, res, 16, hipMemcpyDeviceToHost));
    printf("[%d] timeout=%llu payload_errors=%llu  %.2f us per one-way hop\n", rank, out[0] >> 40, out[1], ms * 1e3 / iters);
    write(wr, &c, 1); read(rd, &c, 1);
    hipIpcCloseMemHandle(peer);
    if (rank == 0) { int st; waitpid(pid, &st, 0); }
    return (out[0] || out[1]) ? 1 : 0;
}
