// Can a bandwidth-bound kernel (k_accept-like) and an ALU-bound kernel (k_props-like) overlap on two streams?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <chrono>
__global__ void __launch_bounds__(256) bw(const double* __restrict__ a, double* __restrict__ f, size_t n, double s)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        f[i] = __builtin_nontemporal_load(&f[i]) + __builtin_nontemporal_load(&a[i]) * s;     // 24 B per element
}
__global__ void __launch_bounds__(256, 4) alu(const double* __restrict__ x, double* __restrict__ o, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = x[i], acc = 0;
        #pragma unroll 1
        for (int k = 0; k < 5; ++k) acc += pow(v + k * 1e-3, 1.3 + 0.01 * k);
        o[i] = acc;
    }
}
int main()
{
    const size_t nb = 60u << 20, na = 5u << 20;           // bw: 60 M x 24 B = 1.5 GB; alu: 5 M x 5 pow
    double *a, *f, *x, *o;
    hipMalloc(&a, nb * 8); hipMalloc(&f, nb * 8); hipMalloc(&x, na * 8); hipMalloc(&o, na * 8);
    hipMemset(a, 0, nb * 8); hipMemset(f, 0, nb * 8);
    std::vector<double> hx(na); for (size_t i = 0; i < na; ++i) hx[i] = 0.1 + (i % 1000) * 1e-3;
    hipMemcpy(x, hx.data(), na * 8, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    auto ms = [&](hipEvent_t p, hipEvent_t q) { float t; hipEventElapsedTime(&t, p, q); return t; };
    for (int bwBlocks : {2048, 1024, 512, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s1); bw<<<bwBlocks, 256, 0, s1>>>(a, f, nb, 0.5); hipEventRecord(e1, s1);
            alu<<<2048, 256, 0, s1>>>(x, o, na); hipEventRecord(e2, s1); hipEventSynchronize(e2);
            const float tb = ms(e0, e1), ta = ms(e1, e2);
            hipDeviceSynchronize();
            hipEventRecord(e0, s1);
            bw<<<bwBlocks, 256, 0, s2>>>(a, f, nb, 0.5); hipEventRecord(e1, s2);
            alu<<<2048, 256, 0, s1>>>(x, o, na); hipEventRecord(e2, s1);
            hipEventSynchronize(e1); hipEventSynchronize(e2);
            hipDeviceSynchronize();
            // wall of the concurrent pair: host clock is simplest
            if (rep) {
                auto t0 = std::chrono::steady_clock::now();
                bw<<<bwBlocks, 256, 0, s2>>>(a, f, nb, 0.5); alu<<<2048, 256, 0, s1>>>(x, o, na);
                hipDeviceSynchronize();
                auto t1 = std::chrono::steady_clock::now();
                auto t2 = std::chrono::steady_clock::now();
                bw<<<bwBlocks, 256, 0, s1>>>(a, f, nb, 0.5); alu<<<2048, 256, 0, s1>>>(x, o, na);
                hipDeviceSynchronize();
                auto t3 = std::chrono::steady_clock::now();
                printf("bw grid %4d: bw alone %.3f ms (%.0f GB/s), alu alone %.3f ms, sequential wall %.3f ms, two streams wall %.3f ms\n", bwBlocks, tb, nb * 24 / tb / 1e6, ta,
                       std::chrono::duration<double, std::milli>(t3 - t2).count(), std::chrono::duration<double, std::milli>(t1 - t0).count());
            }
        }
    }
    return 0;
}
