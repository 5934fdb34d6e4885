// Is ocml's powr (base known to be >= 0) bit-identical to pow on the value ranges of the soil functions, and is it faster?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
extern "C" __device__ double __ocml_powr_f64(double, double);
__global__ void k(const double* x, const double* y, double* a, double* b, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = pow(x[i], y[i]); b[i] = __ocml_powr_f64(x[i], y[i]); }
}
template <int W> __global__ void t(const double* x, const double* y, double* o, int n, int reps)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xv = x[i], yv = y[i], acc = 0;
    for (int r = 0; r < reps; ++r) { acc += W ? __ocml_powr_f64(xv, yv) : pow(xv, yv); xv += 1e-9; }
    o[i] = acc;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), y(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); };
    for (int i = 0; i < n; ++i) {
        const int kind = i & 3;
        if (kind == 0) { x[i] = std::exp(rnd() * 20 - 14); y[i] = 1.05 + rnd() * 1.0; }          // (alpha psi)^n
        else if (kind == 1) { x[i] = 1.0 + std::exp(rnd() * 20 - 14); y[i] = -(0.05 + rnd() * 0.5); }   // (1 + t)^-m
        else if (kind == 2) { x[i] = rnd(); y[i] = 1.0 / (0.05 + rnd() * 0.5); }                  // Se^(1/m)
        else { x[i] = rnd() * 1e-3 + (rnd() < 0.5 ? 0 : rnd()); y[i] = 0.05 + rnd() * 0.5; }      // (1 - x)^m
    }
    double *dx, *dy, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dy, da, db, n);
    std::vector<double> a(n), b(n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    long diff = 0; double worst = 0;
    for (int i = 0; i < n; ++i) if (memcmp(&a[i], &b[i], 8)) { ++diff; worst = std::fmax(worst, std::fabs(a[i] - b[i]) / std::fabs(a[i])); }
    printf("pow vs powr: %ld of %d differ (worst rel %.3e)\n", diff, n, worst);
    long diffh = 0; double worsth = 0;
    for (int i = 0; i < n; ++i) { double h = std::pow(x[i], y[i]); if (memcmp(&a[i], &h, 8)) { ++diffh; worsth = std::fmax(worsth, std::fabs(a[i] - h) / std::fabs(h)); } }
    printf("device pow vs glibc pow: %ld of %d differ (worst rel %.3e)\n", diffh, n, worsth);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (w) t<1><<<n / 256, 256>>>(dx, dy, da, n, 16); else t<0><<<n / 256, 256>>>(dx, dy, da, n, 16);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.3f ms for %d x 16 evaluations = %.1f G/s\n", w ? "powr" : "pow ", ms, n, n * 16.0 / ms / 1e6);
    }
    return 0;
}
