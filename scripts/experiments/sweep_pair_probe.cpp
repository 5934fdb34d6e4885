// Feasibility probe for DESIGN.md 10 "Sweeps": two Jacobi iterations per pass over the coefficient stream on a regular
// NX x NY x NZ grid (layer-major numbering like the product's), against two passes of a k_sweep-like kernel.
// A block owns a patch of W rows x 64 columns and marches through the layers: stage A computes x' of layer t for the whole
// patch (halo included) into an LDS ring of three layers, stage B computes x'' of layer t-1 for the inner (W-2) x 62 cells
// from the ring, with the coefficients kept in registers from the step before.  Checks x'' bit for bit against two
// single sweeps and times both.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/experiments/sweep_pair_probe.cpp -o build_variants/sweep_pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include <cstdlib>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

struct alignas(16) d2 { double x, y; };
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ntload(const d2* p) { const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)); d2 r; r.x = t.x; r.y = t.y; return r; }
constexpr int SLOTS = 10;
__constant__ int cDelta[SLOTS];          // index offsets of the ten neighbours (0 up, 1 down, 2..9 laterals)
constexpr int ORDER[SLOTS] = {0, 2, 3, 4, 5, 6, 7, 8, 9, 1};
constexpr int DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

struct Grid { int NX, NY, NZ; size_t N; const d2* A2; const double *b, *z; };

// one Jacobi sweep, one lane per node, wave64 chunks grid-stride (the product's k_sweep without norm / decisions)
__global__ void __launch_bounds__(256) k_single(Grid g, const double* __restrict__ xin, double* __restrict__ xout)
{
    const size_t chunks = g.N / 64;
    const uint32_t lane = threadIdx.x & 63u;
    for (size_t q = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); q < chunks; q += (size_t)gridDim.x * 4) {
        const size_t i = q * 64 + lane;
        double a[SLOTS], xj[SLOTS];
        #pragma unroll
        for (int p = 0; p < 5; ++p) { const d2 t = ntload(&g.A2[(size_t)p * g.N + i]); a[2 * p] = t.x; a[2 * p + 1] = t.y; }
        const double bi = g.b[i], zi = g.z[i];
        #pragma unroll
        for (int s = 0; s < SLOTS; ++s) xj[s] = (a[s] != 0.) ? xin[i + cDelta[s]] : 0.;
        double xn = bi;
        #pragma unroll
        for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (a[s] != 0.) xn -= a[s] * xj[s]; }
        if (i < (size_t)g.NX * g.NY) xn = (xn < zi) ? zi : xn;
        xout[i] = xn;
    }
}

// k_single + what the product's k_sweep does on top: the node's own x and the scaled norm, a block reduction, a last-block
// hand-off (two-level counter) with a fixed-order sum of the partials; LIST: chunk list + per-chunk descriptor indirection
struct Desc { int delta[SLOTS]; int pad[6]; };
template <bool LIST>
__global__ void __launch_bounds__(256, 8) k_single_full(Grid g, const double* __restrict__ xin, double* __restrict__ xout,
                                                      const uint32_t* __restrict__ list, const Desc* __restrict__ desc,
                                                      double* part, unsigned* arrive, double* result)
{
    const size_t chunks = g.N / 64;
    const uint32_t lane = threadIdx.x & 63u;
    double nrm = 0.;
    for (size_t li = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); li < chunks; li += (size_t)gridDim.x * 4) {
        const size_t q = LIST ? __builtin_amdgcn_readfirstlane(list[li]) : li;
        const size_t i = q * 64 + lane;
        double a[SLOTS], xj[SLOTS];
        #pragma unroll
        for (int p = 0; p < 5; ++p) { const d2 t = ntload(&g.A2[(size_t)p * g.N + i]); a[2 * p] = t.x; a[2 * p + 1] = t.y; }
        const double bi = g.b[i], zi = g.z[i], xi = xin[i];
        if (LIST) {
            const Desc d = desc[q];
            #pragma unroll
            for (int s = 0; s < SLOTS; ++s) xj[s] = xin[(a[s] != 0.) ? i + d.delta[s] : i];
        } else {
            #pragma unroll
            for (int s = 0; s < SLOTS; ++s) xj[s] = xin[(a[s] != 0.) ? i + cDelta[s] : i];
        }
        double xn = bi;
        #pragma unroll
        for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (a[s] != 0.) xn -= a[s] * xj[s]; }
        if (i < (size_t)g.NX * g.NY) xn = (xn < zi) ? zi : xn;
        double d = fabs(xn - xi); const double psi = fabs(xn - zi); if (psi > 1.) d *= (1. / psi);
        nrm += d;
        xout[i] = xn;
    }
    __shared__ double sm[4]; __shared__ int sLast;
    for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
    if (lane == 0) sm[threadIdx.x >> 6] = nrm;
    __syncthreads();
    if (threadIdx.x == 0) {
        // like the product's arrive_last: an agent-scope relaxed store + wait, no __threadfence() (on this part an agent-scope
        // release writes the XCD's L2 back: with it every block pays for the sweep's dirty lines and the kernel takes twice as long)
        __hip_atomic_store(&part[blockIdx.x], (sm[0] + sm[1]) + (sm[2] + sm[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sLast = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (sLast) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!sLast) return;
    double s2 = 0.;
    for (uint32_t k = threadIdx.x; k < gridDim.x; k += 256) s2 += __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off > 0; off >>= 1) s2 += __shfl_down(s2, off, 64);
    if (lane == 0) sm[threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x == 0) *result = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / g.N;
}

// an fp64-transcendental burst like k_props / k_assemble: does the sweep after it run slower (clocks, power)?
__global__ void __launch_bounds__(256, 4) k_alu(const double* __restrict__ x, double* __restrict__ o, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = x[i] * 1e-2, acc = 0;
        #pragma unroll 1
        for (int k = 0; k < 12; ++k) acc += pow(v + k * 1e-3, 1.3 + 0.01 * k);
        o[i] = acc;
    }
}

// two sweeps per pass.  W waves per block (W rows of the patch), inner rows 1..W-2, inner lanes 1..62.
template <int W, int OCC>
__global__ void __launch_bounds__(W * 64, OCC) k_pair(Grid g, const double* __restrict__ xin, double* __restrict__ xout, int patchCols)
{
    __shared__ double ring[3][W][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pr = blockIdx.x / patchCols, pc = blockIdx.x % patchCols;
    int r0 = pr * (W - 2) - 1, c0 = pc * 62 - 1;
    if (r0 + W > g.NY + 1) r0 = g.NY + 1 - W;          // last patches overlap the previous ones instead of hanging over the edge
    if (c0 + 64 > g.NX + 1) c0 = g.NX + 1 - 64;
    const int r = r0 + wave, c = c0 + lane;
    const bool valid = r >= 0 && r < g.NY && c >= 0 && c < g.NX;
    const bool inner = valid && wave >= 1 && wave <= W - 2 && lane >= 1 && lane <= 62;
    const size_t layer = (size_t)g.NX * g.NY;
    const size_t i0 = valid ? (size_t)r * g.NX + c : 0;
    double ap[SLOTS], bp = 0., zp = 0.;
    #pragma unroll
    for (int s = 0; s < SLOTS; ++s) ap[s] = 0.;
    for (int t = 0; t <= g.NZ; ++t) {
        double ac[SLOTS], bc = 0., zc = 0.;
        #pragma unroll
        for (int s = 0; s < SLOTS; ++s) ac[s] = 0.;
        if (t < g.NZ) {                                                     // stage A: x' of layer t
            double x1 = 0.;
            if (valid) {
                const size_t i = (size_t)t * layer + i0;
                double xj[SLOTS];
                #pragma unroll
                for (int p = 0; p < 5; ++p) { const d2 v = ntload(&g.A2[(size_t)p * g.N + i]); ac[2 * p] = v.x; ac[2 * p + 1] = v.y; }
                bc = g.b[i]; zc = (t == 0) ? g.z[i] : 0.;
                #pragma unroll
                for (int s = 0; s < SLOTS; ++s) xj[s] = (ac[s] != 0.) ? xin[i + cDelta[s]] : 0.;
                x1 = bc;
                #pragma unroll
                for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (ac[s] != 0.) x1 -= ac[s] * xj[s]; }
                if (t == 0) x1 = (x1 < zc) ? zc : x1;
            }
            ring[t % 3][wave][lane] = x1;
        }
        __syncthreads();
        if (t >= 1 && inner) {                                              // stage B: x'' of layer t - 1 from the ring
            const int l = t - 1;
            double xj[SLOTS];
            xj[0] = (ap[0] != 0.) ? ring[(l + 2) % 3][wave][lane] : 0.;     // layer l - 1
            xj[1] = (ap[1] != 0.) ? ring[(l + 1) % 3][wave][lane] : 0.;     // layer l + 1
            #pragma unroll
            for (int k = 0; k < 8; ++k) xj[2 + k] = (ap[2 + k] != 0.) ? ring[l % 3][wave + DR[k]][lane + DC[k]] : 0.;
            double x2 = bp;
            #pragma unroll
            for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (ap[s] != 0.) x2 -= ap[s] * xj[s]; }
            if (l == 0) x2 = (x2 < zp) ? zp : x2;
            xout[(size_t)l * layer + i0] = x2;
        }
        #pragma unroll
        for (int s = 0; s < SLOTS; ++s) ap[s] = ac[s];
        bp = bc; zp = zc;
        __syncthreads();
    }
}

// the same with 64 inner columns per patch: the two edge lanes also compute x' of the halo columns (a second, two-lane pass of
// stage A - the arithmetic is free in a kernel that waits for memory), so that 512 columns are 8 patches, not 8.26 -> 9
template <int W>
__device__ __forceinline__ double sweep_node(const Grid& g, size_t i, int t, const double* __restrict__ xin, double (&ac)[SLOTS], double& bc, double& zc)
{
    double xj[SLOTS];
    #pragma unroll
    for (int p = 0; p < 5; ++p) { const d2 v = ntload(&g.A2[(size_t)p * g.N + i]); ac[2 * p] = v.x; ac[2 * p + 1] = v.y; }
    bc = g.b[i]; zc = (t == 0) ? g.z[i] : 0.;
    #pragma unroll
    for (int s = 0; s < SLOTS; ++s) xj[s] = (ac[s] != 0.) ? xin[i + cDelta[s]] : 0.;
    double x1 = bc;
    #pragma unroll
    for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (ac[s] != 0.) x1 -= ac[s] * xj[s]; }
    if (t == 0) x1 = (x1 < zc) ? zc : x1;
    return x1;
}
template <int W>
__global__ void __launch_bounds__((W + 1) * 64) k_pair64(Grid g, const double* __restrict__ xin, double* __restrict__ xout, int patchCols)
{
    __shared__ double ring[3][W][66];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pr = blockIdx.x / patchCols, pc = blockIdx.x % patchCols;
    int r0 = pr * (W - 2) - 1, c0 = pc * 64;
    if (r0 + W > g.NY + 1) r0 = g.NY + 1 - W;
    if (c0 + 64 > g.NX) c0 = g.NX - 64;
    // waves 0..W-1: one patch row each, 64 inner columns; wave W: the two halo columns of all W rows (lane = 2 * row + side)
    const bool halo = wave == W;
    const int prow = halo ? (lane >> 1) : wave;
    const int r = r0 + prow;
    const int c = halo ? ((lane & 1) ? c0 + 64 : c0 - 1) : c0 + lane;
    const int slot = halo ? ((lane & 1) ? 65 : 0) : lane + 1;
    const bool ok = r >= 0 && r < g.NY && c >= 0 && c < g.NX && prow < W;
    const bool inner = !halo && ok && wave >= 1 && wave <= W - 2;
    const size_t layer = (size_t)g.NX * g.NY;
    const size_t i0 = ok ? (size_t)r * g.NX + c : 0;
    double ap[SLOTS], bp = 0., zp = 0.;
    #pragma unroll
    for (int s = 0; s < SLOTS; ++s) ap[s] = 0.;
    for (int t = 0; t <= g.NZ; ++t) {
        double ac[SLOTS], bc = 0., zc = 0.;
        #pragma unroll
        for (int s = 0; s < SLOTS; ++s) ac[s] = 0.;
        if (t < g.NZ && prow < W) ring[t % 3][prow][slot] = ok ? sweep_node<W>(g, (size_t)t * layer + i0, t, xin, ac, bc, zc) : 0.;
        __syncthreads();
        if (t >= 1 && inner) {
            const int l = t - 1;
            double xj[SLOTS];
            xj[0] = (ap[0] != 0.) ? ring[(l + 2) % 3][wave][lane + 1] : 0.;
            xj[1] = (ap[1] != 0.) ? ring[(l + 1) % 3][wave][lane + 1] : 0.;
            #pragma unroll
            for (int k = 0; k < 8; ++k) xj[2 + k] = (ap[2 + k] != 0.) ? ring[l % 3][wave + DR[k]][lane + 1 + DC[k]] : 0.;
            double x2 = bp;
            #pragma unroll
            for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (ap[s] != 0.) x2 -= ap[s] * xj[s]; }
            if (l == 0) x2 = (x2 < zp) ? zp : x2;
            xout[(size_t)l * layer + i0] = x2;
        }
        #pragma unroll
        for (int s = 0; s < SLOTS; ++s) ap[s] = ac[s];
        bp = bc; zp = zc;
        __syncthreads();
    }
}

// k_pair64 with everything the product would need around it: z for every node, the node's own x, x' stored for the inner cells
// (so that a convergence hit after the first of the two iterations can use it), both scaled norms reduced per block and
// summed by the last block (fence-free hand-off)
template <int W>
__global__ void __launch_bounds__((W + 1) * 64, 6) k_pair64_full(Grid g, const double* __restrict__ xin, double* __restrict__ xout1, double* __restrict__ xout,
                                                             int patchCols, double* part, unsigned* arrive, double* result)
{
    __shared__ double ring[3][W][66];
    __shared__ double sm[2][W + 1]; __shared__ int sLast;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pr = blockIdx.x / patchCols, pc = blockIdx.x % patchCols;
    int r0 = pr * (W - 2) - 1, c0 = pc * 64;
    if (r0 + W > g.NY + 1) r0 = g.NY + 1 - W;
    if (c0 + 64 > g.NX) c0 = g.NX - 64;
    const bool halo = wave == W;
    const int prow = halo ? (lane >> 1) : wave;
    const int r = r0 + prow;
    const int c = halo ? ((lane & 1) ? c0 + 64 : c0 - 1) : c0 + lane;
    const int slot = halo ? ((lane & 1) ? 65 : 0) : lane + 1;
    const bool ok = r >= 0 && r < g.NY && c >= 0 && c < g.NX && prow < W;
    const bool inner = !halo && ok && wave >= 1 && wave <= W - 2;
    const size_t layer = (size_t)g.NX * g.NY;
    const size_t i0 = ok ? (size_t)r * g.NX + c : 0;
    double ap[SLOTS], bp = 0., zp = 0., n1 = 0., n2 = 0.;
    #pragma unroll
    for (int s = 0; s < SLOTS; ++s) ap[s] = 0.;
    for (int t = 0; t <= g.NZ; ++t) {
        double ac[SLOTS], bc = 0., zc = 0.;
        #pragma unroll
        for (int s = 0; s < SLOTS; ++s) ac[s] = 0.;
        if (t < g.NZ && prow < W) {
            double x1 = 0.;
            if (ok) {
                const size_t i = (size_t)t * layer + i0;
                double xj[SLOTS];
                #pragma unroll
                for (int p = 0; p < 5; ++p) { const d2 v = ntload(&g.A2[(size_t)p * g.N + i]); ac[2 * p] = v.x; ac[2 * p + 1] = v.y; }
                bc = g.b[i]; zc = g.z[i];
                const double xi = xin[i];
                #pragma unroll
                for (int s = 0; s < SLOTS; ++s) xj[s] = (ac[s] != 0.) ? xin[i + cDelta[s]] : 0.;
                x1 = bc;
                #pragma unroll
                for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (ac[s] != 0.) x1 -= ac[s] * xj[s]; }
                if (t == 0) x1 = (x1 < zc) ? zc : x1;
                if (inner) {
                    double d = fabs(x1 - xi); const double psi = fabs(x1 - zc); if (psi > 1.) d *= (1. / psi);
                    n1 += d;
                    xout1[i] = x1;
                }
            }
            ring[t % 3][prow][slot] = x1;
        }
        __syncthreads();
        if (t >= 1 && inner) {
            const int l = t - 1;
            double xj[SLOTS];
            xj[0] = (ap[0] != 0.) ? ring[(l + 2) % 3][wave][lane + 1] : 0.;
            xj[1] = (ap[1] != 0.) ? ring[(l + 1) % 3][wave][lane + 1] : 0.;
            #pragma unroll
            for (int k = 0; k < 8; ++k) xj[2 + k] = (ap[2 + k] != 0.) ? ring[l % 3][wave + DR[k]][lane + 1 + DC[k]] : 0.;
            const double x1 = ring[l % 3][wave][lane + 1];
            double x2 = bp;
            #pragma unroll
            for (int o = 0; o < SLOTS; ++o) { const int s = ORDER[o]; if (ap[s] != 0.) x2 -= ap[s] * xj[s]; }
            if (l == 0) x2 = (x2 < zp) ? zp : x2;
            double d = fabs(x2 - x1); const double psi = fabs(x2 - zp); if (psi > 1.) d *= (1. / psi);
            n2 += d;
            xout[(size_t)l * layer + i0] = x2;
        }
        #pragma unroll
        for (int s = 0; s < SLOTS; ++s) ap[s] = ac[s];
        bp = bc; zp = zc;
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) { n1 += __shfl_down(n1, off, 64); n2 += __shfl_down(n2, off, 64); }
    if (lane == 0) { sm[0][wave] = n1; sm[1][wave] = n2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s1 = 0., s2 = 0.;
        for (int w = 0; w <= W; ++w) { s1 += sm[0][w]; s2 += sm[1][w]; }
        __hip_atomic_store(&part[blockIdx.x], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&part[2048 + blockIdx.x], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sLast = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (sLast) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!sLast) return;
    double s1 = 0., s2 = 0.;
    for (uint32_t k = threadIdx.x; k < gridDim.x; k += blockDim.x) { s1 += __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s2 += __hip_atomic_load(&part[2048 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    __syncthreads();
    if (lane == 0) { sm[0][wave] = s1; sm[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) { double t1 = 0., t2 = 0.; for (int w = 0; w <= W; ++w) { t1 += sm[0][w]; t2 += sm[1][w]; } result[0] = t1 / g.N; result[1] = t2 / g.N; }
}

int main(int argc, char** argv)
{
    const int NX = argc > 1 ? atoi(argv[1]) : 512, NY = argc > 2 ? atoi(argv[2]) : 512, NZ = 20;
    const size_t N = (size_t)NX * NY * NZ, layer = (size_t)NX * NY;
    std::vector<d2> A2(5 * N);
    std::vector<double> b(N), z(N), x0(N);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.01, 0.09);
    int delta[SLOTS];
    delta[0] = -(int)layer; delta[1] = (int)layer;
    for (int k = 0; k < 8; ++k) delta[2 + k] = DR[k] * NX + DC[k];
    for (int l = 0; l < NZ; ++l) for (int r = 0; r < NY; ++r) for (int c = 0; c < NX; ++c) {
        const size_t i = (size_t)l * layer + (size_t)r * NX + c;
        double a[SLOTS];
        a[0] = (l > 0) ? -U(rng) : 0.; a[1] = (l < NZ - 1) ? -U(rng) : 0.;
        for (int k = 0; k < 8; ++k) { const int rr = r + DR[k], cc = c + DC[k]; a[2 + k] = (rr >= 0 && rr < NY && cc >= 0 && cc < NX) ? -U(rng) * 0.1 : 0.; }
        for (int p = 0; p < 5; ++p) A2[(size_t)p * N + i] = {a[2 * p], a[2 * p + 1]};
        b[i] = 100. + U(rng); z[i] = 100.; x0[i] = 100. + 0.001 * ((i * 2654435761u) % 1000);
    }
    d2* dA; double *db, *dz, *dx0, *dx1, *dx2, *dy;
    CHECK(hipMalloc(&dA, 5 * N * sizeof(d2))); CHECK(hipMalloc(&db, N * 8)); CHECK(hipMalloc(&dz, N * 8));
    CHECK(hipMalloc(&dx0, N * 8)); CHECK(hipMalloc(&dx1, N * 8)); CHECK(hipMalloc(&dx2, N * 8)); CHECK(hipMalloc(&dy, N * 8));
    CHECK(hipMemcpy(dA, A2.data(), 5 * N * sizeof(d2), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), N * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dz, z.data(), N * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dx0, x0.data(), N * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(cDelta), delta, sizeof(delta)));
    CHECK(hipMemset(dy, 0, N * 8));
    Grid g{NX, NY, NZ, N, dA, db, dz};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto&& launch, int reps) { launch(); hipDeviceSynchronize(); hipEventRecord(e0, 0); for (int k = 0; k < reps; ++k) launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / reps; };

    const float tSingle = timeit([&] { k_single<<<2048, 256>>>(g, dx0, dx1); k_single<<<2048, 256>>>(g, dx1, dx2); }, 20);
    printf("%d x %d x %d: two single sweeps: %.1f us (%.1f us each)\n", NX, NY, NZ, tSingle, tSingle / 2);
    {
        std::vector<uint32_t> hl(N / 64); for (size_t q = 0; q < N / 64; ++q) hl[q] = (uint32_t)q;
        std::vector<Desc> hd(N / 64); for (auto& d : hd) for (int s = 0; s < SLOTS; ++s) d.delta[s] = delta[s];
        uint32_t* dl; Desc* dd; double *dpart, *dres; unsigned* darr;
        CHECK(hipMalloc(&dl, hl.size() * 4)); CHECK(hipMalloc(&dd, hd.size() * sizeof(Desc))); CHECK(hipMalloc(&dpart, 4096 * 8)); CHECK(hipMalloc(&dres, 8)); CHECK(hipMalloc(&darr, 4));
        CHECK(hipMemcpy(dl, hl.data(), hl.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dd, hd.data(), hd.size() * sizeof(Desc), hipMemcpyHostToDevice)); CHECK(hipMemset(darr, 0, 4));
        const float tA = timeit([&] { k_single_full<false><<<2048, 256>>>(g, dx0, dx1, dl, dd, dpart, darr, dres); k_single_full<false><<<2048, 256>>>(g, dx1, dx2, dl, dd, dpart, darr, dres); }, 20);
        const float tB = timeit([&] { k_single_full<true><<<2048, 256>>>(g, dx0, dx1, dl, dd, dpart, darr, dres); k_single_full<true><<<2048, 256>>>(g, dx1, dx2, dl, dd, dpart, darr, dres); }, 20);
        printf("  + own x, norm, block reduction, last-block sum: %.1f us each;  + chunk list and descriptor: %.1f us each\n", tA / 2, tB / 2);
    }
    {   // sweeps timed one by one right after an ALU burst of ~0.4 ms, like the product's approximation
        hipEvent_t a0, a1; hipEventCreate(&a0); hipEventCreate(&a1);
        double tot = 0; int cnt = 0; float tb = 0;
        for (int rep = 0; rep < 30; ++rep) {
            hipEventRecord(a0, 0); k_alu<<<2048, 256>>>(dx0, dy, N); hipEventRecord(a1, 0); hipEventSynchronize(a1); hipEventElapsedTime(&tb, a0, a1);
            k_alu<<<2048, 256>>>(dx0, dy, N);
            for (int k = 0; k < 5; ++k) {
                hipEventRecord(a0, 0); k_single<<<2048, 256>>>(g, (k & 1) ? dx1 : dx0, (k & 1) ? dx2 : dx1); hipEventRecord(a1, 0); hipEventSynchronize(a1);
                float t; hipEventElapsedTime(&t, a0, a1); if (rep >= 5 && k >= 1) { tot += t; ++cnt; }
            }
        }
        printf("  single sweeps 2..5 after an fp64 burst of %.0f us, timed one by one with events: %.1f us each\n", tb * 1e3, tot / cnt * 1e3);
        tot = 0; cnt = 0;
        for (int rep = 0; rep < 30; ++rep)
            for (int k = 0; k < 5; ++k) {
                hipEventRecord(a0, 0); k_single<<<2048, 256>>>(g, (k & 1) ? dx1 : dx0, (k & 1) ? dx2 : dx1); hipEventRecord(a1, 0); hipEventSynchronize(a1);
                float t; hipEventElapsedTime(&t, a0, a1); if (rep >= 5) { tot += t; ++cnt; }
            }
        printf("  the same without the burst: %.1f us each\n", tot / cnt * 1e3);
        k_single<<<2048, 256>>>(g, dx0, dx1); k_single<<<2048, 256>>>(g, dx1, dx2); hipDeviceSynchronize();
    }
    std::vector<double> ref(N), got(N);
    CHECK(hipMemcpy(ref.data(), dx2, N * 8, hipMemcpyDeviceToHost));

    auto run_pair = [&](auto kernel, int W, const char* name) {
        const int patchRows = (NY + (W - 2) - 1) / (W - 2), patchCols = (NX + 61) / 62;
        const int blocks = patchRows * patchCols;
        hipMemset(dy, 0, N * 8);
        const float t = timeit([&] { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(W * 64), 0, 0, g, dx0, dy, patchCols); }, 20);
        hipMemcpy(got.data(), dy, N * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < N; ++i) if (memcmp(&got[i], &ref[i], 8) != 0) ++bad;
        printf("%s: %d blocks of %d threads: %.1f us per pair, %zu of %zu values differ from two single sweeps\n", name, blocks, W * 64, t, bad, N);
    };
    run_pair(k_pair<16, 4>, 16, "pair W=16");
    run_pair(k_pair<8, 4>, 8, "pair W=8");
    auto run_pair64 = [&](auto kernel, int W, const char* name) {
        const int patchRows = (NY + (W - 2) - 1) / (W - 2), patchCols = (NX + 63) / 64;
        const int blocks = patchRows * patchCols;
        hipMemset(dy, 0, N * 8);
        const float t = timeit([&] { hipLaunchKernelGGL(kernel, dim3(blocks), dim3((W + 1) * 64), 0, 0, g, dx0, dy, patchCols); }, 20);
        hipMemcpy(got.data(), dy, N * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < N; ++i) if (memcmp(&got[i], &ref[i], 8) != 0) ++bad;
        printf("%s: %d blocks of %d threads: %.1f us per pair, %zu of %zu values differ from two single sweeps\n", name, blocks, (W + 1) * 64, t, bad, N);
    };
    {
        const int W = 10, patchRows = (NY + (W - 2) - 1) / (W - 2), patchCols = (NX + 63) / 64, blocks = patchRows * patchCols;
        double *dpart, *dres, *dx1b; unsigned* darr;
        CHECK(hipMalloc(&dpart, 4096 * 8)); CHECK(hipMalloc(&dres, 16)); CHECK(hipMalloc(&darr, 4)); CHECK(hipMalloc(&dx1b, N * 8)); CHECK(hipMemset(darr, 0, 4));
        if (blocks <= 2048) {
            hipMemset(dy, 0, N * 8);
            const float t = timeit([&] { hipLaunchKernelGGL(k_pair64_full<10>, dim3(blocks), dim3((W + 1) * 64), 0, 0, g, dx0, dx1b, dy, patchCols, dpart, darr, dres); }, 20);
            hipMemcpy(got.data(), dy, N * 8, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < N; ++i) if (memcmp(&got[i], &ref[i], 8) != 0) ++bad;
            std::vector<double> r1(N); hipMemcpy(r1.data(), dx1b, N * 8, hipMemcpyDeviceToHost);
            std::vector<double> s1(N); hipMemcpy(s1.data(), dx1, N * 8, hipMemcpyDeviceToHost);     // dx1 = first single sweep of the reference pair
            size_t bad1 = 0; for (size_t i = 0; i < N; ++i) if (memcmp(&r1[i], &s1[i], 8) != 0) ++bad1;
            double res[2]; hipMemcpy(res, dres, 16, hipMemcpyDeviceToHost);
            printf("pair64 W=10 + z, own x, x' stored, two norms, last-block sum: %d blocks: %.1f us per pair; x'' differs in %zu, x' in %zu values; norms %.6e %.6e\n", blocks, t, bad, bad1, res[0], res[1]);
        }
    }
    run_pair64(k_pair64<10>, 10, "pair64 W=10");
    run_pair64(k_pair64<15>, 15, "pair64 W=15");
    run_pair64(k_pair64<6>, 6, "pair64 W=6");
    return 0;
}
