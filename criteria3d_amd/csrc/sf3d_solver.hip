/*
 * sf3d_solver.hip - HIP/CDNA4 (gfx950) kernels and device-resident step control of the
 * MI355X-native soilFluxes3D water time step.
 *
 * What it computes is the reference's CPU/OpenMP path (cpusolver.cpp:143-190 waterMainLoop,
 * :392-468 waterApproximationLoop, :672-703 solveLinearSystem; water.cpp; soilPhysics.cpp);
 * how it computes it is designed for the MI355X:
 *   - one thread per node over slot-major (coalesced) graph arrays, fp64 throughout, no MFMA
 *     (11-point unstructured stencil); -ffp-contract=off so products and sums round as in the
 *     reference's x86-64 -O2 build;
 *   - properties + boundary fused in one kernel, assembly + diagonal + row normalisation +
 *     Courant maximum fused in one kernel, Jacobi sweep + surface clamp + norm fused in one
 *     kernel, post-solve Se + both mass-balance sums fused in one kernel;
 *   - H / Hold / Hbest are indices into a pool of four head buffers: step begin, rejection,
 *     keep-best and restore-best are index flips, not N-element copies;
 *   - every accept / halve / Courant / convergence decision is taken ON THE DEVICE by a
 *     one-block kernel that reduces the per-block partials in a fixed order (deterministic,
 *     wave64 shuffles + LDS) and advances a stage machine in the control block; every compute
 *     kernel is guarded by that stage, so the host can queue a whole approximation (with a
 *     speculative batch of sweeps) without reading anything back, and polls the control block
 *     once per approximation instead of once per kernel.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      /* types only: the library is dlopen'ed when a multi-GPU run asks for the RCCL exchange */
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "sf3d_model.h"

#define NODATA_D (-9999.0)

/* ======================================================================================= */
/* device helpers                                                                           */
/* ======================================================================================= */

/* Streaming accesses that do not allocate in the caches: the coefficient slabs (80 B/node, read exactly
 * once per sweep by exactly one wave), the static link geometry and the per-link flow sums would
 * otherwise evict x, b and z from L2 / the 256 MiB Infinity Cache between two sweeps.  `NT is a kernel
 * template parameter chosen per launch from DevView::ntStream (a run-time select of the two forms is
 * merged by the compiler into one plain load): off when a rank's whole working set fits the Infinity Cache. */
typedef double sf3d_v2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ sf3d_d2 load_coeff(const sf3d_d2* p)
{
    sf3d_v2 t;
    if (NT) t = __builtin_nontemporal_load(reinterpret_cast<const sf3d_v2*>(p));
    else t = *reinterpret_cast<const sf3d_v2*>(p);
    sf3d_d2 r; r.x = t.x; r.y = t.y; return r;
}
template <bool NT> __device__ __forceinline__ void store_coeff(sf3d_d2* p, double x, double y)
{
    sf3d_v2 t; t.x = x; t.y = y;
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<sf3d_v2*>(p));
    else *reinterpret_cast<sf3d_v2*>(p) = t;
}
template <bool NT, class T> __device__ __forceinline__ T load_stream(const T* p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT, class T> __device__ __forceinline__ void store_stream(T* p, T x)
{
    if (NT) __builtin_nontemporal_store(x, p); else *p = x;
}

/* std::max / std::min semantics (NaN handling included) */
__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }

/* ---- log and pow of the hot kernels: table-driven routines of sf3d_fastmath.inc ----
 * The device library's log / pow are ~90 / ~230 VALU instructions at 1 ulp; the table designs need ~45 / ~85 at 0.51 ulp
 * (measured against mpmath), i.e. they agree with the reference's libm in 99.9 % of the cases instead of most.  The host
 * build of the same text is checked against mpmath and libm (tests/test_fastmath.py) and the device against the host
 * build bit for bit (tests/test_gpu_fastmath.py).  The tables (128 pieces of log, 128 of 2^(j/128): 5 KB) sit in LDS:
 * 64 lanes with 64 different pieces would cost ~64 texture-addresser cycles per vector-memory gather.  Every kernel that
 * evaluates flog() / ppow() calls fm_init() first.  -DSF3D_FAST_LOG=0 / -DSF3D_FAST_POW=0 restore the library calls
 * (pow: ocml's powr, the base is never negative here: saturation degrees, alpha*psi, 1 + t). */
#ifndef SF3D_FAST_LOG
#define SF3D_FAST_LOG 1
#endif
#ifndef SF3D_FAST_POW
#define SF3D_FAST_POW 1
#endif
#define SF3D_FM_FN __device__ __forceinline__
#define SF3D_FM_TABLE __device__ const
struct FmLds { double invc[128], lhi[128], llo[128], ehi[128], elo[128]; };
__device__ __forceinline__ FmLds& fm_lds() { __shared__ FmLds t; return t; }
#define SF3D_FM_LOOKUP(i, e) struct sf3d_flog_entry e; { const FmLds& t_ = fm_lds(); e.invc = t_.invc[i]; e.logc_hi = t_.lhi[i]; e.logc_lo = t_.llo[i]; }
#define SF3D_FM_EXP_LOOKUP(j, e) struct sf3d_fexp_entry e; { const FmLds& t_ = fm_lds(); e.hi = t_.ehi[j]; e.lo = t_.elo[j]; }
#include "sf3d_fastmath.inc"
__device__ __forceinline__ void fm_init()
{
#if SF3D_FAST_LOG || SF3D_FAST_POW
    FmLds& t = fm_lds();
    for (uint32_t k = threadIdx.x; k < 128; k += blockDim.x) {
        t.invc[k] = sf3d_flog_table[k].invc; t.lhi[k] = sf3d_flog_table[k].logc_hi; t.llo[k] = sf3d_flog_table[k].logc_lo;
        t.ehi[k] = sf3d_fexp_table[k].hi; t.elo[k] = sf3d_fexp_table[k].lo;
    }
    __syncthreads();
#endif
}
__device__ __forceinline__ double flog(double x)
{
#if SF3D_FAST_LOG
    return sf3d_flog(x);
#else
    return log(x);
#endif
}
__device__ __forceinline__ double fexp(double x)
{
#if SF3D_FAST_POW
    return sf3d_fexp(x);
#else
    return exp(x);
#endif
}
extern "C" __device__ double __ocml_powr_f64(double, double);
__device__ __forceinline__ double ppow(double x, double y)
{
#if SF3D_FAST_POW
    return sf3d_fpow(x, y);
#else
    return __ocml_powr_f64(x, y);
#endif
}

/* the coefficient matrix of the computeStep in progress */
__device__ __forceinline__ sf3d_d2* cur_A2(const DevView& v) { return v.A2x[v.ctrl->aBuf & 1u]; }

__device__ __forceinline__ int free_buffer(const Ctrl* c)
{
    for (int b = 0; b < SF3D_POOL; ++b)
        if (b != c->cur && b != c->hold && b != c->best) return b;
    return 0;   /* unreachable: at most three of four buffers are in use */
}

/* the two free buffers of the pool of five (paired sweep: x' and x'') */
__device__ __forceinline__ void free_buffers2(const Ctrl* c, int& f1, int& f2)
{
    uint32_t used = (1u << c->cur) | (1u << c->hold);
    if (c->best >= 0) used |= 1u << c->best;
    uint32_t fr = ~used & ((1u << SF3D_POOL) - 1u);      /* at least two bits: three of five buffers are in use at most */
    f1 = __builtin_ctz(fr); fr &= fr - 1u;
    f2 = __builtin_ctz(fr);
}

/* fixed-shape block reductions (256 threads = 4 waves of 64) */
__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double sm[SF3D_BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__device__ __forceinline__ double block_max(double v)
{
    __shared__ double sm[SF3D_BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) v = dmax(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return dmax(dmax(sm[0], sm[1]), dmax(sm[2], sm[3]));
}
/* one block reduces nb partials in a fixed order */
__device__ __forceinline__ double reduce_partials_sum(const double* p, uint32_t nb)
{
    double s = 0.;
    for (uint32_t k = threadIdx.x; k < nb; k += SF3D_BLOCK) s += p[k];
    return block_sum(s);
}
__device__ __forceinline__ double reduce_partials_max(const double* p, uint32_t nb)
{
    double s = 0.;
    for (uint32_t k = threadIdx.x; k < nb; k += SF3D_BLOCK) s = dmax(s, p[k]);
    return block_max(s);
}

/* a / b for the link arithmetic (conductivities, distances, logarithms of conductivity ratios: finite operands of ordinary
 * magnitude).  The compiler's IEEE sequence is 17 instructions - operand scaling, reciprocal seed, two Newton steps, quotient,
 * residual, re-scaling, special-case fix-up - and there are thirty divisions per soil row; this one keeps the seed, the Newton
 * steps and the residual correction (8 instructions, the same <= 1 ulp) and hands everything whose result is not a finite number
 * (b = 0, denormal or infinite operands, nan) to the full sequence, so that the special cases keep the reference's answers - the
 * logarithmic mean of two conductivities one ulp apart really divides by log(1) = 0.
 * OFF by default (-DSF3D_FAST_DIV=1 turns it on): measured at C4, k_assemble 303-305 us with it against 300 us without - the kernel's
 * time is not in its VALU instruction count, so the exactly rounded quotients stay. */
#ifndef SF3D_FAST_DIV
#define SF3D_FAST_DIV 0
#endif
__device__ __forceinline__ double qdiv(double a, double b)
{
#if SF3D_FAST_DIV
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    double q = a * r;
    q = __builtin_fma(__builtin_fma(-b, q, a), r, q);
    if (__builtin_expect(!(fabs(q) < __builtin_inf()), 0)) q = a / b;
    return q;
#else
    return a / b;
#endif
}

/* ---- Math::computeMean (otherFunctions.cpp:7-36) ---- */
/* the logarithm is the table-driven one: every kernel that evaluates a mean has called fm_init() */
__device__ __forceinline__ double mean_of(double v1, double v2, uint32_t type)
{
    if (type == SF3D_MEAN_ARITHMETIC) return (v1 + v2) * 0.5;
    if (type == SF3D_MEAN_GEOMETRIC) { const int sign = (v1 > 0) - (v1 < 0); return sign * sqrt(v1 * v2); }
    return (v1 == v2) ? v1 : qdiv(v1 - v2, flog(qdiv(v1, v2)));
}

/* ---- Soil:: (soilPhysics.cpp) ---- */
__device__ __forceinline__ double se_from_psi(const SoilDev& s, double psi, uint32_t wrc)   /* :91-115 */
{
    if (wrc == SF3D_WRC_VAN_GENUCHTEN) return ppow(1.0 + ppow(s.alpha * psi, s.n), -s.m);
    if (wrc == SF3D_WRC_MODIFIED_VAN_GENUCHTEN) {
        if (psi <= s.he) return 1.0;
        return ppow(1.0 + ppow(s.alpha * psi, s.n), -s.m) * s.invSc;
    }
    return NODATA_D;
}
__device__ __forceinline__ double node_se(const SoilDev& s, double H, double z, uint32_t wrc)  /* :68-83 */
{
    if (H >= z) return 1.;
    return se_from_psi(s, fabs(H - z), wrc);
}
__device__ __forceinline__ double mualem_k(const SoilDev& s, double Se, uint32_t wrc)          /* :181-214 */
{
    if (Se >= 1.0) return s.Ksat;
    double temp;
    if (wrc == SF3D_WRC_VAN_GENUCHTEN) {
        const double sePow = ppow(Se, s.invM);
        temp = 1.0 - ppow(1.0 - sePow, s.m);
    } else if (wrc == SF3D_WRC_MODIFIED_VAN_GENUCHTEN) {
        const double seScPow = ppow(Se * s.Sc, s.invM);
        const double tNum = 1.0 - ppow(1.0 - seScPow, s.m);
        temp = tNum / s.mualemDen;
    } else return NODATA_D;
    /* Mualem tortuosity Se^L: L = 0.5 in every soil table of the application -> correctly rounded sqrt */
    const double seL = (s.L == 0.5) ? sqrt(Se) : ppow(Se, s.L);
    return s.Ksat * seL * (temp * temp);
}
__device__ __forceinline__ double dtheta_dh(const SoilDev& s, double H, double Ho, double z, uint32_t wrc)   /* :224-279 */
{
    const double psiCurr = fabs(dmin(0.0, H - z));
    const double psiPrev = fabs(dmin(0.0, Ho - z));
    if (wrc == SF3D_WRC_VAN_GENUCHTEN) { if (psiCurr == 0.0 && psiPrev == 0.0) return 0.0; }
    else if (wrc == SF3D_WRC_MODIFIED_VAN_GENUCHTEN) { if (psiCurr <= s.he && psiPrev <= s.he) return 0.0; }
    double dSe;
    if (fabs(psiCurr - psiPrev) < 1e-12) {
        const double xx = s.alpha * psiCurr;
        const double onePlus = 1. + ppow(xx, s.n);
        const double t1 = ppow(onePlus, -(s.m + 1.));
        const double t2 = ppow(xx, s.n - 1.);
        dSe = s.alpha * s.n * s.m * t1 * t2;
        if (wrc == SF3D_WRC_MODIFIED_VAN_GENUCHTEN) dSe *= s.invSc;
    } else {
        const double a = se_from_psi(s, psiCurr, wrc), c = se_from_psi(s, psiPrev, wrc);
        dSe = fabs((a - c) / (H - Ho));
    }
    return dSe * (s.thetaS - s.thetaR);
}

/* ======================================================================================= */
/* multi-GPU exchange (device side)                                                          */
/* ======================================================================================= */

#define SYS_STORE(p, x) __hip_atomic_store((p), (x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define SYS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)

/* One block.  All-gather of up to three doubles per rank through the peers' windows, combined in
 * rank order (identical bits on every rank).  op 0 = sum, 1 = max.  Advances the epoch.
 * Returns false (and raises distError / ST_FAIL) when a peer does not answer within 60 s. */
__device__ __forceinline__ bool dist_allgather(const DevView& v, Ctrl* c, double (&vals)[3], int op)
{
    if (v.world <= 1) return true;
    if (v.dist->rccl) {        /* the host queued k_local_reduce + ncclAllGather before this kernel: combine in rank order */
        __shared__ double shR[3];
        if (threadIdx.x == 0) {
            const double* g = v.dist->gathered;
            double acc[3] = {g[0], g[1], g[2]};
            for (int p = 1; p < v.world; ++p)
                for (int k = 0; k < 3; ++k) { const double x = g[3 * p + k]; acc[k] = op ? dmax(acc[k], x) : acc[k] + x; }
            shR[0] = acc[0]; shR[1] = acc[1]; shR[2] = acc[2];
            c->epoch = c->epoch + 1;
        }
        __syncthreads();
        vals[0] = shR[0]; vals[1] = shR[1]; vals[2] = shR[2];
        return true;
    }
    __shared__ double shIn[SF3D_MAX_RANKS][3];
    __shared__ double sh[3];
    __shared__ int shOk;
    const DistView* d = v.dist;
    const uint32_t e = c->epoch, par = e & 1u;
    const unsigned long long tag = (unsigned long long)e + 1ull;
    if (threadIdx.x == 0) shOk = 1;
    __syncthreads();
    if ((int)threadIdx.x < v.world) {            /* thread p talks to rank p: the stores and the waits of all peers overlap */
        const int p = threadIdx.x;
        __threadfence_system();                  /* halo puts of the preceding kernels first */
        DistMail* m = &d->win[p]->mail[par][v.rank];
        SYS_STORE(&m->v[0], vals[0]); SYS_STORE(&m->v[1], vals[1]); SYS_STORE(&m->v[2], vals[2]);
        __threadfence_system();
        __hip_atomic_store(&m->seq, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        DistMail* in = &d->win[v.rank]->mail[par][p];
        bool ok = true;
        const long long t0 = wall_clock64();     /* 100 MHz */
        while (__hip_atomic_load(&in->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 6000000000LL) { ok = false; break; }   /* 60 s at 100 MHz */
        }
        if (ok) for (int k = 0; k < 3; ++k) shIn[p][k] = SYS_LOAD(&in->v[k]);
        else shOk = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc[3] = {0., 0., 0.};
        if (shOk)
            for (int p = 0; p < v.world; ++p)    /* rank order: identical bits on every rank */
                for (int k = 0; k < 3; ++k) {
                    const double x = shIn[p][k];
                    if (p == 0) acc[k] = x;
                    else acc[k] = op ? dmax(acc[k], x) : acc[k] + x;
                }
        sh[0] = acc[0]; sh[1] = acc[1]; sh[2] = acc[2];
        c->epoch = e + 1;
        if (!shOk) { c->distError = 1; c->stage = ST_FAIL; }
    }
    __syncthreads();
    vals[0] = sh[0]; vals[1] = sh[1]; vals[2] = sh[2];
    return shOk != 0;
}

/* whole block: copy field `field` of the payload every neighbour put for epoch parity `par` into dst.
 * Eight independent system-scope loads per thread are in flight before the first store: the payload lives in
 * fine-grained memory (uncached reads, ~1-2 us each), and this runs in ONE block on the critical path of every sweep. */
__device__ __forceinline__ void dist_unpack(const DevView& v, uint32_t par, int field, double* __restrict__ dst)
{
    const DistView* d = v.dist;
    constexpr uint32_t U = 8;
    for (int p = 0; p < v.world; ++p) {
        const uint32_t cnt = d->recvCount[p];
        if (cnt == 0) continue;
        const uint32_t* idx = d->recvIdx[p];
        if (d->rccl) {             /* packed buffer filled by ncclRecv: slot 0 = the sweep iterate or K, slot 1 = waterFlow */
            const double* rb = d->recvBuf[p] + (size_t)(field == DF_FLOW ? 1 : 0) * cnt;
            for (uint32_t k = threadIdx.x; k < cnt; k += SF3D_BLOCK) dst[idx[k]] = rb[k];
            continue;
        }
        const double* src = d->payload[v.rank] + d->recvOff[p] + (size_t)(par * SF3D_DIST_FIELDS + field) * cnt;
        for (uint32_t k0 = threadIdx.x; k0 < cnt; k0 += SF3D_BLOCK * U) {
            double val[U]; uint32_t id[U];
            #pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t k = k0 + u * SF3D_BLOCK;
                if (k < cnt) { id[u] = idx[k]; val[u] = SYS_LOAD(&src[k]); }
            }
            #pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t k = k0 + u * SF3D_BLOCK;
                if (k < cnt) dst[id[u]] = val[u];
            }
        }
    }
}

/* grid-stride: put field values of my boundary nodes into every neighbour's window */
__device__ __forceinline__ void dist_push(const DevView& v, uint32_t par, int field, const double* __restrict__ src)
{
    const DistView* d = v.dist;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int p = 0; p < v.world; ++p) {
        const uint32_t cnt = d->sendCount[p];
        if (cnt == 0) continue;
        const uint32_t* idx = d->sendIdx[p];
        if (d->rccl) {             /* pack for ncclSend */
            double* sb = d->sendBuf[p] + (size_t)(field == DF_FLOW ? 1 : 0) * cnt;
            for (uint32_t k = tid; k < cnt; k += nth) sb[k] = src[idx[k]];
            continue;
        }
        double* dst = d->payload[p] + d->sendOff[p] + (size_t)(par * SF3D_DIST_FIELDS + field) * cnt;
        for (uint32_t k = tid; k < cnt; k += nth) SYS_STORE(&dst[k], src[idx[k]]);
    }
}

/* after k_sweep: the new iterate of my boundary nodes -> neighbours (consumed by their k_decide_sweep) */
__global__ void __launch_bounds__(SF3D_BLOCK) k_push_x(DevView v)
{
    const Ctrl* c = v.ctrl;
    if (c->stage != ST_SWEEP) return;
    dist_push(v, c->epoch & 1u, 0, v.X[free_buffer(c)]);
    __threadfence_system();
}
/* after k_props: K and waterFlow of my boundary nodes -> neighbours (consumed by their k_sync_kf) */
__global__ void __launch_bounds__(SF3D_BLOCK) k_push_kf(DevView v)
{
    const Ctrl* c = v.ctrl;
    if (c->stage != ST_APPROX) return;
    dist_push(v, c->epoch & 1u, DF_K, v.K);
    dist_push(v, c->epoch & 1u, DF_FLOW, v.flow);
    __threadfence_system();
}
/* barrier + halo of K / waterFlow before the assembly reads neighbours */
__global__ void __launch_bounds__(SF3D_BLOCK) k_sync_kf(DevView v)
{
    Ctrl* c = v.ctrl;
    if (c->stage != ST_APPROX) return;
    const uint32_t par = c->epoch & 1u;
    __syncthreads();
    double vals[3] = {0., 0., 0.};
    if (!dist_allgather(v, c, vals, 0)) return;
    dist_unpack(v, par, DF_K, v.K);
    dist_unpack(v, par, DF_FLOW, v.flow);
}

/* SF3D_EXCHANGE=rccl: this rank's partial sums for the ncclAllGather the host queues next.  what: 0 = sum of part0[nb] (sweep norm,
 * storage query), 1 = maximum of part0[nbSurf] (Courant), 2 = sums of part0[nb] and part1[nb] (balance); the same reductions, in the
 * same order, as the decision kernel that follows would make for itself */
__global__ void __launch_bounds__(SF3D_BLOCK) k_local_reduce(DevView v, int what)
{
    double a = 0., b = 0.;
    if (what == 1) a = reduce_partials_max(v.part0, v.nbSurf);
    else { a = reduce_partials_sum(v.part0, v.nb); if (what == 2) b = reduce_partials_sum(v.part1, v.nb); }
    if (threadIdx.x == 0) { double* m = v.dist->mine; m[0] = a; m[1] = b; m[2] = 0.; }
}

/* ---- last-block hand-off inside a launch ---------------------------------------------------
 * Every block calls this once, after a block-wide barrier that follows its last store.  Thread 0
 * publishes the block's partial results write-through (sc1), drains them and arrives on a two-level
 * agent-scope counter (16 shard counters on separate lines, then one top counter: a single counter
 * serialises ~12 ns per block).  Returns true in every thread of the ONE block that arrived last;
 * that block may then read all partials with sc1 loads (MI355X_MICROARCH.md "Valid forms": sc1 payload
 * + drain + agent atomic add, consumer = the workgroup whose add came last). */
__device__ __forceinline__ bool arrive_last(const DevView& v, double p0, double p1, bool twoValues)
{
    __shared__ int sLast;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&v.part0[blockIdx.x], p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (twoValues) __hip_atomic_store(&v.part1[blockIdx.x], p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int shard = blockIdx.x & 15u;
        const unsigned int inShard = (gridDim.x + 15u - shard) >> 4;
        const unsigned int shards = gridDim.x < 16u ? gridDim.x : 16u;
        int last = 0;
        if (__hip_atomic_fetch_add(&v.arrive[16u * (shard + 1u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == inShard - 1u) {
            __hip_atomic_store(&v.arrive[16u * (shard + 1u)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(&v.arrive[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1u;
            if (last) __hip_atomic_store(&v.arrive[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sLast = last;
    }
    __syncthreads();
    return sLast != 0;
}
/* last block only: fixed-order sum of the published partials (same order as reduce_partials_sum) */
__device__ __forceinline__ double sum_published(const double* p, uint32_t nb)
{
    double s = 0.;
    for (uint32_t k = threadIdx.x; k < nb; k += SF3D_BLOCK) s += __hip_atomic_load(&p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return block_sum(s);
}

/* A wave puts the values its chunk owes to neighbouring ranks (multi-GPU, in-kernel halo put).
 * All 64 lanes must call it; `val` is the lane's freshly computed value. */
__device__ __forceinline__ void dist_put_chunk(const DevView& v, uint32_t q, uint32_t lane, uint32_t par, int field, double val)
{
    const DistView* d = v.dist;
    const uint32_t b0 = d->bndStart[q], b1 = d->bndStart[q + 1];           /* wave-uniform */
    for (uint32_t base = b0; base < b1; base += 64) {
        const uint32_t t = base + lane;
        const bool on = t < b1;
        const double x = __shfl(val, on ? (int)d->bndLane[t] : 0, 64);
        if (on) {
            const uint32_t p = d->bndPeer[t];
            double* dst = d->payload[p] + d->sendOff[p] + (size_t)(par * SF3D_DIST_FIELDS + field) * d->sendCount[p] + d->bndSlot[t];
            SYS_STORE(dst, x);
        }
    }
}

/* ======================================================================================= */
/* control kernels (one thread / one block)                                                 */
/* ======================================================================================= */

/* waterMainLoop body head, cpusolver.cpp:155-162 (Hold = H is an index copy) */
__device__ __forceinline__ void begin_attempt(Ctrl* c)
{
    c->dt = dmin(c->dtCurr, c->maxTimeStep);
    c->hold = c->cur;
    c->best = -1;
    c->bestMBR = NODATA_D;
    c->approx = 0;
    c->counters[0]++;
    c->stage = ST_APPROX;
}

__global__ void k_step_begin(Ctrl* c, double maxTimeStep)
{
    c->maxTimeStep = maxTimeStep;
    c->seqCount = 0;
    c->aBuf ^= 1u;              /* the matrix of the step accepted before stays intact for its link flow sums (k_accept_links) */
    begin_attempt(c);
}

/* a refused attempt restores H = Hold (cpusolver.cpp:182-186) and the loop starts the next one */
__device__ __forceinline__ void reject_attempt(Ctrl* c)
{
    c->seSource = 2;                   /* the next attempt starts from Hold again: its Se is what approximation 0 stored in SeHold */
    c->cur = c->hold;
    begin_attempt(c);
}

/* checkCourant, cpusolver.cpp:248-281, then the iteration budget of solver.h:55-59 (one thread) */
__device__ __forceinline__ void courant_decision(Ctrl* c, double cmax)
{
    c->counters[2]++;
    c->courant = cmax;
    c->asmSeq++; c->asmSurfOnly = (cmax < 1.01 || c->dt <= c->dtMin) ? 0u : 1u;
    if (cmax < 1.01 || c->dt <= c->dtMin) {
        uint32_t budget = (uint32_t)((c->approx + 1) * ((float)c->maxIter / (float)c->maxApprox));
        c->iterBudget = budget > 25u ? budget : 25u;
        c->iter = 0;
        c->bestNorm = 1.;
        c->linearValid = 1;
        c->stage = ST_SWEEP;
        return;
    }
    double d = c->dtCurr / cmax;
    int mult = 0;
    while (d < 1.) { d *= 10.; ++mult; }
    d = floor(d);
    for (int k = 0; k < mult; ++k) d /= 10.;
    c->dtCurr = dmax(c->dtMin, d);
    c->counters[4]++;
    if (c->seqCount < 16u) c->seqSweeps[c->seqCount] = 0;
    c->seqCount++;
    reject_attempt(c);
}
__global__ void __launch_bounds__(SF3D_BLOCK) k_decide_courant(DevView v)
{
    Ctrl* c = v.ctrl;
    if (c->stage != ST_APPROX) return;
    double vals[3] = {reduce_partials_max(v.part0, v.nbSurf), 0., 0.};
    if (!dist_allgather(v, c, vals, 1)) return;
    if (threadIdx.x == 0) courant_decision(c, vals[0]);
}

/* solveLinearSystem loop control, cpusolver.cpp:672-703 + :442-447 (one thread) */
__device__ __forceinline__ void sweep_decision(Ctrl* c, int nxt, double norm)
{
    c->cur = nxt;                     /* std::swap(vectorNewX, vectorX), water.cpp:598 */
    c->iter++;
    c->counters[3]++;
    c->lastNorm = norm;
    bool done = false, valid = true;
    if (norm < c->residualTolerance) done = true;
    else if (norm > (c->bestNorm * 10)) { done = true; valid = false; }
    else {
        if (norm < c->bestNorm) c->bestNorm = norm;
        if (c->iter >= c->iterBudget) done = true;
    }
    if (!done) return;
    if (c->seqCount < 16u) c->seqSweeps[c->seqCount] = c->iter;
    c->seqCount++;
    c->linearValid = valid ? 1 : 0;
    if (!valid) c->counters[5]++;
    if (!valid && c->dt > c->dtMin) {
        c->dtCurr = dmax(c->dtMin, c->dtCurr / 2.);
        reject_attempt(c);
    } else
        c->stage = ST_POST;
}

__global__ void __launch_bounds__(SF3D_BLOCK) k_decide_sweep(DevView v)
{
    Ctrl* c = v.ctrl;
    if (c->stage != ST_SWEEP) return;
    const int nxt = free_buffer(c);              /* the buffer k_sweep wrote */
    const uint32_t par = c->epoch & 1u;
    double vals[3] = {reduce_partials_sum(v.part0, v.nb), 0., 0.};
    if (!dist_allgather(v, c, vals, 0)) return;
    if (v.world > 1) dist_unpack(v, par, 0, v.X[nxt]);     /* neighbours' new iterate on my halo */
    if (threadIdx.x != 0) return;
    sweep_decision(c, nxt, vals[0] / v.Nnorm);
}

/* computeCurrentMassBalance, water.cpp:96-123 */
__device__ __forceinline__ void mass_balance(Ctrl* c, double storage, double sink)
{
    BalanceDev b;
    b.storage = storage;
    const double dS = b.storage - c->prevStep.storage;
    b.sinkSource = sink;
    b.MBE = dS - b.sinkSource;
    const double timePct = 0.001 * dmax(c->dt, 30.0) / 3600.;
    double minRef = b.storage * timePct;
    minRef = dmax(minRef, 0.001);
    const double ref = dmax(fabs(b.sinkSource), minRef);
    b.MBR = b.MBE / ref;
    c->curStep = b;
}
/* scalar part of acceptStep, water.cpp:233-237 */
__device__ __forceinline__ void accept_bookkeeping(Ctrl* c)
{
    c->prevStep.storage = c->curStep.storage;
    c->prevStep.sinkSource = c->curStep.sinkSource;
    c->curPeriod.sinkSource += c->curStep.sinkSource;
    c->counters[1]++;
    c->acceptDt = c->dt; c->acceptBuf = c->cur; c->acceptABuf = c->aBuf;
    c->seSource = 1;                   /* Se = Se(H accepted): k_post, or k_restore for a restored best step */
    c->stage = ST_ACCEPT;              /* k_accept (flow sums) is the last kernel of the step */
}
__device__ __forceinline__ void halve_and_reject(Ctrl* c)
{
    c->dtCurr = dmax(c->dtCurr * 0.5, c->dtMin);
    reject_attempt(c);
}

/* evaluateWaterBalance, water.cpp:165-227 (one thread) */
__device__ __forceinline__ void balance_decision(Ctrl* c, double storage, double sink)
{
    c->counters[7]++;
    mass_balance(c, storage, sink);
    const double err = fabs(c->curStep.MBR);
    const uint32_t approx = c->approx;
    if (isnan(err)) {
        if (c->dt > c->dtMin) halve_and_reject(c);
        else if (approx > 0) { c->cur = c->best; c->stage = ST_RESTORE; }
        else c->stage = ST_FAIL;
        return;
    }
    if (err < c->MBRThreshold) {
        accept_bookkeeping(c);
        if (approx < 3 && err < c->MBRThreshold * 0.1 && c->courant < c->courantThreshold)
            c->dtCurr = dmin(c->dtMax, c->dtCurr * 2);
        return;
    }
    if (approx == 0 || err < c->bestMBR) { c->best = c->cur; c->bestMBR = err; }   /* keep-best = index copy */
    if (err > (c->bestMBR * c->instabilityFactor) || approx == (c->maxApprox - 1)) {
        if (c->dt > c->dtMin) { halve_and_reject(c); return; }
        c->cur = c->best;                                                          /* restoreBestStep: H = Hbest */
        c->stage = ST_RESTORE;
        return;
    }
    c->approx = approx + 1;
    c->stage = ST_APPROX;
}
__global__ void __launch_bounds__(SF3D_BLOCK) k_decide_balance(DevView v)
{
    Ctrl* c = v.ctrl;
    if (c->stage != ST_POST) return;
    double vals[3] = {reduce_partials_sum(v.part0, v.nb), reduce_partials_sum(v.part1, v.nb), 0.};
    if (!dist_allgather(v, c, vals, 0)) return;
    if (threadIdx.x == 0) balance_decision(c, vals[0], vals[1]);
}

/* tail of restoreBestStep (water.cpp:266) followed by acceptStep (one thread) */
__device__ __forceinline__ void restore_decision(Ctrl* c, double storage, double sink)
{
    c->counters[6]++;
    mass_balance(c, storage, sink);
    accept_bookkeeping(c);
}
__global__ void __launch_bounds__(SF3D_BLOCK) k_decide_restore(DevView v)
{
    Ctrl* c = v.ctrl;
    if (c->stage != ST_RESTORE) return;
    double vals[3] = {reduce_partials_sum(v.part0, v.nb), reduce_partials_sum(v.part1, v.nb), 0.};
    if (!dist_allgather(v, c, vals, 0)) return;
    if (threadIdx.x == 0) restore_decision(c, vals[0], vals[1]);
}

__global__ void __launch_bounds__(SF3D_BLOCK) k_decide_query(DevView v)
{
    double vals[3] = {reduce_partials_sum(v.part0, v.nb), 0., 0.};
    if (!dist_allgather(v, v.ctrl, vals, 0)) return;
    if (threadIdx.x == 0) v.ctrl->query[0] = vals[0];
}

/* ======================================================================================= */
/* node kernels: one wave64 walks 64-node chunks (grid-stride), one lane per node            */
/* ======================================================================================= */

/* chunk loop: q is wave-uniform (forced into an SGPR so descriptor loads go through the scalar path) */
#define FOR_EACH_CHUNK_IN(v, l0, l1)                                                              \
    const uint32_t lane_ = threadIdx.x & 63u;                                                     \
    const uint32_t wavesTotal_ = gridDim.x * (SF3D_BLOCK / 64);                                   \
    for (uint32_t li_ = __builtin_amdgcn_readfirstlane((l0) + blockIdx.x * (SF3D_BLOCK / 64) + (threadIdx.x >> 6)); \
         li_ < (l1); li_ += wavesTotal_)                                                          \
        for (uint32_t q = __builtin_amdgcn_readfirstlane((v).chunkList[li_]), once_ = 1; once_; once_ = 0)
/* chunks this rank owns (all of them on one GPU), their surface part, their soil part */
#define FOR_EACH_CHUNK(v) FOR_EACH_CHUNK_IN(v, 0u, (v).nList)
/* lane-level ownership: strips are cut at chunk boundaries whenever the numbering allows it, the
 * per-node owner map covers the general case */
#define NOT_MINE(v, i) ((i) >= (v).N || ((v).owner != nullptr && (v).owner[i] != (v).rank))

#include "sf3d_heat.inc"

/* ---- waterFlow = sink (+ evaporation clamp) + boundary flow (water.cpp:632-807) ---- */
template <bool HEAT>
__device__ __forceinline__ double boundary_update(const DevView& v, const Ctrl* c, uint32_t i, double H, double Ho,
                                                  double z, double K, double Se)
{
    const double dt = c->dt;
    double flow = v.sink[i];
    if (i < v.ns && flow < 0) {                                       /* :646-652 */
        const double avgH = 0.5 * (H + Ho);
        const double hs = dmax(0., avgH - z);
        const double maxFlux = -hs * v.size[i] / dt;
        flow = dmax(flow, maxFlux);
    }
    const uint8_t bt = v.btype[i];
    double rate = 0.;
    if (bt != SF3D_BND_NONE) {
        switch (bt) {
            case SF3D_BND_RUNOFF: {                                   /* :661-678 */
                const double avgH = 0.5 * (H + Ho);
                const double hs = dmax(0., avgH - (z + v.pond[i]));
                if (hs < 0.001) break;
                const double maxFlow = (hs * v.size[i]) / dt;
                const double vel = ppow(hs, 2. / 3.) * sqrt(v.bslope[i]) / v.roughness[v.cls[i]];
                const double val = hs * vel * v.bsize[i];
                rate = -dmin(val, maxFlow);
                break; }
            case SF3D_BND_FREE_DRAINAGE:                              /* :680-684, Up-link area */
                rate = -K * v.larea[i];
                break;
            case SF3D_BND_FREE_LATERAL_DRAINAGE:                      /* :686-690 */
                rate = -K * v.bsize[i] * v.bslope[i] * c->lvRatio;
                break;
            case SF3D_BND_PRESCRIBED_TOTAL_POTENTIAL: {               /* :692-706 */
                const SoilDev s = v.soils[v.cls[i]];
                const double L = 1.;
                const double bz = z - L;
                const double pH = v.prescribed[i];
                const double bpsi = pH - bz;
                const double bK = (bpsi >= 0) ? s.Ksat : mualem_k(s, se_from_psi(s, fabs(bpsi), c->wrc), c->wrc);
                const double mk = mean_of(bK, K, c->meanType);
                rate = mk * v.bsize[i] * ((pH - H) / L);
                break; }
            case SF3D_BND_HEAT_SURFACE: {                             /* :708-747: soil evaporation / condensation */
                if (!HEAT || !v.heat.vapor) break;
                const HeatDev& hv = v.heat;
                const SoilDev s = v.soils[v.cls[i]];
                const uint32_t up = v.lto[i];                         /* slot 0; node 0 when there is no Up link, like the reference */
                const bool upLinked = v.lkind[i] != LK_NONE;
                double frac = 0.;
                if (upLinked && up < v.ns) {                          /* getNodeSurfaceWaterFraction, soilPhysics.cpp:317-326 */
                    const double hV = dmax(0., v.X[c->cur][up] - v.z[up]);
                    frac = dmin(1., hV / dmax(0.001, v.pond[up]));
                }
                double evap = 0.;
                if (up < v.ns) {                                      /* computeNodeAtmosphericLatentVaporFlux, heat.cpp:988-1008 */
                    const double Ta = hv.bT[i];
                    const double satC = h_vapor_conc_from_pressure(h_sat_vapor_pressure(Ta - H_ZEROC), Ta);
                    const double boundaryVapor = satC * (hv.bRH[i] / 100.);
                    const double dVapor = boundaryVapor - h_vapor_from_psi_temp(H - z, hv.TX[c->tCur][i]);
                    const double total = 1. / ((1. / hv.bAero[i]) + (1. / hv.bSoilCond[i]));
                    evap = dVapor * total;
                }
                evap = evap / H_RHOW * v.larea[i];
                if (frac > 0.) evap *= (1. - frac);
                const double thetaV = (Se * (s.thetaS - s.thetaR)) + s.thetaR;
                rate = (evap < 0.) ? dmax(evap, -(thetaV - s.thetaR) * v.size[i] / dt)
                                   : dmin(evap, (s.thetaS - s.thetaR) * v.size[i] / dt);
                break; }
            default: rate = 0.; break;                                /* Urban/Road/Culvert */
        }
        if (fabs(rate) < DBL_EPSILON) rate = 0.;                      /* :802-805 */
        else flow += rate;
    }
    if (HEAT && i < v.ns && v.heat.vapor && v.lkind[(size_t)v.N + i] != LK_NONE) {
        /* evaporation of ponded water (water.cpp:719-733).  The reference lets the HeatSurface node below write
         * into this node; here the surface node pulls: same values, same order (this node's own boundary first) */
        const HeatDev& hv = v.heat;
        const uint32_t d = v.lto[(size_t)v.N + i];                    /* slot 1 = Down */
        if (v.btype[d] == SF3D_BND_HEAT_SURFACE && v.lkind[d] != LK_NONE && v.lto[d] == i) {
            const double hV = dmax(0., H - z);
            const double frac = dmin(1., hV / dmax(0.001, v.pond[i]));
            if (frac > 0.) {                                          /* computeNodeAtmosphericLatentSurfaceWaterFlux, heat.cpp:1014-1039 */
                const double Ta = hv.bT[d];
                const double satC = h_vapor_conc_from_pressure(h_sat_vapor_pressure(Ta - H_ZEROC), Ta);
                const double boundaryVapor = satC * (hv.bRH[d] / 100.);
                double surfEvap = ((boundaryVapor - satC) * hv.bAero[d]) / H_RHOW * v.larea[d];
                surfEvap *= frac;
                const double volume = (H - z) * v.size[i];
                surfEvap = dmax(surfEvap, -volume / dt);
                if (bt != SF3D_BND_NONE) rate = surfEvap;             /* overwrites the rate, waterFlow keeps the runoff already added */
                else flow += surfEvap;
            }
        }
    }
    if (bt != SF3D_BND_NONE) v.bflowRate[i] = rate;
    v.flow[i] = flow;
    return flow;
}


/* dtheta/dH with the two saturation degrees already known (soilPhysics.cpp:224-279):
 * Se(psiCurr) is the Se array (post-solve of the previous approximation, same H) and Se(psiPrev)
 * is SeHold (written at approximation 0, when H == Hold) - the same values the reference
 * recomputes with four pow calls per node per approximation. */
__device__ __forceinline__ double dtheta_dh_cached(const SoilDev& s, double H, double Ho, double z, uint32_t wrc,
                                                   double SeCur, double SePrev)
{
    const double psiCurr = fabs(dmin(0.0, H - z));
    const double psiPrev = fabs(dmin(0.0, Ho - z));
    if (wrc == SF3D_WRC_VAN_GENUCHTEN) { if (psiCurr == 0.0 && psiPrev == 0.0) return 0.0; }
    else if (wrc == SF3D_WRC_MODIFIED_VAN_GENUCHTEN) { if (psiCurr <= s.he && psiPrev <= s.he) return 0.0; }
    double dSe;
    if (fabs(psiCurr - psiPrev) < 1e-12) {
        const double xx = s.alpha * psiCurr;
        const double onePlus = 1. + ppow(xx, s.n);
        const double t1 = ppow(onePlus, -(s.m + 1.));
        const double t2 = ppow(xx, s.n - 1.);
        dSe = s.alpha * s.n * s.m * t1 * t2;
        if (wrc == SF3D_WRC_MODIFIED_VAN_GENUCHTEN) dSe *= s.invSc;
    } else
        dSe = fabs((SeCur - SePrev) / (H - Ho));
    return dSe * (s.thetaS - s.thetaR);
}

/* computeCapacity (water.cpp:279-297) + step-begin Se (cpusolver.cpp:165-169) +
 * updateBoundaryWaterData (water.cpp:632-807) */
#ifndef SF3D_PROPS_HEAT_WAVES
#define SF3D_PROPS_HEAT_WAVES 2
#endif
#ifndef SF3D_PROPS_WAVES
#define SF3D_PROPS_WAVES 5     /* 96 VGPRs, no scratch on one GPU (the table-driven pow needs fewer registers than the library's):
                                * 150 -> 135 us at C4 against 4 waves, launched with exactly the 1 280 resident blocks */
#endif
template <int MODE, bool HEAT>
__device__ __forceinline__ void body_props(const DevView& v)
{
    const Ctrl* c = v.ctrl;
    fm_init();
    const double* __restrict__ Xc = v.X[c->cur];
    const double* __restrict__ Xh = v.X[c->hold];
    const uint32_t wrc = c->wrc;
    const uint32_t par = c->epoch & 1u;
    const bool first = c->approx == 0;
    const uint32_t seSource = c->seSource;
    FOR_EACH_CHUNK(v) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        const bool mine = !NOT_MINE(v, i);
        double K = 0., fl = 0.;
        if (mine) {
            const double H = Xc[i], Ho = Xh[i], z = v.z[i];
            double Se = 1.;
            if (i >= v.ns) {
                const SoilDev s = v.soils[v.cls[i]];
                double SeH;
                if (first) {
                    /* cpusolver.cpp:165-169 recomputes Se(H) at the head of every attempt; the same function of the same H was
                     * already evaluated by k_post / k_restore (accepted step) or by the refused attempt: reuse those bits */
                    Se = (seSource == 1) ? v.Se[i] : (seSource == 2) ? v.SeHold[i] : node_se(s, H, z, wrc);
                    SeH = Se; v.Se[i] = Se; v.SeHold[i] = Se;
                }
                else { Se = v.Se[i]; SeH = v.SeHold[i]; }
                K = mualem_k(s, Se, wrc);
                const double dThdH = dtheta_dh_cached(s, H, Ho, z, wrc, Se, SeH);
                double C = v.size[i] * dThdH;
                if (HEAT) {                                   /* vapour terms of computeNodeK / computeCapacity + the node
                                                                 conductivities of the thermal fluxes in the water rows */
                    const HeatDev& hv = v.heat;
                    const double Tm = (hv.TX[c->tCur][i] + hv.TX[c->tOld][i]) * 0.5;   /* getNodeMeanTemperature */
                    const double h = H - z;
                    const double theta = h_theta_signed_psi(s, h, wrc);
                    if (hv.vapor) {
                        K += h_isothermal_vapor_conductivity(s, Tm, h, theta) * (H_G / H_RHOW);
                        C += v.size[i] * h_dthetav_dh(s, h, Tm, dThdH, wrc);
                        hv.wThVap[i] = h_thermal_vapor_conductivity(s, hv.airP[i], Tm, h, theta);
                    }
                    hv.wTm[i] = Tm;
                    hv.wThLiq[i] = h_thermal_liquid_conductivity(Tm - H_ZEROC, h, K);
                }
                v.K[i] = K;
                v.C[i] = C;
            }
            boundary_update<HEAT>(v, c, i, H, Ho, z, K, Se);
            if (MODE == 2) fl = v.flow[i];
        }
        if (MODE == 2) { dist_put_chunk(v, q, lane_, par, DF_K, K); dist_put_chunk(v, q, lane_, par, DF_FLOW, fl); }
    }
    if (MODE != 2) return;
    /* multi GPU: barrier across ranks + halo of K / waterFlow before the assembly reads neighbours */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!arrive_last(v, 0., 0., false)) return;
    double vals[3] = {0., 0., 0.};
    if (!dist_allgather(v, v.ctrl, vals, 0)) return;          /* the barrier: every neighbour's K and waterFlow have landed */
    if (threadIdx.x == 0) v.ctrl->kfEpoch = v.ctrl->epoch;     /* k_halo_copy<0> (next launch, many blocks) copies them in */
}
template <int MODE, bool HEAT>
__global__ void __launch_bounds__(SF3D_BLOCK, HEAT ? SF3D_PROPS_HEAT_WAVES : SF3D_PROPS_WAVES) k_props(DevView v)
{
    if (v.ctrl->stage != ST_APPROX) return;
    body_props<MODE, HEAT>(v);
}

/* The halo copies that used to run in the ONE block that closes an exchange (20 000 system-scope loads at 8-way C4,
 * tens of microseconds on the critical path of every approximation) as a launch of their own with many blocks.
 * WHAT 0: K and waterFlow after k_props; WHAT 1: the final iterate of the sweeps after k_post.  Each copy is tied to the
 * epoch its exchange closed, so a launch queued in vain (guarded no-op batches) copies nothing stale. */
template <int WHAT>
__global__ void __launch_bounds__(SF3D_BLOCK) k_halo_copy(DevView v)
{
    const Ctrl* c = v.ctrl;
    if (v.world <= 1) return;
    const DistView* d = v.dist;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (WHAT == 0) {
        if (c->stage != ST_APPROX || c->kfEpoch != c->epoch) return;
        const uint32_t par = (c->epoch - 1u) & 1u;
        for (int p = 0; p < v.world; ++p) {
            const uint32_t cnt = d->recvCount[p];
            const uint32_t* idx = d->recvIdx[p];
            const double* base = d->payload[v.rank] + d->recvOff[p] + (size_t)(par * SF3D_DIST_FIELDS) * cnt;
            for (uint32_t k = tid; k < cnt; k += nth) {
                const uint32_t i = idx[k];
                v.K[i] = SYS_LOAD(&base[(size_t)DF_K * cnt + k]);
                v.flow[i] = SYS_LOAD(&base[(size_t)DF_FLOW * cnt + k]);
            }
        }
    } else {
        if (c->haloEpoch != c->epoch) return;
        double* __restrict__ dst = v.X[c->haloBuf];
        const uint32_t par = c->haloPar;
        for (int p = 0; p < v.world; ++p) {
            const uint32_t cnt = d->recvCount[p];
            const uint32_t* idx = d->recvIdx[p];
            const double* src = d->payload[v.rank] + d->recvOff[p] + (size_t)(par * SF3D_DIST_FIELDS + DF_X) * cnt;
            for (uint32_t k = tid; k < cnt; k += nth) dst[idx[k]] = SYS_LOAD(&src[k]);
        }
    }
}

/* infiltration, water.cpp:490-539: one end is a surface node, the other a soil node */
__device__ __forceinline__ double infiltration_conductance(const DevView& v, const Ctrl* c, uint32_t i, uint32_t j, size_t e,
                                                           const double* __restrict__ Xc, const double* __restrict__ Xh,
                                                           double Hi, double Hoi, double zi)
{
    const double area = v.larea[e], dist = v.ldist[e];
    const double dt = c->dt;
    const bool iSurf = i < v.ns;
    const uint32_t su = iSurf ? i : j, so = iSurf ? j : i;
    const double Hsu = iSurf ? Hi : Xc[j], Hosu = iSurf ? Hoi : Xh[j];
    const double Hso = iSurf ? Xc[j] : Hi, Hoso = iSurf ? Xh[j] : Hoi;
    const double zsu = iSurf ? zi : v.z[j];
    const SoilDev s = v.soils[v.cls[so]];
    double factor = 1.;
    const uint8_t bt = v.btype[so];
    if (bt == SF3D_BND_URBAN) factor = 0.33;
    else if (bt == SF3D_BND_ROAD) return 0.;
    if (Hso > zsu) return (s.Ksat * factor * area) / dist;
    const double surfH = 0.5 * (Hsu + Hosu);
    const double soilH = 0.5 * (Hso + Hoso);
    double surfaceWater = dmax(surfH - zsu, 0.);
    const double qf = v.flow[su];
    if (qf < 0.) {
        const double bm = (qf * dt) / v.size[su];
        surfaceWater = dmax(0., surfaceWater + bm);
    }
    const double maxInfRate = surfaceWater / dt;
    if (maxInfRate < 2.78e-11) return 0.;
    const double dH = dmax(surfH - soilH, 1e-12);
    const double maxK = maxInfRate * (dist / dH);
    const double meanK = mean_of(s.Ksat, v.K[so], c->meanType);
    return (dmin(factor * meanK, maxK) * area) / dist;
}

/* link conductances: water.cpp:300-343 dispatch, :413-487 runoff, :490-539 infiltration,
 * :542-562 redistribution */
__device__ __forceinline__ double link_conductance(const DevView& v, const Ctrl* c, uint32_t i, uint32_t j, size_t e,
                                                   uint8_t kind, const double* __restrict__ Xc,
                                                   const double* __restrict__ Xh, double Hi, double Hoi,
                                                   double zi, double& courant)
{
    const double area = v.larea[e], dist = v.ldist[e];
    const double dt = c->dt;
    if (kind == LK_SOIL_VERT || kind == LK_SOIL_LAT) {
        double ki = v.K[i], kj = v.K[j];
        if (kind == LK_SOIL_LAT) { ki *= c->lvRatio; kj *= c->lvRatio; }
        return qdiv(mean_of(ki, kj, c->meanType) * area, dist);
    }
    if (kind == LK_RUNOFF) {
        double Ha = 0.5 * (Hi + Hoi);
        double Hb = 0.5 * (Xc[j] + Xh[j]);
        if (c->approx == 0) {
            const double fi = v.flow[i], fj = v.flow[j];
            if (fi > 0) Ha += 0.5 * fi * dt / v.size[i];
            if (fj > 0) Hb += 0.5 * fj * dt / v.size[j];
        }
        const double za = zi + v.pond[i], zb = v.z[j] + v.pond[j];
        const double Hs = dmax(Ha, Hb) - dmax(za, zb);
        if (Hs <= 0.00001) return 0.0;
        if (dist <= 0.0) return 0.0;
        const double rough = 0.5 * (v.roughness[v.cls[i]] + v.roughness[v.cls[j]]);
        if (rough <= 0.0) return 0.0;
        const double Aw = area * Hs;
        const double Hs23 = cbrt(Hs * Hs);
        const double Kij = Aw * Hs23 / (rough * dist);
        const double dH = fabs(Ha - Hb);
        const double slope = (dH > 0.00001) ? dH / dist : 0.0;
        const double vel = Hs23 * sqrt(slope) / rough;
        courant = dmax(courant, vel * dt / dist);
        return Kij;
    }
    return infiltration_conductance(v, c, i, j, e, Xc, Xh, Hi, Hoi, zi);
}

/* computeLinearSystemElement (cpusolver.cpp:348-389, order Up, laterals, Down) +
 * computeDiagonalElement (:335-345) + preconditioningMatrix (:284-305), in two kernels like the
 * reference's two loops (surface rows, Courant check, soil rows - cpusolver.cpp:412-429). */
/* thermal liquid (+ vapour) flux of one soil-soil link in the water rows: computeThermalLiquidFlux /
 * computeThermalVaporFlux with processType::Water (heat.cpp:469-567), added to invariantFluxes (water.cpp:328-340) */
__device__ __forceinline__ void add_thermal_fluxes(const HeatDev& hv, uint32_t i, uint32_t j, double area, double dist3, double& inv)
{
    const double Ti = hv.wTm[i], Tj = hv.wTm[j];
    const double avgL = mean_of(hv.wThLiq[i], hv.wThLiq[j], SF3D_MEAN_LOGARITHMIC);
    inv += (avgL * (Tj - Ti) / dist3) * area;
    if (hv.vapor) {
        const double avgV = mean_of(hv.wThVap[i], hv.wThVap[j], SF3D_MEAN_LOGARITHMIC);
        inv += ((avgV * (Tj - Ti) / dist3) * area) / H_RHOW;
    }
}

/* scales the row (preconditioningMatrix, cpusolver.cpp:284-305), stores it, and leaves the scaled coefficients in k
 * and the right-hand side in the return value for the fused first sweep */
template <bool NT>
__device__ __forceinline__ double store_row(const DevView& v, sf3d_d2* __restrict__ A2w, const ChunkDesc& cd, uint32_t i, double (&k)[SF3D_SLOTS],
                                            double sum, double Hoi, double dt, double invariantFlux, double Ci, double flowi)
{
    const double cdt = Ci / dt;
    const bool raw = v.compatDiag != nullptr;          /* compat: row stored un-normalised, k_compat_rows scales it after the Courant decision */
    const double inv = raw ? 1.0 : 1.0 / (cdt + sum);  /* (x * 1.0 is exact) */
    if (raw) v.compatDiag[i] = cdt + sum;
    if (v.cgDiag != nullptr) v.cgDiag[i] = cdt + sum;
    #pragma unroll
    for (int p = 0; p < SF3D_SLOTS / 2; ++p) {
        k[2 * p] = (k[2 * p] * -1.) * inv; k[2 * p + 1] = (k[2 * p + 1] * -1.) * inv;
        if (cd.kind[2 * p] != CK_NONE || cd.kind[2 * p + 1] != CK_NONE) store_coeff<NT>(&A2w[(size_t)p * v.N + i], k[2 * p], k[2 * p + 1]);
    }
    const double bi = ((cdt * Hoi) + flowi + invariantFlux) * inv;   /* invariantFluxes: 0 without heat (cpusolver.cpp:148,387) */
    v.b[i] = bi;
    return bi;
}

/* rows of chunks [0, qSplit): every surface node (runoff + infiltration links) and, when
 * nrSurfaceNodes is not a multiple of 64, the first soil nodes; any link kind; Courant maximum */
template <bool NT, bool HEAT>
__device__ __forceinline__ double assemble_surface_rows(const DevView& v, uint32_t blk, uint32_t nblk)
{
    const Ctrl* c = v.ctrl;
    const double* __restrict__ Xc = v.X[c->cur];
    const double* __restrict__ Xh = v.X[c->hold];
    sf3d_d2* __restrict__ A2w = cur_A2(v);
    const double dt = c->dt;
    constexpr uint32_t order[SF3D_SLOTS] = {0, 2, 3, 4, 5, 6, 7, 8, 9, 1};
    double courant = 0.;
    const uint32_t lane_ = threadIdx.x & 63u;
    for (uint32_t li_ = __builtin_amdgcn_readfirstlane(blk * (SF3D_BLOCK / 64) + (threadIdx.x >> 6)); li_ < v.nListSurf;
         li_ += nblk * (SF3D_BLOCK / 64)) {
        const uint32_t q = __builtin_amdgcn_readfirstlane(v.asmList[li_]);
        const uint32_t i = q * SF3D_CHUNK + lane_;
        if (NOT_MINE(v, i)) continue;
        const ChunkDesc cd = v.cdesc[q];
        const double Hi = Xc[i], Hoi = Xh[i], zi = v.z[i];
        const double Ci = (i < v.ns) ? v.size[i] : v.C[i], flowi = v.flow[i];     /* surface capacity = area, cpusolver.cpp:151 */
        double k[SF3D_SLOTS];
        double sum = 0., invFlux = 0.;
        #pragma unroll
        for (int o = 0; o < SF3D_SLOTS; ++o) {
            const uint32_t s = order[o];
            double ks = 0.;
            if (cd.kind[s] != CK_NONE) {
                const size_t e = (size_t)s * v.N + i;
                uint8_t kind; uint32_t j;
                if (cd.kind[s] != CK_MIXED) { kind = cd.kind[s]; j = i + cd.delta[s]; }
                else if ((cd.sweepUniform >> s) & 1u) { kind = v.lkind[e]; const uint32_t jj = i + (uint32_t)cd.delta[s]; j = jj < v.N ? jj : i; }
                else { kind = v.lkind[e]; j = v.lto[e]; }
                if (kind != LK_NONE) ks = link_conductance(v, c, i, j, e, kind, Xc, Xh, Hi, Hoi, zi, courant);
                if (HEAT && (kind == LK_SOIL_VERT || kind == LK_SOIL_LAT)) add_thermal_fluxes(v.heat, i, j, v.larea[e], v.heat.hdist[e], invFlux);
            }
            k[s] = ks;
            sum += ks;
        }
        store_row<NT>(v, A2w, cd, i, k, sum, Hoi, dt, invFlux, Ci, flowi);
    }
    return block_max(courant);
}

/* rows of the soil-only chunks [qSplit, nChunks).  Two groups of five slots: all index / area /
 * distance loads and neighbour-K gathers of a group are issued before its first logarithm, so
 * ~20 loads per lane overlap instead of forming dependent round trips (4 waves/SIMD). */
#ifndef SF3D_ASM_HEAT_WAVES
#define SF3D_ASM_HEAT_WAVES 2
#endif
#ifndef SF3D_ASM_SCHED_BARRIER
#define SF3D_ASM_SCHED_BARRIER 1
#endif
#ifndef SF3D_ASM_WAVES
#define SF3D_ASM_WAVES 4
#endif
#ifndef SF3D_ASM_GROUPS
#define SF3D_ASM_GROUPS 2      /* 2 groups of 5 slots (5 groups of 2, 1 of 10: tuning experiments) */
#endif
template <bool NT, bool HEAT>
__device__ __forceinline__ void assemble_soil_rows(const DevView& v, uint32_t blk, uint32_t nblk)
{
    const Ctrl* c = v.ctrl;
    const double* __restrict__ Xc = v.X[c->cur];
    const double* __restrict__ Xh = v.X[c->hold];
    const double dt = c->dt, lvRatio = c->lvRatio;
    sf3d_d2* __restrict__ A2w = cur_A2(v);
    const uint32_t meanType = c->meanType;
    constexpr uint32_t order[SF3D_SLOTS] = {0, 2, 3, 4, 5, 6, 7, 8, 9, 1};
    const uint32_t lane_ = threadIdx.x & 63u;
    for (uint32_t li_ = __builtin_amdgcn_readfirstlane(v.nListSurf + blk * (SF3D_BLOCK / 64) + (threadIdx.x >> 6)); li_ < v.nList;
         li_ += nblk * (SF3D_BLOCK / 64)) {
        const uint32_t q = __builtin_amdgcn_readfirstlane(v.asmList[li_]);
        const uint32_t i = q * SF3D_CHUNK + lane_;
        if (NOT_MINE(v, i)) continue;
        const ChunkDesc cd = v.cdesc[q];                                    /* wave-uniform: scalar load */
        const double Hoi = Xh[i], Ki = v.K[i];
        const double Ci = v.C[i], flowi = v.flow[i];       /* for the row's diagonal and right-hand side: requested with the first loads, not after the last logarithm */
        double k[SF3D_SLOTS];
        double sum = 0., invFlux = 0.;
        #pragma unroll
        for (int g = 0; g < SF3D_ASM_GROUPS; ++g) {
            constexpr int GS = SF3D_SLOTS / SF3D_ASM_GROUPS;          /* slots per group */
            uint32_t j[GS]; uint8_t kd[GS];
            double area[GS], dist[GS], kj[GS];
            #pragma unroll
            for (int t = 0; t < GS; ++t) {
                const uint32_t s = order[g * GS + t];
                kd[t] = LK_NONE; j[t] = i; area[t] = 0.; dist[t] = 1.;
                if (cd.kind[s] != CK_NONE) {
                    const size_t e = (size_t)s * v.N + i;
                    if (cd.kind[s] != CK_MIXED) { kd[t] = cd.kind[s]; j[t] = i + cd.delta[s]; }
                    else if ((cd.sweepUniform >> s) & 1u) {     /* row end: one offset for the nodes that have the link - the gather does not wait for an index */
                        kd[t] = load_stream<NT>(&v.lkind[e]);
                        const uint32_t jj = i + (uint32_t)cd.delta[s];
                        j[t] = jj < v.N ? jj : i;
                    } else { kd[t] = load_stream<NT>(&v.lkind[e]); j[t] = load_stream<NT>(&v.lto[e]); }
                    area[t] = ((cd.areaUniform >> s) & 1u) ? cd.area[s] : load_stream<NT>(&v.larea[e]);
                    dist[t] = load_stream<NT>(&v.ldist[e]);
                }
            }
            #pragma unroll
            for (int t = 0; t < GS; ++t) kj[t] = v.K[j[t]];
            #pragma unroll
            for (int t = 0; t < GS; ++t) {
                const uint32_t s = order[g * GS + t];
                double ks = 0.;
                const double dd = dist[t];
                if (kd[t] == LK_SOIL_LAT) {                                  /* redistribution, water.cpp:542-562 */
                    const double ki = Ki * lvRatio, kn = kj[t] * lvRatio;
                    ks = qdiv(mean_of(ki, kn, meanType) * area[t], dd);
                } else if (kd[t] == LK_SOIL_VERT) {
                    ks = qdiv(mean_of(Ki, kj[t], meanType) * area[t], dd);
                } else if (kd[t] == LK_INFILTRATION) {                       /* the surface node above (layer 1) */
                    ks = infiltration_conductance(v, c, i, j[t], (size_t)s * v.N + i, Xc, Xh, Xc[i], Hoi, v.z[i]);
                }                                                            /* a soil row has no runoff link */
                if (HEAT && (kd[t] == LK_SOIL_LAT || kd[t] == LK_SOIL_VERT))
                    add_thermal_fluxes(v.heat, i, j[t], area[t], v.heat.hdist[(size_t)s * v.N + i], invFlux);
                k[s] = ks;
                sum += ks;
#if SF3D_ASM_SCHED_BARRIER
                if ((t + 1) % SF3D_ASM_SCHED_BARRIER == 0)
                    __builtin_amdgcn_sched_barrier(0);  /* one link's divide -> log -> divide chain at a time (every N-th link with
                                                         * -DSF3D_ASM_SCHED_BARRIER=N): the scheduler would otherwise interleave
                                                         * the five of a group and run out of registers (371 us instead of 304) */
#endif
            }
        }
        store_row<NT>(v, A2w, cd, i, k, sum, Hoi, dt, invFlux, Ci, flowi);
    }
}

/* one launch: blocks [0, nbSurf) assemble the surface rows (and reduce the Courant maximum),
 * blocks [nbSurf, nbSurf + nbSoil) the soil rows.  FUSED: the block that arrives last takes the
 * Courant decision (checkCourant) instead of a separate one-block kernel. */
template <bool FUSED, bool NT, bool HEAT>
__device__ __forceinline__ void body_assemble(const DevView& v)
{
    fm_init();
    double bm = 0.;
    if (blockIdx.x < v.nbSurf) bm = assemble_surface_rows<NT, HEAT>(v, blockIdx.x, v.nbSurf);
    else assemble_soil_rows<NT, HEAT>(v, blockIdx.x - v.nbSurf, v.nbSoil);
    if (!FUSED) {
        if (threadIdx.x == 0 && blockIdx.x < v.nbSurf) v.part0[blockIdx.x] = bm;
        return;
    }
    __syncthreads();
    if (!arrive_last(v, bm, 0., false)) return;
    double m = 0.;                                        /* soil blocks published 0: the maximum is unchanged */
    for (uint32_t k = threadIdx.x; k < gridDim.x; k += SF3D_BLOCK)
        m = dmax(m, __hip_atomic_load(&v.part0[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    double vals[3] = {block_max(m), 0., 0.};
    if (!dist_allgather(v, v.ctrl, vals, 1)) return;
    if (threadIdx.x == 0) courant_decision(v.ctrl, vals[0]);
}
template <bool FUSED, bool NT, bool HEAT>
__global__ void __launch_bounds__(SF3D_BLOCK, HEAT ? SF3D_ASM_HEAT_WAVES : SF3D_ASM_WAVES) k_assemble(DevView v)
{
    if (v.ctrl->stage != ST_APPROX) return;
    body_assemble<FUSED, NT, HEAT>(v);
}

#include "sf3d_cg.inc"          /* k_cg_*: preconditioned conjugate gradients standing in for the linealia hook */

/* quirk-1 compat only.  Mirrors the assembly that has just been decided into the emulated row storage of the reference
 * (computeLinearSystemElement cpusolver.cpp:348-389: every existing link writes its conductance at the running column, only a
 * non-zero one advances it; columns [1, numCols) negated, [0] = diagonal) and then does what the reference's separate
 * preconditioningMatrix pass does (:284-305) - unless the Courant check refused the attempt, in which case the reference had
 * assembled the surface rows only and left them un-normalised. */
__global__ void __launch_bounds__(SF3D_BLOCK) k_compat_rows(DevView v)
{
    Ctrl* c = v.ctrl;
    if (c->asmSeq == c->compatSeq) return;
    const bool surfOnly = c->asmSurfOnly != 0;
    constexpr uint32_t order[SF3D_SLOTS] = {0, 2, 3, 4, 5, 6, 7, 8, 9, 1};
    const size_t N = v.N;
    sf3d_d2* __restrict__ A2c = cur_A2(v);
    FOR_EACH_CHUNK_IN(v, 0u, surfOnly ? v.nListSurf : v.nList) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        if (NOT_MINE(v, i) || (surfOnly && i >= v.ns)) continue;
        const ChunkDesc cd = v.cdesc[q];
        double a[SF3D_SLOTS];
        #pragma unroll
        for (int p = 0; p < SF3D_SLOTS / 2; ++p) { const sf3d_d2 t = A2c[(size_t)p * N + i]; a[2 * p] = t.x; a[2 * p + 1] = t.y; }
        const double diag = v.compatDiag[i];
        const double inv = surfOnly ? 1.0 : 1.0 / diag;
        uint32_t col = 1; bool lastDropped = false;
        #pragma unroll
        for (int o = 0; o < SF3D_SLOTS; ++o) {
            const uint32_t s = order[o];
            const bool exists = (cd.kind[s] == CK_MIXED) ? v.lkind[(size_t)s * N + i] != LK_NONE : cd.kind[s] != CK_NONE;
            if (!exists) continue;
            lastDropped = !(a[s] != 0.);                       /* a = k * -1: -0.0 for a dropped link */
            if (!lastDropped) { v.compatCv[(size_t)col * N + i] = a[s] * inv; ++col; }
        }
        if (lastDropped) v.compatCv[(size_t)col * N + i] = 0.0;   /* the conductance itself, written and never negated */
        v.compatCn[i] = (uint8_t)col;
        v.compatCv[i] = surfOnly ? diag : 1.0;
        if (!surfOnly) {
            #pragma unroll
            for (int p = 0; p < SF3D_SLOTS / 2; ++p) { sf3d_d2 t; t.x = a[2 * p] * inv; t.y = a[2 * p + 1] * inv; A2c[(size_t)p * N + i] = t; }
            v.b[i] *= inv;
        }
    }
    __syncthreads();
    if (!arrive_last(v, 0., 0., false)) return;
    if (threadIdx.x == 0) c->compatSeq = c->asmSeq;
}

/* acceptStep's link flow sums for one row (water.cpp:240-250 + updateLinkFlux :269-277): the stored, row-normalised coefficient as in
 * the reference.  A zero coefficient adds exactly 0 - unless the compat storage exists: then a dropped
 * link reads the slot after the row's last column, like getMatrixElementValue (cpusolver.h:42-52). */
template <bool NT>
__device__ __forceinline__ void accept_links_row(const DevView& v, const sf3d_d2* __restrict__ A2, uint32_t q, uint32_t i, const double* __restrict__ X, double dt)
{
    double a[SF3D_SLOTS], xj[SF3D_SLOTS], f[SF3D_SLOTS];
    uint32_t j[SF3D_SLOTS];
    #pragma unroll
    for (int p = 0; p < SF3D_SLOTS / 2; ++p) { const sf3d_d2 t = load_coeff<NT>(&A2[(size_t)p * v.N + i]); a[2 * p] = t.x; a[2 * p + 1] = t.y; }   /* non-zero only where a link exists */
    const ChunkDesc cd = v.cdesc[q];
    #pragma unroll
    for (int s = 0; s < SF3D_SLOTS; ++s) {
        if (cd.kind[s] != CK_MIXED) j[s] = i + cd.delta[s];
        else if ((cd.sweepUniform >> s) & 1u) { const uint32_t jj = i + (uint32_t)cd.delta[s]; j[s] = jj < v.N ? jj : i; }   /* a node without the link has a zero coefficient */
        else j[s] = v.lto[(size_t)s * v.N + i];
    }
    if (v.compatCv != nullptr) {
        const double stale = v.compatCv[(size_t)v.compatCn[i] * v.N + i] * v.compatCv[i];
        #pragma unroll
        for (int s = 0; s < SF3D_SLOTS; ++s) {
            const bool exists = (cd.kind[s] == CK_MIXED) ? v.lkind[(size_t)s * v.N + i] != LK_NONE : cd.kind[s] != CK_NONE;
            if (exists && !(a[s] != 0.)) a[s] = stale;
        }
    }
    const double Hi = X[i];
    #pragma unroll
    for (int s = 0; s < SF3D_SLOTS; ++s) { xj[s] = X[j[s]]; f[s] = (a[s] != 0.) ? load_stream<NT>(&v.lflowSum[(size_t)s * v.N + i]) : 0.; }
    #pragma unroll
    for (int s = 0; s < SF3D_SLOTS; ++s)
        if (a[s] != 0.) store_stream<NT>(&v.lflowSum[(size_t)s * v.N + i], f[s] + a[s] * (Hi - xj[s]) * dt);
}

/* JacobiWaterCPU, water.cpp:565-601.
 * All coefficient loads, then all neighbour gathers, are issued before the ordered accumulation
 * so that ~30 independent loads per lane are in flight (HBM-bound kernel, 152 algorithmic B/node). */
/* MODE 0: partials only (a separate k_decide_sweep follows).  MODE 1 (single GPU): the block that
 * arrives last also takes the convergence decision.  MODE 2 (multi GPU): each wave additionally puts
 * the new iterate of its boundary nodes into the neighbours' windows, and the last block all-gathers
 * the norm, copies the received halo and decides - one launch per sweep in every configuration. */
template <int MODE, bool NT>
__device__ __forceinline__ void body_sweep(const DevView& v)
{
    const Ctrl* c = v.ctrl;
    const int nxt = free_buffer(c);
    const uint32_t par = c->epoch & 1u;
    const double* __restrict__ xin = v.X[c->cur];
    double* __restrict__ xout = v.X[nxt];
    const sf3d_d2* __restrict__ A2 = cur_A2(v);
    double nrm = 0.;
    FOR_EACH_CHUNK(v) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        const bool mine = !NOT_MINE(v, i);
        double xn = 0.;
        if (mine) {
            double a[SF3D_SLOTS], xj[SF3D_SLOTS];
            uint32_t j[SF3D_SLOTS];
            #pragma unroll
            for (int p = 0; p < SF3D_SLOTS / 2; ++p) { const sf3d_d2 t = load_coeff<NT>(&A2[(size_t)p * v.N + i]); a[2 * p] = t.x; a[2 * p + 1] = t.y; }
            const ChunkDesc cd = v.cdesc[q];                                     /* wave-uniform: scalar load */
            #pragma unroll
            for (int s = 0; s < SF3D_SLOTS; ++s) {
                if (cd.kind[s] != CK_MIXED) j[s] = i + cd.delta[s];              /* offset 0 when the slot is empty */
                else if ((cd.sweepUniform >> s) & 1u) {                          /* row end: the nodes without the link have a zero coefficient */
                    const int64_t jj = (int64_t)i + cd.delta[s];
                    j[s] = (jj < 0 || jj >= (int64_t)v.N) ? i : (uint32_t)jj;
                } else j[s] = load_stream<NT>(&v.lto[(size_t)s * v.N + i]);       /* 0 for a missing link: in range */
            }
            const double bi = v.b[i], zi = v.z[i], xi = xin[i];
            #pragma unroll
            for (int s = 0; s < SF3D_SLOTS; ++s) xj[s] = xin[j[s]];
            if (MODE == 2 && v.haloDirect && cd.pad0 && c->iter > 0) {
                /* multi GPU, second and later sweeps of an approximation: the neighbours' previous iterate sits in my window
                 * (they put it during their previous sweep, parity of the previous epoch) and has not been copied into xin -
                 * the halo copy is done once per approximation by k_post, not once per sweep on the critical path */
                const DistView* d = v.dist;
                const uint32_t parPrev = par ^ 1u;
                #pragma unroll
                for (int s = 0; s < SF3D_SLOTS; ++s) {
                    const uint32_t f = d->fsrc[(size_t)s * v.N + i];
                    if (f != SF3D_FSRC_NONE) {
                        const uint32_t p = f >> 27, k = f & 0x07FFFFFFu;
                        xj[s] = SYS_LOAD(d->payload[v.rank] + d->recvOff[p] + (size_t)(parPrev * SF3D_DIST_FIELDS + DF_X) * d->recvCount[p] + k);
                    }
                }
            }
            xn = bi;
            constexpr uint32_t order[SF3D_SLOTS] = {0, 2, 3, 4, 5, 6, 7, 8, 9, 1};
            #pragma unroll
            for (int o = 0; o < SF3D_SLOTS; ++o) {
                const uint32_t s = order[o];
                if (a[s] != 0.) xn -= a[s] * xj[s];                              /* zero entries are not in the reference's row */
            }
            if (i < v.ns) xn = dmax(xn, zi);
            double d = fabs(xn - xi);
            const double psi = fabs(xn - zi);
            if (psi > 1.) d *= (1. / psi);
            nrm += d;
            xout[i] = xn;
        }
        if (MODE == 2) dist_put_chunk(v, q, lane_, par, 0, xn);
    }
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* every wave drains its puts before the barrier */
    const double bs = block_sum(nrm);
    if (MODE == 0) {
        if (threadIdx.x == 0) v.part0[blockIdx.x] = bs;
        return;
    }
    if (!arrive_last(v, bs, 0., false)) return;
    double vals[3] = {sum_published(v.part0, gridDim.x), 0., 0.};
    if (MODE == 2) {
        if (!dist_allgather(v, v.ctrl, vals, 0)) return;                  /* also the barrier that makes the neighbours' puts visible */
        if (!v.haloDirect) dist_unpack(v, par, DF_X, v.X[nxt]);           /* neighbours' new iterate on my halo */
    }
    if (threadIdx.x == 0) { v.ctrl->singleLaunches++; sweep_decision(v.ctrl, nxt, vals[0] / v.Nnorm); }
}
template <int MODE, bool NT>
__global__ void __launch_bounds__(SF3D_BLOCK) k_sweep(DevView v)
{
    if (v.ctrl->stage != ST_SWEEP) return;
    body_sweep<MODE, NT>(v);
}

#include "sf3d_pair.inc"        /* k_sweep_pair: two Jacobi iterations per pass through an LDS ring */
#include "sf3d_pair_masked.inc" /* k_sweep_pair_masked: the same on layered masked grids (DEM outlines) */

__device__ __forceinline__ void balance_terms(const DevView& v, const Ctrl* c, uint32_t i, double H, double z,
                                              double Se, double& st, double& sk)
{
    double theta;
    if (i >= v.ns) {
        const SoilDev s = v.soils[v.cls[i]];
        theta = (Se * (s.thetaS - s.thetaR)) + s.thetaR;      /* soilPhysics.cpp:38-42 */
    } else theta = dmax(H - z, 0.0);                          /* water.cpp:83 */
    st += theta * v.size[i];
    const double fl = v.flow[i];
    if (fl != 0) sk += fl * c->dt;                            /* water.cpp:136-137 */
}

/* cpusolver.cpp:451-457 (H = x is implicit: H is the current pool buffer) + the two sums of
 * computeCurrentMassBalance (water.cpp:71-90, 130-140) */
#ifndef SF3D_POST_WAVES
#define SF3D_POST_WAVES 8      /* 64 VGPRs: all 2 048 blocks of the grid resident at once (at 70 VGPRs 1 792 are, and the rest form a tail) */
#endif
template <bool FUSED>
__device__ __forceinline__ void body_post(const DevView& v)
{
    const Ctrl* c = v.ctrl;
    fm_init();
    const int cur = c->cur;
    const uint32_t parLastSweep = (c->epoch - 1u) & 1u;       /* the last sweep put its iterate one epoch ago */
    const double* __restrict__ Xc = v.X[cur];
    double st = 0., sk = 0.;
    {   /* software-pipelined chunk loop: the loads of the next chunk (list entry, H, z, class) are issued before the two
         * pow of the current one - the kernel is latency-bound (VALU 48 % busy), three dependent round trips per chunk */
        const uint32_t lane_ = threadIdx.x & 63u;
        const uint32_t wavesTotal_ = gridDim.x * (SF3D_BLOCK / 64);
        uint32_t li_ = __builtin_amdgcn_readfirstlane(blockIdx.x * (SF3D_BLOCK / 64) + (threadIdx.x >> 6));
        uint32_t i = 0; double H = 0., z = 0.; uint16_t cl = 0; bool on = false;
        if (li_ < v.nList) {
            i = __builtin_amdgcn_readfirstlane(v.chunkList[li_]) * SF3D_CHUNK + lane_;
            on = !NOT_MINE(v, i);
            if (on) { H = Xc[i]; z = v.z[i]; cl = v.cls[i]; }
        }
        while (li_ < v.nList) {
            const uint32_t lin = li_ + wavesTotal_;
            uint32_t i2 = 0; double H2 = 0., z2 = 0.; uint16_t cl2 = 0; bool on2 = false;
            if (lin < v.nList) {
                i2 = __builtin_amdgcn_readfirstlane(v.chunkList[lin]) * SF3D_CHUNK + lane_;
                on2 = !NOT_MINE(v, i2);
                if (on2) { H2 = Xc[i2]; z2 = v.z[i2]; cl2 = v.cls[i2]; }
            }
            if (on) {
                double Se = 1.;
                if (i >= v.ns) { Se = node_se(v.soils[cl], H, z, c->wrc); v.Se[i] = Se; }
                balance_terms(v, c, i, H, z, Se, st, sk);
            }
            li_ = lin; i = i2; H = H2; z = z2; cl = cl2; on = on2;
        }
    }
    const double a = block_sum(st), b = block_sum(sk);
    if (!FUSED) {
        if (threadIdx.x == 0) { v.part0[blockIdx.x] = a; v.part1[blockIdx.x] = b; }
        return;
    }
    if (!arrive_last(v, a, b, true)) return;              /* last block: evaluateWaterBalance */
    double vals[3] = {sum_published(v.part0, gridDim.x), sum_published(v.part1, gridDim.x), 0.};
    if (!dist_allgather(v, v.ctrl, vals, 0)) return;
    if (threadIdx.x == 0) {
        if (v.world > 1 && v.haloDirect) {                /* the halo of the final iterate, once per approximation (k_halo_copy<1>, */
            Ctrl* w = v.ctrl;                             /* next launch): the sweeps read foreign neighbours from the window       */
            w->haloBuf = cur; w->haloPar = parLastSweep; w->haloEpoch = w->epoch;
        }
        balance_decision(v.ctrl, vals[0], vals[1]);
    }
}
template <bool FUSED>
__global__ void __launch_bounds__(SF3D_BLOCK, SF3D_POST_WAVES) k_post(DevView v)
{
    if (v.ctrl->stage != ST_POST) return;
    body_post<FUSED>(v);
}

/* restoreBestStep, water.cpp:253-267 */
template <bool FUSED, bool HEAT>
__device__ __forceinline__ void body_restore(const DevView& v)
{
    const Ctrl* c = v.ctrl;
    fm_init();
    const double* __restrict__ Xc = v.X[c->cur];
    const double* __restrict__ Xh = v.X[c->hold];
    const uint32_t par = c->epoch & 1u;
    const bool haloK = HEAT && FUSED && v.world > 1;      /* sharded heat: saveWaterFluxValues reads the neighbours' K */
    double st = 0., sk = 0.;
    FOR_EACH_CHUNK(v) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        double K = 0.;
        if (!NOT_MINE(v, i)) {
            const double H = Xc[i], Ho = Xh[i], z = v.z[i];
            double Se = 1.;
            if (i >= v.ns) {
                const SoilDev s = v.soils[v.cls[i]];
                Se = node_se(s, H, z, c->wrc);
                K = mualem_k(s, Se, c->wrc);
                if (HEAT && v.heat.vapor) {
                    const HeatDev& hv = v.heat;
                    const double Tm = (hv.TX[c->tCur][i] + hv.TX[c->tOld][i]) * 0.5;
                    K += h_isothermal_vapor_conductivity(s, Tm, H - z, h_theta_signed_psi(s, H - z, c->wrc)) * (H_G / H_RHOW);
                }
                v.Se[i] = Se; v.K[i] = K;
            }
            boundary_update<HEAT>(v, c, i, H, Ho, z, K, Se);
            balance_terms(v, c, i, H, z, Se, st, sk);
        }
        if (haloK) dist_put_chunk(v, q, lane_, par, DF_K, K);
    }
    if (haloK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const double a = block_sum(st), b = block_sum(sk);
    if (!FUSED) {
        if (threadIdx.x == 0) { v.part0[blockIdx.x] = a; v.part1[blockIdx.x] = b; }
        return;
    }
    if (!arrive_last(v, a, b, true)) return;
    double vals[3] = {sum_published(v.part0, gridDim.x), sum_published(v.part1, gridDim.x), 0.};
    if (!dist_allgather(v, v.ctrl, vals, 0)) return;
    if (haloK) dist_unpack(v, par, DF_K, v.K);
    if (threadIdx.x == 0) restore_decision(v.ctrl, vals[0], vals[1]);
}
template <bool FUSED, bool HEAT>
__global__ void __launch_bounds__(SF3D_BLOCK) k_restore(DevView v)
{
    if (v.ctrl->stage != ST_RESTORE) return;
    body_restore<FUSED, HEAT>(v);
}

/* acceptStep flow sums, water.cpp:240-250 + updateLinkFlux :269-277 */
template <bool NT>
__device__ __forceinline__ void body_accept(const DevView& v)
{
    const Ctrl* c = v.ctrl;
    const double* __restrict__ X = v.X[c->cur];
    const sf3d_d2* __restrict__ A2 = cur_A2(v);
    const double dt = c->dt;
    FOR_EACH_CHUNK(v) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        if (NOT_MINE(v, i)) continue;
        accept_links_row<NT>(v, A2, q, i, X, dt);
        if (v.btype[i] != SF3D_BND_NONE) v.bflowSum[i] += v.bflowRate[i] * dt;
    }
}
template <bool NT>
__global__ void __launch_bounds__(SF3D_BLOCK) k_accept(DevView v)
{
    if (v.ctrl->stage != ST_ACCEPT) return;
    body_accept<NT>(v);
}


/* The two halves of k_accept for the overlapped mode: the boundary sums stay in the step (the next step's k_props
 * overwrites bflowRate); the link sums - 1.5 GB of traffic at C4, 6 % VALU - run on a second stream next to the next
 * step's k_props (68 % VALU, little traffic), which must only finish before that step's first k_assemble rewrites A2. */
__global__ void __launch_bounds__(SF3D_BLOCK) k_accept_boundary(DevView v)
{
    const Ctrl* c = v.ctrl;
    if (c->stage != ST_ACCEPT) return;
    const double dt = c->dt;
    FOR_EACH_CHUNK(v) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        if (NOT_MINE(v, i)) continue;
        if (v.btype[i] != SF3D_BND_NONE) v.bflowSum[i] += v.bflowRate[i] * dt;
    }
}
template <bool NT>
__global__ void __launch_bounds__(SF3D_BLOCK) k_accept_links(DevView v, int xbuf, uint32_t abuf, double dt)
{
    /* which head buffer, which copy of the matrix and which dt the accepted step had come as ARGUMENTS (the host has just polled
     * them): this kernel runs next to the whole following computeStep, which overwrites those fields of the control block */
    const double* __restrict__ X = v.X[xbuf];
    const sf3d_d2* __restrict__ A2 = v.A2x[abuf & 1u];      /* the accepted step's matrix: the step in progress writes the other copy */
    FOR_EACH_CHUNK(v) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        if (NOT_MINE(v, i)) continue;
        accept_links_row<NT>(v, A2, q, i, X, dt);
    }
}

/* computeTotalWaterContent on the stored state (getTotalWaterContent / initializeBalance) */
__global__ void __launch_bounds__(SF3D_BLOCK) k_storage(DevView v)
{
    const Ctrl* c = v.ctrl;
    const double* __restrict__ Xc = v.X[c->cur];
    double st = 0.;
    FOR_EACH_CHUNK(v) {
        const uint32_t i = q * SF3D_CHUNK + lane_;
        if (NOT_MINE(v, i)) continue;
        const double H = Xc[i], z = v.z[i];
        double theta;
        if (i >= v.ns) { const SoilDev s = v.soils[v.cls[i]]; theta = (v.Se[i] * (s.thetaS - s.thetaR)) + s.thetaR; }
        else theta = dmax(H - z, 0.0);
        st += theta * v.size[i];
    }
    const double a = block_sum(st);
    if (threadIdx.x == 0) v.part0[blockIdx.x] = a;
}

/* ======================================================================================= */
/* host side                                                                                */
/* ======================================================================================= */

namespace {

template <class F> void parallel_for(uint32_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 16) nt = 16;
    if (n < 65536 || nt == 1) { f(0u, n); return; }
    std::vector<std::thread> th;
    const uint32_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const uint32_t a = t * chunk, b = (a + chunk < n) ? a + chunk : n;
        if (a >= b) break;
        th.emplace_back([=] { f(a, b); });
    }
    for (auto& t : th) t.join();
}

const char* kKernelNames[KID_COUNT] = {"k_props", "k_assemble", "k_sweep", "k_post", "k_restore", "k_accept", "k_sweep_pair"};

}  // namespace

/* ---- row-strip partition (host logic; also used by CPU tests through sf3d_dist_owner/halo) ---- */
sf3d_error_t sf3d_compute_partition(const HostModel& m, int rank, int world, Partition& out)
{
    const uint32_t N = m.N, ns = m.ns;
    if (world < 1 || world > SF3D_MAX_RANKS || rank < 0 || rank >= world) return SF3D_PARAMETER_ERROR;
    out.world = world; out.rank = rank;
    out.owner.assign(N, 0);
    out.bounds.assign(world + 1, 0);
    out.send.assign(world, {}); out.recv.assign(world, {});
    for (int r = 0; r <= world; ++r) {
        uint64_t b = (uint64_t)ns * r / world;
        if (r > 0 && r < world) b = (b / SF3D_CHUNK) * SF3D_CHUNK;      /* cut at chunk boundaries */
        out.bounds[r] = (uint32_t)b;
    }
    out.bounds[world] = ns;
    for (int r = 1; r <= world; ++r) if (out.bounds[r] < out.bounds[r - 1]) out.bounds[r] = out.bounds[r - 1];
    if (world == 1) return SF3D_OK;
    /* column id = surface ancestor through Up links (layer-major numbering: one pass suffices) */
    std::vector<uint32_t> col(N, UINT32_MAX);
    for (uint32_t i = 0; i < ns && i < N; ++i) col[i] = i;
    for (int pass = 0; pass < 64; ++pass) {
        bool changed = false, missing = false;
        for (uint32_t i = ns; i < N; ++i) {
            if (col[i] != UINT32_MAX) continue;
            if (m.ltype[0][i] == SF3D_LINK_NONE) { col[i] = 0; changed = true; continue; }   /* orphan: rank 0 */
            const uint32_t up = m.lto[0][i];
            if (col[up] != UINT32_MAX) { col[i] = col[up]; changed = true; } else missing = true;
        }
        if (!missing || !changed) break;
    }
    for (uint32_t i = 0; i < N; ++i) {
        const uint32_t cidx = col[i] == UINT32_MAX ? 0u : col[i];
        int r = 0;
        while (r + 1 < world && cidx >= out.bounds[r + 1]) ++r;
        out.owner[i] = (uint8_t)r;
    }
    for (int s = 0; s < SF3D_SLOTS; ++s)
        for (uint32_t i = 0; i < N; ++i) {
            if (m.ltype[s][i] == SF3D_LINK_NONE) continue;
            if (s >= 2 && s - 2 >= m.nLat[i]) continue;
            const uint32_t j = m.lto[s][i];
            const int a = out.owner[i], b = out.owner[j];
            if (a == b) continue;
            if (a == rank) out.recv[b].push_back(j);       /* my row i reads node j of rank b */
            if (b == rank) out.send[a].push_back(j);       /* rank a's row i reads my node j   */
        }
    for (int r = 0; r < world; ++r) {
        for (auto* v : {&out.send[r], &out.recv[r]}) {
            std::sort(v->begin(), v->end());
            v->erase(std::unique(v->begin(), v->end()), v->end());
        }
    }
    return SF3D_OK;
}

__global__ void k_dist_ping(DistView d, unsigned long long token, long long timeoutTicks, int* out);

struct DeviceSolver::Impl {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;         /* link flow sums of the accepted step, next to the next step's k_props */
    hipEvent_t evLinks[2] = {nullptr, nullptr};   /* one per copy of the matrix: recorded after the link sums that read it */
    bool linksPending[2] = {false, false};
    int overlapAccept = -1;                /* SF3D_OVERLAP_ACCEPT=0 keeps k_accept inside the step */
    DevView v{};
    Ctrl* hostCtrl = nullptr;              /* pinned */
    std::vector<void*> allocs;
    uint32_t N = 0, ns = 0;
    uint32_t lastSweeps = 8;
    uint32_t lastBatches = 1;
    uint32_t predCount = 0, pred[16] = {0};   /* sweeps each approximation of the previous computeStep took (Ctrl::seqSweeps) */
    uint32_t lastHeatSweeps = 8;
    std::vector<uint32_t> gsLevelStart;    /* SF3D_HEAT_GS=1: level l = gsOrder[gsLevelStart[l] .. gsLevelStart[l+1]) */
    uint32_t lastHeatSteps = 1;            /* heat steps (accepted + halved) of the previous computeStep: look-ahead depth */
    double* heatOut[6] = {nullptr};       /* bAero, bSoilCond, bSens, bLat, bRad, bAdv (device) */
    double* heatAirP = nullptr;            /* filled by k_heat_static once z is on the device */
    /* hipGraph cache: one instantiated graph per (with head part, number of queued sweeps) */
    std::vector<std::pair<uint32_t, hipGraphExec_t>> graphs;
    int useFused = -1;                     /* SF3D_FUSED_DECIDE=0 keeps the separate decision kernel */
    int useGraphs = -1;                    /* -1 unknown, 0 off (SF3D_GRAPHS=0), 1 on */
    /* multi-GPU */
    Partition part;
    DistWindow* window = nullptr;          /* fine-grained, IPC-exported */
    size_t windowBytes = 0;
    std::vector<void*> peerMaps;           /* hipIpcOpenMemHandle results */
    DistView hostDist{};
    DistView* devDist = nullptr;
    uint32_t pushBlocks = 0;
    /* RCCL exchange (SF3D_EXCHANGE=rccl, or agreed fall-back when the windows fail): resolved with dlsym, no link-time dependency */
    void* rcclLib = nullptr; ncclComm_t comm = nullptr; bool rcclMode = false;
    ncclResult_t (*pGetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*pCommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*pCommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*pSend)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*pRecv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*pAllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*pAllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*pGroupStart)() = nullptr; ncclResult_t (*pGroupEnd)() = nullptr;
    double *rcclMine = nullptr, *rcclGathered = nullptr;
    bool rcclDistinct = false; unsigned char rcclId[128] = {0};
    bool load_rccl()
    {
        if (pAllGather) return true;
        if (!rcclLib) rcclLib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!rcclLib) rcclLib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!rcclLib) return false;
        #define SF3D_SYM(ptr, name) ptr = reinterpret_cast<decltype(ptr)>(dlsym(rcclLib, name)); if (!ptr) return false
        SF3D_SYM(pGetUniqueId, "ncclGetUniqueId"); SF3D_SYM(pCommInitRank, "ncclCommInitRank"); SF3D_SYM(pCommDestroy, "ncclCommDestroy");
        SF3D_SYM(pSend, "ncclSend"); SF3D_SYM(pRecv, "ncclRecv"); SF3D_SYM(pAllReduce, "ncclAllReduce");
        SF3D_SYM(pGroupStart, "ncclGroupStart"); SF3D_SYM(pGroupEnd, "ncclGroupEnd"); SF3D_SYM(pAllGather, "ncclAllGather");
        #undef SF3D_SYM
        return true;
    }
    /* halo of one or two fields to and from every neighbour, then (optionally) the all-gather of the partial sums */
    void rccl_halo(hipStream_t st, int fields)
    {
        pGroupStart();
        for (int p = 0; p < hostDist.world; ++p) {
            if (p == hostDist.rank) continue;
            if (hostDist.sendCount[p]) pSend(hostDist.sendBuf[p], (size_t)hostDist.sendCount[p] * fields, ncclDouble, p, comm, st);
            if (hostDist.recvCount[p]) pRecv(const_cast<double*>(hostDist.recvBuf[p]), (size_t)hostDist.recvCount[p] * fields, ncclDouble, p, comm, st);
        }
        pGroupEnd();
    }
    void rccl_gather(hipStream_t st) { pAllGather(rcclMine, rcclGathered, 3, ncclDouble, comm, st); }
    /* device slot of every (host slot, node): empty = identity.  Laterals of nodes with fewer links than their chunk's fullest node are
     * moved to the slots where that node keeps the same neighbour offset (sync_to_device), so that a slot means one direction for the
     * whole chunk; host arrays (HostModel, what the API reads and writes) stay in insertion order, the permutation is applied where
     * per-link arrays cross the boundary (flow sums up and down, heat link fluxes down). */
    std::vector<uint8_t> dslot;
    std::vector<double> stage;             /* N x 10 staging area of those transfers */
    uint32_t connectGen = 0;               /* token of the window self-check */
    bool warnedShared = false;
    /* timing */
    int timing = 0;                       /* 0 off, 1 every node kernel, 2 only k_sweep, on every 8th step */
    std::vector<std::pair<const void*, uint32_t>> residentBlocks;   /* kernel -> blocks resident at once (occupancy x CUs) */
    int residentGrids = -1;               /* SF3D_RESIDENT_GRIDS=0: every kernel with the common 2 048-block grid */
    uint32_t pairBlocks = 0;              /* grid of k_sweep_pair (0: the graph is no regular grid, or the paired sweep is off) */
    bool pairMasked = false;              /* the paired sweep runs as k_sweep_pair_masked (layered masked grid) */
    uint64_t stepSeq = 0;
    struct Pair { hipEvent_t a, b; int kid; };
    std::vector<Pair> pending;             /* pairs of the batch in flight, in launch order */
    std::vector<hipEvent_t> freeEvents;
    uint64_t launches[KID_COUNT] = {0};
    double ms[KID_COUNT] = {0};
};

/* on failure: message, then drain the solver stream (async copies from pageable host vectors may still be in flight and the
 * caller is free to modify those vectors as soon as we return), mark the solver unusable, return */
#define HIP_TRY(expr)                                                                          \
    do { hipError_t e_ = (expr);                                                               \
         if (e_ != hipSuccess) {                                                               \
             snprintf(err_, sizeof(err_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
             if (impl_ && impl_->stream) (void)hipStreamSynchronize(impl_->stream);           \
             fatal_ = true;                                                                    \
             return SF3D_SOLVER_ERROR; } } while (0)

DeviceSolver& DeviceSolver::instance() { static DeviceSolver s; return s; }
const char* DeviceSolver::kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kKernelNames[kid] : nullptr; }

sf3d_error_t DeviceSolver::set_device(int dev)
{
    if (!impl_) impl_ = new Impl();
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (dev < 0 || dev >= count) { snprintf(err_, sizeof(err_), "device %d out of range (%d visible)", dev, count); return SF3D_SOLVER_ERROR; }
    impl_->device = dev;
    HIP_TRY(hipSetDevice(dev));
    return SF3D_OK;
}

static uint64_t g_deviceBytes = 0;          /* bytes behind the current model's allocations (sf3d_device_bytes) */
sf3d_error_t DeviceSolver::release()
{
    if (!impl_) return SF3D_OK;
    Impl& I = *impl_;
    if (I.stream) hipStreamSynchronize(I.stream);
    if (I.stream2) hipStreamSynchronize(I.stream2);
    I.linksPending[0] = I.linksPending[1] = false;
    for (auto& p : I.pending) { I.freeEvents.push_back(p.a); I.freeEvents.push_back(p.b); }
    I.pending.clear();
    for (auto& g : I.graphs) hipGraphExecDestroy(g.second);
    I.graphs.clear();
    for (void* p : I.allocs) hipFree(p);
    I.allocs.clear();
    g_deviceBytes = 0;
    if (I.comm && I.pCommDestroy) { I.pCommDestroy(I.comm); I.comm = nullptr; }
    I.rcclMode = false; I.rcclMine = I.rcclGathered = nullptr;
    for (void* p : I.peerMaps) hipIpcCloseMemHandle(p);
    I.peerMaps.clear();
    if (I.window) { hipFree(I.window); I.window = nullptr; }
    I.devDist = nullptr;
    connected_ = false;
    built_ = false;
    fatal_ = false;
    /* the SF3D_* mode switches are read again when the next model is built (tests toggle them between models of one process) */
    I.overlapAccept = I.useFused = I.useGraphs = I.residentGrids = -1;
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::synchronize()
{
    if (!impl_ || !impl_->stream) return SF3D_OK;
    HIP_TRY(hipStreamSynchronize(impl_->stream));
    if (impl_->stream2) { HIP_TRY(hipStreamSynchronize(impl_->stream2)); impl_->linksPending[0] = impl_->linksPending[1] = false; }
    return SF3D_OK;
}

template <class T> static hipError_t dev_alloc(std::vector<void*>& allocs, T*& p, size_t count)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) { allocs.push_back(q); p = static_cast<T*>(q); g_deviceBytes += (uint64_t)(count ? count : 1) * sizeof(T); }
    return e;
}
uint64_t DeviceSolver::device_bytes() const { return g_deviceBytes; }

static void fill_params(Ctrl& c, const ParamsHost& p)
{
    c.MBRThreshold = p.MBRThreshold; c.residualTolerance = p.residualTolerance;
    c.dtMin = p.dtMin; c.dtMax = p.dtMax; c.lvRatio = p.lvRatio;
    c.courantThreshold = p.courantThreshold; c.instabilityFactor = p.instabilityFactor;
    c.maxApprox = p.maxApprox; c.maxIter = p.maxIter; c.wrc = p.wrc; c.meanType = p.meanType;
    c.dtCurr = p.dtCurr;
    c.lineal = p.lineal ? 1u : 0u;
}

sf3d_error_t DeviceSolver::sync_to_device(HostModel& m, const ParamsHost& p)
{
    if (!impl_) impl_ = new Impl();
    Impl& I = *impl_;
    if (I.device < 0) {
        int dev = 0;
        if (const char* lr = getenv("LOCAL_RANK")) dev = atoi(lr);
        int count = 0;
        HIP_TRY(hipGetDeviceCount(&count));
        if (count <= 0) { snprintf(err_, sizeof(err_), "no HIP device visible: the soilFluxes3D product path has no CPU fallback"); return SF3D_SOLVER_ERROR; }
        I.device = dev % count;
    }
    HIP_TRY(hipSetDevice(I.device));
    if (!I.stream) HIP_TRY(hipStreamCreateWithFlags(&I.stream, hipStreamNonBlocking));
    if ((I.linksPending[0] || I.linksPending[1]) && (m.graphDirty || m.stateDirty || m.flowSumsDirty || !built_)) { HIP_TRY(hipStreamSynchronize(I.stream2)); I.linksPending[0] = I.linksPending[1] = false; }
    if (!I.hostCtrl) HIP_TRY(hipHostMalloc((void**)&I.hostCtrl, sizeof(Ctrl), hipHostMallocDefault));

    const uint32_t N = m.N, ns = m.ns;
    const size_t NS = (size_t)N * SF3D_SLOTS;

    if (m.graphDirty || !built_) {
        /* the solver relies on surface nodes being exactly [0, ns) */
        for (uint32_t i = 0; i < N; ++i) {
            if ((m.surf[i] != 0) != (i < ns)) { snprintf(err_, sizeof(err_), "node %u: surface nodes must be exactly the first nrSurfaceNodes indices", i); return SF3D_TOPOGRAPHY_ERROR; }
            if (!m.hasClass[i]) { snprintf(err_, sizeof(err_), "node %u has no soil/surface class", i); return SF3D_MISSING_DATA_ERROR; }
        }
        if (world_ > 1 && connected_) { snprintf(err_, sizeof(err_), "topology changed after sf3d_dist_connect: call sf3d_dist_prepare / export / connect again"); return SF3D_TOPOGRAPHY_ERROR; }
        /* pull anything newer on the device before the arrays are re-created */
        if (built_) { if (m.hostStaleState) fetch_state(m); if (m.hostStaleFlows) fetch_flows(m); if (m.heat && m.hostStaleHeat && I.v.heat.on) fetch_heat(m); }
        release();
        I.N = N; I.ns = ns;
        DevView& v = I.v;
        v = DevView{};
        v.N = N; v.ns = ns; v.Nnorm = m.globalN ? m.globalN : N;
        v.nChunks = (N + SF3D_CHUNK - 1) / SF3D_CHUNK;
        v.qSplit = (ns + SF3D_CHUNK - 1) / SF3D_CHUNK;
        if (v.qSplit > v.nChunks) v.qSplit = v.nChunks;

        /* derived static graph data: link kind and link distance (host, libm - exactly the
         * reference's nodeDistance2D/3D arithmetic, soilPhysics.cpp:328-338) */
        std::vector<uint8_t> kind(NS, LK_NONE);
        std::vector<double> dist(NS, 0.), area(NS, 0.);
        std::vector<uint32_t> to(NS, 0u);
        std::atomic<bool> bad{false};
        {   /* Slot alignment.  setNodeLink puts a node's k-th lateral link into slot 2 + k, so a node at the edge of a grid (fewer
             * laterals) holds OTHER directions in its slots than its 63 chunk mates and the whole chunk falls back to per-lane link
             * kinds and indices (two dependent round trips per group of links in the assembly instead of one: a quarter of the soil
             * chunks of a 512-wide grid).  On the device such a node's laterals sit in the slots where the chunk's fullest node has the
             * same neighbour offset and link kind, in the same relative order - the sums of a row are taken over the existing links in
             * slot order, so every result keeps its bits.  Nodes that do not fit the pattern keep their own order. */
            I.dslot.clear();
            const char* re = getenv("SF3D_SLOT_ALIGN");
            if (!(re && re[0] == '0')) {
                std::vector<uint8_t> ds(NS);
                for (int sl = 0; sl < SF3D_SLOTS; ++sl) std::memset(ds.data() + (size_t)sl * N, sl, N);
                std::atomic<bool> moved{false};
                const uint32_t nq = (N + SF3D_CHUNK - 1) / SF3D_CHUNK;
                auto lateral_kind = [&](uint32_t i, int sl) -> int {      /* 0: no lateral link in host slot sl */
                    if (sl - 2 >= m.nLat[i] || m.ltype[sl][i] == SF3D_LINK_NONE) return 0;
                    const bool si = i >= ns, sj = m.lto[sl][i] >= ns;
                    return (si && sj) ? 1 + (m.ltype[sl][i] == SF3D_LINK_LATERAL ? 0 : 1) : ((!si && !sj) ? 3 : 4);
                };
                parallel_for(nq, [&](uint32_t qa, uint32_t qb) {
                    for (uint32_t q = qa; q < qb; ++q) {
                        const uint32_t i0 = q * SF3D_CHUNK, i1 = (i0 + SF3D_CHUNK < N) ? i0 + SF3D_CHUNK : N;
                        /* the pattern: the node with the most laterals, if they fill its slots 2 .. 2 + n - 1 */
                        uint32_t tmpl = i0; int tn = -1;
                        for (uint32_t i = i0; i < i1; ++i) {
                            int n = 0; bool dense = true;
                            for (int sl = 2; sl < SF3D_SLOTS; ++sl) { const bool on = lateral_kind(i, sl) != 0; if (on) { if (sl - 2 != n) dense = false; ++n; } }
                            if (dense && n > tn) { tn = n; tmpl = i; }
                        }
                        if (tn <= 0) continue;
                        int64_t td[SF3D_SLOTS]; int tk[SF3D_SLOTS];
                        for (int p = 0; p < tn; ++p) { td[p] = (int64_t)m.lto[2 + p][tmpl] - (int64_t)tmpl; tk[p] = lateral_kind(tmpl, 2 + p); }
                        for (uint32_t i = i0; i < i1; ++i) {
                            if (i == tmpl) continue;
                            uint8_t map[SF3D_SLOTS]; bool fits = true, differs = false; int p = 0;
                            for (int sl = 2; sl < SF3D_SLOTS && fits; ++sl) {
                                const int kk = lateral_kind(i, sl);
                                map[sl] = (uint8_t)sl;
                                if (!kk) continue;
                                const int64_t dd = (int64_t)m.lto[sl][i] - (int64_t)i;
                                while (p < tn && !(td[p] == dd && tk[p] == kk)) ++p;
                                if (p >= tn) { fits = false; break; }
                                map[sl] = (uint8_t)(2 + p); if (map[sl] != sl) differs = true;
                                ++p;
                            }
                            if (!fits || !differs) continue;
                            /* host slots without a link take the device slots that are left, so that the map stays a permutation */
                            bool used[SF3D_SLOTS] = {false};
                            for (int sl = 2; sl < SF3D_SLOTS; ++sl) if (lateral_kind(i, sl)) used[map[sl]] = true;
                            int free_ = 2;
                            for (int sl = 2; sl < SF3D_SLOTS; ++sl) {
                                if (lateral_kind(i, sl)) continue;
                                while (used[free_]) ++free_;
                                map[sl] = (uint8_t)free_; used[free_] = true;
                            }
                            for (int sl = 2; sl < SF3D_SLOTS; ++sl) ds[(size_t)sl * N + i] = map[sl];
                            moved = true;
                        }
                    }
                });
                if (moved) I.dslot.swap(ds);
            }
        }
        const uint8_t* dslot = I.dslot.empty() ? nullptr : I.dslot.data();
        parallel_for(N, [&](uint32_t a, uint32_t b) {
            for (uint32_t i = a; i < b; ++i) {
                const int nl = m.nLat[i];
                for (int s = 0; s < SF3D_SLOTS; ++s) {
                    if (s >= 2 && s - 2 >= nl) continue;            /* lateral loop bound, cpusolver.cpp:360 */
                    if (m.ltype[s][i] == SF3D_LINK_NONE) continue;
                    const uint32_t j = m.lto[s][i];
                    const size_t e = (size_t)(dslot ? dslot[(size_t)s * N + i] : s) * N + i;
                    to[e] = j; area[e] = m.larea[s][i];
                    const bool si = i >= ns, sj = j >= ns;
                    if (si && sj) {
                        if (m.ltype[s][i] == SF3D_LINK_LATERAL) {
                            kind[e] = LK_SOIL_LAT;
                            const double dx = m.x[i] - m.x[j], dy = m.y[i] - m.y[j], dz = m.z[i] - m.z[j];
                            double nrm = 0; nrm += dx * dx; nrm += dy * dy; nrm += dz * dz;
                            dist[e] = std::sqrt(nrm);
                        } else { kind[e] = LK_SOIL_VERT; dist[e] = std::fabs(m.z[i] - m.z[j]); }
                    } else if (!si && !sj) {
                        kind[e] = LK_RUNOFF;
                        const double dx = m.x[i] - m.x[j], dy = m.y[i] - m.y[j];
                        double nrm = 0; nrm += dx * dx; nrm += dy * dy;
                        dist[e] = std::sqrt(nrm);
                    } else {
                        kind[e] = LK_INFILTRATION;
                        const uint32_t su = sj ? i : j, so = sj ? j : i;
                        dist[e] = m.z[su] - m.z[so];
                    }
                }
                if (m.btype[i] == SF3D_BND_FREE_DRAINAGE && m.ltype[0][i] == SF3D_LINK_NONE
                    && (m.presetPartition == nullptr || m.presetPartition->owner[i] == rank_)) bad = true;   /* assert water.cpp:682 (a strip-local model's halo nodes may have lost the link: their rows and boundaries are the neighbour's) */
            }
        });
        if (bad) { snprintf(err_, sizeof(err_), "FreeDrainage node without an Up link"); return SF3D_BOUNDARY_ERROR; }

        /* per (64-node chunk, slot) descriptors: uniform kind + uniform index offset => the kernels
         * never read lto/lkind for that chunk and slot */
        const uint32_t nChunks = v.nChunks;
        std::vector<ChunkDesc> cdesc(nChunks);
        parallel_for(nChunks, [&](uint32_t qa, uint32_t qb) {
            for (uint32_t q = qa; q < qb; ++q) {
                const uint32_t i0 = q * SF3D_CHUNK, i1 = (i0 + SF3D_CHUNK < N) ? i0 + SF3D_CHUNK : N;
                ChunkDesc d;
                std::memset(&d, 0, sizeof(d));
                d.rowType = (i1 <= ns) ? 0 : (i0 >= ns ? 1 : 2);
                for (int s = 0; s < SF3D_SLOTS; ++s) {
                    bool any = false, all = true, same = true, sameDelta = true;
                    uint8_t k0 = LK_NONE; int64_t d0 = 0;
                    for (uint32_t i = i0; i < i1; ++i) {
                        const size_t e = (size_t)s * N + i;
                        if (kind[e] == LK_NONE) { all = false; continue; }
                        const int64_t dd = (int64_t)to[e] - (int64_t)i;
                        if (!any) { any = true; k0 = kind[e]; d0 = dd; }
                        else { if (kind[e] != k0 || dd != d0) same = false; if (dd != d0) sameDelta = false; }
                    }
                    const bool fits = d0 >= INT32_MIN && d0 <= INT32_MAX;
                    uint8_t ck = CK_NONE;
                    if (any) ck = (all && same && fits) ? k0 : (uint8_t)CK_MIXED;
                    d.kind[s] = ck;
                    d.ukind[s] = any ? ((same && fits) ? k0 : (uint8_t)CK_MIXED) : (uint8_t)CK_NONE;
                    d.delta[s] = (ck != CK_NONE && ck != CK_MIXED) ? (int32_t)d0 : 0;
                    if (ck == CK_MIXED && sameDelta && fits) { d.delta[s] = (int32_t)d0; d.sweepUniform |= (uint16_t)(1u << s); }
                    if (any) {                                   /* one interface area for the whole chunk? */
                        bool first = true, uni = true; double a0 = 0.;
                        for (uint32_t i = i0; i < i1 && uni; ++i) {
                            const size_t e = (size_t)s * N + i;
                            if (kind[e] == LK_NONE) continue;
                            if (first) { a0 = area[e]; first = false; }
                            else if (area[e] != a0) uni = false;
                        }
                        if (uni) { d.areaUniform |= (uint16_t)(1u << s); d.area[s] = a0; }
                        first = true; uni = true; double d0v = 0.;
                        for (uint32_t i = i0; i < i1 && uni; ++i) {
                            const size_t e = (size_t)s * N + i;
                            if (kind[e] == LK_NONE) continue;
                            if (first) { d0v = dist[e]; first = false; }
                            else if (dist[e] != d0v) uni = false;
                        }
                        if (uni) { d.distUniform |= (uint16_t)(1u << s); d.dist[s] = d0v; }
                    }
                }
                {   /* scalar-geometry path of k_assemble: full 64-node soil chunk, every slot empty or uniform soil-soil */
                    bool su = d.rowType == 1 && i1 - i0 == SF3D_CHUNK, partial = false;
                    for (int s = 0; s < SF3D_SLOTS && su; ++s) {
                        if (d.kind[s] == CK_NONE) continue;
                        if (d.kind[s] == CK_MIXED) partial = true;      /* same kind and offset wherever the link exists (ukind, sweepUniform), some nodes without it */
                        su = (d.ukind[s] == LK_SOIL_VERT || d.ukind[s] == LK_SOIL_LAT) && (d.kind[s] != CK_MIXED || ((d.sweepUniform >> s) & 1u))
                             && ((d.areaUniform >> s) & 1u) && ((d.distUniform >> s) & 1u);
                    }
                    d.soilUniform = su ? (partial ? 2 : 1) : 0;
                }
                cdesc[q] = d;
            }
        });

        /* paired sweep (k_sweep_pair): is the graph a regular NX x NY x NZ grid in layer-major numbering, and which neighbour
         * does every link slot of every node name?  (host logic; the same structure sf3d_get_regular_grid reports) */
        std::vector<uint64_t> pairNode, pairChunk;
        std::vector<int32_t> pairIdxMap;            /* non-empty: the graph is a layered masked grid, not a full box */
        uint32_t pairNX = 0, pairNY = 0, pairNZ = 0;
        {
            const char* pe = getenv("SF3D_PAIR_SWEEP");
            const int want = pe ? (pe[0] == '0' ? 0 : 1) : -1;          /* -1: automatic (large grids only, decided below) */
            bool regular = world_ == 1 && want != 0 && ns >= 64 && N % ns == 0 && N / ns >= 2 && N < (1u << 28);   /* 32-bit byte offsets in the kernel */
            int64_t maxOff = 0;
            if (regular) {
                for (int sl = 2; sl < SF3D_SLOTS && regular; ++sl)
                    for (uint32_t i = 0; i < N; ++i) {
                        const size_t e = (size_t)sl * N + i;
                        if (kind[e] == LK_NONE) continue;
                        const int64_t o = std::llabs((int64_t)to[e] - (int64_t)i);
                        if (o > maxOff) maxOff = o;
                    }
                const int64_t NX = maxOff - 1;
                regular = NX >= 64 && NX % 64 == 0 && ns % (uint64_t)NX == 0 && ns / (uint64_t)NX >= 6;
                if (regular) { pairNX = (uint32_t)NX; pairNY = (uint32_t)(ns / (uint64_t)NX); pairNZ = N / ns; }
            }
            if (regular) {
                pairNode.assign(N, 0);
                std::atomic<bool> irregular{false};
                const int64_t NX = pairNX, NY = pairNY;
                parallel_for(N, [&](uint32_t a, uint32_t b) {
                    for (uint32_t i = a; i < b && !irregular.load(std::memory_order_relaxed); ++i) {
                        const int64_t l = i / ns, r = (i % ns) / NX, cidx = i % NX;
                        uint64_t code = 0; unsigned seen = 0;
                        for (int sl = 0; sl < SF3D_SLOTS; ++sl) {
                            const size_t e = (size_t)sl * N + i;
                            uint64_t nib = SF3D_PAIR_NONE;
                            if (kind[e] != LK_NONE) {
                                const int64_t off = (int64_t)to[e] - (int64_t)i;
                                if (sl == 0) { if (off != -(int64_t)ns) irregular = true; nib = SF3D_PAIR_UP; }
                                else if (sl == 1) { if (off != (int64_t)ns) irregular = true; nib = SF3D_PAIR_DOWN; }
                                else {
                                    const int64_t rr = (off >= 0) ? (off + NX / 2) / NX : -((-off + NX / 2) / NX), cc = off - rr * NX;
                                    if (rr < -1 || rr > 1 || cc < -1 || cc > 1 || (rr == 0 && cc == 0) || r + rr < 0 || r + rr >= NY || cidx + cc < 0 || cidx + cc >= NX
                                        || (int64_t)(to[e] / ns) != l) { irregular = true; break; }
                                    nib = (uint64_t)((rr + 1) * 3 + (cc + 1));
                                    if (seen & (1u << nib)) irregular = true;      /* two links to one neighbour */
                                    seen |= 1u << nib;
                                }
                            }
                            code |= nib << (4 * sl);
                        }
                        pairNode[i] = code;
                    }
                });
                regular = !irregular;
            }
            if (regular) {
                pairChunk.assign(nChunks, 0);
                for (uint32_t q = 0; q < nChunks; ++q) {
                    bool same = (size_t)(q + 1) * SF3D_CHUNK <= N;
                    for (uint32_t k = 1; k < SF3D_CHUNK && same; ++k) same = pairNode[(size_t)q * SF3D_CHUNK + k] == pairNode[(size_t)q * SF3D_CHUNK];
                    if (same) pairChunk[q] = pairNode[(size_t)q * SF3D_CHUNK] | (1ull << 63);
                }
            } else { pairNode.clear(); pairNX = pairNY = pairNZ = 0; }
            /* not a full box: a LAYERED MASKED grid?  (a DEM outline with holes, soil columns of different depth - what
             * Project3D::setCrit3DTopography builds, project3D.cpp:941-1103.)  Every node gets (layer, row, column): layer = number of Up
             * hops to the surface, row / column from the surface node's coordinates; every link must go to one of the 26 grid neighbours. */
            if (!regular && world_ == 1 && want != 0 && ns >= 64 && N < (1u << 28) && m.x.size() == N && m.y.size() == N) {
                bool ok = true;
                std::vector<int32_t> lay(N, -1), row(N, -1), colv(N, -1);
                double cell = 0.;
                for (int sl = 2; sl < SF3D_SLOTS; ++sl)
                    for (uint32_t i = 0; i < ns; ++i) {
                        const size_t e = (size_t)sl * N + i;
                        if (kind[e] == LK_NONE) continue;
                        const double dx = std::fabs(m.x[to[e]] - m.x[i]), dy = std::fabs(m.y[to[e]] - m.y[i]);
                        const double d = dx > 0 ? (dy > 0 ? std::min(dx, dy) : dx) : dy;
                        if (d > 0 && (cell == 0. || d < cell)) cell = d;
                    }
                ok = cell > 0.;
                double xmin = 0, ymin = 0, ymax = 0;
                if (ok) {
                    xmin = m.x[0]; ymin = ymax = m.y[0];
                    for (uint32_t i = 1; i < ns; ++i) { xmin = std::min(xmin, m.x[i]); ymin = std::min(ymin, m.y[i]); ymax = std::max(ymax, m.y[i]); }
                    const bool northFirst = m.y[0] > m.y[ns - 1];
                    for (uint32_t i = 0; i < ns && ok; ++i) {
                        const double fc = (m.x[i] - xmin) / cell, fr = (northFirst ? (ymax - m.y[i]) : (m.y[i] - ymin)) / cell;
                        const long long c = std::llround(fc), r = std::llround(fr);
                        if (std::fabs(fc - (double)c) > 1e-6 || std::fabs(fr - (double)r) > 1e-6 || c > 60000 || r > 60000) ok = false;
                        lay[i] = 0; row[i] = (int32_t)r; colv[i] = (int32_t)c;
                    }
                }
                for (uint32_t i = ns; i < N && ok; ++i) {           /* soil nodes: below the node their Up link names (layer-major: it has a smaller index) */
                    if (kind[i] == LK_NONE || to[i] >= i || lay[to[i]] < 0) { ok = false; break; }
                    lay[i] = lay[to[i]] + 1; row[i] = row[to[i]]; colv[i] = colv[to[i]];
                }
                int32_t mr = 0, mc = 0, ml = 0;
                if (ok) for (uint32_t i = 0; i < N; ++i) { mr = std::max(mr, row[i]); mc = std::max(mc, colv[i]); ml = std::max(ml, lay[i]); }
                const uint32_t gNX = ok ? (uint32_t)((mc + 64) / 64 * 64) : 0, gNY = (uint32_t)mr + 1, gNZ = (uint32_t)ml + 1;
                if (ok && ((uint64_t)gNX * gNY * gNZ > (1ull << 30) || gNZ < 2 || gNY < 6)) ok = false;
                std::vector<int32_t> idxMap;
                if (ok) {
                    idxMap.assign((size_t)gNX * gNY * gNZ, -1);
                    for (uint32_t i = 0; i < N && ok; ++i) {
                        int32_t& cellIdx = idxMap[((size_t)lay[i] * gNY + (size_t)row[i]) * gNX + (size_t)colv[i]];
                        if (cellIdx >= 0) ok = false;                /* two nodes in one cell */
                        cellIdx = (int32_t)i;
                    }
                }
                if (ok) {
                    pairNode.assign(N, 0);
                    std::atomic<bool> bad{false};
                    parallel_for(N, [&](uint32_t a, uint32_t b) {
                        for (uint32_t i = a; i < b && !bad.load(std::memory_order_relaxed); ++i) {
                            uint64_t code = 0; unsigned seen = 0;
                            for (int sl = 0; sl < SF3D_SLOTS; ++sl) {
                                const size_t e = (size_t)sl * N + i;
                                uint64_t nib = SF3D_PAIR_NONE;
                                if (kind[e] != LK_NONE) {
                                    const uint32_t j = to[e];
                                    const int32_t dl = lay[j] - lay[i], dr = row[j] - row[i], dc = colv[j] - colv[i];
                                    if (sl == 0) { if (dl != -1 || dr != 0 || dc != 0) bad = true; nib = SF3D_PAIR_UP; }
                                    else if (sl == 1) { if (dl != 1 || dr != 0 || dc != 0) bad = true; nib = SF3D_PAIR_DOWN; }
                                    else {
                                        if (dl != 0 || dr < -1 || dr > 1 || dc < -1 || dc > 1 || (dr == 0 && dc == 0)) { bad = true; break; }
                                        nib = (uint64_t)((dr + 1) * 3 + (dc + 1));
                                        if (seen & (1u << nib)) bad = true;
                                        seen |= 1u << nib;
                                    }
                                }
                                code |= nib << (4 * sl);
                            }
                            pairNode[i] = code;
                        }
                    });
                    ok = !bad;
                }
                if (ok) { pairNX = gNX; pairNY = gNY; pairNZ = gNZ; pairIdxMap.swap(idxMap); pairChunk.assign(nChunks, 0); }
                else { pairNode.clear(); }
            }
        }

        /* ownership + chunk lists (identity lists on one GPU) */
        {
            if (m.presetPartition) I.part = *m.presetPartition;      /* strip-local model: owners and halo lists in local indices */
            else {
                sf3d_error_t pe = sf3d_compute_partition(m, rank_, world_, I.part);
                if (pe != SF3D_OK) { snprintf(err_, sizeof(err_), "partition failed"); return pe; }
            }
        }
        std::vector<uint32_t> listSurf, listSoil;
        for (uint32_t q = 0; q < nChunks; ++q) {
            bool mine = world_ == 1;
            if (!mine) {
                const uint32_t i0 = q * SF3D_CHUNK, i1 = (i0 + SF3D_CHUNK < N) ? i0 + SF3D_CHUNK : N;
                for (uint32_t i = i0; i < i1 && !mine; ++i) mine = I.part.owner[i] == rank_;
            }
            if (mine) (q < v.qSplit ? listSurf : listSoil).push_back(q);
        }
        {
            const uint32_t per = SF3D_BLOCK / SF3D_CHUNK;
            v.nListSurf = (uint32_t)listSurf.size();
            v.nList = (uint32_t)(listSurf.size() + listSoil.size());
            auto blocks = [&](uint32_t chunks) { uint32_t b = (chunks + per - 1) / per; if (b > SF3D_MAX_BLOCKS) b = SF3D_MAX_BLOCKS; return b; };
            v.nb = blocks(v.nList); if (v.nb == 0) v.nb = 1;
            v.nbSurf = blocks(v.nListSurf); if (v.nbSurf == 0) v.nbSurf = 1;
            v.nbSoil = blocks(v.nList - v.nListSurf);
            v.world = world_; v.rank = rank_;
        }
        if (world_ > 1) {
            /* chunks that owe values to a neighbouring rank go first: their halo puts are on the wire while the interior
             * is still being computed, instead of being the last thing every sweep waits for */
            std::vector<uint8_t> owes(nChunks, 0);
            for (int pr = 0; pr < world_; ++pr) for (uint32_t node : I.part.send[pr]) owes[node / SF3D_CHUNK] = 1;
            auto first = [&](uint32_t q) { return owes[q] != 0; };
            std::stable_partition(listSurf.begin(), listSurf.end(), first);
            std::stable_partition(listSoil.begin(), listSoil.end(), first);
        }
        listSurf.insert(listSurf.end(), listSoil.begin(), listSoil.end());
        {   /* grids of the two halves of k_assemble */
            const uint32_t per = SF3D_BLOCK / SF3D_CHUNK;
            auto blocks = [&](uint32_t chunks) { uint32_t b = (chunks + per - 1) / per; if (b > SF3D_MAX_BLOCKS) b = SF3D_MAX_BLOCKS; return b; };
            v.nbSoil = blocks(v.nList - v.nListSurf);
            if (const char* be = getenv("SF3D_ASM_SOIL_BLOCKS")) { const uint32_t nb = (uint32_t)atoi(be); if (nb > 0 && nb < v.nbSoil) v.nbSoil = nb; }   /* tuning */
        }
        double *z, *size, *pond, *sink, *bslope, *bsize, *prescribed, *roughness, *larea, *ldist;
        uint16_t* cls; uint8_t *btype, *lkind; uint32_t* lto; SoilDev* soils; ChunkDesc* dcdesc;
        HIP_TRY(dev_alloc(I.allocs, z, N)); HIP_TRY(dev_alloc(I.allocs, size, N));
        HIP_TRY(dev_alloc(I.allocs, pond, N)); HIP_TRY(dev_alloc(I.allocs, sink, N));
        HIP_TRY(dev_alloc(I.allocs, cls, N)); HIP_TRY(dev_alloc(I.allocs, btype, N));
        HIP_TRY(dev_alloc(I.allocs, bslope, N)); HIP_TRY(dev_alloc(I.allocs, bsize, N));
        HIP_TRY(dev_alloc(I.allocs, prescribed, N));
        HIP_TRY(dev_alloc(I.allocs, lto, NS)); HIP_TRY(dev_alloc(I.allocs, lkind, NS));
        HIP_TRY(dev_alloc(I.allocs, larea, NS)); HIP_TRY(dev_alloc(I.allocs, ldist, NS));
        HIP_TRY(dev_alloc(I.allocs, v.lflowSum, NS)); HIP_TRY(dev_alloc(I.allocs, v.A2x[0], NS / 2)); HIP_TRY(dev_alloc(I.allocs, v.A2x[1], NS / 2));
        HIP_TRY(dev_alloc(I.allocs, v.b, N)); HIP_TRY(dev_alloc(I.allocs, v.C, N));
        for (int k = 0; k < SF3D_POOL; ++k) HIP_TRY(dev_alloc(I.allocs, v.X[k], N));
        HIP_TRY(dev_alloc(I.allocs, v.Se, N)); HIP_TRY(dev_alloc(I.allocs, v.K, N)); HIP_TRY(dev_alloc(I.allocs, v.SeHold, N));
        HIP_TRY(dev_alloc(I.allocs, dcdesc, (size_t)nChunks));
        uint32_t* dlist; HIP_TRY(dev_alloc(I.allocs, dlist, listSurf.size()));
        if (!listSurf.empty()) HIP_TRY(hipMemcpy(dlist, listSurf.data(), listSurf.size() * 4, hipMemcpyHostToDevice));
        v.chunkList = dlist;
        v.asmList = dlist;              /* the assembly walks the chunks in list order */
        v.owner = nullptr; v.dist = nullptr;
        {   /* bytes one rank touches per sweep: 152 B per owned node.  Below the 256 MiB Infinity Cache the whole
             * sweep working set stays cached between sweeps and bypassing costs ~8 % (measured at 0.98 M nodes);
             * above it the coefficient stream would thrash x, b, z: bypass gains ~13 % at 5.2 M nodes */
            const char* e = getenv("SF3D_NT_STREAM");
            const double sweepBytes = 152.0 * (double)v.nList * SF3D_CHUNK;
            v.ntStream = e ? (e[0] != '0') : (sweepBytes > 256.0 * 1024 * 1024);
        }
        I.pairBlocks = 0; I.pairMasked = false; v.pair.masked = 0;
        if (!pairNode.empty()) {
            /* the paired sweep pays where the sweep streams from HBM (ntStream: above the Infinity Cache) - below that the plain sweep
             * is cache-resident and faster; SF3D_PAIR_SWEEP=1 forces it on any regular grid (tests on small grids) */
            const char* pe = getenv("SF3D_PAIR_SWEEP");
            /* (masked grids, k_sweep_pair_masked: measured at the Ravone project, 5.85 M nodes, 249 us per pair against 2 x 147 us of
             * k_sweep - profiles/README.md; a first version that read the link table was slower than two sweeps) */
            const bool on = pe ? (pe[0] != '0') : (v.ntStream != 0);
            if (on) {
                int cus = 256; { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, I.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount; }
                uint32_t bestW = 0; double bestCost = 1e30;
                const char* we = getenv("SF3D_PAIR_W");
                /* masked grids: the non-empty patches (a patch owns rows pr (W - 2) .. + W - 3 and 64 columns) and the layers they reach */
                std::vector<uint32_t> plist[3]; std::vector<uint8_t> pdepth[3];
                if (!pairIdxMap.empty()) {
                    std::vector<uint8_t> colDepth((size_t)pairNX * pairNY, 0);          /* deepest layer + 1 of every cell's column */
                    for (uint32_t l = 0; l < pairNZ; ++l)
                        for (size_t k = 0; k < colDepth.size(); ++k)
                            if (pairIdxMap[(size_t)l * colDepth.size() + k] >= 0) colDepth[k] = (uint8_t)(l + 1);
                    int wi = 0;
                    for (uint32_t W : {6u, 10u, 14u}) {
                        const uint32_t prs = (pairNY + W - 3) / (W - 2), pcs = pairNX / 64;
                        for (uint32_t pr = 0; pr < prs; ++pr)
                            for (uint32_t pc = 0; pc < pcs; ++pc) {
                                const int64_t r0 = (int64_t)pr * (W - 2) - 1, c0 = (int64_t)pc * 64 - 1;
                                bool any = false; uint8_t depth = 0;
                                for (int64_t r = r0; r < r0 + (int64_t)W; ++r) {
                                    if (r < 0 || r >= (int64_t)pairNY) continue;
                                    for (int64_t cc = c0; cc < c0 + 66; ++cc) {
                                        if (cc < 0 || cc >= (int64_t)pairNX) continue;
                                        const uint8_t d = colDepth[(size_t)r * pairNX + (size_t)cc];
                                        if (d > depth) depth = d;
                                        if (d && r > r0 && r < r0 + (int64_t)W - 1 && cc > c0 && cc < c0 + 65) any = true;      /* an owned cell */
                                    }
                                }
                                if (any) { plist[wi].push_back((pr << 12) | pc); pdepth[wi].push_back(depth); }
                            }
                        ++wi;
                    }
                }
                for (uint32_t W : {6u, 10u, 14u}) {
                    if (pairNY < W || (we && (uint32_t)atoi(we) != W)) continue;
                    const uint64_t blocks = pairIdxMap.empty() ? (uint64_t)((pairNY + W - 3) / (W - 2)) * (pairNX / 64) : (uint64_t)plist[W == 6 ? 0 : (W == 10 ? 1 : 2)].size();
                    if (blocks == 0) continue;
                    const uint64_t resident = (uint64_t)(4 * SF3D_PAIR_WAVES / (W + 1)) * cus;     /* blocks of W + 1 waves per CU */
                    const uint64_t rounds = (blocks + resident - 1) / resident;
                    const double cost = (double)(rounds * resident) / (double)blocks * (double)W / (double)(W - 2);
                    if (cost < bestCost) { bestCost = cost; bestW = W; }
                }
                if (bestW) {
                    uint64_t *dn, *dq;
                    HIP_TRY(dev_alloc(I.allocs, dn, pairNode.size())); HIP_TRY(dev_alloc(I.allocs, dq, pairChunk.size()));
                    HIP_TRY(hipMemcpy(dn, pairNode.data(), pairNode.size() * 8, hipMemcpyHostToDevice));
                    HIP_TRY(hipMemcpy(dq, pairChunk.data(), pairChunk.size() * 8, hipMemcpyHostToDevice));
                    v.pair.NX = pairNX; v.pair.NY = pairNY; v.pair.NZ = pairNZ; v.pair.W = bestW;
                    v.pair.patchCols = pairNX / 64; v.pair.patchRows = (pairNY + bestW - 3) / (bestW - 2);
                    v.pair.nodeCode = dn; v.pair.chunkCode = dq;
                    I.pairBlocks = v.pair.patchCols * v.pair.patchRows;
                    v.pair.masked = 0; I.pairMasked = false;
                    if (!pairIdxMap.empty()) {
                        const int wi = bestW == 6 ? 0 : (bestW == 10 ? 1 : 2);
                        int32_t* dm; uint32_t* dl; uint8_t* dd;
                        HIP_TRY(dev_alloc(I.allocs, dm, pairIdxMap.size())); HIP_TRY(dev_alloc(I.allocs, dl, plist[wi].size())); HIP_TRY(dev_alloc(I.allocs, dd, pdepth[wi].size()));
                        HIP_TRY(hipMemcpy(dm, pairIdxMap.data(), pairIdxMap.size() * 4, hipMemcpyHostToDevice));
                        HIP_TRY(hipMemcpy(dl, plist[wi].data(), plist[wi].size() * 4, hipMemcpyHostToDevice));
                        HIP_TRY(hipMemcpy(dd, pdepth[wi].data(), pdepth[wi].size(), hipMemcpyHostToDevice));
                        v.pair.masked = 1; v.pair.idxMap = dm; v.pair.patchList = dl; v.pair.patchDepth = dd;
                        I.pairBlocks = (uint32_t)plist[wi].size(); I.pairMasked = true;
                    }
                }
            }
        }
        if (world_ > 1) {
            uint8_t* downer; HIP_TRY(dev_alloc(I.allocs, downer, N));
            HIP_TRY(hipMemcpy(downer, I.part.owner.data(), N, hipMemcpyHostToDevice));
            v.owner = downer;
            /* my window: mailboxes + payload areas for what each neighbour puts (2 parities x 2 fields) */
            DistView& d = I.hostDist;
            std::memset(&d, 0, sizeof(d));
            d.world = world_; d.rank = rank_; d.owner = downer;
            uint64_t off = 0;
            uint32_t maxSend = 0;
            for (int pr = 0; pr < world_; ++pr) {
                d.recvCount[pr] = (uint32_t)I.part.recv[pr].size();
                d.recvOff[pr] = off;
                off += (uint64_t)d.recvCount[pr] * 2 * SF3D_DIST_FIELDS;
                d.sendCount[pr] = (uint32_t)I.part.send[pr].size();
                if (d.sendCount[pr] > maxSend) maxSend = d.sendCount[pr];
                uint32_t *si, *ri;
                HIP_TRY(dev_alloc(I.allocs, si, I.part.send[pr].size())); HIP_TRY(dev_alloc(I.allocs, ri, I.part.recv[pr].size()));
                if (d.sendCount[pr]) HIP_TRY(hipMemcpy(si, I.part.send[pr].data(), (size_t)d.sendCount[pr] * 4, hipMemcpyHostToDevice));
                if (d.recvCount[pr]) HIP_TRY(hipMemcpy(ri, I.part.recv[pr].data(), (size_t)d.recvCount[pr] * 4, hipMemcpyHostToDevice));
                d.sendIdx[pr] = si; d.recvIdx[pr] = ri;
            }
            {   /* the send lists regrouped by chunk for the in-kernel puts */
                struct Ent { uint32_t node; uint8_t peer; uint32_t slot; };
                std::vector<Ent> ents;
                for (int pr = 0; pr < world_; ++pr)
                    for (uint32_t k = 0; k < I.part.send[pr].size(); ++k) ents.push_back({I.part.send[pr][k], (uint8_t)pr, k});
                std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.node != b.node ? a.node < b.node : a.peer < b.peer; });
                std::vector<uint32_t> bstart(nChunks + 1, 0), bslot(ents.size());
                std::vector<uint8_t> blane(ents.size()), bpeer(ents.size());
                for (size_t k = 0; k < ents.size(); ++k) {
                    bstart[ents[k].node / SF3D_CHUNK + 1]++;
                    blane[k] = (uint8_t)(ents[k].node % SF3D_CHUNK); bpeer[k] = ents[k].peer; bslot[k] = ents[k].slot;
                }
                for (uint32_t q = 0; q < nChunks; ++q) bstart[q + 1] += bstart[q];
                uint32_t *dbs, *dsl; uint8_t *dla, *dpe;
                HIP_TRY(dev_alloc(I.allocs, dbs, bstart.size())); HIP_TRY(dev_alloc(I.allocs, dsl, bslot.size()));
                HIP_TRY(dev_alloc(I.allocs, dla, blane.size())); HIP_TRY(dev_alloc(I.allocs, dpe, bpeer.size()));
                HIP_TRY(hipMemcpy(dbs, bstart.data(), bstart.size() * 4, hipMemcpyHostToDevice));
                if (!ents.empty()) {
                    HIP_TRY(hipMemcpy(dsl, bslot.data(), bslot.size() * 4, hipMemcpyHostToDevice));
                    HIP_TRY(hipMemcpy(dla, blane.data(), blane.size(), hipMemcpyHostToDevice));
                    HIP_TRY(hipMemcpy(dpe, bpeer.data(), bpeer.size(), hipMemcpyHostToDevice));
                }
                d.bndStart = dbs; d.bndSlot = dsl; d.bndLane = dla; d.bndPeer = dpe;
            }
            if (I.useFused < 0) { const char* fe = getenv("SF3D_FUSED_DECIDE"); I.useFused = (fe && fe[0] == '0') ? 0 : 1; }
            v.haloDirect = I.useFused != 0 ? 1u : 0u;      /* the fused exchange only; SF3D_HALO_DIRECT=0 copies the halo after every sweep */
            if (const char* hd = getenv("SF3D_HALO_DIRECT")) v.haloDirect = (hd[0] == '1' && I.useFused != 0) ? 1u : 0u;
            if (v.haloDirect) {   /* where each foreign neighbour's value arrives in my window; chunks that have any are flagged */
                std::vector<uint32_t> fsrc(NS, SF3D_FSRC_NONE);
                for (uint32_t i = 0; i < N; ++i) {
                    if (I.part.owner[i] != rank_) continue;
                    for (int sl = 0; sl < SF3D_SLOTS; ++sl) {
                        const size_t e = (size_t)sl * N + i;
                        if (kind[e] == LK_NONE) continue;
                        const uint32_t j = to[e];
                        const int pr = I.part.owner[j];
                        if (pr == rank_) continue;
                        const auto& lst = I.part.recv[pr];
                        const auto it = std::lower_bound(lst.begin(), lst.end(), j);
                        if (it == lst.end() || *it != j) { snprintf(err_, sizeof(err_), "partition: node %u read by %u is missing from the halo list of rank %d", j, i, pr); return SF3D_TOPOGRAPHY_ERROR; }
                        fsrc[e] = ((uint32_t)pr << 27) | (uint32_t)(it - lst.begin());
                        cdesc[i / SF3D_CHUNK].pad0 = 1;
                    }
                }
                uint32_t* dfs; HIP_TRY(dev_alloc(I.allocs, dfs, NS));
                HIP_TRY(hipMemcpy(dfs, fsrc.data(), NS * 4, hipMemcpyHostToDevice));
                d.fsrc = dfs;
            }
            I.windowBytes = sizeof(DistWindow) + off * sizeof(double);
            void* w = nullptr;
            HIP_TRY(hipExtMallocWithFlags(&w, I.windowBytes, hipDeviceMallocFinegrained));
            HIP_TRY(hipMemset(w, 0, I.windowBytes));
            I.window = static_cast<DistWindow*>(w);
            d.win[rank_] = I.window;
            d.payload[rank_] = reinterpret_cast<double*>(reinterpret_cast<char*>(w) + sizeof(DistWindow));
            HIP_TRY(dev_alloc(I.allocs, I.devDist, 1));
            I.pushBlocks = (maxSend + SF3D_BLOCK - 1) / SF3D_BLOCK;
            if (I.pushBlocks == 0) I.pushBlocks = 1;
            if (I.pushBlocks > 256) I.pushBlocks = 256;
        }
        if (m.compat) {      /* quirk-1 emulation: calloc'd like the reference's rows (soilFluxes3D.cpp:76-155) */
            HIP_TRY(dev_alloc(I.allocs, v.compatCv, (size_t)N * (SF3D_SLOTS + 2))); HIP_TRY(hipMemset(v.compatCv, 0, (size_t)N * (SF3D_SLOTS + 2) * 8));
            HIP_TRY(dev_alloc(I.allocs, v.compatCn, N)); HIP_TRY(hipMemset(v.compatCn, 0, N));
            HIP_TRY(dev_alloc(I.allocs, v.compatDiag, N)); HIP_TRY(hipMemset(v.compatDiag, 0, (size_t)N * 8));
        }
        if (m.cgArrays && world_ == 1) {      /* device conjugate gradients (SF3D_LINEAL_DEVICE_CG=1) */
            HIP_TRY(dev_alloc(I.allocs, v.cgDiag, N)); HIP_TRY(dev_alloc(I.allocs, v.cgR, N)); HIP_TRY(dev_alloc(I.allocs, v.cgP, N)); HIP_TRY(dev_alloc(I.allocs, v.cgQ, N));
            HIP_TRY(hipMemset(v.cgDiag, 0, (size_t)N * 8)); HIP_TRY(hipMemset(v.cgR, 0, (size_t)N * 8)); HIP_TRY(hipMemset(v.cgP, 0, (size_t)N * 8)); HIP_TRY(hipMemset(v.cgQ, 0, (size_t)N * 8));
        }
        HIP_TRY(dev_alloc(I.allocs, v.flow, N)); HIP_TRY(dev_alloc(I.allocs, v.bflowRate, N));
        HIP_TRY(dev_alloc(I.allocs, v.bflowSum, N));
        {   const size_t np = std::max<size_t>(v.nb + v.nbSurf, I.pairBlocks) + 8;
            HIP_TRY(dev_alloc(I.allocs, v.part0, np)); HIP_TRY(dev_alloc(I.allocs, v.part1, np)); }
        HIP_TRY(dev_alloc(I.allocs, v.arrive, 16 * 17)); HIP_TRY(hipMemset(v.arrive, 0, 16 * 17 * sizeof(unsigned int)));
        HIP_TRY(dev_alloc(I.allocs, v.gridBar, 16)); HIP_TRY(hipMemset(v.gridBar, 0, 16 * sizeof(unsigned int)));
        HIP_TRY(dev_alloc(I.allocs, soils, m.soils.size())); HIP_TRY(dev_alloc(I.allocs, roughness, m.roughness.size()));
        HIP_TRY(dev_alloc(I.allocs, v.ctrl, 1));
        v.z = z; v.size = size; v.pond = pond; v.sink = sink; v.cls = cls; v.btype = btype;
        v.bslope = bslope; v.bsize = bsize; v.prescribed = prescribed;
        v.lto = lto; v.lkind = lkind; v.larea = larea; v.ldist = ldist; v.soils = soils; v.roughness = roughness;
        v.cdesc = dcdesc;

        I.heatAirP = nullptr;
        if (m.heat) {       /* coupled heat transport: state, system, per-node conductivities, boundary and link flux arrays */
            HeatDev& hv = v.heat;
            hv = HeatDev{};
            hv.on = 1; hv.water = m.water ? 1u : 0u; hv.vapor = m.heatVapor ? 1u : 0u; hv.advection = m.heatAdvection ? 1u : 0u;
            hv.save = m.heatSave; hv.wf = p.heatWeightFactor;
            double* tmp = nullptr;
            for (int k = 0; k < 3; ++k) { HIP_TRY(dev_alloc(I.allocs, hv.TX[k], N)); HIP_TRY(hipMemset(hv.TX[k], 0, N * 8)); }
            auto alloc0 = [&](double*& ptr, size_t cnt) -> hipError_t { hipError_t e = dev_alloc(I.allocs, ptr, cnt); if (e == hipSuccess) e = hipMemset(ptr, 0, cnt * 8); return e; };
            HIP_TRY(alloc0(tmp, N)); hv.heatSink = tmp;
            HIP_TRY(alloc0(hv.heatFlux, N)); HIP_TRY(alloc0(hv.invariant, N));
            HIP_TRY(alloc0(hv.hC, N)); HIP_TRY(alloc0(hv.hcapTerm, N)); HIP_TRY(alloc0(hv.hb, N)); HIP_TRY(alloc0(hv.hD, N));
            HIP_TRY(dev_alloc(I.allocs, hv.hA2, NS / 2)); HIP_TRY(hipMemset(hv.hA2, 0, NS * 8));
            HIP_TRY(alloc0(hv.kHeat, N)); HIP_TRY(alloc0(hv.kIsoVap, N)); HIP_TRY(alloc0(hv.hAvg, N));
            HIP_TRY(alloc0(hv.thetaOld, N));
            HIP_TRY(alloc0(tmp, N)); hv.airP = tmp; I.heatAirP = tmp;
            HIP_TRY(alloc0(hv.wThLiq, N)); HIP_TRY(alloc0(hv.wThVap, N)); HIP_TRY(alloc0(hv.wTm, N));
            const double** inputs[9] = {&hv.bHeightWind, &hv.bHeightT, &hv.bRoughH, &hv.bT, &hv.bRH, &hv.bWind, &hv.bNetIrr, &hv.bFixT, &hv.bFixDepth};
            for (auto* pp : inputs) { HIP_TRY(alloc0(tmp, N)); *pp = tmp; }
            double** outputs[6] = {&hv.bAero, &hv.bSoilCond, &hv.bSens, &hv.bLat, &hv.bRad, &hv.bAdv};
            const std::vector<double>* outInit[6] = {&m.bAero, &m.bSoilCond, &m.bSens, &m.bLat, &m.bRad, &m.bAdv};   /* NODATA / 0 of setNodeBoundary */
            for (int k = 0; k < 6; ++k) {
                HIP_TRY(alloc0(*outputs[k], N)); I.heatOut[k] = *outputs[k];
                HIP_TRY(hipMemcpy(*outputs[k], outInit[k]->data(), N * 8, hipMemcpyHostToDevice));
            }
            HIP_TRY(alloc0(hv.lwaterFlux, NS)); HIP_TRY(alloc0(hv.lvaporFlux, NS));
            const int nTypes = (m.heatSave == 2) ? SF3D_FLUX_TYPES : (m.heatSave == 1 ? 1 : 0);
            {   /* never-linked slots hold 0 (calloc in the reference), linked slots NODATA (setNodeLink, soilFluxes3D.cpp:672-678) */
                std::vector<double> init(NS, 0.);
                for (size_t e = 0; e < NS; ++e) if (kind[e] != LK_NONE) init[e] = NODATA_D;
                for (int t = 0; t < nTypes; ++t) {
                    HIP_TRY(dev_alloc(I.allocs, hv.lflux[t], NS));
                    HIP_TRY(hipMemcpy(hv.lflux[t], init.data(), NS * 8, hipMemcpyHostToDevice));
                }
            }
            {   /* nodeDistance3D (soilPhysics.cpp:334-338) of every link between two soil nodes */
                std::vector<double> d3(NS, 1.);
                parallel_for(N, [&](uint32_t a, uint32_t b) {
                    for (uint32_t i = a; i < b; ++i)
                        for (int s = 0; s < SF3D_SLOTS; ++s) {
                            const size_t e = (size_t)s * N + i;
                            if (kind[e] != LK_SOIL_VERT && kind[e] != LK_SOIL_LAT) continue;
                            const uint32_t j = to[e];
                            const double dx = m.x[i] - m.x[j], dy = m.y[i] - m.y[j], dz = m.z[i] - m.z[j];
                            double nrm = 0; nrm += dx * dx; nrm += dy * dy; nrm += dz * dz;
                            d3[e] = std::sqrt(nrm);
                        }
                });
                HIP_TRY(alloc0(tmp, NS)); hv.hdist = tmp;
                HIP_TRY(hipMemcpy(tmp, d3.data(), NS * 8, hipMemcpyHostToDevice));
            }
            I.gsLevelStart.clear();
            if (const char* ge = getenv("SF3D_HEAT_GS")) if (ge[0] == '1') {
                if (world_ > 1) { snprintf(err_, sizeof(err_), "SF3D_HEAT_GS=1 (reference-order Gauss-Seidel) is a single-GPU verification mode"); return SF3D_PARAMETER_ERROR; }
                /* dependency levels of the serial sweep: a node waits for its linked heat nodes with a smaller index */
                std::vector<uint32_t> level(N, 0u);
                uint32_t maxLevel = 0;
                for (uint32_t i = ns; i < N; ++i) {
                    uint32_t l = 0;
                    for (int sl = 0; sl < SF3D_SLOTS; ++sl) {
                        const size_t e = (size_t)sl * N + i;
                        if ((kind[e] == LK_SOIL_VERT || kind[e] == LK_SOIL_LAT) && to[e] < i && level[to[e]] + 1 > l) l = level[to[e]] + 1;
                    }
                    level[i] = l;
                    if (l > maxLevel) maxLevel = l;
                }
                std::vector<uint32_t> start(maxLevel + 2, 0u), order(N > ns ? N - ns : 0);
                for (uint32_t i = ns; i < N; ++i) start[level[i] + 1]++;
                for (uint32_t l = 0; l <= maxLevel; ++l) start[l + 1] += start[l];
                std::vector<uint32_t> pos(start.begin(), start.end() - 1);
                for (uint32_t i = ns; i < N; ++i) order[pos[level[i]]++] = i;
                uint32_t* dord; HIP_TRY(dev_alloc(I.allocs, dord, order.size()));
                if (!order.empty()) HIP_TRY(hipMemcpy(dord, order.data(), order.size() * 4, hipMemcpyHostToDevice));
                hv.gsOrder = dord; hv.gs = 1;
                I.gsLevelStart = start;
            }
            m.heatStateDirty = m.heatSinkDirty = m.heatBoundaryDirty = true;
            m.hostStaleHeat = false;
            for (int t = 0; t < SF3D_FLUX_TYPES; ++t) m.lfluxValid[t] = false;
        }

        std::vector<SoilDev> sd(m.soils.size());
        for (size_t k = 0; k < sd.size(); ++k) {
            const SoilHost& s = m.soils[k];
            sd[k] = SoilDev{s.alpha, s.n, s.m, s.he, s.Sc, 1.0 / s.Sc, s.thetaS, s.thetaR, s.Ksat, s.L, 1.0 / s.m, s.mualemDen, s.clay, s.organicMatter};
        }
        HIP_TRY(hipMemcpy(z, m.z.data(), N * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(size, m.size.data(), N * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(cls, m.cls.data(), N * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(lto, to.data(), NS * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(lkind, kind.data(), NS, hipMemcpyHostToDevice));
        {   std::vector<uint16_t> mask(N, 0);
            for (int sl = 0; sl < SF3D_SLOTS; ++sl) { const uint8_t* kk = kind.data() + (size_t)sl * N; for (uint32_t i = 0; i < N; ++i) if (kk[i] != LK_NONE) mask[i] |= (uint16_t)(1u << sl); }
            uint16_t* dm; HIP_TRY(dev_alloc(I.allocs, dm, N));
            HIP_TRY(hipMemcpy(dm, mask.data(), (size_t)N * 2, hipMemcpyHostToDevice));
            v.lmask = dm; }
        HIP_TRY(hipMemcpy(larea, area.data(), NS * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ldist, dist.data(), NS * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dcdesc, cdesc.data(), cdesc.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
        if (!sd.empty()) HIP_TRY(hipMemcpy(soils, sd.data(), sd.size() * sizeof(SoilDev), hipMemcpyHostToDevice));
        if (!m.roughness.empty()) HIP_TRY(hipMemcpy(roughness, m.roughness.data(), m.roughness.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(v.A2x[0], 0, NS * 8)); HIP_TRY(hipMemset(v.A2x[1], 0, NS * 8)); HIP_TRY(hipMemset(v.b, 0, N * 8)); HIP_TRY(hipMemset(v.C, 0, N * 8));
        HIP_TRY(hipMemset(v.flow, 0, N * 8)); HIP_TRY(hipMemset(v.bflowRate, 0, N * 8)); HIP_TRY(hipMemset(v.SeHold, 0, N * 8));
        for (int k = 0; k < SF3D_POOL; ++k) HIP_TRY(hipMemset(v.X[k], 0, N * 8));
        if (m.heat && I.heatAirP) hipLaunchKernelGGL(k_heat_static, dim3(v.nb), dim3(SF3D_BLOCK), 0, 0, v, I.heatAirP);
        HIP_TRY(hipDeviceSynchronize());      /* null-stream fills must land before the (non-blocking) solver stream runs */

        std::memset(&mirror_, 0, sizeof(mirror_));
        mirror_.cur = 0; mirror_.hold = 0; mirror_.best = -1; mirror_.stage = ST_IDLE;
        mirror_.bestMBR = NODATA_D;
        mirror_.tCur = 0; mirror_.tOld = 0; mirror_.hStage = HS_IDLE;
        built_ = true;
        m.graphDirty = false;
        m.stateDirty = m.sinkDirty = m.pondDirty = m.boundaryDirty = m.flowSumsDirty = m.ctrlDirty = true;
        m.sinkLo = 0; m.sinkHi = UINT32_MAX;
        m.hostStaleState = m.hostStaleFlows = false;
    }

    DevView& v = I.v;
    if (m.stateDirty) {
        HIP_TRY(hipMemcpyAsync(v.X[mirror_.cur], m.H.data(), N * 8, hipMemcpyHostToDevice, I.stream));
        HIP_TRY(hipMemcpyAsync(v.Se, m.Se.data(), N * 8, hipMemcpyHostToDevice, I.stream));
        HIP_TRY(hipMemcpyAsync(v.K, m.K.data(), N * 8, hipMemcpyHostToDevice, I.stream));
        mirror_.seSource = 0; m.ctrlDirty = true;      /* Se now comes from the host's libm: the next attempt recomputes it on the device */
        m.stateDirty = false;
    }
    if (m.sinkDirty) {
        const uint32_t lo = m.sinkLo < N ? m.sinkLo : 0u, hi = m.sinkHi > N ? N : m.sinkHi;      /* only the range the setters touched */
        if (hi > lo) HIP_TRY(hipMemcpyAsync((void*)(v.sink + lo), m.sink.data() + lo, (size_t)(hi - lo) * 8, hipMemcpyHostToDevice, I.stream));
        m.sinkDirty = false; m.sinkLo = 0; m.sinkHi = 0;
    }
    if (m.pondDirty) { HIP_TRY(hipMemcpyAsync((void*)v.pond, m.pond.data(), N * 8, hipMemcpyHostToDevice, I.stream)); m.pondDirty = false; }
    if (m.boundaryDirty) {
        HIP_TRY(hipMemcpyAsync((void*)v.btype, m.btype.data(), N, hipMemcpyHostToDevice, I.stream));
        HIP_TRY(hipMemcpyAsync((void*)v.bslope, m.bslope.data(), N * 8, hipMemcpyHostToDevice, I.stream));
        HIP_TRY(hipMemcpyAsync((void*)v.bsize, m.bsize.data(), N * 8, hipMemcpyHostToDevice, I.stream));
        HIP_TRY(hipMemcpyAsync((void*)v.prescribed, m.prescribed.data(), N * 8, hipMemcpyHostToDevice, I.stream));
        m.boundaryDirty = false;
    }
    if (m.flowSumsDirty) {
        HIP_TRY(hipMemcpyAsync(v.bflowSum, m.bflowSum.data(), N * 8, hipMemcpyHostToDevice, I.stream));
        if (I.dslot.empty()) {
            for (int s = 0; s < SF3D_SLOTS; ++s)
                HIP_TRY(hipMemcpyAsync(v.lflowSum + (size_t)s * N, m.lflowSum[s].data(), N * 8, hipMemcpyHostToDevice, I.stream));
        } else {                                   /* insertion order -> device slots */
            I.stage.resize(NS);
            for (int s = 0; s < SF3D_SLOTS; ++s) {
                const uint8_t* d = I.dslot.data() + (size_t)s * N; const double* src = m.lflowSum[s].data();
                for (uint32_t i = 0; i < N; ++i) I.stage[(size_t)d[i] * N + i] = src[i];
            }
            HIP_TRY(hipMemcpyAsync(v.lflowSum, I.stage.data(), NS * 8, hipMemcpyHostToDevice, I.stream));
            HIP_TRY(hipStreamSynchronize(I.stream));
        }
        m.flowSumsDirty = false;
    }
    if (v.heat.on) {
        HeatDev& hv = v.heat;
        if (m.heatStateDirty) {      /* setNodeTemperature sets temperature and oldTemperature (soilFluxes3D.cpp:1308-1309) */
            if (mirror_.tOld != mirror_.tCur) { mirror_.tOld = mirror_.tCur; m.ctrlDirty = true; }
            HIP_TRY(hipMemcpyAsync(hv.TX[mirror_.tCur], m.temperature.data(), N * 8, hipMemcpyHostToDevice, I.stream));
            m.heatStateDirty = false;
        }
        if (m.heatSinkDirty) { HIP_TRY(hipMemcpyAsync((void*)hv.heatSink, m.heatSink.data(), N * 8, hipMemcpyHostToDevice, I.stream)); m.heatSinkDirty = false; }
        if (m.heatBoundaryDirty) {
            const std::vector<double>* src[9] = {&m.bHeightWind, &m.bHeightT, &m.bRoughH, &m.bT, &m.bRH, &m.bWind, &m.bNetIrr, &m.bFixT, &m.bFixDepth};
            const double* dst[9] = {hv.bHeightWind, hv.bHeightT, hv.bRoughH, hv.bT, hv.bRH, hv.bWind, hv.bNetIrr, hv.bFixT, hv.bFixDepth};
            for (int k = 0; k < 9; ++k) HIP_TRY(hipMemcpyAsync((void*)dst[k], src[k]->data(), N * 8, hipMemcpyHostToDevice, I.stream));
            m.heatBoundaryDirty = false;
        }
    }
    if (m.ctrlDirty || ctrlEdited_) {
        if (mirror_.wrc != p.wrc) mirror_.seSource = 0;      /* another retention curve: the stored Se belongs to the old one */
        fill_params(mirror_, p);
        HIP_TRY(hipMemcpyAsync(v.ctrl, &mirror_, sizeof(Ctrl), hipMemcpyHostToDevice, I.stream));
        m.ctrlDirty = false; ctrlEdited_ = false;
    }
    /* pageable-source async copies return once staged, but be explicit before host buffers change */
    HIP_TRY(hipStreamSynchronize(I.stream));
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::fetch_state(HostModel& m)
{
    Impl& I = *impl_;
    HIP_TRY(hipMemcpyAsync(m.H.data(), I.v.X[mirror_.cur], (size_t)m.N * 8, hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipMemcpyAsync(m.Se.data(), I.v.Se, (size_t)m.N * 8, hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipMemcpyAsync(m.K.data(), I.v.K, (size_t)m.N * 8, hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipStreamSynchronize(I.stream));
    /* the API reports Se = 1 and K = NODATA / 0 for surface nodes as the setters left them */
    m.hostStaleState = false;
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::fetch_flows(HostModel& m)
{
    Impl& I = *impl_;
    const size_t N = m.N;
    if (I.stream2) { HIP_TRY(hipStreamSynchronize(I.stream2)); I.linksPending[0] = I.linksPending[1] = false; }     /* link flow sums of the last step */
    HIP_TRY(hipMemcpyAsync(m.bflowSum.data(), I.v.bflowSum, N * 8, hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipMemcpyAsync(m.bflowRate.data(), I.v.bflowRate, N * 8, hipMemcpyDeviceToHost, I.stream));
    if (I.dslot.empty()) {
        for (int s = 0; s < SF3D_SLOTS; ++s)
            HIP_TRY(hipMemcpyAsync(m.lflowSum[s].data(), I.v.lflowSum + (size_t)s * N, N * 8, hipMemcpyDeviceToHost, I.stream));
        HIP_TRY(hipStreamSynchronize(I.stream));
    } else {                                       /* device slots -> insertion order */
        I.stage.resize(N * SF3D_SLOTS);
        HIP_TRY(hipMemcpyAsync(I.stage.data(), I.v.lflowSum, N * SF3D_SLOTS * 8, hipMemcpyDeviceToHost, I.stream));
        HIP_TRY(hipStreamSynchronize(I.stream));
        for (int s = 0; s < SF3D_SLOTS; ++s) {
            const uint8_t* d = I.dslot.data() + (size_t)s * N; double* dst = m.lflowSum[s].data();
            for (size_t i = 0; i < N; ++i) dst[i] = I.stage[(size_t)d[i] * N + i];
        }
    }
    m.hostStaleFlows = false;
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::fetch_heat(HostModel& m)
{
    Impl& I = *impl_;
    const size_t N = m.N;
    if (!I.v.heat.on) return SF3D_OK;
    HIP_TRY(hipMemcpyAsync(m.temperature.data(), I.v.heat.TX[mirror_.tCur], N * 8, hipMemcpyDeviceToHost, I.stream));
    std::vector<double>* dst[6] = {&m.bAero, &m.bSoilCond, &m.bSens, &m.bLat, &m.bRad, &m.bAdv};
    for (int k = 0; k < 6; ++k) HIP_TRY(hipMemcpyAsync(dst[k]->data(), I.heatOut[k], N * 8, hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipStreamSynchronize(I.stream));
    m.hostStaleHeat = false;
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::fetch_link_flux(HostModel& m, int type)
{
    Impl& I = *impl_;
    if (type < 0 || type >= SF3D_FLUX_TYPES || !I.v.heat.on || !I.v.heat.lflux[type]) return SF3D_MISSING_DATA_ERROR;
    const size_t NS = (size_t)m.N * SF3D_SLOTS;
    m.lfluxCache[type].resize(NS);
    if (I.dslot.empty()) {
        HIP_TRY(hipMemcpyAsync(m.lfluxCache[type].data(), I.v.heat.lflux[type], NS * 8, hipMemcpyDeviceToHost, I.stream));
        HIP_TRY(hipStreamSynchronize(I.stream));
    } else {                                       /* device slots -> insertion order */
        I.stage.resize(NS);
        HIP_TRY(hipMemcpyAsync(I.stage.data(), I.v.heat.lflux[type], NS * 8, hipMemcpyDeviceToHost, I.stream));
        HIP_TRY(hipStreamSynchronize(I.stream));
        const size_t N = m.N;
        for (int s = 0; s < SF3D_SLOTS; ++s) {
            const uint8_t* d = I.dslot.data() + (size_t)s * N; double* dst = m.lfluxCache[type].data() + (size_t)s * N;
            for (size_t i = 0; i < N; ++i) dst[i] = I.stage[(size_t)d[i] * N + i];
        }
    }
    m.lfluxValid[type] = true;
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::heat_query(HostModel& m, const ParamsHost& p, int what, uint32_t node, double h, double* out)
{
    sf3d_error_t e = sync_to_device(m, p);
    if (e != SF3D_OK) return e;
    Impl& I = *impl_;
    if (!I.v.heat.on) return SF3D_MISSING_DATA_ERROR;
    hipLaunchKernelGGL(k_heat_query, dim3(1), dim3(1), 0, I.stream, I.v, what, node, h);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&I.hostCtrl->query[1], &I.v.ctrl->query[1], sizeof(double), hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipStreamSynchronize(I.stream));
    *out = I.hostCtrl->query[1];
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::heat_storage(HostModel& m, const ParamsHost& p, double* out)
{
    sf3d_error_t e = sync_to_device(m, p);
    if (e != SF3D_OK) return e;
    Impl& I = *impl_;
    if (!I.v.heat.on) return SF3D_MISSING_DATA_ERROR;
    hipLaunchKernelGGL(k_heat_storage, dim3(I.v.nb), dim3(SF3D_BLOCK), 0, I.stream, I.v);
    hipLaunchKernelGGL(k_decide_query, dim3(1), dim3(SF3D_BLOCK), 0, I.stream, I.v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(I.hostCtrl, I.v.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipStreamSynchronize(I.stream));
    mirror_ = *I.hostCtrl;
    *out = mirror_.query[0];
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::dist_prepare(int rank, int world)
{
    if (world < 1 || world > SF3D_MAX_RANKS || rank < 0 || rank >= world) { snprintf(err_, sizeof(err_), "dist_prepare: bad rank/world %d/%d", rank, world); return SF3D_PARAMETER_ERROR; }
    if (built_) release();            /* a model built for another (or the same) partition: drop it, sf3d_initialize follows */
    world_ = world; rank_ = rank; connected_ = false;
    return SF3D_OK;
}

/* builds the device state (so the window exists) and returns what the peers need to reach it */
sf3d_error_t DeviceSolver::dist_export(HostModel& m, const ParamsHost& p, DistBlob* out)
{
    sf3d_error_t e = sync_to_device(m, p);
    if (e != SF3D_OK) return e;
    std::memset(out, 0, sizeof(*out));
    out->world = world_; out->rank = rank_; out->nodes = m.globalN ? m.globalN : m.N;
    if (world_ == 1) return SF3D_OK;
    Impl& I = *impl_;
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, I.window));
    static_assert(sizeof(h) <= sizeof(out->ipcHandle), "ipc handle size");
    std::memcpy(out->ipcHandle, &h, sizeof(h));
    for (int r = 0; r < world_; ++r) { out->recvOff[r] = I.hostDist.recvOff[r]; out->recvCount[r] = I.hostDist.recvCount[r]; }
    if (hipDeviceGetPCIBusId(out->pciBusId, (int)sizeof(out->pciBusId), I.device) != hipSuccess) out->pciBusId[0] = 0;
    out->generation = ++I.connectGen;
    {   /* the communicator id only when the RCCL exchange was asked for: ncclGetUniqueId starts RCCL's bootstrap root (listener thread and
         * socket), which the default window path has no use for */
        const char* xe = getenv("SF3D_EXCHANGE");
        if (rank_ == 0 && xe && std::strcmp(xe, "rccl") == 0 && I.load_rccl()) {
            ncclUniqueId id;
            if (I.pGetUniqueId(&id) == ncclSuccess) { static_assert(sizeof(id) <= sizeof(out->ncclId), "ncclUniqueId size"); std::memcpy(out->ncclId, &id, sizeof(id)); }
        }
    }
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::dist_connect(const DistBlob* all)
{
    if (world_ == 1) { connected_ = true; return SF3D_OK; }
    if (!built_) { snprintf(err_, sizeof(err_), "dist_connect before dist_export"); return SF3D_SOLVER_ERROR; }
    Impl& I = *impl_;
    HIP_TRY(hipSetDevice(I.device));
    static_assert(SF3D_MAX_RANKS >= 8, "a node has eight MI355X");
    DistView& d = I.hostDist;
    char why[200] = {0};                  /* why the window exchange is not usable on this rank (empty: it is) */
    for (int r = 0; r < world_; ++r) {
        const DistBlob& b = all[r];
        if ((int)b.world != world_ || (int)b.rank != r || b.nodes != all[rank_].nodes) { snprintf(err_, sizeof(err_), "dist_connect: blob %d does not match (world %u rank %u nodes %llu)", r, b.world, b.rank, (unsigned long long)b.nodes); return SF3D_PARAMETER_ERROR; }
        if (r == rank_) continue;
        /* both sides derived the lists from the same global graph: counts must agree */
        if (b.recvCount[rank_] != d.sendCount[r]) { snprintf(err_, sizeof(err_), "dist_connect: rank %d expects %u nodes from me, I send %u", r, b.recvCount[rank_], d.sendCount[r]); return SF3D_TOPOGRAPHY_ERROR; }
        d.sendOff[r] = b.recvOff[rank_];
        hipIpcMemHandle_t h;
        std::memcpy(&h, b.ipcHandle, sizeof(h));
        void* ptr = nullptr;
        const hipError_t oe = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (oe != hipSuccess) {
            (void)hipGetLastError();
            if (!why[0]) snprintf(why, sizeof(why), "hipIpcOpenMemHandle of rank %d's window failed: %s", r, hipGetErrorString(oe));
            d.win[r] = I.window; d.payload[r] = d.payload[rank_];        /* never dereferenced in the RCCL mode; keeps the view well-formed */
            continue;
        }
        I.peerMaps.push_back(ptr);
        d.win[r] = static_cast<DistWindow*>(ptr);
        d.payload[r] = reinterpret_cast<double*>(static_cast<char*>(ptr) + sizeof(DistWindow));
    }
    /* ---- which physical GPUs: distinct?  peer-reachable? ---- */
    const bool shareOk = getenv("SF3D_BENCH_SHARE_GPU") && getenv("SF3D_BENCH_SHARE_GPU")[0] == '1';
    const char* mine = all[rank_].pciBusId;
    bool distinct = true;
    for (int a = 0; a < world_; ++a)
        for (int r = a + 1; r < world_; ++r)
            if (!all[a].pciBusId[0] || !all[r].pciBusId[0] || std::strncmp(all[a].pciBusId, all[r].pciBusId, sizeof(all[r].pciBusId)) == 0) distinct = false;
    for (int r = 0; r < world_; ++r) {
        if (r == rank_ || !mine[0] || !all[r].pciBusId[0]) continue;
        if (std::strncmp(mine, all[r].pciBusId, sizeof(all[r].pciBusId)) == 0) {
            if (!shareOk && !I.warnedShared) { fprintf(stderr, "sf3d: warning: ranks %d and %d run on the same GPU (%s): one process per GPU is the intended layout (set LOCAL_RANK / sf3d_set_device)\n", rank_, r, mine); I.warnedShared = true; }
            continue;
        }
        int peerDev = -1, can = 1;
        if (hipDeviceGetByPCIBusId(&peerDev, all[r].pciBusId) == hipSuccess && peerDev >= 0 && peerDev != I.device
            && hipDeviceCanAccessPeer(&can, I.device, peerDev) == hipSuccess && !can && !why[0])
            snprintf(why, sizeof(why), "GPU %s (rank %d) cannot access GPU %s (rank %d) peer-to-peer", mine, rank_, all[r].pciBusId, r);
        (void)hipGetLastError();     /* a peer that is not visible to this process (one device per process) is checked by the ping alone */
    }
    const char* xe = getenv("SF3D_EXCHANGE");
    const bool forceRccl = xe && std::strcmp(xe, "rccl") == 0;
    HIP_TRY(hipMemcpy(I.devDist, &d, sizeof(DistView), hipMemcpyHostToDevice));
    I.v.dist = I.devDist;
    /* ---- one value through every window, both ways, within a bound ---- */
    if (!why[0] && !forceRccl) {
        double timeoutS = 5.0;
        if (const char* te = getenv("SF3D_DIST_PING_TIMEOUT_S")) { const double t = atof(te); if (t > 0) timeoutS = t; }
        int* dres = nullptr;
        HIP_TRY(hipMalloc(&dres, sizeof(int) * SF3D_MAX_RANKS));
        HIP_TRY(hipMemset(dres, 0, sizeof(int) * SF3D_MAX_RANKS));
        const unsigned long long token = 0x5F3D000000000000ull + (unsigned long long)all[0].generation;    /* the same on every rank */
        hipLaunchKernelGGL(k_dist_ping, dim3(1), dim3(64), 0, I.stream, d, token, (long long)(timeoutS * 1e8), dres);
        int res[SF3D_MAX_RANKS] = {0};
        hipError_t pe = hipMemcpyAsync(res, dres, sizeof(res), hipMemcpyDeviceToHost, I.stream);
        if (pe == hipSuccess) pe = hipStreamSynchronize(I.stream);
        (void)hipFree(dres);
        if (pe != hipSuccess) { snprintf(err_, sizeof(err_), "dist_connect: window self-check failed: %s (cross-device IPC mapping unusable?)", hipGetErrorString(pe)); fatal_ = true; return SF3D_SOLVER_ERROR; }
        int pos = 0; char who[96] = {0};
        for (int r = 0; r < world_; ++r) if (!res[r]) pos += snprintf(who + pos, sizeof(who) - pos, " %d", r);
        if (pos) snprintf(why, sizeof(why), "no answer through the window of rank(s)%s within %.0f s (peer not connected, or device-initiated stores do not cross GPUs here)", who, timeoutS);
    }
    /* what this rank found; the launcher gathers the ranks' findings and calls dist_finalize with the common decision */
    distStatus_ = (why[0] || forceRccl) ? 1 : 0;
    std::snprintf(distWhy_, sizeof(distWhy_), "%s", forceRccl ? "SF3D_EXCHANGE=rccl" : why);
    I.rcclDistinct = distinct;
    std::memcpy(I.rcclId, all[0].ncclId, sizeof(I.rcclId));
    connected_ = false;                   /* the ranks' common decision comes with sf3d_dist_finalize: one rank whose windows failed while the
                                           * others' passed must not leave the others spinning in the in-kernel exchange */
    return SF3D_OK;
}

/* the common decision of all ranks: keep the windows (all passed), or - only with SF3D_EXCHANGE=rccl - exchange over RCCL */
sf3d_error_t DeviceSolver::dist_finalize(bool useRccl)
{
    if (world_ == 1) { connected_ = true; return SF3D_OK; }
    if (!built_ || !impl_->v.dist) { snprintf(err_, sizeof(err_), "dist_finalize before dist_connect"); return SF3D_SOLVER_ERROR; }
    Impl& I = *impl_;
    if (!useRccl) {
        if (distStatus_ != 0) { snprintf(err_, sizeof(err_), "dist_finalize(windows), but rank %d: %.160s", rank_, distWhy_); return SF3D_SOLVER_ERROR; }
        connected_ = true;
        return SF3D_OK;
    }
    HIP_TRY(hipSetDevice(I.device));
    DistView& d = I.hostDist;
    const char* xe = getenv("SF3D_EXCHANGE");
    if (!(xe && std::strcmp(xe, "rccl") == 0)) {
        /* The RCCL exchange is opt-in: it has not run on hardware yet (its collectives are queued unconditionally inside look-ahead
         * batches and have no bounded wait), so a failed window check is a hard error on every rank instead of a silent switch */
        snprintf(err_, sizeof(err_), "rank %d: the window exchange failed its self-check on some rank (%.150s); the RCCL exchange is opt-in (SF3D_EXCHANGE=rccl)",
                 rank_, distWhy_[0] ? distWhy_ : "this rank's windows passed");
        fatal_ = true;
        return SF3D_SOLVER_ERROR;
    }
    bool haveId = false;
    for (size_t k = 0; k < sizeof(I.rcclId); ++k) if (I.rcclId[k]) haveId = true;
    if ((xe && std::strcmp(xe, "ipc") == 0) || !I.rcclDistinct || !haveId || !I.load_rccl()) {
        snprintf(err_, sizeof(err_), "rank %d: the window exchange is not usable (%.120s) and RCCL is not available either (%s)", rank_, distWhy_,
                 !I.rcclDistinct ? "ranks share a GPU: RCCL needs one GPU per rank" : (!haveId ? "no communicator id in rank 0's blob" : "SF3D_EXCHANGE=ipc / librccl not loadable"));
        return SF3D_SOLVER_ERROR;
    }
    if (I.v.heat.on) { snprintf(err_, sizeof(err_), "the RCCL exchange does not carry the coupled heat step (window exchange only)"); return SF3D_SOLVER_ERROR; }
    ncclUniqueId id;
    std::memcpy(&id, I.rcclId, sizeof(id));
    if (I.pCommInitRank(&I.comm, world_, id, rank_) != ncclSuccess) { I.comm = nullptr; snprintf(err_, sizeof(err_), "rank %d: ncclCommInitRank failed", rank_); return SF3D_SOLVER_ERROR; }
    HIP_TRY(dev_alloc(I.allocs, I.rcclMine, 4)); HIP_TRY(dev_alloc(I.allocs, I.rcclGathered, (size_t)3 * world_ + 1));
    d.rccl = 1; d.mine = I.rcclMine; d.gathered = I.rcclGathered;
    for (int r = 0; r < world_; ++r) {
        double *sb = nullptr, *rb = nullptr;
        HIP_TRY(dev_alloc(I.allocs, sb, (size_t)d.sendCount[r] * 2)); HIP_TRY(dev_alloc(I.allocs, rb, (size_t)d.recvCount[r] * 2));
        d.sendBuf[r] = sb; d.recvBuf[r] = rb;
    }
    I.v.haloDirect = 0;                  /* sweeps read the halo from the arrays the unpack filled, not from a window */
    I.useFused = 0;                      /* the exchange sits BETWEEN the kernels: separate decision kernels */
    I.rcclMode = true;
    if (rank_ == 0) fprintf(stderr, "sf3d: multi-GPU exchange over RCCL (ncclSend/ncclRecv halos + ncclAllGather of the partial sums)%s%s\n", distWhy_[0] ? ": " : "", distWhy_);
    HIP_TRY(hipMemcpy(I.devDist, &d, sizeof(DistView), hipMemcpyHostToDevice));
    connected_ = true;
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::total_water_content(HostModel& m, const ParamsHost& p, double* out)
{
    sf3d_error_t e = sync_to_device(m, p);
    if (e != SF3D_OK) return e;
    if (world_ > 1 && !connected_) { snprintf(err_, sizeof(err_), "multi-GPU model used before sf3d_dist_connect / sf3d_dist_finalize%s%.150s", distWhy_[0] ? ": " : "", distWhy_); return SF3D_SOLVER_ERROR; }
    Impl& I = *impl_;
    hipLaunchKernelGGL(k_storage, dim3(I.v.nb), dim3(SF3D_BLOCK), 0, I.stream, I.v);
    if (I.rcclMode) { hipLaunchKernelGGL(k_local_reduce, dim3(1), dim3(SF3D_BLOCK), 0, I.stream, I.v, 0); I.rccl_gather(I.stream); }
    hipLaunchKernelGGL(k_decide_query, dim3(1), dim3(SF3D_BLOCK), 0, I.stream, I.v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(I.hostCtrl, I.v.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, I.stream));
    HIP_TRY(hipStreamSynchronize(I.stream));
    mirror_ = *I.hostCtrl;
    *out = mirror_.query[0];
    return SF3D_OK;
}

/* start-up self-check of the sharded path (dist_connect): thread p stores a token into rank p's window through the IPC mapping
 * (system-scope store over xGMI) and waits until rank p's token shows up in the own window; out[p] = 1 answered, 0 timed out */
__global__ void k_dist_ping(DistView d, unsigned long long token, long long timeoutTicks, int* out)
{
    const int p = threadIdx.x;
    if (p >= d.world) return;
    if (p == d.rank) { out[p] = 1; return; }
    __hip_atomic_store(&d.win[p]->ping[d.rank], token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();        /* 100 MHz */
    int ok = 1;
    while (__hip_atomic_load(&d.win[d.rank]->ping[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != token) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeoutTicks) { ok = 0; break; }
    }
    out[p] = ok;
}

__global__ void k_device_log(const double* x, double* y, uint32_t n)
{
    fm_init();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = flog(x[i]);
}
__global__ void k_device_exp(const double* x, double* y, uint32_t n)
{
    fm_init();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fexp(x[i]);
}
__global__ void k_device_pow(const double* x, const double* e, double* y, uint32_t n)
{
    fm_init();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = ppow(x[i], e[i]);
}

/* test hooks: the kernels' logarithm / pow on host values */
namespace {
struct DevBuf {                      /* scratch device array, released on every exit path */
    double* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}
sf3d_error_t DeviceSolver::device_log(uint32_t n, const double* x, double* out, bool exponential)
{
    if (n == 0) return SF3D_OK;
    DevBuf dx, dy;
    HIP_TRY(hipMalloc(&dx.p, (size_t)n * 8));
    HIP_TRY(hipMalloc(&dy.p, (size_t)n * 8));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
    if (exponential) hipLaunchKernelGGL(k_device_exp, dim3((n + 255) / 256), dim3(256), 0, 0, dx.p, dy.p, n);
    else hipLaunchKernelGGL(k_device_log, dim3((n + 255) / 256), dim3(256), 0, 0, dx.p, dy.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dy.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::device_pow(uint32_t n, const double* x, const double* y, double* out)
{
    if (n == 0) return SF3D_OK;
    DevBuf dx, dy, dz;
    HIP_TRY(hipMalloc(&dx.p, (size_t)n * 8));
    HIP_TRY(hipMalloc(&dy.p, (size_t)n * 8));
    HIP_TRY(hipMalloc(&dz.p, (size_t)n * 8));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy.p, y, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_device_pow, dim3((n + 255) / 256), dim3(256), 0, 0, dx.p, dy.p, dz.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dz.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::timing(int enable)
{
    if (!impl_) impl_ = new Impl();
    impl_->timing = enable;
    if (enable) { for (int k = 0; k < KID_COUNT; ++k) { impl_->launches[k] = 0; impl_->ms[k] = 0; } }
    return SF3D_OK;
}

sf3d_error_t DeviceSolver::stats(int kid, uint64_t* launches, double* ms, uint64_t* nodes)
{
    if (!impl_ || kid < 0 || kid >= KID_COUNT) return SF3D_INDEX_ERROR;
    *launches = impl_->launches[kid]; *ms = impl_->ms[kid]; *nodes = impl_->N;
    return SF3D_OK;
}

/* One computeStep (soilFluxes3D.cpp:1785-1821 -> CPUSolver::run -> waterMainLoop).
 * The host queues one approximation per batch - every kernel guarded by the device stage -
 * and polls the control block once per batch. */
sf3d_error_t DeviceSolver::step(HostModel& m, ParamsHost& p, double maxTimeStep, double* dtOut)
{
    sf3d_error_t e = sync_to_device(m, p);
    if (e != SF3D_OK) return e;
    if (world_ > 1 && !connected_) { snprintf(err_, sizeof(err_), "multi-GPU model used before sf3d_dist_connect / sf3d_dist_finalize%s%.150s", distWhy_[0] ? ": " : "", distWhy_); return SF3D_SOLVER_ERROR; }
    Impl& I = *impl_;
    const DevView& v = I.v;
    const dim3 grid(v.nb), block(SF3D_BLOCK), one(1);
    const bool multi = world_ > 1;
    const bool heatOn = v.heat.on != 0;
    if (I.useFused < 0) { const char* e = getenv("SF3D_FUSED_DECIDE"); I.useFused = (e && e[0] == '0') ? 0 : 1; }
    if (I.residentGrids < 0) { const char* e = getenv("SF3D_RESIDENT_GRIDS"); I.residentGrids = (e && e[0] == '0') ? 0 : 1; }
    const bool compat = v.compatCv != nullptr;   /* quirk-1 emulation: rows are stored raw and normalised by k_compat_rows after the Courant decision */
    /* grids of the register-heavy kernels: exactly as many blocks as are resident at once - equal work per block, no tail
     * round (at 70 VGPRs only 1 792 of 2 048 blocks fit and the remaining 256 ran alone afterwards).  Every kernel walks the
     * chunk list with its own gridDim, and the fused reductions count gridDim partials, so any grid is valid. */
    auto resident = [&](const void* fn) -> dim3 {
        uint32_t nbk = 0;
        for (auto& r : I.residentBlocks) if (r.first == fn) nbk = r.second;
        if (nbk == 0) {
            int perCu = 0, dev = 0; hipDeviceProp_t prop;
            nbk = SF3D_MAX_BLOCKS;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, fn, SF3D_BLOCK, 0) == hipSuccess && perCu > 0
                && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                nbk = (uint32_t)perCu * (uint32_t)prop.multiProcessorCount;
            if (I.residentGrids == 0) nbk = SF3D_MAX_BLOCKS;
            I.residentBlocks.push_back({fn, nbk});
        }
        return dim3(v.nb < nbk ? v.nb : nbk);
    };
    if (heatOn && multi) I.useFused = 1;       /* the sharded heat step exists only in the fused-exchange form */
    const bool fused = !multi && I.useFused;   /* sweep + convergence decision in one launch (single GPU) */
    const bool fusedMulti = multi && I.useFused; /* + in-kernel halo puts and all-gather (multi GPU) */
    const bool linealOn = p.lineal && v.cgDiag != nullptr && !multi && I.useFused;     /* the linealia stand-in: device conjugate gradients */
    const bool pairOn = fused && I.pairBlocks != 0 && !linealOn;   /* k_sweep_pair instead of k_sweep */
    const dim3 pgrid(I.pushBlocks ? I.pushBlocks : 1);
    hipStream_t st = I.stream;

    /* mode 2 samples: the sweeps of every 8th computeStep carry HIP events (eager launches); the other
     * steps replay hipGraphs, so the measurement costs ~1 % instead of ~6 % */
    const bool timedStep = I.timing == 1 || (I.timing == 2 && (I.stepSeq++ % 8 == 0));
    if (I.overlapAccept < 0) { const char* oe = getenv("SF3D_OVERLAP_ACCEPT"); I.overlapAccept = (oe && oe[0] == '0') ? 0 : 1; }
    /* single GPU only: a second stream per process is a second hardware queue, and ranks that share a GPU (functional
     * multi-rank tests) oversubscribe the queues - their spinning exchange kernels then wait for time slices (measured: 45x slower) */
    if (I.overlapAccept && !multi && !I.stream2) { HIP_TRY(hipStreamCreateWithFlags(&I.stream2, hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&I.evLinks[0], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&I.evLinks[1], hipEventDisableTiming)); }
    /* accepted step: link flow sums on a second stream next to the next step's k_props (untimed steps only, so that the
     * per-kernel event timing of --time-all-kernels stays a sequence of exclusive launches) */
    const bool overlap = I.overlapAccept && !multi && I.stream2 && !timedStep && I.timing != 1 && v.compatCv == nullptr;   /* (compat: k_compat_rows rewrites what the sums read) */
    auto timed = [&](int kid, auto launch) {
        if (!timedStep || (I.timing == 2 && kid != KID_SWEEP && kid != KID_SWEEP_PAIR)) { launch(); return; }
        hipEvent_t a, b;
        if (I.freeEvents.size() >= 2) { a = I.freeEvents.back(); I.freeEvents.pop_back(); b = I.freeEvents.back(); I.freeEvents.pop_back(); }
        else { hipEventCreate(&a); hipEventCreate(&b); }
        hipEventRecord(a, st); launch(); hipEventRecord(b, st);
        I.pending.push_back({a, b, kid});
    };

    if (heatOn) {                           /* computeStep head, soilFluxes3D.cpp:1787-1791 */
        if (v.heat.save == 2) hipLaunchKernelGGL(k_heat_reset_water_fluxes, grid, block, 0, st, v);
        hipLaunchKernelGGL(k_heat_conductance, grid, block, 0, st, v);
        for (int t = 0; t < SF3D_FLUX_TYPES; ++t) m.lfluxValid[t] = false;
    }
    uint32_t stage = ST_ACCEPT;
    if (m.water) {
    {   /* this step assembles into the copy of the matrix that the step before the last one used (k_step_begin flips Ctrl::aBuf): the
         * link flow sums that read it - queued two steps ago on the second stream - must be done; so must they before this step's
         * sweeps reuse the head buffer they read.  The sums of the LAST step run next to this whole step - unless this step adds its
         * own sums inside the step (event-timed steps): then both must be done first. */
        const uint32_t wb = (mirror_.aBuf ^ 1u) & 1u;
        if (I.linksPending[wb]) { HIP_TRY(hipStreamWaitEvent(st, I.evLinks[wb], 0)); I.linksPending[wb] = false; }
        if (!overlap && I.linksPending[wb ^ 1u]) { HIP_TRY(hipStreamWaitEvent(st, I.evLinks[wb ^ 1u], 0)); I.linksPending[wb ^ 1u] = false; }
    }
    {
    hipLaunchKernelGGL(k_step_begin, one, one, 0, st, v.ctrl, maxTimeStep);
    stage = ST_APPROX;                      /* k_step_begin opens the first attempt */
    uint64_t before[8], atStart[8];
    std::memcpy(before, mirror_.counters, sizeof(before));
    std::memcpy(atStart, mirror_.counters, sizeof(atStart));
    int guard = 0;
    uint64_t pairBefore = mirror_.pairLaunches, singleBefore = mirror_.singleLaunches;

    const dim3 propsGrid = resident((const void*)k_props<0, false>), propsHeatGrid = resident((const void*)k_props<0, true>);
    const dim3 acceptGrid = v.ntStream ? resident((const void*)k_accept<true>) : resident((const void*)k_accept<false>);
    /* one approximation's worth of guarded kernels */
    auto enqueue_props = [&] {
            if (heatOn && multi) {      /* sharded heat always uses the fused exchange; halo conductivities are recomputed locally */
                timed(KID_PROPS, [&] { hipLaunchKernelGGL((k_props<2, true>), grid, block, 0, st, v); });
                hipLaunchKernelGGL(k_halo_copy<0>, pgrid, block, 0, st, v);
                hipLaunchKernelGGL(k_heat_halo_water, pgrid, block, 0, st, v);
            }
            else if (heatOn) timed(KID_PROPS, [&] { hipLaunchKernelGGL((k_props<0, true>), propsHeatGrid, block, 0, st, v); });
            else if (multi && fusedMulti) { timed(KID_PROPS, [&] { hipLaunchKernelGGL((k_props<2, false>), grid, block, 0, st, v); }); hipLaunchKernelGGL(k_halo_copy<0>, pgrid, block, 0, st, v); }
            else timed(KID_PROPS, [&] { hipLaunchKernelGGL((k_props<0, false>), propsGrid, block, 0, st, v); });
            if (multi && !fusedMulti) {
                hipLaunchKernelGGL(k_push_kf, pgrid, block, 0, st, v);
                if (I.rcclMode) I.rccl_halo(st, 2);
                hipLaunchKernelGGL(k_sync_kf, one, block, 0, st, v);
            }
    };
    static const bool asmNtOff = getenv("SF3D_ASM_NT") && getenv("SF3D_ASM_NT")[0] == '0';      /* tuning: cacheable stores of the rows */
    const bool asmNT = v.ntStream && !asmNtOff;
    auto enqueue_batch = [&](bool withHead, bool withTail, bool skipProps, uint32_t chunk) {
        if (withHead) {
            if (!skipProps) enqueue_props();
            if (heatOn && I.useFused) timed(KID_ASSEMBLE, [&] { hipLaunchKernelGGL((k_assemble<true, false, true>), dim3(v.nbSurf + v.nbSoil), block, 0, st, v); });
            else if (heatOn) { timed(KID_ASSEMBLE, [&] { hipLaunchKernelGGL((k_assemble<false, false, true>), dim3(v.nbSurf + v.nbSoil), block, 0, st, v); }); hipLaunchKernelGGL(k_decide_courant, one, block, 0, st, v); }
            else if (I.useFused) timed(KID_ASSEMBLE, [&] { if (asmNT) hipLaunchKernelGGL((k_assemble<true, true, false>), dim3(v.nbSurf + v.nbSoil), block, 0, st, v); else hipLaunchKernelGGL((k_assemble<true, false, false>), dim3(v.nbSurf + v.nbSoil), block, 0, st, v); });
            else {
                timed(KID_ASSEMBLE, [&] { if (asmNT) hipLaunchKernelGGL((k_assemble<false, true, false>), dim3(v.nbSurf + v.nbSoil), block, 0, st, v); else hipLaunchKernelGGL((k_assemble<false, false, false>), dim3(v.nbSurf + v.nbSoil), block, 0, st, v); });
                if (I.rcclMode) { hipLaunchKernelGGL(k_local_reduce, one, block, 0, st, v, 1); I.rccl_gather(st); }
                hipLaunchKernelGGL(k_decide_courant, one, block, 0, st, v);
            }
        }
        if (withHead && compat) hipLaunchKernelGGL(k_compat_rows, grid, block, 0, st, v);
        if (linealOn) {     /* conjugate gradients instead of Jacobi sweeps: `chunk` counts Jacobi sweeps, CG needs far fewer iterations */
            hipLaunchKernelGGL(k_cg_init, grid, block, 0, st, v);
            for (uint32_t k = 0; k < chunk; ++k) {
                timed(KID_SWEEP, [&] { hipLaunchKernelGGL(k_cg_matvec, grid, block, 0, st, v); hipLaunchKernelGGL(k_cg_update, grid, block, 0, st, v);
                                       hipLaunchKernelGGL(k_cg_dir, grid, block, 0, st, v); });
            }
            hipLaunchKernelGGL(k_cg_finish, grid, block, 0, st, v);
        } else
        if (pairOn) {      /* two Jacobi iterations per launch (regular grid, one GPU).  `chunk` = expected iterations + 1: floor(expected / 2)
                            * pairs, a single sweep when the expectation is odd (122 us instead of a 175 us pair whose second half
                            * would be thrown away), then one more pair as the margin (a guarded no-op when the expectation holds) */
            const dim3 pgr(I.pairBlocks), pbl((v.pair.W + 1) * 64);
            const uint32_t expected = chunk > 1 ? chunk - 1 : 1;
            static const bool oddSingle = !(getenv("SF3D_PAIR_ODD_SINGLE") && getenv("SF3D_PAIR_ODD_SINGLE")[0] == '0');
            const uint32_t nPairs = oddSingle ? expected / 2 + 1 : (chunk + 1) / 2, singleAfter = (oddSingle && (expected & 1u)) ? expected / 2 : UINT32_MAX;
            for (uint32_t k = 0; k < nPairs; ++k) {
                if (k == singleAfter) timed(KID_SWEEP, [&] { if (v.ntStream) hipLaunchKernelGGL((k_sweep<1, true>), grid, block, 0, st, v); else hipLaunchKernelGGL((k_sweep<1, false>), grid, block, 0, st, v); });
                timed(KID_SWEEP_PAIR, [&] {
                    if (I.pairMasked) {
                        switch (v.pair.W * 2 + (v.ntStream ? 1 : 0)) {
                            case 12: hipLaunchKernelGGL((k_sweep_pair_masked<6, false>), pgr, pbl, 0, st, v); break;
                            case 13: hipLaunchKernelGGL((k_sweep_pair_masked<6, true>), pgr, pbl, 0, st, v); break;
                            case 20: hipLaunchKernelGGL((k_sweep_pair_masked<10, false>), pgr, pbl, 0, st, v); break;
                            case 21: hipLaunchKernelGGL((k_sweep_pair_masked<10, true>), pgr, pbl, 0, st, v); break;
                            case 28: hipLaunchKernelGGL((k_sweep_pair_masked<14, false>), pgr, pbl, 0, st, v); break;
                            default: hipLaunchKernelGGL((k_sweep_pair_masked<14, true>), pgr, pbl, 0, st, v); break;
                        }
                        return;
                    }
                    switch (v.pair.W * 2 + (v.ntStream ? 1 : 0)) {
                        case 12: hipLaunchKernelGGL((k_sweep_pair<6, false>), pgr, pbl, 0, st, v); break;
                        case 13: hipLaunchKernelGGL((k_sweep_pair<6, true>), pgr, pbl, 0, st, v); break;
                        case 20: hipLaunchKernelGGL((k_sweep_pair<10, false>), pgr, pbl, 0, st, v); break;
                        case 21: hipLaunchKernelGGL((k_sweep_pair<10, true>), pgr, pbl, 0, st, v); break;
                        case 28: hipLaunchKernelGGL((k_sweep_pair<14, false>), pgr, pbl, 0, st, v); break;
                        default: hipLaunchKernelGGL((k_sweep_pair<14, true>), pgr, pbl, 0, st, v); break;
                    }
                });
            }
        } else
        for (uint32_t k = 0; k < chunk; ++k) {
            if (fused) { timed(KID_SWEEP, [&] { if (v.ntStream) hipLaunchKernelGGL((k_sweep<1, true>), grid, block, 0, st, v); else hipLaunchKernelGGL((k_sweep<1, false>), grid, block, 0, st, v); }); continue; }
            if (fusedMulti) { timed(KID_SWEEP, [&] { if (v.ntStream) hipLaunchKernelGGL((k_sweep<2, true>), grid, block, 0, st, v); else hipLaunchKernelGGL((k_sweep<2, false>), grid, block, 0, st, v); }); continue; }
            timed(KID_SWEEP, [&] { if (v.ntStream) hipLaunchKernelGGL((k_sweep<0, true>), grid, block, 0, st, v); else hipLaunchKernelGGL((k_sweep<0, false>), grid, block, 0, st, v); });
            if (multi) hipLaunchKernelGGL(k_push_x, pgrid, block, 0, st, v);
            if (I.rcclMode) { I.rccl_halo(st, 1); hipLaunchKernelGGL(k_local_reduce, one, block, 0, st, v, 0); I.rccl_gather(st); }
            hipLaunchKernelGGL(k_decide_sweep, one, block, 0, st, v);
        }
        if (I.useFused) { timed(KID_POST, [&] { hipLaunchKernelGGL(k_post<true>, grid, block, 0, st, v); }); if (multi && v.haloDirect) hipLaunchKernelGGL(k_halo_copy<1>, pgrid, block, 0, st, v); }
        else {
            timed(KID_POST, [&] { hipLaunchKernelGGL(k_post<false>, grid, block, 0, st, v); });
            if (I.rcclMode) { hipLaunchKernelGGL(k_local_reduce, one, block, 0, st, v, 2); I.rccl_gather(st); }
            hipLaunchKernelGGL(k_decide_balance, one, block, 0, st, v);
        }
        if (withTail) {     /* restore-best and the flow sums of the accepted step: once per poll group */
            if (heatOn && I.useFused) timed(KID_RESTORE, [&] { hipLaunchKernelGGL((k_restore<true, true>), grid, block, 0, st, v); });
            else if (I.useFused) timed(KID_RESTORE, [&] { hipLaunchKernelGGL((k_restore<true, false>), grid, block, 0, st, v); });
            else {
                timed(KID_RESTORE, [&] { if (heatOn) hipLaunchKernelGGL((k_restore<false, true>), grid, block, 0, st, v); else hipLaunchKernelGGL((k_restore<false, false>), grid, block, 0, st, v); });
                if (I.rcclMode) { hipLaunchKernelGGL(k_local_reduce, one, block, 0, st, v, 2); I.rccl_gather(st); }
                hipLaunchKernelGGL(k_decide_restore, one, block, 0, st, v);
            }
            if (overlap) hipLaunchKernelGGL(k_accept_boundary, grid, block, 0, st, v);
            else timed(KID_ACCEPT, [&] { if (v.ntStream) hipLaunchKernelGGL(k_accept<true>, acceptGrid, block, 0, st, v); else hipLaunchKernelGGL(k_accept<false>, acceptGrid, block, 0, st, v); });
        }
    };

    /* Look-ahead: the previous step needed `lastBatches` approximations; queue that many guarded
     * batches before the first poll (a batch queued in vain costs ~25 no-op launches, a poll costs
     * a full host round trip while the GPU idles).  After the first poll continue one at a time. */
    /* The ~25 launches of a batch are replayed from an instantiated hipGraph (one per shape): small
     * grids are bound by the host's launch rate otherwise.  Event timing needs eager launches. */
    if (I.useGraphs < 0) { const char* e = getenv("SF3D_GRAPHS"); I.useGraphs = (e && e[0] == '0') ? 0 : 1; }
    /* how many sweeps to queue for the `index`-th approximation of this step: what the same approximation of the previous step took,
     * plus one (Ctrl::seqSweeps; a sweep queued in vain is a guarded no-op of ~4 us, one too few costs a poll; sweeps that a later
     * batch queued simply continue an unfinished approximation); without a record, the last count plus two */
    auto predicted_sweeps = [&](uint32_t index, bool withHead) -> uint32_t {
        uint32_t chunk;
        if (withHead && index < I.predCount && index < 16u) { chunk = I.pred[index] + 1; if (chunk < 2) chunk = 2; }
        else { chunk = I.lastSweeps + 2; if (chunk < 4) chunk = 4; }
        if (chunk > 40) chunk = 40;
        return chunk;
    };
    auto launch_batch = [&](bool withHead, bool withTail, bool skipProps, uint32_t chunk) -> hipError_t {
        if (!I.useGraphs || timedStep || I.rcclMode) { enqueue_batch(withHead, withTail, skipProps, chunk); return hipSuccess; }   /* (RCCL calls are queued eagerly) */
        const uint32_t key = (withHead ? 1u : 0u) | (withTail ? 2u : 0u) | (chunk << 2) | (skipProps ? 1u << 20 : 0u) | (overlap ? 1u << 21 : 0u) | (pairOn ? 1u << 22 : 0u) | (linealOn ? 1u << 23 : 0u);
        for (auto& g : I.graphs) if (g.first == key) return hipGraphLaunch(g.second, st);
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) return e;
        enqueue_batch(withHead, withTail, skipProps, chunk);
        e = hipStreamEndCapture(st, &graph);
        if (e != hipSuccess) return e;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e != hipSuccess) return e;
        I.graphs.push_back({key, exec});
        return hipGraphLaunch(exec, st);
    };

    uint32_t look = I.lastBatches < 1 ? 1 : (I.lastBatches > 6 ? 6 : I.lastBatches);
    uint32_t batchIndex = 0;
    while (true) {
        for (uint32_t bq = 0; bq < look; ++bq) {
            const bool head = bq > 0 || stage == ST_APPROX;
            HIP_TRY(launch_batch(head, bq + 1 == look, false, predicted_sweeps(batchIndex, head)));
            if (head) ++batchIndex;
        }
        look = 1;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(I.hostCtrl, v.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const Ctrl& c = *I.hostCtrl;

        if (timedStep) {
            /* attribute event pairs only to launches that really ran: how many of each kernel ran
             * comes from the device counters; guarded no-op launches are the shortest of a group */
            uint64_t ran[KID_COUNT];
            ran[KID_PROPS] = ran[KID_ASSEMBLE] = c.counters[2] - before[2];
            ran[KID_SWEEP] = pairOn ? c.singleLaunches - singleBefore : c.counters[3] - before[3];
            ran[KID_SWEEP_PAIR] = c.pairLaunches - pairBefore;
            ran[KID_POST] = c.counters[7] - before[7];
            ran[KID_RESTORE] = c.counters[6] - before[6];
            ran[KID_ACCEPT] = c.counters[1] - before[1];
            std::vector<float> el[KID_COUNT];
            for (auto& pr : I.pending) {
                float t = 0.f;
                if (hipEventElapsedTime(&t, pr.a, pr.b) == hipSuccess) el[pr.kid].push_back(t);
                I.freeEvents.push_back(pr.a); I.freeEvents.push_back(pr.b);
            }
            I.pending.clear();
            for (int k = 0; k < KID_COUNT; ++k) {
                std::sort(el[k].begin(), el[k].end(), [](float x, float y) { return x > y; });
                for (size_t q = 0; q < el[k].size() && q < ran[k]; ++q) { I.ms[k] += el[k][q]; I.launches[k]++; }
            }
        }
        std::memcpy(before, c.counters, sizeof(before));
        pairBefore = c.pairLaunches; singleBefore = c.singleLaunches;

        stage = c.stage;
        if (stage != ST_SWEEP && c.iter > 0) I.lastSweeps = c.iter;
        if (stage == ST_ACCEPT || stage == ST_FAIL) break;   /* ST_ACCEPT: bookkeeping done and k_accept has run */
        if (++guard > 1000000) { snprintf(err_, sizeof(err_), "step state machine did not terminate (stage %u)", stage); return SF3D_SOLVER_ERROR; }
    }
    {   /* approximations this step took (rejected attempts included) = batches the next one will queue up front */
        const uint64_t used = I.hostCtrl->counters[2] - atStart[2];
        I.lastBatches = used < 1 ? 1u : (uint32_t)used;
        I.predCount = I.hostCtrl->seqCount < 16u ? I.hostCtrl->seqCount : 16u;
        for (uint32_t k = 0; k < I.predCount; ++k) I.pred[k] = I.hostCtrl->seqSweeps[k];
    }
    }
    if (overlap && stage == ST_ACCEPT) {
        /* the main stream is drained (the poll just read the control block): no dependency to express for the launch */
        uint32_t lcap = 384u; if (const char* le = getenv("SF3D_LINKS_BLOCKS")) lcap = (uint32_t)atoi(le);      /* measured at C4: 2048 -1 %, 512 / 256 +1 %, 128 -4 % */
        const dim3 lgrid(v.nb > lcap ? lcap : v.nb);        /* a streaming kernel: few enough waves that k_props fits next to it */
        const Ctrl& hc = *I.hostCtrl;
        const uint32_t rb = hc.acceptABuf & 1u;
        if (v.ntStream) hipLaunchKernelGGL(k_accept_links<true>, lgrid, block, 0, I.stream2, v, hc.acceptBuf, rb, hc.acceptDt);
        else hipLaunchKernelGGL(k_accept_links<false>, lgrid, block, 0, I.stream2, v, hc.acceptBuf, rb, hc.acceptDt);
        HIP_TRY(hipEventRecord(I.evLinks[rb], I.stream2));
        I.linksPending[rb] = true;
    }
    }   /* if (m.water) */

    if (heatOn && stage == ST_ACCEPT) {
        /* heat part of computeStep (soilFluxes3D.cpp:1802-1818): every kernel is guarded by Ctrl::hStage; one batch =
         * boundary, node properties, rows, a run of sweeps, balance, flux bookkeeping; polled once per batch */
        hipLaunchKernelGGL(k_heat_begin, one, one, 0, st, v.ctrl, maxTimeStep, m.water ? 1 : 0);
        const dim3 gSaveWater = resident((const void*)k_heat_save_water), gBoundary = resident((const void*)k_heat_boundary),
                   gHProps = resident((const void*)k_heat_props), gHAsm = resident((const void*)k_heat_assemble),
                   gHPost = resident((const void*)k_heat_post), gHSave = resident((const void*)k_heat_save);
        hipLaunchKernelGGL(k_heat_save_water, gSaveWater, block, 0, st, v);
        int hguard = 0;
        uint32_t lookH = I.lastHeatSteps < 1 ? 1 : (I.lastHeatSteps > 8 ? 8 : I.lastHeatSteps);
        if (v.heat.gs) lookH = 1;              /* one launch per dependency level: keep the queue short */
        if (I.useGraphs < 0) { const char* e = getenv("SF3D_GRAPHS"); I.useGraphs = (e && e[0] == '0') ? 0 : 1; }
        /* one poll group = lookH guarded heat steps (look-ahead: as many as the last computeStep needed, <= 8), replayed
         * from an instantiated hipGraph per shape like the water batches */
        auto enqueue_heat = [&](uint32_t steps, uint32_t chunk) {
            for (uint32_t bq = 0; bq < steps; ++bq) {
                hipLaunchKernelGGL(k_heat_boundary, gBoundary, block, 0, st, v);
                hipLaunchKernelGGL(k_heat_props, gHProps, block, 0, st, v);
                hipLaunchKernelGGL(k_heat_assemble, gHAsm, block, 0, st, v);
                if (v.heat.gs) {
                    for (uint32_t k = 0; k < chunk; ++k) {
                        hipLaunchKernelGGL(k_heat_gs_begin, grid, block, 0, st, v);
                        for (size_t l = 0; l + 1 < I.gsLevelStart.size(); ++l) {
                            const uint32_t first = I.gsLevelStart[l], cnt = I.gsLevelStart[l + 1] - first;
                            if (cnt) hipLaunchKernelGGL(k_heat_gs_level, dim3((cnt + SF3D_BLOCK - 1) / SF3D_BLOCK), block, 0, st, v, first, cnt);
                        }
                        hipLaunchKernelGGL(k_heat_gs_decide, one, one, 0, st, v.ctrl);
                    }
                } else
                for (uint32_t k = 0; k < chunk; ++k) hipLaunchKernelGGL(k_heat_sweep, grid, block, 0, st, v);
                hipLaunchKernelGGL(k_heat_post, gHPost, block, 0, st, v);
                if (v.heat.save != 0) hipLaunchKernelGGL(k_heat_save, gHSave, block, 0, st, v);
            }
        };
        auto launch_heat = [&](uint32_t steps, uint32_t chunk) -> hipError_t {
            if (!I.useGraphs || v.heat.gs) { enqueue_heat(steps, chunk); return hipSuccess; }
            const uint32_t key = 0x80000000u | (chunk << 8) | (steps << 1) | (v.heat.save != 0 ? 1u : 0u);
            for (auto& g : I.graphs) if (g.first == key) return hipGraphLaunch(g.second, st);
            hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e != hipSuccess) return e;
            enqueue_heat(steps, chunk);
            e = hipStreamEndCapture(st, &graph);
            if (e != hipSuccess) return e;
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (e != hipSuccess) return e;
            I.graphs.push_back({key, exec});
            return hipGraphLaunch(exec, st);
        };
        while (true) {
            uint32_t chunk = ((I.lastHeatSweeps + 2 + 3) / 4) * 4;      /* multiples of four: few graph shapes */
            if (chunk < 4) chunk = 4;
            if (chunk > 64) chunk = 64;
            HIP_TRY(launch_heat(lookH, chunk));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(I.hostCtrl, v.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const Ctrl& c = *I.hostCtrl;
            if (c.hSweepsLast > 0) I.lastHeatSweeps = c.hSweepsLast;
            if (getenv("SF3D_HEAT_DEBUG") && getenv("SF3D_HEAT_DEBUG")[0] == '2') fprintf(stderr, "gpu heatLoop stage %u next dt %g outer %g/%g sweeps %u MBR %.6e storage %.12e sink %.6e courant %.6e\n", c.hStage, c.hDt, c.hOuterDt, c.hOuterSum, c.hSweepsLast, c.heatCur.MBR, c.heatCur.storage, c.heatCur.sinkSource, c.hCourant);
            if (c.hStage == HS_FINISHED) { I.lastHeatSteps = c.hRows + c.hPad; break; }
            if (c.hStage == HS_IDLE) { snprintf(err_, sizeof(err_), "heat step did not start (water stage %u)", c.stage); stage = ST_FAIL; fatal_ = true; break; }
            if (++hguard > 1000000) { snprintf(err_, sizeof(err_), "heat state machine did not terminate (stage %u)", c.hStage); return SF3D_SOLVER_ERROR; }
        }
        if (getenv("SF3D_HEAT_DEBUG")) fprintf(stderr, "heat: water dt %g: %u heat steps accepted, %u halved, last dt %g, last sweeps %u, courant %g, MBR %g\n", I.hostCtrl->dtWater, I.hostCtrl->hRows, I.hostCtrl->hPad, I.hostCtrl->hDt, I.hostCtrl->hSweepsLast, I.hostCtrl->hCourant, I.hostCtrl->heatCur.MBR);
        m.hostStaleHeat = true;
    } else if (!m.water) {
        HIP_TRY(hipMemcpyAsync(I.hostCtrl, v.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    mirror_ = *I.hostCtrl;
    if (mirror_.distError) { snprintf(err_, sizeof(err_), "rank %d: a peer did not answer within the bounded wait (multi-GPU exchange)", rank_); fatal_ = true; }
    else if (stage != ST_ACCEPT && !fatal_) snprintf(err_, sizeof(err_), "the mass balance is not a number at the minimum time step (stepNan, cpusolver.cpp:176-180): state left as the reference leaves it");
    p.dtCurr = mirror_.dtCurr;
    *dtOut = mirror_.dt;
    m.hostStaleState = true;
    m.hostStaleFlows = true;
    return (stage == ST_ACCEPT) ? SF3D_OK : SF3D_SOLVER_ERROR;
}
