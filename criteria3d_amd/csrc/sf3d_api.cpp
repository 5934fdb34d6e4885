/*
 * sf3d_api.cpp - the C ABI of include/sf3d.h for the HIP product library (libsf3d_hip.so).
 *
 * Mirrors the reference's API layer (agrolib/soilFluxes3D/soilFluxes3D.cpp): same validation
 * rules, return codes and sentinels, same call-order assumptions.  Setters write a host staging
 * model and raise dirty flags (O(1), no device traffic); the time step itself runs on the GPU
 * (sf3d_solver.hip).  There is NO CPU implementation of the time step in this library: without a
 * HIP device computeStep fails loudly (message on stderr, NaN returned).
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <sys/mman.h>
#include <unistd.h>

#include "sf3d_model.h"

namespace {

HostModel M;
ParamsHost P;                                   /* persists across re-initialisation */
std::vector<std::vector<uint16_t>> soil1D;      /* soil1DIndices, soilFluxes3D.cpp:39 */
bool useLineal = false; int linealMethod = 0;
struct Bal { double storage = 0, sinkSource = 0, MBE = 0, MBR = 0; };
Bal curPeriod, wholePeriod;                     /* period balances live on the host (computePeriod) */
struct HeatBal { double storage = 0, sinkSource = 0, MBE = 0, MBR = 0; };
HeatBal heatCurPeriod, heatWholePeriod;         /* balanceDataCurrentPeriod / WholePeriod, heat members */
struct HeatFlagsHost { bool vapor = false, advection = false; uint8_t save = 0; } HF;   /* simulationFlags_t: survives re-initialisation */
uint64_t counterBase[8] = {0};

DeviceSolver& dev() { return DeviceSolver::instance(); }

double errValue(sf3d_error_t e)                  /* getDoubleErrorValue, types.h:42-64 */
{
    switch (e) {
        case SF3D_OK: return 0;
        case SF3D_INDEX_ERROR: return SF3D_VAL_INDEX_ERROR;
        case SF3D_MEMORY_ERROR: return SF3D_VAL_MEMORY_ERROR;
        case SF3D_TOPOGRAPHY_ERROR: return SF3D_VAL_TOPOGRAPHY_ERROR;
        case SF3D_BOUNDARY_ERROR: return SF3D_VAL_BOUNDARY_ERROR;
        case SF3D_MISSING_DATA_ERROR: return SF3D_VAL_MISSING_DATA_ERROR;
        case SF3D_PARAMETER_ERROR: return SF3D_VAL_PARAMETER_ERROR;
        default: return SF3D_VAL_INDEX_ERROR;
    }
}

/* ---- host-side soil functions used by the state setters/getters only (soilPhysics.cpp) ---- */
double seFromPsi(const SoilHost& s, double psi)                     /* :91-115 */
{
    switch (P.wrc) {
        case SF3D_WRC_VAN_GENUCHTEN: return std::pow(1.0 + std::pow(s.alpha * psi, s.n), -s.m);
        case SF3D_WRC_MODIFIED_VAN_GENUCHTEN:
            if (psi <= s.he) return 1.0;
            return std::pow(1.0 + std::pow(s.alpha * psi, s.n), -s.m) * (1.0 / s.Sc);
        default: return SF3D_NODATA;
    }
}
double nodeSe(uint32_t i)                                           /* :68-83 */
{
    if (M.H[i] >= M.z[i]) return 1.;
    return seFromPsi(M.soils[M.cls[i]], std::fabs(M.H[i] - M.z[i]));
}
double mualemK(const SoilHost& s, double Se)                        /* :181-214 */
{
    if (Se >= 1.0) return s.Ksat;
    const double invM = 1.0 / s.m;
    double temp;
    switch (P.wrc) {
        case SF3D_WRC_VAN_GENUCHTEN: temp = 1.0 - std::pow(1.0 - std::pow(Se, invM), s.m); break;
        case SF3D_WRC_MODIFIED_VAN_GENUCHTEN:
            temp = (1.0 - std::pow(1.0 - std::pow(Se * s.Sc, invM), s.m)) / s.mualemDen; break;
        default: return SF3D_NODATA;
    }
    return s.Ksat * std::pow(Se, s.L) * (temp * temp);
}
double thetaFromSe(const SoilHost& s, double Se) { return (Se * (s.thetaS - s.thetaR)) + s.thetaR; }   /* :38-42 */
/* computeNodeK as the state setters call it (soilFluxes3D.cpp:831,859,881,903): Mualem + the isothermal vapour
 * conductivity when latent heat is on (soilPhysics.cpp:164-172, heat.cpp:831-845, 1080-1087, 1136-1176) */
double nodeKHost(uint32_t i)
{
    const SoilHost& s = M.soils[M.cls[i]];
    double k = mualemK(s, M.Se[i]);
    if (M.heat && HF.vapor) {
        const double T = (M.temperature[i] + M.temperature[i]) * 0.5;      /* temperature == oldTemperature outside a step */
        const double h = M.H[i] - M.z[i];
        const double theta = (h >= 0.) ? s.thetaS : thetaFromSe(s, seFromPsi(s, std::fabs(h)));
        const double vDiff = (0.0000212 * std::pow(T / 273.15, 2.)) * 0.66 * std::pow(s.thetaS - theta, 1.);
        const double svp = 611 * std::exp(17.502 * (T - 273.15) / ((T - 273.15) + 240.97));
        const double svc = (svp * 0.018 / (8.31447215 * T));
        const double rh = std::exp(0.018 * h * 9.80665 / (8.31447215 * T));
        const double vConc = svc * rh;
        k += ((vDiff * vConc * 0.018) / (8.31447215 * T)) * (9.80665 / 1000.);
    }
    return k;
}
double nodeTheta(uint32_t i) { return M.surf[i] ? 1. : thetaFromSe(M.soils[M.cls[i]], M.Se[i]); }      /* :26-32 */
double thetaFromSignedPsi(uint32_t i, double psi)                   /* :50-61 */
{
    if (M.surf[i]) return 1.;
    const SoilHost& s = M.soils[M.cls[i]];
    if (psi >= 0.) return s.thetaS;
    return thetaFromSe(s, seFromPsi(s, std::fabs(psi)));
}
double seFromTheta(const SoilHost& s, double theta)                 /* :123-134 */
{
    if (theta >= s.thetaS) return 1.;
    if (theta < s.thetaR) return 0.;
    return (theta - s.thetaR) / (s.thetaS - s.thetaR);
}
double nodePsi(uint32_t i)                                          /* :140-158 */
{
    const SoilHost& s = M.soils[M.cls[i]];
    double temp;
    switch (P.wrc) {
        case SF3D_WRC_VAN_GENUCHTEN: temp = std::pow(1. / M.Se[i], 1. / s.m) - 1.; break;
        case SF3D_WRC_MODIFIED_VAN_GENUCHTEN: temp = std::pow(1. / (M.Se[i] * s.Sc), 1. / s.m) - 1; break;
        default: return SF3D_NODATA;
    }
    return (1. / s.alpha) * std::pow(temp, 1. / s.n);
}

/* ---- multi-GPU: strip-local device models --------------------------------------------------------------------------------------
 * The caller builds the same GLOBAL model on every rank (the API is global: a drop-in keeps its caller).  What goes to the device of
 * rank r is only what its rows touch: the nodes it owns plus the one-cell halo, renumbered 0..n-1 in global order (surface nodes
 * first, layer-major: the numbering the kernels rely on survives, and so do the chunk-uniform link offsets inside a layer), links
 * remapped, owners and halo lists handed over in local indices (HostModel::presetPartition).  Device memory and upload time per rank
 * are then ~1/world of the global model's (+ halo) instead of all of it.  M stays the staging copy the setters and getters see; L
 * is what the solver sees; dirty parts go M -> L before every device call, fetched fields come back L -> M for the local nodes
 * (halo included: the device's halo values are as current as its last exchange, exactly what the global-index path keeps).
 * SF3D_DIST_LOCAL=0 keeps the global-index path of rounds 1-2 (every rank uploads the whole model): the checker of this one. */
struct LocalModel {
    bool on = false;                 /* decided at sf3d_dist_prepare */
    bool built = false;
    bool trimmed = false;            /* trimHostStaging() has run: M holds this rank's nodes only (the other pages were given back) */
    HostModel L;
    std::vector<uint32_t> l2g;
    std::vector<int32_t> g2l;
    Partition part;                  /* local indices */
    Partition gpart;                 /* global indices: what sf3d_dist_owner / sf3d_dist_halo answer from once M is trimmed */
};
LocalModel LM;
int distRank = 0, distWorld = 1;

template <class T> void gatherTo(std::vector<T>& dst, const std::vector<T>& src)
{
    dst.resize(LM.l2g.size());
    if (src.size() < M.N) { std::fill(dst.begin(), dst.end(), T()); return; }
    for (size_t k = 0; k < LM.l2g.size(); ++k) dst[k] = src[LM.l2g[k]];
}
template <class T> void scatterFrom(std::vector<T>& dst, const std::vector<T>& src)
{
    if (dst.size() < M.N || src.size() < LM.l2g.size()) return;
    for (size_t k = 0; k < LM.l2g.size(); ++k) dst[LM.l2g[k]] = src[k];
}

/* (re)build L from M: the rank's nodes, links remapped, partition in local indices */
sf3d_error_t buildLocal()
{
    Partition& gp = LM.gpart;
    sf3d_error_t e = sf3d_compute_partition(M, distRank, distWorld, gp);
    if (e == SF3D_MISSING_DATA_ERROR && gp.missingFrom != UINT32_MAX)
        fprintf(stderr, "sf3d: rank %d: node %u of this rank's strip is linked to node %u, which was never given a soil / surface class - a strip-local "
                        "build must stage the whole one-cell ring of columns around the strip\n", distRank, gp.missingFrom, gp.missingTo);
    if (e != SF3D_OK) return e;
    LM.trimmed = false;
    std::vector<uint8_t> in(M.N, 0);
    for (uint32_t i = 0; i < M.N; ++i) if (gp.owner[i] == distRank) in[i] = 1;
    for (int p = 0; p < distWorld; ++p) for (uint32_t i : gp.recv[p]) in[i] = 1;
    LM.l2g.clear(); LM.g2l.assign(M.N, -1);
    for (uint32_t i = 0; i < M.N; ++i) if (in[i]) { LM.g2l[i] = (int32_t)LM.l2g.size(); LM.l2g.push_back(i); }
    HostModel& L = LM.L;
    L = HostModel();
    L.initialized = M.initialized; L.solverReady = M.solverReady; L.water = M.water; L.heat = M.heat; L.solutes = M.solutes;
    L.cgArrays = M.cgArrays; L.compat = M.compat; L.heatVapor = M.heatVapor; L.heatAdvection = M.heatAdvection; L.heatSave = M.heatSave;
    L.N = (uint32_t)LM.l2g.size();
    L.ns = 0; for (uint32_t g : LM.l2g) if (g < M.ns) ++L.ns;
    L.globalN = M.N;
    gatherTo(L.x, M.x); gatherTo(L.y, M.y); gatherTo(L.z, M.z); gatherTo(L.size, M.size); gatherTo(L.surf, M.surf);
    gatherTo(L.hasClass, M.hasClass); gatherTo(L.cls, M.cls); gatherTo(L.btype, M.btype); gatherTo(L.bslope, M.bslope); gatherTo(L.bsize, M.bsize);
    gatherTo(L.bflowRate, M.bflowRate); gatherTo(L.bflowSum, M.bflowSum); gatherTo(L.prescribed, M.prescribed); gatherTo(L.nLat, M.nLat);
    for (int s = 0; s < SF3D_SLOTS; ++s) {
        gatherTo(L.ltype[s], M.ltype[s]); gatherTo(L.lto[s], M.lto[s]); gatherTo(L.larea[s], M.larea[s]); gatherTo(L.lflowSum[s], M.lflowSum[s]);
        for (uint32_t k = 0; k < L.N; ++k) {
            if (L.ltype[s][k] == SF3D_LINK_NONE) { L.lto[s][k] = 0; continue; }
            const int32_t t = LM.g2l[L.lto[s][k]];
            if (t < 0) { L.ltype[s][k] = SF3D_LINK_NONE; L.lto[s][k] = 0; }      /* a halo node's link out of the local set: its row is never computed */
            else L.lto[s][k] = (uint32_t)t;
        }
    }
    gatherTo(L.Se, M.Se); gatherTo(L.K, M.K); gatherTo(L.H, M.H); gatherTo(L.sink, M.sink); gatherTo(L.pond, M.pond);
    L.soils = M.soils; L.roughness = M.roughness;
    if (M.heat) {
        gatherTo(L.temperature, M.temperature); gatherTo(L.heatSink, M.heatSink);
        gatherTo(L.bHeightWind, M.bHeightWind); gatherTo(L.bHeightT, M.bHeightT); gatherTo(L.bRoughH, M.bRoughH); gatherTo(L.bT, M.bT); gatherTo(L.bRH, M.bRH);
        gatherTo(L.bWind, M.bWind); gatherTo(L.bNetIrr, M.bNetIrr); gatherTo(L.bFixT, M.bFixT); gatherTo(L.bFixDepth, M.bFixDepth);
        gatherTo(L.bAero, M.bAero); gatherTo(L.bSoilCond, M.bSoilCond); gatherTo(L.bSens, M.bSens); gatherTo(L.bLat, M.bLat); gatherTo(L.bRad, M.bRad); gatherTo(L.bAdv, M.bAdv);
    }
    /* the partition in local indices (lists stay sorted: the local numbering is monotonic in the global one, so both ends of an
     * exchange still enumerate its nodes in the same order) */
    Partition& lp = LM.part;
    lp = Partition(); lp.world = distWorld; lp.rank = distRank; lp.bounds = gp.bounds;
    lp.owner.resize(L.N);
    for (uint32_t k = 0; k < L.N; ++k) lp.owner[k] = gp.owner[LM.l2g[k]];
    lp.send.assign(distWorld, {}); lp.recv.assign(distWorld, {});
    for (int p = 0; p < distWorld; ++p) {
        for (uint32_t g : gp.send[p]) lp.send[p].push_back((uint32_t)LM.g2l[g]);
        for (uint32_t g : gp.recv[p]) lp.recv[p].push_back((uint32_t)LM.g2l[g]);
    }
    L.presetPartition = &LM.part;
    /* everything is new to the device */
    L.graphDirty = L.stateDirty = L.sinkDirty = L.pondDirty = L.boundaryDirty = L.flowSumsDirty = L.ctrlDirty = true;
    L.sinkLo = 0; L.sinkHi = UINT32_MAX;
    L.heatStateDirty = L.heatSinkDirty = L.heatBoundaryDirty = true;
    M.graphDirty = M.stateDirty = M.sinkDirty = M.pondDirty = M.boundaryDirty = M.flowSumsDirty = M.ctrlDirty = false;
    M.heatStateDirty = M.heatSinkDirty = M.heatBoundaryDirty = false;
    M.hostStaleState = M.hostStaleFlows = M.hostStaleHeat = false;
    LM.built = true;
    return SF3D_OK;
}

/* ---- host staging per rank -----------------------------------------------------------------------------------------------------
 * The caller builds the GLOBAL model on every rank (the API is global).  Once the ranks are connected the topology is frozen (an
 * edit is an error until a new prepare / export / connect round), so the global staging copy M is only ever read and written at
 * THIS rank's nodes - gathers into L, scatters out of it, setters of its own nodes.  trimHostStaging() then gives the pages of every
 * array of M that hold only OTHER ranks' nodes back to the system (madvise MADV_DONTNEED: the vectors stay valid, the pages read as
 * zero and are re-faulted if a caller insists on writing there - bulk setters skip such nodes) and frees the arrays of L that only the
 * graph build reads (coordinates, link tables, areas).  What stays resident per rank is its strip + halo: at eight ranks of C4 about a
 * fifth of the single-rank figure (sf3d_host_bytes, asserted in tests/test_gpu_multirank.py).  Partition queries answer from the copy
 * kept in LM.gpart. */
template <class F> void forEachArray(HostModel& h, F f)
{
    f(h.x); f(h.y); f(h.z); f(h.size); f(h.surf); f(h.hasClass); f(h.cls); f(h.btype); f(h.bslope); f(h.bsize); f(h.bflowRate); f(h.bflowSum);
    f(h.prescribed); f(h.nLat);
    for (int s = 0; s < SF3D_SLOTS; ++s) { f(h.ltype[s]); f(h.lto[s]); f(h.larea[s]); f(h.lflowSum[s]); }
    f(h.Se); f(h.K); f(h.H); f(h.sink); f(h.pond);
    f(h.temperature); f(h.heatSink); f(h.bHeightWind); f(h.bHeightT); f(h.bRoughH); f(h.bT); f(h.bRH); f(h.bWind); f(h.bNetIrr); f(h.bFixT); f(h.bFixDepth);
    f(h.bAero); f(h.bSoilCond); f(h.bSens); f(h.bLat); f(h.bRad); f(h.bAdv);
    for (int t = 0; t < SF3D_FLUX_TYPES; ++t) f(h.lfluxCache[t]);
}
template <class T> void releasePages(std::vector<T>& v, size_t a, size_t b)      /* elements [a, b): whole pages inside the range */
{
    if (v.size() < b || b <= a) return;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    uintptr_t lo = (uintptr_t)(v.data() + a), hi = (uintptr_t)(v.data() + b);
    lo = (lo + page - 1) / page * page; hi = hi / page * page;
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_DONTNEED);
}
template <class T> uint64_t residentBytes(const std::vector<T>& v)
{
    if (v.empty()) return 0;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = (uintptr_t)v.data() / page * page, hi = ((uintptr_t)(v.data() + v.size()) + page - 1) / page * page;
    std::vector<unsigned char> vec((hi - lo) / page);
    if (mincore((void*)lo, hi - lo, vec.data()) != 0) return (uint64_t)v.size() * sizeof(T);
    uint64_t n = 0;
    for (unsigned char c : vec) n += c & 1u;
    return n * page;
}
void trimHostStaging()
{
    if (!LM.on || !LM.built || LM.trimmed || LM.g2l.size() != M.N) return;
    if (const char* e = getenv("SF3D_DIST_TRIM_HOST")) if (e[0] == '0') return;
    std::vector<std::pair<size_t, size_t>> runs;          /* maximal runs of nodes that are not this rank's (owned or halo) */
    for (size_t i = 0; i < M.N;) {
        if (LM.g2l[i] >= 0) { ++i; continue; }
        size_t j = i;
        while (j < M.N && LM.g2l[j] < 0) ++j;
        if (j - i >= 512) runs.push_back({i, j});
        i = j;
    }
    forEachArray(M, [&](auto& v) { if (v.size() >= M.N) for (auto& r : runs) releasePages(v, r.first, r.second); });
    HostModel& L = LM.L;                                  /* read by the graph build only: no rebuild while connected */
    auto drop = [](auto& v) { std::remove_reference_t<decltype(v)>().swap(v); };
    drop(L.x); drop(L.y); drop(L.z); drop(L.size); drop(L.surf); drop(L.hasClass); drop(L.cls); drop(L.nLat);
    for (int s = 0; s < SF3D_SLOTS; ++s) { drop(L.ltype[s]); drop(L.lto[s]); drop(L.larea[s]); }
    LM.trimmed = true;
}
inline bool skippedByTrim(uint32_t i) { return LM.trimmed && i < LM.g2l.size() && LM.g2l[i] < 0; }

bool needState(); bool needFlows(); bool needHeatState();
/* the model the device works on, brought up to date with what the setters changed in M since the last call */
HostModel& deviceModel()
{
    if (!LM.on) return M;
    if (!LM.built || M.graphDirty) {
        if (LM.built && dev().ready() && dev().connected()) {
            /* the topology changed AFTER sf3d_dist_connect: the windows, the halo lists and the peers' view of this rank belong to the old
             * graph.  Nothing is torn down behind the launcher's back: the local model is marked stale and the device call that follows
             * fails with SF3D_TOPOGRAPHY_ERROR (sync_to_device) until a new sf3d_dist_prepare / export / connect / finalize round. */
            LM.L.graphDirty = true;
            return LM.L;
        }
        if (LM.built && dev().ready()) {                         /* a re-built topology before the ranks are connected: what the device knows better goes back to M first, */
            needState(); needFlows(); if (M.heat) needHeatState();
            dev().release();                                     /* then windows, arrays and graphs go, like a re-initialisation */
        }
        if (buildLocal() != SF3D_OK) {
            /* (the partition fails only for a bad rank / world, which sf3d_dist_prepare has refused already) - never hand a half-built
             * model to the solver: an empty one fails its upload loudly */
            fprintf(stderr, "sf3d: strip-local model: partition failed (bad rank / world, or a strip-local build without its halo columns)\n");
            LM.L = HostModel(); LM.built = false;
        }
        return LM.L;
    }
    HostModel& L = LM.L;
    if (M.stateDirty) { gatherTo(L.H, M.H); gatherTo(L.Se, M.Se); gatherTo(L.K, M.K); L.stateDirty = true; L.hostStaleState = false; M.stateDirty = false; }
    if (M.sinkDirty) { gatherTo(L.sink, M.sink); L.sinkDirty = true; L.sinkLo = 0; L.sinkHi = UINT32_MAX; M.sinkDirty = false; }
    if (M.pondDirty) { gatherTo(L.pond, M.pond); L.pondDirty = true; M.pondDirty = false; }
    if (M.boundaryDirty) { gatherTo(L.btype, M.btype); gatherTo(L.bslope, M.bslope); gatherTo(L.bsize, M.bsize); gatherTo(L.prescribed, M.prescribed); L.boundaryDirty = true; M.boundaryDirty = false; }
    if (M.flowSumsDirty) {
        gatherTo(L.bflowSum, M.bflowSum);
        for (int s = 0; s < SF3D_SLOTS; ++s) gatherTo(L.lflowSum[s], M.lflowSum[s]);
        L.flowSumsDirty = true; L.hostStaleFlows = false; M.flowSumsDirty = false;
    }
    if (M.ctrlDirty) { L.ctrlDirty = true; M.ctrlDirty = false; }
    if (M.heat) {
        if (M.heatStateDirty) { gatherTo(L.temperature, M.temperature); L.heatStateDirty = true; L.hostStaleHeat = false; M.heatStateDirty = false; }
        if (M.heatSinkDirty) { gatherTo(L.heatSink, M.heatSink); L.heatSinkDirty = true; M.heatSinkDirty = false; }
        if (M.heatBoundaryDirty) {
            gatherTo(L.bHeightWind, M.bHeightWind); gatherTo(L.bHeightT, M.bHeightT); gatherTo(L.bRoughH, M.bRoughH); gatherTo(L.bT, M.bT); gatherTo(L.bRH, M.bRH);
            gatherTo(L.bWind, M.bWind); gatherTo(L.bNetIrr, M.bNetIrr); gatherTo(L.bFixT, M.bFixT); gatherTo(L.bFixDepth, M.bFixDepth);
            L.heatBoundaryDirty = true; M.heatBoundaryDirty = false;
        }
    }
    return L;
}
/* is the host copy of a field older than the device's? (asked of the model the device works on) */
bool staleState() { return LM.on ? (LM.built && LM.L.hostStaleState) : M.hostStaleState; }
bool staleFlows() { return LM.on ? (LM.built && LM.L.hostStaleFlows) : M.hostStaleFlows; }
bool staleHeat() { return LM.on ? (LM.built && LM.L.hostStaleHeat) : M.hostStaleHeat; }

/* make the host copy of a field current before it is read or partially overwritten */
bool needState()
{
    if (staleState() && dev().ready()) {
        HostModel& D = LM.on ? LM.L : M;
        if (dev().fetch_state(D) != SF3D_OK) { fprintf(stderr, "sf3d: %s\n", dev().last_error()); return false; }
        if (LM.on) { scatterFrom(M.H, D.H); scatterFrom(M.Se, D.Se); scatterFrom(M.K, D.K); }
        /* surface conventions of the setters (soilFluxes3D.cpp:819-820, 880-881) */
        for (uint32_t i = 0; i < M.ns; ++i) { M.Se[i] = 1.; }
    }
    return true;
}
bool needFlows()
{
    if (staleFlows() && dev().ready()) {
        HostModel& D = LM.on ? LM.L : M;
        if (dev().fetch_flows(D) != SF3D_OK) { fprintf(stderr, "sf3d: %s\n", dev().last_error()); return false; }
        if (LM.on) {
            scatterFrom(M.bflowSum, D.bflowSum); scatterFrom(M.bflowRate, D.bflowRate);
            for (int s = 0; s < SF3D_SLOTS; ++s) scatterFrom(M.lflowSum[s], D.lflowSum[s]);
        }
    }
    return true;
}

template <class T> void reset(std::vector<T>& v, size_t n) { v.assign(n, T()); }

#define NEED_INIT_E   if (!M.initialized) return SF3D_MEMORY_ERROR
#define NEED_NODE_E(i) if ((i) >= M.N) return SF3D_INDEX_ERROR
#define NEED_INIT_D   if (!M.initialized) return errValue(SF3D_MEMORY_ERROR)
/* (a node of another rank's strip once the staging copy is trimmed: NODATA, like the bulk getters - sf3d_dist_owner says whose it is) */
/* (ONE statement - an else-if chain closed by an empty else - so that it stays whole under an unbraced if) */
#define NEED_NODE_D(i) if ((i) >= M.N) return errValue(SF3D_INDEX_ERROR); else if (skippedByTrim(i)) return (double)SF3D_NODATA; else (void)0

}  // namespace

extern "C" {

const char* sf3d_backend_name(void) { return "hip"; }

sf3d_error_t sf3d_clean(void)                                       /* soilFluxes3D.cpp:218-304 */
{
    if (!M.initialized) return SF3D_OK;
    dev().release();
    M = HostModel();
    LM.built = false; LM.trimmed = false; LM.L = HostModel(); LM.l2g.clear(); LM.g2l.clear(); LM.gpart = Partition();       /* (LM.on stays: sf3d_dist_prepare comes before sf3d_initialize) */
    return SF3D_OK;
}

sf3d_error_t sf3d_initialize(uint32_t n, uint32_t ns, uint8_t nLat, int w, int h, int s, sf3d_heat_save_t saveMode)   /* :49-178 */
{
    sf3d_error_t c = sf3d_clean();
    if (c != SF3D_OK) return c;
    M.water = w != 0; M.heat = h != 0; M.solutes = s != 0;
    { const char* ce = getenv("SF3D_COMPAT_STALE_LINK_FLOW"); M.compat = ce && ce[0] == '1'; }
    { const char* ge = getenv("SF3D_LINEAL_DEVICE_CG"); M.cgArrays = ge && ge[0] == '1'; P.lineal = useLineal && M.cgArrays; }
    if (M.heat) { HF.vapor = true; HF.advection = true; HF.save = saveMode; }        /* :58-65 */
    M.heatVapor = HF.vapor; M.heatAdvection = HF.advection; M.heatSave = HF.save;
    M.N = n; M.ns = ns;
    if (nLat > 8) return SF3D_PARAMETER_ERROR;
    try {
        reset(M.x, n); reset(M.y, n); reset(M.z, n); reset(M.size, n); reset(M.surf, n);
        reset(M.hasClass, n); reset(M.cls, n); reset(M.btype, n); reset(M.bslope, n); reset(M.bsize, n);
        reset(M.bflowRate, n); reset(M.bflowSum, n); reset(M.prescribed, n); reset(M.nLat, n);
        for (int k = 0; k < SF3D_SLOTS; ++k) { reset(M.ltype[k], n); reset(M.lto[k], n); reset(M.larea[k], n); reset(M.lflowSum[k], n); }
        reset(M.Se, n); reset(M.K, n); reset(M.H, n); reset(M.sink, n); reset(M.pond, n);
        if (M.heat) {
            for (auto* v : {&M.temperature, &M.heatSink, &M.bHeightWind, &M.bHeightT, &M.bRoughH, &M.bT, &M.bRH, &M.bWind, &M.bNetIrr,
                            &M.bFixT, &M.bFixDepth, &M.bAero, &M.bSoilCond, &M.bSens, &M.bLat, &M.bRad, &M.bAdv}) reset(*v, n);
        }
    } catch (const std::bad_alloc&) { return SF3D_MEMORY_ERROR; }
    /* a rank of a multi-GPU run may stage only its strip and the ring of columns around it (sf3d_dist_bounds): the freshly zeroed
     * pages go back to the system now and come back - still zero - where a setter writes, so what a rank never stages costs nothing */
    if (LM.on && distWorld > 1) forEachArray(M, [&](auto& v) { releasePages(v, 0, v.size()); });
    M.initialized = true;
    if (P.dtCurr == SF3D_NODATA) P.dtCurr = P.dtMax;               /* CPUSolver::initialize, cpusolver.cpp:30-31 */
    M.solverReady = true;
    std::memset(counterBase, 0, sizeof(counterBase));
    curPeriod = Bal(); wholePeriod = Bal();
    heatCurPeriod = HeatBal(); heatWholePeriod = HeatBal();
    return SF3D_OK;
}

sf3d_error_t sf3d_initialize_balance(void)                          /* soilFluxes3D.cpp:184-197, water.cpp:35-65 */
{
    if (!M.initialized) return SF3D_MEMORY_ERROR;
    double wc = 0.;
    sf3d_error_t e = dev().total_water_content(deviceModel(), P, &wc);
    if (e != SF3D_OK) { fprintf(stderr, "sf3d: initializeBalance: %s\n", dev().last_error()); return e; }
    Ctrl& c = dev().ctrl();
    wholePeriod.storage = curPeriod.storage = wc;
    c.curStep.storage = c.prevStep.storage = wc;
    c.curStep.sinkSource = c.prevStep.sinkSource = 0.; c.curPeriod.sinkSource = 0.;
    curPeriod.sinkSource = wholePeriod.sinkSource = 0.;
    c.curStep.MBR = 0.; wholePeriod.MBR = 0.; c.curStep.MBE = 0.; wholePeriod.MBE = 0.;
    dev().push_ctrl();
    needFlows();
    if (LM.trimmed) {          /* this rank's nodes only: the other pages of the staging copy were given back (and read as zero anyway) */
        for (uint32_t g : LM.l2g) { for (int k = 0; k < SF3D_SLOTS; ++k) M.lflowSum[k][g] = 0.; M.bflowSum[g] = 0.; }
    } else {
        for (int k = 0; k < SF3D_SLOTS; ++k) std::fill(M.lflowSum[k].begin(), M.lflowSum[k].end(), 0.);
        std::fill(M.bflowSum.begin(), M.bflowSum.end(), 0.);
    }
    M.flowSumsDirty = true;
    if (M.heat) {                                                   /* initializeHeatBalance, heat.cpp:31-53 */
        double hs = 0.;
        e = dev().heat_storage(deviceModel(), P, &hs);
        if (e != SF3D_OK) { fprintf(stderr, "sf3d: initializeBalance (heat): %s\n", dev().last_error()); return e; }
        Ctrl& ch = dev().ctrl();
        heatWholePeriod = HeatBal(); heatCurPeriod = HeatBal();
        heatWholePeriod.storage = heatCurPeriod.storage = hs;
        ch.heatCur = HeatBalanceDev{hs, 0., 0., 0.}; ch.heatPrev = HeatBalanceDev{hs, 0., 0., 0.};
        ch.heatPeriodSink = 0.;
        dev().push_ctrl();
    } else heatWholePeriod.MBR = 1.;
    return SF3D_OK;
}
sf3d_error_t sf3d_initialize_log(const char*, const char*) { return SF3D_OK; }   /* MCR logging is not built (parallel.pri:15-16) */
sf3d_error_t sf3d_close_log(void) { return SF3D_OK; }
sf3d_error_t sf3d_initialize_heat_flag(sf3d_heat_save_t save, int adv, int latent)   /* soilFluxes3D.cpp:325-332 */
{
    HF.save = save; HF.advection = adv != 0; HF.vapor = latent != 0;
    if (M.initialized && (M.heatSave != HF.save || M.heatAdvection != HF.advection || M.heatVapor != HF.vapor)) {
        M.heatVapor = HF.vapor; M.heatAdvection = HF.advection; M.heatSave = HF.save;
        M.graphDirty = true;                  /* which flux arrays exist and which kernels' terms are live change */
    }
    return SF3D_OK;
}

uint32_t sf3d_set_threads_number(uint32_t n)                        /* soilFluxes3D.cpp:340-361: clamp and report; the GPU path has no host threads to set */
{
    uint32_t hw = std::thread::hardware_concurrency();
    if (n < 1 || n > hw) n = hw;
    if (M.solverReady) P.numThreads = n;
    return n;
}
/* setUseLineal, soilFluxes3D.cpp:367-372.  The third-party liblinealia is not part of the repository and is never loaded; by default the
 * library keeps its own Jacobi (= the reference with useLineal = false).  With SF3D_LINEAL_DEVICE_CG=1 in the environment a true value
 * selects the device's Jacobi-preconditioned conjugate gradients for the linear systems (one GPU; parity with linealia is unpinned by
 * construction - there is no binary to compare with - so it is checked against the Jacobi solution instead, tests/test_gpu_cg.py) */
void sf3d_set_use_lineal(int v)
{
    if (!M.solverReady) return;
    useLineal = v != 0;
    const bool want = useLineal && M.cgArrays;
    if (want != P.lineal) { P.lineal = want; M.ctrlDirty = true; }
}
void sf3d_set_lineal_method(int v) { if (M.solverReady) linealMethod = v; }

sf3d_error_t sf3d_set_soil_properties(uint16_t nrSoil, uint8_t nrHorizon, double alpha, double n, double m,
                                      double he, double thetaR, double thetaS, double kSat, double L,
                                      double om, double clay)       /* soilFluxes3D.cpp:395-449 */
{
    if (alpha <= 0 || n <= 1.0 || m <= 0.0 || m >= 1.0 || he < 0.0 || kSat <= 0.0 || thetaR < 0.0 ||
        thetaR >= 1.0 || thetaS <= 0.0 || thetaS > 1.0 || thetaR > thetaS)
        return SF3D_PARAMETER_ERROR;
    for (const SoilHost& s : M.soils)
        if (s.soilNumber == nrSoil && s.horizonNumber == nrHorizon) return SF3D_PARAMETER_ERROR;
    if (M.soils.size() > std::numeric_limits<uint16_t>::max()) return SF3D_MEMORY_ERROR;
    SoilHost s{nrSoil, nrHorizon, alpha, n, m, he, 0., thetaS, thetaR, kSat, L, om, clay, 0.};
    s.Sc = std::pow(1. + std::pow(alpha * he, n), -m);
    s.mualemDen = 1.0 - std::pow(1.0 - std::pow(s.Sc, 1.0 / m), m);
    if (nrSoil >= soil1D.size()) soil1D.resize(nrSoil + 1);
    if (nrHorizon >= soil1D[nrSoil].size()) soil1D[nrSoil].resize(nrHorizon + 1);
    M.soils.push_back(s);
    soil1D[nrSoil][nrHorizon] = (uint16_t)(M.soils.size() - 1);
    M.graphDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_surface_properties(uint16_t idx, double roughness)   /* soilFluxes3D.cpp:457-467 */
{
    if (roughness < 0) return SF3D_PARAMETER_ERROR;
    if (idx >= M.roughness.size()) M.roughness.resize(idx + 1);
    M.roughness[idx] = roughness;
    M.graphDirty = true;
    return SF3D_OK;
}

sf3d_error_t sf3d_set_numerical_parameters(double minDt, double maxDt, uint16_t maxIter, uint16_t maxApprox,
                                           uint8_t resExp, uint8_t mbrExp)   /* soilFluxes3D.cpp:474-520 */
{
    if (minDt < 0.01) minDt = 0.01;
    if (minDt > 3600.) minDt = 3600.;
    if (maxDt < 60) maxDt = 60;
    if (maxDt > 3600.) maxDt = 3600.;
    if (maxDt < minDt) maxDt = minDt;
    if (maxIter < 20) maxIter = 20;
    if (maxIter > 1000) maxIter = 1000;
    if (maxApprox < 1) maxApprox = 1;
    if (maxApprox > 50) maxApprox = 50;
    if (resExp < 5) resExp = 5;
    if (resExp > 12) resExp = 12;
    if (mbrExp < 1) mbrExp = 1;
    if (mbrExp > 9) mbrExp = 9;
    if (!M.solverReady) return SF3D_MEMORY_ERROR;
    P.MBRThreshold = std::pow(10.0, -mbrExp);
    P.residualTolerance = std::pow(10.0, -resExp);
    P.dtMin = minDt; P.dtMax = maxDt; P.maxApprox = maxApprox; P.maxIter = maxIter;
    M.ctrlDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_hydraulic_properties(sf3d_wrc_t wrc, sf3d_mean_t mean, float ratio)   /* soilFluxes3D.cpp:531-548 */
{
    if ((ratio < 0.1) || (ratio > 100)) return SF3D_PARAMETER_ERROR;
    if (!M.solverReady) return SF3D_MEMORY_ERROR;
    P.wrc = wrc; P.meanType = mean; P.lvRatio = ratio;
    M.ctrlDirty = true;
    return SF3D_OK;
}

sf3d_error_t sf3d_set_culvert(uint32_t, double, double, double, double) { return SF3D_BOUNDARY_ERROR; }

sf3d_error_t sf3d_set_node_boundary(uint32_t i, sf3d_boundary_t bt, double slope, double area)   /* soilFluxes3D.cpp:689-725 */
{
    NEED_INIT_E; NEED_NODE_E(i);      /* the reference does not check; an out-of-range index there corrupts memory */
    M.btype[i] = bt;
    M.boundaryDirty = true;
    if (bt == SF3D_BND_NONE) return SF3D_OK;
    M.bslope[i] = slope; M.bsize[i] = area;
    if (M.water) {
        needFlows();
        M.bflowRate[i] = 0.; M.bflowSum[i] = 0.; M.prescribed[i] = SF3D_NODATA;
        M.flowSumsDirty = true;
    }
    if (M.heat) {                                                   /* soilFluxes3D.cpp:706-722 */
        M.bHeightWind[i] = M.bHeightT[i] = M.bRoughH[i] = M.bAero[i] = M.bSoilCond[i] = SF3D_NODATA;
        M.bT[i] = M.bRH[i] = M.bWind[i] = M.bNetIrr[i] = SF3D_NODATA;
        M.bRad[i] = M.bLat[i] = M.bSens[i] = M.bAdv[i] = 0.;
        M.bFixT[i] = M.bFixDepth[i] = SF3D_NODATA;
        M.heatBoundaryDirty = true;
    }
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node(uint32_t i, double x, double y, double z, double v, int isSurf, sf3d_boundary_t bt,
                           double slope, double barea)              /* soilFluxes3D.cpp:595-629 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    M.x[i] = x; M.y[i] = y; M.z[i] = z; M.size[i] = v;
    M.surf[i] = isSurf != 0;
    sf3d_set_node_boundary(i, bt, slope, barea);
    if (M.water) { M.pond[i] = isSurf ? 0.0001f : SF3D_NODATA; M.sink[i] = 0.; M.pondDirty = M.sinkDirty = true; M.sinkLo = 0; M.sinkHi = UINT32_MAX; }
    if (M.heat && !isSurf) {                                        /* :620-626: soil nodes start at 20 degrees C */
        needHeatState();
        M.temperature[i] = 273.15 + 20; M.heatSink[i] = 0.;
        M.heatStateDirty = M.heatSinkDirty = true;
    }
    M.graphDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_link(uint32_t i, uint32_t j, sf3d_link_t dir, double area)   /* soilFluxes3D.cpp:636-683 */
{
    NEED_INIT_E;
    if (i >= M.N || j >= M.N) return SF3D_INDEX_ERROR;
    int s;
    switch (dir) {
        case SF3D_LINK_UP: s = 0; break;
        case SF3D_LINK_DOWN: s = 1; break;
        case SF3D_LINK_LATERAL:
            if (M.nLat[i] == 8) return SF3D_TOPOGRAPHY_ERROR;
            s = 2 + M.nLat[i]; M.nLat[i]++;
            break;
        default: return SF3D_PARAMETER_ERROR;
    }
    M.ltype[s][i] = dir; M.lto[s][i] = j; M.larea[s][i] = area;
    if (M.water) { needFlows(); M.lflowSum[s][i] = 0.; M.flowSumsDirty = true; }
    M.graphDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_soil(uint32_t i, uint16_t soil, uint16_t horizon)   /* soilFluxes3D.cpp:734-750 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    if (M.surf[i]) return SF3D_INDEX_ERROR;
    if (soil >= soil1D.size() || horizon >= soil1D[soil].size()) return SF3D_PARAMETER_ERROR;
    /* the index table outlives re-initialisation like the reference's (soilFluxes3D.cpp:39 is never cleared): an entry left
     * from an earlier model points past the present soil list - the reference would store a dangling pointer there */
    if (soil1D[soil][horizon] >= M.soils.size()) return SF3D_PARAMETER_ERROR;
    M.cls[i] = soil1D[soil][horizon]; M.hasClass[i] = 1;
    M.graphDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_surface(uint32_t i, uint16_t surfaceIndex)   /* soilFluxes3D.cpp:758-775 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    if (surfaceIndex >= M.roughness.size()) return SF3D_PARAMETER_ERROR;
    if (!M.surf[i]) return SF3D_INDEX_ERROR;
    M.cls[i] = surfaceIndex; M.hasClass[i] = 1;
    M.graphDirty = true;
    return SF3D_OK;
}

sf3d_error_t sf3d_set_node_pond(uint32_t i, double pond)            /* soilFluxes3D.cpp:783-796 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    if (!M.surf[i]) return SF3D_INDEX_ERROR;
    M.pond[i] = pond; M.pondDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_water_content(uint32_t i, double wc)     /* soilFluxes3D.cpp:803-835 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    if (wc < 0.) return SF3D_PARAMETER_ERROR;
    if (!needState()) return SF3D_SOLVER_ERROR;
    if (M.surf[i]) { M.H[i] = M.z[i] + wc; M.Se[i] = 1.; M.K[i] = 0.; }
    else {
        if (wc > 1.) return SF3D_PARAMETER_ERROR;
        const SoilHost& s = M.soils[M.cls[i]];
        M.Se[i] = seFromTheta(s, wc);
        M.H[i] = M.z[i] - nodePsi(i);
        M.K[i] = nodeKHost(i);
    }
    M.stateDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_degree_of_saturation(uint32_t i, double se)   /* soilFluxes3D.cpp:842-862 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    if (M.surf[i]) return SF3D_INDEX_ERROR;
    if ((se < 0.) || (se > 1.)) return SF3D_PARAMETER_ERROR;
    if (!needState()) return SF3D_SOLVER_ERROR;
    M.Se[i] = se;
    M.H[i] = M.z[i] - nodePsi(i);
    M.K[i] = nodeKHost(i);
    M.stateDirty = true;
    return SF3D_OK;
}
static sf3d_error_t setH(uint32_t i, double H)                      /* soilFluxes3D.cpp:877-883, 899-905 */
{
    if (!needState()) return SF3D_SOLVER_ERROR;
    M.H[i] = H;
    if (M.surf[i]) { M.Se[i] = 1.; M.K[i] = SF3D_NODATA; }
    else { M.Se[i] = nodeSe(i); M.K[i] = nodeKHost(i); }
    M.stateDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_matric_potential(uint32_t i, double psi) { NEED_INIT_E; NEED_NODE_E(i); return setH(i, M.z[i] + psi); }
sf3d_error_t sf3d_set_node_total_potential(uint32_t i, double H) { NEED_INIT_E; NEED_NODE_E(i); return setH(i, H); }
sf3d_error_t sf3d_set_node_water_sink_source(uint32_t i, double q)  /* soilFluxes3D.cpp:934-945 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    M.sink[i] = q;
    if (!M.sinkDirty) { M.sinkLo = i; M.sinkHi = i + 1; M.sinkDirty = true; }
    else { if (i < M.sinkLo) M.sinkLo = i; if (i + 1 > M.sinkHi) M.sinkHi = i + 1; }
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_prescribed_total_potential(uint32_t i, double v)   /* soilFluxes3D.cpp:913-927 */
{
    NEED_INIT_E; NEED_NODE_E(i);
    if (M.btype[i] != SF3D_BND_PRESCRIBED_TOTAL_POTENTIAL) return SF3D_BOUNDARY_ERROR;
    M.prescribed[i] = v; M.boundaryDirty = true;
    return SF3D_OK;
}

/* ---- getters (soilFluxes3D.cpp:951-1277) ---- */
double sf3d_get_node_water_content(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); needState(); return M.surf[i] ? (M.H[i] - M.z[i]) : nodeTheta(i); }
double sf3d_get_node_maximum_water_content(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); if (M.surf[i]) return errValue(SF3D_INDEX_ERROR); return M.soils[M.cls[i]].thetaS; }
double sf3d_get_node_minimum_water_content(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); if (M.surf[i]) return errValue(SF3D_INDEX_ERROR); return M.soils[M.cls[i]].thetaR; }
double sf3d_get_node_available_water_content(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); needState(); return M.surf[i] ? (M.H[i] - M.z[i]) : std::max(0., nodeTheta(i) - thetaFromSignedPsi(i, -160)); }
double sf3d_get_node_water_deficit(uint32_t i, double fc)
{ NEED_INIT_D; NEED_NODE_D(i); if (M.surf[i]) return 0.; needState(); return thetaFromSignedPsi(i, -fc) - nodeTheta(i); }
double sf3d_get_node_degree_of_saturation(uint32_t i)
{
    NEED_INIT_D; NEED_NODE_D(i); needState();
    if (!M.surf[i]) return M.Se[i];
    const double cur = M.H[i] - M.z[i], mx = 0.001;
    return cur <= 0 ? 0 : (cur > mx ? 1. : cur / mx);
}
double sf3d_get_node_water_conductivity(uint32_t i) { NEED_INIT_D; NEED_NODE_D(i); needState(); return M.K[i]; }
double sf3d_get_node_matric_potential(uint32_t i) { NEED_INIT_D; NEED_NODE_D(i); needState(); return M.H[i] - M.z[i]; }
double sf3d_get_node_total_potential(uint32_t i) { NEED_INIT_D; NEED_NODE_D(i); needState(); return M.H[i]; }
double sf3d_get_node_pond(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); if (!M.surf[i]) return errValue(SF3D_INDEX_ERROR); return M.pond[i]; }
double sf3d_get_node_max_water_flow(uint32_t i, sf3d_link_t dir)
{
    NEED_INIT_D; NEED_NODE_D(i); needFlows();
    double mx = 0.;
    switch (dir) {
        case SF3D_LINK_UP: return M.lflowSum[0][i];
        case SF3D_LINK_DOWN: return M.lflowSum[1][i];
        case SF3D_LINK_LATERAL:
            for (int l = 0; l < M.nLat[i]; ++l) mx = std::max(mx, M.lflowSum[2 + l][i]);
            return mx;
        default: return errValue(SF3D_INDEX_ERROR);
    }
}
double sf3d_get_node_sum_lateral_water_flow(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); needFlows(); double s = 0.; for (int l = 0; l < M.nLat[i]; ++l) s += M.lflowSum[2 + l][i]; return s; }
double sf3d_get_node_sum_lateral_water_flow_in(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); needFlows(); double s = 0.; for (int l = 0; l < M.nLat[i]; ++l) if (M.lflowSum[2 + l][i] > 0) s += M.lflowSum[2 + l][i]; return s; }
double sf3d_get_node_sum_lateral_water_flow_out(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); needFlows(); double s = 0.; for (int l = 0; l < M.nLat[i]; ++l) if (M.lflowSum[2 + l][i] < 0) s += M.lflowSum[2 + l][i]; return s; }
double sf3d_get_node_boundary_water_flow(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); if (M.btype[i] == SF3D_BND_NONE) return errValue(SF3D_BOUNDARY_ERROR); needFlows(); return M.bflowSum[i]; }
double sf3d_get_total_boundary_water_flow(sf3d_boundary_t bt)       /* soilFluxes3D.cpp:1240-1250, index order */
{ needFlows(); double s = 0.0; for (uint32_t i = 0; i < M.N; ++i) if (M.btype[i] == bt) s += M.bflowSum[i]; return s; }
double sf3d_get_total_water_content(void)                           /* soilFluxes3D.cpp:1256-1259 */
{
    if (!M.initialized) return -1;
    double wc = 0.;
    if (dev().total_water_content(deviceModel(), P, &wc) != SF3D_OK) { fprintf(stderr, "sf3d: getTotalWaterContent: %s\n", dev().last_error()); return std::nan(""); }
    return wc;
}
double sf3d_get_water_storage(void) { return dev().ctrl().curStep.storage; }
double sf3d_get_water_mbr(void) { return wholePeriod.MBR; }

/* ---- heat setters / getters (soilFluxes3D.cpp:1283-1752).  Without isComputeHeat the reference's arrays do not
 * exist: MissingDataError.  Getters pull device results lazily. ---- */
#define HEAT_OFF_E if (!M.heat) return SF3D_MISSING_DATA_ERROR
#define HEAT_OFF_D if (!M.heat) return errValue(SF3D_MISSING_DATA_ERROR)
static bool needHeat()
{
    if (staleHeat() && dev().ready()) {
        HostModel& D = LM.on ? LM.L : M;
        if (dev().fetch_heat(D) != SF3D_OK) { fprintf(stderr, "sf3d: %s\n", dev().last_error()); return false; }
        if (LM.on) {
            scatterFrom(M.temperature, D.temperature); scatterFrom(M.bAero, D.bAero); scatterFrom(M.bSoilCond, D.bSoilCond);
            scatterFrom(M.bSens, D.bSens); scatterFrom(M.bLat, D.bLat); scatterFrom(M.bRad, D.bRad); scatterFrom(M.bAdv, D.bAdv);
        }
    }
    return true;
}
namespace { bool needHeatState() { return needHeat(); } }
sf3d_error_t sf3d_set_node_heat_sink_source(uint32_t i, double v) { NEED_INIT_E; NEED_NODE_E(i); HEAT_OFF_E; M.heatSink[i] = v; M.heatSinkDirty = true; return SF3D_OK; }
sf3d_error_t sf3d_set_node_temperature(uint32_t i, double v)
{ NEED_INIT_E; NEED_NODE_E(i); HEAT_OFF_E; needHeat(); M.temperature[i] = v; M.heatStateDirty = true; return SF3D_OK; }
sf3d_error_t sf3d_set_node_boundary_fixed_temperature(uint32_t i, double t, double depth)
{
    NEED_INIT_E; NEED_NODE_E(i); HEAT_OFF_E;
    if (M.btype[i] != SF3D_BND_PRESCRIBED_TOTAL_POTENTIAL && M.btype[i] != SF3D_BND_FREE_DRAINAGE) return SF3D_BOUNDARY_ERROR;
    M.bFixT[i] = t; M.bFixDepth[i] = depth; M.heatBoundaryDirty = true;
    return SF3D_OK;
}
#define HEAT_BND_SET(NAME, FIELD, CHECK) sf3d_error_t NAME(uint32_t i, double v) { NEED_INIT_E; NEED_NODE_E(i); HEAT_OFF_E; \
    if (M.btype[i] == SF3D_BND_NONE) return SF3D_BOUNDARY_ERROR; CHECK; M.FIELD[i] = v; M.heatBoundaryDirty = true; return SF3D_OK; }
HEAT_BND_SET(sf3d_set_node_boundary_height_wind, bHeightWind, )
HEAT_BND_SET(sf3d_set_node_boundary_height_temperature, bHeightT, )
HEAT_BND_SET(sf3d_set_node_boundary_net_irradiance, bNetIrr, )
HEAT_BND_SET(sf3d_set_node_boundary_temperature, bT, )
HEAT_BND_SET(sf3d_set_node_boundary_relative_humidity, bRH, )
HEAT_BND_SET(sf3d_set_node_boundary_roughness, bRoughH, if (v < 0) return SF3D_PARAMETER_ERROR)
HEAT_BND_SET(sf3d_set_node_boundary_wind_speed, bWind, if ((v < 0.) || (v > 1000.)) return SF3D_PARAMETER_ERROR)

double sf3d_get_node_temperature(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); if (!M.heat || M.surf[i]) return errValue(SF3D_TOPOGRAPHY_ERROR); if (!needHeat()) return std::nan(""); return M.temperature[i]; }
static double heatQuery(int what, uint32_t i, double h)
{
    double out = std::nan("");
    HostModel& D = deviceModel();
    if (LM.on) {                       /* node index in the strip-local numbering; only nodes of this rank's strip can be asked */
        if (i >= LM.g2l.size() || LM.g2l[i] < 0) { fprintf(stderr, "sf3d: heat query: node %u is not on this rank\n", i); return out; }
        i = (uint32_t)LM.g2l[i];
    }
    if (dev().heat_query(D, P, what, i, h, &out) != SF3D_OK) fprintf(stderr, "sf3d: heat query failed: %s\n", dev().last_error());
    return out;
}
double sf3d_get_node_heat_conductivity(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); if (!M.heat || M.surf[i]) return errValue(SF3D_TOPOGRAPHY_ERROR); return heatQuery(0, i, 0.); }
double sf3d_get_node_vapor(uint32_t i)
{ NEED_INIT_D; NEED_NODE_D(i); if (!M.water || !M.heat || !M.heatVapor) return errValue(SF3D_MISSING_DATA_ERROR); if (M.surf[i]) return errValue(SF3D_TOPOGRAPHY_ERROR); return heatQuery(1, i, 0.); }
double sf3d_get_node_heat_storage(uint32_t i, double h)
{ NEED_INIT_D; NEED_NODE_D(i); HEAT_OFF_D; if (M.surf[i]) return errValue(SF3D_TOPOGRAPHY_ERROR); return heatQuery(2, i, h); }
static double linkHeatFlux(int slot, uint32_t i, sf3d_flux_t t)                              /* getLinkHeatFlux, heat.cpp:639-658 */
{
    if (!M.heat) return SF3D_NODATA;
    if (M.heatSave == 1) { if (t != 0) return SF3D_NODATA; }
    else if (M.heatSave != 2 || t >= SF3D_FLUX_TYPES) return SF3D_NODATA;
    if (LM.on) {
        HostModel& D = deviceModel();
        if (i >= LM.g2l.size() || LM.g2l[i] < 0) return SF3D_NODATA;                       /* not on this rank */
        if (!D.lfluxValid[t]) {
            if (!dev().ready()) return (M.ltype[slot][i] != SF3D_LINK_NONE) ? SF3D_NODATA : 0.;
            if (dev().fetch_link_flux(D, t) != SF3D_OK) { fprintf(stderr, "sf3d: %s\n", dev().last_error()); return std::nan(""); }
        }
        return D.lfluxCache[t][(size_t)slot * D.N + (uint32_t)LM.g2l[i]];
    }
    if (!M.lfluxValid[t]) {
        if (!dev().ready()) return (M.ltype[slot][i] != SF3D_LINK_NONE) ? SF3D_NODATA : 0.;   /* setNodeLink initialisation */
        if (dev().fetch_link_flux(M, t) != SF3D_OK) { fprintf(stderr, "sf3d: %s\n", dev().last_error()); return std::nan(""); }
    }
    return M.lfluxCache[t][(size_t)slot * M.N + i];
}
double sf3d_get_node_heat_max_flux(uint32_t i, sf3d_link_t dir, sf3d_flux_t t)              /* soilFluxes3D.cpp:1580-1612 */
{
    NEED_INIT_D; NEED_NODE_D(i);
    if (!M.heat || M.surf[i]) return errValue(SF3D_TOPOGRAPHY_ERROR);
    switch (dir) {
        case SF3D_LINK_UP: return linkHeatFlux(0, i, t);
        case SF3D_LINK_DOWN: return linkHeatFlux(1, i, t);
        case SF3D_LINK_LATERAL: {
            double mx = 0.;
            for (int l = 0; l < 8; ++l) { const double f = linkHeatFlux(2 + l, i, t); if (f > std::fabs(mx)) mx = f; }
            return mx; }
        default: return errValue(SF3D_INDEX_ERROR);
    }
}
#define HEAT_BND_GET(NAME, FIELD, NEEDS_VAPOR) double NAME(uint32_t i) { NEED_INIT_D; NEED_NODE_D(i); \
    if (NEEDS_VAPOR ? (!M.water || !M.heat || !M.heatVapor) : !M.heat) return errValue(SF3D_MISSING_DATA_ERROR); \
    if (M.btype[i] != SF3D_BND_HEAT_SURFACE) return errValue(SF3D_BOUNDARY_ERROR); if (!needHeat()) return std::nan(""); return M.FIELD[i]; }
HEAT_BND_GET(sf3d_get_node_boundary_advective_flux, bAdv, true)
HEAT_BND_GET(sf3d_get_node_boundary_latent_flux, bLat, true)
HEAT_BND_GET(sf3d_get_node_boundary_radiative_flux, bRad, false)
HEAT_BND_GET(sf3d_get_node_boundary_sensible_flux, bSens, false)
HEAT_BND_GET(sf3d_get_node_boundary_aerodynamic_conductance, bAero, false)
HEAT_BND_GET(sf3d_get_node_boundary_soil_conductance, bSoilCond, false)
double sf3d_get_heat_mbr(void) { return heatWholePeriod.MBR; }
double sf3d_get_heat_mbe(void) { return heatWholePeriod.MBE; }

/* ---- computation ---- */
double sf3d_compute_step(double maxDt)                               /* soilFluxes3D.cpp:1785-1821 */
{
    if (!M.water && !M.heat) return std::min(maxDt, P.dtMax);
    if (!M.initialized || !M.solverReady) { fprintf(stderr, "sf3d: computeStep before initializeSF3D\n"); return std::nan(""); }
    if (dev().fatal()) { fprintf(stderr, "sf3d: computeStep refused after a fatal device failure: %s\n", dev().last_error()); return std::nan(""); }
    double dt = std::nan("");
    sf3d_error_t e = dev().step(deviceModel(), P, maxDt, &dt);
    if (e != SF3D_OK) {
        /* the reference's computeStep ignores run()'s error code (soilFluxes3D.cpp:1796) and returns the dt of a stepNan attempt;
         * so does this one - after saying so - unless the device itself failed (no dt, peer time-out, heat step not started) */
        fprintf(stderr, "sf3d: computeStep: %s\n", dev().last_error());
        if (!(dt == dt) || dev().fatal()) return std::nan("");
    }
    return dt;
}
void sf3d_compute_period(double period)                              /* soilFluxes3D.cpp:1760-1777, water.cpp:143-156 */
{
    if (!M.initialized) return;
    if (dev().ready()) { dev().ctrl().curPeriod.sinkSource = 0.; dev().ctrl().heatPeriodSink = 0.; dev().push_ctrl(); }
    double t = 0.;
    while (t < period) {
        const double dt = sf3d_compute_step(period - t);
        if (!(dt > 0.)) return;                                      /* device failure: do not spin */
        t += dt;
    }
    if (M.water) {
        const Ctrl& c = dev().ctrl();
        curPeriod.sinkSource = c.curPeriod.sinkSource;
        wholePeriod.sinkSource += curPeriod.sinkSource;
        const double dSp = c.curStep.storage - curPeriod.storage;
        const double dSh = c.curStep.storage - wholePeriod.storage;
        curPeriod.MBE = dSp - curPeriod.sinkSource;
        wholePeriod.MBE = dSh - wholePeriod.sinkSource;
        const double ref = std::max(0.001, wholePeriod.sinkSource);
        wholePeriod.MBR = wholePeriod.MBE / ref;
        curPeriod.storage = c.curStep.storage;
    }
    if (M.heat) {                                                   /* updateHeatBalanceDataWholePeriod, heat.cpp:400-413 */
        const Ctrl& c = dev().ctrl();
        heatCurPeriod.sinkSource = c.heatPeriodSink;
        heatWholePeriod.sinkSource += heatCurPeriod.sinkSource;
        const double dSp = c.heatCur.storage - heatCurPeriod.storage;
        const double dSh = c.heatCur.storage - heatWholePeriod.storage;
        heatCurPeriod.MBE = dSp - heatCurPeriod.sinkSource;
        heatWholePeriod.MBE = dSh - heatWholePeriod.sinkSource;
        const double ref = std::max(1., std::fabs(heatWholePeriod.sinkSource));
        heatWholePeriod.MBR = heatWholePeriod.MBE / ref;
        heatCurPeriod.storage = c.heatCur.storage;
    }
}

/* ---- extensions ---- */
sf3d_error_t sf3d_set_nodes(uint32_t first, uint32_t count, const double* x, const double* y, const double* z,
                            const double* v, const uint8_t* surf, const uint8_t* bt, const double* sl, const double* ba)
{
    for (uint32_t k = 0; k < count; ++k) {
        sf3d_error_t e = sf3d_set_node(first + k, x[k], y[k], z[k], v[k], surf[k], bt ? bt[k] : 0, sl ? sl[k] : 0., ba ? ba[k] : 0.);
        if (e != SF3D_OK) return e;
    }
    return SF3D_OK;
}
sf3d_error_t sf3d_set_node_links(uint64_t count, const uint32_t* node, const uint32_t* linked, const uint8_t* dir, const double* area)
{
    for (uint64_t k = 0; k < count; ++k) {
        sf3d_error_t e = sf3d_set_node_link(node[k], linked[k], dir[k], area[k]);
        if (e != SF3D_OK) return e;
    }
    return SF3D_OK;
}
#define BULK_SET(NAME, CALL, ...)                                                     \
    sf3d_error_t NAME(uint32_t first, uint32_t count, __VA_ARGS__)                     \
    { for (uint32_t k = 0; k < count; ++k) { if (skippedByTrim(first + k)) continue; sf3d_error_t e = CALL; if (e != SF3D_OK) return e; } return SF3D_OK; }
BULK_SET(sf3d_set_nodes_soil, sf3d_set_node_soil(first + k, s[k], h ? h[k] : 0), const uint16_t* s, const uint16_t* h)
BULK_SET(sf3d_set_nodes_surface, sf3d_set_node_surface(first + k, s[k]), const uint16_t* s)
BULK_SET(sf3d_set_nodes_pond, sf3d_set_node_pond(first + k, v[k]), const double* v)
BULK_SET(sf3d_set_nodes_matric_potential, sf3d_set_node_matric_potential(first + k, v[k]), const double* v)
BULK_SET(sf3d_set_nodes_total_potential, sf3d_set_node_total_potential(first + k, v[k]), const double* v)
BULK_SET(sf3d_set_nodes_water_sink_source, sf3d_set_node_water_sink_source(first + k, v[k]), const double* v)
#define BULK_GET(NAME, CALL)                                                          \
    sf3d_error_t NAME(uint32_t first, uint32_t count, double* out)                     \
    { for (uint32_t k = 0; k < count; ++k) out[k] = skippedByTrim(first + k) ? SF3D_NODATA : CALL(first + k); return SF3D_OK; }      /* (another rank's node: sf3d_dist_owner says whose) */
BULK_GET(sf3d_get_nodes_total_potential, sf3d_get_node_total_potential)
BULK_GET(sf3d_get_nodes_degree_of_saturation, sf3d_get_node_degree_of_saturation)
BULK_GET(sf3d_get_nodes_water_content, sf3d_get_node_water_content)
BULK_GET(sf3d_get_nodes_water_conductivity, sf3d_get_node_water_conductivity)
BULK_GET(sf3d_get_nodes_boundary_water_flow, sf3d_get_node_boundary_water_flow)
BULK_SET(sf3d_set_nodes_temperature, sf3d_set_node_temperature(first + k, v[k]), const double* v)
BULK_SET(sf3d_set_nodes_heat_sink_source, sf3d_set_node_heat_sink_source(first + k, v[k]), const double* v)
BULK_GET(sf3d_get_nodes_temperature, sf3d_get_node_temperature)
sf3d_error_t sf3d_set_nodes_boundary_heat(int field, uint32_t count, const uint32_t* nodes, const double* v)
{
    typedef sf3d_error_t (*setter_t)(uint32_t, double);
    static const setter_t setters[7] = {sf3d_set_node_boundary_height_wind, sf3d_set_node_boundary_height_temperature,
        sf3d_set_node_boundary_roughness, sf3d_set_node_boundary_temperature, sf3d_set_node_boundary_relative_humidity,
        sf3d_set_node_boundary_wind_speed, sf3d_set_node_boundary_net_irradiance};
    if (field < 0 || field > 6) return SF3D_PARAMETER_ERROR;
    for (uint32_t k = 0; k < count; ++k) { const sf3d_error_t e = setters[field](nodes[k], v[k]); if (e != SF3D_OK) return e; }
    return SF3D_OK;
}

sf3d_error_t sf3d_get_counters(uint64_t out[8])
{
    const Ctrl& c = dev().ctrl();
    for (int k = 0; k < 8; ++k) out[k] = dev().ready() ? c.counters[k] : 0;
    out[7] = dev().ready() ? c.earlyCourant : 0;      /* (slot 7 of the device block counts balance decisions: internal) */
    return SF3D_OK;
}
sf3d_error_t sf3d_get_sweep_launches(uint64_t* single, uint64_t* paired)
{
    if (!single || !paired) return SF3D_PARAMETER_ERROR;
    const Ctrl& c = dev().ctrl();
    *single = dev().ready() ? c.singleLaunches : 0; *paired = dev().ready() ? c.pairLaunches : 0;
    return SF3D_OK;
}
sf3d_error_t sf3d_get_resident_launches(uint64_t* loops)
{
    if (!loops) return SF3D_PARAMETER_ERROR;
    *loops = dev().ready() ? dev().ctrl().residentLaunches : 0;
    return SF3D_OK;
}
sf3d_error_t sf3d_get_heat_counters(uint64_t out[4])
{
    if (!out) return SF3D_PARAMETER_ERROR;
    const Ctrl& c = dev().ctrl();
    for (int k = 0; k < 4; ++k) out[k] = dev().ready() ? c.heatCounters[k] : 0;
    return SF3D_OK;
}
double sf3d_get_linear_residual(void) { return dev().ready() ? dev().ctrl().lastNorm : -9999.; }
double sf3d_get_time_step(void) { return P.dtCurr; }
sf3d_error_t sf3d_set_time_step(double dt)
{
    if (!M.solverReady) return SF3D_MEMORY_ERROR;
    if (!(dt > 0.)) return SF3D_PARAMETER_ERROR;
    P.dtCurr = std::min(std::max(dt, P.dtMin), P.dtMax);
    M.ctrlDirty = true;
    return SF3D_OK;
}
sf3d_error_t sf3d_reset_solver_state(void) { P = ParamsHost(); M.ctrlDirty = true; return SF3D_OK; }
sf3d_error_t sf3d_set_surface_nodes_number(uint32_t ns)
{
    if (!M.initialized) return SF3D_MEMORY_ERROR;
    if (ns > M.N) return SF3D_INDEX_ERROR;
    if (ns != M.ns) { M.ns = ns; M.graphDirty = true; }
    return SF3D_OK;
}
sf3d_error_t sf3d_set_device(int d)
{
    sf3d_error_t e = dev().set_device(d);
    if (e != SF3D_OK) fprintf(stderr, "sf3d: set_device: %s\n", dev().last_error());
    return e;
}
sf3d_error_t sf3d_synchronize(void)
{
    /* push every pending host edit (sinks, ponds, state, parameters) to the device, then drain the stream */
    if (M.initialized && M.solverReady) {
        sf3d_error_t e = dev().sync_to_device(deviceModel(), P);
        if (e != SF3D_OK) { fprintf(stderr, "sf3d: synchronize: %s\n", dev().last_error()); return e; }
    }
    return dev().synchronize();
}
/* ---- multi-GPU ---- */
int sf3d_dist_blob_bytes(void) { return (int)sizeof(DistBlob); }
sf3d_error_t sf3d_dist_prepare(int rank, int world)
{
    /* a connected, trimmed strip keeps only its own part of the graph: another (rank, world) needs the whole model again */
    if (LM.trimmed && M.initialized && (rank != distRank || world != distWorld)) {
        fprintf(stderr, "sf3d: dist_prepare(%d, %d) on a connected strip of (%d, %d): the staging copy holds this strip only - sf3d_clean / re-initialise first\n", rank, world, distRank, distWorld);
        return SF3D_TOPOGRAPHY_ERROR;
    }
    sf3d_error_t e = dev().dist_prepare(rank, world);
    if (e != SF3D_OK) fprintf(stderr, "sf3d: %s\n", dev().last_error());
    else {
        const char* le = getenv("SF3D_DIST_LOCAL");
        distRank = rank; distWorld = world;
        LM.on = world > 1 && !(le && le[0] == '0');      /* strip-local device models (SF3D_DIST_LOCAL=0: every rank uploads the global model) */
        LM.built = false;
    }
    return e;
}
sf3d_error_t sf3d_dist_export(void* blob)
{
    if (!M.initialized) return SF3D_MEMORY_ERROR;
    sf3d_error_t e = dev().dist_export(deviceModel(), P, static_cast<DistBlob*>(blob));
    if (e != SF3D_OK) fprintf(stderr, "sf3d: dist_export: %s\n", dev().last_error());
    return e;
}
sf3d_error_t sf3d_dist_connect(const void* blobs)
{
    sf3d_error_t e = dev().dist_connect(static_cast<const DistBlob*>(blobs));
    if (e != SF3D_OK) fprintf(stderr, "sf3d: dist_connect: %s\n", dev().last_error());
    return e;
}
int sf3d_dist_status(void) { return dev().dist_status(); }
sf3d_error_t sf3d_dist_finalize(int mode)
{
    if (mode < 0 || mode > 2) return SF3D_PARAMETER_ERROR;
    sf3d_error_t e = dev().dist_finalize(mode);
    if (e != SF3D_OK) fprintf(stderr, "sf3d: dist_finalize: %s\n", dev().last_error());
    else trimHostStaging();          /* connected: the topology is frozen, M is needed at this rank's nodes only */
    return e;
}
int sf3d_dist_transport(void) { return dev().dist_transport(); }
sf3d_error_t sf3d_dist_stats(double* out, int capacity) { return out ? dev().dist_stats(out, capacity) : SF3D_PARAMETER_ERROR; }
uint64_t sf3d_host_bytes(void)
{
    uint64_t n = 0;
    forEachArray(M, [&](auto& v) { n += residentBytes(v); });
    if (LM.on) forEachArray(LM.L, [&](auto& v) { n += residentBytes(v); });
    return n;
}
/* partition queries are host logic: they work without a device */
sf3d_error_t sf3d_dist_owner(int world, uint32_t first, uint32_t count, int32_t* out)
{
    if (!M.initialized) return SF3D_MEMORY_ERROR;
    if ((uint64_t)first + count > M.N) return SF3D_INDEX_ERROR;
    auto rankOf = [](uint8_t o) -> int32_t { return o == SF3D_OWNER_NONE ? -1 : (int32_t)o; };      /* -1: a node this rank never staged (strip-local build) */
    if (LM.trimmed && world == distWorld && LM.gpart.owner.size() == M.N) {      /* (M no longer holds the other ranks' links) */
        for (uint32_t k = 0; k < count; ++k) out[k] = rankOf(LM.gpart.owner[first + k]);
        return SF3D_OK;
    }
    /* the trimmed staging copy no longer holds the other ranks' links: a partition for another world cannot be derived from it
     * (it would silently hand every foreign soil node to rank 0) */
    if (LM.trimmed) return SF3D_MISSING_DATA_ERROR;
    Partition part;
    sf3d_error_t e = sf3d_compute_partition(M, LM.on ? distRank : 0, world, part);
    if (e != SF3D_OK) return e;
    for (uint32_t k = 0; k < count; ++k) out[k] = rankOf(part.owner[first + k]);
    return SF3D_OK;
}
sf3d_error_t sf3d_dist_bounds(uint32_t nrSurfaceNodes, int world, uint32_t* bounds) { return sf3d_partition_bounds(nrSurfaceNodes, world, bounds); }
/* Is the node graph a regular NX x NY x NZ grid in the layer-major numbering i = (l NY + r) NX + c with the ten-link stencil
 * (up, down, up to eight laterals to the 8-neighbourhood of the same layer)?  Host logic over the staged links; groundwork for
 * the two-iterations-per-pass sweep, which needs exactly this structure.  dr/dc: row / column step of lateral
 * slot k = 0..7 at a node that has all eight; nodes on the grid's edge have fewer laterals and fill their slots in their own
 * order (setNodeLink puts the k-th lateral of a node into slot 2 + k), so a kernel decodes their steps from the link targets. */
sf3d_error_t sf3d_get_regular_grid(uint32_t* nx, uint32_t* ny, uint32_t* nz, int8_t* dr, int8_t* dc)
{
    if (!M.initialized) return SF3D_MEMORY_ERROR;
    if (!nx || !ny || !nz || !dr || !dc) return SF3D_PARAMETER_ERROR;
    const uint64_t N = M.N, ns = M.ns;
    if (ns == 0 || N % ns != 0) return SF3D_MISSING_DATA_ERROR;
    const uint64_t NZ = N / ns;
    int64_t maxOff = 0;                                    /* NX + 1: the diagonal step */
    uint64_t full = N;                                     /* first node with eight laterals */
    for (uint64_t i = 0; i < N; ++i) {
        int cnt = 0;
        for (int k = 0; k < 8; ++k) {
            if (M.ltype[2 + k][i] == SF3D_LINK_NONE) continue;
            ++cnt;
            const int64_t o = llabs((int64_t)M.lto[2 + k][i] - (int64_t)i);
            if (o > maxOff) maxOff = o;
        }
        if (cnt == 8 && full == N) full = i;
    }
    const int64_t NX = maxOff - 1;
    if (full == N || NX < 2 || ns % (uint64_t)NX != 0) return SF3D_MISSING_DATA_ERROR;
    const int64_t NY = (int64_t)(ns / (uint64_t)NX);
    if (NY < 2) return SF3D_MISSING_DATA_ERROR;
    auto decode = [&](int64_t off, int64_t& rr, int64_t& cc) {
        rr = (off >= 0) ? (off + NX / 2) / NX : -((-off + NX / 2) / NX);
        cc = off - rr * NX;
        return rr >= -1 && rr <= 1 && cc >= -1 && cc <= 1 && !(rr == 0 && cc == 0);
    };
    for (uint64_t i = 0; i < N; ++i) {
        const int64_t l = (int64_t)(i / ns), r = (int64_t)((i % ns) / (uint64_t)NX), c = (int64_t)(i % (uint64_t)NX);
        if (M.ltype[0][i] != SF3D_LINK_NONE && (l == 0 || M.lto[0][i] != i - ns)) return SF3D_MISSING_DATA_ERROR;
        if (M.ltype[1][i] != SF3D_LINK_NONE && (l == (int64_t)NZ - 1 || M.lto[1][i] != i + ns)) return SF3D_MISSING_DATA_ERROR;
        unsigned seen = 0;
        for (int k = 0; k < 8; ++k) {
            if (M.ltype[2 + k][i] == SF3D_LINK_NONE) continue;
            int64_t rr, cc;
            if (!decode((int64_t)M.lto[2 + k][i] - (int64_t)i, rr, cc)) return SF3D_MISSING_DATA_ERROR;
            if (r + rr < 0 || r + rr >= NY || c + cc < 0 || c + cc >= NX) return SF3D_MISSING_DATA_ERROR;
            const unsigned bit = 1u << ((rr + 1) * 3 + (cc + 1));
            if (seen & bit) return SF3D_MISSING_DATA_ERROR;              /* two links to the same neighbour */
            seen |= bit;
        }
    }
    *nx = (uint32_t)NX; *ny = (uint32_t)NY; *nz = (uint32_t)NZ;
    for (int k = 0; k < 8; ++k) {
        int64_t rr, cc;
        decode((int64_t)M.lto[2 + k][full] - (int64_t)full, rr, cc);
        dr[k] = (int8_t)rr; dc[k] = (int8_t)cc;
    }
    return SF3D_OK;
}
sf3d_error_t sf3d_dist_halo(int rank, int world, int peer, int direction, uint32_t capacity, uint32_t* out, uint32_t* count)
{
    if (!M.initialized) return SF3D_MEMORY_ERROR;
    if (peer < 0 || peer >= world) return SF3D_PARAMETER_ERROR;
    Partition part;
    const bool cached = LM.trimmed && world == distWorld && rank == distRank && (int)LM.gpart.send.size() == world;      /* (M no longer holds the other ranks' links) */
    if (!cached) {
        if (LM.trimmed) return SF3D_MISSING_DATA_ERROR;          /* another rank's lists cannot be derived from a trimmed staging copy */
        sf3d_error_t e = sf3d_compute_partition(M, rank, world, part);
        if (e != SF3D_OK) return e;
    }
    const Partition& pp = cached ? LM.gpart : part;
    const std::vector<uint32_t>& l = direction == 0 ? pp.send[peer] : pp.recv[peer];
    *count = (uint32_t)l.size();
    if (out) { if (capacity < l.size()) return SF3D_MEMORY_ERROR; std::copy(l.begin(), l.end(), out); }
    return SF3D_OK;
}

uint64_t sf3d_device_bytes(void) { return dev().device_bytes(); }
sf3d_error_t sf3d_kernel_timing(int mode) { return dev().timing(mode); }
int sf3d_kernel_count(void) { return KID_COUNT; }
const char* sf3d_kernel_name(int k) { return DeviceSolver::kernel_name(k); }
sf3d_error_t sf3d_kernel_stats(int k, uint64_t* n, double* ms, uint64_t* nodes) { return dev().stats(k, n, ms, nodes); }
sf3d_error_t sf3d_device_log(uint32_t count, const double* x, double* out) { return (count && (!x || !out)) ? SF3D_PARAMETER_ERROR : dev().device_log(count, x, out); }
sf3d_error_t sf3d_device_exp(uint32_t count, const double* x, double* out) { return (count && (!x || !out)) ? SF3D_PARAMETER_ERROR : dev().device_log(count, x, out, 1); }
sf3d_error_t sf3d_device_cbrt(uint32_t count, const double* x, double* out) { return (count && (!x || !out)) ? SF3D_PARAMETER_ERROR : dev().device_log(count, x, out, 2); }
sf3d_error_t sf3d_device_norm_sum(uint32_t count, const double* x, uint32_t blocks, int association, double* out) { return ((count && !x) || !out) ? SF3D_PARAMETER_ERROR : dev().device_norm_sum(count, x, blocks, association, out); }
sf3d_error_t sf3d_device_pow(uint32_t count, const double* x, const double* y, double* out) { return (count && (!x || !y || !out)) ? SF3D_PARAMETER_ERROR : dev().device_pow(count, x, y, out); }

} /* extern "C" */
