/*
 * sf3d_model.h - host-side staging model of the MI355X soilFluxes3D library and the interface
 * of the device solver.  Setters of the C ABI only touch this staging copy (O(1) host work per
 * call, SURVEY.md 7 "chatty API"); the device is brought up to date lazily at the next
 * computeStep / initializeBalance, and getters pull a field back only when the device holds a
 * newer version of it.
 */
#ifndef SF3D_MODEL_H
#define SF3D_MODEL_H

#include <cstdint>
#include <vector>

#include "sf3d.h"
#include "sf3d_device.h"

struct SoilHost {                      /* soilData_t, types.h:104-121 */
    uint16_t soilNumber; uint8_t horizonNumber;
    double alpha, n, m, he, Sc, thetaS, thetaR, Ksat, L, organicMatter, clay;
    double mualemDen;                  /* 1 - (1 - Sc^(1/m))^m, soilPhysics.cpp:201-204 */
};

struct ParamsHost {                    /* SolverParameters, types.h:291-315 (persist across re-init) */
    double MBRThreshold = 1e-3, residualTolerance = 1e-10;
    double dtMin = 1, dtMax = 600, dtCurr = SF3D_NODATA;
    uint16_t maxApprox = 10, maxIter = 200;
    uint8_t wrc = SF3D_WRC_MODIFIED_VAN_GENUCHTEN, meanType = SF3D_MEAN_LOGARITHMIC;
    double lvRatio = 4., courantThreshold = 0.5, instabilityFactor = 10.;
    double heatWeightFactor = 0.5;     /* types.h:307 */
    uint32_t numThreads = 1;
    bool lineal = false;               /* setUseLineal(true) AND SF3D_LINEAL_DEVICE_CG=1: device conjugate gradients instead of Jacobi sweeps */
};

struct Partition;
struct HostModel {
    bool initialized = false, solverReady = false;
    bool water = true, heat = false, solutes = false;
    bool cgArrays = false;           /* allocate the conjugate-gradient work vectors (SF3D_LINEAL_DEVICE_CG=1 at sf3d_initialize) */
    bool compat = false;             /* SF3D_COMPAT_STALE_LINK_FLOW=1 at sf3d_initialize: reproduce quirk 1's stale-slot reads */
    uint32_t N = 0, ns = 0;
    std::vector<double> x, y, z, size;
    std::vector<uint8_t> surf, hasClass;
    std::vector<uint16_t> cls;
    std::vector<uint8_t> btype;
    std::vector<double> bslope, bsize, bflowRate, bflowSum, prescribed;
    std::vector<uint8_t> nLat;
    std::vector<uint8_t> ltype[SF3D_SLOTS];
    std::vector<uint32_t> lto[SF3D_SLOTS];
    std::vector<double> larea[SF3D_SLOTS], lflowSum[SF3D_SLOTS];
    std::vector<double> Se, K, H, sink, pond;
    std::vector<SoilHost> soils;
    std::vector<double> roughness;
    /* coupled heat transport (heat.cpp): flags, state, atmosphere / fixed-temperature boundary inputs, boundary outputs */
    bool heatVapor = false, heatAdvection = false; uint8_t heatSave = 0;
    std::vector<double> temperature, heatSink;
    std::vector<double> bHeightWind, bHeightT, bRoughH, bT, bRH, bWind, bNetIrr, bFixT, bFixDepth;
    std::vector<double> bAero, bSoilCond, bSens, bLat, bRad, bAdv;
    std::vector<double> lfluxCache[SF3D_FLUX_TYPES]; bool lfluxValid[SF3D_FLUX_TYPES] = {false};   /* [10][N], fetched on demand */
    bool heatStateDirty = true, heatSinkDirty = true, heatBoundaryDirty = true;
    bool hostStaleHeat = false;      /* temperature and boundary outputs are newer on the device */

    /* what the device lacks */
    bool graphDirty = true;      /* topology / classes / boundary geometry: rebuild + full upload */
    bool stateDirty = true;      /* H (and Se, K) set through the API                             */
    bool sinkDirty = true, pondDirty = true, boundaryDirty = true, flowSumsDirty = true;
    bool ctrlDirty = true;       /* parameters or balances edited on the host                     */
    uint32_t sinkLo = 0, sinkHi = UINT32_MAX;   /* node range [lo, hi) touched since the last sink upload (hourly sinks usually cover the
                                                  * surface nodes only: 2 MB instead of 42 MB at 512 x 512 x 20) */
    /* multi-GPU, strip-local models (sf3d_api.cpp LocalModel): this model holds only the nodes a rank's rows touch (owned + one-cell
     * halo) in LOCAL numbering; the partition comes ready-made in local indices and the mean norm of the Jacobi sweep divides by the
     * GLOBAL node count */
    const struct Partition* presetPartition = nullptr;
    uint32_t globalN = 0;                       /* 0: this IS the global model */
    /* what the host lacks (device is newer) */
    bool hostStaleState = false; /* H, Se, K                                                      */
    bool hostStaleFlows = false; /* bflowRate, bflowSum, lflowSum                                 */
};

/* The two-colour heat sweep (sf3d_heat.inc: odd layers from the old iterate, then even layers with their Up / Down neighbours taken
 * from the values the first half has just written) is a valid, deterministic Gauss-Seidel ordering only if "layer parity" colours the
 * vertical links: layer = hops to the surface through the Up links, defined where the node above has the SMALLER index (layer-major
 * numbering), and every Up / Down link must join nodes of opposite parity.  A numbering the API accepts but this does not hold for
 * (bottom-up columns, a Down link whose target's Up is another node) would make the second half read same-colour values that the
 * same launch is writing - a race (round 4's advice).  Such graphs keep the Jacobi sweep. */
inline bool heat_two_colour_valid(const HostModel& m)
{
    std::vector<uint8_t> lp(m.N, 0);
    for (uint32_t i = m.ns; i < m.N; ++i)
        if (m.ltype[0][i] != SF3D_LINK_NONE) {
            if (m.lto[0][i] >= i) return false;
            lp[i] = (uint8_t)(lp[m.lto[0][i]] + 1);
        }
    for (uint32_t i = m.ns; i < m.N; ++i)
        for (int s = 0; s < 2; ++s)
            if (m.ltype[s][i] != SF3D_LINK_NONE && ((lp[i] ^ lp[m.lto[s][i]]) & 1u) == 0u) return false;
    return true;
}

/* Row-strip partition of the node graph for `world` ranks (host logic, no device needed):
 * a node belongs to the rank that owns its surface-cell column (the surface node reached by
 * walking Up links); surface nodes [0, ns) are cut into `world` contiguous index ranges at
 * multiples of 64.  send[p] / recv[p] are the sorted unique node lists exchanged with rank p
 * (what p's rows read from my strip / what my rows read from p's strip). */
#define SF3D_OWNER_NONE 255u       /* Partition::owner of a node that was never staged on this rank (strip-local build) */
struct Partition {
    int world = 1, rank = 0;
    std::vector<uint8_t> owner;                       /* [N]; SF3D_OWNER_NONE: absent */
    uint32_t missingFrom = UINT32_MAX, missingTo = UINT32_MAX;   /* first link from one of this rank's nodes to an absent node (SF3D_MISSING_DATA_ERROR) */
    std::vector<uint32_t> bounds;                     /* [world + 1] surface-index strip bounds */
    std::vector<std::vector<uint32_t>> send, recv;    /* [world] */
};
sf3d_error_t sf3d_compute_partition(const HostModel& m, int rank, int world, Partition& out);
sf3d_error_t sf3d_partition_bounds(uint32_t ns, int world, uint32_t* bounds);      /* [world + 1] surface-index strip bounds: a function of ns and world alone */

/* what each rank publishes to the others before the first step (sf3d_dist_export/connect) */
struct DistBlob {
    unsigned char ipcHandle[64];
    uint64_t recvOff[SF3D_MAX_RANKS];
    uint32_t recvCount[SF3D_MAX_RANKS];
    uint32_t world, rank;
    uint64_t nodes;
    uint32_t generation;            /* export count of the rank; rank 0's value stamps the self-check token of a connect */
    uint32_t heatJacobiOnly;        /* 1: this rank's part of the graph does not admit the two-colour heat sweep (heat_two_colour_valid): every rank then sweeps by Jacobi */
    unsigned char ncclId[128];     /* rank 0: ncclUniqueId of the run's RCCL communicator (all zero when RCCL is not available) */
    char pciBusId[32];             /* which physical GPU the rank runs on (start-up self-check: distinct, peer-reachable devices) */
    char shmName[48];              /* POSIX shared-memory object holding a second copy of the rank's window in HOST memory: the fall-back
                                    * transport when the device windows cannot be opened or do not carry device-initiated stores */
    uint64_t windowBytes;
    uint32_t resBlocks, resCapacity; /* resident sweep loop (sf3d_resident.inc): blocks of this rank's grid (0: its strip does not fit, or the loop is off) and how many such
                                      * blocks its GPU holds at once - the loop runs on every rank or on none (its edge rows hand over tagged records, the single
                                      * sweeps plain values), and ranks that share a GPU must fit on it TOGETHER (persistent kernels that have to take turns
                                      * wait for each other's time slices) */
    uint32_t pairRecords, pairPad;   /* paired pass with record hand-over (sf3d_pair.inc, DIST): nonzero = this rank runs the paired pass and hands its edge rows over as
                                      * records - used only if EVERY rank says so (a rank on single sweeps puts plain values and needs two exchanges per two iterations) */
};

/* The device half.  All methods return an sf3d_error_t; HIP failures map to SF3D_SOLVER_ERROR
 * and leave a message retrievable with last_error(). */
class DeviceSolver {
public:
    static DeviceSolver& instance();
    sf3d_error_t set_device(int dev);
    /* bring the device up to date with every dirty part of `m` (allocates on first use) */
    sf3d_error_t sync_to_device(HostModel& m, const ParamsHost& p);
    /* one computeStep: returns the accepted dt through *dt and refreshes the Ctrl mirror */
    sf3d_error_t step(HostModel& m, ParamsHost& p, double maxTimeStep, double* dt);
    /* storage = sum(theta * size) over the device state (water.cpp:71-90) */
    sf3d_error_t total_water_content(HostModel& m, const ParamsHost& p, double* out);
    sf3d_error_t fetch_state(HostModel& m);      /* H, Se, K   -> host */
    sf3d_error_t fetch_flows(HostModel& m);      /* flow sums  -> host */
    /* heat */
    sf3d_error_t fetch_heat(HostModel& m);       /* temperature, boundary fluxes and conductances -> host */
    sf3d_error_t fetch_link_flux(HostModel& m, int type);   /* one flux type [10][N] -> m.lfluxCache[type] */
    sf3d_error_t heat_query(HostModel& m, const ParamsHost& p, int what, uint32_t node, double h, double* out);
    sf3d_error_t heat_storage(HostModel& m, const ParamsHost& p, double* out);   /* computeCurrentHeatStorage() */
    sf3d_error_t release();
    sf3d_error_t synchronize();
    Ctrl& ctrl() { return mirror_; }
    void push_ctrl() { ctrlEdited_ = true; }
    bool ready() const { return built_; }
    bool connected() const { return world_ > 1 && connected_; }   /* a multi-rank model whose windows are set up */
    const char* last_error() const { return err_; }
    /* true after a failure that leaves the device state unusable (peer time-out of the multi-GPU exchange, a heat step that
     * did not start, a HIP error): computeStep then refuses to go on instead of stalling once more per call */
    bool fatal() const { return fatal_; }
    /* multi-GPU */
    sf3d_error_t dist_prepare(int rank, int world);
    sf3d_error_t dist_export(HostModel& m, const ParamsHost& p, DistBlob* out);
    sf3d_error_t dist_connect(const DistBlob* all);
    int dist_status() const { return distStatus_; }
    int dist_transport() const;
    sf3d_error_t dist_stats(double* out, int capacity);
    void print_hops() const;
    sf3d_error_t dist_finalize(int mode);      /* 0 device windows (all ranks passed), 1 RCCL (opt-in), 2 host-memory windows (fall-back) */
    int world() const { return world_; }
    int rank() const { return rank_; }
    /* instrumentation */
    sf3d_error_t timing(int mode);               /* 0 off, 1 all node kernels, 2 only the Jacobi sweep */
    sf3d_error_t stats(int kid, uint64_t* launches, double* ms, uint64_t* nodes);
    static const char* kernel_name(int kid);
    sf3d_error_t device_log(uint32_t n, const double* x, double* out, int which = 0);   /* test hook: the kernels' log (0) / exp (1) / cbrt (2) */
    sf3d_error_t device_pow(uint32_t n, const double* x, const double* y, double* out);   /* test hook: the property kernels' pow */
    sf3d_error_t device_norm_sum(uint32_t n, const double* x, uint32_t blocks, int assoc, double* out);   /* test hook: the sweep kernels' double-double norm sum */
    uint64_t device_bytes() const;               /* bytes of device memory the model's arrays take (sum of the allocations) */

private:
    DeviceSolver() = default;
    struct Impl;
    Impl* impl_ = nullptr;
    Ctrl mirror_{};
    bool built_ = false, ctrlEdited_ = false;
    int world_ = 1, rank_ = 0;
    bool connected_ = false;
    bool fatal_ = false;
    int distStatus_ = 0;             /* after dist_connect: 0 the windows passed their self-check, 1 they did not / RCCL was asked for */
    char distWhy_[200] = {0};
    char err_[256] = {0};
    friend struct Impl;
};

#endif
