/*
 * sf3d_device.h - structures shared by the host model (sf3d_api.cpp) and the HIP solver
 * (sf3d_solver.hip) of the MI355X-native soilFluxes3D library.
 *
 * Data layout in HBM (all fp64 unless noted, node arrays in caller order: layer-major,
 * surface nodes first - the ordering the reference relies on):
 *   static graph   lto[10][N] u32, lkind[10][N] u8, larea[10][N], ldist[10][N]  (slot-major:
 *                  slot 0 Up, 1 Down, 2..9 laterals; consecutive threads read consecutive
 *                  nodes of one slot => coalesced), z, size, pond, cls u16, btype u8,
 *                  bslope, bsize, prescribed; per (64-node chunk, slot) descriptors ckind/cdelta
 *                  that replace lto/lkind reads wherever the numbering is locally regular
 *   soil classes   SoilDev[] (gathered by cls, L2/scalar-cache resident)
 *   state          X[5][N]: pool of head buffers; H, Hold and Hbest are INDICES into the
 *                  pool kept in Ctrl (no copies on step begin / reject / keep-best / restore)
 *                  Se, K, C, flow, sink, bflowRate, bflowSum, lflowSum[10][N]
 *   linear system  A2[5][N] row-normalised off-diagonals (static ELL, zeros kept; slots paired so
 *                  every access is 16 B per lane), b
 *   control block  Ctrl: solver parameters, adaptive dt, stage of the step state machine,
 *                  balances, counters - decisions are taken on the device, by the block of the producing
 *                  kernel that arrives last (or by 1-block kernels with SF3D_FUSED_DECIDE=0)
 *   heat           HeatDev (coupled heat transport, sf3d_heat.inc): temperature pool TX[3][N], heat system
 *                  hA2[5][N]/hD/hb, per-node conductivities, boundary and link-flux arrays
 */
#ifndef SF3D_DEVICE_H
#define SF3D_DEVICE_H

#include <stdint.h>

#define SF3D_SLOTS 10
#define SF3D_POOL 5              /* H, Hold, Hbest + two free buffers: the paired sweep writes x' and x'' in one pass */
#define SF3D_BLOCK 256
#define SF3D_CHUNK 64            /* one wave64 = 64 consecutive nodes */
#define SF3D_MAX_BLOCKS 2048     /* 8 blocks of 256 threads per CU x 256 CUs */
#define CK_NONE 0                /* chunk descriptor: no node of the chunk has this link slot        */
#define CK_MIXED 255             /* chunk descriptor: per-node lkind/lto must be read                 */

/* link kinds (derived once per graph change from the two end nodes, water.cpp:308-324) */
enum : uint8_t { LK_NONE = 0, LK_SOIL_VERT = 1, LK_SOIL_LAT = 2, LK_RUNOFF = 3, LK_INFILTRATION = 4 };

/* stages of the device-resident step state machine (cpusolver.cpp:143-190, 392-468) */
enum : uint32_t {
    ST_IDLE = 0,        /* between computeStep calls                                         */
    ST_ATTEMPT = 1,     /* (unused: an attempt begins inline in k_step_begin / on rejection)   */
    ST_APPROX = 2,      /* properties + boundary + assembly of approximation `approx`        */
    ST_SWEEP = 3,       /* Jacobi sweeps running                                             */
    ST_POST = 4,        /* H = x, Se, balance sums                                           */
    ST_RESTORE = 5,     /* restoreBestStep: H = Hbest, Se, K, boundary, balance              */
    ST_ACCEPT = 6,      /* step accepted: bookkeeping done, k_accept adds the flow sums        */
    ST_DONE = 7,        /* (unused)                                                          */
    ST_FAIL = 8         /* stepNan                                                           */
};

/* 16-byte pair; plain struct so host code can use it too */
struct alignas(16) sf3d_d2 { double x, y; };
struct alignas(8) sf3d_f2 { float x, y; };

struct SoilDev {        /* soilData_t (types.h:104-121) + per-soil constants */
    double alpha, n, m, he, Sc, invSc, thetaS, thetaR, Ksat, L, invM, mualemDen;
    double clay, organicMatter;   /* heat only */
};

/* per-chunk link descriptor (240 bytes, fetched through the scalar path once per chunk) */
struct ChunkDesc {
    int32_t delta[SF3D_SLOTS];          /* j - i when kind[s] is a uniform link kind, else 0 */
    uint8_t kind[SF3D_SLOTS];           /* CK_NONE | LK_* (uniform) | CK_MIXED */
    uint8_t rowType;                    /* 0 all surface nodes, 1 all soil nodes, 2 straddles nrSurfaceNodes */
    uint8_t pad0;                       /* multi GPU: 1 = some node of the chunk has a neighbour owned by another rank */
    uint16_t areaUniform;               /* bit s: every link of slot s in the chunk has interface area area[s] */
    uint16_t sweepUniform;              /* bit s: kind[s] is CK_MIXED only because some nodes lack the link - every existing one has
                                         * j = i + delta[s] (row ends of a regular grid, whose laterals the upload aligns with their chunk's
                                         * slots): neither the sweep nor the assembly reads lto there */
    uint16_t distUniform;               /* bit s: every link of slot s in the chunk has the link distance dist[s] */
    uint8_t soilUniform;                /* 1: a soil-only chunk whose slots are all either empty or one uniform soil-soil link kind with uniform offset,
                                         * area and distance (interior of a regular grid below layer 1): k_assemble_uniform's scalar-geometry rows;
                                         * 2: the same except that some nodes lack some of the links (row ends: DevView::lmask says which) */
    uint8_t pad1[5];
    uint8_t ukind[SF3D_SLOTS];          /* the one link kind of the nodes that HAVE slot s (= kind[s] when that is uniform; CK_MIXED if they differ) */
    uint8_t pad2[6];
    double area[SF3D_SLOTS];            /* (cell size and layer thickness make it constant over regular grids) */
    double dist[SF3D_SLOTS];
};

struct BalanceDev { double storage, sinkSource, MBE, MBR; };
struct HeatBalanceDev { double storage, sinkSource, MBE, MBR; };

/* stages of the heat part of a computeStep (soilFluxes3D.cpp:1802-1818, CPUSolver::run(Heat)
 * cpusolver.cpp:77-91, heatLoop :471-605): three nested loops driven on the device */
enum : uint32_t {
    HS_IDLE = 0,
    HS_SAVE_WATER = 1,  /* saveWaterFluxValues: link water / vapour fluxes of the accepted water step  */
    HS_BOUNDARY = 2,    /* updateBoundaryHeatData + boundary Courant check                              */
    HS_ASSEMBLE = 3,    /* heatLoop: node properties, then rows                                          */
    HS_SWEEP = 4,       /* iterations of the heat linear system                                          */
    HS_POST = 5,        /* evaluateHeatBalance                                                           */
    HS_SAVE = 6,        /* saveHeatFluxValues of an accepted heat step                                   */
    HS_FINISHED = 7
};
#define SF3D_FLUX_TYPES 9   /* numTotalFluxTypes, types.h:28 */

struct Ctrl {
    /* ---- parameters (SolverParameters, types.h:291-315) ---- */
    double MBRThreshold, residualTolerance, dtMin, dtMax, lvRatio, courantThreshold, instabilityFactor;
    uint32_t maxApprox, maxIter;
    uint32_t wrc, meanType;
    /* ---- adaptive state ---- */
    double dtCurr;          /* deltaTcurr                                  */
    double dt;              /* dt of the running attempt                   */
    double maxTimeStep;     /* argument of computeStep                     */
    double courant;         /* nodeGrid.CourantWater                       */
    double bestMBR;         /* _bestMBRerror                               */
    double bestNorm;        /* bestErrorNorm of solveLinearSystem          */
    double lastNorm;
    uint32_t stage, approx, iter, iterBudget;
    int32_t cur, hold, best; /* indices into the X pool; best = -1 when none */
    uint32_t linearValid;
    uint32_t seSource;      /* where Se(H) of the state an attempt starts from already is: 0 nowhere (state came from the host), 1 the Se
                               array (k_post / k_restore of the accepted step), 2 SeHold (the refused attempt started from the same H) */
    uint32_t epoch;         /* exchange counter, identical on every rank (multi-GPU) */
    uint32_t distError;     /* 1 = a bounded wait for a peer expired */
    uint32_t distSilent;    /* ... bit p: rank p was the one that did not answer */
    uint32_t pairRecTimeout; /* 1: a block of a paired pass waited longer than the bound for a neighbouring strip's record (PairGrid::records) */
    uint32_t kfEpoch, haloEpoch, haloPar; int32_t haloBuf;   /* multi GPU: which exchange the many-block halo copies (k_halo_copy) belong to */
    int32_t acceptBuf;      /* pool index of the accepted H: the link flow sums of the step are added from it ... */
    uint32_t aBuf, acceptABuf;   /* which A2x the step in progress uses / the accepted step used */
    double acceptDt;        /* ... with this dt, possibly while the next step's first kernels already run */
    /* ---- balances (balanceData_t x4, soilFluxes3D.cpp:37) ---- */
    BalanceDev curStep, prevStep, curPeriod, wholePeriod;
    /* ---- heat part of the step ---- */
    uint32_t hStage, hIter, hIterBudget, hRows;
    int32_t tCur, tOld;       /* indices into the temperature pool (Told = T and T = Told are index copies) */
    uint32_t hSweepsLast, hPad;
    double dtWater;           /* accepted water step (or min(maxTimeStep, dtMax) without water)          */
    double hOuterDt, hOuterSum;   /* dtHeat / dtHeatSum of computeStep's loop                            */
    double hMaxStep, hDt, hDone;  /* maxTimeStep / dtHeat / sumHeatTime of CPUSolver::run(Heat)          */
    double hCourant, hNorm;
    double hCacheDt;          /* heat step the cached water contents of k_heat_props (HeatDev::thAvgC / thNewC) belong to: H at the heat time level
                               * depends on dtHeat / dtWater only, so sub-steps of equal length - what the boundary Courant rule produces - share them */
    unsigned long long hNormBits;   /* running maximum of |dx| of a Gauss-Seidel sweep (bit pattern of a non-negative double) */
    HeatBalanceDev heatCur, heatPrev;
    double heatPeriodSink;    /* balanceDataCurrentPeriod.heatSinkSource                                 */
    /* ---- query results (getTotalWaterContent etc.) ---- */
    double query[2];
    /* ---- quirk-1 compat (SF3D_COMPAT_STALE_LINK_FLOW=1): which assembly k_compat_rows has to mirror into the emulated row storage ---- */
    /* ---- Jacobi-preconditioned conjugate gradients standing in for the linealia hook (setUseLineal, cpusolver.cpp:608-669) ---- */
    uint32_t lineal;          /* 1: an approximation's linear system is solved by the k_cg_* kernels instead of Jacobi sweeps */
    uint32_t cgState;         /* 0 idle, 1 iterating, 2 converged / out of budget: k_cg_finish clamps the surface and hands over to ST_POST */
    int32_t cgX;              /* pool index of the iterate (a free buffer: Hold may be the same buffer as H) */
    uint32_t cgPad;
    double cgRho, cgAlpha, cgBeta, cgBnorm2, cgRes2;
    uint32_t seqCount, seqSweeps[16];   /* Jacobi iterations of the 1st, 2nd, ... approximation of the computeStep in progress (0 for one the Courant check
                                          * refused): the host queues that many sweeps (+1) for the same approximation of the NEXT step */
    uint32_t haloLocalTag;    /* paired pass with record hand-over ended by its FIRST decision: epoch + 1 of that moment - the halo of H = x' is in the X pool
                               * already (the blocks that waited for the records stored them), k_post must not arm the copy from the window */
    uint32_t barTimeout;      /* 1: a block waited longer than the bound at a grid barrier (blocks not co-resident?): the step failed */
    uint64_t residentLaunches;  /* k_sweep_resident launches that really ran (one per approximation: all its Jacobi iterations) */
    /* paired sweep on a strip (multi GPU): k_sweep_pair leaves the second iteration of the rows next to a neighbouring strip to
     * k_sweep_bnd, which needs the neighbours' first iterate: pending = 1 between the two launches, x'' goes to pool buffer pairX2 and
     * the (all-gathered) second norm of the rows k_sweep_pair did itself waits in pairNorm2 */
    uint32_t pairPending; int32_t pairX2;
    double pairNorm2, pairNorm2Lo;      /* (a double-double: sf3d_physics.inc "the norm of a Jacobi sweep") */
    uint64_t singleLaunches;  /* k_sweep launches that really ran (next to paired sweeps: the odd iteration of an approximation) */
    uint64_t pairLaunches;    /* k_sweep_pair launches that really ran (guarded no-op launches do not count): event attribution */
    uint32_t asmSeq;          /* counts Courant decisions (= assemblies)                                          */
    uint32_t asmSurfOnly;     /* 1: the Courant check refused the attempt - the reference had assembled only the surface rows by then */
    uint32_t compatSeq;       /* last assembly mirrored                                                            */
    uint32_t compatPad;
    uint32_t probeBudget;     /* early Courant check: approximations it still runs for - refilled by every Courant refusal, used up by checks that pass
                               * (a run whose Courant number sits just below 1 without ever being refused - a real catchment at its Courant-limited
                               * dt - would otherwise pay the check before every approximation for nothing) */
    uint32_t probePad;
    uint64_t earlyCourant;    /* attempts the early Courant check (k_courant_probe) refused before the full properties + assembly ran: a subset of counters[4] */
    /* ---- work counters (include/sf3d.h sf3d_get_counters / sf3d_get_heat_counters) ---- */
    uint64_t counters[8];
    uint64_t heatCounters[4];   /* heat steps accepted / halved (heatLoop returned true / false), boundary Courant reductions of dtHeat, linear-solver sweeps */
};

/* ---- multi-GPU: one process per GPU, row strips of surface-cell columns --------
 * Every rank holds full-size (global-index) arrays but computes only the chunks it owns.  Values a
 * rank's rows read from a neighbouring strip (one-cell halo) are PUT by their owner into the
 * reader's fine-grained, IPC-mapped window right after they are produced, then copied into the
 * reader's own arrays by its next decision kernel, which also all-gathers the partial sums through
 * the same windows (system-scope stores + epoch-stamped flags, double-buffered by epoch parity). */
#define SF3D_MAX_RANKS 16
struct DistMail { unsigned long long w[8]; };      /* four values (both norms of a paired pass, each a double-double) as tagged records: (low 32 bits | tag << 32), (high 32 bits | tag << 32), tag = epoch + 1 - value and flag in one word, one round trip */
struct DistWindow {                     /* head of each rank's window; payload doubles follow */
    DistMail mail[2][SF3D_MAX_RANKS];   /* [epoch parity][source rank] */
    unsigned long long ping[SF3D_MAX_RANKS];   /* start-up self-check: peer p stores a token here through its mapping of this window */
    unsigned long long rrec[2][SF3D_MAX_RANKS][4];   /* [epoch parity][source rank]: that rank's partial norm of a resident-loop iteration as tagged records
                                                       * ((hi, lo) of a double-double, two words each; tag = epoch + 1): every block of every rank reads them
                                                       * here and adds them in rank order - no mailbox round, no fence, no second hand-over inside the rank */
};
/* record hand-over of the masked paired pass, per node (one load instead of a walk through the chunk's send list / a look-up chain):
 * where a foreign local node's record arrives in my window, and where an owned node's record goes in its reader's window - positions in
 * doubles from the window's payload, parity 0, field DF_RECLO; parity 1 lies SF3D_DIST_FIELDS * cnt further, DF_RECHI cnt further */
struct RecGet { uint32_t off0, cnt; };                        /* off0 = SF3D_FSRC_NONE: the node is mine */
struct RecPut { uint32_t off0, cnt, peer, nDest; };           /* nDest: readers of the node (0: none, 1: the entry describes it, 2+: walk the list) */
struct DistView {
    int32_t world, rank;
    DistWindow* win[SF3D_MAX_RANKS];    /* win[rank] = own window (local pointer), others IPC-mapped */
    double* payload[SF3D_MAX_RANKS];    /* payload base of each rank's window */
    /* what I send to peer p: node indices (device array), count, and where it lands in p's payload */
    const uint32_t* sendIdx[SF3D_MAX_RANKS]; uint32_t sendCount[SF3D_MAX_RANKS]; uint64_t sendOff[SF3D_MAX_RANKS];
    /* what I receive from peer p: node indices, count, offset in MY payload */
    const uint32_t* recvIdx[SF3D_MAX_RANKS]; uint32_t recvCount[SF3D_MAX_RANKS]; uint64_t recvOff[SF3D_MAX_RANKS];
    const uint8_t* owner;               /* [N] owning rank of each node (null when world == 1) */
    /* the same send lists indexed by chunk, for kernels that put their boundary values themselves:
     * entries [bndStart[q], bndStart[q+1]) belong to chunk q; each names the lane (node - 64 q), the
     * destination rank and the position in that rank's send list */
    const uint32_t* bndStart;           /* [nChunks + 1] */
    const uint8_t* bndLane; const uint8_t* bndPeer; const uint32_t* bndSlot;
    /* [10][N]: where the value of a FOREIGN neighbour arrives in my window (SF3D_FSRC_NONE for local neighbours); read only
     * in chunks whose descriptor is flagged (ChunkDesc::pad0): the sweeps take foreign neighbours straight from the payload */
    const uint32_t* fsrc;
    const RecGet* recGet; const RecPut* recPut;      /* [N] each (masked paired pass with record hand-over; null otherwise) */
    /* SF3D_EXCHANGE=rccl (or the automatic fall-back when the windows fail their self-check): halos travel as paired ncclSend /
     * ncclRecv of packed buffers and the partial sums as an ncclAllGather, all queued by the host between the kernels (the form
     * SURVEY.md 8e sketches); the decision kernels then combine `gathered` in rank order exactly like the window mailboxes */
    int32_t rccl;
    double* mine;                        /* [4] this rank's partial sums (k_local_reduce) */
    const double* gathered;              /* [world][4] after ncclAllGather */
    double* sendBuf[SF3D_MAX_RANKS];     /* per peer: [2 fields][sendCount] packed halo values */
    const double* recvBuf[SF3D_MAX_RANKS];
    /* wait statistics of the window exchange (device memory of this rank, zeroed at connect): [0] epochs closed, [1 + p] ticks (100 MHz)
     * spent waiting for rank p's mailbox, summed over the epochs, [1 + SF3D_MAX_RANKS + p] the longest single wait - what a rank loses
     * per exchange to the slowest of its peers (sf3d_dist_stats; bench.py prints it per rank) */
    unsigned long long* stats;
    long long spinTicks;                 /* how long (100 MHz ticks) a rank waits for a peer's mail before the step fails: 10 s, SF3D_DIST_TIMEOUT_S (round 5: 60 s - a
                                          * dead rank cost every survivor a minute before the error) */
};
/* payload layout per (receiver, source p): [parity 0/1][field][count] doubles at offset off[p];
 * field 0 = the iterate of a sweep (x of the water system / T of the heat system), 1 = K, 2 = waterFlow.  Separate
 * fields because the x of the LAST sweep of an approximation is consumed late (by the readers' k_post), after a fast
 * neighbour may already have put the K of the next approximation into the same parity. */
#define SF3D_DIST_FIELDS 9
enum { DF_X = 0, DF_K = 1, DF_FLOW = 2, DF_RECLO = 3, DF_RECHI = 4, DF_KLO = 5, DF_KHI = 6, DF_FLO = 7, DF_FHI = 8 };      /* (3, 4: the iterate as tagged records, for the resident sweep loop's in-launch hand-off - sf3d_resident.inc) */
#define SF3D_FSRC_NONE 0xFFFFFFFFu      /* fsrc: (source rank << 27) | position in that rank's send list */

/* coupled heat transport (heat.cpp): everything the heat kernels and the heat terms of the water kernels
 * need; `on` = 0 leaves every pointer null */
struct HeatDev {
    uint32_t on, water, vapor, advection, save;    /* simulationFlags_t, types.h:189-197 */
    uint32_t gs;                                    /* 1: verification mode, the reference's serial Gauss-Seidel order (level-scheduled) */
    uint32_t redBlack;                              /* 1 (default): the heat sweep is a Gauss-Seidel sweep in two colours, odd layers then even layers; 0: Jacobi (SF3D_HEAT_SWEEP=jacobi) */
    const uint8_t* lpar;                            /* [N] layer of the node (hops to the surface through Up links), low bit = its colour */
    const uint32_t* colourList[2]; uint32_t nColour[2];   /* chunks that hold a node of an odd layer [0] / of an even layer [1], in list order (sharded runs: plus the chunks that owe values to a neighbour, in both) */
    const uint32_t* gsOrder;                        /* heat nodes sorted by dependency level (gs mode only) */
    double wf;                                      /* heatWeightFactor, types.h:307 */
    double* TX[3];                                  /* temperature pool: T, Told and the sweep buffers are indices (Ctrl::tCur/tOld) */
    const double* heatSink;
    double *heatFlux, *invariant;                   /* heatData.heatFlux, waterData.invariantFluxes */
    double *hC, *hcapTerm, *hb, *hD;                /* heat capacity x V, water-content-change term, rhs, diagonal */
    sf3d_d2* hA2;                                   /* [5][N] normalised off-diagonals (slots paired like A2); 0 where the slot is no heat link */
    const double* hdist;                            /* [10][N] nodeDistance3D of every link between two soil nodes */
    /* per-node conductivities evaluated once per node instead of once per link end (same arguments => same bits) */
    const double* airP;                             /* [N] computePressure_fromAltitude(z): a pow of a static input, evaluated once per node */
    double* thetaOld;                               /* [N] theta(Hold - z): constant over the heat sub-steps of one water step */
    double *thAvgC, *thNewC;                        /* [N] theta(mean h) and theta(H - z) at the heat time level of Ctrl::hCacheDt */
    double *kHeat, *kIsoVap, *hAvg;                 /* heat process: Campbell conductivity, isothermal vapour conductivity at (T, mean h); mean h */
    double *wThLiq, *wThVap, *wTm;                  /* water process: thermal liquid / vapour conductivity at (mean T, H - z); mean T */
    /* atmosphere boundary (HeatSurface nodes) and fixed-temperature boundary, full-length arrays */
    const double *bHeightWind, *bHeightT, *bRoughH, *bT, *bRH, *bWind, *bNetIrr, *bFixT, *bFixDepth;
    double *bAero, *bSoilCond, *bSens, *bLat, *bRad, *bAdv;
    sf3d_f2* lwvFlux;                               /* [10][N] (water, vapour) flux of the link: the reference keeps them as floats (heat.cpp saveWaterFluxValues),
                                                     * one 8-byte pair per link here - half the bytes of two doubles holding the same float values */
    double* lflux[SF3D_FLUX_TYPES];                 /* [10][N] each; allocated per heatFluxSaveMode_t */
};

/* ---- paired Jacobi sweep (k_sweep_pair): regular NX x NY x NZ grids in layer-major numbering (one GPU, or one row strip of it) ---
 * Per node a 40-bit code, one nibble per link slot: 0..8 = lateral neighbour (dr + 1) * 3 + (dc + 1) of the same layer,
 * 9 = the node above (i - NX NY), 10 = the node below, 15 = no link.  Nodes on the grid's edge fill their lateral slots in
 * their own order (setNodeLink puts the k-th lateral into slot 2 + k), hence a code per node (nodeLat); chunkCode[q] carries the code of
 * chunk q with bit 63 set when all 64 nodes share it (interior), so that the kernel loads it on the scalar unit.  Slot 0 only ever
 * holds 9 or 15 and slot 1 only 10 or 15 (the host build refuses anything else): the kernels take the vertical neighbours from where
 * they are by construction and decode only the eight lateral nibbles (bits 8 .. 39). */
#define SF3D_PAIR_NONE 15u
#define SF3D_PAIR_UP 9u
#define SF3D_PAIR_DOWN 10u
struct PairGrid {
    uint32_t NX, NY, NZ;                /* 0 when the graph is not such a grid (k_sweep then) */
    uint32_t W;                         /* rows of a block's patch (W - 2 owned rows + one halo row on either side) */
    uint32_t patchCols, patchRows;      /* NX / 64, ceil(NY / (W - 2)) */
    uint32_t ownLo, ownHi;              /* rows [ownLo, ownHi) of the grid are this rank's (0, NY on one GPU); a row beyond them is the halo row of a
                                         * neighbouring strip: its x' comes through the window, the second iteration of the rows next to it is k_sweep_bnd's */
    const uint32_t* nodeLat;            /* [N] the eight lateral nibbles of a node's code (bits 8 .. 39) */
    const uint64_t* chunkCode;          /* [N / 64] the whole code of a chunk whose 64 nodes share it, bit 63 set; 0 otherwise */
    /* layered MASKED grids (k_sweep_pair_masked: DEM outlines, soil columns of different depth): NX x NY x NZ is the bounding grid */
    uint32_t masked;                    /* 1: the fields below describe the graph, chunkCode is unused */
    const int32_t* idxMap;              /* [(l NY + r) NX + c] node index of the cell, -1 where there is none */
    const uint32_t* patchList;          /* [blocks] patches that hold nodes: (band of W - 2 rows << 12) | first column (NX <= 4096; else column / 64) */
    const uint8_t* patchDepth;          /* [blocks] bits 0-6: layers the patch (halo included) reaches; bit 7 (multi GPU): a column of another rank lies in the patch's
                                         * footprint - after the first pass of an approximation such blocks take those cells' old iterate from the window */
    uint32_t records;                   /* multi GPU: 1 = ONE launch and ONE exchange per pass - the first iterate of a neighbouring strip's row arrives as tagged
                                         * records (DF_RECLO / DF_RECHI) while the pass runs, the blocks whose ring holds such a row wait for them layer by
                                         * layer, every owned row gets both iterations here and k_sweep_bnd is not launched (sf3d_pair.inc, DIST) */
    /* ... on a regular grid the records of an edge row sit in the neighbour's window at (layer) * stride + column from a base - the host checks the
     * send and receive lists for it - so that neither side looks anything up between computing a value and storing it, or before polling */
    uint32_t recFast;                   /* 1: the five fields below hold */
    int32_t sidePeer[2];                /* the rank above (side 0) / below (1) the strip; -1 at an edge of the grid */
    uint32_t putBase[2], getBase[2];    /* position of (layer 0, column 0) of my edge row in that rank's send list / of its edge row (my halo row) in my receive list */
    uint32_t recStride;                 /* positions per layer (the same in all four lists: NX) */
    const uint8_t* role;                /* [N] multi GPU (masked grids): 0 = another rank's node (halo: never computed here), 1 = mine, both iterations in
                                         * k_sweep_pair_masked, 2 = mine in a chunk that reads a neighbouring strip (ChunkDesc::pad0): first iteration +
                                         * put there, second iteration in k_sweep_bnd.  null on one GPU */
};

/* ---- resident-coefficient sweep loop (k_sweep_resident, sf3d_resident.inc): a regular grid (one GPU) or one row strip of it whose rows
 * fit on chip - every block keeps the normalised rows of its patch (PR rows x 64 columns x NZ layers) in registers for ALL the Jacobi
 * iterations of an approximation, the iterate of the patch and its one-cell halo in an LDS tile; iterations are separated by a grid
 * barrier (all blocks co-resident: the host sizes the grid from the occupancy query) that also carries the norm. */
struct ResGrid {
    uint32_t on;                        /* 0: this graph / grid does not fit (the sweeps are separate launches then) */
    uint32_t NX, NY, NZ;                /* the (rank-local) grid */
    uint32_t PR;                        /* rows of a block's patch */
    uint32_t patchCols, rowGroups;      /* NX / 64, ceil(owned rows / PR): blocks = patchCols * rowGroups, one (or two) per CU */
    uint32_t ownLo, ownHi;              /* rows [ownLo, ownHi) are this rank's (0, NY on one GPU) */
    uint32_t K, NW;                     /* (row, layer) chunks a wave keeps, waves of a block: PR * NZ <= K * NW */
    const uint32_t* nodeLat;            /* [N] the eight lateral nibbles of a node's code (PairGrid) */
    const uint64_t* chunkCode;          /* [N / 64] */
    unsigned int* bar;                  /* [0] iterations of all launches so far: the tag base of the records below */
    unsigned long long* rec;            /* [2 parities][2 halves][N] the new iterate as tagged records (a double = two words {32 data bits, 32-bit tag}) */
    unsigned long long* prec;           /* [2][4][blocks] the blocks' partial norms ((hi, lo) x two halves) */
    unsigned long long* gpub;           /* [2][4] multi GPU: the all-gathered norm, published by block 0 */
    unsigned long long* prec2;          /* [8][blocks] the blocks' sums of the mass balance (the fused post-solve part: storage and sink term, double-doubles, two halves each) */
    uint32_t fusedPost;                 /* 1: the loop's launch also does k_post's work when the solve ends well (SF3D_RESIDENT_POST=0: k_post stays a launch of its own) */
    double* pub;                        /* tuning builds (SF3D_RES_PROFILE): phase timers of block 0 */
    const uint32_t* haloSrc;            /* multi GPU: [2 sides][NZ][NX] where the value of a cell of the foreign halo row above (side 0) / below (1) arrives
                                         * in my window ((source rank << 27) | position in its send list; SF3D_FSRC_NONE: no such row) */
    uint32_t forceTimeout;              /* tests (SF3D_RESIDENT_FAIL_TEST): this launch gives up in its second iteration as if a wait had expired */
    int32_t sidePeer[2];                /* ... the rank that row belongs to (-1: the strip lies at that edge of the grid) */
    uint32_t recFast, putBase[2];       /* 1: my edge row's cells lie at putBase[side] + layer * NX + column in that rank's send list (sf3d_host_build.inc
                                         * edge_rows_direct): the records are stored without a walk through the chunk's send list */
};

struct DevView {
    uint32_t N, ns, nb;                 /* nodes, surface nodes, blocks of SF3D_BLOCK threads */
    uint32_t Nnorm;                     /* node count the mean norm of a sweep divides by: N, or the global count for a strip-local model */
    uint32_t nChunks;                   /* ceil(N / 64): one wave processes one chunk at a time */
    uint32_t qSplit;                    /* chunks [0, qSplit) hold every surface node (runoff/infiltration rows, generic
                                           assembly kernel); chunks [qSplit, nChunks) are soil-only */
    uint32_t nbSurf, nbSoil;            /* grid sizes of the two assembly kernels */
    double probeMin;                    /* early Courant check (k_courant_probe) runs while the last Courant number is at least this; < 0: never */
    /* chunks this rank computes (all chunks when world == 1): whole list, its surface part
     * [0, nListSurf) and its soil part [nListSurf, nList) */
    const uint32_t* chunkList; uint32_t nList, nListSurf;
    const uint32_t* asmList;            /* = chunkList: the order the assembly walks the chunks in */
    uint32_t pLo, pHi;                  /* the part of the list a launch of k_props walks: [0, nList) unless the approximation is queued in slabs */
    uint32_t aLo, aHi;                  /* ... a launch of k_assemble's soil blocks: [nListSurf, nList) */
    const uint32_t* bndList; uint32_t nBnd;   /* multi GPU, paired sweep: the owned chunks that read a neighbouring strip (ChunkDesc::pad0): k_sweep_bnd's rows */
    const uint16_t* lmask;      /* bit s: the node has a link in (device) slot s */
    uint32_t haloDirect;                /* multi GPU: the sweeps read foreign neighbours straight from the window and the halo is copied once
                                           per approximation (k_post) instead of once per sweep, off the critical path (SF3D_HALO_DIRECT=0: old way) */
    uint32_t ntStream;                  /* 1: streamed-once arrays (coefficients, link geometry, flow sums) bypass the caches
                                           (they would evict x, b, z from the 256 MiB Infinity Cache); 0 when the whole
                                           working set of a rank fits the cache (small grids, 8-way sharding) */
    const DistView* dist;               /* device copy; null when world == 1 */
    int32_t world, rank;
    const uint8_t* owner;               /* = dist->owner, null when world == 1 */
    /* per (chunk, slot): when every node of the chunk has the link with the same kind and the same
     * index offset j - i, ckind = that kind and cdelta = that offset (no per-node index traffic);
     * CK_NONE when no node has it (slot skipped); CK_MIXED otherwise */
    const ChunkDesc* cdesc;             /* [nChunks], 64 B each: one scalar load per chunk */
    const double *z, *size, *pond, *sink;
    const uint16_t* cls;
    const uint8_t* btype;
    const double *bslope, *bsize, *prescribed;
    const uint32_t* lto;                /* [10][N] */
    const uint8_t* lkind;               /* [10][N] */
    const uint8_t* probeMask;           /* [ns] early Courant check: bit s - 2 set = lateral slot s of the surface node is the end that evaluates its
                                         * runoff link (the Courant term of a link is bit-identical from both ends: the one with the larger
                                         * neighbour index evaluates it, or the only one where the link has no reverse) */
    const double *larea, *ldist;        /* [10][N] interface area (read only where a chunk's areas differ) and link distance */
    double* lflowSum;                   /* [10][N] */
    sf3d_d2* A2x[2];                    /* two copies of [5][N] row-normalised off-diagonals, slots paired (2p, 2p+1): 16-byte accesses.  The
                                         * computeStep in progress assembles and sweeps A2x[Ctrl::aBuf]; k_step_begin flips aBuf, so the matrix of
                                         * the step accepted before stays intact while its link flow sums are still being added (k_accept_links on the
                                         * second stream, next to the whole next step instead of next to its k_props only) */
    double *b, *C;
    double* X[SF3D_POOL];
    double *Se, *SeHold, *K, *flow, *bflowRate, *bflowSum;   /* SeHold = Se(Hold), written at approximation 0 */
    /* quirk-1 compat, null unless SF3D_COMPAT_STALE_LINK_FLOW=1: the reference's compacted row storage matrixA.values[row][0..10]
     * (compatCv[c][row]) and numColsInRow (compatCn), shared by the water and the heat assembly, mirrored write for write so that
     * the slot a dropped link's search falls into (cpusolver.h:42-52) holds what it holds in the reference; compatDiag: the
     * diagonal of the row k_assemble has just written UN-normalised (k_compat_rows normalises after the Courant decision, like
     * the reference's separate preconditioning pass) */
    double* compatCv; uint8_t* compatCn; double* compatDiag;
    /* conjugate gradients (null unless the device CG is enabled): diagonal of the un-normalised rows, residual, direction, A p */
    double *cgDiag, *cgR, *cgP, *cgQ;
    double *part0, *part1;              /* per-block partials [nb] */
    double *part2, *part3;              /* the second norm of a paired pass (both norms are double-doubles: part0/1 and part2/3) */
    unsigned int* arrive;               /* block arrival counter of the fused sweep + decision kernel */
    unsigned int* gridBar;              /* arrival counter of the persistent step kernel's grid barrier (monotonic) */
    const SoilDev* soils;
    const double* roughness;
    PairGrid pair;
    ResGrid res;
    Ctrl* ctrl;
    HeatDev heat;
};

/* kernels instrumented by sf3d_kernel_timing (ids index the arrays in the solver) */
enum { KID_PROPS = 0, KID_ASSEMBLE, KID_SWEEP, KID_POST, KID_RESTORE, KID_ACCEPT, KID_SWEEP_PAIR, KID_SWEEP_RES, KID_COUNT };

#endif
