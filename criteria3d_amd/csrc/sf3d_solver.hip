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
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

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

#include "sf3d_physics.inc"
#include "sf3d_control.inc"
#include "sf3d_phases.inc"
#include "sf3d_host_build.inc"
#include "sf3d_host_step.inc"
