// drives the C1 column through the retired v1 API names on top of the product
#include "soilFluxes3D_v1_api.h"
#include <cstdio>
using namespace soilFluxes3D::v1;
int main()
{
    const long N = 100; const double dz = 0.02, area = 1.0, n = 1.56;
    if (initializeFluxes(N, 100, 8, true, false, false) != 0) return 2;
    setSurfaceProperties(0, 0.05);
    setSoilProperties(0, 0, 3.6, n, 1 - 1 / n, 0.1, 0.078, 0.43, 2.9e-6, 0.5, 0.01, 0.2);
    for (long i = 0; i < N; ++i) {
        if (i == 0) setNode(0, 0, 0, 0.0, area, true, false, 99, 0, 0);
        else setNode(i, 0, 0, -(dz * (i - 0.5)), area * dz, false, i == N - 1, 3 /* BOUNDARY_FREEDRAINAGE */, 0, (float)area);
        if (i > 0) setNodeLink(i, i - 1, 1, (float)area);
        if (i < N - 1) setNodeLink(i, i + 1, 2, (float)area);
    }
    setNodeSurface(0, 0); setNodePond(0, 0.002);
    for (long i = 1; i < N; ++i) setNodeSoil(i, 0, 0);
    setHydraulicProperties(1 /* MODIFIEDVANGENUCHTEN */, 1 /* MEAN_LOGARITHMIC */, 10.f);
    setNumericalParameters(1, 3600, 150, 10, 10, 3);
    setMatricPotential(0, 0.0);
    for (long i = 1; i < N; ++i) setMatricPotential(i, -3.0);
    initializeBalance();
    for (int h = 0; h < 2; ++h) {
        setWaterSinkSource(0, 5e-3 / 3600. * area);
        double t = 0; int steps = 0;
        while (t < 3600) { double dt = computeStep(3600 - t); if (!(dt > 0)) return 3; t += dt; ++steps; }
        printf("h%d steps=%d H1=%.12g H99=%.12g storage=%.12g drain=%.12g\n", h, steps, getTotalPotential(1), getTotalPotential(99),
               getWaterStorage(), getBoundaryWaterSumFlow(3));
    }
    cleanMemory();
    return 0;
}
