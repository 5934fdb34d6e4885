// A caller written against the reference's public header (soilFluxes3D.h): the C1 column with coupled heat, driven
// through the soilFluxes3D::v2 C++ symbols exactly as bin/CRITERIA3D would.  criteria3d_amd/build.py compiles it against
// the reference's OWN header when /root/reference is mounted (-DSF3D_USE_REFERENCE_HEADER), else against
// shim/soilFluxes3D_api.h, and links it to shim/libsoilFluxes3D_mi355x.so: no source change on the caller's side.
#ifdef SF3D_USE_REFERENCE_HEADER
#include "soilFluxes3D.h"
#else
#include "soilFluxes3D_api.h"
#endif
#include <cstdio>
#ifdef SF3D_DEMO_LINEALIA
#include "linealiaLib.h"      /* the application's main.cpp:81 asks the solver library for the optional linealia back end */
#endif
using namespace soilFluxes3D;
int main()
{
#ifdef SF3D_DEMO_LINEALIA
    printf("linealia loaded: %d\n", (int)LinealiaLib::instance().load());
#endif
    const SF3Duint_t N = 22; const double dz = 0.05, area = 1.0, n = 1.56;
    if (initializeSF3D(N, 1, 8, true, true, false, heatFluxSaveMode_t::Total) != SF3Derror_t::SF3Dok) return 2;
    initializeHeatFlag(heatFluxSaveMode_t::Total, false, true);
    setSurfaceProperties(0, 0.05);
    setSoilProperties(0, 0, 3.6, n, 1 - 1 / n, 0.1, 0.078, 0.43, 2.9e-6, 0.5, 0.01, 0.2);
    for (SF3Duint_t i = 0; i < N; ++i) {
        boundaryType_t bt = boundaryType_t::NoBoundary;
        if (i == 1) bt = boundaryType_t::HeatSurface;
        if (i == N - 1) bt = boundaryType_t::FreeDrainage;
        if (i == 0) setNode(0, 0, 0, 0.0, area, true, bt, 0, 0);
        else setNode(i, 0, 0, -(dz * (i - 0.5)), area * dz, false, bt, 0, area);
        if (i > 0) setNodeLink(i, i - 1, linkType_t::Up, area);
        if (i < N - 1) setNodeLink(i, i + 1, linkType_t::Down, area);
    }
    setNodeSurface(0, 0); setNodePond(0, 0.002);
    for (SF3Duint_t i = 1; i < N; ++i) setNodeSoil(i, 0, 0);
    setHydraulicProperties(WRCModel::ModifiedVanGenuchten, meanType_t::Logarithmic, 10.f);
    setNumericalParameters(1, 3600, 150, 10, 10, 3);
    setThreadsNumber(1);
    for (SF3Duint_t i = 0; i < N; ++i) setNodeTemperature(i, 288.15 - 2.0 * (i == 0 ? 0.0 : dz * (i - 0.5)));
    setNodeMatricPotential(0, 0.0);
    for (SF3Duint_t i = 1; i < N; ++i) setNodeMatricPotential(i, -3.0);
    setNodeBoundaryHeightWind(1, 2.0); setNodeBoundaryHeightTemperature(1, 2.0); setNodeBoundaryRoughness(1, 0.01);
    setNodeBoundaryFixedTemperature(N - 1, 285.15, 0.5);
    if (initializeBalance() != SF3Derror_t::SF3Dok) return 3;
    for (int h = 0; h < 2; ++h) {
        setNodeBoundaryTemperature(1, 290.0 + h); setNodeBoundaryRelativeHumidity(1, 60.0); setNodeBoundaryWindSpeed(1, 2.0);
        setNodeBoundaryNetIrradiance(1, 100.0);
        setNodeWaterSinkSource(0, (h == 0 ? 1e-3 : 0.0) / 3600. * area);
        double t = 0; int steps = 0;
        while (t < 3600) { double dt = computeStep(3600 - t); if (!(dt > 0)) return 4; t += dt; ++steps; }
        printf("h%d steps=%d H1=%.12g T1=%.12g T10=%.12g storage=%.12g sens=%.12g flux=%.9g\n", h, steps, getNodeTotalPotential(1),
               getNodeTemperature(1), getNodeTemperature(10), getWaterStorage(), getNodeBoundarySensibleFlux(1),
               getNodeHeatMaxFlux(2, linkType_t::Down, fluxTypes_t::HeatTotal));
    }
    cleanSF3D();
    return 0;
}
