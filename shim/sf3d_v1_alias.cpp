/*
 * sf3d_v1_alias.cpp - alias layer: the retired `soilFluxes3D::v1` entry points
 * (old/old_soilFluxes3D.h:16-113; `initializeFluxes`, `setNode`, `setNodeLink`, `computePeriod`, ...)
 * forwarded to the C ABI of include/sf3d.h with v1's integer codes and enumerations translated.
 * The numerics are the current (v2) solver's: v1's own Gauss-Seidel water solver is retired code.
 */
#include "soilFluxes3D_v1_api.h"
#include "sf3d.h"

#include <cstdint>

namespace {

long g_nrNodes = 0;
uint32_t g_surfaceCount = 0;
bool g_surfaceCountDirty = false;

int code(sf3d_error_t e)            /* commonConstants.h:95-105 */
{
    switch (e) {
        case SF3D_OK: return 0;
        case SF3D_INDEX_ERROR: return -1111;
        case SF3D_MEMORY_ERROR: return -2222;
        case SF3D_TOPOGRAPHY_ERROR: return -3333;
        case SF3D_BOUNDARY_ERROR: return -4444;
        case SF3D_MISSING_DATA_ERROR: return -9999;
        case SF3D_PARAMETER_ERROR: return -7777;
        default: return 100;         /* CRIT3D_ERROR */
    }
}
sf3d_boundary_t boundary(int v1Type)  /* commonConstants.h:118-128 -> types.h:98 */
{
    switch (v1Type) {
        case 2: return SF3D_BND_RUNOFF;
        case 3: return SF3D_BND_FREE_DRAINAGE;
        case 4: return SF3D_BND_FREE_LATERAL_DRAINAGE;
        case 5: return SF3D_BND_PRESCRIBED_TOTAL_POTENTIAL;
        case 10: return SF3D_BND_URBAN;
        case 11: return SF3D_BND_ROAD;
        case 12: return SF3D_BND_CULVERT;
        case 20: return SF3D_BND_HEAT_SURFACE;
        case 30: return SF3D_BND_SOLUTE_FLUX;
        default: return SF3D_BND_NONE;
    }
}
bool badIndex(long i) { return i < 0 || i >= g_nrNodes; }
void flushSurfaceCount()
{
    if (g_surfaceCountDirty) { sf3d_set_surface_nodes_number(g_surfaceCount); g_surfaceCountDirty = false; }
}

}  // namespace

#define IDX(i) if (badIndex(i)) return -1111
#define IDXD(i) if (badIndex(i)) return -1111.0
#define U32(i) static_cast<uint32_t>(i)

namespace soilFluxes3D::v1 {

int test() { return 0; }
void cleanMemory() { sf3d_clean(); g_nrNodes = 0; g_surfaceCount = 0; g_surfaceCountDirty = false; }

int initializeFluxes(long nrNodes, int nrLayers, int nrLateralLinks, bool w, bool h, bool s)
{
    if (nrNodes < 0 || nrNodes > 0xFFFFFFFFL || nrLateralLinks < 0 || nrLateralLinks > 255) return -7777;
    /* v1 does not know the number of surface nodes at this point: start from nodes / layers and
     * correct it from the isSurface flags of setNode before the first step */
    uint32_t guess = nrLayers > 0 ? U32(nrNodes / nrLayers) : 0;
    int rc = code(sf3d_initialize(U32(nrNodes), guess, static_cast<uint8_t>(nrLateralLinks), w, h, s, 0));
    if (rc == 0) { g_nrNodes = nrNodes; g_surfaceCount = 0; g_surfaceCountDirty = true; }
    return rc;
}
void initializeHeat(short save, bool adv, bool lat) { sf3d_initialize_heat_flag(static_cast<uint8_t>(save), adv, lat); }

int setNumericalParameters(double minDt, double maxDt, int maxIter, int maxApprox, int residualTolerance, double MBRThreshold)
{
    /* v1 takes both tolerances as exponents too (old_soilFluxes3D.cpp:175-185) */
    if (maxIter < 0) maxIter = 0;
    if (maxIter > 65535) maxIter = 65535;
    if (maxApprox < 0) maxApprox = 0;
    if (maxApprox > 65535) maxApprox = 65535;
    if (residualTolerance < 0) residualTolerance = 0;
    if (residualTolerance > 255) residualTolerance = 255;
    double m = MBRThreshold < 0 ? 0 : (MBRThreshold > 255 ? 255 : MBRThreshold);
    return code(sf3d_set_numerical_parameters(minDt, maxDt, static_cast<uint16_t>(maxIter), static_cast<uint16_t>(maxApprox),
                                              static_cast<uint8_t>(residualTolerance), static_cast<uint8_t>(m)));
}
int setThreadsNumber(int n) { return static_cast<int>(sf3d_set_threads_number(n < 0 ? 0u : U32(n))); }

int setNode(long i, float x, float y, double z, double v, bool isSurface, bool isBoundary, int bt, float slope, float area)
{
    IDX(i);
    int rc = code(sf3d_set_node(U32(i), x, y, z, v, isSurface, isBoundary ? boundary(bt) : SF3D_BND_NONE, slope, area));
    if (rc == 0 && isSurface) { ++g_surfaceCount; g_surfaceCountDirty = true; }
    return rc;
}
int setNodeLink(long i, long j, short direction, float S0)
{
    IDX(i); IDX(j);
    if (direction < 1 || direction > 3) return -7777;
    return code(sf3d_set_node_link(U32(i), U32(j), static_cast<uint8_t>(direction), S0));   /* UP 1, DOWN 2, LATERAL 3 in both APIs */
}
int setCulvert(long i, double r, double s, double w, double h) { IDX(i); return code(sf3d_set_culvert(U32(i), r, s, w, h)); }

int setSoilProperties(int nrSoil, int nrHorizon, double a, double n, double m, double he, double tr, double ts, double ks, double L,
                      double om, double clay)
{
    if (nrSoil < 0 || nrSoil > 65535 || nrHorizon < 0 || nrHorizon > 255) return -7777;
    return code(sf3d_set_soil_properties(static_cast<uint16_t>(nrSoil), static_cast<uint8_t>(nrHorizon), a, n, m, he, tr, ts, ks, L, om, clay));
}
int setNodeSoil(long i, int soil, int horizon)
{
    IDX(i);
    if (soil < 0 || soil > 65535 || horizon < 0 || horizon > 65535) return -7777;
    flushSurfaceCount();
    return code(sf3d_set_node_soil(U32(i), static_cast<uint16_t>(soil), static_cast<uint16_t>(horizon)));
}
int setSurfaceProperties(int idx, double roughness)
{ if (idx < 0 || idx > 65535) return -7777; return code(sf3d_set_surface_properties(static_cast<uint16_t>(idx), roughness)); }
int setNodeSurface(long i, int idx)
{ IDX(i); if (idx < 0 || idx > 65535) return -7777; return code(sf3d_set_node_surface(U32(i), static_cast<uint16_t>(idx))); }
int setNodePond(long i, double pond) { IDX(i); return code(sf3d_set_node_pond(U32(i), pond)); }

int setHydraulicProperties(int wrc, int meanType, float ratio)
{
    /* v1: MEAN_GEOMETRIC 0, MEAN_LOGARITHMIC 1 (commonConstants.h:112-113) */
    const sf3d_mean_t mt = meanType == 0 ? SF3D_MEAN_GEOMETRIC : SF3D_MEAN_LOGARITHMIC;
    if (wrc < 0 || wrc > 2) return -7777;
    return code(sf3d_set_hydraulic_properties(static_cast<uint8_t>(wrc), mt, ratio));
}
int setWaterContent(long i, double v) { IDX(i); flushSurfaceCount(); return code(sf3d_set_node_water_content(U32(i), v)); }
int setDegreeOfSaturation(long i, double v) { IDX(i); flushSurfaceCount(); return code(sf3d_set_node_degree_of_saturation(U32(i), v)); }
int setMatricPotential(long i, double v) { IDX(i); flushSurfaceCount(); return code(sf3d_set_node_matric_potential(U32(i), v)); }
int setTotalPotential(long i, double v) { IDX(i); flushSurfaceCount(); return code(sf3d_set_node_total_potential(U32(i), v)); }
int setPrescribedTotalPotential(long i, double v) { IDX(i); return code(sf3d_set_node_prescribed_total_potential(U32(i), v)); }
int setWaterSinkSource(long i, double v) { IDX(i); return code(sf3d_set_node_water_sink_source(U32(i), v)); }

double getWaterContent(long i) { IDXD(i); return sf3d_get_node_water_content(U32(i)); }
double getMaximumWaterContent(long i) { IDXD(i); return sf3d_get_node_maximum_water_content(U32(i)); }
double getAvailableWaterContent(long i) { IDXD(i); return sf3d_get_node_available_water_content(U32(i)); }
double getWaterDeficit(long i, double fc) { IDXD(i); return sf3d_get_node_water_deficit(U32(i), fc); }
double getTotalWaterContent() { flushSurfaceCount(); return sf3d_get_total_water_content(); }
double getDegreeOfSaturation(long i) { IDXD(i); return sf3d_get_node_degree_of_saturation(U32(i)); }
double getBoundaryWaterFlow(long i) { IDXD(i); return sf3d_get_node_boundary_water_flow(U32(i)); }
double getBoundaryWaterSumFlow(int bt) { return sf3d_get_total_boundary_water_flow(boundary(bt)); }
double getMatricPotential(long i) { IDXD(i); return sf3d_get_node_matric_potential(U32(i)); }
double getTotalPotential(long i) { IDXD(i); return sf3d_get_node_total_potential(U32(i)); }
double getWaterMBR() { return sf3d_get_water_mbr(); }
double getWaterConductivity(long i) { IDXD(i); return sf3d_get_node_water_conductivity(U32(i)); }
double getWaterFlow(long i, short direction)
{ IDXD(i); if (direction < 1 || direction > 3) return -1111.0; return sf3d_get_node_max_water_flow(U32(i), static_cast<uint8_t>(direction)); }
double getSumLateralWaterFlow(long i) { IDXD(i); return sf3d_get_node_sum_lateral_water_flow(U32(i)); }
double getSumLateralWaterFlowIn(long i) { IDXD(i); return sf3d_get_node_sum_lateral_water_flow_in(U32(i)); }
double getSumLateralWaterFlowOut(long i) { IDXD(i); return sf3d_get_node_sum_lateral_water_flow_out(U32(i)); }
double getWaterStorage() { return sf3d_get_water_storage(); }
double getPond(long i) { IDXD(i); return sf3d_get_node_pond(U32(i)); }

int setHeatSinkSource(long i, double v) { IDX(i); return code(sf3d_set_node_heat_sink_source(U32(i), v)); }
int setTemperature(long i, double v) { IDX(i); return code(sf3d_set_node_temperature(U32(i), v)); }
int setHeatBoundaryHeightWind(long i, double v) { IDX(i); return code(sf3d_set_node_boundary_height_wind(U32(i), v)); }
int setHeatBoundaryHeightTemperature(long i, double v) { IDX(i); return code(sf3d_set_node_boundary_height_temperature(U32(i), v)); }
int setHeatBoundaryTemperature(long i, double v) { IDX(i); return code(sf3d_set_node_boundary_temperature(U32(i), v)); }
int setHeatBoundaryRelativeHumidity(long i, double v) { IDX(i); return code(sf3d_set_node_boundary_relative_humidity(U32(i), v)); }
int setHeatBoundaryRoughness(long i, double v) { IDX(i); return code(sf3d_set_node_boundary_roughness(U32(i), v)); }
int setHeatBoundaryWindSpeed(long i, double v) { IDX(i); return code(sf3d_set_node_boundary_wind_speed(U32(i), v)); }
int setHeatBoundaryNetIrradiance(long i, double v) { IDX(i); return code(sf3d_set_node_boundary_net_irradiance(U32(i), v)); }
int setFixedTemperature(long i, double t, double d) { IDX(i); return code(sf3d_set_node_boundary_fixed_temperature(U32(i), t, d)); }

double getTemperature(long i) { IDXD(i); return sf3d_get_node_temperature(U32(i)); }
double getHeatConductivity(long i) { IDXD(i); return sf3d_get_node_heat_conductivity(U32(i)); }
double getHeat(long i, double h) { IDXD(i); return sf3d_get_node_heat_storage(U32(i), h); }
double getNodeVapor(long i) { IDXD(i); return sf3d_get_node_vapor(U32(i)); }
float getHeatFlux(long i, short direction, int fluxType)
{ if (badIndex(i) || direction < 1 || direction > 3 || fluxType < 0 || fluxType > 8) return -1111.f;
  return static_cast<float>(sf3d_get_node_heat_max_flux(U32(i), static_cast<uint8_t>(direction), static_cast<uint8_t>(fluxType))); }
double getBoundarySensibleFlux(long i) { IDXD(i); return sf3d_get_node_boundary_sensible_flux(U32(i)); }
double getBoundaryAdvectiveFlux(long i) { IDXD(i); return sf3d_get_node_boundary_advective_flux(U32(i)); }
double getBoundaryLatentFlux(long i) { IDXD(i); return sf3d_get_node_boundary_latent_flux(U32(i)); }
double getBoundaryRadiativeFlux(long i) { IDXD(i); return sf3d_get_node_boundary_radiative_flux(U32(i)); }
double getBoundaryAerodynamicConductance(long i) { IDXD(i); return sf3d_get_node_boundary_aerodynamic_conductance(U32(i)); }
double getBoundarySoilConductance(long i) { IDXD(i); return sf3d_get_node_boundary_soil_conductance(U32(i)); }
double getHeatMBR() { return sf3d_get_heat_mbr(); }
double getHeatMBE() { return sf3d_get_heat_mbe(); }

void initializeBalance() { flushSurfaceCount(); sf3d_initialize_balance(); }
void computePeriod(double t) { flushSurfaceCount(); sf3d_compute_period(t); }
double computeStep(double t) { flushSurfaceCount(); return sf3d_compute_step(t); }

}  // namespace soilFluxes3D::v1
