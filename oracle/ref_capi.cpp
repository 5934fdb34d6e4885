/*
 * ref_capi.cpp - TEST INFRASTRUCTURE, not product code.
 *
 * Thin forwarding wrapper that exposes the UNMODIFIED reference library
 * (/root/reference/agrolib/soilFluxes3D, compiled where it lies by oracle/Makefile into
 * oracle/_ref/libsf3d_ref.so) through the C ABI of include/sf3d.h, so the same Python
 * harness can drive the reference, the CPU restatement and the HIP product.
 * Nothing here computes anything: every function is one call into soilFluxes3D::v2.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the result.
 */
#include "soilFluxes3D.h"   /* the reference's own header, found via -I in oracle/Makefile */
#include "sf3d.h"

#include <string>

using namespace soilFluxes3D;

#define E(x) static_cast<sf3d_error_t>(x)

extern "C" {

const char* sf3d_backend_name(void) { return "reference"; }

sf3d_error_t sf3d_initialize(uint32_t n, uint32_t ns, uint8_t nl, int w, int h, int s, sf3d_heat_save_t m)
{ return E(initializeSF3D(n, ns, nl, w != 0, h != 0, s != 0, static_cast<heatFluxSaveMode_t>(m))); }
sf3d_error_t sf3d_initialize_balance(void) { return E(initializeBalance()); }
sf3d_error_t sf3d_initialize_log(const char* a, const char* b) { return E(initializeLog(std::string(a ? a : ""), std::string(b ? b : ""))); }
sf3d_error_t sf3d_clean(void) { return E(cleanSF3D()); }
sf3d_error_t sf3d_close_log(void) { return E(closeLog()); }
sf3d_error_t sf3d_initialize_heat_flag(sf3d_heat_save_t m, int adv, int lat)
{ return E(initializeHeatFlag(static_cast<heatFluxSaveMode_t>(m), adv != 0, lat != 0)); }
uint32_t sf3d_set_threads_number(uint32_t n) { return setThreadsNumber(n); }
void sf3d_set_use_lineal(int v) { setUseLineal(v != 0); }
void sf3d_set_lineal_method(int v) { setLinealMethod(v); }

sf3d_error_t sf3d_set_soil_properties(uint16_t a, uint8_t b, double c, double d, double e, double f,
                                      double g, double h, double i, double j, double k, double l)
{ return E(setSoilProperties(a, b, c, d, e, f, g, h, i, j, k, l)); }
sf3d_error_t sf3d_set_surface_properties(uint16_t i, double r) { return E(setSurfaceProperties(i, r)); }
sf3d_error_t sf3d_set_numerical_parameters(double a, double b, uint16_t c, uint16_t d, uint8_t e, uint8_t f)
{ return E(setNumericalParameters(a, b, c, d, e, f)); }
sf3d_error_t sf3d_set_hydraulic_properties(sf3d_wrc_t a, sf3d_mean_t b, float c)
{ return E(setHydraulicProperties(static_cast<WRCModel>(a), static_cast<meanType_t>(b), c)); }

/* the reference's setCulvert writes through a never-allocated pointer;
 * it is not forwarded */
sf3d_error_t sf3d_set_culvert(uint32_t, double, double, double, double) { return SF3D_BOUNDARY_ERROR; }
sf3d_error_t sf3d_set_node(uint32_t i, double x, double y, double z, double v, int s, sf3d_boundary_t b, double sl, double ba)
{ return E(setNode(i, x, y, z, v, s != 0, static_cast<boundaryType_t>(b), sl, ba)); }
sf3d_error_t sf3d_set_node_link(uint32_t i, uint32_t j, sf3d_link_t d, double a)
{ return E(setNodeLink(i, j, static_cast<linkType_t>(d), a)); }
sf3d_error_t sf3d_set_node_boundary(uint32_t i, sf3d_boundary_t b, double s, double a)
{ return E(setNodeBoundary(i, static_cast<boundaryType_t>(b), s, a)); }
sf3d_error_t sf3d_set_node_soil(uint32_t i, uint16_t s, uint16_t h) { return E(setNodeSoil(i, s, h)); }
sf3d_error_t sf3d_set_node_surface(uint32_t i, uint16_t s) { return E(setNodeSurface(i, s)); }

sf3d_error_t sf3d_set_node_pond(uint32_t i, double v) { return E(setNodePond(i, v)); }
sf3d_error_t sf3d_set_node_water_content(uint32_t i, double v) { return E(setNodeWaterContent(i, v)); }
sf3d_error_t sf3d_set_node_degree_of_saturation(uint32_t i, double v) { return E(setNodeDegreeOfSaturation(i, v)); }
sf3d_error_t sf3d_set_node_matric_potential(uint32_t i, double v) { return E(setNodeMatricPotential(i, v)); }
sf3d_error_t sf3d_set_node_total_potential(uint32_t i, double v) { return E(setNodeTotalPotential(i, v)); }
sf3d_error_t sf3d_set_node_water_sink_source(uint32_t i, double v) { return E(setNodeWaterSinkSource(i, v)); }
sf3d_error_t sf3d_set_node_prescribed_total_potential(uint32_t i, double v) { return E(setNodePrescribedTotalPotential(i, v)); }

double sf3d_get_node_water_content(uint32_t i) { return getNodeWaterContent(i); }
double sf3d_get_node_maximum_water_content(uint32_t i) { return getNodeMaximumWaterContent(i); }
double sf3d_get_node_minimum_water_content(uint32_t i) { return getNodeMinimumWaterContent(i); }
double sf3d_get_node_available_water_content(uint32_t i) { return getNodeAvailableWaterContent(i); }
double sf3d_get_node_water_deficit(uint32_t i, double fc) { return getNodeWaterDeficit(i, fc); }
double sf3d_get_node_degree_of_saturation(uint32_t i) { return getNodeDegreeOfSaturation(i); }
double sf3d_get_node_water_conductivity(uint32_t i) { return getNodeWaterConductivity(i); }
double sf3d_get_node_matric_potential(uint32_t i) { return getNodeMatricPotential(i); }
double sf3d_get_node_total_potential(uint32_t i) { return getNodeTotalPotential(i); }
double sf3d_get_node_pond(uint32_t i) { return getNodePond(i); }
double sf3d_get_node_max_water_flow(uint32_t i, sf3d_link_t d) { return getNodeMaxWaterFlow(i, static_cast<linkType_t>(d)); }
double sf3d_get_node_sum_lateral_water_flow(uint32_t i) { return getNodeSumLateralWaterFlow(i); }
double sf3d_get_node_sum_lateral_water_flow_in(uint32_t i) { return getNodeSumLateralWaterFlowIn(i); }
double sf3d_get_node_sum_lateral_water_flow_out(uint32_t i) { return getNodeSumLateralWaterFlowOut(i); }
double sf3d_get_node_boundary_water_flow(uint32_t i) { return getNodeBoundaryWaterFlow(i); }
double sf3d_get_total_boundary_water_flow(sf3d_boundary_t b) { return getTotalBoundaryWaterFlow(static_cast<boundaryType_t>(b)); }
double sf3d_get_total_water_content(void) { return getTotalWaterContent(); }
double sf3d_get_water_storage(void) { return getWaterStorage(); }
double sf3d_get_water_mbr(void) { return getWaterMBR(); }

sf3d_error_t sf3d_set_node_heat_sink_source(uint32_t i, double v) { return E(setNodeHeatSinkSource(i, v)); }
sf3d_error_t sf3d_set_node_temperature(uint32_t i, double v) { return E(setNodeTemperature(i, v)); }
sf3d_error_t sf3d_set_node_boundary_fixed_temperature(uint32_t i, double t, double d) { return E(setNodeBoundaryFixedTemperature(i, t, d)); }
sf3d_error_t sf3d_set_node_boundary_height_wind(uint32_t i, double v) { return E(setNodeBoundaryHeightWind(i, v)); }
sf3d_error_t sf3d_set_node_boundary_height_temperature(uint32_t i, double v) { return E(setNodeBoundaryHeightTemperature(i, v)); }
sf3d_error_t sf3d_set_node_boundary_net_irradiance(uint32_t i, double v) { return E(setNodeBoundaryNetIrradiance(i, v)); }
sf3d_error_t sf3d_set_node_boundary_temperature(uint32_t i, double v) { return E(setNodeBoundaryTemperature(i, v)); }
sf3d_error_t sf3d_set_node_boundary_relative_humidity(uint32_t i, double v) { return E(setNodeBoundaryRelativeHumidity(i, v)); }
sf3d_error_t sf3d_set_node_boundary_roughness(uint32_t i, double v) { return E(setNodeBoundaryRoughness(i, v)); }
sf3d_error_t sf3d_set_node_boundary_wind_speed(uint32_t i, double v) { return E(setNodeBoundaryWindSpeed(i, v)); }

double sf3d_get_node_temperature(uint32_t i) { return getNodeTemperature(i); }
double sf3d_get_node_heat_conductivity(uint32_t i) { return getNodeHeatConductivity(i); }
double sf3d_get_node_vapor(uint32_t i) { return getNodeVapor(i); }
double sf3d_get_node_heat_storage(uint32_t i, double h) { return getNodeHeatStorage(i, h); }
double sf3d_get_node_heat_max_flux(uint32_t i, sf3d_link_t d, sf3d_flux_t f) { return getNodeHeatMaxFlux(i, static_cast<linkType_t>(d), static_cast<fluxTypes_t>(f)); }
double sf3d_get_node_boundary_advective_flux(uint32_t i) { return getNodeBoundaryAdvectiveFlux(i); }
double sf3d_get_node_boundary_latent_flux(uint32_t i) { return getNodeBoundaryLatentFlux(i); }
double sf3d_get_node_boundary_radiative_flux(uint32_t i) { return getNodeBoundaryRadiativeFlux(i); }
double sf3d_get_node_boundary_sensible_flux(uint32_t i) { return getNodeBoundarySensibleFlux(i); }
double sf3d_get_node_boundary_aerodynamic_conductance(uint32_t i) { return getNodeBoundaryAerodynamicConductance(i); }
double sf3d_get_node_boundary_soil_conductance(uint32_t i) { return getNodeBoundarySoilConductance(i); }
double sf3d_get_heat_mbr(void) { return getHeatMBR(); }
double sf3d_get_heat_mbe(void) { return getHeatMBE(); }

void sf3d_compute_period(double t) { computePeriod(t); }
double sf3d_compute_step(double t) { return computeStep(t); }

/* ---- extensions: loops over the reference's scalar API -------------------------------- */

sf3d_error_t sf3d_set_nodes(uint32_t first, uint32_t count, const double* x, const double* y, const double* z,
                            const double* v, const uint8_t* surf, const uint8_t* bt, const double* sl, const double* ba)
{
    for (uint32_t k = 0; k < count; ++k) {
        sf3d_error_t e = sf3d_set_node(first + k, x[k], y[k], z[k], v[k], surf[k], bt ? bt[k] : 0,
                                       sl ? sl[k] : 0., ba ? ba[k] : 0.);
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
    { for (uint32_t k = 0; k < count; ++k) { sf3d_error_t e = CALL; if (e != SF3D_OK) return e; } return SF3D_OK; }
BULK_SET(sf3d_set_nodes_soil, sf3d_set_node_soil(first + k, s[k], h ? h[k] : 0), const uint16_t* s, const uint16_t* h)
BULK_SET(sf3d_set_nodes_surface, sf3d_set_node_surface(first + k, s[k]), const uint16_t* s)
BULK_SET(sf3d_set_nodes_pond, sf3d_set_node_pond(first + k, v[k]), const double* v)
BULK_SET(sf3d_set_nodes_matric_potential, sf3d_set_node_matric_potential(first + k, v[k]), const double* v)
BULK_SET(sf3d_set_nodes_total_potential, sf3d_set_node_total_potential(first + k, v[k]), const double* v)
BULK_SET(sf3d_set_nodes_water_sink_source, sf3d_set_node_water_sink_source(first + k, v[k]), const double* v)
#define BULK_GET(NAME, CALL)                                                          \
    sf3d_error_t NAME(uint32_t first, uint32_t count, double* out)                     \
    { for (uint32_t k = 0; k < count; ++k) out[k] = CALL(first + k); return SF3D_OK; }
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

sf3d_error_t sf3d_get_counters(uint64_t*) { return SF3D_MISSING_DATA_ERROR; }
/* Solver::getTimeStep is public on the reference's global `solver` object but that object is
 * not part of the public header; not available here */
double sf3d_get_linear_residual(void) { return -9999.; }   /* not tracked by the unmodified reference */
double sf3d_get_time_step(void) { return SF3D_VAL_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_set_time_step(double) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_reset_solver_state(void) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_set_surface_nodes_number(uint32_t) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_set_device(int) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_synchronize(void) { return SF3D_OK; }
sf3d_error_t sf3d_kernel_timing(int) { return SF3D_MISSING_DATA_ERROR; }
int sf3d_kernel_count(void) { return 0; }
const char* sf3d_kernel_name(int) { return nullptr; }
sf3d_error_t sf3d_kernel_stats(int, uint64_t*, double*, uint64_t*) { return SF3D_MISSING_DATA_ERROR; }
int sf3d_libm_set(void) { return 1; }      /* the reference calls the C library */
sf3d_error_t sf3d_device_log(uint32_t, const double*, double*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_device_exp(uint32_t, const double*, double*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_device_cbrt(uint32_t, const double*, double*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_device_norm_sum(uint32_t, const double*, uint32_t, int, double*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_get_sweep_launches(uint64_t*, uint64_t*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_get_resident_launches(uint64_t*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_get_heat_counters(uint64_t*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_device_pow(uint32_t, const double*, const double*, double*) { return SF3D_MISSING_DATA_ERROR; }

/* multi-GPU entry points exist only in the HIP product */
uint64_t sf3d_device_bytes(void) { return 0; }
uint64_t sf3d_host_bytes(void) { return 0; }
int sf3d_dist_blob_bytes(void) { return 0; }
sf3d_error_t sf3d_dist_prepare(int, int) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_dist_export(void*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_dist_connect(const void*) { return SF3D_MISSING_DATA_ERROR; }
int sf3d_dist_status(void) { return 0; }
sf3d_error_t sf3d_dist_finalize(int) { return SF3D_MISSING_DATA_ERROR; }
int sf3d_dist_transport(void) { return 0; }
sf3d_error_t sf3d_dist_stats(double*, int) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_dist_bounds(uint32_t, int, uint32_t*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_get_regular_grid(uint32_t*, uint32_t*, uint32_t*, int8_t*, int8_t*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_dist_owner(int, uint32_t, uint32_t, int32_t*) { return SF3D_MISSING_DATA_ERROR; }
sf3d_error_t sf3d_dist_halo(int, int, int, int, uint32_t, uint32_t*, uint32_t*) { return SF3D_MISSING_DATA_ERROR; }

} /* extern "C" */
