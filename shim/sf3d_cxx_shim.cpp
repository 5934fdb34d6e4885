/*
 * sf3d_cxx_shim.cpp - the drop-in translation unit: defines the reference's 70
 * `soilFluxes3D::v2::*` functions (Itanium-mangled, NOT extern "C" - SURVEY.md 8b / App. E) by
 * forwarding each to the flat C ABI of include/sf3d.h, so bin/CRITERIA3D (and VINE3D) link the
 * MI355X library in place of agrolib/soilFluxes3D with no source change.
 *
 * Compile it either against the reference's own header (-DSF3D_USE_REFERENCE_HEADER
 * -I<ref>/agrolib/soilFluxes3D -I<ref>/agrolib/mathFunctions) or against shim/soilFluxes3D_api.h;
 * both give the same symbols.  It also provides LinealiaLib::load() for main.cpp:81 when built
 * with -DSF3D_WITH_LINEALIA_STUB against the reference's lineal/linealiaLib.h.
 */
#ifdef SF3D_USE_REFERENCE_HEADER
#include "soilFluxes3D.h"
#else
#include "soilFluxes3D_api.h"
#endif
#include "sf3d.h"

namespace soilFluxes3D { inline namespace v2 {

#define ERR(x) static_cast<SF3Derror_t>(x)
#define U8(x) static_cast<std::uint8_t>(x)

SF3Derror_t initializeSF3D(SF3Duint_t n, SF3Duint_t ns, u8_t nl, bool w, bool h, bool s, heatFluxSaveMode_t m)
{ return ERR(sf3d_initialize(n, ns, nl, w, h, s, U8(m))); }
SF3Derror_t initializeBalance() { return ERR(sf3d_initialize_balance()); }
SF3Derror_t initializeLog(const std::string& a, const std::string& b) { return ERR(sf3d_initialize_log(a.c_str(), b.c_str())); }
SF3Derror_t cleanSF3D() { return ERR(sf3d_clean()); }
SF3Derror_t closeLog() { return ERR(sf3d_close_log()); }
SF3Derror_t initializeHeatFlag(heatFluxSaveMode_t m, bool adv, bool lat) { return ERR(sf3d_initialize_heat_flag(U8(m), adv, lat)); }
u32_t setThreadsNumber(u32_t n) { return sf3d_set_threads_number(n); }
void setUseLineal(bool v) { sf3d_set_use_lineal(v); }
void setLinealMethod(int v) { sf3d_set_lineal_method(v); }

SF3Derror_t setSoilProperties(u16_t a, u8_t b, double c, double d, double e, double f, double g, double h, double i,
                              double j, double k, double l)
{ return ERR(sf3d_set_soil_properties(a, b, c, d, e, f, g, h, i, j, k, l)); }
SF3Derror_t setSurfaceProperties(u16_t i, double r) { return ERR(sf3d_set_surface_properties(i, r)); }
SF3Derror_t setNumericalParameters(double a, double b, u16_t c, u16_t d, u8_t e, u8_t f)
{ return ERR(sf3d_set_numerical_parameters(a, b, c, d, e, f)); }
SF3Derror_t setHydraulicProperties(WRCModel a, meanType_t b, float c) { return ERR(sf3d_set_hydraulic_properties(U8(a), U8(b), c)); }

SF3Derror_t setCulvert(SF3Duint_t i, double r, double s, double w, double h) { return ERR(sf3d_set_culvert(i, r, s, w, h)); }
SF3Derror_t setNode(SF3Duint_t i, double x, double y, double z, double v, bool s, boundaryType_t b, double sl, double ba)
{ return ERR(sf3d_set_node(i, x, y, z, v, s, U8(b), sl, ba)); }
SF3Derror_t setNodeLink(SF3Duint_t i, SF3Duint_t j, linkType_t d, double a) { return ERR(sf3d_set_node_link(i, j, U8(d), a)); }
SF3Derror_t setNodeBoundary(SF3Duint_t i, boundaryType_t b, double s, double a) { return ERR(sf3d_set_node_boundary(i, U8(b), s, a)); }
SF3Derror_t setNodeSoil(SF3Duint_t i, u16_t s, u16_t h) { return ERR(sf3d_set_node_soil(i, s, h)); }
SF3Derror_t setNodeSurface(SF3Duint_t i, u16_t s) { return ERR(sf3d_set_node_surface(i, s)); }

SF3Derror_t setNodePond(SF3Duint_t i, double v) { return ERR(sf3d_set_node_pond(i, v)); }
SF3Derror_t setNodeWaterContent(SF3Duint_t i, double v) { return ERR(sf3d_set_node_water_content(i, v)); }
SF3Derror_t setNodeDegreeOfSaturation(SF3Duint_t i, double v) { return ERR(sf3d_set_node_degree_of_saturation(i, v)); }
SF3Derror_t setNodeMatricPotential(SF3Duint_t i, double v) { return ERR(sf3d_set_node_matric_potential(i, v)); }
SF3Derror_t setNodeTotalPotential(SF3Duint_t i, double v) { return ERR(sf3d_set_node_total_potential(i, v)); }
SF3Derror_t setNodeWaterSinkSource(SF3Duint_t i, double v) { return ERR(sf3d_set_node_water_sink_source(i, v)); }
SF3Derror_t setNodePrescribedTotalPotential(SF3Duint_t i, double v) { return ERR(sf3d_set_node_prescribed_total_potential(i, v)); }

double getNodeWaterContent(SF3Duint_t i) { return sf3d_get_node_water_content(i); }
double getNodeMaximumWaterContent(SF3Duint_t i) { return sf3d_get_node_maximum_water_content(i); }
double getNodeMinimumWaterContent(SF3Duint_t i) { return sf3d_get_node_minimum_water_content(i); }
double getNodeAvailableWaterContent(SF3Duint_t i) { return sf3d_get_node_available_water_content(i); }
double getNodeWaterDeficit(SF3Duint_t i, double fc) { return sf3d_get_node_water_deficit(i, fc); }
double getNodeDegreeOfSaturation(SF3Duint_t i) { return sf3d_get_node_degree_of_saturation(i); }
double getNodeWaterConductivity(SF3Duint_t i) { return sf3d_get_node_water_conductivity(i); }
double getNodeMatricPotential(SF3Duint_t i) { return sf3d_get_node_matric_potential(i); }
double getNodeTotalPotential(SF3Duint_t i) { return sf3d_get_node_total_potential(i); }
double getNodePond(SF3Duint_t i) { return sf3d_get_node_pond(i); }
double getNodeMaxWaterFlow(SF3Duint_t i, linkType_t d) { return sf3d_get_node_max_water_flow(i, U8(d)); }
double getNodeSumLateralWaterFlow(SF3Duint_t i) { return sf3d_get_node_sum_lateral_water_flow(i); }
double getNodeSumLateralWaterFlowIn(SF3Duint_t i) { return sf3d_get_node_sum_lateral_water_flow_in(i); }
double getNodeSumLateralWaterFlowOut(SF3Duint_t i) { return sf3d_get_node_sum_lateral_water_flow_out(i); }
double getNodeBoundaryWaterFlow(SF3Duint_t i) { return sf3d_get_node_boundary_water_flow(i); }
double getTotalBoundaryWaterFlow(boundaryType_t b) { return sf3d_get_total_boundary_water_flow(U8(b)); }
double getTotalWaterContent() { return sf3d_get_total_water_content(); }
double getWaterStorage() { return sf3d_get_water_storage(); }
double getWaterMBR() { return sf3d_get_water_mbr(); }

SF3Derror_t setNodeHeatSinkSource(SF3Duint_t i, double v) { return ERR(sf3d_set_node_heat_sink_source(i, v)); }
SF3Derror_t setNodeTemperature(SF3Duint_t i, double v) { return ERR(sf3d_set_node_temperature(i, v)); }
SF3Derror_t setNodeBoundaryFixedTemperature(SF3Duint_t i, double t, double d) { return ERR(sf3d_set_node_boundary_fixed_temperature(i, t, d)); }
SF3Derror_t setNodeBoundaryHeightWind(SF3Duint_t i, double v) { return ERR(sf3d_set_node_boundary_height_wind(i, v)); }
SF3Derror_t setNodeBoundaryHeightTemperature(SF3Duint_t i, double v) { return ERR(sf3d_set_node_boundary_height_temperature(i, v)); }
SF3Derror_t setNodeBoundaryNetIrradiance(SF3Duint_t i, double v) { return ERR(sf3d_set_node_boundary_net_irradiance(i, v)); }
SF3Derror_t setNodeBoundaryTemperature(SF3Duint_t i, double v) { return ERR(sf3d_set_node_boundary_temperature(i, v)); }
SF3Derror_t setNodeBoundaryRelativeHumidity(SF3Duint_t i, double v) { return ERR(sf3d_set_node_boundary_relative_humidity(i, v)); }
SF3Derror_t setNodeBoundaryRoughness(SF3Duint_t i, double v) { return ERR(sf3d_set_node_boundary_roughness(i, v)); }
SF3Derror_t setNodeBoundaryWindSpeed(SF3Duint_t i, double v) { return ERR(sf3d_set_node_boundary_wind_speed(i, v)); }

double getNodeTemperature(SF3Duint_t i) { return sf3d_get_node_temperature(i); }
double getNodeHeatConductivity(SF3Duint_t i) { return sf3d_get_node_heat_conductivity(i); }
double getNodeVapor(SF3Duint_t i) { return sf3d_get_node_vapor(i); }
double getNodeHeatStorage(SF3Duint_t i, double h) { return sf3d_get_node_heat_storage(i, h); }
double getNodeHeatMaxFlux(SF3Duint_t i, linkType_t d, fluxTypes_t f) { return sf3d_get_node_heat_max_flux(i, U8(d), U8(f)); }
double getNodeBoundaryAdvectiveFlux(SF3Duint_t i) { return sf3d_get_node_boundary_advective_flux(i); }
double getNodeBoundaryLatentFlux(SF3Duint_t i) { return sf3d_get_node_boundary_latent_flux(i); }
double getNodeBoundaryRadiativeFlux(SF3Duint_t i) { return sf3d_get_node_boundary_radiative_flux(i); }
double getNodeBoundarySensibleFlux(SF3Duint_t i) { return sf3d_get_node_boundary_sensible_flux(i); }
double getNodeBoundaryAerodynamicConductance(SF3Duint_t i) { return sf3d_get_node_boundary_aerodynamic_conductance(i); }
double getNodeBoundarySoilConductance(SF3Duint_t i) { return sf3d_get_node_boundary_soil_conductance(i); }
double getHeatMBR() { return sf3d_get_heat_mbr(); }
double getHeatMBE() { return sf3d_get_heat_mbe(); }

void computePeriod(double t) { sf3d_compute_period(t); }
double computeStep(double t) { return sf3d_compute_step(t); }

}}  // namespace soilFluxes3D::v2

#ifdef SF3D_WITH_LINEALIA_STUB
/* bin/CRITERIA3D/main.cpp:81 calls LinealiaLib::instance().load() (lineal/linealiaLib.h; defined by lineal/linealiaLib.cpp, which
 * leaves the link line together with the rest of agrolib/soilFluxes3D).  The MI355X library never dlopens the third-party
 * liblinealia: "not loaded" is the truthful answer, and the application then runs with useLineal = false
 * (src/project3D/project3D.cpp:53).  Only the members a caller outside the solver can reach are defined. */
#include "linealiaLib.h"
LinealiaLib& LinealiaLib::instance() { static LinealiaLib one; return one; }
LinealiaLib::LinealiaLib() : lib("liblinealia") {}
bool LinealiaLib::load() { return false; }
bool LinealiaLib::isLoaded() const { return false; }
#endif
