/*
 * soilFluxes3D_api.h - declarations of the reference's public C++ API surface
 * (namespace soilFluxes3D, inline namespace v2; agrolib/soilFluxes3D/soilFluxes3D.h:9-104 and the
 * enums of types.h:36-199), written out independently so the drop-in shim can be compiled and its
 * exported symbols checked where the reference tree is not available (GPU box, CI).
 * Only what takes part in name mangling is declared: enum names, their uint8 underlying type and
 * the 70 function signatures.  A CRITERIA3D maintainer compiles shim/sf3d_cxx_shim.cpp against
 * the reference's own soilFluxes3D.h instead - the mangled names are identical,
 * tests/test_abi.py::test_shim_exports_reference_symbols pins them to SURVEY.md App. E.
 */
#pragma once
#include <cstdint>
#include <string>

namespace soilFluxes3D { inline namespace v2 {

using SF3Duint_t = std::uint32_t;
using u8_t = std::uint8_t;
using u16_t = std::uint16_t;
using u32_t = std::uint32_t;

enum class meanType_t : u8_t { Arithmetic, Geometric, Logarithmic };
enum class SF3Derror_t : u8_t { SF3Dok, IndexError, MemoryError, TopographyError, BoundaryError,
                                MissingDataError, ParameterError, SolverError, FileError };
enum class boundaryType_t : u8_t { NoBoundary, Runoff, FreeDrainage, FreeLateralDrainage,
                                   PrescribedTotalWaterPotential, Urban, Road, Culvert, HeatSurface, SoluteFlux };
enum class linkType_t : u8_t { NoLink, Up, Down, Lateral };
enum class WRCModel : u8_t { VanGenuchten, ModifiedVanGenuchten, Campbell };
enum class heatFluxSaveMode_t : std::uint8_t { None, Total, All };
enum class fluxTypes_t : u8_t { HeatTotal, HeatDiffusive, HeatLatentIsothermal, HeatLatentThermal, HeatAdvective,
                                WaterLiquidIsothermal, WaterLiquidThermal, WaterVaporIsothermal, WaterVaporThermal };

SF3Derror_t initializeSF3D(SF3Duint_t nrNodes, SF3Duint_t nrSurfaceNodes, u8_t nrLateralLinks, bool isComputeWater,
                           bool isComputeHeat, bool isComputeSolutes, heatFluxSaveMode_t HFsm = heatFluxSaveMode_t::None);
SF3Derror_t initializeBalance();
SF3Derror_t initializeLog(const std::string& logPath, const std::string& projectName);
SF3Derror_t cleanSF3D();
SF3Derror_t closeLog();
SF3Derror_t initializeHeatFlag(heatFluxSaveMode_t saveModeHeat, bool isComputeAdvectiveFlux, bool isComputeLatentHeat);
u32_t setThreadsNumber(u32_t nrThreads);
void setUseLineal(bool value);
void setLinealMethod(int value);

SF3Derror_t setSoilProperties(u16_t nrSoil, u8_t nrHorizon, double VG_alpha, double VG_n, double VG_m, double VG_he,
                              double thetaR, double thetaS, double kSat, double MualemL, double organicMatter, double clay);
SF3Derror_t setSurfaceProperties(u16_t surfaceIndex, double roughness);
SF3Derror_t setNumericalParameters(double minDeltaT, double maxDeltaT, u16_t maxIterationNumber,
                                   u16_t maxApproximationsNumber, u8_t ResidualToleranceExponent, u8_t MBRThresholdExponent);
SF3Derror_t setHydraulicProperties(WRCModel waterRetentionCurve, meanType_t conductivityMeanType, float conductivityHorizVertRatio);

SF3Derror_t setCulvert(SF3Duint_t nodeIndex, double roughness, double slope, double width, double height);
SF3Derror_t setNode(SF3Duint_t index, double x, double y, double z, double volume_or_area, bool isSurface,
                    boundaryType_t boundaryType, double slope = 0, double boundaryArea = 0);
SF3Derror_t setNodeLink(SF3Duint_t nodeIndex, SF3Duint_t linkIndex, linkType_t direction, double interfaceArea);
SF3Derror_t setNodeBoundary(SF3Duint_t nodeIndex, boundaryType_t boundaryType, double slope, double boundaryArea);
SF3Derror_t setNodeSoil(SF3Duint_t nodeIndex, u16_t soilIndex, u16_t horizonIndex);
SF3Derror_t setNodeSurface(SF3Duint_t nodeIndex, u16_t surfaceIndex);

SF3Derror_t setNodePond(SF3Duint_t nodeIndex, double pond);
SF3Derror_t setNodeWaterContent(SF3Duint_t nodeIndex, double waterContent);
SF3Derror_t setNodeDegreeOfSaturation(SF3Duint_t nodeIndex, double degreeOfSaturation);
SF3Derror_t setNodeMatricPotential(SF3Duint_t nodeIndex, double matricPotential);
SF3Derror_t setNodeTotalPotential(SF3Duint_t nodeIndex, double totalPotential);
SF3Derror_t setNodeWaterSinkSource(SF3Duint_t nodeIndex, double waterSinkSource);
SF3Derror_t setNodePrescribedTotalPotential(SF3Duint_t nodeIndex, double prescribedTotalPotential);

double getNodeWaterContent(SF3Duint_t nodeIndex);
double getNodeMaximumWaterContent(SF3Duint_t nodeIndex);
double getNodeMinimumWaterContent(SF3Duint_t nodeIndex);
double getNodeAvailableWaterContent(SF3Duint_t nodeIndex);
double getNodeWaterDeficit(SF3Duint_t nodeIndex, double fieldCapacity);
double getNodeDegreeOfSaturation(SF3Duint_t nodeIndex);
double getNodeWaterConductivity(SF3Duint_t nodeIndex);
double getNodeMatricPotential(SF3Duint_t nodeIndex);
double getNodeTotalPotential(SF3Duint_t nodeIndex);
double getNodePond(SF3Duint_t nodeIndex);
double getNodeMaxWaterFlow(SF3Duint_t nodeIndex, linkType_t linkDirection);
double getNodeSumLateralWaterFlow(SF3Duint_t nodeIndex);
double getNodeSumLateralWaterFlowIn(SF3Duint_t nodeIndex);
double getNodeSumLateralWaterFlowOut(SF3Duint_t nodeIndex);
double getNodeBoundaryWaterFlow(SF3Duint_t nodeIndex);
double getTotalBoundaryWaterFlow(boundaryType_t boundaryType);
double getTotalWaterContent();
double getWaterStorage();
double getWaterMBR();

SF3Derror_t setNodeHeatSinkSource(SF3Duint_t nodeIndex, double heatSinkSource);
SF3Derror_t setNodeTemperature(SF3Duint_t nodeIndex, double temperature);
SF3Derror_t setNodeBoundaryFixedTemperature(SF3Duint_t nodeIndex, double fixedTemperature, double depth);
SF3Derror_t setNodeBoundaryHeightWind(SF3Duint_t nodeIndex, double heightWind);
SF3Derror_t setNodeBoundaryHeightTemperature(SF3Duint_t nodeIndex, double heightTemperature);
SF3Derror_t setNodeBoundaryNetIrradiance(SF3Duint_t nodeIndex, double netIrradiance);
SF3Derror_t setNodeBoundaryTemperature(SF3Duint_t nodeIndex, double temperature);
SF3Derror_t setNodeBoundaryRelativeHumidity(SF3Duint_t nodeIndex, double relativeHumidity);
SF3Derror_t setNodeBoundaryRoughness(SF3Duint_t nodeIndex, double roughness);
SF3Derror_t setNodeBoundaryWindSpeed(SF3Duint_t nodeIndex, double windSpeed);

double getNodeTemperature(SF3Duint_t nodeIndex);
double getNodeHeatConductivity(SF3Duint_t nodeIndex);
double getNodeVapor(SF3Duint_t nodeIndex);
double getNodeHeatStorage(SF3Duint_t nodeIndex, double h);
double getNodeHeatMaxFlux(SF3Duint_t nodeIndex, linkType_t linkDirection, fluxTypes_t fluxType);
double getNodeBoundaryAdvectiveFlux(SF3Duint_t nodeIndex);
double getNodeBoundaryLatentFlux(SF3Duint_t nodeIndex);
double getNodeBoundaryRadiativeFlux(SF3Duint_t nodeIndex);
double getNodeBoundarySensibleFlux(SF3Duint_t nodeIndex);
double getNodeBoundaryAerodynamicConductance(SF3Duint_t nodeIndex);
double getNodeBoundarySoilConductance(SF3Duint_t nodeIndex);
double getHeatMBR();
double getHeatMBE();

void computePeriod(double timePeriod);
double computeStep(double maxTimeStep);

}}  // namespace soilFluxes3D::v2
