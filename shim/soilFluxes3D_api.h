/*
 * soilFluxes3D_api.h - name-mangling-relevant declarations of the reference's public C++ API
 * (namespace soilFluxes3D, inline namespace v2; agrolib/soilFluxes3D/soilFluxes3D.h:9-104, enums of
 * types.h:36-199), written as a generated-style table so the drop-in shim can be compiled, and its
 * exported symbols checked, where the reference tree is not available (GPU box, CI).
 * Only what takes part in Itanium mangling matters: namespace, enum type names with their uint8
 * underlying type, function names and parameter TYPES.  A CRITERIA3D maintainer compiles
 * shim/sf3d_cxx_shim.cpp against the reference's own header instead; the resulting
 * symbols are identical - tests/test_abi.py pins them to the list of SURVEY.md App. E.
 */
#pragma once
#include <cstdint>
#include <string>

namespace soilFluxes3D {
inline namespace v2 {

typedef std::uint32_t SF3Duint_t;
typedef std::uint8_t u8_t;
typedef std::uint16_t u16_t;
typedef std::uint32_t u32_t;

/* enumerations: explicit values, same order as types.h */
enum class SF3Derror_t : u8_t {
    SF3Dok = 0,
    IndexError = 1,
    MemoryError = 2,
    TopographyError = 3,
    BoundaryError = 4,
    MissingDataError = 5,
    ParameterError = 6,
    SolverError = 7,
    FileError = 8
};
enum class boundaryType_t : u8_t {
    NoBoundary = 0,
    Runoff = 1,
    FreeDrainage = 2,
    FreeLateralDrainage = 3,
    PrescribedTotalWaterPotential = 4,
    Urban = 5,
    Road = 6,
    Culvert = 7,
    HeatSurface = 8,
    SoluteFlux = 9
};
enum class linkType_t : u8_t { NoLink = 0, Up = 1, Down = 2, Lateral = 3 };
enum class WRCModel : u8_t { VanGenuchten = 0, ModifiedVanGenuchten = 1, Campbell = 2 };
enum class meanType_t : u8_t { Arithmetic = 0, Geometric = 1, Logarithmic = 2 };
enum class heatFluxSaveMode_t : std::uint8_t { None = 0, Total = 1, All = 2 };
enum class fluxTypes_t : u8_t {
    HeatTotal = 0,
    HeatDiffusive = 1,
    HeatLatentIsothermal = 2,
    HeatLatentThermal = 3,
    HeatAdvective = 4,
    WaterLiquidIsothermal = 5,
    WaterLiquidThermal = 6,
    WaterVaporIsothermal = 7,
    WaterVaporThermal = 8
};

/* ---- declaration generators ---------------------------------------------------------------- */
#define SF3D_STATUS0(fn) SF3Derror_t fn();
#define SF3D_NODE_SET1(fn) SF3Derror_t fn(SF3Duint_t node, double value);
#define SF3D_NODE_GET(fn) double fn(SF3Duint_t node);
#define SF3D_SCALAR_GET(fn) double fn();

/* life cycle */
SF3D_STATUS0(initializeBalance)
SF3D_STATUS0(cleanSF3D)
SF3D_STATUS0(closeLog)
SF3Derror_t initializeSF3D(SF3Duint_t n, SF3Duint_t nSurface, u8_t nLateral, bool water, bool heat, bool solutes,
                           heatFluxSaveMode_t mode = heatFluxSaveMode_t::None);
SF3Derror_t initializeLog(const std::string& path, const std::string& project);
SF3Derror_t initializeHeatFlag(heatFluxSaveMode_t mode, bool advective, bool latent);
u32_t setThreadsNumber(u32_t n);
void setUseLineal(bool on);
void setLinealMethod(int method);

/* classes and solver parameters */
SF3Derror_t setSoilProperties(u16_t soil, u8_t horizon, double alpha, double n, double m, double he, double thetaR,
                              double thetaS, double kSat, double L, double organicMatter, double clay);
SF3Derror_t setSurfaceProperties(u16_t surface, double roughness);
SF3Derror_t setNumericalParameters(double dtMin, double dtMax, u16_t maxIterations, u16_t maxApproximations,
                                   u8_t residualExponent, u8_t mbrExponent);
SF3Derror_t setHydraulicProperties(WRCModel curve, meanType_t mean, float horizVertRatio);

/* topology */
SF3Derror_t setNode(SF3Duint_t node, double x, double y, double z, double size, bool surface, boundaryType_t boundary,
                    double slope = 0, double boundaryArea = 0);
SF3Derror_t setNodeLink(SF3Duint_t node, SF3Duint_t linked, linkType_t direction, double area);
SF3Derror_t setNodeBoundary(SF3Duint_t node, boundaryType_t boundary, double slope, double area);
SF3Derror_t setCulvert(SF3Duint_t node, double roughness, double slope, double width, double height);
SF3Derror_t setNodeSoil(SF3Duint_t node, u16_t soil, u16_t horizon);
SF3Derror_t setNodeSurface(SF3Duint_t node, u16_t surface);

/* water state: (node, value) setters */
SF3D_NODE_SET1(setNodePond)
SF3D_NODE_SET1(setNodeWaterContent)
SF3D_NODE_SET1(setNodeDegreeOfSaturation)
SF3D_NODE_SET1(setNodeMatricPotential)
SF3D_NODE_SET1(setNodeTotalPotential)
SF3D_NODE_SET1(setNodeWaterSinkSource)
SF3D_NODE_SET1(setNodePrescribedTotalPotential)

/* water state: per-node getters */
SF3D_NODE_GET(getNodeWaterContent)
SF3D_NODE_GET(getNodeMaximumWaterContent)
SF3D_NODE_GET(getNodeMinimumWaterContent)
SF3D_NODE_GET(getNodeAvailableWaterContent)
SF3D_NODE_GET(getNodeDegreeOfSaturation)
SF3D_NODE_GET(getNodeWaterConductivity)
SF3D_NODE_GET(getNodeMatricPotential)
SF3D_NODE_GET(getNodeTotalPotential)
SF3D_NODE_GET(getNodePond)
SF3D_NODE_GET(getNodeSumLateralWaterFlow)
SF3D_NODE_GET(getNodeSumLateralWaterFlowIn)
SF3D_NODE_GET(getNodeSumLateralWaterFlowOut)
SF3D_NODE_GET(getNodeBoundaryWaterFlow)
double getNodeWaterDeficit(SF3Duint_t node, double fieldCapacity);
double getNodeMaxWaterFlow(SF3Duint_t node, linkType_t direction);
double getTotalBoundaryWaterFlow(boundaryType_t boundary);
SF3D_SCALAR_GET(getTotalWaterContent)
SF3D_SCALAR_GET(getWaterStorage)
SF3D_SCALAR_GET(getWaterMBR)

/* heat: (node, value) setters */
SF3D_NODE_SET1(setNodeHeatSinkSource)
SF3D_NODE_SET1(setNodeTemperature)
SF3D_NODE_SET1(setNodeBoundaryHeightWind)
SF3D_NODE_SET1(setNodeBoundaryHeightTemperature)
SF3D_NODE_SET1(setNodeBoundaryNetIrradiance)
SF3D_NODE_SET1(setNodeBoundaryTemperature)
SF3D_NODE_SET1(setNodeBoundaryRelativeHumidity)
SF3D_NODE_SET1(setNodeBoundaryRoughness)
SF3D_NODE_SET1(setNodeBoundaryWindSpeed)
SF3Derror_t setNodeBoundaryFixedTemperature(SF3Duint_t node, double temperature, double depth);

/* heat: getters */
SF3D_NODE_GET(getNodeTemperature)
SF3D_NODE_GET(getNodeHeatConductivity)
SF3D_NODE_GET(getNodeVapor)
SF3D_NODE_GET(getNodeBoundaryAdvectiveFlux)
SF3D_NODE_GET(getNodeBoundaryLatentFlux)
SF3D_NODE_GET(getNodeBoundaryRadiativeFlux)
SF3D_NODE_GET(getNodeBoundarySensibleFlux)
SF3D_NODE_GET(getNodeBoundaryAerodynamicConductance)
SF3D_NODE_GET(getNodeBoundarySoilConductance)
double getNodeHeatStorage(SF3Duint_t node, double h);
double getNodeHeatMaxFlux(SF3Duint_t node, linkType_t direction, fluxTypes_t flux);
SF3D_SCALAR_GET(getHeatMBR)
SF3D_SCALAR_GET(getHeatMBE)

/* time stepping */
void computePeriod(double seconds);
double computeStep(double maxSeconds);

#undef SF3D_STATUS0
#undef SF3D_NODE_SET1
#undef SF3D_NODE_GET
#undef SF3D_SCALAR_GET

}  // inline namespace v2
}  // namespace soilFluxes3D
