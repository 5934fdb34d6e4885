/*
 * sf3d.h - flat C ABI of the MI355X-native soilFluxes3D time-step library.
 *
 * This is the drop-in boundary.  Every entry point below replaces one
 * function of the reference's public API `namespace soilFluxes3D::v2`
 * (agrolib/soilFluxes3D/soilFluxes3D.h:9-104); the citation after each prototype is the
 * reference declaration (header line) and definition (soilFluxes3D.cpp line) it stands in
 * for.  Argument order, meaning, units, validation rules and return/sentinel values are the
 * reference's.  Only plain scalars and caller-owned arrays cross the boundary; the library
 * owns all solver memory.  One process-global model instance, not re-entrant (as the
 * reference).
 *
 * Three shared libraries export this same ABI:
 *   criteria3d_amd/csrc/libsf3d_hip.so   the product: HIP/gfx950 kernels (no CPU fallback)
 *   oracle/libsf3d_oracle.so              CPU restatement of the reference algorithm (tests only)
 *   oracle/_ref/libsf3d_ref.so            the unmodified reference sources behind a thin
 *                                         forwarding wrapper (tests / cpu_baseline only)
 *
 * `shim/sf3d_cxx_shim.cpp` re-exports the product under the reference's 70 C++ mangled
 * names so bin/CRITERIA3D links it unchanged.
 */
#ifndef SF3D_H
#define SF3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (all uint8-backed, values identical to types.h) ------------------------- */

/* SF3Derror_t, types.h:39-40 */
typedef uint8_t sf3d_error_t;
enum {
    SF3D_OK = 0, SF3D_INDEX_ERROR = 1, SF3D_MEMORY_ERROR = 2, SF3D_TOPOGRAPHY_ERROR = 3,
    SF3D_BOUNDARY_ERROR = 4, SF3D_MISSING_DATA_ERROR = 5, SF3D_PARAMETER_ERROR = 6,
    SF3D_SOLVER_ERROR = 7, SF3D_FILE_ERROR = 8
};

/* boundaryType_t, types.h:98-99 */
typedef uint8_t sf3d_boundary_t;
enum {
    SF3D_BND_NONE = 0, SF3D_BND_RUNOFF = 1, SF3D_BND_FREE_DRAINAGE = 2,
    SF3D_BND_FREE_LATERAL_DRAINAGE = 3, SF3D_BND_PRESCRIBED_TOTAL_POTENTIAL = 4,
    SF3D_BND_URBAN = 5, SF3D_BND_ROAD = 6, SF3D_BND_CULVERT = 7, SF3D_BND_HEAT_SURFACE = 8,
    SF3D_BND_SOLUTE_FLUX = 9
};

/* linkType_t, types.h:101 */
typedef uint8_t sf3d_link_t;
enum { SF3D_LINK_NONE = 0, SF3D_LINK_UP = 1, SF3D_LINK_DOWN = 2, SF3D_LINK_LATERAL = 3 };

/* WRCModel, types.h:135 */
typedef uint8_t sf3d_wrc_t;
enum { SF3D_WRC_VAN_GENUCHTEN = 0, SF3D_WRC_MODIFIED_VAN_GENUCHTEN = 1, SF3D_WRC_CAMPBELL = 2 };

/* meanType_t, types.h:36 */
typedef uint8_t sf3d_mean_t;
enum { SF3D_MEAN_ARITHMETIC = 0, SF3D_MEAN_GEOMETRIC = 1, SF3D_MEAN_LOGARITHMIC = 2 };

/* heatFluxSaveMode_t, types.h:186 ; fluxTypes_t, types.h:199 */
typedef uint8_t sf3d_heat_save_t;   /* 0 None, 1 Total, 2 All */
typedef uint8_t sf3d_flux_t;        /* 0 HeatTotal ... 8 WaterVaporThermal */

/* getter sentinels, types.h:42-64 + commonConstants.h:95-100 */
#define SF3D_VAL_INDEX_ERROR        (-1111.0)
#define SF3D_VAL_MEMORY_ERROR       (-2222.0)
#define SF3D_VAL_TOPOGRAPHY_ERROR   (-3333.0)
#define SF3D_VAL_BOUNDARY_ERROR     (-4444.0)
#define SF3D_VAL_MISSING_DATA_ERROR (-9999.0)
#define SF3D_VAL_PARAMETER_ERROR    (-7777.0)
#define SF3D_NODATA                 (-9999.0)

/* ---- initialisation and memory ------------------------------------------------------ */

/* initializeSF3D            soilFluxes3D.h:9   soilFluxes3D.cpp:49-178 */
sf3d_error_t sf3d_initialize(uint32_t nrNodes, uint32_t nrSurfaceNodes, uint8_t nrLateralLinks,
                             int isComputeWater, int isComputeHeat, int isComputeSolutes,
                             sf3d_heat_save_t heatSaveMode);
/* initializeBalance         soilFluxes3D.h:13  soilFluxes3D.cpp:184-197, water.cpp:35-65 */
sf3d_error_t sf3d_initialize_balance(void);
/* initializeLog             soilFluxes3D.h:14  soilFluxes3D.cpp:203-212 (no-op unless MCR) */
sf3d_error_t sf3d_initialize_log(const char* logPath, const char* projectName);
/* cleanSF3D                 soilFluxes3D.h:16  soilFluxes3D.cpp:218-304 */
sf3d_error_t sf3d_clean(void);
/* closeLog                  soilFluxes3D.h:17  soilFluxes3D.cpp:310-319 */
sf3d_error_t sf3d_close_log(void);
/* initializeHeatFlag        soilFluxes3D.h:19  soilFluxes3D.cpp:325-332 */
sf3d_error_t sf3d_initialize_heat_flag(sf3d_heat_save_t saveMode, int computeAdvectiveFlux,
                                       int computeLatentHeat);
/* setThreadsNumber          soilFluxes3D.h:21  soilFluxes3D.cpp:340-361 */
uint32_t sf3d_set_threads_number(uint32_t nrThreads);
/* setUseLineal              soilFluxes3D.h:22  soilFluxes3D.cpp:367-371 */
void sf3d_set_use_lineal(int value);
/* setLinealMethod           soilFluxes3D.h:23  soilFluxes3D.cpp:376-380 */
void sf3d_set_lineal_method(int value);

/* ---- soil / surface classes ---------------------------------------------------------- */

/* setSoilProperties         soilFluxes3D.h:26-28  soilFluxes3D.cpp:395-449 */
sf3d_error_t sf3d_set_soil_properties(uint16_t nrSoil, uint8_t nrHorizon, double VG_alpha,
                                      double VG_n, double VG_m, double VG_he, double thetaR,
                                      double thetaS, double kSat, double MualemL,
                                      double organicMatter, double clay);
/* setSurfaceProperties      soilFluxes3D.h:29  soilFluxes3D.cpp:457-467 */
sf3d_error_t sf3d_set_surface_properties(uint16_t surfaceIndex, double roughness);

/* ---- solver parameters ---------------------------------------------------------------- */

/* setNumericalParameters    soilFluxes3D.h:32  soilFluxes3D.cpp:474-520 */
sf3d_error_t sf3d_set_numerical_parameters(double minDeltaT, double maxDeltaT,
                                           uint16_t maxIterationNumber,
                                           uint16_t maxApproximationsNumber,
                                           uint8_t residualToleranceExponent,
                                           uint8_t MBRThresholdExponent);
/* setHydraulicProperties    soilFluxes3D.h:33  soilFluxes3D.cpp:531-548 */
sf3d_error_t sf3d_set_hydraulic_properties(sf3d_wrc_t waterRetentionCurve,
                                           sf3d_mean_t conductivityMeanType,
                                           float conductivityHorizVertRatio);

/* ---- topology -------------------------------------------------------------------------- */

/* setCulvert                soilFluxes3D.h:36  soilFluxes3D.cpp:551-588
 * (the reference dereferences a never-allocated array here; this ABI returns
 *  SF3D_BOUNDARY_ERROR = "unsupported") */
sf3d_error_t sf3d_set_culvert(uint32_t nodeIndex, double roughness, double slope, double width,
                              double height);
/* setNode                   soilFluxes3D.h:37  soilFluxes3D.cpp:595-629 */
sf3d_error_t sf3d_set_node(uint32_t index, double x, double y, double z, double volume_or_area,
                           int isSurface, sf3d_boundary_t boundaryType, double slope,
                           double boundaryArea);
/* setNodeLink               soilFluxes3D.h:38  soilFluxes3D.cpp:636-683 */
sf3d_error_t sf3d_set_node_link(uint32_t nodeIndex, uint32_t linkIndex, sf3d_link_t direction,
                                double interfaceArea);
/* setNodeBoundary           soilFluxes3D.h:39  soilFluxes3D.cpp:689-725 */
sf3d_error_t sf3d_set_node_boundary(uint32_t nodeIndex, sf3d_boundary_t boundaryType,
                                    double slope, double boundaryArea);
/* setNodeSoil               soilFluxes3D.h:42  soilFluxes3D.cpp:734-750 */
sf3d_error_t sf3d_set_node_soil(uint32_t nodeIndex, uint16_t soilIndex, uint16_t horizonIndex);
/* setNodeSurface            soilFluxes3D.h:43  soilFluxes3D.cpp:758-775 */
sf3d_error_t sf3d_set_node_surface(uint32_t nodeIndex, uint16_t surfaceIndex);

/* ---- water state setters --------------------------------------------------------------- */

/* setNodePond               soilFluxes3D.h:46  soilFluxes3D.cpp:783-796 */
sf3d_error_t sf3d_set_node_pond(uint32_t nodeIndex, double pond);
/* setNodeWaterContent       soilFluxes3D.h:47  soilFluxes3D.cpp:803-835 */
sf3d_error_t sf3d_set_node_water_content(uint32_t nodeIndex, double waterContent);
/* setNodeDegreeOfSaturation soilFluxes3D.h:48  soilFluxes3D.cpp:842-862 */
sf3d_error_t sf3d_set_node_degree_of_saturation(uint32_t nodeIndex, double degreeOfSaturation);
/* setNodeMatricPotential    soilFluxes3D.h:49  soilFluxes3D.cpp:869-884 */
sf3d_error_t sf3d_set_node_matric_potential(uint32_t nodeIndex, double matricPotential);
/* setNodeTotalPotential     soilFluxes3D.h:50  soilFluxes3D.cpp:891-906 */
sf3d_error_t sf3d_set_node_total_potential(uint32_t nodeIndex, double totalPotential);
/* setNodeWaterSinkSource    soilFluxes3D.h:51  soilFluxes3D.cpp:934-945 */
sf3d_error_t sf3d_set_node_water_sink_source(uint32_t nodeIndex, double waterSinkSource);
/* setNodePrescribedTotalPotential  soilFluxes3D.h:52  soilFluxes3D.cpp:913-927 */
sf3d_error_t sf3d_set_node_prescribed_total_potential(uint32_t nodeIndex,
                                                      double prescribedTotalPotential);

/* ---- water getters --------------------------------------------------------------------- */

/* getNodeWaterContent            soilFluxes3D.h:55  soilFluxes3D.cpp:951-961 */
double sf3d_get_node_water_content(uint32_t nodeIndex);
/* getNodeMaximumWaterContent     soilFluxes3D.h:56  soilFluxes3D.cpp:967-979 */
double sf3d_get_node_maximum_water_content(uint32_t nodeIndex);
/* getNodeMinimumWaterContent     soilFluxes3D.h:57  soilFluxes3D.cpp:986-998 */
double sf3d_get_node_minimum_water_content(uint32_t nodeIndex);
/* getNodeAvailableWaterContent   soilFluxes3D.h:58  soilFluxes3D.cpp:1005-1015 */
double sf3d_get_node_available_water_content(uint32_t nodeIndex);
/* getNodeWaterDeficit            soilFluxes3D.h:59  soilFluxes3D.cpp:1022-1034 */
double sf3d_get_node_water_deficit(uint32_t nodeIndex, double fieldCapacity);
/* getNodeDegreeOfSaturation      soilFluxes3D.h:60  soilFluxes3D.cpp:1041-1056 */
double sf3d_get_node_degree_of_saturation(uint32_t nodeIndex);
/* getNodeWaterConductivity       soilFluxes3D.h:61  soilFluxes3D.cpp:1062-1071 */
double sf3d_get_node_water_conductivity(uint32_t nodeIndex);
/* getNodeMatricPotential         soilFluxes3D.h:62  soilFluxes3D.cpp:1077-1086 */
double sf3d_get_node_matric_potential(uint32_t nodeIndex);
/* getNodeTotalPotential          soilFluxes3D.h:63  soilFluxes3D.cpp:1092-1101 */
double sf3d_get_node_total_potential(uint32_t nodeIndex);
/* getNodePond                    soilFluxes3D.h:64  soilFluxes3D.cpp:1107-1119 */
double sf3d_get_node_pond(uint32_t nodeIndex);
/* getNodeMaxWaterFlow            soilFluxes3D.h:65  soilFluxes3D.cpp:1126-1156 */
double sf3d_get_node_max_water_flow(uint32_t nodeIndex, sf3d_link_t linkDirection);
/* getNodeSumLateralWaterFlow     soilFluxes3D.h:66  soilFluxes3D.cpp:1162-1176 */
double sf3d_get_node_sum_lateral_water_flow(uint32_t nodeIndex);
/* getNodeSumLateralWaterFlowIn   soilFluxes3D.h:67  soilFluxes3D.cpp:1182-1196 */
double sf3d_get_node_sum_lateral_water_flow_in(uint32_t nodeIndex);
/* getNodeSumLateralWaterFlowOut  soilFluxes3D.h:68  soilFluxes3D.cpp:1202-1216 */
double sf3d_get_node_sum_lateral_water_flow_out(uint32_t nodeIndex);
/* getNodeBoundaryWaterFlow       soilFluxes3D.h:69  soilFluxes3D.cpp:1222-1234 */
double sf3d_get_node_boundary_water_flow(uint32_t nodeIndex);
/* getTotalBoundaryWaterFlow      soilFluxes3D.h:70  soilFluxes3D.cpp:1240-1250 */
double sf3d_get_total_boundary_water_flow(sf3d_boundary_t boundaryType);
/* getTotalWaterContent           soilFluxes3D.h:71  soilFluxes3D.cpp:1256-1259 */
double sf3d_get_total_water_content(void);
/* getWaterStorage                soilFluxes3D.h:72  soilFluxes3D.cpp:1265-1268 */
double sf3d_get_water_storage(void);
/* getWaterMBR                    soilFluxes3D.h:73  soilFluxes3D.cpp:1274-1277 */
double sf3d_get_water_mbr(void);

/* ---- heat setters (heat transport: heat.cpp, heatLoop cpusolver.cpp:471-605).
 * Where the reference would write through arrays it only allocates when isComputeHeat was requested
 * (soilFluxes3D.cpp:100-165) these return SF3D_MISSING_DATA_ERROR instead of crashing. -------------- */

/* setNodeHeatSinkSource             soilFluxes3D.h:76  soilFluxes3D.cpp:1283-1294 */
sf3d_error_t sf3d_set_node_heat_sink_source(uint32_t nodeIndex, double heatSinkSource);
/* setNodeTemperature                soilFluxes3D.h:77  soilFluxes3D.cpp:1300-1312 */
sf3d_error_t sf3d_set_node_temperature(uint32_t nodeIndex, double temperature);
/* setNodeBoundaryFixedTemperature   soilFluxes3D.h:78  soilFluxes3D.cpp:1318-1336 */
sf3d_error_t sf3d_set_node_boundary_fixed_temperature(uint32_t nodeIndex,
                                                      double fixedTemperature, double depth);
/* setNodeBoundaryHeightWind         soilFluxes3D.h:79  soilFluxes3D.cpp:1342-1356 */
sf3d_error_t sf3d_set_node_boundary_height_wind(uint32_t nodeIndex, double heightWind);
/* setNodeBoundaryHeightTemperature  soilFluxes3D.h:80  soilFluxes3D.cpp:1362-1376 */
sf3d_error_t sf3d_set_node_boundary_height_temperature(uint32_t nodeIndex, double heightTemperature);
/* setNodeBoundaryNetIrradiance      soilFluxes3D.h:81  soilFluxes3D.cpp:1382-1396 */
sf3d_error_t sf3d_set_node_boundary_net_irradiance(uint32_t nodeIndex, double netIrradiance);
/* setNodeBoundaryTemperature        soilFluxes3D.h:82  soilFluxes3D.cpp:1402-1416 */
sf3d_error_t sf3d_set_node_boundary_temperature(uint32_t nodeIndex, double temperature);
/* setNodeBoundaryRelativeHumidity   soilFluxes3D.h:83  soilFluxes3D.cpp:1422-1436 */
sf3d_error_t sf3d_set_node_boundary_relative_humidity(uint32_t nodeIndex, double relativeHumidity);
/* setNodeBoundaryRoughness          soilFluxes3D.h:84  soilFluxes3D.cpp:1442-1459 */
sf3d_error_t sf3d_set_node_boundary_roughness(uint32_t nodeIndex, double roughness);
/* setNodeBoundaryWindSpeed          soilFluxes3D.h:85  soilFluxes3D.cpp:1465-1482 */
sf3d_error_t sf3d_set_node_boundary_wind_speed(uint32_t nodeIndex, double windSpeed);

/* ---- heat getters ------------------------------------------------------------------------ */

/* getNodeTemperature                soilFluxes3D.h:88  soilFluxes3D.cpp:1488-1500 */
double sf3d_get_node_temperature(uint32_t nodeIndex);
/* getNodeHeatConductivity           soilFluxes3D.h:89  soilFluxes3D.cpp:1506-1519 */
double sf3d_get_node_heat_conductivity(uint32_t nodeIndex);
/* getNodeVapor                      soilFluxes3D.h:90  soilFluxes3D.cpp:1525-1540 */
double sf3d_get_node_vapor(uint32_t nodeIndex);
/* getNodeHeatStorage                soilFluxes3D.h:91  soilFluxes3D.cpp:1547-1570 */
double sf3d_get_node_heat_storage(uint32_t nodeIndex, double h);
/* getNodeHeatMaxFlux                soilFluxes3D.h:92  soilFluxes3D.cpp:1578-1611 */
double sf3d_get_node_heat_max_flux(uint32_t nodeIndex, sf3d_link_t linkDirection, sf3d_flux_t fluxType);
/* getNodeBoundaryAdvectiveFlux      soilFluxes3D.h:93  soilFluxes3D.cpp:1617-1632 */
double sf3d_get_node_boundary_advective_flux(uint32_t nodeIndex);
/* getNodeBoundaryLatentFlux         soilFluxes3D.h:94  soilFluxes3D.cpp:1638-1653 */
double sf3d_get_node_boundary_latent_flux(uint32_t nodeIndex);
/* getNodeBoundaryRadiativeFlux      soilFluxes3D.h:95  soilFluxes3D.cpp:1659-1674 */
double sf3d_get_node_boundary_radiative_flux(uint32_t nodeIndex);
/* getNodeBoundarySensibleFlux       soilFluxes3D.h:96  soilFluxes3D.cpp:1680-1695 */
double sf3d_get_node_boundary_sensible_flux(uint32_t nodeIndex);
/* getNodeBoundaryAerodynamicConductance  soilFluxes3D.h:97  soilFluxes3D.cpp:1701-1716 */
double sf3d_get_node_boundary_aerodynamic_conductance(uint32_t nodeIndex);
/* getNodeBoundarySoilConductance    soilFluxes3D.h:98  soilFluxes3D.cpp:1722-1737 */
double sf3d_get_node_boundary_soil_conductance(uint32_t nodeIndex);
/* getHeatMBR                        soilFluxes3D.h:99  soilFluxes3D.cpp:1743-1746 */
double sf3d_get_heat_mbr(void);
/* getHeatMBE                        soilFluxes3D.h:100 soilFluxes3D.cpp:1750-1753 */
double sf3d_get_heat_mbe(void);

/* ---- computation --------------------------------------------------------------------------- */

/* computePeriod   soilFluxes3D.h:103  soilFluxes3D.cpp:1760-1777 */
void sf3d_compute_period(double timePeriod);
/* computeStep     soilFluxes3D.h:104  soilFluxes3D.cpp:1785-1821  -> accepted dt [s] */
double sf3d_compute_step(double maxTimeStep);

/* ==================================================================================== */
/* Extensions (not in the reference API).  None of them changes what the 70 calls above  */
/* compute; they remove per-node call overhead and expose the counters  */
/* the algorithmic-bytes model needs.                                      */
/* ==================================================================================== */

/* "hip" | "oracle" | "reference" */
const char* sf3d_backend_name(void);

/* Bulk forms: element k applies the scalar call above to node first+k (same validation, the
 * first non-OK code is returned and processing stops there).  NULL optional arrays mean the
 * scalar default. */
sf3d_error_t sf3d_set_nodes(uint32_t first, uint32_t count, const double* x, const double* y,
                            const double* z, const double* volume_or_area,
                            const uint8_t* isSurface, const uint8_t* boundaryType,
                            const double* slope, const double* boundaryArea);
/* count link records: (node[k], linked[k], direction[k], area[k]) applied in array order */
sf3d_error_t sf3d_set_node_links(uint64_t count, const uint32_t* node, const uint32_t* linked,
                                 const uint8_t* direction, const double* interfaceArea);
sf3d_error_t sf3d_set_nodes_soil(uint32_t first, uint32_t count, const uint16_t* soilIndex,
                                 const uint16_t* horizonIndex);
sf3d_error_t sf3d_set_nodes_surface(uint32_t first, uint32_t count, const uint16_t* surfaceIndex);
sf3d_error_t sf3d_set_nodes_pond(uint32_t first, uint32_t count, const double* pond);
sf3d_error_t sf3d_set_nodes_matric_potential(uint32_t first, uint32_t count, const double* psi);
sf3d_error_t sf3d_set_nodes_total_potential(uint32_t first, uint32_t count, const double* H);
sf3d_error_t sf3d_set_nodes_water_sink_source(uint32_t first, uint32_t count, const double* q);
/* out[k] = getter(first+k) */
sf3d_error_t sf3d_get_nodes_total_potential(uint32_t first, uint32_t count, double* out);
sf3d_error_t sf3d_get_nodes_degree_of_saturation(uint32_t first, uint32_t count, double* out);
sf3d_error_t sf3d_get_nodes_water_content(uint32_t first, uint32_t count, double* out);
sf3d_error_t sf3d_get_nodes_water_conductivity(uint32_t first, uint32_t count, double* out);
sf3d_error_t sf3d_get_nodes_boundary_water_flow(uint32_t first, uint32_t count, double* out);
/* heat counterparts: setNodeTemperature / setNodeHeatSinkSource / getNodeTemperature over a range
 * (surface nodes answer getNodeTemperature's TopographyError value, soilFluxes3D.cpp:1496-1497) */
sf3d_error_t sf3d_set_nodes_temperature(uint32_t first, uint32_t count, const double* temperature);
sf3d_error_t sf3d_set_nodes_heat_sink_source(uint32_t first, uint32_t count, const double* q);
sf3d_error_t sf3d_get_nodes_temperature(uint32_t first, uint32_t count, double* out);
/* one atmospheric boundary field for a list of nodes = count calls of the scalar setter of
 * soilFluxes3D.h:79-85 (same validation; stops at the first error):
 * field 0 heightWind, 1 heightTemperature, 2 roughness, 3 temperature, 4 relativeHumidity,
 * 5 windSpeed, 6 netIrradiance */
sf3d_error_t sf3d_set_nodes_boundary_heat(int field, uint32_t count, const uint32_t* nodes, const double* values);

/* Work counters since sf3d_initialize (the same events SURVEY.md App. B instruments in
 * cpusolver.cpp): out[0] attempts (waterMainLoop iterations), [1] accepted steps,
 * [2] approximations, [3] Jacobi sweeps, [4] Courant rejections, [5] linear-solver failures,
 * [6] restore-best calls, [7] those of [4] that the HIP product's early Courant check decided before the full
 * approximation ran (an implementation detail of the product, 0 in the CPU libraries: the early check changes no
 * result and no other counter).  The "reference" backend cannot count (unmodified sources) and returns
 * SF3D_MISSING_DATA_ERROR. */
sf3d_error_t sf3d_get_counters(uint64_t out[8]);
/* How the Jacobi iterations counted in out[3] above were launched by the HIP product since sf3d_initialize: single sweeps (one
 * iteration per pass over the coefficients) and PAIRED passes (two iterations per pass; on a row strip of a multi-GPU
 * run: the pass + its boundary launch count as one).  Implementation detail of the product - no result depends on it; the CPU
 * libraries return SF3D_MISSING_DATA_ERROR. */
sf3d_error_t sf3d_get_sweep_launches(uint64_t* single_sweeps, uint64_t* paired_passes);
/* ... and RESIDENT loops: launches that made ALL Jacobi iterations of an approximation with the rows kept on chip (grids that fit:
 * csrc/sf3d_resident.inc; neither of the two counts above moves then).  Same status: an implementation detail, SF3D_MISSING_DATA_ERROR
 * from the CPU libraries. */
sf3d_error_t sf3d_get_resident_launches(uint64_t* resident_loops);
/* Work counters of the heat part since sf3d_initialize (heatLoop, cpusolver.cpp:471-605; computeStep's heat loop,
 * soilFluxes3D.cpp:1802-1818): out[0] heat steps accepted (heatLoop returned true), [1] heat steps halved (|MBR| > 1: false),
 * [2] reductions of dtHeat by the boundary Courant rule (updateBoundaryHeatData returned false, heat.cpp:329-339),
 * [3] sweeps of the linear solver (Gauss-Seidel in the reference and the oracle, Jacobi in the HIP product: not comparable).
 * The "reference" backend cannot count and returns SF3D_MISSING_DATA_ERROR. */
sf3d_error_t sf3d_get_heat_counters(uint64_t out[4]);

/* Stopping quantity of the LAST linear solve of the water system: Jacobi - the mean scaled update of the last sweep
 * (JacobiWaterCPU's norm, water.cpp:592-600); conjugate gradients (the linealia stand-in) - the relative residual
 * ||r~|| / ||b~|| of the normalised system.  -9999 where it is not tracked (the unmodified reference). */
double sf3d_get_linear_residual(void);
/* current adaptive time step deltaTcurr [s] (Solver::getTimeStep, solver.h:36) */
double sf3d_get_time_step(void);
/* Restore the adaptive time step of a checkpointed run (the reference keeps deltaTcurr only in
 * memory - SURVEY.md 5 "deltaTcurr is not checkpointed" - although Solver::setTimeStep exists,
 * solver.h:77-86).  Together with sf3d_get_time_step and the state setters/getters this resumes a
 * simulation on the exact trajectory.  The value is clamped to [deltaTmin, deltaTmax]. */
sf3d_error_t sf3d_set_time_step(double deltaT);
/* Put every persistent solver parameter back to its fresh-process default (SolverParameters,
 * types.h:291-315: deltaTcurr = NODATA -> deltaTmax at the next initialize, deltaTmax = 600, ...).
 * The reference keeps them across re-initialisations (global CPUSolverObject, SURVEY.md 8a
 * quirk 4), so two models run in one process influence each other; tests and benchmarks call
 * this before sf3d_initialize to start exactly like a new process.
 * "reference" backend: SF3D_MISSING_DATA_ERROR (not reachable through its public header). */
sf3d_error_t sf3d_reset_solver_state(void);

/* Declare how many of the first node indices are surface nodes when that was not known at
 * sf3d_initialize time (the retired v1 API, old/old_soilFluxes3D.h:16, has no such argument; the
 * v1 alias layer counts the isSurface flags of setNode and calls this before the first step). */
sf3d_error_t sf3d_set_surface_nodes_number(uint32_t nrSurfaceNodes);

/* ---- device-side instrumentation (product backend only; others return MISSING_DATA) ------ */

/* Select the HIP device for this process before sf3d_initialize (default: LOCAL_RANK or 0). */
sf3d_error_t sf3d_set_device(int device);
/* Upload every pending host-side edit (sinks, ponds, state, parameters) to the device and block
 * until all queued device work of the solver stream has finished. */
sf3d_error_t sf3d_synchronize(void);
/* Per-kernel HIP-event timing on the solver's own stream.  mode 1 records an event pair around
 * every launch of the kernels listed by sf3d_kernel_name(), mode 2 only around the Jacobi sweeps of
 * every 8th computeStep (the dominant kernel, sampled: ~1 % overhead), mode 0 stops.  Guarded launches that did no work are not counted. */
sf3d_error_t sf3d_kernel_timing(int mode);
/* number of instrumented kernels; name of kernel k (NULL if out of range) */
int          sf3d_kernel_count(void);
const char*  sf3d_kernel_name(int k);
/* launches, total milliseconds and nodes processed per launch of kernel k since timing was
 * enabled (drains the event pool) */
sf3d_error_t sf3d_kernel_stats(int k, uint64_t* launches, double* total_ms, uint64_t* nodes_per_launch);
/* Which elementary functions (log / exp / pow / cbrt) the kernels evaluate: 1 = the operations of the reference build's C library
 * (glibc 2.35: soilPhysics.cpp:68-279, otherFunctions.cpp:35, water.cpp:389-469, heat.cpp:702-845 call it), bit for bit - the default build;
 * 0 = the library's own 0.50-ulp table routines of earlier rounds (a build option, -DSF3D_LIBM_GLIBC=0: DESIGN.md 4). */
int sf3d_libm_set(void);
/* Test hooks: the logarithm the link kernels use for the logarithmic mean (a table-driven routine) evaluated on
 * the device for `count` host values; tests compare it bit for bit with the host build of the same source. */
sf3d_error_t sf3d_device_log(uint32_t count, const double* x, double* out);
/* the same for the exp of the heat kernels */
sf3d_error_t sf3d_device_exp(uint32_t count, const double* x, double* out);
/* the same for the cbrt of the runoff links' Manning term (x >= 0) */
sf3d_error_t sf3d_device_cbrt(uint32_t count, const double* x, double* out);
/* the norm of a Jacobi sweep (JacobiWaterCPU, water.cpp:592-600: a sum of N terms) as the sweep kernels add it: `count` host terms summed on
 * the device over `blocks` workgroups in k_sweep's association (0) or the paired passes' (1); out[0] = the double-double sum rounded once
 * (the same double for every association and block count: tests hold it against an exactly rounded sum), out[1] = the same association
 * in plain doubles (differs from one association to the next - what the double-double is there for) */
sf3d_error_t sf3d_device_norm_sum(uint32_t count, const double* x, uint32_t blocks, int association, double* out);
/* the same for the pow of the soil-property kernels: out[k] = x[k]^y[k], x >= 0 */
sf3d_error_t sf3d_device_pow(uint32_t count, const double* x, const double* y, double* out);

/* ---- multi-GPU: one process per GPU, row strips of surface-cell columns ------
 * Every rank builds the SAME global model through the setters above; the library uploads and computes only
 * the strip it owns (+ one-cell halo, renumbered locally) and exchanges halos with its neighbours device-to-device (HIP-IPC windows
 * over xGMI, device-side flags).  Call order: sf3d_dist_prepare(rank, world) -> sf3d_initialize ...
 * graph / state setters ... -> sf3d_dist_export(blob) -> [the launcher all-gathers the blobs, e.g.
 * torch.distributed / MPI] -> sf3d_dist_connect(all blobs) -> sf3d_initialize_balance -> steps.
 * Getters answer for the nodes a rank owns (sf3d_dist_owner); scalar balances are global. */
/* bytes of device memory behind the model's arrays on this rank (0 for libraries without a device); with strip-local device models
 * (default for world > 1; SF3D_DIST_LOCAL=0: every rank uploads the whole global model) about 1 / world of the single-rank figure + halo */
uint64_t     sf3d_device_bytes(void);
/* bytes of HOST memory resident behind the library's staging model (the global copy the setters write + a rank's strip-local copy),
 * counted page by page (mincore).  Multi GPU: once the ranks are connected the pages of the global copy that hold only other ranks'
 * nodes are given back to the system and the strip-local copy drops what only the graph build reads: a rank of an eight-way C4 run
 * keeps about a fifth of the single-rank figure (SF3D_DIST_TRIM_HOST=0 keeps everything). */
uint64_t     sf3d_host_bytes(void);
int          sf3d_dist_blob_bytes(void);
sf3d_error_t sf3d_dist_prepare(int rank, int world);
sf3d_error_t sf3d_dist_export(void* blob_out);
sf3d_error_t sf3d_dist_connect(const void* blobs_of_all_ranks);
/* Which exchange: sf3d_dist_connect opens the peers' windows (device memory, HIP IPC) and checks them (one value through every window,
 * both ways, bounded).  sf3d_dist_status() then says what THIS rank found: 0 = its windows work, 1 = they do not (or SF3D_EXCHANGE asks
 * for another transport).  The launcher all-gathers that one value and calls sf3d_dist_finalize(mode) with the SAME mode on every rank -
 * REQUIRED: the model is not connected before it.
 *   mode 0  keep the device windows (every rank said 0);
 *   mode 2  some rank's device windows failed: the same exchange through HOST-MEMORY windows - every rank has published a POSIX
 *           shared-memory copy of its window in its blob; all ranks map and register them (hipHostRegister) and repeat the self-check
 *           through them.  Slower (both ends of an exchange cross PCIe), same protocol, same bits.  SF3D_EXCHANGE=host forces it;
 *   mode 1  only with SF3D_EXCHANGE=rccl: all ranks join the communicator whose id rank 0 put into its blob (collective:
 *           ncclCommInitRank) and move halos with ncclSend/ncclRecv and the partial sums with ncclAllGather (its sequencing runs
 *           in the tests over a shared-memory stand-in for librccl, SF3D_RCCL_LIB; never between two physical GPUs yet);
 *           without that opt-in mode 1 is an error on every rank. */
int          sf3d_dist_status(void);
sf3d_error_t sf3d_dist_finalize(int mode);
/* which exchange a connected multi-rank model uses: 0 none (one rank, or not connected), 1 device windows, 2 host-memory windows, 3 RCCL */
int          sf3d_dist_transport(void);
/* What the window exchange costs this rank (a connected multi-rank model; diagnostics for a first contact between real GPUs - bench.py
 * prints it per rank): out[0] = exchange epochs closed since the connect (one per Jacobi iteration, halo of K / waterFlow, balance ...);
 * for every rank p: out[1 + 3 p] = flag-hop latency to rank p measured at sf3d_dist_finalize [us] (a system-scope store into p's window
 * seen by p's polling load, handed back and forth 256 times; 0 for the own rank and for the RCCL transport), out[2 + 3 p] = mean wait
 * for rank p's mailbox per epoch [us] (hop + how far p runs behind), out[3 + 3 p] = the longest single wait [us].
 * capacity >= 1 + 3 * world doubles, else SF3D_MEMORY_ERROR; SF3D_MISSING_DATA_ERROR without a connected multi-rank model. */
sf3d_error_t sf3d_dist_stats(double* out, int capacity);
/* owning rank of nodes first..first+count-1 for a world of `world` ranks (host logic, no device); -1: a node this rank never staged */
/* Host logic, no device needed: SF3D_OK and the shape if the staged node graph is a regular NX x NY x NZ grid in layer-major
 * numbering i = (l NY + r) NX + c with the ten-link stencil (slot 0 up, 1 down, laterals to the 8-neighbourhood of the layer;
 * dr[k], dc[k]: row / column step of lateral slot k at a node that has all eight - edge nodes fill their slots in their own order),
 * SF3D_MISSING_DATA_ERROR otherwise (irregular DEM outlines, other numberings).  dr, dc: 8 entries each. */
sf3d_error_t sf3d_get_regular_grid(uint32_t* nx, uint32_t* ny, uint32_t* nz, int8_t* dr, int8_t* dc);
sf3d_error_t sf3d_dist_owner(int world, uint32_t first, uint32_t count, int32_t* owner_out);
/* STRIP-LOCAL BUILD.  The setters are global (the reference's API is), but a rank of a multi-GPU run need not stage the whole model:
 * rank r owns the surface-cell columns whose surface node index lies in [bounds[r], bounds[r + 1]) - a function of nrSurfaceNodes and
 * world alone, returned here without a model (bounds: world + 1 entries; cuts at multiples of 64) - and computes only their rows.  It
 * is enough to call setNode / setNodeLink / setNodeSoil / setNodeSurface / the state setters for the nodes of those columns AND of the
 * one-cell ring of columns around them (every column whole, with all its links), after sf3d_dist_prepare and initializeSF3D with the
 * GLOBAL node counts and with GLOBAL indices.  A node that never receives its soil / surface class is absent: it costs no memory
 * (sf3d_host_bytes), joins no exchange list and reads as NODATA; a link from one of the rank's own nodes to an absent node is
 * SF3D_MISSING_DATA_ERROR at the first device call (a forgotten halo column), and the halo counts of the two sides of every exchange
 * are compared at sf3d_dist_connect.  Staging everything (the global build) stays valid and gives the same bits. */
sf3d_error_t sf3d_dist_bounds(uint32_t nrSurfaceNodes, int world, uint32_t* bounds);
/* halo lists of `rank` in a world of `world`: direction 0 = nodes sent to `peer`, 1 = nodes received
 * from `peer` (sorted global indices; pass out = NULL to query the count) */
sf3d_error_t sf3d_dist_halo(int rank, int world, int peer, int direction, uint32_t capacity,
                            uint32_t* out, uint32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* SF3D_H */
