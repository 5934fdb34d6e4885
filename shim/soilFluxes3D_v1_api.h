/*
 * soilFluxes3D_v1_api.h - the retired first-generation API `soilFluxes3D::v1`
 * (agrolib/soilFluxes3D/old/old_soilFluxes3D.h:16-113; not built by the reference any more,
 * SURVEY.md 8f-4) re-declared so that older callers - and BASELINE.json's north_star, which names
 * `initializeFluxes / setNode / setNodeLink / computePeriod` - can link the MI355X library.
 * Plain C++ linkage like the reference's non-DLL build (old_macro.h:14-18).  The three QString
 * debug-log getters of v1 are not provided (Qt-only, never called by the applications).
 * Integer codes are v1's: CRIT3D_OK 0, INDEX_ERROR -1111, ... (commonConstants.h:95-105);
 * boundary types 2 Runoff, 3 FreeDrainage, 4 FreeLateralDrainage, 5 PrescribedTotalPotential,
 * 10 Urban, 11 Road, 12 Culvert, 20 HeatSurface, 30 SoluteFlux, 99 none (:118-128); link directions
 * UP 1, DOWN 2, LATERAL 3 (:91-93); mean types 0 geometric, 1 logarithmic (:112-113).
 */
#pragma once

namespace soilFluxes3D::v1 {

int test();
void cleanMemory();
int initializeFluxes(long nrNodes, int nrLayers, int nrLateralLinks, bool isComputeWater, bool isComputeHeat, bool isComputeSolutes);
void initializeHeat(short saveHeatFluxes_, bool computeAdvectiveHeat, bool computeLatentHeat);
int setNumericalParameters(double minDeltaT, double maxDeltaT, int maxIterationNumber, int maxApproximationsNumber,
                           int ResidualTolerance, double MBRThreshold);
int setThreadsNumber(int nrThreads);

int setNode(long myIndex, float x, float y, double z, double volume_or_area, bool isSurface, bool isBoundary,
            int boundaryType, float slope, float boundaryArea);
int setNodeLink(long nodeIndex, long linkIndex, short direction, float S0);
int setCulvert(long myIndex, double roughness, double slope, double width, double height);

int setSoilProperties(int nrSoil, int nrHorizon, double VG_alpha, double VG_n, double VG_m, double VG_he, double ThetaR,
                      double ThetaS, double Ksat, double L, double organicMatter, double clay);
int setNodeSoil(long nodeIndex, int soilIndex, int horizonIndex);
int setSurfaceProperties(int surfaceIndex, double Roughness);
int setNodeSurface(long nodeIndex, int surfaceIndex);
int setNodePond(long nodeIndex, double pond);

int setHydraulicProperties(int waterRetentionCurve, int conductivityMeanType, float conductivityHorizVertRatio);
int setWaterContent(long index, double myWaterContent);
int setDegreeOfSaturation(long nodeIndex, double degreeOfSaturation);
int setMatricPotential(long index, double psi);
int setTotalPotential(long index, double totalPotential);
int setPrescribedTotalPotential(long index, double prescribedTotalPotential);
int setWaterSinkSource(long index, double sinkSource);

double getWaterContent(long nodeIndex);
double getMaximumWaterContent(long nodeIndex);
double getAvailableWaterContent(long nodeIndex);
double getWaterDeficit(long index, double fieldCapacity);
double getTotalWaterContent();
double getDegreeOfSaturation(long nodeIndex);
double getBoundaryWaterFlow(long nodeIndex);
double getBoundaryWaterSumFlow(int boundaryType);
double getMatricPotential(long nodeIndex);
double getTotalPotential(long nodeIndex);
double getWaterMBR();
double getWaterConductivity(long nodeIndex);
double getWaterFlow(long nodeIndex, short direction);
double getSumLateralWaterFlow(long nodeIndex);
double getSumLateralWaterFlowIn(long nodeIndex);
double getSumLateralWaterFlowOut(long nodeIndex);
double getWaterStorage();
double getPond(long nodeIndex);

int setHeatSinkSource(long nodeIndex, double myHeatFlow);
int setTemperature(long nodeIndex, double myT);
int setHeatBoundaryHeightWind(long nodeIndex, double myHeight);
int setHeatBoundaryHeightTemperature(long nodeIndex, double myHeight);
int setHeatBoundaryTemperature(long nodeIndex, double myTemperature);
int setHeatBoundaryRelativeHumidity(long nodeIndex, double myRelativeHumidity);
int setHeatBoundaryRoughness(long nodeIndex, double myRoughness);
int setHeatBoundaryWindSpeed(long nodeIndex, double myWindSpeed);
int setHeatBoundaryNetIrradiance(long nodeIndex, double myNetIrradiance);
int setFixedTemperature(long nodeIndex, double myT, double myDepth);

double getTemperature(long nodeIndex);
double getHeatConductivity(long nodeIndex);
double getHeat(long nodeIndex, double h);
double getNodeVapor(long nodeIndex);
float getHeatFlux(long nodeIndex, short myDirection, int fluxType);
double getBoundarySensibleFlux(long nodeIndex);
double getBoundaryAdvectiveFlux(long nodeIndex);
double getBoundaryLatentFlux(long nodeIndex);
double getBoundaryRadiativeFlux(long nodeIndex);
double getBoundaryAerodynamicConductance(long nodeIndex);
double getBoundarySoilConductance(long nodeIndex);
double getHeatMBR();
double getHeatMBE();

void initializeBalance();
void computePeriod(double timePeriod);
double computeStep(double maxTime);

}  // namespace soilFluxes3D::v1
