/*
 * ref_project_prep.cpp - TEST INFRASTRUCTURE, not product code.
 *
 * Driver over the UNMODIFIED reference libraries agrolib/gis and agrolib/soil (compiled where they lie by
 * oracle/Makefile target `ref-project` into oracle/_ref/ravone_prep).  It runs the reference's own functions on the
 * Ravone project inputs and dumps what they return, so that tests/golden/make_ravone_project.py can store those
 * outputs as the pins of the host-side restatement (criteria3d_amd/project3d.py):
 *
 *   gis::openRaster / readEsriGridFlt          DEM, soil map, land-use map               (gisIO.cpp)
 *   gis::resampleGrid(.., aggrPrevailing, 0)   maps onto the DEM header                  (project3D.cpp:673,699)
 *   gis::computeSlopeAspectMaps                radiationMaps->slopeMap / aspectMap       (solarRadiation.cpp:65)
 *   gis::isBoundaryRunoff                      Project3D::setLateralBoundary             (project3D.cpp:851-873)
 *   soil::setHorizon                           loadSoil's per-horizon call               (soilDbTools.cpp:427)
 *   soil::getHorizonIndex                      setCrit3DNodeSoil                          (project3D.cpp:1213)
 *
 * Nothing here restates reference arithmetic: the driver only moves data in and out.  The database rows arrive as a
 * text file written by the generator with python's sqlite3 (agrolib/utilities does not compile against the image's
 * Qt 5.9: QFileInfo::fileTime is Qt 5.10), so the QVariant -> double conversions of soilDbTools.cpp:316-355 are the
 * generator's and are cited there.
 *
 * usage: ravone_prep <dem> <soilMap> <landUse> <soil_rows.txt> <outdir>      (rasters as .flt paths)
 */
#include "gis.h"
#include "soil.h"
#include "commonConstants.h"
#include "basicMath.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

static void dump_f32(const std::string& path, const gis::Crit3DRasterGrid& g)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    for (int r = 0; r < g.header->nrRows; r++) fwrite(g.value[r], sizeof(float), size_t(g.header->nrCols), f);
    fclose(f);
}

int main(int argc, char** argv)
{
    if (argc != 6) { fprintf(stderr, "usage: %s dem soilMap landUse soil_rows.txt outdir\n", argv[0]); return 2; }
    const std::string out = argv[5];
    std::string err;

    gis::Crit3DRasterGrid dem, rawSoil, rawLand, soilMap, landUse, slope, aspect;
    const int utmZone = 32;                                          // Ravone.ini [location] utm_zone
    if (!gis::openRaster(argv[1], &dem, utmZone, err)) { fprintf(stderr, "dem: %s\n", err.c_str()); return 1; }
    if (!gis::openRaster(argv[2], &rawSoil, utmZone, err)) { fprintf(stderr, "soil map: %s\n", err.c_str()); return 1; }
    if (!gis::openRaster(argv[3], &rawLand, utmZone, err)) { fprintf(stderr, "land use: %s\n", err.c_str()); return 1; }
    gis::resampleGrid(rawSoil, &soilMap, dem.header, aggrPrevailing, 0);       // project3D.cpp:699
    gis::resampleGrid(rawLand, &landUse, dem.header, aggrPrevailing, 0);       // project3D.cpp:673
    if (!gis::computeSlopeAspectMaps(dem, &slope, &aspect)) { fprintf(stderr, "slope/aspect failed\n"); return 1; }

    dump_f32(out + "/dem.f32", dem);
    dump_f32(out + "/soilmap.f32", soilMap);
    dump_f32(out + "/landuse.f32", landUse);
    dump_f32(out + "/slope.f32", slope);
    dump_f32(out + "/aspect.f32", aspect);

    std::ifstream in(argv[4]);
    if (!in) { fprintf(stderr, "cannot read %s\n", argv[4]); return 1; }
    std::string tag;
    int n;

    // land unit ids (land_units.id_unit): layer 0 of the index map holds a node wherever the land-use map names one
    // of them (Project3D::setIndexMaps, project3D.cpp:779-787 -> getLandUnitIndexRowCol :1476-1492)
    in >> tag >> n;
    std::vector<int> unitId(n);
    for (int i = 0; i < n; i++) in >> unitId[i];

    gis::Crit3DIndexGrid index0;
    index0.initializeGrid(*dem.header);
    const long noIndex = long(index0.header->flag);
    long current = 0;
    for (int r = 0; r < dem.header->nrRows; r++)
        for (int c = 0; c < dem.header->nrCols; c++)
        {
            index0.value[r][c] = noIndex;
            if (isEqual(dem.value[r][c], dem.header->flag)) continue;
            bool found = (n <= 1);
            if (!found)
            {
                int id = int(landUse.value[r][c]);
                if (id != int(landUse.header->flag))
                    for (int k = 0; k < n; k++) if (unitId[k] == id) found = true;
            }
            if (found) index0.value[r][c] = current++;
        }

    {
        FILE* f = fopen((out + "/boundary.u8").c_str(), "wb");
        for (int r = 0; r < dem.header->nrRows; r++)
            for (int c = 0; c < dem.header->nrCols; c++)
            {
                unsigned char b = gis::isBoundaryRunoff(index0, dem, aspect, r, c) ? 1 : 0;      // project3D.cpp:865
                fwrite(&b, 1, 1, f);
            }
        fclose(f);
    }

    // texture classes (van_genuchten table; the assignments of loadVanGenuchtenParameters, soilDbTools.cpp:171-192,
    // are plain copies apart from m and sc, which are reference formulas evaluated by the generator and checked there)
    std::vector<soil::Crit3DTextureClass> texture(13);
    std::vector<soil::Crit3DGeotechnicsClass> geotechnics(19);
    soil::Crit3DFittingOptions fitting;                                // Project3D's default-constructed options
    in >> tag >> n;
    for (int i = 0; i < n; i++)
    {
        int id; std::string name;
        in >> id >> name;
        for (auto& ch : name) if (ch == '_') ch = ' ';
        auto& t = texture[size_t(id)];
        t.classNameUSDA = name;
        in >> t.vanGenuchten.alpha >> t.vanGenuchten.n >> t.vanGenuchten.he >> t.vanGenuchten.m >> t.vanGenuchten.sc
           >> t.vanGenuchten.thetaR >> t.vanGenuchten.refThetaS >> t.waterConductivity.kSat >> t.waterConductivity.l;
        t.vanGenuchten.thetaS = t.vanGenuchten.refThetaS;
    }

    FILE* fs = fopen((out + "/horizons_out.txt").c_str(), "w");
    int nSoils;
    in >> tag >> nSoils;
    for (int s = 0; s < nSoils; s++)
    {
        int idSoil, nh; std::string code;
        in >> idSoil >> code >> nh;
        soil::Crit3DSoil mySoil;
        mySoil.initialize(code, nh);
        fprintf(fs, "SOIL %d %s %d\n", idSoil, code.c_str(), nh);
        for (int h = 0; h < nh; h++)
        {
            auto& d = mySoil.horizon[size_t(h)].dbData;
            in >> d.horizonNr >> d.upperDepth >> d.lowerDepth >> d.sand >> d.silt >> d.clay >> d.coarseFragments
               >> d.organicMatter >> d.bulkDensity >> d.thetaSat >> d.kSat;
            std::string herr;
            auto& hz = mySoil.horizon[size_t(h)];
            bool ok = soil::setHorizon(hz, texture, geotechnics, fitting, herr);               // soilDbTools.cpp:427
            fprintf(fs, "H %d %d %.17g %.17g %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n",
                    ok ? 1 : 0, herr.empty() ? 0 : 1, hz.upperDepth, hz.lowerDepth, hz.texture.classUSDA, hz.coarseFragments,
                    hz.organicMatter, hz.bulkDensity, hz.vanGenuchten.alpha, hz.vanGenuchten.n, hz.vanGenuchten.m,
                    hz.vanGenuchten.he, hz.vanGenuchten.thetaR, hz.vanGenuchten.thetaS, hz.waterConductivity.kSat,
                    hz.waterConductivity.l, hz.texture.clay, hz.fieldCapacity);
        }
        // horizon lookup at a ladder of depths (setCrit3DNodeSoil's call, project3D.cpp:1213), on the horizons as set
        fprintf(fs, "IDX");
        mySoil.nrHorizons = unsigned(nh);
        for (int k = 0; k <= 40; k++) fprintf(fs, " %d", soil::getHorizonIndex(mySoil, 0.05 * k));
        fprintf(fs, "\n");
    }
    fclose(fs);
    printf("ravone_prep: %d x %d cells, %ld surface nodes, %d soils\n", dem.header->nrRows, dem.header->nrCols, current, nSoils);
    return 0;
}
