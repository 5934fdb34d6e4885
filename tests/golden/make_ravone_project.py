#!/usr/bin/env python3
"""Generate tests/golden/ravone_project.npz: the inputs of BASELINE config 5 (the Ravone PROJECT, not only its DEM) as
data, plus what the UNMODIFIED reference code returns for them.  Build container only (/root/reference mounted).

    make -C oracle ref-project && python tests/golden/make_ravone_project.py

Inputs kept (data, not source): the soil-map and land-use rasters of DATA/PROJECT/Ravone (ids as int16), the rows of
tables soils / horizons / van_genuchten of SOIL/soil_ER_2021.db and land_units of DATA/crop_Ravone.db exactly as sqlite
holds them (NULL, '' and numeric text preserved - the conversions are the restatement's job), the [soilWaterFluxes]
parameters of SETTINGS/parameters.ini.  The DEM is the existing tests/golden/ravone_dem_519x1208.npz.

Expected outputs kept (pins for criteria3d_amd/project3d.py; produced by oracle/_ref/ravone_prep = agrolib/gis + agrolib/soil
compiled where they lie + oracle/ref_project_prep.cpp):
  * soil::setHorizon for every horizon of every soil of the database (ok flag, depths, texture class, coarse fragments, organic
    matter, bulk density, van Genuchten alpha/n/m/he/thetaR/thetaS, Ksat, L, clay, field capacity) and soil::getHorizonIndex
    on a ladder of depths;
  * gis::computeSlopeAspectMaps on the DEM: sha256 of the two float32 maps + the maps themselves on rows 380:540, cols
    200:360 (the 160 x 160 window the window tests use);
  * gis::isBoundaryRunoff on every cell (bit-packed);
  * gis::resampleGrid(aggrPrevailing) of the two rasters onto the DEM header (identity here: sha256 compared).
"""
import hashlib
import json
import sqlite3
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from criteria3d_amd import esri, project3d as p3  # noqa: E402

REF = Path("/root/reference")
PRJ = REF / "DATA" / "PROJECT" / "Ravone"
OUT = Path(__file__).resolve().parent
PREP = ROOT / "oracle" / "_ref" / "ravone_prep"
WINDOW = (380, 540, 200, 360)


def fmt(v: float) -> str:
    return repr(float(v))


def main():
    if not PREP.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref-project"], check=True)
    soil_db = sqlite3.connect(f"file:{PRJ / 'SOIL' / 'soil_ER_2021.db'}?mode=ro", uri=True)
    crop_db = sqlite3.connect(f"file:{PRJ / 'DATA' / 'crop_Ravone.db'}?mode=ro", uri=True)
    # loadAllSoils: "SELECT id_soil, soil_code, name FROM soils" (soilDbTools.cpp:832) - table order
    soils = [list(r) for r in soil_db.execute("SELECT id_soil, soil_code FROM soils")]
    cols = ["soil_code"] + list(p3.HORIZON_KEYS)
    horizons = {}
    for code in {s[1] for s in soils}:
        # loadSoilData: "SELECT * FROM horizons WHERE soil_code='..' ORDER BY horizon_nr" (:283-284)
        rows = soil_db.execute(f"SELECT {', '.join(cols)} FROM horizons WHERE soil_code=? ORDER BY horizon_nr", (code,)).fetchall()
        horizons[code] = [dict(zip(cols[1:], r[1:])) for r in rows]
    vg = [list(r) for r in soil_db.execute("SELECT id_texture, texture, alpha, n, he, theta_r, theta_s, k_sat, l FROM van_genuchten ORDER BY id_texture")]
    land_units = [list(r) for r in crop_db.execute("SELECT id_unit, name, description, id_landuse, id_crop, roughness, pond FROM land_units")]
    ini = {}
    sect = None
    for line in open(PRJ / "SETTINGS" / "parameters.ini"):
        line = line.strip()
        if line.startswith("["):
            sect = line.strip("[]")
        elif "=" in line and sect == "soilWaterFluxes":
            k, v = line.split("=", 1)
            ini[k] = v
    tables = dict(soils=soils, horizons=horizons, van_genuchten=vg, land_units=land_units, soilWaterFluxes=ini)

    # rows for the driver: the conversions of convert_horizon_row are applied HERE, so the driver's setHorizon sees exactly
    # what loadSoilData would have stored in dbData
    textures = p3.texture_classes([tuple(r) for r in vg])
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        with open(tmp / "rows.txt", "w") as f:
            f.write(f"UNITS {len(land_units)} " + " ".join(str(int(r[0])) for r in land_units) + "\n")
            f.write("VG 12\n")
            for r in vg:
                t = textures[int(r[0])]
                f.write(" ".join([str(int(r[0])), str(r[1]).replace(" ", "_")] + [fmt(t[k]) for k in
                        ("alpha", "n", "he", "m", "sc", "theta_r", "ref_theta_s", "ksat", "l")]) + "\n")
            listed = [s for s in soils if s[0] is not None and s[1] not in (None, "") and horizons.get(s[1])]
            f.write(f"SOILS {len(listed)}\n")
            for sid, code in listed:
                rows = [p3.convert_horizon_row(r) for r in horizons[code]]
                f.write(f"{int(sid)} {code.replace(' ', '_')} {len(rows)}\n")
                for r in rows:
                    f.write(" ".join([str(r["horizon_nr"])] + [fmt(r[k]) for k in p3.HORIZON_KEYS[1:]]) + "\n")
        out = subprocess.run([str(PREP), str(REF / "DATA" / "DEM" / "DEM_Ravone.flt"), str(PRJ / "SOIL" / "soilMap_Ravone.flt"),
                              str(PRJ / "MAPS" / "landUse_Ravone.flt"), str(tmp / "rows.txt"), str(tmp)], check=True,
                             stdout=subprocess.PIPE, text=True).stdout
        print(out.strip())
        dem, hdr = esri.read_grid(REF / "DATA" / "DEM" / "DEM_Ravone")
        shape = dem.shape
        rd = {k: np.fromfile(tmp / f"{k}.f32", dtype="<f4").reshape(shape) for k in ("dem", "soilmap", "landuse", "slope", "aspect")}
        boundary = np.fromfile(tmp / "boundary.u8", dtype=np.uint8).reshape(shape)
        ref_h, ref_idx = [], []
        for line in open(tmp / "horizons_out.txt"):
            p = line.split()
            if p[0] == "SOIL":
                ref_h.append(dict(id=int(p[1]), code=p[2], rows=[]))
            elif p[0] == "H":
                ref_h[-1]["rows"].append([float(v) for v in p[1:]])
            elif p[0] == "IDX":
                ref_idx.append([int(v) for v in p[1:]])

    # the fixtures hold what the application reads, the reference's own reader must agree with ours
    fixture_dem, _ = esri.load_dem_fixture(OUT / "ravone_dem_519x1208.npz")
    assert np.array_equal(rd["dem"], fixture_dem), "DEM fixture differs from gis::openRaster"
    soil_raw, _ = esri.read_grid(PRJ / "SOIL" / "soilMap_Ravone")
    land_raw, _ = esri.read_grid(PRJ / "MAPS" / "landUse_Ravone")
    assert np.array_equal(rd["soilmap"], soil_raw) and np.array_equal(rd["landuse"], land_raw), "resampleGrid is not the identity"
    assert np.array_equal(soil_raw, soil_raw.astype(np.int16)) and np.array_equal(land_raw, land_raw.astype(np.int16))

    r0, r1, c0, c1 = WINDOW
    nh = max(len(s["rows"]) for s in ref_h)
    href = np.full((len(ref_h), nh, 18), np.nan)
    for i, s in enumerate(ref_h):
        for j, r in enumerate(s["rows"]):
            href[i, j] = r
    np.savez_compressed(
        OUT / "ravone_project.npz",
        soil_map=soil_raw.astype(np.int16), land_use=land_raw.astype(np.int16),
        tables_json=np.array(json.dumps(tables)),
        ref_horizons=href, ref_horizon_soil_id=np.array([s["id"] for s in ref_h]), ref_horizon_count=np.array([len(s["rows"]) for s in ref_h]),
        ref_horizon_index_ladder=np.array(ref_idx, np.int32),
        ref_horizon_columns=np.array("ok has_error upper lower class_usda coarse organic_matter bulk_density alpha n m he theta_r theta_s ksat l clay field_capacity"),
        ref_slope_sha256=np.array(hashlib.sha256(rd["slope"].tobytes()).hexdigest()),
        ref_aspect_sha256=np.array(hashlib.sha256(rd["aspect"].tobytes()).hexdigest()),
        ref_slope_window=rd["slope"][r0:r1, c0:c1], ref_aspect_window=rd["aspect"][r0:r1, c0:c1], window=np.array(WINDOW),
        ref_boundary_bits=np.packbits(boundary.astype(bool)), shape=np.array(shape))
    print("wrote", OUT / "ravone_project.npz", (OUT / "ravone_project.npz").stat().st_size, "bytes;",
          len(ref_h), "soils,", int(sum(len(s['rows']) for s in ref_h)), "horizons,", int(boundary.sum()), "runoff-boundary cells")


if __name__ == "__main__":
    main()
