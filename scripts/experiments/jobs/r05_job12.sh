#!/bin/bash
# round 5 job 12: with the reference's own serial Gauss-Seidel order on the device (SF3D_HEAT_GS=1) and the C library's elementary functions, is the
# coupled heat step bit-identical to the oracle too?  (the test prints GS / two-colour / Jacobi against the oracle)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_heat.py -q -s -k "reference_order" > gpurun_out/r05_job12_heat_gs.log 2>&1; grep -E "GS vs oracle|heat sweeps|passed|failed" gpurun_out/r05_job12_heat_gs.log
python - <<'PY' > gpurun_out/r05_job12_heat_gs_bits.txt 2>&1
import os, numpy as np
os.environ["SF3D_HEAT_GS"] = "1"
from criteria3d_amd import capi, catchment as cm
from tests import checkers
gpu, ora = capi.load_product(), checkers.load_oracle()
for name, m, heat, hours in (("40x40x6 12 soils, water + latent heat", cm.with_heat_surface(cm.catchment_model(40, 40, 6, heterogeneous=True)), cm.Heat(water=True, latent=True, save_mode=0), 2),
                             ("heat only 24x24x6", cm.with_heat_surface(cm.catchment_model(24, 24, 6)), cm.Heat(water=False, latent=False, save_mode=1), 2)):
    res = []
    for sf in (gpu, ora):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=4, heat=heat)
        out = []
        for h in range(hours):
            cm.apply_heat_forcing(sf, m, h)
            _, dts = cm.run_hour(sf, m, 4.0 if h == 0 else 0.0)
            out.append((np.array(dts), sf.temperature(0, m.n)[m.ns:], sf.total_potential(0, m.n)))
        res.append(out); sf.lib.sf3d_clean()
    for h, ((gd, gT, gH), (od, oT, oH)) in enumerate(zip(*res)):
        print(f"{name}, hour {h}: steps {len(gd)} dt equal {np.array_equal(gd, od)}; T bit-identical {np.array_equal(gT, oT)} (max rel {np.max(np.abs(gT-oT)/oT):.2e}, {int((gT!=oT).sum())} of {gT.size} differ); H bit-identical {np.array_equal(gH, oH)} (max rel {np.max(np.abs(gH-oH)/np.maximum(np.abs(oH),1e-9)):.2e})")
PY
cat gpurun_out/r05_job12_heat_gs_bits.txt | tail -6
