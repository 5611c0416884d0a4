"""The one downlink case of the fuzz soak (MI_LTE_FUZZ_SEED=17) whose soft bits differ from the reference's in the exact stage: which soft
bit, and the equalised symbol behind it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fuzz_cases as fz  # noqa: E402
import openlte_amd as m  # noqa: E402
from oracle import pyoracle  # noqa: E402

R = pyoracle.ref_big()
ctx = m.Context(0)
for chunk in range(20):
    cases = fz.draw_dl_cases(1000, 1000 + chunk + 100000 * 17)
    hit = [i for i, c in enumerate(cases) if c["n_rb"] == 100 and c["cell"] == 303 and c["sf"] == 7 and c["tbs"] == 5800 and c["mod"] == 1]
    if not hit:
        continue
    r = fz.run_ref_dl(R, cases)
    for i in hit:
        c = cases[i]
        cfg = m.DlCfg(c["fft"], c["n_rb"], c["n_ant"], m.IQ_I8)
        al = m.make_alloc(0, c["mod"], c["tbs"], c["prb0"], c["rnti"], c["rv"], c["tx_mode"], c["prb1"])
        grid = np.ascontiguousarray(r["planes"][i, :2 + 2 * c["n_ant"]]).reshape(-1)
        d_sub = ctx.to_device(grid)
        plan = ctx.pdsch_plan(cfg, c["n_sym"], [al])
        st, bits = plan.run(d_sub, np.array([c["sf"]], np.uint32), np.array([c["cell"]], np.uint32))
        e = plan.soft_bits(0)
        ne = int(r["n_soft"][i])
        want = r["soft"][i, :ne]
        d = np.nonzero(e[:ne] != want)[0]
        print("chunk", chunk, "case", i, "soft bits", ne, "differing at", d.tolist(), "got", e[d].tolist(), "want", want[d].tolist())
        # the resource elements in mapping order (single port), and the equalised symbol of the differing soft bit
        P = r["planes"][i]
        res = []
        for L in range(c["n_sym"], 14):
            l7 = L % 7
            for p in (c["prb0"] if L < 7 else c["prb1"]):
                for j in range(12):
                    if (l7 == 0 and c["cell"] % 6 == j % 6) or (l7 == 4 and (c["cell"] + 3) % 6 == j % 6):
                        continue
                    res.append((L, p * 12 + j))
        for k in d:
            L, sc = res[k // 2]
            yr, yi, hr, hi = (np.float32(P[q, L, sc]) for q in (0, 1, 2, 3))
            hn = np.float32(hr * hr) + np.float32(hi * hi)
            xr = (np.float32(yr * hr) + np.float32(yi * hi)) / hn
            xi = (np.float32(yi * hr) - np.float32(yr * hi)) / hn
            print("  soft bit", k, "symbol", k // 2, "RE", (L, sc), "x =", float(xr), float(xi), "ratio re/im", float(xr) / float(xi) if xi else None,
                  "host atan2f", float(np.arctan2(np.float32(xi), np.float32(xr))), repr(np.arctan2(np.float32(xi), np.float32(xr))))
        plan.close(); d_sub.free()
