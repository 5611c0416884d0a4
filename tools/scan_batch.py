#!/usr/bin/env python3
"""Batch cell scan of an int8 I,Q capture on one MI355X, straight on the library's batch entry points.

The per-call shim (shim/scan_demo.cc over shim/liblte_phy_shim.cc) proves the drop-in claim one `liblte_phy_*` call at a time;
this is the same receiver the way the batch API is meant to be used (LTE_fdd_dl_file_scan's state machine,
LTE_fdd_dl_fs_samp_buf.cc:277-600, with every loop over subframes turned into one launch): the capture goes to HBM once, then

    coarse timing -> PSS + fine timing -> SSS                      (per correlation peak; mi_lte_coarse_timing_run / find_pss / find_sss)
    front end for subframe 0 of every frame, 4 ports -> PBCH        (mi_lte_dl_frontend_batch, mi_lte_pbch_decode_run)
    front end for EVERY subframe of every frame                     (one launch)
    PCFICH + PDCCH common search space for every subframe           (one launch, mi_lte_pdcch_decode_run)
    PDSCH decode of every allocation the DCIs announce              (one plan per control-region size, mi_lte_pdsch_decode_run)

and only CFIs, DCIs, verdicts and transport blocks come back.  (No frequency correction between the stages: the scanner's
caller-side derotation, LTE_fdd_dl_fs_samp_buf.cc:696-713, is not part of liblte_phy.)  Usage: scan_batch.py <capture.bin> <fs in MHz> [--json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FFT_OF_FS = {1.92: 128, 3.84: 256, 7.68: 512, 15.36: 1024, 30.72: 2048}
RB_OF_BW = [6, 15, 25, 50, 75, 100]
PHICH_RES = [1.0 / 6, 0.5, 1.0, 2.0]


def parse_mib(mib):
    """MasterInformationBlock (36.331 6.2.2): dl-Bandwidth(3) phich-Duration(1) phich-Resource(2) systemFrameNumber(8) spare(10)."""
    return dict(N_rb_dl=RB_OF_BW[min((mib >> 21) & 7, 5)], phich_dur=(mib >> 20) & 1, phich_res=(mib >> 18) & 3, sfn_div_4=(mib >> 10) & 0xFF)


def scan(ctx, iq, fft, n_slots=160, max_cells=8):
    import openlte_amd as m
    t0 = time.perf_counter()
    sc = 2048 // fft
    n_sf, n_frame, look = 30720 // sc, 307200 // sc, 4400 // sc + 1
    d_iq = ctx.to_device(np.ascontiguousarray(iq))
    cfg6 = m.DlCfg(fft, 6, 1, m.IQ_I8)  # before the MIB only the centre six resource blocks are known to exist
    need = int(ctx.L.mi_lte_coarse_timing_samples(fft, n_slots))
    report = dict(samples=len(iq), frames=len(iq) / n_frame, cells=[])
    if len(iq) < need:
        return report
    timing = ctx.coarse_timing_dev(cfg6, d_iq, None, n_slots)
    report["coarse_peaks"] = timing.n_corr_peaks
    seen = set()
    for p in range(timing.n_corr_peaks):
        ss, n_id_2, _, thresh, _ = ctx.find_pss_dev(cfg6, d_iq, None, list(timing.symb_starts[p]))
        found, n_id_1, frame_start, _ = ctx.find_sss_dev(cfg6, d_iq, None, n_id_2, ss, thresh)
        cell = 3 * n_id_1 + n_id_2
        if not found or cell in seen or len(seen) >= max_cells:
            continue
        frame_start %= n_frame
        n_frames = (len(iq) - frame_start - look) // n_frame
        if n_frames < 1:
            continue
        # ---- PBCH on subframe 0 of every frame (4-port estimates), first frame that decodes wins
        starts0 = (frame_start + n_frame * np.arange(n_frames)).astype(np.uint64)
        cfg4 = m.DlCfg(fft, 6, 4, m.IQ_I8)
        d_st, d_sf, d_cell = ctx.to_device(starts0), ctx.to_device(np.zeros(n_frames, np.uint32)), ctx.to_device(np.full(n_frames, cell, np.uint32))
        d_sub = ctx.alloc(n_frames * ctx.subframe_floats(4) * 4)
        ctx.dl_frontend_dev(cfg4, d_iq, None, d_st, d_sf, d_cell, n_frames, d_sub)
        n_ant, off, mib = ctx.pbch_decode_dev(cfg4, d_sub, d_cell, n_frames)
        for b in (d_st, d_sf, d_cell, d_sub):
            b.free()
        ok = np.flatnonzero(n_ant > 0)
        if len(ok) == 0:
            continue
        seen.add(cell)
        f0 = int(ok[0])
        info = parse_mib(int(mib[f0]))
        sfn0 = (info["sfn_div_4"] << 2) + int(off[f0]) - f0  # SFN of the first whole frame of the capture
        rec = dict(cell=cell, frame_start=int(frame_start), N_ant=int(n_ant[f0]), frames=int(n_frames), pbch_frames_decoded=int(len(ok)), sfn_first_frame=sfn0, **info)
        # ---- every subframe of every frame: front end, control region, shared channel
        N_rb, N_ant = info["N_rb_dl"], int(n_ant[f0])
        cfg = m.DlCfg(fft, N_rb, N_ant, m.IQ_I8)
        n = n_frames * 10
        idx = np.arange(n)
        d_st = ctx.to_device((frame_start + n_sf * idx).astype(np.uint64))
        sfs = (idx % 10).astype(np.uint32)
        d_sf, d_cell = ctx.to_device(sfs), ctx.to_device(np.full(n, cell, np.uint32))
        d_sub = ctx.alloc(n * ctx.subframe_floats(N_ant) * 4)
        ctx.dl_frontend_dev(cfg, d_iq, None, d_st, d_sf, d_cell, n, d_sub)
        plan = ctx.pdcch_plan(cfg, [cell], PHICH_RES[info["phich_res"]])
        rc, cfi, nsym, ndci, dci = plan.decode_raw(d_sub, d_sf, d_cell, n)
        plan.close()
        rec["subframes"] = n
        rec["cfi_decoded"] = int((cfi > 0).sum())
        by_nsym = {}
        for u in range(n):
            got = set()
            for k in range(int(ndci[u])):
                d = dci[6 * u + k]
                key = (d.alloc.rnti, d.payload, d.format)  # the same DCI often decodes at aggregation levels 4 and 8
                if not d.alloc_valid or key in got or d.alloc.tbs + 24 > 6144 or d.alloc.N_prb == 0:
                    continue
                got.add(key)
                a = m.PdschAlloc.from_buffer_copy(d.alloc)
                a.unit = u
                by_nsym.setdefault(int(nsym[u]), []).append((u, a, d.mcs))
        blocks = []
        for ns, lst in sorted(by_nsym.items()):
            try:
                pl = ctx.pdsch_plan(cfg, ns, [a for _, a, _ in lst])
            except m.MiLteError:
                continue
            st, bits = run_plan(ctx, pl, d_sub, d_sf, d_cell, len(lst))
            pl.close()
            for (u, a, mcs), s_, b_ in zip(lst, st, bits):
                blocks.append(dict(sfn=sfn0 + u // 10, subframe=u % 10, n_symbs=ns, rnti=int(a.rnti), tbs=int(a.tbs), N_prb=int(a.N_prb), rv=int(a.rv_idx),
                                   mcs=int(mcs), crc_ok=bool(s_ == 0), bits=np.asarray(b_, np.uint8) if s_ == 0 else None))
        blocks.sort(key=lambda r: (r["sfn"], r["subframe"]))
        rec["transport_blocks"] = blocks
        for b in (d_st, d_sf, d_cell, d_sub):
            b.free()
        report["cells"].append(rec)
    d_iq.free()
    report["seconds"] = time.perf_counter() - t0
    return report


def run_plan(ctx, pl, d_sub, d_sf, d_cell, n_alloc):
    import numpy as np
    d_out, d_st = ctx.alloc(n_alloc * pl.out_stride), ctx.alloc(n_alloc * 4)
    pl.run_dev(d_sub, d_sf, d_cell, d_out, d_st)
    st = d_st.download(np.int32)
    bits = d_out.download(np.uint8).reshape(n_alloc, pl.out_stride)
    d_out.free()
    d_st.free()
    return st, [bits[i] for i in range(n_alloc)]


def main():
    import openlte_amd as m
    if len(sys.argv) < 3:
        print(__doc__)
        return 2
    fft = FFT_OF_FS[min(FFT_OF_FS, key=lambda f: abs(f - float(sys.argv[2])))]
    iq = np.fromfile(sys.argv[1], np.int8)
    iq = iq[:len(iq) // 2 * 2].reshape(-1, 2)
    ctx = m.Context(0)
    scan(ctx, iq, fft)  # warm-up: HIP start-up, tables
    rep = scan(ctx, iq, fft)
    ctx.close()
    if "--json" in sys.argv:
        for c in rep["cells"]:
            for b in c["transport_blocks"]:
                b["bits"] = None if b["bits"] is None else "".join(map(str, b["bits"][:b["tbs"]]))
        print(json.dumps(rep))
        return 0
    print("capture: %d samples (%.1f frames), %d coarse peak(s), scanned in %.1f ms" % (rep["samples"], rep["frames"], rep.get("coarse_peaks", 0), 1e3 * rep["seconds"]))
    for c in rep["cells"]:
        print("cell %d: frame start %d, %d antenna port(s), MIB: N_rb_dl=%d phich_dur=%d phich_res=%d sfn=%d (PBCH decoded in %d of %d frames)"
              % (c["cell"], c["frame_start"], c["N_ant"], c["N_rb_dl"], c["phich_dur"], c["phich_res"], c["sfn_first_frame"], c["pbch_frames_decoded"], c["frames"]))
        ok = [b for b in c["transport_blocks"] if b["crc_ok"]]
        print("  %d subframes, CFI decoded in %d, %d PDSCH transport blocks announced, %d decoded (CRC ok)" % (c["subframes"], c["cfi_decoded"], len(c["transport_blocks"]), len(ok)))
        for b in ok[:12]:
            print("  sfn %d subframe %d: CFI-symbols=%d rnti=0x%04X tbs=%d N_prb=%d rv=%d" % (b["sfn"], b["subframe"], b["n_symbs"], b["rnti"], b["tbs"], b["N_prb"], b["rv"]))
    print("%d cell(s) found" % len(rep["cells"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
