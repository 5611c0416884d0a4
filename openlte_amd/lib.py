"""ctypes binding of libmi_lte.so (include/mi_lte.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SOFT_F32, SOFT_I8, SOFT_I16 = 0, 1, 2
TURBO_REF, TURBO_BCJR, TURBO_BCJR_BLOCK, TURBO_BCJR_EARLY = 0, 1, 2, 3  # BCJR_BLOCK: one code block per wavefront, one launch (a handful of blocks; its own model)
_SOFT_OF_DTYPE = {np.dtype(np.float32): SOFT_F32, np.dtype(np.int8): SOFT_I8, np.dtype(np.int16): SOFT_I16}


IQ_I8, IQ_F32_PLANAR = 0, 1
IQ_ALL_ROWS = 0x100  # OR into DlCfg.sample_format: symbol row 15 also for one or two ports (MI_LTE_IQ_ALL_ROWS)
CE_COMPACT = 0x200   # OR into DlCfg.sample_format of the front end AND the PDSCH plan: magnitude / phase rows instead of estimate rows (MI_LTE_CE_COMPACT)


class DlCfg(C.Structure):
    """mi_lte_dl_cfg"""
    _fields_ = [("fft_size", C.c_uint32), ("N_rb_dl", C.c_uint32), ("N_ant", C.c_uint32), ("sample_format", C.c_uint32)]


class UlCfg(C.Structure):
    """mi_lte_ul_cfg: the liblte_phy_ul_init arguments the PUSCH DMRS depends on"""
    _fields_ = [(n, C.c_uint32) for n in ("group_assignment_pusch", "group_hopping_enabled", "sequence_hopping_enabled",
                                          "cyclic_shift", "cyclic_shift_dci")]


class PrachCfg(C.Structure):
    """mi_lte_prach_cfg"""
    _fields_ = [(n, C.c_uint32) for n in ("root_seq_idx", "preamble_format", "zczc", "hs_flag", "freq_offset")]


class PdschAlloc(C.Structure):
    """mi_lte_pdsch_alloc"""
    _fields_ = [(n, C.c_uint32) for n in ("unit", "mod_type", "tbs", "rv_idx", "tx_mode", "rnti", "N_prb", "n_pdcch_symbs")] + \
               [("prb", (C.c_uint8 * 112) * 2)]


def make_alloc(unit, mod_type, tbs, prbs, rnti, rv_idx=0, tx_mode=1, prbs_slot1=None, n_pdcch_symbs=0):
    a = PdschAlloc()
    a.unit, a.mod_type, a.tbs, a.rv_idx, a.tx_mode, a.rnti, a.N_prb = unit, mod_type, tbs, rv_idx, tx_mode, rnti, len(prbs)
    a.n_pdcch_symbs = n_pdcch_symbs
    for i, p in enumerate(prbs):
        a.prb[0][i] = p
        a.prb[1][i] = (prbs_slot1 or prbs)[i]
    return a


def tile_allocs(template, n_units):
    """The allocation list of n_units units that all carry the same template (a list of PdschAlloc for one unit): a ctypes array of
    n_units * len(template) structs with `unit` = 0 .. n_units - 1, built by numpy (half a million Python-made structs take a minute)."""
    k, sz = len(template), C.sizeof(PdschAlloc)
    base = np.frombuffer((PdschAlloc * k)(*template), np.uint8).reshape(k, sz)
    full = np.ascontiguousarray(np.broadcast_to(base, (n_units, k, sz))).reshape(n_units * k, sz)
    off = PdschAlloc.unit.offset
    full[:, off:off + 4] = np.repeat(np.arange(n_units, dtype=np.uint32), k).view(np.uint8).reshape(-1, 4)
    return (PdschAlloc * (n_units * k)).from_buffer(full)


class PipelineDevStats(C.Structure):
    """mi_lte_pipeline_dev_stats"""
    _fields_ = [("device", C.c_uint32), ("chunks", C.c_uint32), ("units", C.c_uint32), ("n_cpus", C.c_uint32), ("numa_node", C.c_int32), ("reserved", C.c_int32),
                ("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("wall_s", C.c_double), ("h2d_s", C.c_double), ("kernel_s", C.c_double), ("d2h_s", C.c_double)]


class PucchRes(C.Structure):
    """mi_lte_pucch_res"""
    _fields_ = [("unit", C.c_uint32), ("format", C.c_uint32), ("N_1_p_pucch", C.c_uint32)]


class PdcchDci(C.Structure):
    """mi_lte_pdcch_dci"""
    _fields_ = [(n, C.c_uint32) for n in ("rnti", "format", "candidate", "n_bits", "payload", "mcs", "alloc_valid", "reserved")] + [("alloc", PdschAlloc)]


class CoarseTiming(C.Structure):
    """mi_lte_coarse_timing = LIBLTE_PHY_COARSE_TIMING_STRUCT"""
    _fields_ = [("freq_offset", C.c_float * 5), ("symb_starts", (C.c_uint32 * 7) * 5), ("n_corr_peaks", C.c_uint32)]


class TxAlloc(C.Structure):
    """mi_lte_tx_alloc: LIBLTE_PHY_ALLOCATION_STRUCT as the transmit functions read it"""
    _fields_ = [("msg", C.POINTER(C.c_uint8) * 2), ("msg_bits", C.c_uint32 * 2)] + \
               [(n, C.c_uint32) for n in ("pre_coder_type", "mod_type", "chan_type", "tbs", "rv_idx", "N_prb")] + [("prb", (C.c_uint32 * 110) * 2)] + \
               [(n, C.c_uint32) for n in ("N_codewords", "N_layers", "tx_mode", "rnti", "mcs", "tpc", "ndi", "dl_alloc")]


class MiLteError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libmi_lte.so")


def build_library():
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")])
    return library_path()


def load_library():
    """Load libmi_lte.so; raises if it has not been built -- there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise MiLteError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(the HIP extension is the only implementation; no CPU fallback exists)" % path)
    L = C.CDLL(path)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    L.mi_lte_version.restype = C.c_int
    L.mi_lte_device_count.restype = C.c_int
    L.mi_lte_build_id.restype = C.c_char_p
    L.mi_lte_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mi_lte_ctx_destroy.argtypes = [vp]
    L.mi_lte_last_error.argtypes = [vp]
    L.mi_lte_last_error.restype = C.c_char_p
    L.mi_lte_device_name.argtypes = [vp]
    L.mi_lte_device_name.restype = C.c_char_p
    L.mi_lte_last_kernels.argtypes = [vp]
    L.mi_lte_last_kernels.restype = C.c_char_p
    L.mi_lte_stream.argtypes = [vp]
    L.mi_lte_stream.restype = vp
    L.mi_lte_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.mi_lte_free.argtypes = [vp, vp]
    L.mi_lte_memset.argtypes = [vp, vp, C.c_int, sz]
    L.mi_lte_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.mi_lte_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.mi_lte_sync.argtypes = [vp]
    L.mi_lte_timer_start.argtypes = [vp]
    L.mi_lte_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.mi_lte_profile_enable.argtypes = [vp, C.c_int]
    L.mi_lte_profile_reset.argtypes = [vp]
    L.mi_lte_profile_report.argtypes = [vp]
    L.mi_lte_profile_report.restype = C.c_char_p
    L.mi_lte_subframe_floats.argtypes = [u32]
    L.mi_lte_subframe_floats.restype = sz
    L.mi_lte_dl_frontend_batch.argtypes = [vp, C.POINTER(DlCfg), vp, vp, vp, vp, vp, u32, vp]
    L.mi_lte_pdsch_plan_create.argtypes = [vp, C.POINTER(DlCfg), u32, vp, u32, C.POINTER(vp)]
    L.mi_lte_pdsch_plan_destroy.argtypes = [vp, vp]
    L.mi_lte_pdsch_plan_set_decoder.argtypes = [vp, u32, u32, C.c_int]
    L.mi_lte_pdsch_plan_set_output.argtypes = [vp, u32]
    L.mi_lte_host_alloc.argtypes = [sz]
    L.mi_lte_host_alloc.restype = vp
    L.mi_lte_host_alloc_on.argtypes = [C.c_int, sz]
    L.mi_lte_host_alloc_on.restype = vp
    L.mi_lte_host_alloc_node.argtypes = [vp]
    L.mi_lte_host_alloc_node.restype = C.c_int
    L.mi_lte_device_numa_node.argtypes = [C.c_int]
    L.mi_lte_device_numa_node.restype = C.c_int
    L.mi_lte_dl_pipeline_device_stats.argtypes = [vp, u32, C.POINTER(PipelineDevStats)]
    L.mi_lte_host_free.argtypes = [vp]
    L.mi_lte_pdsch_plan_create_dynamic.argtypes = [vp, C.POINTER(DlCfg), u32, sz, C.POINTER(vp)]
    L.mi_lte_pdsch_plan_assign.argtypes = [vp, vp, u32, vp, u32]
    L.mi_lte_pdsch_plan_n_alloc.argtypes = [vp]
    L.mi_lte_pdsch_plan_n_alloc.restype = u32
    L.mi_lte_pdsch_alloc_decodable.argtypes = [C.POINTER(DlCfg), C.POINTER(PdschAlloc), u32]
    L.mi_lte_pdsch_alloc_decodable.restype = C.c_int
    L.mi_lte_iq_i8_to_planar.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.mi_lte_freq_shift_run.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_float, u32]
    L.mi_lte_dl_pipeline_create.argtypes = [C.c_int, C.POINTER(DlCfg), u32, vp, u32, u32, u32, C.POINTER(vp)]
    L.mi_lte_dl_pipeline_create_multi.argtypes = [C.POINTER(C.c_int), u32, C.POINTER(DlCfg), u32, vp, u32, sz, u32, u32, C.POINTER(vp)]
    L.mi_lte_dl_pipeline_n_devices.argtypes = [vp]
    L.mi_lte_dl_pipeline_n_devices.restype = u32
    L.mi_lte_dl_pipeline_run_units.argtypes = [vp, vp, vp, vp, u32, vp, vp, u32, vp, vp]
    L.mi_lte_dl_pipeline_run_capture.argtypes = [vp, vp, C.c_uint64, C.c_uint64, u32, u32, u32, vp, vp, u32, vp, vp]
    L.mi_lte_dl_pipeline_destroy.argtypes = [vp]
    L.mi_lte_dl_pipeline_out_stride.argtypes = [vp]
    L.mi_lte_dl_pipeline_out_stride.restype = u32
    L.mi_lte_dl_pipeline_unit_samples.argtypes = [vp]
    L.mi_lte_dl_pipeline_unit_samples.restype = sz
    L.mi_lte_dl_pipeline_last_error.argtypes = [vp]
    L.mi_lte_dl_pipeline_last_error.restype = C.c_char_p
    L.mi_lte_dl_pipeline_run.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    L.mi_lte_pdsch_plan_out_stride.argtypes = [vp]
    L.mi_lte_pdsch_plan_out_stride.restype = u32
    L.mi_lte_pdsch_decode_run.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.mi_lte_pdsch_plan_soft_bits.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(vp)]
    L.mi_lte_ul_subframe_floats.restype = sz
    L.mi_lte_ul_frontend_batch.argtypes = [vp, C.POINTER(DlCfg), vp, vp, vp, u32, vp]
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L.mi_lte_ul_dmrs_pusch.argtypes = [C.POINTER(UlCfg), u32, u32, u32, f32p, f32p, f32p, f32p]
    L.mi_lte_pusch_plan_create.argtypes = [vp, C.POINTER(DlCfg), C.POINTER(UlCfg), u32p, u32p, u32, vp, u32, C.POINTER(vp)]
    L.mi_lte_pusch_plan_destroy.argtypes = [vp, vp]
    L.mi_lte_pusch_plan_out_stride.argtypes = [vp]
    L.mi_lte_pusch_plan_out_stride.restype = u32
    L.mi_lte_pusch_decode_run.argtypes = [vp, vp, vp, vp, vp]
    L.mi_lte_pusch_plan_soft_bits.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u32)]
    L.mi_lte_prach_plan_create.argtypes = [vp, C.POINTER(DlCfg), C.POINTER(PrachCfg), C.POINTER(vp)]
    L.mi_lte_prach_plan_create_roots.argtypes = [vp, C.POINTER(DlCfg), C.POINTER(PrachCfg), f32p, f32p, u32, C.POINTER(vp)]
    L.mi_lte_prach_plan_destroy.argtypes = [vp, vp]
    L.mi_lte_prach_plan_n_roots.argtypes = [vp]
    L.mi_lte_prach_plan_n_roots.restype = u32
    L.mi_lte_prach_occasion_samples.argtypes = [vp]
    L.mi_lte_prach_occasion_samples.restype = u32
    L.mi_lte_prach_detect_run.argtypes = [vp, vp, vp, vp, vp, u32, u32p, u32p, u32p]
    L.mi_lte_prach_detect_launch.argtypes = [vp, vp, vp, vp, vp, u32]
    L.mi_lte_prach_detect_fetch.argtypes = [vp, vp, u32p, u32p, u32p, u32]
    L.mi_lte_device_copy_rate.argtypes = [vp, C.c_size_t, u32, C.POINTER(C.c_double)]
    L.mi_lte_device_copy_rates.argtypes = [vp, C.POINTER(C.c_double)]
    L.mi_lte_pdcch_plan_create.argtypes = [vp, C.POINTER(DlCfg), C.c_float, u32, u32, u32p, u32, C.POINTER(vp)]
    L.mi_lte_pdcch_plan_destroy.argtypes = [vp, vp]
    L.mi_lte_pdcch_decode_run.argtypes = [vp, vp, vp, vp, vp, u32, u32p, u32p, u32p, u32p, C.POINTER(PdcchDci)]
    L.mi_lte_pucch_decode_run.argtypes = [vp, u32, u32, vp, C.POINTER(PucchRes), f32p, u32, u8p, u32p, u32p]
    L.mi_lte_coarse_timing_samples.argtypes = [u32, u32]
    L.mi_lte_coarse_timing_samples.restype = C.c_size_t
    L.mi_lte_coarse_timing_run.argtypes = [vp, C.POINTER(DlCfg), vp, vp, C.c_uint64, u32, C.POINTER(CoarseTiming)]
    L.mi_lte_find_pss_run.argtypes = [vp, C.POINTER(DlCfg), vp, vp, C.c_uint64, u32p, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mi_lte_find_sss_run.argtypes = [vp, C.POINTER(DlCfg), vp, vp, C.c_uint64, u32, u32p, C.c_float, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.mi_lte_pbch_decode_run.argtypes = [vp, C.POINTER(DlCfg), vp, vp, u32, u32p, u32p, u32p]
    L.mi_lte_pdcch_re_tables.argtypes = [u32, u32, u32, C.c_float, u32, u32p, u32p]
    L.mi_lte_dci_1a_unpack.argtypes = [u32, u32, u32, u32, u32, C.POINTER(PdcchDci)]
    L.mi_lte_dci_1c_unpack.argtypes = [u32, u32, u32, u32, u32, C.POINTER(PdcchDci)]
    L.mi_lte_turbo_decode_batch.argtypes = [vp, vp, C.c_int, u32, u32, C.c_int, u32, C.c_int, vp]
    L.mi_lte_set_turbo_small_batch.argtypes = [vp, u32]
    L.mi_lte_turbo_early_exit_iterations.argtypes = [vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]
    L.mi_lte_turbo_scratch_bytes.argtypes = [u32, u32]
    L.mi_lte_turbo_scratch_bytes.restype = sz
    L.mi_lte_tbs.argtypes = [u32, u32]
    L.mi_lte_tbs.restype = u32
    # the transmit side (host code): the handle and the grid functions
    L.mi_lte_tx_create.argtypes = [C.POINTER(vp)]
    L.mi_lte_tx_destroy.argtypes = [vp]
    L.mi_lte_pdsch_channel_encode.argtypes = [vp, u32, u32, C.POINTER(TxAlloc), u32, u32, u32, u32, u32, f32p, f32p]
    L.mi_lte_bch_channel_encode.argtypes = [vp, u32, u32, u8p, u32, u32, u32, u32, f32p, f32p]
    L.mi_lte_map_crs.argtypes = [u32, u32, u32, u32, u32, f32p, f32p]
    L.mi_lte_map_pss.argtypes = [u32, u32, u32, u32, f32p, f32p]
    L.mi_lte_map_sss.argtypes = [u32, u32, u32, u32, u32, u32, f32p, f32p]
    L.mi_lte_create_dl_subframe.argtypes = [u32, u32, u32, u32, f32p, f32p, u32, f32p, f32p]
    _LIB = L
    return L


class DeviceBuffer:
    """A block of HBM owned by a Context (hipMalloc through the C-ABI)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._check(ctx.L.mi_lte_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.L.mi_lte_memcpy_h2d(self.ctx.h, self.ptr + offset, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, count=None, offset=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dtype.itemsize
        out = np.empty(count, dtype)
        self.ctx._check(self.ctx.L.mi_lte_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def zero(self):
        self.ctx._check(self.ctx.L.mi_lte_memset(self.ctx.h, self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            self.ctx.L.mi_lte_free(self.ctx.h, self.ptr)
            self.ptr = None


class PdschPlan:
    """mi_lte_pdsch_plan: device copy of an allocation list, grouped by code-block size.  allocs=None: a dynamic plan of the given
    capacity (mi_lte_pdsch_plan_create_dynamic), filled and re-filled by assign()."""

    def __init__(self, ctx, cfg, n_pdcch_symbs, allocs, max_alloc=0, max_soft_bytes=0):
        self.ctx = ctx
        h = C.c_void_p()
        if allocs is None:
            ctx._check(ctx.L.mi_lte_pdsch_plan_create_dynamic(ctx.h, C.byref(cfg), max_alloc, max_soft_bytes, C.byref(h)))
            self.h, self.n_alloc, self.tbs = h, 0, []
        else:
            self.n_alloc = len(allocs)
            arr = allocs if isinstance(allocs, C.Array) else (PdschAlloc * len(allocs))(*allocs)  # (a ready-made ctypes array: tile_allocs below)
            ctx._check(ctx.L.mi_lte_pdsch_plan_create(ctx.h, C.byref(cfg), n_pdcch_symbs, C.cast(arr, C.c_void_p), len(allocs),
                                                      C.byref(h)))
            self.h = h
            self.tbs = None if isinstance(allocs, C.Array) else [a.tbs for a in allocs]  # (a ctypes array: read on demand, run() below)
            self._arr = arr
        self.out_stride = ctx.L.mi_lte_pdsch_plan_out_stride(h)

    def assign(self, n_pdcch_symbs, allocs):
        """Re-plan a dynamic plan (no allocation, no wait); allocations may carry their own n_pdcch_symbs."""
        arr = (PdschAlloc * len(allocs))(*allocs)
        self.ctx._check(self.ctx.L.mi_lte_pdsch_plan_assign(self.ctx.h, self.h, n_pdcch_symbs, C.cast(arr, C.c_void_p), len(allocs)))
        self.n_alloc, self.tbs = len(allocs), [a.tbs for a in allocs]
        self.out_stride = self.ctx.L.mi_lte_pdsch_plan_out_stride(self.h)

    def set_decoder(self, mode, n_iter=8, qpp_spec=0):
        """TURBO_REF (default, the reference's decoder) or TURBO_BCJR (max-log-MAP, n_iter iterations)."""
        self.ctx._check(self.ctx.L.mi_lte_pdsch_plan_set_decoder(self.h, mode, n_iter, qpp_spec))

    def set_packed(self, packed=True):
        """Eight bits per byte (first bit in the most significant position) instead of the reference's one bit per byte; changes out_stride."""
        self.ctx._check(self.ctx.L.mi_lte_pdsch_plan_set_output(self.h, 1 if packed else 0))
        self.packed = bool(packed)
        self.out_stride = self.ctx.L.mi_lte_pdsch_plan_out_stride(self.h)

    def run_dev(self, d_subframes, d_sf, d_cell, d_out, d_status):
        self.ctx._check(self.ctx.L.mi_lte_pdsch_decode_run(self.ctx.h, self.h, d_subframes.ptr, d_sf.ptr, d_cell.ptr,
                                                           d_out.ptr, d_status.ptr))

    def run(self, d_subframes, subfr_num, n_id_cell):
        """Returns (status int32 [n_alloc], list of uint8 bit arrays)."""
        ctx = self.ctx
        d_sf, d_cell = ctx.to_device(np.asarray(subfr_num, np.uint32)), ctx.to_device(np.asarray(n_id_cell, np.uint32))
        d_out, d_st = ctx.alloc(self.n_alloc * self.out_stride), ctx.alloc(4 * self.n_alloc)
        d_out.zero()
        try:
            self.run_dev(d_subframes, d_sf, d_cell, d_out, d_st)
            st = d_st.download(np.int32)
            bits = d_out.download(np.uint8).reshape(self.n_alloc, self.out_stride)
            if getattr(self, "packed", False):  # back to one bit per byte for the caller
                return st, [np.unpackbits(bits[a, :(self._tbs()[a] + 7) // 8])[:self._tbs()[a]] for a in range(self.n_alloc)]
            return st, [bits[a, :self._tbs()[a]] for a in range(self.n_alloc)]
        finally:
            for b in (d_sf, d_cell, d_out, d_st):
                b.free()

    def _tbs(self):
        if self.tbs is None:
            self.tbs = [a.tbs for a in self._arr]
        return self.tbs

    def soft_bits(self, alloc):
        """Descrambled int8 soft bits of one allocation (stage tap)."""
        pe, pn = C.c_void_p(), C.c_void_p()
        self.ctx._check(self.ctx.L.mi_lte_pdsch_plan_soft_bits(self.h, alloc, C.byref(pe), C.byref(pn)))
        n = np.empty(1, np.uint32)
        self.ctx._check(self.ctx.L.mi_lte_memcpy_d2h(self.ctx.h, n.ctypes.data, pn.value, 4))
        out = np.empty(int(n[0]), np.int8)
        self.ctx._check(self.ctx.L.mi_lte_memcpy_d2h(self.ctx.h, out.ctypes.data, pe.value, out.nbytes))
        return out

    def close(self):
        if self.h:
            self.ctx.L.mi_lte_pdsch_plan_destroy(self.ctx.h, self.h)
            self.h = None


class HostBuffer:
    """Pinned host memory (mi_lte_host_alloc) viewed as a numpy array: what the host-batch pipeline wants its arrays in."""

    def __init__(self, shape, dtype, device=None):
        """device: bind the pages to that device's NUMA node (mi_lte_host_alloc_on); self.node says whether that happened (-1: no)."""
        self.L = load_library()
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        self.ptr = self.L.mi_lte_host_alloc(max(n, 1)) if device is None else self.L.mi_lte_host_alloc_on(device, max(n, 1))
        if not self.ptr:
            raise MiLteError("mi_lte_host_alloc(%d) failed" % n)
        self.node = self.L.mi_lte_host_alloc_node(self.ptr)
        self.arr = np.frombuffer((C.c_uint8 * max(n, 1)).from_address(self.ptr), dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr:
            self.arr = None
            self.L.mi_lte_host_free(self.ptr)
            self.ptr = None


class DlPipeline:
    """mi_lte_dl_pipeline: whole-chain batches from host buffers, chunks overlapped on several lanes."""

    def __init__(self, device, cfg, n_pdcch_symbs, unit_allocs, chunk_units, n_lanes=3, max_alloc_per_unit=0, max_soft_bytes_per_unit=0):
        """device: one ordinal or a list of them (chunk c of a run goes to devices[c % len]); unit_allocs: the allocation template every
        unit carries, or None for per-unit lists only (then max_alloc_per_unit bounds a unit's list)."""
        self.L = load_library()
        devs = list(device) if isinstance(device, (list, tuple)) else [device]
        darr = (C.c_int * len(devs))(*devs)
        h = C.c_void_p()
        if unit_allocs is not None:
            arr = (PdschAlloc * len(unit_allocs))(*unit_allocs)
            rc = self.L.mi_lte_dl_pipeline_create_multi(darr, len(devs), C.byref(cfg), n_pdcch_symbs, C.cast(arr, C.c_void_p), len(unit_allocs),
                                                        max_soft_bytes_per_unit, chunk_units, n_lanes, C.byref(h))
        else:
            rc = self.L.mi_lte_dl_pipeline_create_multi(darr, len(devs), C.byref(cfg), n_pdcch_symbs, None, max_alloc_per_unit, max_soft_bytes_per_unit,
                                                        chunk_units, n_lanes, C.byref(h))
        if rc != 0:
            raise MiLteError("mi_lte_dl_pipeline_create_multi failed: %d" % rc)
        self.h, self.n_alloc = h, len(unit_allocs) if unit_allocs is not None else 0
        self.out_stride = self.L.mi_lte_dl_pipeline_out_stride(h)
        self.unit_samples = self.L.mi_lte_dl_pipeline_unit_samples(h)
        self.n_devices = self.L.mi_lte_dl_pipeline_n_devices(h)
        self.tbs = [a.tbs for a in unit_allocs] if unit_allocs is not None else []

    def _err(self, what, rc):
        raise MiLteError("%s failed: %d (%s)" % (what, rc, self.L.mi_lte_dl_pipeline_last_error(self.h).decode()))

    def device_stats(self):
        """What every device slot did in the last run (mi_lte_dl_pipeline_device_stats), as a list of dicts."""
        out = []
        for i in range(self.n_devices):
            st = PipelineDevStats()
            if self.L.mi_lte_dl_pipeline_device_stats(self.h, i, C.byref(st)) != 0:
                raise MiLteError("mi_lte_dl_pipeline_device_stats failed")
            out.append({k: getattr(st, k) for k, _ in PipelineDevStats._fields_ if k != "reserved"})
        return out

    def run_units(self, h_iq, h_sf, h_cell, n_units, allocs, first, n_pdcch_symbs, h_out, h_status):
        """Per-unit allocation lists: allocs (ctypes array of PdschAlloc, sorted by unit), first uint32 [n_units + 1]."""
        first = np.ascontiguousarray(first, np.uint32)
        rc = self.L.mi_lte_dl_pipeline_run_units(self.h, h_iq.ctypes.data, h_sf.ctypes.data, h_cell.ctypes.data, n_units, C.cast(allocs, C.c_void_p),
                                                 first.ctypes.data, n_pdcch_symbs, h_out.ctypes.data, h_status.ctypes.data)
        if rc != 0:
            self._err("mi_lte_dl_pipeline_run_units", rc)

    def run_capture(self, h_capture, first_start, n_subframes, first_sf, cell, allocs, first, n_pdcch_symbs, h_out, h_status):
        """One contiguous int8 capture [n_samples, 2], split on subframe boundaries with the look-ahead halo."""
        first = np.ascontiguousarray(first, np.uint32)
        rc = self.L.mi_lte_dl_pipeline_run_capture(self.h, h_capture.ctypes.data, h_capture.shape[0], first_start, n_subframes, first_sf, cell,
                                                   C.cast(allocs, C.c_void_p), first.ctypes.data, n_pdcch_symbs, h_out.ctypes.data, h_status.ctypes.data)
        if rc != 0:
            self._err("mi_lte_dl_pipeline_run_capture", rc)

    def run(self, h_iq, h_sf, h_cell, n_units, h_out, h_status):
        """h_iq int8 [n_units, unit_samples, 2], h_sf / h_cell uint32 [n_units], h_out uint8 [n_units * n_alloc, out_stride], h_status int32:
        numpy arrays, ideally views of HostBuffer (pinned)."""
        rc = self.L.mi_lte_dl_pipeline_run(self.h, h_iq.ctypes.data, h_sf.ctypes.data, h_cell.ctypes.data, n_units, h_out.ctypes.data, h_status.ctypes.data)
        if rc != 0:
            raise MiLteError("mi_lte_dl_pipeline_run failed: %d (%s)" % (rc, self.L.mi_lte_dl_pipeline_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.L.mi_lte_dl_pipeline_destroy(self.h)
            self.h = None


class PuschPlan:
    """mi_lte_pusch_plan: PUSCH allocations (one per scheduled UE) over a batch of uplink subframe units."""

    def __init__(self, ctx, cfg, ulcfg, unit_subfr_num, unit_n_id_cell, allocs):
        self.ctx, self.n_alloc = ctx, len(allocs)
        arr = (PdschAlloc * len(allocs))(*allocs)
        h = C.c_void_p()
        sf, cell = np.ascontiguousarray(unit_subfr_num, np.uint32), np.ascontiguousarray(unit_n_id_cell, np.uint32)
        ctx._check(ctx.L.mi_lte_pusch_plan_create(ctx.h, C.byref(cfg), C.byref(ulcfg), sf, cell, len(sf), C.cast(arr, C.c_void_p),
                                                  len(allocs), C.byref(h)))
        self.h = h
        self.out_stride = ctx.L.mi_lte_pusch_plan_out_stride(h)
        self.tbs = [a.tbs for a in allocs]

    def run_dev(self, d_subframes, d_out, d_status):
        self.ctx._check(self.ctx.L.mi_lte_pusch_decode_run(self.ctx.h, self.h, d_subframes.ptr, d_out.ptr, d_status.ptr))

    def run(self, d_subframes):
        """Returns (status int32 [n_alloc], list of uint8 bit arrays)."""
        ctx = self.ctx
        d_out, d_st = ctx.alloc(self.n_alloc * self.out_stride), ctx.alloc(4 * self.n_alloc)
        d_out.zero()
        try:
            self.run_dev(d_subframes, d_out, d_st)
            st = d_st.download(np.int32)
            bits = d_out.download(np.uint8).reshape(self.n_alloc, self.out_stride)
            return st, [bits[a, :self.tbs[a]] for a in range(self.n_alloc)]
        finally:
            d_out.free()
            d_st.free()

    def soft_bits(self, alloc):
        """De-interleaved, descrambled int8 soft bits of one allocation (stage tap)."""
        pe, n = C.c_void_p(), C.c_uint32()
        self.ctx._check(self.ctx.L.mi_lte_pusch_plan_soft_bits(self.h, alloc, C.byref(pe), C.byref(n)))
        out = np.empty(int(n.value), np.int8)
        self.ctx._check(self.ctx.L.mi_lte_memcpy_d2h(self.ctx.h, out.ctypes.data, pe.value, out.nbytes))
        return out

    def close(self):
        if self.h:
            self.ctx.L.mi_lte_pusch_plan_destroy(self.ctx.h, self.h)
            self.h = None


class PrachPlan:
    """mi_lte_prach_plan: root-sequence spectra of one cell's PRACH configuration."""

    def __init__(self, ctx, cfg, prach_cfg, roots_fft=None):
        self.ctx = ctx
        h = C.c_void_p()
        if roots_fft is None:
            ctx._check(ctx.L.mi_lte_prach_plan_create(ctx.h, C.byref(cfg), C.byref(prach_cfg), C.byref(h)))
        else:
            re, im = np.ascontiguousarray(roots_fft[0], np.float32), np.ascontiguousarray(roots_fft[1], np.float32)
            ctx._check(ctx.L.mi_lte_prach_plan_create_roots(ctx.h, C.byref(cfg), C.byref(prach_cfg), re, im, re.shape[0], C.byref(h)))
        self.h = h
        self.n_roots = ctx.L.mi_lte_prach_plan_n_roots(h)
        self.occasion_samples = ctx.L.mi_lte_prach_occasion_samples(h)

    def detect_dev(self, d_a, d_b, d_start, n_occ):
        """(N_det_pre, det_pre, det_ta) uint32 arrays, one entry per occasion."""
        out = [np.zeros(n_occ, np.uint32) for _ in range(3)]
        self.ctx._check(self.ctx.L.mi_lte_prach_detect_run(self.ctx.h, self.h, d_a.ptr, d_b.ptr if d_b is not None else None, d_start.ptr,
                                                            n_occ, out[0], out[1], out[2]))
        return out

    def launch_dev(self, d_a, d_b, d_start, n_occ):
        """Queue the detection of n_occ occasions and return at once (mi_lte_prach_detect_launch); fetch() has the verdicts."""
        self.ctx._check(self.ctx.L.mi_lte_prach_detect_launch(self.ctx.h, self.h, d_a.ptr, d_b.ptr if d_b is not None else None, d_start.ptr, n_occ))
        self._pending = n_occ

    def fetch(self):
        """(N_det_pre, det_pre, det_ta) of the launch before it (waits for that launch only)."""
        n = self._pending
        out = [np.zeros(n, np.uint32) for _ in range(3)]
        self.ctx._check(self.ctx.L.mi_lte_prach_detect_fetch(self.ctx.h, self.h, out[0], out[1], out[2], n))
        self._pending = 0
        return out

    def detect(self, iq, occ_start):
        d_a, d_s = self.ctx.to_device(iq.astype(np.int8)), self.ctx.to_device(np.asarray(occ_start, np.uint64))
        try:
            return self.detect_dev(d_a, None, d_s, len(occ_start))
        finally:
            d_a.free()
            d_s.free()

    def close(self):
        if self.h:
            self.ctx.L.mi_lte_prach_plan_destroy(self.ctx.h, self.h)
            self.h = None


class PdcchPlan:
    """mi_lte_pdcch_plan: PCFICH / PDCCH resource-element tables of the listed cells."""

    def __init__(self, ctx, cfg, cells, phich_res=1.0, phich_dur_extended=0, per_port_estimates=False):
        self.ctx = ctx
        h = C.c_void_p()
        cells = np.ascontiguousarray(cells, np.uint32)
        ctx._check(ctx.L.mi_lte_pdcch_plan_create(ctx.h, C.byref(cfg), float(phich_res), int(phich_dur_extended), 1 if per_port_estimates else 0, cells,
                                                  len(cells), C.byref(h)))
        self.h = h

    def decode_raw(self, d_subframes, d_sf, d_cell, n_units):
        """(rc[n], cfi[n], n_symbs[n], n_dci[n], ctypes array PdcchDci[n * 6]); the buffers are the plan's and are reused by the next call."""
        if getattr(self, "_n", None) != n_units:
            self._n = n_units
            self._out = [np.zeros(n_units, np.uint32) for _ in range(4)]
            self._dci = (PdcchDci * (6 * n_units))()
        rc, cfi, nsym, ndci = self._out
        self.ctx._check(self.ctx.L.mi_lte_pdcch_decode_run(self.ctx.h, self.h, d_subframes.ptr, d_sf.ptr, d_cell.ptr, n_units, rc, cfi, nsym, ndci,
                                                            self._dci))
        return rc, cfi, nsym, ndci, self._dci

    def decode_dev(self, d_subframes, d_sf, d_cell, n_units):
        """(rc[n], cfi[n], n_symbs[n], list of per-unit lists of PdcchDci)"""
        rc, cfi, nsym, ndci, dci = self.decode_raw(d_subframes, d_sf, d_cell, n_units)
        return rc.copy(), cfi.copy(), nsym.copy(), [[dci[6 * u + k] for k in range(int(ndci[u]))] for u in range(n_units)]

    def close(self):
        if self.h:
            self.ctx.L.mi_lte_pdcch_plan_destroy(self.ctx.h, self.h)
            self.h = None


class Transmitter:
    """The host-side transmit functions of one cell (mi_lte_tx): the reference's liblte_phy_map_crs / _pss / _sss, _bch_channel_encode,
    _pdsch_channel_encode and _create_dl_subframe on a grid of the reference's layout (tx_symb_re / _im: [4][16][1200]).  Host code, no GPU."""

    def __init__(self, fft_size, n_rb_dl, n_id_cell, n_ant=1):
        self.L = load_library()
        self.fft, self.n_rb, self.cell, self.n_ant = fft_size, n_rb_dl, n_id_cell, n_ant
        self.h = C.c_void_p()
        if self.L.mi_lte_tx_create(C.byref(self.h)) != 0:
            raise MiLteError("mi_lte_tx_create failed")
        self.re, self.im = np.zeros((4, 16, 1200), np.float32), np.zeros((4, 16, 1200), np.float32)
        self._keep = []

    def _g(self):
        return self.re, self.im

    def _ok(self, rc, what):
        if rc != 0:
            raise MiLteError("%s: %d (the reference's LIBLTE_ERROR_INVALID_INPUTS)" % (what, rc))

    def clear(self):
        self.re[:], self.im[:] = 0, 0

    def signals(self, subfr_num):
        """CRS in every subframe, PSS / SSS in subframes 0 and 5 (what LTE_fdd_dl_file_gen maps first)."""
        re, im = self._g()
        if subfr_num in (0, 5):
            self._ok(self.L.mi_lte_map_pss(self.n_rb, 12, self.cell % 3, self.n_ant, re, im), "mi_lte_map_pss")
            self._ok(self.L.mi_lte_map_sss(self.n_rb, 12, subfr_num, self.cell // 3, self.cell % 3, self.n_ant, re, im), "mi_lte_map_sss")
        self._ok(self.L.mi_lte_map_crs(self.n_rb, 12, subfr_num, self.cell, self.n_ant, re, im), "mi_lte_map_crs")

    def bch(self, mib_bits, sfn):
        re, im = self._g()
        b = np.ascontiguousarray(mib_bits, np.uint8)
        self._ok(self.L.mi_lte_bch_channel_encode(self.h, self.n_rb, 12, b, len(b), self.cell, self.n_ant, sfn, re, im), "mi_lte_bch_channel_encode")

    def pdsch(self, subfr_num, n_pdcch_symbs, allocs):
        """allocs: (PdschAlloc, transport-block bits) pairs -- the receive side's own allocation struct, so that the same list goes to a PDSCH plan."""
        arr = (TxAlloc * len(allocs))()
        self._keep = []
        for t, (a, bits) in zip(arr, allocs):
            b = np.ascontiguousarray(bits, np.uint8)
            self._keep.append(b)
            t.msg[0], t.msg_bits[0] = b.ctypes.data_as(C.POINTER(C.c_uint8)), len(b)
            t.pre_coder_type, t.mod_type, t.chan_type, t.tbs, t.rv_idx, t.N_prb = 0, a.mod_type, 0, a.tbs, a.rv_idx, a.N_prb
            t.N_codewords, t.N_layers, t.tx_mode, t.rnti = 1, 1, a.tx_mode, a.rnti
            for s in range(2):
                for i in range(a.N_prb):
                    t.prb[s][i] = a.prb[s][i]
        re, im = self._g()
        self._ok(self.L.mi_lte_pdsch_channel_encode(self.h, self.n_rb, 12, arr, len(allocs), n_pdcch_symbs, self.cell, self.n_ant, subfr_num, re, im), "mi_lte_pdsch_channel_encode")

    def samples(self, ant=0):
        """OFDM modulation of the grid's 14 symbols: (i, q) float32 of one subframe."""
        s = self.fft // 128
        n = 14 * self.fft + 2 * 10 * s + 12 * 9 * s
        i, q = np.zeros(n, np.float32), np.zeros(n, np.float32)
        re, im = self._g()
        self._ok(self.L.mi_lte_create_dl_subframe(self.fft, 12 * self.n_rb, 10 * s, 9 * s, re, im, ant, i, q),
                 "mi_lte_create_dl_subframe")
        return i, q

    def close(self):
        if self.h:
            self.L.mi_lte_tx_destroy(self.h)
            self.h = None


def pdcch_re_tables(n_rb_dl, n_ant, cell, phich_res, n_symbs):
    """(pcfich[16], cand[6, 288]) uint32 grid indices l*1200 + k (host function of the library)."""
    pc, cand = np.zeros(16, np.uint32), np.zeros((6, 288), np.uint32)
    rc = load_library().mi_lte_pdcch_re_tables(n_rb_dl, n_ant, cell, float(phich_res), n_symbs, pc, cand)
    if rc != 0:
        raise MiLteError("mi_lte_pdcch_re_tables failed: %d" % rc)
    return pc, cand


def dci_unpack(fmt, payload, n_bits, rnti, n_rb_dl, n_ant):
    """(rc, PdcchDci): the library's DCI 1A (fmt 0) / 1C (fmt 1) unpacker (host arithmetic)."""
    d = PdcchDci()
    L = load_library()
    f = L.mi_lte_dci_1a_unpack if fmt == 0 else L.mi_lte_dci_1c_unpack
    return f(int(payload), int(n_bits), int(rnti), int(n_rb_dl), int(n_ant), C.byref(d)), d


def ul_dmrs_pusch(ulcfg, n_id_cell, n_subfr, n_prb):
    """float32 [4, 12*n_prb]: dmrs_0_re, dmrs_0_im, dmrs_1_re, dmrs_1_im (host function of the library)."""
    out = np.zeros((4, 12 * n_prb), np.float32)
    rc = load_library().mi_lte_ul_dmrs_pusch(C.byref(ulcfg), n_id_cell, n_subfr, n_prb, out[0], out[1], out[2], out[3])
    if rc != 0:
        raise MiLteError("mi_lte_ul_dmrs_pusch failed: %d" % rc)
    return out


class Context:
    """One GPU + one stream.  Mirrors the role of LIBLTE_PHY_STRUCT: all scratch lives here."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.mi_lte_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise MiLteError("mi_lte_ctx_create(device=%d) failed with %d: no usable gfx950 GPU "
                             "(the product path has no CPU fallback)" % (device, rc))
        self.h = h
        self.device = device

    def _check(self, rc):
        if rc != 0:
            raise MiLteError("libmi_lte error %d: %s" % (rc, self.L.mi_lte_last_error(self.h).decode()))

    @property
    def device_name(self):
        return self.L.mi_lte_device_name(self.h).decode()

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.nbytes).upload(arr)

    def sync(self):
        self._check(self.L.mi_lte_sync(self.h))

    def timer_start(self):
        self._check(self.L.mi_lte_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.L.mi_lte_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, on=True):
        self._check(self.L.mi_lte_profile_enable(self.h, 1 if on else 0))
        self._check(self.L.mi_lte_profile_reset(self.h))

    def profile_report(self):
        """{kernel: (launches, total_ms)} for every launch bracketed since profile() / the last reset."""
        out = {}
        for item in self.L.mi_lte_profile_report(self.h).decode().split(";"):
            if item:
                name, n, ms = item.rsplit(":", 2)
                out[name] = (int(n), float(ms))
        return out

    def last_kernels(self):
        return self.L.mi_lte_last_kernels(self.h).decode()

    # ---- front end --------------------------------------------------------------------------
    def subframe_floats(self, n_ant):
        return self.L.mi_lte_subframe_floats(n_ant)

    def dl_frontend_dev(self, cfg, d_a, d_b, d_start, d_sf, d_cell, n_units, d_subframes):
        self._check(self.L.mi_lte_dl_frontend_batch(self.h, C.byref(cfg), d_a.ptr, d_b.ptr if d_b is not None else None,
                                                    d_start.ptr, d_sf.ptr, d_cell.ptr, n_units, d_subframes.ptr))

    def dl_frontend(self, cfg, samples, unit_start, subfr_num, n_id_cell):
        """Host convenience.  samples: int8 [..., 2] interleaved I,Q, or a (i, q) pair of float32 arrays.
        Returns float32 [n_units, 2 + 2*N_ant, 16, 1200] = (symb_re, symb_im, ce_re[p].., ce_im[p]..)."""
        n = len(unit_start)
        if isinstance(samples, tuple):
            d_a, d_b = self.to_device(samples[0].astype(np.float32)), self.to_device(samples[1].astype(np.float32))
            cfg.sample_format = IQ_F32_PLANAR
        else:
            d_a, d_b = self.to_device(samples.astype(np.int8)), None
            cfg.sample_format = IQ_I8
        d_start = self.to_device(np.asarray(unit_start, np.uint64))
        d_sf = self.to_device(np.asarray(subfr_num, np.uint32))
        d_cell = self.to_device(np.asarray(n_id_cell, np.uint32))
        nf = self.subframe_floats(cfg.N_ant)
        d_out = self.alloc(n * nf * 4)
        d_out.zero()
        try:
            self.dl_frontend_dev(cfg, d_a, d_b, d_start, d_sf, d_cell, n, d_out)
            return d_out.download(np.float32).reshape(n, 2 + 2 * cfg.N_ant, 16, 1200)
        finally:
            for b in (d_a, d_b, d_start, d_sf, d_cell, d_out):
                if b is not None:
                    b.free()

    # ---- uplink ------------------------------------------------------------------------------
    def ul_subframe_floats(self):
        return self.L.mi_lte_ul_subframe_floats()

    def ul_frontend_dev(self, cfg, d_a, d_b, d_start, n_units, d_subframes):
        self._check(self.L.mi_lte_ul_frontend_batch(self.h, C.byref(cfg), d_a.ptr, d_b.ptr if d_b is not None else None,
                                                    d_start.ptr, n_units, d_subframes.ptr))

    def ul_frontend(self, cfg, samples, unit_start, keep=False):
        """Host convenience: int8 [..., 2] samples -> float32 [n_units, 2, 16, 1200] (symb_re, symb_im); keep=True
        also returns the device buffer (caller frees)."""
        n = len(unit_start)
        d_a = self.to_device(samples.astype(np.int8))
        cfg.sample_format = IQ_I8
        d_start = self.to_device(np.asarray(unit_start, np.uint64))
        d_out = self.alloc(n * self.ul_subframe_floats() * 4)
        d_out.zero()
        try:
            self.ul_frontend_dev(cfg, d_a, None, d_start, n, d_out)
            host = d_out.download(np.float32).reshape(n, 2, 16, 1200)
            return (host, d_out) if keep else host
        finally:
            d_a.free()
            d_start.free()
            if not keep:
                d_out.free()

    def ul_subframe_decode(self, fft_size, n_rb_ul, i_samps, q_samps, subfr_num, cell, ulcfg, allocs, pucch=(), pucch_tables=None):
        """mi_lte_ul_subframe_decode_host: one uplink subframe (float32 sample arrays) in one call.  allocs: list of PdschAlloc; pucch: list of
        (format 0/1/2, N_1_p_pucch) with tables float32 [n, 352].  Returns (status int32 [n_alloc], list of bit arrays (None where the CRC failed),
        (pucch bits uint8 [n, 2], n_bits, rc))."""
        L = self.L
        if not getattr(L, "_ul_sf_bound", False):
            L.mi_lte_ul_subframe_decode_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(UlCfg), C.c_void_p,
                                                         C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                         C.c_void_p, C.c_void_p]
            L._ul_sf_bound = True
        n, npu = len(allocs), len(pucch)
        arr = (PdschAlloc * max(n, 1))(*allocs)
        i_s, q_s = np.ascontiguousarray(i_samps, np.float32), np.ascontiguousarray(q_samps, np.float32)
        out, nb, st = np.zeros((max(n, 1), 6144), np.uint8), np.zeros(max(n, 1), np.uint32), np.full(max(n, 1), -7, np.int32)
        pr = (PucchRes * max(npu, 1))(*[PucchRes(0, f, n1) for (f, n1) in pucch])
        tabs = np.ascontiguousarray(pucch_tables if pucch_tables is not None else np.zeros((1, 352)), np.float32)
        pb, pnb, prc = np.zeros((max(npu, 1), 2), np.uint8), np.zeros(max(npu, 1), np.uint32), np.zeros(max(npu, 1), np.uint32)
        rc = L.mi_lte_ul_subframe_decode_host(self.h, fft_size, n_rb_ul, i_s.ctypes.data, q_s.ctypes.data, subfr_num, cell, C.byref(ulcfg), C.cast(arr, C.c_void_p), n,
                                              out.ctypes.data, 6144, nb.ctypes.data, st.ctypes.data, C.cast(pr, C.c_void_p), tabs.ctypes.data, npu, pb.ctypes.data,
                                              pnb.ctypes.data, prc.ctypes.data)
        if rc < 0:
            self._check(rc)
        if rc != 0:
            raise MiLteError("mi_lte_ul_subframe_decode_host: invalid arguments (%d)" % rc)
        return st[:n], [out[a, :nb[a]].copy() if st[a] == 0 else None for a in range(n)], (pb[:npu], pnb[:npu], prc[:npu])

    def prach_plan(self, cfg, prach_cfg, roots_fft=None):
        return PrachPlan(self, cfg, prach_cfg, roots_fft)

    def pucch_decode_dev(self, n_rb_ul, d_subframes, res, tables):
        """res: list of (unit, format 0/1/2, N_1_p_pucch); tables float32 [n, 352] (mi_lte.h).  Returns (bits uint8 [n, 2], n_bits, rc)."""
        n = len(res)
        arr = (PucchRes * n)(*[PucchRes(*r) for r in res])
        bits, nb, rc = np.zeros((n, 2), np.uint8), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        self._check(self.L.mi_lte_pucch_decode_run(self.h, n_rb_ul, 1, d_subframes.ptr, arr, np.ascontiguousarray(tables, np.float32).reshape(-1), n,
                                                   bits.reshape(-1), nb, rc))
        return bits, nb, rc

    def coarse_timing_dev(self, cfg, d_a, d_b, n_slots, start=0):
        """CoarseTiming for samples resident in HBM (d_b None for int8 interleaved)."""
        out = CoarseTiming()
        self._check(self.L.mi_lte_coarse_timing_run(self.h, C.byref(cfg), d_a.ptr, d_b.ptr if d_b is not None else None, start, n_slots, C.byref(out)))
        return out

    def find_pss_dev(self, cfg, d_a, d_b, symb_starts, start=0):
        """(symb_starts[7] rewritten, N_id_2, pss_symb, pss_thresh, freq_offset)"""
        ss = np.ascontiguousarray(symb_starts, np.uint32).copy()
        n2, ps, th, fo = C.c_uint32(), C.c_uint32(), C.c_float(), C.c_float()
        self._check(self.L.mi_lte_find_pss_run(self.h, C.byref(cfg), d_a.ptr, d_b.ptr if d_b is not None else None, start, ss, C.byref(n2), C.byref(ps),
                                               C.byref(th), C.byref(fo)))
        return ss, n2.value, ps.value, th.value, fo.value

    def find_sss_dev(self, cfg, d_a, d_b, n_id_2, symb_starts, pss_thresh, start=0):
        """(found, N_id_1, frame_start_idx, symb_starts[7])"""
        ss = np.ascontiguousarray(symb_starts, np.uint32).copy()
        n1, fs, found = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self.L.mi_lte_find_sss_run(self.h, C.byref(cfg), d_a.ptr, d_b.ptr if d_b is not None else None, start, n_id_2, ss, pss_thresh,
                                               C.byref(n1), C.byref(fs), C.byref(found)))
        return bool(found.value), n1.value, fs.value, ss

    def pbch_decode_dev(self, cfg, d_subframes, d_cell, n_units):
        """(N_ant[n] (0 = not decoded), offset[n], mib[n] = the 24 BCH bits, first bit in bit 23); cfg.N_ant must be 4."""
        out = [np.zeros(n_units, np.uint32) for _ in range(3)]
        self._check(self.L.mi_lte_pbch_decode_run(self.h, C.byref(cfg), d_subframes.ptr, d_cell.ptr, n_units, out[0], out[1], out[2]))
        return out

    def pdcch_plan(self, cfg, cells, phich_res=1.0, per_port_estimates=False):
        return PdcchPlan(self, cfg, cells, phich_res, 0, per_port_estimates)

    def pusch_plan(self, cfg, ulcfg, unit_subfr_num, unit_n_id_cell, allocs):
        return PuschPlan(self, cfg, ulcfg, unit_subfr_num, unit_n_id_cell, allocs)

    # ---- PDSCH ------------------------------------------------------------------------------
    def pdsch_plan(self, cfg, n_pdcch_symbs, allocs):
        return PdschPlan(self, cfg, n_pdcch_symbs, allocs)

    def pdsch_plan_dynamic(self, cfg, max_alloc, max_soft_bytes):
        return PdschPlan(self, cfg, 0, None, max_alloc, max_soft_bytes)

    def iq_to_planar(self, d_iq, n_samples, d_i, d_q):
        self._check(self.L.mi_lte_iq_i8_to_planar(self.h, d_iq.ptr, n_samples, d_i.ptr, d_q.ptr))

    def freq_shift(self, d_i, d_q, first_index, n_samples, freq_offset, fs):
        """LTE_fdd_dl_fs_samp_buf::freq_shift on planar float samples in HBM, in place."""
        self._check(self.L.mi_lte_freq_shift_run(self.h, d_i.ptr, d_q.ptr, first_index, n_samples, float(freq_offset), int(fs)))

    # ---- turbo -------------------------------------------------------------------------------
    def turbo_decode_dev(self, d_soft, soft_type, K, n_cb, d_out, mode=TURBO_REF, n_iter=8, qpp_spec=False):
        """Device-pointer form: everything already resident in HBM (what bench.py times)."""
        self._check(self.L.mi_lte_turbo_decode_batch(self.h, d_soft.ptr, soft_type, K, n_cb, mode, n_iter,
                                                     1 if qpp_spec else 0, d_out.ptr))

    def turbo_decode(self, soft, K, mode=TURBO_REF, n_iter=8, qpp_spec=False):
        """Host convenience: soft is [n_cb, 3*(K+4)] (float32 / int8 / int16), interleaved d[i*3+x]
        exactly as the reference's turbo_decode takes it; returns [n_cb, K] uint8 hard bits."""
        soft = np.ascontiguousarray(soft)
        assert soft.ndim == 2 and soft.shape[1] == 3 * (K + 4), soft.shape
        n_cb = soft.shape[0]
        d_in = self.to_device(soft)
        d_out = self.alloc(n_cb * K)
        try:
            self.turbo_decode_dev(d_in, _SOFT_OF_DTYPE[soft.dtype], K, n_cb, d_out, mode, n_iter, qpp_spec)
            return d_out.download(np.uint8).reshape(n_cb, K)
        finally:
            d_in.free()
            d_out.free()

    def device_copy_rate(self, nbytes=1 << 30, reps=10):
        """GB/s (read + written) of a 16-bytes-per-lane copy kernel on this device: the measured streaming rate."""
        out = C.c_double()
        self._check(self.L.mi_lte_device_copy_rate(self.h, C.c_size_t(nbytes), reps, C.byref(out)))
        return out.value

    def device_copy_rates(self):
        """The last device_copy_rate call's three kernel shapes: (one access per thread, the same non-temporal, grid-stride loop), GB/s."""
        out = (C.c_double * 3)()
        self._check(self.L.mi_lte_device_copy_rates(self.h, out))
        return tuple(round(v, 1) for v in out)

    def turbo_early_exit_iterations(self):
        """Iterations each tile pair (128 code blocks) of the last TURBO_BCJR_EARLY decode ran: uint32 [n_pairs]."""
        n, ni = C.c_uint32(), C.c_uint32()
        out = np.zeros(1 << 16, np.uint32)
        self._check(self.L.mi_lte_turbo_early_exit_iterations(self.h, out.ctypes.data, len(out), C.byref(n), C.byref(ni)))
        return out[:n.value].copy()

    def set_turbo_merged(self, on=True):
        """Several code-block sizes in one decode: one launch set over all of them (default) or, off, size by size (mi_lte_set_turbo_merged)."""
        self.L.mi_lte_set_turbo_merged.argtypes = [C.c_void_p, C.c_uint32]
        self._check(self.L.mi_lte_set_turbo_merged(self.h, 1 if on else 0))

    def set_turbo_small_batch(self, n_cb_max):
        """Code blocks per decode up to which the REF decoder's state-parallel trellis kernel runs (0: always the lock-step one)."""
        self._check(self.L.mi_lte_set_turbo_small_batch(self.h, int(n_cb_max)))

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_lte_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
