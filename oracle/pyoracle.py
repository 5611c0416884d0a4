"""ctypes bindings for the two CPU oracles.  TEST INFRASTRUCTURE ONLY.

* ``port()``  -> oracle/liblte_oracle.so : our plain-C restatement (oracle/lte_oracle.c)
* ``ref()``   -> oracle/_ref/libref_oracle.so : the reference's own liblte_phy.cc compiled in place
                 (oracle/ref/Makefile); ``None`` when it has not been built / shipped.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (openlte_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = None
_REF = None
_REF_TRIED = False
_REF_BIG = False

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u32 = C.c_uint32


class LoCfg(C.Structure):
    _fields_ = [(n, u32) for n in ("N_samps_per_symb", "N_samps_cp_l_0", "N_samps_cp_l_else", "N_samps_per_slot",
                                   "N_samps_per_subfr", "N_rb_dl", "N_sc_rb_dl", "FFT_size", "FFT_pad_size")]


class LoAlloc(C.Structure):
    _fields_ = [(n, u32) for n in ("mod_type", "tbs", "rv_idx", "N_prb", "tx_mode", "rnti", "pre_coder_type",
                                   "N_codewords")] + [("prb", u32 * 110)]


class LoSubframe(C.Structure):
    _fields_ = [("rx_symb_re", C.c_float * (16 * 1200)), ("rx_symb_im", C.c_float * (16 * 1200)),
                ("rx_ce_re", C.c_float * (4 * 16 * 1200)), ("rx_ce_im", C.c_float * (4 * 16 * 1200)), ("num", u32)]

    def arr(self, name):
        a = np.ctypeslib.as_array(getattr(self, name))
        return a.reshape((16, 1200)) if "symb" in name else a.reshape((4, 16, 1200))


class RefDlCase(C.Structure):
    """ref_dl_case (oracle/ref/ref_fuzz.cc)"""
    _fields_ = [(n, u32) for n in ("fs_enum", "N_rb_dl", "N_ant", "N_id_cell", "subfr_num", "N_pdcch_symbs", "mod_type", "tbs", "rv_idx",
                                   "tx_mode", "rnti", "N_prb")] + \
               [("prb", (C.c_uint8 * 112) * 2), ("snr_db", C.c_float), ("peak", C.c_float), ("gain_re", C.c_float * 4),
                ("gain_im", C.c_float * 4), ("delay", u32), ("seed", u32), ("rc_tx", C.c_int32), ("rc_fe", C.c_int32), ("rc", C.c_int32),
                ("N_out", u32), ("N_soft", u32)]


class RefUlAllocCase(C.Structure):
    """ref_ul_alloc_case"""
    _fields_ = [(n, u32) for n in ("unit", "mod_type", "tbs", "rnti", "N_prb")] + [("prb", C.c_uint8 * 112), ("rc", C.c_int32), ("N_out", u32),
                                                                                     ("N_soft", u32)]


class RefUlUnitCase(C.Structure):
    """ref_ul_unit_case"""
    _fields_ = [(n, u32) for n in ("fs_enum", "N_rb_ul", "N_id_cell", "subfr_num", "group_assignment_pusch", "group_hopping_enabled",
                                   "sequence_hopping_enabled", "cyclic_shift", "cyclic_shift_dci")] + [("rc_fe", C.c_int32)]


def make_alloc(mod_type, tbs, prbs, rnti, rv_idx=0, tx_mode=1, pre_coder_type=0, n_codewords=1):
    a = LoAlloc()
    a.mod_type, a.tbs, a.rv_idx, a.N_prb, a.tx_mode, a.rnti = mod_type, tbs, rv_idx, len(prbs), tx_mode, rnti
    a.pre_coder_type, a.N_codewords = pre_coder_type, n_codewords
    for i, p in enumerate(prbs):
        a.prb[i] = p
    return a


def build_port():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liblte_oracle.so"])


def build_ref():
    """Compile the reference in place (only possible where /root/reference exists)."""
    if os.path.isdir("/root/reference/liblte/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref")])
        return True
    return False


def port():
    global _PORT
    if _PORT is not None:
        return _PORT
    path = os.path.join(_HERE, "liblte_oracle.so")
    if not os.path.exists(path):
        build_port()
    L = C.CDLL(path)
    L.lo_cfg_init.argtypes = [C.POINTER(LoCfg), u32, u32]
    L.lo_qpp_map_ref.argtypes = [u32, u16p]
    L.lo_qpp_map_spec.argtypes = [u32, u16p]
    L.lo_prs_c.argtypes = [u32, u32, u8p]
    L.lo_generate_crs.argtypes = [u32, u32, u32, f32p, f32p]
    L.lo_crc24a.argtypes = [u8p, u32, u8p]
    L.lo_viterbi_siso.argtypes = [i8p, u32, i8p]
    L.lo_fb_soft.argtypes = [i8p, u32, i8p]
    L.lo_turbo_decode_ref.argtypes = [f32p, u32, u8p]
    L.lo_turbo_decode_ref_taps.argtypes = [f32p, u32, u8p, i8p]
    L.lo_turbo_encode.argtypes = [u8p, u32, u8p]
    L.lo_rate_unmatch_turbo.argtypes = [f32p, u32, u32, u32, u32, u32, u32, u32, u32, f32p]
    L.lo_rate_unmatch_turbo.restype = u32
    L.lo_rate_match_turbo.argtypes = [u8p, u32, u32, u32, u32, u32, u32, u32, u32, u8p]
    L.lo_modulation_demapper.argtypes = [f32p, f32p, u32, u32, i8p]
    L.lo_modulation_demapper.restype = u32
    L.lo_pre_decoder_dl.argtypes = [f32p, f32p, f32p, f32p, u32, u32, u32, f32p, f32p]
    L.lo_pre_decoder_dl.restype = u32
    L.lo_segmentation_params.argtypes = [u32] + [C.POINTER(u32)] * 5
    L.lo_dlsch_channel_decode.argtypes = [f32p, u32, u32, u32, u32, u32, u32, u8p, C.POINTER(u32), C.c_void_p]
    L.lo_samples_to_symbols_dl.argtypes = [C.POINTER(LoCfg), f32p, f32p, u32, u32, f32p, f32p]
    L.lo_get_dl_subframe_and_ce.argtypes = [C.POINTER(LoCfg), f32p, f32p, u32, u32, u32, u32, C.POINTER(LoSubframe)]
    L.lo_pdsch_channel_decode.argtypes = [C.POINTER(LoCfg), C.POINTER(LoSubframe), C.POINTER(LoAlloc), u32, u32, u32,
                                          u8p, C.POINTER(u32), C.c_void_p, C.POINTER(u32)]
    L.lo_time_turbo_decode_ref.argtypes = [f32p, u32, u32, u8p]
    L.lo_time_turbo_decode_ref.restype = C.c_double
    if hasattr(L, "lo_turbo_decode_bcjr"):
        L.lo_turbo_decode_bcjr.argtypes = [np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS"), u32, u32, C.c_int, u8p]
    if hasattr(L, "lo_turbo_decode_bcjr_block"):
        L.lo_turbo_decode_bcjr_block.argtypes = [np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS"), u32, u32, C.c_int, u8p]
        L.lo_bcjr_block_seg_len.argtypes = [u32]
        L.lo_bcjr_block_seg_len.restype = u32
    _PORT = L
    return L


def ref_big():
    """The reference compiled through the sed of oracle/ref/Makefile that only enlarges the PDSCH scratch literals (SURVEY 7.1
    `oracle_big`): allocations of more than 10 000 soft bits.  None if it has not been built / shipped."""
    global _REF_BIG
    if _REF_BIG is not False:
        return _REF_BIG
    path = os.path.join(_HERE, "_ref", "libref_oracle_big.so")
    if not os.path.exists(path):
        try:
            if not build_ref() or not os.path.exists(path):
                _REF_BIG = None
                return None
        except Exception:
            _REF_BIG = None
            return None
    _REF_BIG = _bind_ref(C.CDLL(path))
    return _REF_BIG


_REF_F32 = False


def ref_f32fft():
    """The compiled reference linked against the single-precision FFT stand-in (oracle/ref/fftw_shim_f32.c): what bench.py's
    cpu_baseline legs TIME.  No parity test goes through it.  None if it has not been built / shipped."""
    global _REF_F32
    if _REF_F32 is not False:
        return _REF_F32
    path = os.path.join(_HERE, "_ref", "libref_oracle_f32fft.so")
    if not os.path.exists(path):
        try:
            if not build_ref() or not os.path.exists(path):
                _REF_F32 = None
                return None
        except Exception:
            _REF_F32 = None
            return None
    _REF_F32 = _bind_ref(C.CDLL(path))
    return _REF_F32


def ref():
    """The compiled reference, or None if oracle/_ref has not been built/shipped."""
    global _REF, _REF_TRIED
    if _REF_TRIED:
        return _REF
    _REF_TRIED = True
    path = os.path.join(_HERE, "_ref", "libref_oracle.so")
    if not os.path.exists(path):
        try:
            if not build_ref():
                return None
        except Exception:
            return None
    _REF = _bind_ref(C.CDLL(path))
    return _REF


def _bind_ref(L):
    vp = C.c_void_p
    L.ref_phy_new.restype = vp
    L.ref_phy_new.argtypes = [C.c_int] * 4
    L.ref_phy_new_phich.restype = vp
    L.ref_phy_new_phich.argtypes = [C.c_int] * 4 + [C.c_float]
    L.ref_phy_free.argtypes = [vp]
    L.ref_sizeof_phy_struct.restype = C.c_size_t
    L.ref_sizeof_subframe_struct.restype = C.c_size_t
    L.ref_turbo_encode.argtypes = [vp, u8p, u32, u8p]
    L.ref_turbo_encode.restype = u32
    L.ref_turbo_decode.argtypes = [vp, f32p, u32, u8p]
    L.ref_turbo_decode_batch.argtypes = [vp, f32p, u32, u32, u8p, u32]
    L.ref_turbo_decode_batch.restype = C.c_double
    L.ref_viterbi_siso.argtypes = [vp, i8p, u32, i8p]
    L.ref_viterbi_siso.restype = u32
    L.ref_conv_encode_soft_g03.argtypes = [vp, i8p, u32, i8p]
    L.ref_conv_encode_soft_g03.restype = u32
    L.ref_rate_match_turbo.argtypes = [vp, u8p, u32, u32, u32, u32, u32, u32, u32, u32, u8p]
    L.ref_rate_unmatch_turbo.argtypes = [vp, f32p, u32, u32, u32, u32, u32, u32, u32, u32, f32p]
    L.ref_rate_unmatch_turbo.restype = u32
    L.ref_modulation_demapper.argtypes = [f32p, f32p, u32, u32, i8p]
    L.ref_modulation_demapper.restype = u32
    L.ref_modulation_mapper.argtypes = [u8p, u32, u32, f32p, f32p]
    L.ref_modulation_mapper.restype = u32
    L.ref_generate_prs_c.argtypes = [u32, u32, u32p]
    L.ref_generate_crs.argtypes = [u32, u32, u32, f32p, f32p]
    L.ref_calc_crc24a.argtypes = [u8p, u32, u8p]
    L.ref_pre_decoder_dl.argtypes = [f32p, f32p, f32p, f32p, u32, u32, u32, f32p, f32p, C.POINTER(u32)]
    L.ref_dlsch_channel_decode.argtypes = [vp, f32p, u32, u32, u32, u32, u32, u32, u8p, C.POINTER(u32)]
    L.ref_dlsch_channel_encode.argtypes = [vp, u8p, u32, u32, u32, u32, u32, u8p]
    L.ref_dlsch_channel_encode.restype = u32
    L.ref_subframe_new.restype = vp
    L.ref_subframe_free.argtypes = [vp]
    L.ref_subframe_clear_tx.argtypes = [vp, u32]
    L.ref_subframe_ptr.argtypes = [vp, C.c_int]
    L.ref_subframe_ptr.restype = C.POINTER(C.c_float)
    L.ref_subframe_set_num.argtypes = [vp, u32]
    L.ref_map_crs.argtypes = [vp, vp, u32, u32]
    L.ref_pdsch_channel_encode.argtypes = [vp, vp, C.POINTER(LoAlloc), u32, u8p, u32, u32, u32, u32]
    L.ref_create_dl_subframe.argtypes = [vp, vp, u32, f32p, f32p]
    L.ref_get_dl_subframe_and_ce.argtypes = [vp, f32p, f32p, u32, u32, u32, u32, vp]
    L.ref_pdsch_channel_decode.argtypes = [vp, vp, C.POINTER(LoAlloc), u32, u32, u32, u8p, C.POINTER(u32)]
    L.ref_pdsch_soft_bits_ptr.argtypes = [vp]
    L.ref_pdsch_soft_bits_ptr.restype = C.POINTER(C.c_int8)
    L.ref_pdsch_descramb_bits_ptr.argtypes = [vp]
    L.ref_pdsch_descramb_bits_ptr.restype = C.POINTER(C.c_float)
    L.ref_dlsch_rx_d_bits_ptr.argtypes = [vp]
    L.ref_dlsch_rx_d_bits_ptr.restype = C.POINTER(C.c_float)
    L.ref_dlsch_c_bits_ptr.argtypes = [vp]
    L.ref_dlsch_c_bits_ptr.restype = C.POINTER(C.c_uint8)
    L.ref_samples_to_symbols_dl.argtypes = [vp, f32p, f32p, u32, u32, f32p, f32p]
    L.ref_time_get_dl_subframe_and_ce.argtypes = [vp, f32p, f32p, u32, u32, u32, u32, vp, u32]
    L.ref_time_get_dl_subframe_and_ce.restype = C.c_double
    if hasattr(L, "ref_time_dl_chain"):  # absent from a libref_oracle.so built before round 2
        L.ref_time_dl_chain.argtypes = [vp, f32p, f32p, u32, u32, vp, C.POINTER(LoAlloc), u32, u32, u32, C.POINTER(u32)]
        L.ref_time_dl_chain.restype = C.c_double
    # uplink
    L.ref_ul_init.argtypes = [vp, u32, u32, u32, u32, u32, u32]
    L.ref_get_pusch_dmrs.argtypes = [vp, u32, u32, f32p]
    L.ref_get_ul_subframe.argtypes = [vp, f32p, f32p, vp]
    L.ref_pusch_channel_decode.argtypes = [vp, vp, C.POINTER(LoAlloc), u32, u32, u8p, C.POINTER(u32)]
    if hasattr(L, "ref_pusch_channel_decode_slots"):
        L.ref_pusch_channel_decode_slots.argtypes = [vp, vp, C.POINTER(LoAlloc), C.POINTER(u32), u32, u32, u8p, C.POINTER(u32)]
    L.ref_pusch_soft_bits_ptr.argtypes = [vp]
    L.ref_pusch_soft_bits_ptr.restype = C.POINTER(C.c_int8)
    L.ref_ulsch_rx_g_bits_ptr.argtypes = [vp]
    L.ref_ulsch_rx_g_bits_ptr.restype = C.POINTER(C.c_float)
    L.ref_pusch_d_re_ptr.argtypes = [vp]
    L.ref_pusch_d_re_ptr.restype = C.POINTER(C.c_float)
    L.ref_pusch_d_im_ptr.argtypes = [vp]
    L.ref_pusch_d_im_ptr.restype = C.POINTER(C.c_float)
    L.ref_ul_init_prach.argtypes = [vp, u32, u32, u32, u32, u32]
    L.ref_detect_prach.argtypes = [vp, f32p, f32p, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.ref_prach_n_roots.argtypes = [vp]
    L.ref_prach_n_roots.restype = u32
    L.ref_get_prach_root_fft.argtypes = [vp, u32, f32p, f32p]
    if hasattr(L, "ref_get_prach_root_seq"):  # (added in round 4: an older prebuilt harness does not have it)
        L.ref_get_prach_root_seq.argtypes = [vp, u32, f32p, f32p]
    L.ref_time_pusch.argtypes = [vp, f32p, f32p, vp, C.POINTER(LoAlloc), u32, u32, u32]
    L.ref_time_pusch.restype = C.c_double
    # control channels
    L.ref_pdcch_channel_encode.argtypes = [vp, vp, u32, C.POINTER(LoAlloc), u32p, u32, u32, u32, C.c_float]
    L.ref_pdcch_channel_decode.argtypes = [vp, vp, u32, u32, C.c_float, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(LoAlloc), u32p, u32p]
    L.ref_time_pdcch.argtypes = [vp, vp, u32, u32, C.c_float, u32]
    L.ref_time_pdcch.restype = C.c_double
    L.ref_dci_unpack.argtypes = [u32, u8p, u32, u32, u32, u32, C.POINTER(LoAlloc), C.POINTER(u32), u32p]
    L.ref_bch_channel_encode.argtypes = [vp, vp, u8p, u32, u32, u32]
    L.ref_bch_channel_decode.argtypes = [vp, vp, u32, C.POINTER(u32), u8p, C.POINTER(u32)]
    L.ref_time_bch.argtypes = [vp, vp, u32, u32]
    L.ref_time_bch.restype = C.c_double
    L.ref_find_coarse_timing.argtypes = [vp, f32p, f32p, u32, C.POINTER(u32), f32p, u32p]
    L.ref_find_pss.argtypes = [vp, f32p, f32p, u32p, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ref_find_sss.argtypes = [vp, f32p, f32p, u32, u32p, C.c_float, C.POINTER(u32), C.POINTER(u32)]
    L.ref_time_coarse_timing.argtypes = [vp, f32p, f32p, u32, u32]
    L.ref_time_coarse_timing.restype = C.c_double
    L.ref_ul_init_pucch.argtypes = [vp, u32, u32, u32, u32, u32]
    L.ref_pucch_decode.argtypes = [vp, vp, u32, u32, u32, u32, u8p, C.POINTER(u32)]
    L.ref_get_pucch_tables.argtypes = [vp, u32, u32, f32p]
    L.ref_get_n_rb_ul.argtypes = [vp]
    L.ref_get_n_rb_ul.restype = u32
    if hasattr(L, "ref_dl_cases_run"):  # absent from a libref_oracle.so built before round 3
        sz = C.c_size_t
        L.ref_dl_case_sizeof.restype = sz
        L.ref_ul_alloc_case_sizeof.restype = sz
        L.ref_ul_unit_case_sizeof.restype = sz
        L.ref_pdsch_soft_capacity.restype = sz
        assert L.ref_dl_case_sizeof() == C.sizeof(RefDlCase) and L.ref_ul_alloc_case_sizeof() == C.sizeof(RefUlAllocCase)
        assert L.ref_ul_unit_case_sizeof() == C.sizeof(RefUlUnitCase)
        L.ref_dl_cases_run.argtypes = [vp, u32, C.c_int, vp, sz, vp, sz, vp, sz, vp, sz, vp, C.c_int]
        L.ref_ul_cases_run.argtypes = [vp, u32, vp, u32, vp, sz, vp, vp, sz, vp, sz, C.c_int]
    return L


FS_ENUM = {128: 0, 256: 1, 512: 2, 1024: 3, 2048: 4}  # LIBLTE_PHY_FS_ENUM (liblte_phy.h:196-203)


def ref_subframe_view(L, sf, which, n_ant_dim=False):
    p = L.ref_subframe_ptr(sf, which)
    n = 4 * 16 * 1200 if n_ant_dim else 16 * 1200
    a = np.ctypeslib.as_array(p, shape=(n,))
    return a.reshape((4, 16, 1200)) if n_ant_dim else a.reshape((16, 1200))
