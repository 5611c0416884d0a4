"""openlte_amd -- MI355X-native LTE downlink receive chain behind the liblte_phy hot-path API.

The product is ``libmi_lte.so`` (hand-written HIP for gfx950 + a C-ABI, ``include/mi_lte.h``);
this package is only its ctypes binding plus small host-side helpers.  There is no CPU fallback:
importing works anywhere (so the C-ABI can be inspected), but creating a ``Context`` without the
built library or without a gfx950 GPU raises.
"""
from .lib import (Transmitter, TxAlloc, DlCfg, UlCfg, PrachCfg, CoarseTiming, PdcchDci, dci_unpack, pdcch_re_tables, ul_dmrs_pusch, PdschAlloc, make_alloc, tile_allocs, IQ_I8, IQ_F32_PLANAR, IQ_ALL_ROWS, CE_COMPACT, Context, DeviceBuffer, HostBuffer, DlPipeline, MiLteError, build_library, library_path, load_library,
                  SOFT_F32, SOFT_I8, SOFT_I16, TURBO_REF, TURBO_BCJR, TURBO_BCJR_BLOCK, TURBO_BCJR_EARLY)

__all__ = ["Transmitter", "TxAlloc", "DlCfg", "UlCfg", "PrachCfg", "CoarseTiming", "PdcchDci", "dci_unpack", "pdcch_re_tables", "ul_dmrs_pusch", "PdschAlloc", "make_alloc", "tile_allocs", "IQ_I8", "IQ_F32_PLANAR", "IQ_ALL_ROWS", "CE_COMPACT", "Context", "DeviceBuffer", "HostBuffer", "DlPipeline", "MiLteError", "build_library", "library_path", "load_library",
           "SOFT_F32", "SOFT_I8", "SOFT_I16", "TURBO_REF", "TURBO_BCJR", "TURBO_BCJR_BLOCK", "TURBO_BCJR_EARLY"]
