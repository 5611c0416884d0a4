// liblte_phy_shim.cc -- the binding a maintainer of the reference adds to run the DL receive hot
// path -- and, since round 1, the uplink PUSCH receive path -- on an MI355X.  It is compiled AGAINST THE REFERENCE'S OWN HEADER (liblte/hdr/liblte_phy.h from
// their tree; nothing from the reference is copied here) and defines, with the reference's exact
// C++ signatures, the public functions of the receive paths (and liblte_phy_cleanup, which owns the GPU context's lifetime):
//
//     liblte_phy_get_dl_subframe_and_ce   liblte_phy.h:1170-1177   (impl. liblte_phy.cc:5905-6200)
//     liblte_phy_pdsch_channel_decode     liblte_phy.h:906-913     (impl. liblte_phy.cc:3690-3853)
//     liblte_phy_rate_unmatch_turbo       liblte_phy.h:1311-1323   (impl. liblte_phy.cc:11246-11490)
//     liblte_phy_get_ul_subframe          liblte_phy.h:1190-1193   (impl. liblte_phy.cc:6209-6236)
//     liblte_phy_pusch_channel_decode     liblte_phy.h:722-728     (impl. liblte_phy.cc:2801-2935)
//     liblte_phy_detect_prach             liblte_phy.h:862-868     (impl. liblte_phy.cc:3299-3479)
//     liblte_phy_pdcch_channel_decode     liblte_phy.h:1012-1020   (impl. liblte_phy.cc:4519-5135)
//     liblte_phy_bch_channel_decode       liblte_phy.h:947-953     (impl. liblte_phy.cc:3968-4105)
//     liblte_phy_pucch_format_1_1a_1b_channel_decode   liblte_phy.h:775-782 (impl. liblte_phy.cc:2961-3146)
//     liblte_phy_dl_find_coarse_timing_and_freq_offset  liblte_phy.h:1134-1138 (impl. liblte_phy.cc:5697-5852)
//     liblte_phy_find_pss_and_fine_timing liblte_phy.h:1068-1075   (impl. liblte_phy.cc:5306-5510)
//     liblte_phy_find_sss                 liblte_phy.h:1106-1113   (impl. liblte_phy.cc:5578-5687)
//
// by forwarding to libmi_lte.so's C-ABI (include/mi_lte.h).  The reference's own definitions of
// these symbols are kept out of the link by compiling liblte_phy.cc with
//     -Dliblte_phy_get_dl_subframe_and_ce=liblte_phy_get_dl_subframe_and_ce_cpu   (etc.)
// so every other liblte_phy_* function (sync, PBCH, PDCCH, TX side, UL) stays on the reference's CPU
// code and keeps working unchanged -- see INTEGRATION.md and shim/Makefile.
//
// LIBLTE_PHY_STRUCT stays byte-identical (callers read its fields directly); the GPU
// context lives in a side table keyed by the struct pointer.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "liblte_phy.h" // the reference's header, from -I<reference>/liblte/hdr -I<reference>/cmn_hdr
#include "mi_lte.h"
#include "liblte_phy_ext.h"
#include <algorithm>

namespace {
// One GPU context per LIBLTE_PHY_STRUCT, created on first use and destroyed with the struct (liblte_phy_cleanup below).  The
// reference's contract is one call at a time per struct (all of its mutable state lives there); the context has the same
// contract -- one stream, one scratch, one staged subframe -- and the entry's mutex enforces it for callers that share a struct
// between threads anyway.
struct Entry {
    mi_lte_ctx *ctx = nullptr;
    std::mutex  mu;
    // what liblte_phy_ul_init was given (the reference turns it into tables inside the struct; here the library's generators are asked per
    // call and their answers kept): PUSCH reference-signal configuration, the cell the tables are FOR, DMRS per (subframe, N_prb)
    mi_lte_ul_cfg ul = {0, 0, 0, 0, 0};
    uint32_t      ul_cell = 0, n_cs_an = 0, delta_pucch_shift = 1;
    std::map<uint32_t, std::vector<float>> dmrs, pucch; // keys: subframe * 256 + N_prb / + N_1_p_pucch
    // the transmit side's scratch (host memory only; made by the first transmit call, own-lifecycle build)
    mi_lte_tx    *tx    = nullptr;
    mi_lte_tx_ul *tx_ul = nullptr;
    ~Entry() { if (tx) mi_lte_tx_destroy(tx); if (tx_ul) mi_lte_tx_ul_destroy(tx_ul); }
};
std::mutex                                            g_mu;
std::map<LIBLTE_PHY_STRUCT *, std::shared_ptr<Entry>> g_ctx;

// The entry is SHARED between the table and every call that is using it: liblte_phy_cleanup on another thread takes it out of the table and
// destroys the context under the entry's mutex, but the Entry (and its mutex) lives until the last caller lets go -- a late caller then
// finds ctx == nullptr and fails with the reference's error instead of touching freed state.
// Entries are made by liblte_phy_init (below) and by nothing else: a struct this table does not know -- one that liblte_phy_cleanup has
// already taken out, or one that never came from liblte_phy_init -- gets an entry without a context, i.e. the reference's error, and
// never a fresh GPU context keyed by a dead address that the next struct malloc places there would silently inherit.
std::shared_ptr<Entry> make_entry()
{
    auto        e  = std::make_shared<Entry>();
    const char *dv = getenv("MI_LTE_DEVICE");
    if (mi_lte_ctx_create(dv ? atoi(dv) : 0, &e->ctx) != MI_LTE_OK) e->ctx = nullptr; // no GPU: every call below fails loudly
    // a caller that never writes into a LIBLTE_PHY_SUBFRAME_STRUCT between the front end and the decodes (LTE_fdd_dl_fs_samp_buf.cc does not)
    // can say so: the decodes then take the copy in HBM on the struct's address alone instead of hashing its contents
    if (e->ctx && getenv("MI_LTE_SHIM_EXPLICIT_CACHE")) mi_lte_host_cache_set_mode(e->ctx, MI_LTE_HOST_CACHE_EXPLICIT);
    return e;
}
std::shared_ptr<Entry> entry_for(LIBLTE_PHY_STRUCT *phy)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto                        it = g_ctx.find(phy);
    if (it != g_ctx.end()) return it->second;
    return std::make_shared<Entry>(); // unknown struct: ctx == nullptr, the call fails
}
// the context of a struct, locked for the duration of the enclosing call (the test of ctx is made under the lock: cleanup nulls it there)
#define MI_LOCKED_CTX(phy, fail)                                                                                                   \
    std::shared_ptr<Entry> entry_ = entry_for(phy);                                                                                \
    std::lock_guard<std::mutex> call_lock_(entry_->mu);                                                                            \
    if (!entry_->ctx) return fail;                                                                                                 \
    mi_lte_ctx *c = entry_->ctx

// the entry of a struct locked for a call that needs no GPU (the transmit side): the struct must be one liblte_phy_init made
#define MI_LOCKED_ENTRY(phy, fail)                                                                                                 \
    std::shared_ptr<Entry> entry_ = entry_for(phy);                                                                                \
    std::lock_guard<std::mutex> call_lock_(entry_->mu);                                                                            \
    if (!entry_->tx && mi_lte_tx_create(&entry_->tx) != MI_LTE_OK) return fail;                                                    \
    mi_lte_tx *t = entry_->tx

void to_tx_alloc(const LIBLTE_PHY_ALLOCATION_STRUCT *a, mi_lte_tx_alloc *o)
{
    memset(o, 0, sizeof(*o));
    for (int i = 0; i < 2; i++) o->msg[i] = a->msg[i].msg, o->msg_bits[i] = a->msg[i].N_bits;
    o->pre_coder_type = (uint32_t)a->pre_coder_type, o->mod_type = (uint32_t)a->mod_type, o->chan_type = (uint32_t)a->chan_type;
    o->tbs = a->tbs, o->rv_idx = a->rv_idx, o->N_prb = a->N_prb, o->N_codewords = a->N_codewords, o->N_layers = a->N_layers, o->tx_mode = a->tx_mode;
    o->rnti = a->rnti, o->mcs = a->mcs, o->tpc = a->tpc, o->ndi = a->ndi, o->dl_alloc = a->dl_alloc;
    memcpy(o->prb, a->prb, sizeof(o->prb));
}

void to_mi_alloc(const LIBLTE_PHY_ALLOCATION_STRUCT *a, mi_lte_pdsch_alloc *o)
{
    memset(o, 0, sizeof(*o));
    o->mod_type = (uint32_t)a->mod_type;
    o->tbs      = a->tbs;
    o->rv_idx   = a->rv_idx;
    o->tx_mode  = a->tx_mode;
    o->rnti     = a->rnti;
    o->N_prb    = a->N_prb;
    for (uint32 s = 0; s < 2; s++)
        for (uint32 i = 0; i < a->N_prb && i < 110; i++) o->prb[s][i] = (uint8_t)a->prb[s][i];
}
} // namespace

// ---- lifetime of a LIBLTE_PHY_STRUCT.  Two builds of this file:
//
//   default                     : liblte_phy_init / liblte_phy_cleanup WRAP the reference's own (compiled under the names *_cpu, shim/Makefile):
//                                 the struct is the reference's in every field, so the entry points that are NOT replaced (TX side, UL init,
//                                 TBS helpers) keep working next to the replaced ones.
//   -DMI_LTE_SHIM_OWN_LIFECYCLE : liblte_phy_init / liblte_phy_cleanup / liblte_phy_update_n_rb_dl are DEFINED here and no object of the
//                                 reference's PHY is linked at all (shim/Makefile: scan_gpu_pure = scan_demo.cc + liblte_rrc + this file +
//                                 libmi_lte.so).  That is the receive side of LTE_fdd_dl_file_scan: its ten liblte_phy calls are the three
//                                 above and seven replaced entry points.  The struct is the header's (layout is ABI); the fields a caller or
//                                 a replaced entry point reads are filled as liblte_phy.cc:2210-2335 and :2592-2647 fill them
//                                 (shim/lifecycle_check.cc compares them with the reference's for every sampling rate x bandwidth), the
//                                 reference's private work areas are zero, its FFTW plans and the transmitter's PDCCH permutation /
//                                 CRS storage (:2292-2307) do not exist -- nothing on the receive path reads them.
namespace {
void register_struct(LIBLTE_PHY_STRUCT *phy)
{
    std::shared_ptr<Entry> e = make_entry(), stale;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto                        it = g_ctx.find(phy);
        if (it != g_ctx.end()) stale = it->second; // a struct freed behind the shim's back (never cleaned up) whose address came round again
        g_ctx[phy] = e;
    }
    if (stale) {
        std::lock_guard<std::mutex> call(stale->mu);
        if (stale->ctx) mi_lte_ctx_destroy(stale->ctx);
        stale->ctx = nullptr;
    }
}
void unregister_struct(LIBLTE_PHY_STRUCT *phy)
{
    std::shared_ptr<Entry> e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto                        it = g_ctx.find(phy);
        if (it != g_ctx.end()) { e = it->second; g_ctx.erase(it); }
    }
    if (!e) return;
    std::lock_guard<std::mutex> call(e->mu); // waits for a call that is still running on it
    if (e->ctx && getenv("MI_LTE_SHIM_STATS")) { // how often a decode call found the struct it was handed already on the device
        uint64_t reuse = 0, upload = 0;
        uint32_t plans = 0;
        mi_lte_host_cache_stats(e->ctx, &reuse, &upload, &plans);
        fprintf(stderr, "mi_lte shim: device subframe reused %llu times, uploaded %llu times; %u cached plans\n", (unsigned long long)reuse,
                (unsigned long long)upload, plans);
    }
    if (e->ctx) mi_lte_ctx_destroy(e->ctx);
    e->ctx = nullptr; // a caller that was blocked on the mutex, or still holds the entry, fails cleanly
}
} // namespace

#ifndef MI_LTE_SHIM_OWN_LIFECYCLE
// liblte_phy_init (liblte_phy.h:606-612, impl. liblte_phy.cc:2210-2335): the reference's own initialisation builds the struct -- tables, FFT
// plans, everything the entry points that are NOT replaced read -- and the struct it returns is registered here with a GPU context of its own.
LIBLTE_ERROR_ENUM liblte_phy_init_cpu(LIBLTE_PHY_STRUCT **phy_struct, LIBLTE_PHY_FS_ENUM fs, uint16 N_id_cell, uint8 N_ant, uint32 N_rb_dl, uint32 N_sc_rb_dl,
                                      float phich_res);
LIBLTE_ERROR_ENUM liblte_phy_init(LIBLTE_PHY_STRUCT **phy_struct, LIBLTE_PHY_FS_ENUM fs, uint16 N_id_cell, uint8 N_ant, uint32 N_rb_dl, uint32 N_sc_rb_dl,
                                  float phich_res)
{
    const LIBLTE_ERROR_ENUM err = liblte_phy_init_cpu(phy_struct, fs, N_id_cell, N_ant, N_rb_dl, N_sc_rb_dl, phich_res);
    if (err != LIBLTE_SUCCESS || !phy_struct || !*phy_struct) return err;
    register_struct(*phy_struct);
    return err;
}

// liblte_phy_cleanup (liblte_phy.h:636, impl. liblte_phy.cc:2524-2543): the struct's GPU context goes first -- stream, scratch, staging buffers
// and cached plans are released, and a later struct that malloc places at the same address starts with a fresh context -- then the
// reference's own cleanup (compiled under the name liblte_phy_cleanup_cpu, shim/Makefile) frees the struct.
LIBLTE_ERROR_ENUM liblte_phy_cleanup_cpu(LIBLTE_PHY_STRUCT *phy_struct);
LIBLTE_ERROR_ENUM liblte_phy_cleanup(LIBLTE_PHY_STRUCT *phy_struct)
{
    unregister_struct(phy_struct);
    return liblte_phy_cleanup_cpu(phy_struct);
}

// liblte_phy_ul_init (liblte_phy.h:613-625, impl. liblte_phy.cc:2337-2517): the reference's own (compiled under the name liblte_phy_ul_init_cpu)
// fills the struct for the entry points that are NOT replaced -- the transmit side reads its DMRS and PRACH tables -- and the configuration
// is kept here: the replaced receive side asks the library's own generators (ul_rs.cc, prach_sets.hpp), exactly as the build without the
// reference's PHY object does, and reads none of the struct's uplink tables.
LIBLTE_ERROR_ENUM liblte_phy_ul_init_cpu(LIBLTE_PHY_STRUCT *phy_struct, uint16 N_id_cell, uint32 prach_root_seq_idx, uint32 prach_preamble_format, uint32 prach_zczc,
                                         bool prach_hs_flag, uint8 group_assignment_pusch, bool group_hopping_enabled, bool sequence_hopping_enabled,
                                         uint8 cyclic_shift, uint8 cyclic_shift_dci, uint8 N_cs_an, uint8 delta_pucch_shift);
LIBLTE_ERROR_ENUM liblte_phy_ul_init(LIBLTE_PHY_STRUCT *phy_struct, uint16 N_id_cell, uint32 prach_root_seq_idx, uint32 prach_preamble_format, uint32 prach_zczc,
                                     bool prach_hs_flag, uint8 group_assignment_pusch, bool group_hopping_enabled, bool sequence_hopping_enabled, uint8 cyclic_shift,
                                     uint8 cyclic_shift_dci, uint8 N_cs_an, uint8 delta_pucch_shift)
{
    const LIBLTE_ERROR_ENUM err = liblte_phy_ul_init_cpu(phy_struct, N_id_cell, prach_root_seq_idx, prach_preamble_format, prach_zczc, prach_hs_flag,
                                                         group_assignment_pusch, group_hopping_enabled, sequence_hopping_enabled, cyclic_shift, cyclic_shift_dci,
                                                         N_cs_an, delta_pucch_shift);
    if (err != LIBLTE_SUCCESS || phy_struct == NULL) return err;
    std::shared_ptr<Entry>      e = entry_for(phy_struct);
    std::lock_guard<std::mutex> call(e->mu);
    e->ul      = mi_lte_ul_cfg{group_assignment_pusch, group_hopping_enabled ? 1u : 0u, sequence_hopping_enabled ? 1u : 0u, cyclic_shift, cyclic_shift_dci};
    e->ul_cell = N_id_cell;
    e->n_cs_an = N_cs_an;
    e->delta_pucch_shift = (uint32_t)delta_pucch_shift + 1; // (generate_dmrs_pucch is handed delta_pucch_shift + 1, liblte_phy.cc:2414)
    e->dmrs.clear();
    e->pucch.clear();
    return err;
}
#else
// liblte_phy_update_n_rb_dl (liblte_phy.h:620-621, impl. liblte_phy.cc:2592-2647): a bandwidth is accepted when its sub-carriers fit the
// sampling rate's transform (at 30.72 MHz the reference accepts anything); N_rb_ul follows N_rb_dl, the pad is the unused half band.
LIBLTE_ERROR_ENUM liblte_phy_update_n_rb_dl(LIBLTE_PHY_STRUCT *phy_struct, uint32 N_rb_dl)
{
    if (phy_struct == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    static const uint32 rb_of_bw[6] = {LIBLTE_PHY_N_RB_DL_1_4MHZ, LIBLTE_PHY_N_RB_DL_3MHZ, LIBLTE_PHY_N_RB_DL_5MHZ,
                                       LIBLTE_PHY_N_RB_DL_10MHZ, LIBLTE_PHY_N_RB_DL_15MHZ, LIBLTE_PHY_N_RB_DL_20MHZ};
    uint32 n_bw; // how many of the six bandwidths the sampling rate carries
    switch (phy_struct->N_samps_per_symb) {
    case LIBLTE_PHY_N_SAMPS_PER_SYMB_1_92MHZ:  n_bw = 1; break;
    case LIBLTE_PHY_N_SAMPS_PER_SYMB_3_84MHZ:  n_bw = 2; break;
    case LIBLTE_PHY_N_SAMPS_PER_SYMB_7_68MHZ:  n_bw = 3; break;
    case LIBLTE_PHY_N_SAMPS_PER_SYMB_15_36MHZ: n_bw = 4; break;
    case LIBLTE_PHY_N_SAMPS_PER_SYMB_30_72MHZ: n_bw = 6; break;
    default: return LIBLTE_ERROR_INVALID_INPUTS;
    }
    bool ok = n_bw == 6; // (30.72 MHz: no test in the reference, :2606-2609)
    for (uint32 b = 0; b < n_bw && !ok; b++) ok = rb_of_bw[b] == N_rb_dl;
    if (!ok) return LIBLTE_ERROR_INVALID_INPUTS;
    phy_struct->FFT_size     = phy_struct->N_samps_per_symb; // LIBLTE_PHY_FFT_SIZE_* = LIBLTE_PHY_N_SAMPS_PER_SYMB_* (liblte_phy.h:100-104, :142-170)
    phy_struct->N_rb_dl      = N_rb_dl;
    phy_struct->N_rb_ul      = N_rb_dl;
    phy_struct->FFT_pad_size = (phy_struct->FFT_size - N_rb_dl * phy_struct->N_sc_rb_dl) / 2;
    return LIBLTE_SUCCESS;
}

// liblte_phy_init (liblte_phy.h:606-612, impl. liblte_phy.cc:2210-2335), receive side: the sampling-rate geometry, the bandwidth (through
// liblte_phy_update_n_rb_dl, whose verdict the reference ignores here, too), the PHICH group count, and a GPU context of the struct's own.
LIBLTE_ERROR_ENUM liblte_phy_init(LIBLTE_PHY_STRUCT **phy_struct, LIBLTE_PHY_FS_ENUM fs, uint16 N_id_cell, uint8 N_ant, uint32 N_rb_dl, uint32 N_sc_rb_dl,
                                  float phich_res)
{
    if (phy_struct == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    LIBLTE_PHY_STRUCT *p = (LIBLTE_PHY_STRUCT *)calloc(1, sizeof(LIBLTE_PHY_STRUCT));
    *phy_struct = p;
    if (p == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    uint32 scale; // 30.72 MHz / fs
    switch (fs) {
    case LIBLTE_PHY_FS_30_72MHZ: scale = 1; break;
    case LIBLTE_PHY_FS_15_36MHZ: scale = 2; break;
    case LIBLTE_PHY_FS_7_68MHZ:  scale = 4; break;
    case LIBLTE_PHY_FS_3_84MHZ:  scale = 8; break;
    case LIBLTE_PHY_FS_1_92MHZ:  scale = 16; break;
    default: scale = 0; break; // (the reference leaves the fields unset)
    }
    if (scale) {
        p->fs                = 30720000 / scale;
        p->N_samps_per_symb  = LIBLTE_PHY_N_SAMPS_PER_SYMB_30_72MHZ / scale;
        p->N_samps_cp_l_0    = LIBLTE_PHY_N_SAMPS_CP_L_0_30_72MHZ / scale;
        p->N_samps_cp_l_else = LIBLTE_PHY_N_SAMPS_CP_L_ELSE_30_72MHZ / scale;
        p->N_samps_per_slot  = LIBLTE_PHY_N_SAMPS_PER_SLOT_30_72MHZ / scale;
        p->N_samps_per_subfr = LIBLTE_PHY_N_SAMPS_PER_SUBFR_30_72MHZ / scale;
        p->N_samps_per_frame = LIBLTE_PHY_N_SAMPS_PER_FRAME_30_72MHZ / scale;
    }
    p->N_sc_rb_dl = N_sc_rb_dl;
    p->N_sc_rb_ul = LIBLTE_PHY_N_SC_RB_UL;
    (void)liblte_phy_update_n_rb_dl(p, N_rb_dl);
    p->N_ant   = N_ant;
    p->ul_init = false;
    // PHICH groups (36.211 6.9): ceil(N_g * N_rb_dl / 8), twice that with the extended prefix
    const uint32 groups = (uint32)ceilf((float)phich_res * ((float)p->N_rb_dl / (float)8));
    const bool   normal = LIBLTE_PHY_N_SC_RB_DL_NORMAL_CP == p->N_sc_rb_dl;
    p->N_group_phich = normal ? groups : 2 * groups;
    p->N_sf_phich    = normal ? 4 : 2;
    if (LIBLTE_PHY_INIT_N_ID_CELL_UNKNOWN != N_id_cell) p->N_id_cell_crs = N_id_cell; // (stored with the reference's CRS cache, which the replaced front end does not read)
    register_struct(p);
    return LIBLTE_SUCCESS;
}

// liblte_phy_cleanup (liblte_phy.h:636, impl. liblte_phy.cc:2524-2543): the GPU context, then the struct.
LIBLTE_ERROR_ENUM liblte_phy_cleanup(LIBLTE_PHY_STRUCT *phy_struct)
{
    if (phy_struct == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    unregister_struct(phy_struct);
    free(phy_struct);
    return LIBLTE_SUCCESS;
}

// liblte_phy_ul_init (liblte_phy.h:613-625, impl. liblte_phy.cc:2337-2517), receive side of PUSCH and PRACH: the reference fills the struct with
// DMRS tables for every (subframe, N_prb), PUCCH sequence tables, the cell's PRACH root sequences and their spectra, and a dozen FFTW plans.
// Here the configuration is kept (struct fields where the reference has fields for it, the side table otherwise) and the library's own
// generators (ul_rs.cc, prach_sets.hpp) are asked when a decode needs them.
LIBLTE_ERROR_ENUM liblte_phy_ul_init(LIBLTE_PHY_STRUCT *phy_struct, uint16 N_id_cell, uint32 prach_root_seq_idx, uint32 prach_preamble_format, uint32 prach_zczc,
                                     bool prach_hs_flag, uint8 group_assignment_pusch, bool group_hopping_enabled, bool sequence_hopping_enabled, uint8 cyclic_shift,
                                     uint8 cyclic_shift_dci, uint8 N_cs_an, uint8 delta_pucch_shift)
{
    if (phy_struct == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    std::shared_ptr<Entry> e = entry_for(phy_struct);
    std::lock_guard<std::mutex> call(e->mu);
    e->ul      = mi_lte_ul_cfg{group_assignment_pusch, group_hopping_enabled ? 1u : 0u, sequence_hopping_enabled ? 1u : 0u, cyclic_shift, cyclic_shift_dci};
    e->ul_cell = N_id_cell;
    e->n_cs_an = N_cs_an;
    e->delta_pucch_shift = (uint32_t)delta_pucch_shift + 1; // (generate_dmrs_pucch is handed delta_pucch_shift + 1, liblte_phy.cc:2414)
    e->dmrs.clear();
    e->pucch.clear();
    phy_struct->prach_root_seq_idx    = prach_root_seq_idx; // (prach_preamble_seq_gen, liblte_phy.cc:7157-7172)
    phy_struct->prach_preamble_format = prach_preamble_format;
    phy_struct->prach_zczc            = prach_zczc;
    phy_struct->prach_hs_flag         = prach_hs_flag;
    phy_struct->prach_N_zc            = prach_preamble_format == 4 ? 139 : 839;
    mi_lte_prach_cfg pc = {prach_root_seq_idx, prach_preamble_format, prach_zczc, prach_hs_flag ? 1u : 0u, 0};
    uint32_t         roots[64], n_roots = 0;
    phy_struct->prach_N_x_u = mi_lte_prach_root_set(&pc, roots, &n_roots) == MI_LTE_OK ? n_roots : 0;
    const uint32 down = 30720000 / phy_struct->fs; // occasion geometry in samples of this rate (liblte_phy.cc:2430-2467)
    static const uint32 t_seq[5] = {24576, 24576, 2 * 24576, 2 * 24576, 4096}, t_cp[5] = {3168, 21024, 6240, 21024, 448};
    const uint32 f = prach_preamble_format > 4 ? 4 : prach_preamble_format;
    phy_struct->prach_T_fft      = (f == 4 ? 4096 : 24576) / down;
    phy_struct->prach_T_seq      = t_seq[f] / down;
    phy_struct->prach_T_cp       = t_cp[f] / down;
    phy_struct->prach_delta_f_RA = f == 4 ? 7500 : 1250;
    phy_struct->prach_phi        = f == 4 ? 2 : 7;
    phy_struct->ul_init          = true;
    return LIBLTE_SUCCESS;
}

// liblte_phy_ul_cleanup (liblte_phy.h:639, impl. liblte_phy.cc:2544-2582)
LIBLTE_ERROR_ENUM liblte_phy_ul_cleanup(LIBLTE_PHY_STRUCT *phy_struct)
{
    if (phy_struct == NULL || !phy_struct->ul_init) return LIBLTE_ERROR_INVALID_INPUTS;
    std::shared_ptr<Entry> e = entry_for(phy_struct);
    std::lock_guard<std::mutex> call(e->mu);
    e->dmrs.clear();
    e->pucch.clear();
    phy_struct->ul_init = false;
    return LIBLTE_SUCCESS;
}
#endif

LIBLTE_ERROR_ENUM liblte_phy_get_dl_subframe_and_ce(LIBLTE_PHY_STRUCT *phy_struct, float *i_samps, float *q_samps,
                                                    uint32 frame_start_idx, uint8 subfr_num, uint32 N_id_cell, uint8 N_ant,
                                                    LIBLTE_PHY_SUBFRAME_STRUCT *subframe)
{
    if (phy_struct == NULL || i_samps == NULL || q_samps == NULL || !(N_ant == 1 || N_ant == 2 || N_ant == 4) || subframe == NULL)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    subframe->num = subfr_num;
    int rc = mi_lte_get_dl_subframe_and_ce_host(c, phy_struct->N_samps_per_symb, phy_struct->N_rb_dl, i_samps, q_samps, frame_start_idx,
                                                subfr_num, N_id_cell, N_ant, &subframe->rx_symb_re[0][0], &subframe->rx_symb_im[0][0],
                                                &subframe->rx_ce_re[0][0][0], &subframe->rx_ce_im[0][0][0]);
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
}

LIBLTE_ERROR_ENUM liblte_phy_pdsch_channel_decode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe,
                                                  LIBLTE_PHY_ALLOCATION_STRUCT *alloc, uint32 N_pdcch_symbs, uint32 N_id_cell,
                                                  uint8 N_ant, uint8 *out_bits, uint32 *N_out_bits)
{
    if (phy_struct == NULL || subframe == NULL || alloc == NULL || N_id_cell > 503 || out_bits == NULL || N_out_bits == NULL)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    mi_lte_pdsch_alloc a;
    to_mi_alloc(alloc, &a);
    int rc = mi_lte_pdsch_channel_decode_host(c, phy_struct->N_rb_dl, &subframe->rx_symb_re[0][0], &subframe->rx_symb_im[0][0],
                                              &subframe->rx_ce_re[0][0][0], &subframe->rx_ce_im[0][0][0], subframe->num, &a, N_pdcch_symbs,
                                              N_id_cell, N_ant, out_bits, N_out_bits);
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_DECODE_FAIL;
}

void liblte_phy_rate_unmatch_turbo(LIBLTE_PHY_STRUCT *phy_struct, float *e_bits, uint32 N_e_bits, uint8 *dummy_bits,
                                   uint32 N_dummy_bits, uint32 N_codeblocks, uint32 tx_mode, uint32 N_soft, uint32 M_dl_harq,
                                   LIBLTE_PHY_CHAN_TYPE_ENUM chan_type, uint32 rv_idx, float *d_bits, uint32 *N_d_bits)
{
    (void)dummy_bits; // only its length matters: a uint8 can never equal RX_NULL_BIT
    if (phy_struct == NULL || e_bits == NULL || d_bits == NULL || N_d_bits == NULL) return;
    MI_LOCKED_CTX(phy_struct, (void)0);
    uint32_t n = 0;
    if (0 == mi_lte_rate_unmatch_turbo_host(c, e_bits, N_e_bits, N_dummy_bits, N_codeblocks, tx_mode, N_soft, M_dl_harq,
                                                 (uint32_t)chan_type, rv_idx, d_bits, &n))
        *N_d_bits = n;
}

// ---- uplink (LTE_fdd_enodeb's radio thread: LTE_fdd_enb_phy.cc:832-917)

LIBLTE_ERROR_ENUM liblte_phy_get_ul_subframe(LIBLTE_PHY_STRUCT *phy_struct, float *i_samps, float *q_samps,
                                             LIBLTE_PHY_SUBFRAME_STRUCT *subframe)
{
    if (phy_struct == NULL || i_samps == NULL || q_samps == NULL || subframe == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    int rc = mi_lte_get_ul_subframe_host(c, phy_struct->N_samps_per_symb, phy_struct->N_rb_ul, i_samps, q_samps,
                                         &subframe->rx_symb_re[0][0], &subframe->rx_symb_im[0][0]);
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
}

LIBLTE_ERROR_ENUM liblte_phy_pusch_channel_decode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe,
                                                  LIBLTE_PHY_ALLOCATION_STRUCT *alloc, uint32 N_id_cell, uint8 N_ant, uint8 *out_bits,
                                                  uint32 *N_out_bits)
{
    if (phy_struct == NULL || subframe == NULL || alloc == NULL || out_bits == NULL || N_out_bits == NULL || !phy_struct->ul_init ||
        alloc->N_prb == 0 || alloc->N_prb >= LIBLTE_PHY_N_RB_UL_MAX || subframe->num > 9)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    mi_lte_pdsch_alloc a;
    to_mi_alloc(alloc, &a);
    const uint32 sf = subframe->num, np = alloc->N_prb;
    // the reference signals come from the library's own generator (mi_lte_ul_dmrs_pusch restates generate_dmrs_pusch, liblte_phy.cc:6745-6990,
    // value for value: tests/test_uplink_cpu.py), asked once per (subframe, N_prb) with the configuration and the cell liblte_phy_ul_init was
    // given -- not from the tables the reference's liblte_phy_ul_init leaves in the struct (pusch_dmrs_*), in either build
    std::vector<float> &tab = entry_->dmrs[sf * 256u + np];
    if (tab.empty()) {
        tab.resize((size_t)4 * 12 * np);
        if (mi_lte_ul_dmrs_pusch(&entry_->ul, entry_->ul_cell, sf, np, &tab[0], &tab[12 * np], &tab[24 * np], &tab[36 * np]) != MI_LTE_OK) {
            tab.clear();
            return LIBLTE_ERROR_INVALID_INPUTS;
        }
    }
    const float *d0r = &tab[0], *d0i = &tab[12 * np], *d1r = &tab[24 * np], *d1i = &tab[36 * np];
    int rc = mi_lte_pusch_channel_decode_host(c, phy_struct->N_rb_ul, &subframe->rx_symb_re[0][0], &subframe->rx_symb_im[0][0], sf, &a,
                                              N_id_cell, N_ant, d0r, d0i, d1r, d1i, out_bits, N_out_bits);
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS; // the reference's own failure code on this path (:2809, :2929)
}

// Not a reference symbol (shim/liblte_phy_ext.h): the receive half of a TTI in one call -- mi_lte_ul_subframe_decode_host
LIBLTE_ERROR_ENUM liblte_phy_ul_subframe_decode(LIBLTE_PHY_STRUCT *phy_struct, float *i_samps, float *q_samps, uint8 subfr_num, uint32 N_id_cell,
                                                LIBLTE_PHY_ALLOCATION_STRUCT *allocs, uint32 N_allocs, uint8 *out_bits, uint32 *N_out_bits,
                                                LIBLTE_ERROR_ENUM *status, LIBLTE_PHY_PUCCH_FORMAT_ENUM *pucch_format, uint32 *N_1_p_pucch,
                                                uint32 N_pucch, uint8 *pucch_bits, uint32 *N_pucch_bits, LIBLTE_ERROR_ENUM *pucch_status)
{
    if (phy_struct == NULL || i_samps == NULL || q_samps == NULL || !phy_struct->ul_init || subfr_num > 9 || N_allocs > LIBLTE_PHY_UL_SUBFRAME_MAX_ALLOC ||
        N_pucch > 32 || (N_allocs && (allocs == NULL || out_bits == NULL || N_out_bits == NULL || status == NULL)) ||
        (N_pucch && (pucch_format == NULL || N_1_p_pucch == NULL || pucch_bits == NULL || N_pucch_bits == NULL || pucch_status == NULL)))
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    std::vector<mi_lte_pdsch_alloc> al(N_allocs ? N_allocs : 1);
    for (uint32 k = 0; k < N_allocs; k++) {
        to_mi_alloc(&allocs[k], &al[k]);
        // (what liblte_phy_pusch_channel_decode itself refuses: reported per allocation below)
        if (allocs[k].N_prb == 0 || allocs[k].N_prb >= LIBLTE_PHY_N_RB_UL_MAX) al[k].N_prb = 0;
    }
    std::vector<mi_lte_pucch_res> pr(N_pucch ? N_pucch : 1);
    std::vector<float>            tabs((size_t)(N_pucch ? N_pucch : 1) * MI_LTE_PUCCH_TAB_FLOATS);
    for (uint32 r = 0; r < N_pucch; r++) {
        if (!(pucch_format[r] == LIBLTE_PHY_PUCCH_FORMAT_1 || pucch_format[r] == LIBLTE_PHY_PUCCH_FORMAT_1A || pucch_format[r] == LIBLTE_PHY_PUCCH_FORMAT_1B) ||
            N_1_p_pucch[r] >= LIBLTE_PHY_N_RB_UL_MAX / 2)
            return LIBLTE_ERROR_INVALID_INPUTS;
        pr[r] = mi_lte_pucch_res{0, (uint32_t)pucch_format[r], N_1_p_pucch[r]};
        std::vector<float> &tab = entry_->pucch[subfr_num * 256u + N_1_p_pucch[r]]; // (as in liblte_phy_pucch_format_1_1a_1b_channel_decode below)
        if (tab.empty()) {
            tab.resize(MI_LTE_PUCCH_TAB_FLOATS);
            if (mi_lte_ul_pucch_tables(&entry_->ul, entry_->ul_cell, subfr_num, N_1_p_pucch[r], entry_->n_cs_an, entry_->delta_pucch_shift, phy_struct->N_ant, &tab[0]) != MI_LTE_OK) {
                tab.clear();
                return LIBLTE_ERROR_INVALID_INPUTS;
            }
        }
        std::copy(tab.begin(), tab.end(), tabs.begin() + (size_t)r * MI_LTE_PUCCH_TAB_FLOATS);
    }
    static thread_local std::vector<uint8_t> bits;
    bits.resize((size_t)LIBLTE_PHY_UL_SUBFRAME_MAX_ALLOC * 6144);
    std::vector<uint32_t> nb(N_allocs ? N_allocs : 1), pnb(N_pucch ? N_pucch : 1), prc(N_pucch ? N_pucch : 1);
    std::vector<int32_t>  st(N_allocs ? N_allocs : 1);
    std::vector<uint8_t>  pb(2 * (size_t)(N_pucch ? N_pucch : 1));
    int rc = mi_lte_ul_subframe_decode_host(c, phy_struct->N_samps_per_symb, phy_struct->N_rb_ul, i_samps, q_samps, subfr_num, N_id_cell, &entry_->ul, al.data(), N_allocs,
                                            bits.data(), 6144, nb.data(), st.data(), pr.data(), tabs.data(), N_pucch, pb.data(), pnb.data(), prc.data());
    if (rc != 0) return LIBLTE_ERROR_INVALID_INPUTS;
    for (uint32 k = 0; k < N_allocs; k++) {
        status[k] = st[k] == 0 && nb[k] <= LIBLTE_MAX_MSG_SIZE ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
        if (status[k] == LIBLTE_SUCCESS) {
            std::copy(bits.begin() + (size_t)k * 6144, bits.begin() + (size_t)k * 6144 + nb[k], out_bits + (size_t)k * LIBLTE_MAX_MSG_SIZE);
            N_out_bits[k] = nb[k];
        }
    }
    for (uint32 r = 0; r < N_pucch; r++) {
        pucch_bits[2 * r] = pb[2 * r];
        if (pnb[r] == 2) pucch_bits[2 * r + 1] = pb[2 * r + 1];
        N_pucch_bits[r] = pnb[r];
        pucch_status[r] = prc[r] == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
    }
    return LIBLTE_SUCCESS;
}

LIBLTE_ERROR_ENUM liblte_phy_detect_prach(LIBLTE_PHY_STRUCT *phy_struct, float *samps_re, float *samps_im, uint32 freq_offset,
                                          uint32 *N_det_pre, uint32 *det_pre, uint32 *det_ta)
{
    if (phy_struct == NULL || samps_re == NULL || samps_im == NULL || N_det_pre == NULL || det_pre == NULL || det_ta == NULL ||
        !phy_struct->ul_init || phy_struct->prach_preamble_format > 4)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    mi_lte_prach_cfg pc = {phy_struct->prach_root_seq_idx, phy_struct->prach_preamble_format, phy_struct->prach_zczc,
                           phy_struct->prach_hs_flag ? 1u : 0u, freq_offset};
    // no spectra handed over: the plan generates the cell's root set itself (prach.hip / prach_sets.hpp) -- the struct's prach_x_u_fft_* tables are
    // not read, in either build
    int rc = mi_lte_detect_prach_host(c, phy_struct->N_samps_per_symb, phy_struct->N_rb_ul, &pc, NULL, NULL, 0, samps_re, samps_im, N_det_pre, det_pre, det_ta);
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
}

// ---- control region: PCFICH + PDCCH common search space (LTE_fdd_dl_fs_samp_buf.cc:445-470)

LIBLTE_ERROR_ENUM liblte_phy_pdcch_channel_decode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_cell,
                                                  uint8 N_ant, float phich_res, LIBLTE_RRC_PHICH_DURATION_ENUM phich_dur,
                                                  LIBLTE_PHY_PCFICH_STRUCT *pcfich, LIBLTE_PHY_PHICH_STRUCT *phich, LIBLTE_PHY_PDCCH_STRUCT *pdcch)
{
    if (phy_struct == NULL || subframe == NULL || pcfich == NULL || phich == NULL || pdcch == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    // MI_LTE_PDCCH_PER_PORT=1 selects the standard transmit-diversity combiner instead of the reference's arithmetic (mi_lte.h)
    const char      *pp    = getenv("MI_LTE_PDCCH_PER_PORT");
    const uint32_t   flags = (pp && atoi(pp)) ? MI_LTE_PDCCH_PER_PORT_ESTIMATES : 0u;
    uint32_t         cfi = 0, n_symbs = 0, n_dci = 0, n_reg = 0, k[75];
    mi_lte_pdcch_dci dci[MI_LTE_PDCCH_MAX_DCI];
    pcfich->N_reg = 4;
    if (mi_lte_ctrl_reg_positions(phy_struct->N_rb_dl, N_id_cell, phich_res, pcfich->k, pcfich->n, &n_reg, k) != MI_LTE_OK)
        return LIBLTE_ERROR_INVALID_INPUTS;
    int rc = mi_lte_pdcch_channel_decode_host(c, phy_struct->N_rb_dl, &subframe->rx_symb_re[0][0], &subframe->rx_symb_im[0][0],
                                              &subframe->rx_ce_re[0][0][0], &subframe->rx_ce_im[0][0][0], subframe->num, N_id_cell, N_ant, phich_res,
                                              phich_dur == LIBLTE_RRC_PHICH_DURATION_NORMAL ? 0u : 1u, flags, &cfi, &n_symbs, &n_dci, dci);
    if (rc < 0) return LIBLTE_ERROR_INVALID_INPUTS;
    if (rc == 3) return LIBLTE_ERROR_INVALID_CRC; // PCFICH: nothing else is written, as in the reference (:4563-4570)
    pcfich->cfi               = cfi;
    phy_struct->N_group_phich = n_reg / 3;
    phich->N_reg              = n_reg;
    for (uint32 i = 0; i < n_reg; i++) phich->k[i] = k[i];
    pdcch->N_symbs = n_symbs;
    pdcch->N_alloc = n_dci;
    for (uint32 a = 0; a < n_dci; a++) { // the fields dci_1a_unpack / dci_1c_unpack write (:13273-13378, :13400-13611)
        LIBLTE_PHY_ALLOCATION_STRUCT *o = &pdcch->alloc[a];
        const mi_lte_pdsch_alloc     &m = dci[a].alloc;
        o->N_prb  = m.N_prb;
        o->mcs    = (uint8)dci[a].mcs;
        if (dci[a].format == 0) o->rv_idx = m.rv_idx;
        for (uint32 i = 0; i < m.N_prb && i < LIBLTE_PHY_N_RB_DL_MAX; i++) { o->prb[0][i] = m.prb[0][i]; o->prb[1][i] = m.prb[1][i]; }
        o->mod_type       = LIBLTE_PHY_MODULATION_TYPE_QPSK;
        o->pre_coder_type = LIBLTE_PHY_PRE_CODER_TYPE_TX_DIVERSITY;
        o->tx_mode        = m.tx_mode;
        o->N_codewords    = 1;
        o->tbs            = m.tbs;
        o->rnti           = (uint16)m.rnti;
    }
    return rc == 0 ? LIBLTE_SUCCESS : rc == 4 ? LIBLTE_ERROR_INVALID_CONTENTS : LIBLTE_ERROR_INVALID_INPUTS;
}

LIBLTE_ERROR_ENUM liblte_phy_bch_channel_decode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_cell, uint8 *N_ant,
                                                uint8 *out_bits, uint32 *N_out_bits, uint8 *offset)
{
    if (phy_struct == NULL || subframe == NULL || N_id_cell > 503 || N_ant == NULL || out_bits == NULL || N_out_bits == NULL || offset == NULL)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    int rc = mi_lte_bch_channel_decode_host(c, phy_struct->N_rb_dl, &subframe->rx_symb_re[0][0], &subframe->rx_symb_im[0][0], &subframe->rx_ce_re[0][0][0],
                                            &subframe->rx_ce_im[0][0][0], N_id_cell, N_ant, out_bits, N_out_bits, offset);
    return rc == 0 ? LIBLTE_SUCCESS : rc == 2 ? LIBLTE_ERROR_DECODE_FAIL : LIBLTE_ERROR_INVALID_INPUTS;
}

// ---- initial synchronisation (LTE_fdd_dl_fs_samp_buf.cc:277-395)

LIBLTE_ERROR_ENUM liblte_phy_dl_find_coarse_timing_and_freq_offset(LIBLTE_PHY_STRUCT *phy_struct, float *i_samps, float *q_samps, uint32 N_slots,
                                                                   LIBLTE_PHY_COARSE_TIMING_STRUCT *timing_struct)
{
    if (phy_struct == NULL || i_samps == NULL || q_samps == NULL || timing_struct == NULL) return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    mi_lte_coarse_timing t;
    if (mi_lte_dl_find_coarse_timing_host(c, phy_struct->N_samps_per_symb, phy_struct->N_rb_dl, i_samps, q_samps, N_slots, &t) != 0)
        return LIBLTE_ERROR_INVALID_INPUTS;
    timing_struct->n_corr_peaks = t.n_corr_peaks;
    for (uint32 i = 0; i < t.n_corr_peaks && i < LIBLTE_PHY_N_MAX_ROUGH_CORR_SEARCH_PEAKS; i++) { // the reference writes the found peaks only
        timing_struct->freq_offset[i] = t.freq_offset[i];
        for (uint32 j = 0; j < 7; j++) timing_struct->symb_starts[i][j] = t.symb_starts[i][j];
    }
    return LIBLTE_SUCCESS;
}

LIBLTE_ERROR_ENUM liblte_phy_find_pss_and_fine_timing(LIBLTE_PHY_STRUCT *phy_struct, float *i_samps, float *q_samps, uint32 *symb_starts,
                                                      uint32 *N_id_2, uint32 *pss_symb, float *pss_thresh, float *freq_offset)
{
    if (phy_struct == NULL || i_samps == NULL || q_samps == NULL || symb_starts == NULL || N_id_2 == NULL || pss_symb == NULL || pss_thresh == NULL)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    float f = 0;
    int   rc = mi_lte_find_pss_host(c, phy_struct->N_samps_per_symb, phy_struct->N_rb_dl, i_samps, q_samps, symb_starts, N_id_2, pss_symb, pss_thresh, &f);
    if (rc == 0 && freq_offset) *freq_offset = f;
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
}

LIBLTE_ERROR_ENUM liblte_phy_find_sss(LIBLTE_PHY_STRUCT *phy_struct, float *i_samps, float *q_samps, uint32 N_id_2, uint32 *symb_starts, float pss_thresh,
                                      uint32 *N_id_1, uint32 *frame_start_idx)
{
    if (phy_struct == NULL || i_samps == NULL || q_samps == NULL || symb_starts == NULL || N_id_1 == NULL || frame_start_idx == NULL)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    int rc = mi_lte_find_sss_host(c, phy_struct->N_samps_per_symb, phy_struct->N_rb_dl, i_samps, q_samps, N_id_2, symb_starts, pss_thresh, N_id_1, frame_start_idx);
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
}

// ---- PUCCH formats 1 / 1a / 1b (LTE_fdd_enb_phy.cc:867)

#ifdef MI_LTE_SHIM_OWN_LIFECYCLE
int32 W_5_4_1_2[3][4] = {{1, 1, 1, 1}, {1, -1, 1, -1}, {1, -1, -1, 1}}; // 36.211 table 5.4.1-2; external like the reference's (liblte_phy.cc:161): callers name it
#endif

LIBLTE_ERROR_ENUM liblte_phy_pucch_format_1_1a_1b_channel_decode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe,
                                                                 LIBLTE_PHY_PUCCH_FORMAT_ENUM format, uint32 N_id_cell, uint8 N_ant, uint32 N_1_p_pucch,
                                                                 uint8 *out_bits, uint32 *N_out_bits)
{
    (void)N_id_cell;
    if (phy_struct == NULL || subframe == NULL || !(format == LIBLTE_PHY_PUCCH_FORMAT_1 || format == LIBLTE_PHY_PUCCH_FORMAT_1A || format == LIBLTE_PHY_PUCCH_FORMAT_1B) ||
        out_bits == NULL || N_out_bits == NULL || subframe->num > 9 || N_1_p_pucch >= LIBLTE_PHY_N_RB_UL_MAX / 2 || N_ant != 1)
        return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_CTX(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    const uint32 N = subframe->num, n = N_1_p_pucch;
    // the library's own generator (mi_lte_ul_pucch_tables restates generate_dmrs_pucch, liblte_phy.cc:6986-7129, value for value:
    // tests/test_uplink_cpu.py), asked once per (subframe, resource) with what liblte_phy_ul_init was given -- not the struct's pucch_* tables, in either build
    if (!phy_struct->ul_init) return LIBLTE_ERROR_INVALID_INPUTS;
    std::vector<float> &tab = entry_->pucch[N * 256u + n];
    if (tab.empty()) {
        tab.resize(MI_LTE_PUCCH_TAB_FLOATS);
        if (mi_lte_ul_pucch_tables(&entry_->ul, entry_->ul_cell, N, n, entry_->n_cs_an, entry_->delta_pucch_shift, phy_struct->N_ant, &tab[0]) != MI_LTE_OK) {
            tab.clear();
            return LIBLTE_ERROR_INVALID_INPUTS;
        }
    }
    const float *t = &tab[0];
    uint32_t nb = 0;
    int rc = mi_lte_pucch_decode_host(c, phy_struct->N_rb_ul, &subframe->rx_symb_re[0][0], &subframe->rx_symb_im[0][0], (uint32_t)format, N_ant, N_1_p_pucch, t,
                                      out_bits, &nb);
    if (rc == 0 || rc == 1) *N_out_bits = nb;
    return rc == 0 ? LIBLTE_SUCCESS : LIBLTE_ERROR_INVALID_INPUTS;
}

// ---- the scheduler-side helpers (SURVEY 8b's "CPU pass-through" functions), own-lifecycle build only: with them a caller's link needs no
// object of the reference's PHY for anything but transmitting.  Host arithmetic in libmi_lte.so (sched.cc); each is compared with the
// compiled reference over its whole argument range by `lifecycle_check helpers`.
#ifdef MI_LTE_SHIM_OWN_LIFECYCLE
// liblte_phy.h:1210, liblte_phy.cc:6251-6357 -- LTE_fdd_enb_mac.cc (scheduler), LTE_fdd_dl_file_gen
LIBLTE_ERROR_ENUM liblte_phy_get_tbs_mcs_and_n_prb_for_dl(uint32 N_bits, uint32 N_subframe, uint32 N_rb_dl, uint16 rnti, uint32 *tbs, uint8 *mcs, uint32 *N_prb)
{
    return (LIBLTE_ERROR_ENUM)mi_lte_get_tbs_mcs_and_n_prb_for_dl(N_bits, N_subframe, N_rb_dl, rnti, tbs, mcs, N_prb);
}
// liblte_phy.h:1234, liblte_phy.cc:6359-6407
LIBLTE_ERROR_ENUM liblte_phy_get_tbs_and_n_prb_for_dl(uint32 N_bits, uint32 N_rb_dl, uint8 mcs, uint32 *tbs, uint32 *N_prb)
{
    return (LIBLTE_ERROR_ENUM)mi_lte_get_tbs_and_n_prb_for_dl(N_bits, N_rb_dl, mcs, tbs, N_prb);
}
// liblte_phy.h:1253, liblte_phy.cc:6409-6475
LIBLTE_ERROR_ENUM liblte_phy_get_tbs_mcs_and_n_prb_for_ul(uint32 N_bits, uint32 N_rb_ul, uint32 *tbs, uint8 *mcs, uint32 *N_prb)
{
    return (LIBLTE_ERROR_ENUM)mi_lte_get_tbs_mcs_and_n_prb_for_ul(N_bits, N_rb_ul, tbs, mcs, N_prb);
}
// liblte_phy.h:1271, liblte_phy.cc:6477-6505 (no argument checks there either; called under the MAC's sys_info_sem, LTE_fdd_enb_phy.cc:357-369: reads two fields)
LIBLTE_ERROR_ENUM liblte_phy_get_n_cce(LIBLTE_PHY_STRUCT *phy_struct, float phich_res, uint32 N_pdcch_symbs, uint8 N_ant, uint32 *N_cce)
{
    (void)phich_res;
    *N_cce = mi_lte_get_n_cce(phy_struct->N_rb_dl, phy_struct->N_group_phich, N_pdcch_symbs, N_ant);
    return LIBLTE_SUCCESS;
}
// liblte_phy.h:830, liblte_phy.cc:3183-3217
void liblte_phy_pucch_map_sr_config_idx(uint32 i_sr, uint32 *sr_periodicity, uint32 *N_offset_sr) { mi_lte_pucch_map_sr_config_idx(i_sr, sr_periodicity, N_offset_sr); }
// liblte_phy.h:1337, liblte_phy.cc:9753-9865
void liblte_phy_code_block_segmentation(uint8 *b_bits, uint32 N_b_bits, uint32 *N_codeblocks, uint32 *N_filler_bits, uint8 *c_bits, uint32 N_c_bits_max, uint32 *N_c_bits)
{
    mi_lte_code_block_segmentation(b_bits, N_b_bits, N_codeblocks, N_filler_bits, c_bits, N_c_bits_max, N_c_bits);
}
// liblte_phy.h:1357, liblte_phy.cc:9875-9987
void liblte_phy_code_block_desegmentation(uint8 *c_bits, uint32 *N_c_bits, uint32 N_c_bits_max, uint32 tbs, uint8 *b_bits, uint32 N_b_bits)
{
    mi_lte_code_block_desegmentation(c_bits, N_c_bits, N_c_bits_max, tbs, b_bits, N_b_bits);
}

// ---- the transmit side (2: "CPU pass-through"), own-lifecycle build only: host code in libmi_lte.so (tx.cc ...) behind the reference's
// signatures, the struct's fields handed over as arguments, its long-lived scratch in the entry.  `lifecycle_check tx` compares each with the
// compiled reference.
static_assert(sizeof(((LIBLTE_PHY_SUBFRAME_STRUCT *)0)->tx_symb_re) == MI_LTE_TX_GRID_FLOATS * sizeof(float), "grid layout");
static_assert(sizeof(((LIBLTE_PHY_ALLOCATION_STRUCT *)0)->prb) == sizeof(((mi_lte_tx_alloc *)0)->prb), "PRB list layout");
// liblte_phy.h:1288, liblte_phy.cc:11081-11237
void liblte_phy_rate_match_turbo(LIBLTE_PHY_STRUCT *phy_struct, uint8 *d_bits, uint32 N_d_bits, uint32 N_codeblocks, uint32 tx_mode, uint32 N_soft, uint32 M_dl_harq,
                                 LIBLTE_PHY_CHAN_TYPE_ENUM chan_type, uint32 rv_idx, uint32 N_e_bits, uint8 *e_bits)
{
    (void)phy_struct; // (the reference uses it for scratch only)
    mi_lte_rate_match_turbo(d_bits, N_d_bits, N_codeblocks, tx_mode, N_soft, M_dl_harq, (uint32_t)chan_type, rv_idx, N_e_bits, e_bits);
}
// liblte_phy.h:888, liblte_phy.cc:3489-3688
LIBLTE_ERROR_ENUM liblte_phy_pdsch_channel_encode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_PDCCH_STRUCT *pdcch, uint32 N_id_cell, uint8 N_ant, LIBLTE_PHY_SUBFRAME_STRUCT *subframe)
{
    if (!phy_struct || !pdcch || !subframe || pdcch->N_alloc > LIBLTE_PHY_PDCCH_MAX_ALLOC) return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_ENTRY(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    mi_lte_tx_alloc al[LIBLTE_PHY_PDCCH_MAX_ALLOC];
    for (uint32 i = 0; i < pdcch->N_alloc; i++) to_tx_alloc(&pdcch->alloc[i], &al[i]);
    return (LIBLTE_ERROR_ENUM)mi_lte_pdsch_channel_encode(t, phy_struct->N_rb_dl, phy_struct->N_sc_rb_dl, al, pdcch->N_alloc, pdcch->N_symbs, N_id_cell, N_ant, subframe->num,
                                                          &subframe->tx_symb_re[0][0][0], &subframe->tx_symb_im[0][0][0]);
}
// liblte_phy.h:927, liblte_phy.cc:3863-3966
LIBLTE_ERROR_ENUM liblte_phy_bch_channel_encode(LIBLTE_PHY_STRUCT *phy_struct, uint8 *in_bits, uint32 N_in_bits, uint32 N_id_cell, uint8 N_ant, LIBLTE_PHY_SUBFRAME_STRUCT *subframe,
                                                uint32 sfn)
{
    if (!phy_struct || !in_bits || !subframe) return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_ENTRY(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    return (LIBLTE_ERROR_ENUM)mi_lte_bch_channel_encode(t, phy_struct->N_rb_dl, phy_struct->N_sc_rb_dl, in_bits, N_in_bits, N_id_cell, N_ant, sfn, &subframe->tx_symb_re[0][0][0],
                                                        &subframe->tx_symb_im[0][0][0]);
}
// liblte_phy.h:1034, liblte_phy.cc:5144-5263
LIBLTE_ERROR_ENUM liblte_phy_map_crs(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_cell, uint8 N_ant)
{
    if (!phy_struct || !subframe) return LIBLTE_ERROR_INVALID_INPUTS;
    return (LIBLTE_ERROR_ENUM)mi_lte_map_crs(phy_struct->N_rb_dl, phy_struct->N_sc_rb_dl, subframe->num, N_id_cell, N_ant, &subframe->tx_symb_re[0][0][0], &subframe->tx_symb_im[0][0][0]);
}
// liblte_phy.h:1051, liblte_phy.cc:5265-5304
LIBLTE_ERROR_ENUM liblte_phy_map_pss(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_2, uint8 N_ant)
{
    if (!phy_struct || !subframe) return LIBLTE_ERROR_INVALID_INPUTS;
    return (LIBLTE_ERROR_ENUM)mi_lte_map_pss(phy_struct->N_rb_dl, phy_struct->N_sc_rb_dl, N_id_2, N_ant, &subframe->tx_symb_re[0][0][0], &subframe->tx_symb_im[0][0][0]);
}
// liblte_phy.h:1089, liblte_phy.cc:5520-5576
LIBLTE_ERROR_ENUM liblte_phy_map_sss(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_1, uint32 N_id_2, uint8 N_ant)
{
    if (!phy_struct || !subframe) return LIBLTE_ERROR_INVALID_INPUTS;
    return (LIBLTE_ERROR_ENUM)mi_lte_map_sss(phy_struct->N_rb_dl, phy_struct->N_sc_rb_dl, subframe->num, N_id_1, N_id_2, N_ant, &subframe->tx_symb_re[0][0][0],
                                             &subframe->tx_symb_im[0][0][0]);
}
// liblte_phy.h:988, liblte_phy.cc:4113-4517
static_assert(sizeof(LIBLTE_PHY_PCFICH_STRUCT) == sizeof(mi_lte_pcfich) && sizeof(LIBLTE_PHY_PHICH_STRUCT) == sizeof(mi_lte_phich) && sizeof(bool) == 1, "control structs");
LIBLTE_ERROR_ENUM liblte_phy_pdcch_channel_encode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_PCFICH_STRUCT *pcfich, LIBLTE_PHY_PHICH_STRUCT *phich, LIBLTE_PHY_PDCCH_STRUCT *pdcch,
                                                  uint32 N_id_cell, uint8 N_ant, float phich_res, LIBLTE_RRC_PHICH_DURATION_ENUM phich_dur, LIBLTE_PHY_SUBFRAME_STRUCT *subframe)
{
    (void)phich_res; // (the reference does not read it here either: the group count is the struct's)
    if (!phy_struct || !pcfich || !phich || !pdcch || !subframe || pdcch->N_alloc > LIBLTE_PHY_PDCCH_MAX_ALLOC) return LIBLTE_ERROR_INVALID_INPUTS;
    MI_LOCKED_ENTRY(phy_struct, LIBLTE_ERROR_INVALID_INPUTS);
    mi_lte_tx_alloc al[LIBLTE_PHY_PDCCH_MAX_ALLOC];
    for (uint32 i = 0; i < pdcch->N_alloc; i++) to_tx_alloc(&pdcch->alloc[i], &al[i]);
    const int rc = mi_lte_pdcch_channel_encode(t, phy_struct->N_rb_dl, phy_struct->N_rb_ul, phy_struct->N_sc_rb_dl, phy_struct->N_group_phich, phy_struct->N_sf_phich,
                                               (mi_lte_pcfich *)pcfich, (mi_lte_phich *)phich, al, pdcch->N_alloc, &pdcch->N_symbs, N_id_cell, N_ant, (uint32_t)phich_dur, subframe->num,
                                               &subframe->tx_symb_re[0][0][0], &subframe->tx_symb_im[0][0][0]);
    for (uint32 i = 0; i < pdcch->N_alloc; i++) pdcch->alloc[i].tbs = al[i].tbs; // (the DCI's transport block size, as dci_1a_pack leaves it: liblte_phy.cc:13206, :13232)
    return (LIBLTE_ERROR_ENUM)rc;
}
// liblte_phy.h:704, liblte_phy.cc:2664-2799 (the UE side of the reference's loop-back: nothing in LTE_fdd_enodeb calls it)
LIBLTE_ERROR_ENUM liblte_phy_pusch_channel_encode(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_ALLOCATION_STRUCT *alloc, uint32 N_id_cell, uint8 N_ant, LIBLTE_PHY_SUBFRAME_STRUCT *subframe)
{
    if (!phy_struct || !alloc || !subframe || !phy_struct->ul_init) return LIBLTE_ERROR_INVALID_INPUTS; // (without liblte_phy_ul_init the reference reads tables nobody filled)
    std::shared_ptr<Entry> entry_ = entry_for(phy_struct);
    std::lock_guard<std::mutex> call_lock_(entry_->mu);
    if (!entry_->tx_ul && mi_lte_tx_ul_create(&entry_->tx_ul) != MI_LTE_OK) return LIBLTE_ERROR_INVALID_INPUTS;
    mi_lte_tx_alloc al;
    to_tx_alloc(alloc, &al);
    return (LIBLTE_ERROR_ENUM)mi_lte_pusch_channel_encode(entry_->tx_ul, &entry_->ul, entry_->ul_cell, phy_struct->N_rb_ul, phy_struct->N_sc_rb_ul, &al, N_id_cell, N_ant, subframe->num,
                                                          &subframe->tx_symb_re[0][0][0], &subframe->tx_symb_im[0][0][0]);
}
// liblte_phy.h:845, liblte_phy.cc:3219-3297
LIBLTE_ERROR_ENUM liblte_phy_generate_prach(LIBLTE_PHY_STRUCT *phy_struct, uint32 preamble_idx, uint32 freq_offset, float *samps_re, float *samps_im)
{
    if (!phy_struct || !samps_re || !samps_im || !phy_struct->ul_init) return LIBLTE_ERROR_INVALID_INPUTS;
    const mi_lte_prach_cfg pc = {phy_struct->prach_root_seq_idx, phy_struct->prach_preamble_format, phy_struct->prach_zczc, phy_struct->prach_hs_flag ? 1u : 0u, freq_offset};
    return (LIBLTE_ERROR_ENUM)mi_lte_generate_prach(&pc, phy_struct->FFT_size, phy_struct->N_rb_ul, phy_struct->N_sc_rb_ul, preamble_idx, freq_offset, samps_re, samps_im);
}
// liblte_phy.h:1152, liblte_phy.cc:5862-5903
LIBLTE_ERROR_ENUM liblte_phy_create_dl_subframe(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint8 ant, float *i_samps, float *q_samps)
{
    if (!phy_struct || !subframe) return LIBLTE_ERROR_INVALID_INPUTS;
    return (LIBLTE_ERROR_ENUM)mi_lte_create_dl_subframe(phy_struct->N_samps_per_symb, phy_struct->FFT_size - 2 * phy_struct->FFT_pad_size, phy_struct->N_samps_cp_l_0,
                                                        phy_struct->N_samps_cp_l_else, &subframe->tx_symb_re[0][0][0], &subframe->tx_symb_im[0][0][0], ant, i_samps, q_samps);
}
#endif
