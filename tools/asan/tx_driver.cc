// ASan / UBSan driver for the host-side transmit functions (CPU build only): random inputs through every entry point, no comparison -- memory and UB only
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mi_lte.h"
static uint32_t x = 88172645u;
static uint32_t rnd() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
int main(int argc, char **argv)
{
    std::vector<float> re(MI_LTE_TX_GRID_FLOATS), im(MI_LTE_TX_GRID_FLOATS), si(40000), sq(40000), pr(80000), pq(80000);
    mi_lte_tx *t = nullptr; mi_lte_tx_ul *tu = nullptr;
    mi_lte_tx_create(&t); mi_lte_tx_ul_create(&tu);
    static const uint32_t rbs[6] = {6, 15, 25, 50, 75, 100}, ffts[6] = {128, 256, 512, 1024, 2048, 2048};
    std::vector<uint8_t> msg(20000);
    long calls = 0, ok = 0;
    const int n_it = argc > 1 ? atoi(argv[1]) : 3000;
    for (int it = 0; it < n_it; it++) {
        const int b = rnd() % 6; const uint32_t n_rb = rbs[b], n_ant = 1u << (rnd() % 3), cell = rnd() % 504, sf = rnd() % 10;
        for (auto &m : msg) m = rnd() & 1;
        mi_lte_tx_alloc al[6]; memset(al, 0, sizeof al);
        const uint32_t n_al = 1 + rnd() % 6;
        for (uint32_t a = 0; a < n_al; a++) {
            al[a].msg[0] = msg.data(); al[a].msg[1] = msg.data() + 7000; al[a].msg_bits[0] = rnd() % 6000; al[a].msg_bits[1] = rnd() % 6000;
            al[a].mod_type = rnd() % 4; al[a].chan_type = rnd() % 5 == 0 ? 2 : 0; al[a].pre_coder_type = rnd() % 9 == 0; al[a].tbs = 1 + rnd() % (rnd() % 4 ? 3000 : 31000);
            al[a].rv_idx = rnd() % 4; al[a].N_prb = rnd() % (n_rb + 2); al[a].N_codewords = 1 + (rnd() % 7 == 0); al[a].N_layers = 1; al[a].tx_mode = 1 + rnd() % 9; al[a].rnti = rnd() & 0xFFFF;
            al[a].mcs = rnd() % 29; al[a].tpc = rnd() % 4; al[a].ndi = rnd() & 1;
            for (uint32_t i = 0; i < 110; i++) al[a].prb[0][i] = rnd() % (n_rb + (rnd() % 50 == 0)), al[a].prb[1][i] = rnd() % n_rb;
        }
        uint32_t nsym = 1 + rnd() % 4;
        calls += 8;
        ok += 0 == mi_lte_map_crs(n_rb, 12, sf, cell, n_ant, re.data(), im.data());
        ok += 0 == mi_lte_map_pss(n_rb, 12, rnd() % 4, n_ant, re.data(), im.data());
        ok += 0 == mi_lte_map_sss(n_rb, 12, sf, rnd() % 168, rnd() % 3, n_ant, re.data(), im.data());
        ok += 0 == mi_lte_bch_channel_encode(t, n_rb, 12, msg.data(), 24, cell, n_ant, rnd() % 1024, re.data(), im.data());
        mi_lte_pcfich pc; mi_lte_phich ph; memset(&pc, 0, sizeof pc); memset(&ph, 0, sizeof ph);
        pc.cfi = rnd() % 5;
        const uint32_t n_group = 1 + rnd() % 25;
        for (uint32_t g = 0; g < 25; g++) for (int q = 0; q < 8; q++) ph.present[g][q] = rnd() % 3 == 0, ph.b[g][q] = rnd() & 1;
        ok += 0 == mi_lte_pdcch_channel_encode(t, n_rb, n_rb, 12, n_group, 4, &pc, &ph, al, n_al, &nsym, cell, n_ant, rnd() % 2, sf, re.data(), im.data());
        ok += 0 == mi_lte_pdsch_channel_encode(t, n_rb, 12, al, n_al, 1 + rnd() % 4, cell, n_ant, sf, re.data(), im.data());
        ok += 0 == mi_lte_create_dl_subframe(ffts[b], 12 * n_rb, ffts[b] * 10 / 128, ffts[b] * 9 / 128, re.data(), im.data(), rnd() % 4, si.data(), sq.data());
        mi_lte_ul_cfg ul = {rnd() % 30, rnd() & 1, rnd() & 1, rnd() % 8, rnd() % 8};
        al[0].N_layers = 1; al[0].N_codewords = 1;
        ok += 0 == mi_lte_pusch_channel_encode(tu, &ul, cell, n_rb, 12, &al[0], cell, 1, sf, re.data(), im.data());
        if (it % 40 == 0) {
            mi_lte_prach_cfg pc2 = {rnd() % 838, rnd() % 5, rnd() % 16, rnd() % 2, 0};
            if (pc2.preamble_format == 4) pc2.root_seq_idx %= 138, pc2.zczc %= 7;
            if (mi_lte_generate_prach_len(ffts[b], pc2.preamble_format) <= 80000) { calls++; ok += 0 == mi_lte_generate_prach(&pc2, ffts[b], n_rb, 12, rnd() % 64, rnd() % (n_rb - 5), pr.data(), pq.data()); }
        }
        std::vector<uint8_t> d(3 * 6148), e(40000);
        for (auto &v : d) v = rnd() & 1;
        const uint32_t K = 40 + 8 * (rnd() % 760);
        mi_lte_rate_match_turbo(d.data(), 3 * (K + 4), rnd() % 4, rnd() % 10, rnd() % 300000, rnd() % 10, rnd() % 4, rnd() % 4, rnd() % 30000, e.data());
    }
    printf("asan driver: %ld calls, %ld accepted\n", calls, ok);
    mi_lte_tx_destroy(t); mi_lte_tx_ul_destroy(tu);
    return 0;
}
