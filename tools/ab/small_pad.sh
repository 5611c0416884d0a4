cd "$(dirname "$0")/../.."
cp openlte_amd/libmi_lte.so _ko/lib_BASE.so
for v in BASE PAD8 PAD16 BASE; do cp _ko/lib_$v.so openlte_amd/libmi_lte.so; echo "== $v"; timeout 100 python tools/siso_modes_sweep.py 2>&1 | grep -E "K 6144 +1 blocks"; done
cp _ko/lib_BASE.so openlte_amd/libmi_lte.so
