#!/bin/bash
# per-call forms: spinning on hipStreamQuery (BASE at the time) against spinning on a word that a one-thread kernel behind the call's work stores
# (FLAG: the experimental branch of mi_stream_wait_polling, built with -DMI_FLAG_WAIT before it became the library's wait; results: profiles/r02k_scan_timing.txt)
cd "$(dirname "$0")/../.."
cp openlte_amd/libmi_lte.so _ko/lib_BASE.so
for v in BASE FLAG BASE FLAG; do
  cp _ko/lib_$v.so openlte_amd/libmi_lte.so; echo "== $v"
  timeout 100 bash tools/scan_timing.sh 2>&1 | grep "gpu" | sed "s/.*= //" | tr "\n" " "; echo
  timeout 100 python tools/ul_timing.py 2>&1 | tail -3 | head -2 | sed "s/.*decode): //"
done
cp _ko/lib_BASE.so openlte_amd/libmi_lte.so
