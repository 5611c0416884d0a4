#!/bin/bash
# per-call forms: spinning on hipStreamQuery (BASE at the time) against spinning on a word that a one-thread kernel behind the call's work stores
# (FLAG: ctx.cc built with -DMI_FLAG_WAIT from commit 2f0f.. of round 2; the variant has since become the library's wait, mi_stream_wait_polling)
cd /root/repo
cp openlte_amd/libmi_lte.so _ko/lib_BASE.so
for v in BASE FLAG BASE FLAG; do
  cp _ko/lib_$v.so openlte_amd/libmi_lte.so; echo "== $v"
  timeout 100 bash tools/scan_timing.sh 2>&1 | grep "gpu" | sed "s/.*= //" | tr "\n" " "; echo
  timeout 100 python tools/ul_timing.py 2>&1 | tail -3 | head -2 | sed "s/.*decode): //"
done
cp _ko/lib_BASE.so openlte_amd/libmi_lte.so
