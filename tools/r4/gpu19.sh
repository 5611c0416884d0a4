#!/bin/bash
cd /root/repo
o=gpurun_out/r04p; mkdir -p $o
cp openlte_amd/libmi_lte.so _ko/lib_BASE.so
for v in BASE V2TERMS BASE V2TERMS BASE V2TERMS; do cp _ko/lib_$v.so openlte_amd/libmi_lte.so; echo "== $v"; timeout 120 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1; done | tee $o/variants_siso_terms.txt
cp _ko/lib_BASE.so openlte_amd/libmi_lte.so
