#!/bin/bash
# On the GPU box: tools/ab/run_variants.sh <workload> [bench args]  -- per-kernel ms of every _ko/lib_*.so variant, the current library first and last
cp openlte_amd/libmi_lte.so _ko/lib_BASE.so
for v in BASE $(ls _ko | sed 's/^lib_//;s/\.so$//' | grep -v BASE) BASE; do
  cp _ko/lib_$v.so openlte_amd/libmi_lte.so; echo "== $v"
  timeout ${AB_TIMEOUT:-90} python tools/ab/bench_kernels.py "$@" --no-cpu-baseline 2>&1 | tail -1 # (a variant that hangs must not take the GPU call with it)
done
cp _ko/lib_BASE.so openlte_amd/libmi_lte.so
