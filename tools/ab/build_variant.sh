#!/bin/bash
# A/B builds of one kernel file: tools/ab/build_variant.sh NAME file-stem [-DFLAG ...]  ->  _ko/lib_NAME.so
# (the other objects are taken from openlte_amd/csrc as they are; run `make -C openlte_amd/csrc` first).  Together with
# tools/ab/run_variants.sh this is how the per-kernel tuning decisions in DESIGN.md 6.1 were measured: all variants in ONE gpurun
# call, on the same box, base variant first and last.
set -e
n=$1; f=$2; shift; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/_ko
cd $root/openlte_amd/csrc
objs=""
for o in ctx bcjr chain frontend pdcch prach sync turbo uplink; do if [ "$o" = "$f" ]; then objs="$objs /tmp/${f}_$n.o"; else objs="$objs $o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $f.hip -o /tmp/${f}_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs hostapi.o pipeline.o synth.o ul_rs.o ul_synth.o sched.o -o $root/_ko/lib_$n.so
