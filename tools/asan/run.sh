#!/bin/bash
# tools/asan/run.sh [iterations]: the host-side transmit functions (tx*.cc, sched.cc and what they use) built with g++ -fsanitize=address,undefined
# and driven with random -- also out-of-range -- inputs: memory errors and undefined behaviour only, no comparison (that is lifecycle_check tx).
# CPU only; GPU sanitizers are not available on this pool.
set -e
cd "$(dirname "$0")/../.."
OUT=${TMPDIR:-/tmp}/mi_lte_asan_tx
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Iinclude tools/asan/tx_driver.cc \
    openlte_amd/csrc/tx.cc openlte_amd/csrc/tx_ctrl.cc openlte_amd/csrc/tx_ul.cc openlte_amd/csrc/sched.cc openlte_amd/csrc/synth.cc openlte_amd/csrc/ul_rs.cc -o $OUT
ASAN_OPTIONS=detect_leaks=1 $OUT ${1:-3000}
