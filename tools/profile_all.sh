#!/bin/bash
# On the GPU box: the profile set DESIGN.md 6 cites, under gpurun_out/ (tag = $1): kernel trace + HBM counters for chain / uplink / turbo,
# SQ instruction counters for chain / uplink, and one bench line per workload.
set -u
TAG=${1:-r01e}
for w in chain uplink turbo; do bash tools/profile_bench.sh ${TAG}_$w --workload $w > /dev/null 2>&1; done
for w in chain uplink; do
  bash tools/pmc_sq.sh ${TAG}a_$w --workload $w 2>&1 | grep "k_" > gpurun_out/sq_$w.txt
  SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/pmc_sq.sh ${TAG}b_$w --workload $w 2>&1 | grep "k_" >> gpurun_out/sq_$w.txt
done
mkdir -p gpurun_out/bench_$TAG
for w in chain uplink turbo frontend control sync; do python bench.py --workload $w > gpurun_out/bench_$TAG/$w.json 2> gpurun_out/bench_$TAG/$w.err; done
python bench.py --workload turbo --decoder bcjr > gpurun_out/bench_$TAG/turbo_bcjr.json 2>/dev/null
python bench.py --workload chain --decoder bcjr --no-cpu-baseline > gpurun_out/bench_$TAG/chain_bcjr.json 2>/dev/null
ls gpurun_out/bench_$TAG
