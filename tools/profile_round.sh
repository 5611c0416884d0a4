#!/bin/bash
# tools/profile_round.sh <tag>: the round's profile set on the GPU box (under gpurun_out/; tools/summarize_profile.py turns it into profiles/<tag>_*):
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes for chain / frontend / uplink, SQ counters for chain and uplink, one bench line per workload.
set -u
TAG=${1:-r05}
cd "$(dirname "$0")/.."
for w in chain frontend uplink; do bash tools/profile_bench.sh ${TAG}_$w --workload $w > /dev/null 2>&1; done
for w in chain uplink; do
  bash tools/pmc_sq.sh ${TAG}a_$w --workload $w 2>&1 | grep "k_" > gpurun_out/sq_${TAG}_$w.txt
  SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/pmc_sq.sh ${TAG}b_$w --workload $w 2>&1 | grep "k_" >> gpurun_out/sq_${TAG}_$w.txt
done
mkdir -p gpurun_out/bench_$TAG
for w in uplink turbo frontend frontend2 control sync; do python bench.py --workload $w > gpurun_out/bench_$TAG/$w.json 2> gpurun_out/bench_$TAG/$w.err; done
python bench.py > gpurun_out/bench_$TAG/chain.json 2> gpurun_out/bench_$TAG/chain.err
ls gpurun_out/bench_$TAG gpurun_out/prof_${TAG}_chain
