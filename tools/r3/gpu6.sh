#!/bin/bash
# round 3, call 6: the profile set of the round (kernel trace + FETCH / WRITE PMC passes, SQ counters), the strong-scaling capture runs, the suite
cd /root/repo
mkdir -p gpurun_out/r03f
bash tools/profile_bench.sh r03_chain --workload chain > /dev/null 2>&1
bash tools/profile_bench.sh r03_turbo_bcjr --workload turbo --decoder bcjr > /dev/null 2>&1
bash tools/profile_bench.sh r03_turbo_bcjr_early --workload turbo --decoder bcjr_early > /dev/null 2>&1
bash tools/profile_bench.sh r03_uplink --workload uplink > /dev/null 2>&1
bash tools/profile_bench.sh r03_frontend2 --workload frontend2 > /dev/null 2>&1
bash tools/pmc_sq.sh r03a_chain --workload chain 2>&1 | grep "k_" > gpurun_out/r03f/sq_chain.txt
SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/pmc_sq.sh r03b_chain --workload chain 2>&1 | grep "k_" >> gpurun_out/r03f/sq_chain.txt
bash tools/pmc_sq.sh r03a_turbo_bcjr --workload turbo --decoder bcjr 2>&1 | grep "k_" > gpurun_out/r03f/sq_turbo_bcjr.txt
for g in 1 2 4; do timeout 600 python bench.py --strong --gpus $g --oversubscribe --steps 3 --warmup 1 --units 16384 2>&1 | tail -1 > gpurun_out/r03f/bench_strong_$g.json; done
python - <<'PY'
import json
for g in (1, 2, 4):
    d = json.loads(open('/root/repo/gpurun_out/r03f/bench_strong_%d.json' % g).read().strip().splitlines()[-1])
    print(g, d['value'], d['crc_pass'], d['h2d_GBps'])
PY
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee gpurun_out/r03f/pytest_gpu.txt
du -sh gpurun_out/prof_r03_* | tail -8
