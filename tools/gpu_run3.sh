set -u
mkdir -p gpurun_out/r02c
python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" | tail -30 > gpurun_out/r02c/pytest.txt
tail -4 gpurun_out/r02c/pytest.txt
python bench.py > gpurun_out/r02c/bench_chain.json 2> gpurun_out/r02c/bench_chain.err
tail -3 gpurun_out/r02c/bench_chain.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c/bench_chain.json'))
print(d['value'], d['ms_per_step'], d.get('crc_pass'), d.get('sampled_subframes_equal_cpu_restatement'), d.get('from_host_buffers'))
print(d.get('cpu_baseline_all_cores'))
PY
cat gpurun_out/scan_timing_*.txt
