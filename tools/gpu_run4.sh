set -u
mkdir -p gpurun_out/r02d
python -m pytest tests/test_bcjr_gpu.py tests/test_dropin_gpu.py "tests/test_chain_gpu.py::test_packed_output_equals_the_unpacked_bits" -m gpu -q -x 2>&1 | grep -v "^ERROR: DCI" > gpurun_out/r02d/pytest.txt
tail -40 gpurun_out/r02d/pytest.txt
python bench.py --workload turbo --decoder bcjr --no-cpu-baseline > gpurun_out/r02d/bench_turbo_bcjr.json 2> gpurun_out/r02d/bench_turbo_bcjr.err
tail -3 gpurun_out/r02d/bench_turbo_bcjr.err
python -c "
import json
d=json.load(open('gpurun_out/r02d/bench_turbo_bcjr.json'))
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()}, d.get('sampled_blocks_equal_tx_bits'))
"
