set -u
mkdir -p gpurun_out/r02g
python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" > gpurun_out/r02g/pytest.txt
tail -4 gpurun_out/r02g/pytest.txt
bash tools/profile_all.sh r02g > gpurun_out/r02g/profile_all.log 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r02g/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
"
cat gpurun_out/scan_timing_*.txt
