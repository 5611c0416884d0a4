set -u
mkdir -p gpurun_out/r02f
python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" > gpurun_out/r02f/pytest.txt
tail -5 gpurun_out/r02f/pytest.txt
bash tools/profile_all.sh r02f > gpurun_out/r02f/profile_all.log 2>&1
tail -3 gpurun_out/r02f/profile_all.log
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r02f/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
"
