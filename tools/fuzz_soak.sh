#!/bin/bash
# tools/fuzz_soak.sh <first seed> <last seed>: the differential tests of tests/test_fuzz_gpu.py over further seeds (the suite runs seed 0);
# one line per seed and the report of each under gpurun_out/fuzz_soak/.  On the GPU box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/fuzz_soak
for s in $(seq $1 $2); do
  MI_LTE_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -1 | sed "s/^/seed $s: /" | tee -a gpurun_out/fuzz_soak/summary.txt
  cp gpurun_out/fuzz_report.json gpurun_out/fuzz_soak/report_seed_$s.json
done
python - <<'PY'
import glob, json
tot = dict(downlink=0, dl_identical=0, ce_phase_ties=0, uplink=0, ul_diff_bits=0, ul_bits=0, control=0, pbch=0, prach=0, sync=0)
for f in sorted(glob.glob("gpurun_out/fuzz_soak/report_seed_*.json")):
    d = json.load(open(f))
    tot["downlink"] += d["downlink"]["cases"]; tot["dl_identical"] += d["downlink"]["identical_soft_bits_verdict_and_bits"]
    tot["ce_phase_ties"] += len(d["downlink"].get("channel_estimate_phase_ties", []))
    u = d.get("uplink", {})
    tot["uplink"] += u.get("allocations", 0); tot["ul_diff_bits"] += u.get("soft_bits_differing", 0); tot["ul_bits"] += u.get("soft_bits", 0)
    tot["control"] += d.get("control_region", {}).get("subframes", 0); tot["pbch"] += d.get("pbch", {}).get("units", 0)
    tot["prach"] += d.get("prach", {}).get("occasions", 0); tot["sync"] += d.get("sync", {}).get("captures", 0)
print(json.dumps(tot))
open("gpurun_out/fuzz_soak/totals.json", "w").write(json.dumps(tot, indent=1))
PY
