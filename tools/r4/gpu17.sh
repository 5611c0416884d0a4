#!/bin/bash
# round 4: the differential suite over 12 further seeds on the round's kernels (SISO forward pass, transform, BCJR addressing, DPP reductions)
cd /root/repo
rm -rf gpurun_out/fuzz_soak; mkdir -p gpurun_out/fuzz_soak
bash tools/r3/fuzz_soak.sh 401 412 2>&1 | tail -16
mkdir -p gpurun_out/fuzz_soak_r04; cp gpurun_out/fuzz_soak/summary.txt gpurun_out/fuzz_soak/totals.json gpurun_out/fuzz_soak_r04/ 2>/dev/null
python - <<'PY'
import glob, json
vals = []
for f in sorted(glob.glob("gpurun_out/fuzz_soak/report_seed_*.json")):
    u = json.load(open(f)).get("uplink", {})
    vals += u.get("differing_soft_bits_position_library_reference_and_the_symbols_other_bit", [])
json.dump(vals, open("gpurun_out/fuzz_soak_r04/uplink_differing_soft_bits.json", "w"), indent=1)
print(len(vals), "differing uplink soft bits over the soak:", vals[:6])
PY
