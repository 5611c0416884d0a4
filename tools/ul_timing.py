#!/usr/bin/env python3
"""Per-call uplink path: what a subframe of the shim's liblte_phy_get_ul_subframe + one liblte_phy_pusch_channel_decode per UE costs,
GPU build against the all-CPU build of the same caller (shim/dropin_ul_demo.cc with UL_DEMO_REPEAT; 20 MHz, three UEs) -- and the same
subframe through the shim's one-call form (liblte_phy_ul_subframe_decode, UL_DEMO_ONE_CALL)."""
import os, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_dropin_gpu as t
with tempfile.TemporaryDirectory() as d:
    args = t._ul_demo_args(d)
    env = dict(os.environ, UL_DEMO_REPEAT="200", UL_DEMO_ONE_CALL="1")
    for name in ("dropin_ul_gpu", "dropin_ul_gpu", "dropin_ul_gpu_pure", "dropin_ul_cpu"):
        r = subprocess.run([os.path.join(root, "shim", "_build", name)] + args, capture_output=True, text=True, env=env, timeout=600)
        print(name, [l for l in r.stderr.splitlines() if l.startswith("timing")])
