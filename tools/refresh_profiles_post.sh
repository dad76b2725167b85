#!/bin/bash
# Build-container side of tools/refresh_profiles.sh: copy the judged summaries into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/refresh
R=${LH_ROUND:-r06}
cp $O/bench.json profiles/${R}_bench.json
cp "$(find $O/stats -name "${R}_kernel_stats.csv" | head -1)" profiles/${R}_bench_kernel_stats.csv
cp $O/layers.txt profiles/${R}_layers.txt
python tools/pmc_traffic.py "$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)" \
    "$(find $O/pmc_write -name '*counter_collection.csv' | head -1)" profiles/${R}_pmc_hbm_traffic.txt profiles/${R}_pmc_traffic.json
python tools/pmc_mfma_step.py "$(find $O/pmc_mfma -name '*counter_collection.csv' | head -1)" profiles/${R}_pmc_mfma_step.txt > /dev/null
cp "$(find $O/stats_hrnet -name '*kernel_stats.csv' | head -1)" profiles/${R}_hrnet_w32_bs32_kernel_stats.csv
cp $O/layers_hrnet.txt profiles/${R}_hrnet_w32_bs32_layers.txt
python tools/step_timeline.py "$(find $O/stats -name "${R}_kernel_trace.csv" | head -1)" profiles/${R}_step_timeline.txt
python tools/step_timeline.py "$(find $O/stats_hrnet -name '*kernel_trace.csv' | head -1)" profiles/${R}_hrnet_w32_bs32_step_timeline.txt
cp "$(find $O/stats_c5 -name '*kernel_stats.csv' | head -1)" profiles/${R}_c5_infer384_kernel_stats.csv
cp $O/layers_c5.txt profiles/${R}_c5_infer384_layers.txt
python tools/pmc_traffic.py "$(find $O/pmc_fetch_c5 -name '*counter_collection.csv' | head -1)" \
    "$(find $O/pmc_write_c5 -name '*counter_collection.csv' | head -1)" profiles/${R}_c5_pmc_hbm_traffic.txt /tmp/c5_traffic.json
ls -la profiles/
# the bench line was printed before the PMC passes of the same session existed: fill its roofline.traffic from them
python - <<'PY'
import json, os
R = os.environ.get("LH_ROUND", "r06")
b = json.load(open(f"profiles/{R}_bench.json"))
pmc = json.load(open(f"profiles/{R}_pmc_traffic.json"))
k = b.get("roofline", {}).get("kernel")
if k in pmc:                     # always from the PMC passes of THIS session (the bench line itself carries the previous committed profile's)
    b["roofline"]["traffic"] = pmc[k]["read_bytes_per_launch"] + pmc[k]["write_bytes_per_launch"]
    b["roofline"]["traffic_note"] = ("bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same session (separate passes, corrected per the "
                                     f"MI355X guide), profiles/{R}_pmc_hbm_traffic.txt")
    b["roofline"]["traffic_source"] = f"PMC passes of the same session (profiles/{R}_pmc_traffic.json)"
    print("roofline.traffic <-", b["roofline"]["traffic"])
st = pmc.get("__train_step__")
if st:
    alg = b["roofline"]["step_traffic"]["algorithmic_bytes"] if b["roofline"].get("step_traffic") else 12064000000
    b["roofline"]["step_traffic"] = {"bytes": st["bytes"], "read_bytes": st["read_bytes"], "write_bytes": st["write_bytes"], "algorithmic_bytes": alg,
                                     "over_algorithmic": round(st["bytes"] / alg, 3),
                                     "whole_process_bytes_per_step": st.get("whole_process_bytes_per_step"),
                                     "source": f"PMC passes of the same session (profiles/{R}_pmc_traffic.json: `bench.py --train-only`, {st['steps']} training steps behind the first Adam launch)"}
    print("roofline.step_traffic <-", b["roofline"]["step_traffic"]["bytes"], b["roofline"]["step_traffic"]["over_algorithmic"])
json.dump(b, open(f"profiles/{R}_bench.json", "w"))
PY
