#!/bin/bash
# The fp64 issue-rate probe with the GPU's clocks sampled beside it: rocm-smi's current sclk every 0.2 s while the probe runs.
# usage (GPU box): scripts/valu_probe_with_clocks.sh > gpurun_out/valu_rate_probe.txt
( for i in $(seq 1 60); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.2; done > gpurun_out/valu_probe_sclk_samples.txt ) &
SAMPLER=$!
timeout -k 10 120 build/probe/valu_rate_probe
rc=$?
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
echo "--- rocm-smi sclk samples taken while the probe ran (distinct lines, count) ---"
sort gpurun_out/valu_probe_sclk_samples.txt | uniq -c | sort -rn | head -8
exit $rc
