#!/bin/bash
# one two-process run of tools/sharing_probe_victims.py on the one GPU of the box; logs under gpurun_out/share/
set -u
mkdir -p gpurun_out/share
rm -rf /tmp/ac_share_probe3
timeout -k 10 400 python tools/sharing_probe_victims.py aggressor > gpurun_out/share/victims_aggressor.log 2>&1 &
A=$!
timeout -k 10 400 python tools/sharing_probe_victims.py victim > gpurun_out/share/victims_victim.log 2>&1
V=$?
wait $A
echo "victim rc=$V aggressor rc=$?"
grep -v amdgpu.ids gpurun_out/share/victims_aggressor.log | tail -n 5
grep -v amdgpu.ids gpurun_out/share/victims_victim.log | tail -n 40
