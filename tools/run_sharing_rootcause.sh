#!/bin/bash
# one two-process run of tools/sharing_probe_rootcause.py on the one GPU of the box; logs under gpurun_out/share/
set -u
mkdir -p gpurun_out/share
rm -rf /tmp/ac_share_probe
timeout -k 10 420 python tools/sharing_probe_rootcause.py aggressor > gpurun_out/share/aggressor.log 2>&1 &
A=$!
timeout -k 10 420 python tools/sharing_probe_rootcause.py victim > gpurun_out/share/victim.log 2>&1
V=$?
wait $A
echo "victim rc=$V aggressor rc=$?"
tail -n 60 gpurun_out/share/victim.log
tail -n 5 gpurun_out/share/aggressor.log
