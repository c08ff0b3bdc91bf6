#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/overlap_streams_probe.py 1 2 3 1 > gpurun_out/r04_overlap_streams_probe.txt 2>&1; tail -5 gpurun_out/r04_overlap_streams_probe.txt
