#!/bin/bash
# The GPU test tier as the driver runs it, with its log kept: gpurun_out/pytest_gpu.log always, and a copy under
# gpurun_out/incidents/ when the run fails or the process dies (tools/keep_incidents.sh then moves it to profiles/incidents/).
# Usage (on the GPU box, from the repo root): tools/gpu_suite.sh [extra pytest arguments]
mkdir -p gpurun_out/incidents
python -X faulthandler -m pytest tests -x -q -m gpu "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then
	cp gpurun_out/pytest_gpu.log "gpurun_out/incidents/pytest_gpu_$(date +%Y%m%d_%H%M%S)_rc$rc.log"
fi
exit $rc
