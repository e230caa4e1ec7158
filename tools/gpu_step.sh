#!/bin/bash
# helper for gpurun command lists: run one GPU step under a timeout; a step that had to be killed ends the
# whole call (no further GPU step after a hang), any other failure is logged and the list goes on.
# usage: source tools/gpu_step.sh; step <seconds> <logfile> <command...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
step() {
  local secs=$1 log=$2; shift 2
  echo "[step $(date +%H:%M:%S)] $*"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "[step] rc=$rc ($log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[step] killed at its limit: stopping this call"; exit 1; fi
  return 0
}
