#!/bin/bash
# Run GPU steps in sequence on the box, each under its own timeout; a step that FAILS lets the next one run, a step
# that is KILLED at its limit (or dies on a signal) ends the chain — no further GPU step after a hang.
#   bash tools/gpu_steps.sh "<secs> <name> <command...>" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for step in "$@"; do
  set -- $step
  secs=$1; name=$2; shift 2
  echo "=== step $name (limit ${secs}s): $*"
  timeout -k 10 $secs bash -c "$*" > gpurun_out/$name.log 2>&1
  rc=$?
  echo "=== step $name exit $rc"; tail -n 12 gpurun_out/$name.log
  if [ $rc -ge 124 ]; then echo "=== step $name was killed: stopping the chain"; exit $rc; fi
done
exit 0
