#!/bin/bash
# Run a list of GPU steps from a file (one command per line, '#' comments), each under its own `timeout -k 10 <limit>`, logging to
# gpurun_out/<tag>/<n>_<name>.log.  A step that fails normally does not stop the list; a step that TIMES OUT or is KILLED does
# (no further GPU step after a hang).  Line format:  <limit_seconds> <name> <command...>
# usage: tools/gpu_steps.sh <tag> <steps-file>
tag=$1; steps=$2
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
n=0
while IFS= read -r line; do
  case "$line" in ''|'#'*) continue;; esac
  limit=${line%% *}; rest=${line#* }; name=${rest%% *}; cmd=${rest#* }
  n=$((n+1)); log=$out/$(printf %02d $n)_$name.log
  echo "[$(date +%H:%M:%S)] step $n $name (limit ${limit}s)"
  timeout -k 10 $limit bash -c "$cmd" > $log 2>&1
  rc=$?
  echo "[$(date +%H:%M:%S)] step $n $name rc=$rc" | tee -a $out/summary.txt
  tail -3 $log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping" | tee -a $out/summary.txt; exit 1; fi
done < $steps
exit 0
