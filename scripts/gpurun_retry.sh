#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 / "transient"): scripts/gpurun_retry.sh <timeout> <script>
for i in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun --timeout $1 -- "bash $2" 2>&1)
  echo "$out" | tail -60
  if ! echo "$out" | grep -q "status=transient"; then exit 0; fi
  sleep 60
done
