#!/bin/bash
# Submit tools/_gpu_job.sh to the GPU pool; when every slot is busy (nothing ran, nothing charged) wait and ask again.
# usage: tools/gpu_submit.sh [timeout_seconds]
T=${1:-900}
for attempt in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- 'bash tools/_gpu_job.sh' > gpurun_out/_call.log 2>&1
  rc=$?
  if grep -q "status=transient" gpurun_out/_call.log; then sleep 150; continue; fi
  break
done
tail -40 gpurun_out/_call.log
exit $rc
