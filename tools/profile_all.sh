#!/bin/bash
# All rocprofv3 passes of the four bench workloads (tools/profile_frame.py) and the HIP API trace of a stop-motion loop;
# on the GPU box.  Results under gpurun_out/prof_all, to be copied into profiles/.
out=gpurun_out/prof_all
mkdir -p $out
for w in cfg1 cfg2 cfg3 cfg4; do
  timeout -k 10 900 python tools/profile_frame.py $w $out trace fp32 lanes wait l2 fetch write || exit 1
done
timeout -k 10 600 python tools/profile_frame.py cfg5 $out trace || exit 1
cd /tmp && export TMPDIR=/tmp
for n in 10 30; do
  timeout -k 10 200 rocprofv3 --hip-trace --stats -d $GRAFT_REPO_ROOT/$out/sm_hip_$n -o sm -f csv -- python3 $GRAFT_REPO_ROOT/tools/stop_motion_time.py 10000 512 $n > $GRAFT_REPO_ROOT/$out/sm_$n.log 2>&1 || exit 1
done
