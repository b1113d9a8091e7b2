#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
for S in "0,0" "2,2" "3,3" "4,4" "6,6" "3,4" "4,3"; do
  if [ "$S" = "0,0" ]; then unset PN_MS_SLICES; else export PN_MS_SLICES=$S; fi
  echo "PN_MS_SLICES=$S" >> $O/slices.log
  timeout 300 python tools/kbench.py meanshift_batch >> $O/slices.log 2>&1
done
unset PN_MS_SLICES
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 600 python bench.py --workload cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $O/b.json 2>$O/b.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/step_gaps.py $O/tr/b_kernel_trace.csv 3 > $O/gaps.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
grep -v amdgpu.ids $O/slices.log; cat $O/gaps.txt
