#!/bin/bash
# round 6, call m: census of the device copies / fills of one cfg5 step by issuing site
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout 900 python tools/probes/copy_census.py > $O/copy_census.txt 2> $O/copy_census.err
wc -l $O/copy_census.txt; tail -3 $O/copy_census.err
