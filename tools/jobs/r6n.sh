#!/bin/bash
# round 6, call n: tensor-library operators of one cfg5 step by exact package line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
timeout 900 python tools/probes/op_lines.py > $O/op_lines.txt 2> $O/op_lines.err
wc -l $O/op_lines.txt; tail -3 $O/op_lines.err
