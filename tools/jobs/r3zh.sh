#!/bin/bash
# round 3, job zh: issue-priority variants of the column-pass ping-pong (cycles per tile, in-kernel timers)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3zh
mkdir -p $O
v() { PN_EXTRA_HIPCC_FLAGS="-DMS_TIMING $2" python -m parsenet_codebase_amd.build > $O/build_$1.log 2>&1
      for i in 1 2; do PARSENET_MS_SPARSE=0 timeout 300 python tools/ms_timing.py 2>&1 | grep "PASS 2" | sed "s/^/$1 /"; done; }
v "G1=2,EW=0,G2A=3,G2B=0(default)" ""
v "G1=2,EW=0,G2A=3,G2B=1" "-DX3_P_G2B=1"
v "G1=1,EW=0,G2A=2,G2B=1" "-DX3_P_G1=1 -DX3_P_G2A=2 -DX3_P_G2B=1"
v "G1=0,EW=0,G2A=3,G2B=0" "-DX3_P_G1=0"
v "G1=2,EW=1,G2A=3,G2B=0" "-DX3_P_EW=1"
v "G1=3,EW=0,G2A=3,G2B=0" "-DX3_P_G1=3"
v "G1=2,EW=0,G2A=3,G2B=2" "-DX3_P_G2B=2"
