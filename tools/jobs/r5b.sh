# small-k kernel: where does the time go?  KSK_EXP=2: no flushes; 3: no flushes, no LDS appends; 4: MFMAs + one add per row
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5b
mkdir -p $O
for E in 4 3 2; do
  PN_EXTRA_HIPCC_FLAGS="-DKSK_EXP=$E" python -m parsenet_codebase_amd.build > $O/build_$E.txt 2>&1
  PN_KNN_SMALLK=1 python tools/kbench.py smallk 2>&1 | grep "one pass" | sed "s/^/EXP=$E /" | cut -c1-80
done
