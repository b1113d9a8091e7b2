#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache.pt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
