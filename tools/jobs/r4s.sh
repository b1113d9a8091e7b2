cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
export PARSENET_PRETRAIN_CACHE=/tmp/pt_cache.pt
python tools/xproc_probe.py 1 > /dev/null 2>&1
for i in 1 2 3; do python tools/xproc_probe.py 6 2>/dev/null | grep "^step" > gpurun_out/r4s/run_$i.txt; done
diff gpurun_out/r4s/run_1.txt gpurun_out/r4s/run_2.txt && echo "1 == 2"
diff gpurun_out/r4s/run_1.txt gpurun_out/r4s/run_3.txt && echo "1 == 3"
cat gpurun_out/r4s/run_1.txt | cut -c1-200
