# the large per-point products of the SplineNets on the bf16 x 3 GEMM (opt-in) against rocBLAS, which picks 40-50 TFLOP/s
# kernels for some segment counts (0.35-0.62 ms launches in 2 of 5 steps): alternating cfg5 lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5u
mkdir -p $O
export PARSENET_PRETRAIN_CACHE=/tmp/pretrain_cache_r5u.pt
for i in 1 2 3; do for V in "0 2" "1 10" "1 2"; do
set -- $V
PARSENET_GEMM_X3=$1 PARSENET_GEMM_X3_MIN_GFLOP=$2 timeout 600 python bench.py --no-cpu-baseline > $O/b_$1_$2_$i.json 2> $O/b_$1_$2_$i.err
python - <<PY
import json
d=json.load(open("$O/b_$1_$2_$i.json"))
print("gemm_x3=$1 min_gflop=$2", round(d["value"],1), round(d["ms_per_step"],2))
PY
done; done
