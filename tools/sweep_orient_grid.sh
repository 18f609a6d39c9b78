python -m pytest tests -x -q -m gpu -k "extract or golden" > gpurun_out/t_og.log 2>&1; tail -3 gpurun_out/t_og.log
for w in kitti tum; do
  for cfg in "0 23" "2 0" "3 0" "4 0" "6 0" "8 0"; do
    set -- $cfg
    ORBFE_ORIENT_GRID=$1 ORBFE_PAD_ORIENT=$2 python bench.py --full-line --no-detail --workload $w --no-e2e --no-cpu-baseline > gpurun_out/b_og.json 2> gpurun_out/b_og.err
    echo "== $w grid=$1 pad=$2"; python tools/show_bench.py gpurun_out/b_og.json | grep -E "value|orient"
  done
done
