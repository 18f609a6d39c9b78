# A/B of the persistent-grid cap of k_octree ($ORBFE_OCTREE_GRID workgroups per CU, 0 = one workgroup per item)
python -m pytest tests -x -q -m gpu -k "extract or golden or octree" > gpurun_out/t_ocg.log 2>&1; tail -3 gpurun_out/t_ocg.log
for w in kitti tum; do
  for g in 0 2 3 4 6; do
    ORBFE_OCTREE_GRID=$g python bench.py --full-line --no-detail --workload $w --no-e2e --no-cpu-baseline > gpurun_out/b_ocg.json 2> gpurun_out/b_ocg.err
    echo "== $w octree grid=$g"; python tools/show_bench.py gpurun_out/b_ocg.json | grep -E "value|octree"
  done
done
