# same-box A/B of two builds of the library: tools/ab_lib.sh path/to/liborbfe_A.so [workloads]   (B = the in-tree build)
A=$1; WL=${2:-"kitti tum euroc"}
for rep in 1 2; do
  for lib in "$A" ""; do
    for w in $WL; do
      ORBFE_LIB=$lib python bench.py --workload $w --no-e2e --no-cpu-baseline > gpurun_out/b_abl.json 2> gpurun_out/b_abl.err
      echo "[${lib:-in-tree}] $(python tools/show_bench.py gpurun_out/b_abl.json | grep -E 'value|pyramid|fast|orient' | tr '\n' ' ' | sed -E 's/ +/ /g' | sed -E 's/hbm_excl [0-9.]+ valu_frac [0-9.]+ busy [0-9.e+-]+//g' | cut -c1-260)"
    done
  done
done
