# same-box A/B of environment knobs over the three workloads: tools/ab_env.sh "A=1 B=2" "A=0" ...   (each argument = one configuration)
for cfg in "$@"; do
  for w in kitti tum euroc; do
    env $cfg python bench.py --workload $w --no-e2e --no-cpu-baseline > gpurun_out/b_ab.json 2> gpurun_out/b_ab.err
    echo "[$cfg] $(python tools/show_bench.py gpurun_out/b_ab.json | grep -E 'value' | sed -E 's/.*(KITTI|TUM|EuRoC|tum|euroc).* value ([0-9]+) .*ms\/step ([0-9.]+).*/\1 \2 \3/')"
  done
done
