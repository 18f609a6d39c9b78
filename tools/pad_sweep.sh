run() { echo "== $*"; env "$@" python bench.py --full-line --no-detail --workload tum --no-cpu-baseline --no-e2e --input-cache /tmp/ic 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=j['roofline']; print(round(j['value']), round(j['ms_per_step'],3), {s:round(v['ms_per_step_exclusive'],2) for s,v in rf['stages'].items()})"; }
run A=1
run ORBFE_PAD_RESIZE=40
run ORBFE_PAD_OCTREE=28
run ORBFE_PAD_RESIZE=40 ORBFE_PAD_OCTREE=28
run ORBFE_PAD_RESIZE=53 ORBFE_PAD_OCTREE=41
run ORBFE_PAD_RESIZE=40 ORBFE_PAD_OCTREE=28 ORBFE_PAD_ORIENT=40
run ORBFE_PAD_RESIZE=40 ORBFE_PAD_OCTREE=28 ORBFE_PAD_ORIENT=10
run ORBFE_PAD_RESIZE=20 ORBFE_PAD_OCTREE=15
run ORBFE_PAD_BLUR=16
run A=1
