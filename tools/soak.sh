set -o pipefail
# the full GPU suite, then ~11 minutes of differential fuzzing against the oracle; every step appends to gpurun_out/ so the call is never silent
python -m pytest tests -m gpu -q > gpurun_out/t_final.log 2>&1; rc=$?; tail -2 gpurun_out/t_final.log
test $rc -eq 0 || exit $rc
: > gpurun_out/r04_fuzz.txt
timeout -k 10 260 python tools/fuzz_parity.py 200 61 | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
timeout -k 10 200 python tools/fuzz_parity.py 120 62 batch | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
timeout -k 10 200 python tools/fuzz_matchers.py 140 63 | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
timeout -k 10 260 python tools/fuzz_stereo_bow.py 200 64 | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
cat gpurun_out/r04_fuzz.txt
python tools/matcher_latency.py > gpurun_out/r04_matcher_latency.txt 2>&1 || exit 1
python tools/claim_probe.py 100 > gpurun_out/r04_claim_probe.txt 2>&1 || exit 1
{ python tools/single_frame_latency.py 300 kitti pinned; python tools/single_frame_latency.py 300 vga pinned; python tools/single_frame_latency.py 300 kitti stages; python tools/single_frame_latency.py 300 vga stages; } > gpurun_out/r04_single_frame_latency.txt 2>&1 || exit 1
tail -3 gpurun_out/r04_single_frame_latency.txt
