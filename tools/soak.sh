set -o pipefail
python -m pytest tests -m gpu -q > gpurun_out/t_final.log 2>&1; rc=$?; tail -2 gpurun_out/t_final.log
test $rc -eq 0 || exit $rc
: > gpurun_out/r04_fuzz.txt
timeout -k 10 260 python tools/fuzz_parity.py 200 61 | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
timeout -k 10 200 python tools/fuzz_parity.py 120 62 batch | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
timeout -k 10 160 python tools/fuzz_matchers.py 100 63 | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
timeout -k 10 260 python tools/fuzz_stereo_bow.py 200 64 | tail -1 >> gpurun_out/r04_fuzz.txt || exit 1
cat gpurun_out/r04_fuzz.txt
