set -o pipefail
: > gpurun_out/r04_fuzz_2.txt
timeout -k 10 330 python tools/fuzz_parity.py 280 81 | tail -1 >> gpurun_out/r04_fuzz_2.txt || exit 1
timeout -k 10 200 python tools/fuzz_parity.py 140 82 batch | tail -1 >> gpurun_out/r04_fuzz_2.txt || exit 1
timeout -k 10 260 python tools/fuzz_matchers.py 200 83 | tail -1 >> gpurun_out/r04_fuzz_2.txt || exit 1
timeout -k 10 300 python tools/fuzz_stereo_bow.py 240 84 | tail -1 >> gpurun_out/r04_fuzz_2.txt || exit 1
cat gpurun_out/r04_fuzz_2.txt
