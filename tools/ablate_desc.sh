#!/bin/bash
# k_desc.hip with the global gathers removed (DESC_ABLATE bit 0: descriptor samples, bit 1: moment rows)
cd "$(dirname "$0")/../orb_slam2_annotate_amd/csrc"
for a in 1 2 3 0; do
  rm -f build/k_desc.o
  if [ "$a" = "0" ]; then make -j8 >/dev/null 2>&1; else make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DDESC_ABLATE=$a" >/dev/null 2>&1; fi
  (cd ../.. && python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate=$a', 'orient_desc_ms', d['roofline']['stage_ms_per_step_exclusive']['orient_desc'])")
done
