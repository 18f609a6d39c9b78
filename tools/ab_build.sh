#!/bin/bash
# Builds a second copy of the library with extra compiler flags for same-box A/B runs of compile-time variants:
#   tools/ab_build.sh exp -DORBFE_RESIZE_WAVES=8      -> orb_slam2_annotate_amd/liborbfe_exp.so
#   ORBFE_LIB=$PWD/orb_slam2_annotate_amd/liborbfe_exp.so python bench.py --full-line --no-detail ...
set -e
TAG=$1; shift
cd "$(dirname "$0")/../orb_slam2_annotate_amd/csrc"
make -j8 OBJDIR=build_$TAG LIB=../liborbfe_$TAG.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fdenormal-fp-math=ieee -Wall -Wno-unused-result $*"
