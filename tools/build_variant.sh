#!/bin/bash
# Builds a variant of the library for A/B runs: tools/build_variant.sh <name> "<extra compiler flags>" -> build/lib_<name>.so
# (a private copy of csrc/ so that object files of the shipped build are not disturbed)
set -e
name=$1; extra=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/build/variants/$name
mkdir -p $dst/csrc $root/build/variants/include
cp $root/spin-torque-rl-gym_amd/csrc/*.hip $root/spin-torque-rl-gym_amd/csrc/*.hpp $root/spin-torque-rl-gym_amd/csrc/Makefile $dst/csrc/
# the sources include ../../include/spintorque_hip.h relative to csrc/
cp $root/include/spintorque_hip.h $root/build/variants/include/
make -s -j4 -C $dst/csrc ARCH=gfx950 EXTRA="$extra" OUT=$root/build/lib_$name.so
ls -la $root/build/lib_$name.so
