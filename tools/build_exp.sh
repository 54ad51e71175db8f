#!/bin/bash
# Tuning build: recompile ONE translation unit with extra flags and link it with the regular objects into
# dafs_amd/_exp/libdafs_NAME.so (use with DAFS_HIP_LIB=dafs_amd/_exp/libdafs_NAME.so).  Needs a regular build first.
#   bash tools/build_exp.sh NAME pairhmm3.hip [extra hipcc flags...]
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/dafs_amd/_exp
ff=""
case $src in pairhmm3.hip|pairhmm5.hip) ff="-fno-slp-vectorize";; esac
xflag=""
case $src in *.cpp) xflag="-x hip";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-missing-braces -fno-fast-math $ff "$@" \
  $xflag -c $root/dafs_amd/csrc/$src -o $root/dafs_amd/_exp/${src%.*}_$name.o
objs=$(ls $root/dafs_amd/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/dafs_amd/_exp/libdafs_$name.so $objs $root/dafs_amd/_exp/${src%.*}_$name.o
echo $root/dafs_amd/_exp/libdafs_$name.so
