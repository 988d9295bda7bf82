#!/bin/bash
# usage: scripts/build_variant.sh <name> [-DTVR_X=1 ...]  ->  jittor-myc-nerfs_amd/lib/variants/libtvr_<name>.so
# A/B and diagnostic builds of the same ABI (select with TVR_LIB_PATH); the default library is untouched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$R/jittor-myc-nerfs_amd/lib/variants
mkdir -p $out/obj_$name
cd $R/jittor-myc-nerfs_amd/csrc
for f in $(sed -n 's/^SRCS *:= *//p' Makefile | sed 's/\.hip//g'); do        # the Makefile's list: a variant library exports every symbol
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off $(sed -n "s/^FLAGS_$f *:= *//p" Makefile) -std=c++17 -Wno-unused-function "$@" -c $f.hip -o $out/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libtvr_$name.so $out/obj_$name/*.o
echo built $out/libtvr_$name.so
