#!/bin/bash
# usage: build_variant.sh <tree> <name> <patch-mode>   -> <tree>/jittor-myc-nerfs_amd/lib/libtvr_<name>.so
set -e
T=$1; name=$2; mode=$3
C=$T/jittor-myc-nerfs_amd/csrc
W=/tmp/isa_patch_w_$name; rm -rf $W; mkdir -p $W; cd $W
LL=/opt/rocm/lib/llvm/bin
FL="-O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Wno-unused-function"
/opt/rocm/bin/hipcc $FL --save-temps -c $C/tvr_shade.hip -o shade_orig.o >/dev/null 2>&1
S=tvr_shade-hip-amdgcn-amd-amdhsa-gfx950.s
python3 $(dirname $0)/patch_isa.py $S patched.s $mode
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c patched.s -o dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o dev.out dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev.out -output=dev.hipfb
/opt/rocm/bin/hipcc $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang dev.hipfb -c $C/tvr_shade.hip -o shade.o
mkdir -p $T/jittor-myc-nerfs_amd/lib
OBJS=""
for f in $(sed -n 's/^SRCS *:= *//p' $C/Makefile); do b=${f%.hip}; if [ $b != tvr_shade ]; then [ -f $T/jittor-myc-nerfs_amd/lib/obj/$b.o ] || (mkdir -p $T/jittor-myc-nerfs_amd/lib/obj && /opt/rocm/bin/hipcc $FL -c $C/$f -o $T/jittor-myc-nerfs_amd/lib/obj/$b.o); OBJS="$OBJS $T/jittor-myc-nerfs_amd/lib/obj/$b.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/jittor-myc-nerfs_amd/lib/libtvr_$name.so shade.o $OBJS
echo built $name: $(grep -c v_mfma patched.s) mfma, $(grep -c "s_nop 15" patched.s) nop15
