#!/bin/bash
# Experimental build: the kernels' device assembly with VOP2 FP32 mul / add / sub / fmac re-encoded as VOP3 (tools/vop3_rewrite.py) -> librtx_hip_vop3.so
# (select with RTX_LIB_PATH).  Everything else is the product build.
set -e
cd "$(dirname "$0")/../royaltracer-dx_amd"
L=/opt/rocm/lib/llvm/bin; B=build_vop3; mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS --cuda-device-only -S -o $B/k.s csrc/rtx_kernels.hip
python3 ../tools/vop3_rewrite.py $B/k.s $B/k3.s
$L/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/k3.s -o $B/k_dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/k.out $B/k_dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$B/k.out -output=$B/k.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/k.hipfb -c csrc/rtx_kernels.hip -o $B/rtx_kernels.o
OBJS=$(ls build/csrc/*.o build/host/*.o | grep -v rtx_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librtx_hip_vop3.so $B/rtx_kernels.o $OBJS
ls -la librtx_hip_vop3.so
