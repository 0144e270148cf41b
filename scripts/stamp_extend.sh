#!/bin/bash
# Cycle anatomy of the prefill kernel's steady-state loop: builds the kernel and the standalone driver with -DEXT_STAMPS
# (in-kernel cycle stamps per segment) into tools/ko/ and prints where the cycles go.  Build here, run on the GPU box:
#   scripts/stamp_extend.sh build && gpurun -- scripts/stamp_extend.sh run
set -e
cd "$(dirname "$0")/.."
C=mini-sglang_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
if [ "$1" = build ]; then
    mkdir -p tools/ko
    make -C $C >/dev/null
    /opt/rocm/bin/hipcc $FLAGS -DEXT_STAMPS -c $C/paged_extend.hip -o /tmp/pe_stamps.o
    /opt/rocm/bin/hipcc $FLAGS -DEXT_STAMPS -c $C/paged_extend_wide.hip -o /tmp/pew_stamps.o
    /opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -DEXT_STAMPS -c tools/extend_bench.hip -o /tmp/eb_stamps.o
    OBJS=$(ls $C/*.o | grep -v "paged_extend.o\|paged_extend_wide.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/eb_stamps.o /tmp/pe_stamps.o /tmp/pew_stamps.o $OBJS -o tools/ko/extend_stamps
    # the four-wave kernel with the loop stamped as a whole only: undisturbed cycles per iteration and the clock in the loop
    /opt/rocm/bin/hipcc $FLAGS -DEXT_STAMPS=2 -c $C/paged_extend_wide.hip -o /tmp/pew_stamps2.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/eb_stamps.o /tmp/pe_stamps.o /tmp/pew_stamps2.o $OBJS -o tools/ko/extend_clock
else
    timeout -k 10 120 tools/ko/extend_stamps --bs 2 --sq 16384 --steps 5 --warmup 2
    timeout -k 10 120 tools/ko/extend_clock --bs 2 --sq 16384 --steps 40 --warmup 10
    timeout -k 10 120 tools/ko/extend_stamps --bs 2 --sq 16384 --steps 5 --warmup 2 --wide 0
    timeout -k 10 120 tools/ko/extend_stamps --bs 2 --sq 16384 --steps 5 --warmup 2 --heads 1
fi
