#!/bin/bash
# Time line of the matrix-core decode kernel's work groups (entry, page table, first tile, loop end, wave merge, exit) from
# in-kernel stamps of the chip-wide 100 MHz counter: builds kernel + driver with -DDEC_STAMPS into tools/ko/ (build, here),
# prints the distribution over the work groups of one launch (run, on the GPU box).
set -e
cd "$(dirname "$0")/.."
C=mini-sglang_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
if [ "$1" = build ]; then
    mkdir -p tools/ko
    make -C $C >/dev/null
    /opt/rocm/bin/hipcc $FLAGS -DDEC_STAMPS -c $C/paged_decode_mfma.hip -o /tmp/pdm_stamps.o
    /opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -DDEC_STAMPS -c tools/decode_bench.hip -o /tmp/db_stamps.o
    OBJS=$(ls $C/*.o | grep -v "paged_decode_mfma.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/db_stamps.o /tmp/pdm_stamps.o $OBJS -o tools/ko/decode_stamps
else
    S="--hq 8 --hkv 1 --ctx 131072 --ps 64 --layers 8 --steps 40 --warmup 5"
    for cfg in "--bs 1 --chunk 512 --splits 256 --depth 4" "--bs 1 --chunk 512 --splits 256 --depth 2" "--bs 1 --chunk 512 --splits 256 --waves 8" \
               "--bs 8 --chunk 4096 --splits 32 --depth 4" "--bs 8 --chunk 4096 --splits 32 --depth 2"; do
        echo "== $cfg"
        timeout -k 10 100 tools/ko/decode_stamps $S $cfg
    done
fi
