#!/bin/bash
# A/B timing of variants of the four-wave prefill kernel in one GPU job: every variant is a copy of
# mini-sglang_amd/csrc/paged_extend_wide.hip linked with the standalone driver into tools/ko/ab_<name>.
#   scripts/ab_extend.sh build name1=path/to/variant1.hip [name2=... ...]     (here, on the CPU)
#   gpurun -- scripts/ab_extend.sh run [rounds]                                (on the GPU box; rounds are interleaved)
set -e
cd "$(dirname "$0")/.."
C=mini-sglang_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -I$C"
if [ "$1" = build ]; then
    shift
    mkdir -p tools/ko
    rm -f tools/ko/ab_* tools/ko/lib_*
    make -C $C >/dev/null
    # (every variant carries the two whole-loop stamps of -DEXT_STAMPS=2: cycles per steady-state iteration and the clock)
    /opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -DEXT_STAMPS -c tools/extend_bench.hip -o /tmp/eb_ab.o
    /opt/rocm/bin/hipcc $FLAGS -DEXT_STAMPS=2 -c $C/paged_extend.hip -o /tmp/pe_ab.o
    OBJS="/tmp/pe_ab.o $(ls $C/*.o | grep -v "paged_extend_wide.o\|paged_extend.o")"
    for nv in "$@"; do
        name=${nv%%=*}; src=${nv#*=}; extra=""
        case "$src" in *:*) extra=${src#*:}; src=${src%%:*};; esac
        cp "$src" $C/_ab_variant.hip
        /opt/rocm/bin/hipcc $FLAGS -DEXT_STAMPS=2 $extra -c $C/_ab_variant.hip -o /tmp/ab_$name.o
        rm -f $C/_ab_variant.hip
        /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/eb_ab.o /tmp/ab_$name.o $OBJS -o tools/ko/ab_$name
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ab_$name.o $OBJS -o tools/ko/lib_$name.so  # for MSGL_HIP_LIB=
        echo "built tools/ko/ab_$name"
    done
else
    rounds=${2:-3}
    for r in $(seq $rounds); do
        for b in $(ls tools/ko/ab_*); do
            echo "== $(basename $b) round $r"
            timeout -k 10 120 $b --bs 1 --sq 16384 --steps 10 --warmup 3
            timeout -k 10 120 $b --bs 1 --sq 4096 --prefix 65536 --ps 64 --hq 8 --hkv 1 --steps 10 --warmup 3
        done
    done
fi
