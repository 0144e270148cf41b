# A/B two builds of libmsgl_hip.so on the shape sweep: the in-tree one and exp/libmsgl_nt.so
mkdir -p /tmp/nt && cp exp/libmsgl_nt.so /tmp/nt/libmsgl_hip.so
echo "== default"; timeout -k 10 400 python -u scripts/sweep_shapes.py cfg2 cfg3ii tp8 cfg5 2>&1 | grep -v amdgpu.ids | cut -c1-75,150-260
echo "== nt"; MSGL_HIP_LIB=/tmp/nt/libmsgl_hip.so timeout -k 10 400 python -u scripts/sweep_shapes.py cfg2 cfg3ii tp8 cfg5 2>&1 | grep -v amdgpu.ids | cut -c1-75,150-260
