# work-group shape experiments on small grids (MSGL_EXTEND_NH overrides the heuristic)
for args in "--bs 1 --sq 4096 --hq 8 --hkv 1 --ps 64" "--bs 1 --sq 2048 --hq 32 --hkv 8" "--bs 4 --sq 1024 --hq 32 --hkv 8" "--bs 2 --sq 8192 --hq 32 --hkv 8"; do
  for nh in 1 2; do
    MSGL_EXTEND_NH=$nh timeout -k 10 60 ./tools/extend_bench $args --steps 20 | cut -c1-130; echo "  ^ NH=$nh"
  done
done
