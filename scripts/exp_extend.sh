# timing experiments: variant libraries under build/<name>/ against the same driver
for v in BASE NOBARRIER NOSTAGE BOTH; do
  for rep in 1 2; do
    if [ $v = BASE ]; then timeout -k 10 60 ./tools/extend_bench --bs 2 --sq 8192 --steps 20 | cut -c80-140
    else LD_LIBRARY_PATH=build/$v timeout -k 10 60 ./tools/extend_bench --bs 2 --sq 8192 --steps 20 | cut -c80-140; fi
    echo "  ^ $v"
  done
done
