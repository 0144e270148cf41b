# rocprofv3 evidence for the headline decode kernel.  Pass 1: kernel-trace stats of the python benchmark
# (the command bench.py's number comes from).  Passes 2/3: HBM counters, each in its own run, on the
# standalone driver (same kernel, same shapes, hipMalloc memory).
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_decode
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-e2e --no-configs --no-extend --steps 40 > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 120 ./tools/decode_bench --chunk 4096 --splits 2 --steps 10 > $OUT/driver.log 2>&1 || exit 4
cat $OUT/driver.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- ./tools/decode_bench --chunk 4096 --splits 2 --steps 4 --warmup 1 > $OUT/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- ./tools/decode_bench --chunk 4096 --splits 2 --steps 4 --warmup 1 > $OUT/write.log 2>&1 || exit 3
find $OUT -name "*.csv" | head -20
