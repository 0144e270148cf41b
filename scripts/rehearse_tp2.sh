# bench.py's N = 2 control flow on ONE GPU over gloo (what tests/test_xgmi_allreduce_gpu.py runs), with per-rank logs.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 MSGL_BENCH_TRACE=1
ARGS="--gpus 2 --dist-backend gloo --steps 3 --warmup 1 --bs 16 --ctx 1024 --layers 2 --e2e-steps 2 --e2e-ctx 512 --mixed-bs 16 --secondary-deadline 400 ${EXTRA:-}"
RANK=1 LOCAL_RANK=1 timeout -k 10 500 python -u bench.py $ARGS > gpurun_out/rehearse_r1.log 2>&1 &
RANK=0 LOCAL_RANK=0 timeout -k 10 500 python -u bench.py $ARGS > gpurun_out/rehearse_r0.log 2>&1
rc=$?
wait
echo "rank 0 exit $rc"
tail -c 3000 gpurun_out/rehearse_r0.log
