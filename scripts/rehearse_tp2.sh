# Rehearsal of bench.py's N = 2 control flow on a one-GPU box: two ranks share the GPU, gloo carries the collectives
# (so the numbers mean nothing; what is checked is that every rank-agreement step, the per-rank head split,
# the all-reduce trial children and the final JSON line work; whole-model decode is skipped on this backend).  A rank that is still running after 400 s prints its stack and exits.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4
mkdir -p gpurun_out
ARGS="--gpus 2 --dist-backend gloo --bs 32 --ctx 2048 --steps 4 --warmup 2 --secondary-deadline 400 $*"
RUN='import faulthandler, runpy, sys; faulthandler.dump_traceback_later(400, exit=True); sys.argv = ["bench.py"] + sys.argv[1:]; runpy.run_path("bench.py", run_name="__main__")'
( while sleep 30; do date >> gpurun_out/rehearse_heartbeat.log; done ) &
HB=$!
RANK=1 LOCAL_RANK=1 timeout -k 10 500 python -c "$RUN" $ARGS > gpurun_out/rehearse_r1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 500 python -c "$RUN" $ARGS > gpurun_out/rehearse_r0.json 2> gpurun_out/rehearse_r0.err
RC=$?
wait $P1
R1=$?
kill $HB
echo "rank0 exit $RC, rank1 exit $R1"
tail -c 2500 gpurun_out/rehearse_r0.json
