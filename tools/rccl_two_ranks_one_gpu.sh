#!/bin/bash
# tools/rccl_two_ranks_one_gpu.sh NAME: does RCCL accept TWO ranks on ONE device on this box?  (tests/_dist_worker.py gpu-rccl with world size 2,
# both ranks on GPU 0, each under its own `timeout`.)  If it does, the N > 1 collective path (ncclCommInitRank with a broadcast id, the in-place
# all-gather of segment partials, the ordered segment sum) runs for real on a one-GPU box; if not, the refusal is recorded.
out=gpurun_out/$1; mkdir -p "$out"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=0 LOCAL_RANK=0 timeout -k 10 150 python3 tests/_dist_worker.py gpu-rccl "$out" > "$out/rank0.log" 2>&1 &
p0=$!
RANK=1 LOCAL_RANK=0 timeout -k 10 150 python3 tests/_dist_worker.py gpu-rccl "$out" > "$out/rank1.log" 2>&1 &
p1=$!
wait $p0; r0=$?
wait $p1; r1=$?
echo "rank 0 exit $r0, rank 1 exit $r1"
tail -6 "$out/rank0.log"; tail -6 "$out/rank1.log"
ls "$out"
