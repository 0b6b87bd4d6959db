#!/bin/bash
# N ranks sharing one GPU through the file communicator: the driver's N-GPU run (BASELINE configs[2],
# strong scaling) end to end on a 1-GPU box; the timings mean nothing (file all-reduce, one GPU)
N=${1:-8}
export ML_COMM_BACKEND=file WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=29577
pids=()
for r in $(seq 1 $((N-1))); do RANK=$r LOCAL_RANK=$r python bench.py --gpus $N --steps 3 --warmup 1 --cpu-rows 0 > gpurun_out/rank$r.out 2> gpurun_out/rank$r.err & pids+=($!); done
RANK=0 LOCAL_RANK=0 python bench.py --gpus $N --steps 3 --warmup 1 --cpu-rows 0 2> gpurun_out/rank0.err | cut -c1-900
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done; echo "other ranks rc=$rc"; tail -2 gpurun_out/rank1.err
