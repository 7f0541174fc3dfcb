#!/bin/bash
# PMC pass (own run, kernel-trace only): SQ instruction/cycle counters per kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
export ULCX_ASYNC_FB=0
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu --streams ${1:-1024} --blocks 16 > gpurun_out/pmc_bench.json 2> gpurun_out/pmc.err
ls gpurun_out/pmc/*/ | head
cp $(find gpurun_out/pmc -name "*counter_collection.csv" | head -1) gpurun_out/pmc_counters.csv
