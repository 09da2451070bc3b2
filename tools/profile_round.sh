#!/bin/bash
# Profiling recipe of a round (run on the GPU box from the repo root):  bash tools/profile_round.sh r02
# Writes everything under gpurun_out/<tag>_*; condense afterwards with  python tools/collect_profiles.py <tag>.
# --pmc passes are separate runs with --kernel-trace only (never combined with sys/hip traces); the program after
# `--` is python3 itself (no env/bash hop).
set -eo pipefail
tag=${1:-r02}
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
# the driver's form of the bench line (steps 20, warm-up 5) and the default form
python3 bench.py --steps 20 --warmup 5 > $o/${tag}_bench_driver.json 2> $o/${tag}_bench_driver.err
python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-empirical --no-secondary > $o/${tag}_trace.log 2>&1
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_nf4dq_ffn -- python3 bench.py --workload nf4dq_ffn --no-cpu-baseline --steps 50 > $o/${tag}_nf4dq_ffn.json 2> $o/${tag}_nf4dq_ffn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_int8_4096 -- python3 bench.py --workload int8_4096 --no-cpu-baseline --steps 50 > $o/${tag}_int8_4096.json 2> $o/${tag}_int8_4096.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_nf4_m1 -- python3 bench.py --workload nf4_m1 --no-cpu-baseline --steps 20 > $o/${tag}_nf4_m1.json 2> $o/${tag}_nf4_m1.err
echo "workload traces done"
small="--no-cpu-baseline --no-gemv --no-empirical --no-secondary --steps 5 --warmup 3 --reps 2 --prewarm-ms 50"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/${tag}_fetch -- python3 bench.py $small > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/${tag}_write -- python3 bench.py $small > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $o/${tag}_tcc -- python3 bench.py $small > /dev/null 2>&1
echo "traffic passes done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/${tag}_sq -- python3 bench.py $small > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $o/${tag}_lds -- python3 bench.py $small > /dev/null 2>&1
echo "sq passes done"
# HBM traffic of the M = 1 GEMV (64 rotating layers), of the int8 GEMM and of OutlierAwareLinear's GEMM: FETCH_SIZE and WRITE_SIZE in separate passes
for wl in nf4_m1 int8_4096 nf4dq_ffn outlier; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/${tag}_${wl}_fetch -- python3 bench.py --workload $wl $small > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/${tag}_${wl}_write -- python3 bench.py --workload $wl $small > /dev/null 2>&1
done
echo "gemv / int8 traffic passes done"
