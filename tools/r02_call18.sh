#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step_cfg3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload cfg3_cartpole_32k_x_32k --envs 4096 --no-cpu-baseline --no-learn --no-other-configs --steps 3 --warmup 1 --env-steps 4"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/run1.json 2> $OUT/run1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/run2.json 2> $OUT/run2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/run3.json 2> $OUT/run3.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/tcc -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/run4.json 2> $OUT/run4.err
echo done
