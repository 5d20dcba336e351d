#!/bin/bash
# Usage: tools/pmc_passes.sh OUTDIR PROGRAM [ARGS...]      e.g.  tools/pmc_passes.sh gpurun_out/pmc_c3 python3 tests/prof_mesh.py bunny
# Runs PROGRAM under rocprofv3 once per counter set (counters that do not fit one pass are collected in separate runs,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); --pmc is never combined with a trace domain.  PROGRAM itself follows `--`
# (no env/bash -c hop: the profiler's preloaded library has already initialised the GPU).
set -e
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
sets=(
 "sq1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
 "sq2:SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
 "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "fetch:FETCH_SIZE"
 "write:WRITE_SIZE"
)
for s in "${sets[@]}"; do
    name=${s%%:*}; ctrs=${s#*:}
    echo "pass $name: $ctrs"
    rocprofv3 --pmc $ctrs --output-format csv -d "$out/$name" -o p -- "$@" > "$out/$name.log" 2>&1 || echo "pass $name FAILED (see $out/$name.log)"
done
