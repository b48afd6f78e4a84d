#!/bin/bash
# Runs ON the GPU box (through gpurun): kernel trace + the PMC passes of one bench.py workload.
#   profiles/tools/profile_gpu.sh <out_dir under gpurun_out> [bench.py arguments...]
# Counters are collected in separate rocprofv3 runs (--pmc only; never combined with trace domains),
# FETCH_SIZE and WRITE_SIZE in different passes (TCC slots), SQ counters eight at a time.
set -u
OUT=$1; shift
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-variants $*"
run() {  # name, rocprofv3 options...
  local name=$1; shift
  ( cd /tmp && rocprofv3 "$@" -d "$OUT" -o "$name" --output-format csv -- python "$REPO/bench.py" $ARGS ) > "$OUT/$name.log" 2>&1
}
# PASSES (default: all) selects the runs; STEPS_TRACE overrides bench.py's default steps for the traced run
PASSES=${PASSES:-"trace fetch write sq1 sq2 sq3 grbm"}
want() { case " $PASSES " in *" $1 "*) return 0;; esac; return 1; }
ARGS="--no-cpu-baseline --no-variants $* ${STEPS_TRACE:-}"
want trace && run trace --kernel-trace --stats
STEPS="--steps 2 --warmup 1"
ARGS="--no-cpu-baseline --no-variants $* $STEPS"
want fetch && run fetch --pmc FETCH_SIZE
want write && run write --pmc WRITE_SIZE
want sq1 && run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
want sq2 && run sq2 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM
want sq3 && run sq3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT
want grbm && run grbm --pmc GRBM_GUI_ACTIVE
ls -la "$OUT" | head -50
