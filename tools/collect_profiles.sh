#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/prof_rNN_cC/ :
#   tools/collect_profiles.sh r02 [config]
# 1. kernel trace + stats of the eager bench run (every launch visible, as bench.py's own HIP-event pass)
# 2. three separate PMC passes (HBM fetch, HBM write, SQ/MFMA cycles) — never combined with other trace domains
# 3. the bench line itself (HIP-graph replay)
tag=${1:-r02}
cfg=${2:-2}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_c${cfg}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
COMMON="--config $cfg --repeats 1 --no-cpu-baseline --no-roofline --no-calibration --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $B --steps 20 --warmup 3 $COMMON > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $out/pmc_sq.log 2>&1
# 2b. (round 6, verdict r05 item 4) the LDS / vector-memory view of the forward / backward machines, three more passes
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_lds -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $out/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $out/pmc_vmem -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $out/pmc_vmem.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $out/pmc_issue -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $out/pmc_issue.log 2>&1
cd $GRAFT_REPO_ROOT && python3 bench.py --config $cfg --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_line.json
find $out -name "*.csv" | head -20
