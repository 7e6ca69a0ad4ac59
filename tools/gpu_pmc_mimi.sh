#!/bin/bash
# GPU box: SQ / MFMA / HBM counters of one batch-64 x 125-frame Mimi decode (tools/pmc_mimi.py), one rocprofv3 pass per counter set
# (counter passes carry --kernel-trace only).  usage: tools/gpu_pmc_mimi.sh TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1
run() {  # name counters...
  local name=$1; shift
  mkdir -p gpurun_out/${tag}_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${tag}_$name -o pmc -- python3 tools/pmc_mimi.py > gpurun_out/${tag}_$name.log 2>&1 || { echo "$name failed"; tail -3 gpurun_out/${tag}_$name.log; return 1; }
  python3 tools/pmc_summary.py $(ls gpurun_out/${tag}_$name/*counter_collection.csv | head -1) "" > gpurun_out/${tag}_$name.txt 2>&1
  rm -f gpurun_out/${tag}_$name/*kernel_trace.csv
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS &&
run lds SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
for n in sq lds fetch write; do echo "## $n"; grep -A1 "k_gemm5\|k_resblock\|k_gemm_wres\|k_attn_window\|k_layernorm\|k_mimi" gpurun_out/${tag}_$n.txt | cut -c1-260 | head -60; done
