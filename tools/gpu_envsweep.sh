#!/bin/bash
# GPU box: does a HIP-runtime knob move the dependent-launch floor?  headline (graph replay) + plain launches per setting
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() {
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-b1 --no-traffic --steps 6 > gpurun_out/r2o_env.json 2> gpurun_out/r2o_env.err
  python3 -c "import json,sys;d=json.load(open('gpurun_out/r2o_env.json'));print('$1', 'graph', d['ms_per_step'], 'plain', d.get('plain_launches',{}).get('ms_per_step'), 'two_engines', d.get('two_engines',{}).get('value'))" 2>&1 | tail -1
}
run baseline
ROC_SYSTEM_SCOPE_SIGNAL=0 run ROC_SYSTEM_SCOPE_SIGNAL=0
AMD_OPT_FLUSH=0 run AMD_OPT_FLUSH=0
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
HIP_FORCE_DEV_KERNARG=1 run HIP_FORCE_DEV_KERNARG=1
DEBUG_HIP_KERNARG_COPY_OPT=0 run DEBUG_HIP_KERNARG_COPY_OPT=0
ROC_USE_FGS_KERNARG=0 run ROC_USE_FGS_KERNARG=0
GPU_MAX_HW_QUEUES=8 run GPU_MAX_HW_QUEUES=8
run baseline
