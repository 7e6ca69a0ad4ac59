#!/bin/bash
# GPU box, round 3: usage tools/gpu_r3.sh TAG STEP...   steps: tests | bench | quick[:lib.so] | stamps[:lib.so] | prof | full
# quick = the headline workload without the extras (phases + mean step-linear launch), optionally on another build of the library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; shift
rc_all=0
for step in "$@"; do
  name=${step%%:*}; lib=""; [[ "$step" == *:* ]] && lib=${step#*:}
  sfx=""; [ -n "$lib" ] && sfx="_$(basename $lib .so)"
  [ -n "$lib" ] && export PTTS_LIB_PATH="$GRAFT_REPO_ROOT/$lib" || unset PTTS_LIB_PATH
  case $name in
    tests) tools/gpu_tests.sh $tag; rc=$?; [ $rc -ne 0 ] && rc_all=$rc ;;
    quick) timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-b1 --no-traffic --no-two-engines --steps 8 > gpurun_out/${tag}_quick$sfx.json 2> gpurun_out/${tag}_quick$sfx.err; echo "quick$sfx rc=$?"
           python3 -c "import json;d=json.load(open('gpurun_out/${tag}_quick$sfx.json'));r=d['roofline'];print('quick$sfx', d['value'],d['ms_per_step'],r['avg_launch_us'],r.get('phases_ms'))" ;;
    quickb1) timeout -k 10 300 python3 bench.py --workload b1_5s_f32 --no-cpu-baseline --no-b1 --no-traffic --no-two-engines --steps 12 --warmup 3 > gpurun_out/${tag}_quickb1$sfx.json 2> gpurun_out/${tag}_quickb1$sfx.err; echo "quickb1$sfx rc=$?"
           python3 -c "import json;d=json.load(open('gpurun_out/${tag}_quickb1$sfx.json'));r=d['roofline'];print('quickb1$sfx', d['value'],d['ms_per_step'],r['avg_launch_us'],r.get('phases_ms'))" ;;
    stamps) timeout -k 10 200 python3 tools/step_stamps.py > gpurun_out/${tag}_stamps$sfx.txt 2>&1; echo "stamps$sfx rc=$?"; grep -c pro= gpurun_out/${tag}_stamps$sfx.txt ;;
    bench) timeout -k 10 900 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 1200 gpurun_out/${tag}_bench.json; echo ;;
    prof) mkdir -p gpurun_out/${tag}_prof$sfx
          timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof$sfx -o prof -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-b1 --no-two-engines --no-traffic > gpurun_out/${tag}_prof$sfx.log 2>&1; echo "rocprof$sfx rc=$?"
          python3 tools/trace_summary.py $(ls gpurun_out/${tag}_prof$sfx/*kernel_trace.csv | head -1) 70 > gpurun_out/${tag}_by_grid$sfx.txt; head -24 gpurun_out/${tag}_by_grid$sfx.txt
          rm -f gpurun_out/${tag}_prof$sfx/*kernel_trace.csv ;;
    realckpt) # the real-checkpoint tests on a stand-in: synthetic full-size file + an oracle-made fixture (plumbing check; pins nothing)
          python3 -c "import sys; sys.path.insert(0,'.'); import ptts_amd; p=ptts_amd.load(); p.synth.write_safetensors('/tmp/full_f32.safetensors', p.synth.make_checkpoint(p.synth.SynthConfig.full(), seed=1234), dtype='F32')" &&
          python3 tools/oracle_fixture.py /tmp/full_f32.safetensors /tmp/fx.json &&
          PTTS_CHECKPOINT=/tmp/full_f32.safetensors POCKETTTS_NATIVE_PY_FIXTURE=/tmp/fx.json timeout -k 10 600 python3 -m pytest tests/test_real_checkpoint.py -q -p no:cacheprovider > gpurun_out/${tag}_realckpt.log 2>&1; echo "realckpt rc=$?"; tail -n 12 gpurun_out/${tag}_realckpt.log ;;
    *) echo "unknown step $step" ;;
  esac
done
exit $rc_all
