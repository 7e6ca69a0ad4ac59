#!/bin/bash
# Builds the ablation library: every decoder product with the hi half of its activation split only (-DPTTS_ABLATE_LO, csrc/device_util.h) -> tools/probes/hi_only/build/libptts_hip.so
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../../.." && pwd); src=$root/go-pocket-tts_amd
mkdir -p "$here/build/obj"
objs=""
for f in $(sed -n 's/^SRC *= *//p' "$src/Makefile"); do
  o="$here/build/obj/$(basename $f).o"; extra=""
  case $f in *skinny.hip|*attn_step.hip) extra="-mllvm -amdgpu-kernarg-preload-count=10";; *ffn_fused.hip) extra="-fno-slp-vectorize";; esac
  if [ ! -f "$o" ] || [ "$src/$f" -nt "$o" ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wno-unused-result --offload-arch=gfx950 -DPTTS_ABLATE_LO $extra -x hip -c "$src/$f" -o "$o" & fi
  objs="$objs $o"
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,-soname,libptts_hip.so -o "$here/build/libptts_hip.so" $objs -ldl
echo "built $here/build/libptts_hip.so"
