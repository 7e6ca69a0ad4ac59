#!/bin/bash
# Build container: patch the device ISA of the rebuilt FIRST CUT of ffn_fused.hip and link a library variant around it.
# Preparation (see README.md): /tmp/hz/fc_A.hip = csrc/ffn_fused.hip of commit ceed6dd..HEAD~ with the two fences removed and absolute includes;
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip -S --cuda-device-only fc_A.hip -o fc_A.s
# usage: patch_build.sh NAME 'python-expr transforming the text s'  -> tools/probes/mfma_hazard/build/libptts_fc_NAME.so (PTTS_LIB_PATH)
set -e
L=/opt/rocm/lib/llvm/bin
name=$1
python3 - "$name" "$2" <<'PY'
import sys,re
name,expr=sys.argv[1],sys.argv[2]
s=open('/tmp/hz/fc_A.s').read()
s=eval(expr)
open(f'/tmp/hz/fc_{name}.s','w').write(s)
PY
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c fc_$name.s -o dev_$name.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o dev_$name.out dev_$name.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev_$name.out -output=fc_$name.hipfb
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wno-unused-result --offload-arch=gfx950 --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang fc_$name.hipfb -x hip -c fc_A.hip -o ffn_$name.o 2>&1 | grep -i "error" || true
OBJ=$(ls /root/repo/go-pocket-tts_amd/build/*.o | grep -v "ffn_fused\|gemv\|capi_hooks")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,-soname,libptts_hip.so -o /root/repo/tools/probes/mfma_hazard/build/libptts_fc_$name.so $OBJ ffn_$name.o -ldl
echo built $name
