#!/bin/bash
# device assembly of csrc/pointwise.hip -> base.s ; every *.s in this directory -> *.hsaco (code objects for tools/stale_read/asm_delta.py)
set -e
cd "$(dirname "$0")"
LLVM=/opt/rocm/lib/llvm/bin
if [ ! -f base.s ] || [ "$1" = "--regen" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I../../../vi_depth_completion_amd/csrc -o base.s ../../../vi_depth_completion_amd/csrc/pointwise.hip 2>/dev/null
fi
for f in *.s; do
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$f" -o "${f%.s}.o"
  $LLVM/ld.lld -shared "${f%.s}.o" -o "${f%.s}.hsaco"
  rm -f "${f%.s}.o"
done
ls -la *.hsaco
