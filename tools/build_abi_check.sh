#!/bin/bash
# Builds tools/abi_check.bin (torch-free hardware check of the staged kernels through the C ABI; tools/abi_check.cpp) against the
# in-tree pytorch-gan_amd/csrc/libmigan.so; `host` builds the same program against tests/hipemu's execution model instead
# (checks the harness itself on a machine without a GPU: /tmp/abi_check_host).
set -e
cd "$(dirname "$0")"
if [ "$1" = "host" ]; then
  python -c "import sys; sys.path.insert(0, '../tests'); from hipemu import build_emu; print(build_emu.build())"
  /opt/rocm/lib/llvm/bin/clang++ -O1 -std=c++17 -DABI_CHECK_HOST abi_check.cpp -o /tmp/abi_check_host \
    -L../tests/hipemu/_build -lmigan_emu -Wl,-rpath,"$(pwd)/../tests/hipemu/_build" -pthread
  echo "built /tmp/abi_check_host (run with MIGAN_K7_GRID=8)"
else
  /opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 abi_check.cpp -o abi_check.bin \
    -L../pytorch-gan_amd/csrc -lmigan -Wl,-rpath,'$ORIGIN/../pytorch-gan_amd/csrc'
  echo "built tools/abi_check.bin"
fi
