#!/bin/bash
# CPU-only sanitizer pass (GPU ASan / XNACK runs are not available on the pool): AddressSanitizer + UBSan builds of
#   (1) the plain-C oracle (gcc)                       -> tests/test_oracle.py, tests/test_generator.py
#   (2) the HOST side of libpogema_amd.so (hipcc, -fno-gpu-sanitize): C-ABI argument handling, host generator
#       -> tests/test_generator.py, tests/test_abi.py, tests/test_oracle.py through PGX_LIB
# Nothing is installed in-tree: sanitized libraries live in /tmp; the normal oracle .so is restored afterwards.
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fPIC -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -shared \
    -o /tmp/libpogema_oracle_asan.so oracle/pogema_oracle.c
cp oracle/libpogema_oracle.so /tmp/libpogema_oracle_backup.so
trap 'cp /tmp/libpogema_oracle_backup.so oracle/libpogema_oracle.so' EXIT
cp /tmp/libpogema_oracle_asan.so oracle/libpogema_oracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle.py tests/test_generator.py -q -x
cp /tmp/libpogema_oracle_backup.so oracle/libpogema_oracle.so
(cd pogema_amd/csrc && /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address \
    -fno-gpu-sanitize -shared-libsan -x hip pgx_kernels.hip pgx_reset.hip pgx_buffers.hip pgx_nprng.hip pgx_api.cpp -shared \
    -o /tmp/libpogema_amd_asan.so -pthread)
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 PGX_LIB=/tmp/libpogema_amd_asan.so \
    LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1) \
    python -m pytest tests/test_generator.py tests/test_abi.py tests/test_oracle.py tests/test_nprng.py tests/test_npgen.py -q -x
