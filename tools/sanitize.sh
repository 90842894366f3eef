#!/bin/bash
# CPU-side sanitizer pass (GPU AddressSanitizer is not available on this pool): the oracle (gcc) and the product's host
# code (hipcc, device instrumentation off) built with ASan + UBSan, then the CPU tests that exercise them - loader incl.
# the 300-file differential fuzz, the PBRT parser incl. its 300-scene fuzz, BVH builder, camera, tiling, PNG writer,
# solver/guided/view oracle tests, the fast tree's builder and host walk.
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/ptmi_sanitize; mkdir -p $OUT
gcc -O1 -g -std=gnu11 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -march=x86-64-v2 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o $OUT/libptmi_oracle_asan.so oracle/ptmi_oracle.c -lm
for f in csrc/kernels.hip csrc/radiosity.hip csrc/dist.hip csrc/c_api.cpp host/application_state.cpp host/bvh.cpp host/file_manager.cpp host/pbrt_loader.cpp host/wide_bvh.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
      -fno-slp-vectorize -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -c cuda-pathtracer_amd/$f -o $OUT/$(basename $f).o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -shared -fPIC -o $OUT/libptmi_asan.so $OUT/*.o -ldl
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 OMP_NUM_THREADS=4
echo "== oracle under gcc ASan/UBSan"
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) PTMI_ORACLE_LIB=$OUT/libptmi_oracle_asan.so \
  python -m pytest tests/test_oracle_known_answers.py tests/test_oracle_vs_ref.py tests/test_radiosity_solver.py tests/test_radiosity_view.py \
                   tests/test_guided_oracle.py tests/test_numerics_contract.py tests/test_host_logic.py -x -q -m "not gpu"
echo "== product host code under clang ASan/UBSan"
LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) PTMI_LIB=$OUT/libptmi_asan.so \
  python -m pytest tests/test_host_logic.py tests/test_pbrt_loader.py tests/test_fast_tree.py -x -q -m "not gpu"
