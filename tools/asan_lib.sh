#!/bin/bash
# The library's HOST side under sanitizers (SURVEY.md §5 "race detection / sanitizers").  Two builds of libmmplace with
# the host code instrumented (the device code is untouched: GPU AddressSanitizer is not available on this pool):
#   libmmplace_asan.so   -fsanitize=address,undefined
#   libmmplace_tsan.so   -fsanitize=thread
# and, against each: tools/micro/stress.cc (request threads x committer x registry events x submission threads x resident kernel,
# instrumented itself, so no LD_PRELOAD) and the multi-threaded GPU tests through Python with the sanitizer runtime preloaded.
#   build (here or on the GPU box):  bash tools/asan_lib.sh build
#   run (GPU box):                   bash tools/asan_lib.sh run [seconds]      -> gpurun_out/sanitizers/*.log
set -u
cd "$(dirname "$0")/.."
V=modelmesh_amd/lib/variants
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)")
mkdir -p $V
build() {
  for s in asan tsan; do
    if [ $s = asan ]; then F="-fsanitize=address,undefined"; else F="-fsanitize=thread"; fi
    if [ ! -f $V/libmmplace_$s.so ] || [ modelmesh_amd/csrc/mmplace.hip -nt $V/libmmplace_$s.so ] || [ modelmesh_amd/csrc/place_kernel.hpp -nt $V/libmmplace_$s.so ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC $F -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-function \
        modelmesh_amd/csrc/mmplace.hip -o $V/libmmplace_$s.so -ldl -lpthread || exit 1
    fi
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 $F -fno-gpu-sanitize -fno-omit-frame-pointer -Iinclude tools/micro/stress.cc -L$V -lmmplace_$s \
      -Wl,-rpath,$PWD/$V -lpthread -o $V/stress_$s || exit 1
  done
  echo "built: $(ls $V | tr '\n' ' ')"
}
run() {
  secs=${1:-10}
  OUT=gpurun_out/sanitizers
  mkdir -p $OUT
  export TMPDIR=/tmp
  # the HIP runtime maps memory where the sanitizers' shadow lives unless told not to protect the gap; leaks of the runtime's own are not ours
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:allocator_may_return_null=1
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
  # (races INSIDE libamdhip64 / libhsa-runtime64 are not this library's: the suppressions name them by module)
  export TSAN_OPTIONS="suppressions=$PWD/tools/tsan.supp:halt_on_error=0:second_deadlock_stack=1:report_signal_unsafe=0:history_size=4"
  for s in asan tsan; do
    echo "== stress, $s"; timeout 600 $V/stress_$s $secs 8 > $OUT/stress_$s.log 2>&1; echo "exit $?" >> $OUT/stress_$s.log
    tail -2 $OUT/stress_$s.log; grep -c "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer" $OUT/stress_$s.log
  done
  echo "== hostile arguments, asan"; timeout 600 $V/stress_asan fuzz 20000 > $OUT/fuzz_asan.log 2>&1; echo "exit $?" >> $OUT/fuzz_asan.log
  tail -2 $OUT/fuzz_asan.log; grep -c "ERROR: AddressSanitizer\|runtime error:" $OUT/fuzz_asan.log
  # Through Python, ThreadSanitizer only: under a preloaded AddressSanitizer runtime the HIP runtime does not initialise inside the
  # interpreter on the GPU box (its start-up allocation dies in the sanitizer's allocator; with allocator_may_return_null it faults) —
  # the AddressSanitizer legs are the two C++ programs above.  torch's bundled runtime does not start under the preload either: the library
  # binds /opt/rocm's itself (MMP_NO_TORCH_PRELOAD=1) and the suites that need torch tensors stay with stress.cc.
  export MMP_NO_TORCH_PRELOAD=1
  echo "== pytest, tsan"
  MMP_LIB_PATH=$PWD/$V/libmmplace_tsan.so LD_PRELOAD=$RT/libclang_rt.tsan-x86_64.so timeout 2400 setarch -R python -m pytest -p no:cacheprovider tests/test_resident_gpu.py tests/test_delta_commit_gpu.py tests/test_churn_gpu.py tests/test_registry_upsert_gpu.py tests/test_miss_gpu.py tests/test_route_gpu.py tests/test_cache_replay_gpu.py tests/test_shortlist_memo_gpu.py tests/test_long_memo_gpu.py -m gpu -q > $OUT/pytest_tsan.log 2>&1
  echo "exit $?" >> $OUT/pytest_tsan.log; tail -3 $OUT/pytest_tsan.log; grep -c "WARNING: ThreadSanitizer" $OUT/pytest_tsan.log
}
case ${1:-build} in
  build) build ;;
  run) shift; run "$@" ;;
esac
