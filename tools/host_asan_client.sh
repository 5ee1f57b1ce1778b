#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the C-ABI (device code untouched: GPU sanitizers are not available on this pool).
#   build container:  bash tools/host_asan_client.sh build      -> build/asan/{libspintorque_hip.so, c_client_asan}
#   GPU box (gpurun): bash tools/host_asan_client.sh run        -> examples/c_client.c through reset / step / step_many / state / counters
# The client is linked by ROCm's clang (the library's sanitizer runtime is clang's; gcc's libasan lacks its symbols).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
case "${1:-}" in
build)
  bash $root/tools/build_variant.sh hostasan "-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -g"
  mkdir -p $root/build/asan && cp $root/build/lib_hostasan.so $root/build/asan/libspintorque_hip.so
  /opt/rocm/lib/llvm/bin/clang -std=c99 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -I$root/include $root/examples/c_client.c \
      -o $root/build/asan/c_client_asan -L$root/build/asan -lspintorque_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
  ;;
run)
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
  for c in "rk4 1000 3" "rk45 300 2" "rk4 70000 2" "rk45 5000 1"; do
    timeout -k 10 200 $root/build/asan/c_client_asan $c /tmp/asan_client.bin
  done
  ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
