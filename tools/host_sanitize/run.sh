#!/bin/bash
# Host-only, ASan + UBSan instrumented build of csrc/*.hip and the driver that walks the ABI's host code (no GPU needed).
#   tools/host_sanitize/run.sh        -> exit code 0 and "all checks passed" when clean
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=${TMPDIR:-/tmp}/hb_host_san.$$
mkdir -p "$out"
trap 'rm -rf "$out"' EXIT
python3 -c "import sys; sys.path.insert(0, '$root'); from henbun_amd import _build; _build._generate_jit_prelude()"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
for f in runtime elementwise rng variational gram linalg sgp adam comm jit; do
  $HIPCC --offload-host-only $SAN -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -I"$root/include" -c "$root/henbun_amd/csrc/$f.hip" -o "$out/$f.o" &
done
wait
# a host-only object still refers to its device image (__hip_fatbin_<hash>, registered lazily by the HIP runtime and
# only looked at when a kernel is launched -- which nothing here does): empty stand-ins keep the link closed
nm -u "$out"/*.o | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u | awk '{print "extern \"C\" { __attribute__((visibility(\"default\"))) char " $1 "[4096] = {0}; }"}' > "$out/fatbin_stubs.cpp"
$HIPCC --offload-host-only -x c++ -fPIC -c "$out/fatbin_stubs.cpp" -o "$out/fatbin_stubs.o"
$HIPCC $SAN -shared -o "$out/libhb_host_san.so" "$out"/*.o
$HIPCC --offload-host-only $SAN -std=c++17 -x c++ "$root/tools/host_sanitize/driver.cpp" -L"$out" -lhb_host_san -Wl,-rpath,"$out" -o "$out/driver"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$out/driver"
