#!/bin/bash
# fault_build.sh -- cordic_amd/lib_fault.so: the library with ONE ordering
# edge removed (cordic_group.cpp: the wait of a job's kernels for the previous
# job's forwarded pieces, -DCORDIC_FAULT_SKIP_JOB_ORDER).  Test infrastructure:
# tests/test_group.py::test_async_shim_catches_a_missing_job_order runs the
# back-to-back jobs over the asynchronous RCCL stand-in against this build and
# expects WRONG gathered data -- the proof that the stand-in can see what the
# round-3 one (exchange complete inside ncclGroupEnd) could not.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/cordic_amd/csrc"
# only cordic_group.cpp differs, and it sits on the public C ABI (plus one
# exported probe launcher): the fault library is that ONE object with the
# product library as its dependency ($ORIGIN), ~60 KB instead of a second copy
# of every kernel.  dlsym() on its handle finds cordic_group_* here first and
# everything else in libcordic_amd.so.
[ -f "$ROOT/cordic_amd/libcordic_amd.so" ] || { echo "no product library: lib_fault.so not rebuilt"; exit 0; }
mkdir -p build_fault
rm -f build_fault/*.o
g++ -O3 -std=c++17 -fPIC -fwrapv -Wall -Wno-unused-function -I"$ROOT/include" -I. \
	-ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
	-DCORDIC_FAULT_SKIP_JOB_ORDER -c cordic_group.cpp -o build_fault/cordic_group.o
g++ -shared -fPIC -o "$ROOT/cordic_amd/lib_fault.so" build_fault/cordic_group.o \
	-L"$ROOT/cordic_amd" -l:libcordic_amd.so -Wl,-rpath,'$ORIGIN' \
	-L/opt/rocm/lib -lamdhip64 -ldl -lpthread
echo "built cordic_amd/lib_fault.so"
