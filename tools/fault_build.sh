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
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}	# as cordic_amd/csrc/Makefile
ARCH=${ARCH:-gfx950}
cd "$ROOT/cordic_amd/csrc"
# the product build's objects are reused; a snapshot (the GPU boxes get the
# libraries without build/) or `make clean` leaves none: nothing to do here
# then -- the test that wants lib_fault.so skips when it is missing
ls build/*.o > /dev/null 2>&1 || { echo "no product objects: lib_fault.so not rebuilt"; exit 0; }
mkdir -p build_fault
find build_fault -type l -delete	# objects of an earlier source layout
# every other object is identical: reuse the product build's
for o in build/*.o; do
	b=$(basename $o)
	[ "$b" = cordic_group.o ] || ln -sf ../$o build_fault/$b
done
g++ -O3 -std=c++17 -fPIC -fwrapv -Wall -Wno-unused-function -I"$ROOT/include" -I. \
	-ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
	-DCORDIC_FAULT_SKIP_JOB_ORDER -c cordic_group.cpp -o build_fault/cordic_group.o
"$HIPCC" --offload-arch="$ARCH" -shared -fPIC -o "$ROOT/cordic_amd/lib_fault.so" \
	build_fault/*.o -ldl
echo "built cordic_amd/lib_fault.so"
