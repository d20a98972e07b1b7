#!/bin/bash
# bash tools/dev/build_variant_sched.sh name strategy|none : the throughput engine with another machine-scheduler strategy
name=$1; strat=$2
rm -rf build/v_$name; mkdir -p build/v_$name ablibs
FL=""; [ "$strat" != none ] && FL="-mllvm -amdgpu-sched-strategy=$strat"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wl,-Bsymbolic $FL -mllvm -pragma-unroll-threshold=200000 zk-paillier_amd/csrc/zkp_api.hip -ldl -o build/v_$name/lib.so -save-temps=obj > build/v_$name/log.txt 2>&1 || { echo "$name FAILED"; tail -3 build/v_$name/log.txt; exit 1; }
mv build/v_$name/zkp_api-hip-amdgcn-amd-amdhsa-gfx950.s build/v_$name/isa.s; rm -f build/v_$name/zkp_api*
cp build/v_$name/lib.so ablibs/$name.so; echo "$name done"
