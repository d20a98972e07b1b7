#!/bin/bash
# bash tools/dev/r2l5_variants.sh: builds timing-probe variants of the latency engine into ablibs/ (half the rows per product, no barriers, both)
# — they compute garbage; tools/dev/r2l5_time.py times one 256-Enc launch on each ($ZKP_HIP_LAT_LIB) to split a slot into rows | barriers | the rest.
mkdir -p ablibs
F="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wl,-Bsymbolic -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -pragma-unroll-threshold=200000 -DZKP_W=9 -DZKP_SECONDARY_ENGINE"
/opt/rocm/bin/hipcc $F -DZKP_R2L5_DEV_ROW_DIVISOR=2 zk-paillier_amd/csrc/zkp_api.hip -o ablibs/lat_half_rows.so &
/opt/rocm/bin/hipcc $F -DZKP_R2L5_DEV_NO_BARRIERS=1 zk-paillier_amd/csrc/zkp_api.hip -o ablibs/lat_no_barriers.so &
/opt/rocm/bin/hipcc $F -DZKP_R2L5_DEV_NO_BARRIERS=1 -DZKP_R2L5_DEV_ROW_DIVISOR=2 zk-paillier_amd/csrc/zkp_api.hip -o ablibs/lat_half_rows_no_barriers.so &
wait
ls -la ablibs/
