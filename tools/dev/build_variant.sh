#!/bin/bash
# bash tools/dev/build_variant.sh name [-Dflags...]: builds ablibs/name.so — the throughput engine exactly as __graft_entry__.build() builds it
# (three translation units), with the extra flags — and keeps its device assembly under build/v_name/.  The in-tree libraries are put back.
# ZKP_HIP_LIB=ablibs/name.so selects a variant (tools/ab.sh runs A/B pairs on one box).  One variant at a time: build() owns build/v_isa/.
name=$1; shift
mkdir -p ablibs build/v_$name
cp zk-paillier_amd/libzkp_hip.so /tmp/libzkp_hip.keep.so; cp zk-paillier_amd/libzkp_hip_lat.so /tmp/libzkp_hip_lat.keep.so
rm -rf /tmp/v_isa.keep; cp -r build/v_isa /tmp/v_isa.keep 2>/dev/null
ZKP_EXTRA_FLAGS="$*" python - <<'PY' > build/v_$name/log.txt 2>&1 || { echo "$name FAILED"; tail -5 build/v_$name/log.txt; }
import os, __graft_entry__ as g
os.utime(g.LAT, None)                      # (the latency engine is not rebuilt: only the throughput engine is the variant)
srcs = [g.LIB]
os.remove(g.LIB)
g.build()
PY
cp zk-paillier_amd/libzkp_hip.so ablibs/$name.so && cp build/v_isa/isa*.s build/v_$name/ && echo "$name done"
cp /tmp/libzkp_hip.keep.so zk-paillier_amd/libzkp_hip.so; cp /tmp/libzkp_hip_lat.keep.so zk-paillier_amd/libzkp_hip_lat.so
[ -d /tmp/v_isa.keep ] && { rm -rf build/v_isa; cp -r /tmp/v_isa.keep build/v_isa; }      # (tests/test_isa_quality.py reads the assembly of the IN-TREE library)
