bash tools/ab.sh ablibs/main.so ablibs/w16.so ablibs/w64f.so ablibs/v3.so 2>&1 | tee gpurun_out/ab_windows.txt
bash profiles/r03_collect_all.sh main > gpurun_out/pmc_main.log 2>&1
ZKP_HIP_LIB=$PWD/ablibs/w16.so bash profiles/r03_collect_all.sh w16 > gpurun_out/pmc_w16.log 2>&1
tail -3 gpurun_out/pmc_main.log gpurun_out/pmc_w16.log
