ls /sys/class/drm/ | head; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done 2>&1 | head -30
rocm-smi --showclocks 2>&1 | head -30
python -c "import amdsmi; print('amdsmi ok')" 2>&1 | tail -1
ls /sys/class/drm/card*/device/hwmon/*/ 2>/dev/null | head -40
cat /sys/class/drm/card*/device/hwmon/*/freq1_input 2>/dev/null | head
