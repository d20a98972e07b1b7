#!/bin/bash
# bash tools/dev/clock_while.sh <command ...>: runs the command and prints min / mean / max of the shader clock (hwmon freq1_input, MHz) sampled every ~5 ms while it ran
F=$(ls /sys/class/drm/card*/device/hwmon/*/freq1_input 2>/dev/null | head -1)
T=$(mktemp)
( while true; do cat $F >> $T 2>/dev/null; sleep 0.005; done ) & S=$!
"$@"
kill $S 2>/dev/null; wait $S 2>/dev/null
awk '{v=$1/1e6; s+=v; n++; if(min==""||v<min)min=v; if(v>max)max=v} END {printf("{\"sclk_mhz_min\": %.0f, \"sclk_mhz_mean\": %.0f, \"sclk_mhz_max\": %.0f, \"samples\": %d}\n", min, s/n, max, n)}' $T
rm -f $T
