#!/usr/bin/env python3
"""The multiply-add roofline WITH the clock it was measured at: runs csrc/microbench/mad_sustained (v_mad_u64_u32, 16 independent
accumulators, 8 waves per SIMD, kernels of 18 ms ... 4 s) while sampling the shader clock and board power from sysfs, and prints for
every kernel: lane-MAD/s, the mean clock during the kernel, and cycles per wave instruction per SIMD AT THAT CLOCK.
    python tools/dev/mad_peak_with_clock.py > gpurun_out/mad_sustained_with_clock_r03.jsonl"""
import glob, json, os, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "mad_sustained"          # or: mad_random_operands
src = os.path.join(ROOT, "zk-paillier_amd", "csrc", "microbench", name + ".hip")
exe = "/tmp/" + name
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-mllvm", "-pragma-unroll-threshold=200000", src, "-o", exe])
# sysfs hwmon of the GPU this process sees as device 0 (the box exposes the hwmon of every GPU of its node): by PCI bus id
import ctypes
hip = ctypes.CDLL("libamdhip64.so"); buf = ctypes.create_string_buffer(64)
assert hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0
bus = buf.value.decode().lower()
h = glob.glob(f"/sys/bus/pci/devices/{bus}/hwmon/hwmon*")[0]
freq, power = os.path.join(h, "freq1_input"), os.path.join(h, "power1_input")
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        try: samples.append((time.monotonic(), int(open(freq).read()) / 1e9, int(open(power).read()) / 1e6))
        except (OSError, ValueError): pass
        stop.wait(0.02)
t = threading.Thread(target=sampler, daemon=True); t.start()
p = subprocess.Popen([exe] + sys.argv[2:], stdout=subprocess.PIPE, text=True)
last = time.monotonic()
for line in p.stdout:
    now = time.monotonic()
    rec = json.loads(line)
    dur = rec["ms"] * 1e-3
    win = [s for s in samples if now - dur * 0.9 <= s[0] <= now - dur * 0.05]       # the kernel's own time span (it ended just now)
    if win:
        ghz = sum(s[1] for s in win) / len(win)
        rec.update({"clock_ghz_mean": round(ghz, 4), "clock_samples": len(win), "power_w_mean": round(sum(s[2] for s in win) / len(win), 1),
                    "cycles_per_wave_instr_per_simd_at_sampled_clock": round(ghz * 1e9 * 256 * 4 * 64 / rec["lane_mad_per_s"], 3),
                    "lane_mad_per_s_over_16_lanes_per_clk_per_simd": round(rec["lane_mad_per_s"] / (16 * 1024 * ghz * 1e9), 4)})
        if "cycles_per_substep_per_wave_at_2.4GHz" in rec:
            rec["cycles_per_substep_per_wave_at_sampled_clock"] = round(rec["cycles_per_substep_per_wave_at_2.4GHz"] * ghz / 2.4, 1)
            rec["substeps_per_s_per_simd"] = round(ghz * 1e9 / rec["cycles_per_substep_per_wave_at_sampled_clock"], 1)
        if "valu_instr_per_mad" in rec:
            rec["cycles_per_valu_instr_per_simd_at_sampled_clock"] = round(rec["cycles_per_wave_instr_per_simd_at_sampled_clock"] / rec["valu_instr_per_mad"], 3)
    print(json.dumps(rec), flush=True)
stop.set(); p.wait()
