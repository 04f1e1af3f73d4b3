#!/usr/bin/env python3
"""Samples the GPU's shader clock and power (sysfs hwmon of the amdgpu device) while a command runs:
    python tools/clock_watch.py -- python bench.py --steps 40 --cpu-sample 0
Prints min / median / max of what it saw (10 ms period).  Diagnostic for DESIGN.md 6: the list kernels are bound by
dependent scalar instructions, i.e. by the shader clock, and the firmware lowers that clock when HBM-bound kernels run."""
import glob
import os
import subprocess
import sys
import threading
import time


def my_card():
    """sysfs device directory of HIP device 0 (the host may have other users' GPUs beside ours); asked in a child process
    so that this process never initialises the GPU"""
    code = ("import ctypes as C; h = C.CDLL('libamdhip64.so'); b = C.create_string_buffer(64); "
            "print(b.value.decode() if h.hipDeviceGetPCIBusId(b, 64, 0) == 0 else '')")
    try:
        bus = subprocess.check_output([sys.executable, "-c", code], text=True).strip().lower()
    except Exception:
        bus = ""
    for d in glob.glob("/sys/class/drm/card*/device"):
        if bus and os.path.realpath(d).lower().endswith(bus):
            return d
    return None


def find():
    out = {}
    mine = my_card()
    pattern = (mine + "/hwmon/hwmon*") if mine else "/sys/class/drm/card*/device/hwmon/hwmon*"
    print("watching", pattern)
    for hw in glob.glob(pattern):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(hw, []).append(p)
    return out


def main():
    cmd = sys.argv[sys.argv.index("--") + 1:]
    files = find()
    if not files:
        print("no amdgpu hwmon files readable")
    samples = {}
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            for hw, ps in files.items():
                for p in ps:
                    try:
                        v = int(open(p).read().strip())
                    except Exception:
                        continue
                    samples.setdefault(p, []).append(v)
            time.sleep(0.01)

    t = threading.Thread(target=watch, daemon=True)
    t.start()
    rc = subprocess.call(cmd)
    stop.set()
    t.join()
    # the busiest card (highest peak power) is the one the command ran on: its clock while it drew the most
    pw = {p: v for p, v in samples.items() if "power1" in p}
    if pw:
        top = max(pw, key=lambda p: max(pw[p]))
        hw = os.path.dirname(top)
        fq = samples.get(os.path.join(hw, "freq1_input"))
        if fq:
            n = min(len(fq), len(pw[top]))
            pairs = sorted(zip(pw[top][:n], fq[:n]))
            for lo, hi in ((0, 400), (400, 800), (800, 1100), (1100, 2000)):
                f = sorted(fr / 1e6 for (w, fr) in pairs if lo <= w / 1e6 < hi)
                if f:
                    print("busiest card %s: power %4d-%4d W: %5d samples, shader clock min %6.0f median %6.0f max %6.0f MHz; cap %s W" % (
                        hw[-40:], lo, hi, len(f), f[0], f[len(f) // 2], f[-1],
                        open(os.path.join(hw, "power1_cap")).read().strip()[:-6] if os.path.exists(os.path.join(hw, "power1_cap")) else "?"))
    for p, v in sorted(samples.items()):
        s = sorted(v)
        unit = 1e6 if "freq" in p else 1e6  # Hz -> MHz, microwatt -> W
        print("%-70s n %5d  min %8.1f  p10 %8.1f  median %8.1f  p90 %8.1f  max %8.1f" %
              (p[-60:], len(s), s[0] / unit, s[len(s) // 10] / unit, s[len(s) // 2] / unit, s[9 * len(s) // 10] / unit, s[-1] / unit))
    sys.exit(rc)


if __name__ == "__main__":
    main()
