#!/usr/bin/env python3
"""Sample GPU power / clock / temperature from sysfs (hwmon) while bench.py runs, to tell a
power-capped clock from a compute- or memory-bound one.  Usage: power_probe.py [bench args...]"""
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "power1_cap_max", "freq1_input", "freq2_input",
                     "temp1_input", "temp2_input", "temp3_input"):
            v = read(os.path.join(hw, name))
            if v is not None:
                out.setdefault(hw, {})[name] = v
    return out


def main():
    print("sensors at rest:", sensors(), flush=True)
    args = sys.argv[1:] or ["--steps", "1500", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-timing"]
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, text=True)
    t0 = time.time()
    rows = []
    while p.poll() is None:
        s = sensors()
        rows.append((time.time() - t0, s))
        time.sleep(0.1)
    out = p.stdout.read()
    busy = []
    for t, s in rows:
        for hw, d in s.items():
            pw = d.get("power1_average") or d.get("power1_input")
            line = "t=%6.2f %s power_W=%s sclk_MHz=%s mclk_MHz=%s temp=%s/%s/%s" % (
                t, hw.split("/")[4], int(pw) / 1e6 if pw else None,
                int(d["freq1_input"]) / 1e6 if "freq1_input" in d else None,
                int(d["freq2_input"]) / 1e6 if "freq2_input" in d else None,
                d.get("temp1_input"), d.get("temp2_input"), d.get("temp3_input"))
            busy.append(line)
    step = max(1, len(busy) // 60)
    print("\n".join(busy[::step]))
    print(out.strip()[:400])


if __name__ == "__main__":
    main()
