#!/usr/bin/env python3
"""Shader clock and package power of GPU 0 while a command runs (sysfs, sampled every 20 ms): does the chip hold its clock under this load?
usage: clock_watch.py <label> -- <command ...>   -> one JSON line {label, samples, sclk_mhz: {min, median, p90, max}, power_w: {...}, busy_pct: {...}}"""
import glob, json, statistics, subprocess, sys, threading, time
i = sys.argv.index("--"); label = " ".join(sys.argv[1:i]); cmd = sys.argv[i + 1:]
import os
def our_card():
    """the DRM card of HIP device 0 of this environment (a box shows all eight cards of its host; only one is ours): matched by PCI address"""
    try:
        out = subprocess.check_output([sys.executable, "-c", "import torch; p = torch.cuda.get_device_properties(0); print('%04x:%02x:%02x' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))"], text=True).strip()
    except Exception:
        return None
    for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if "-" in os.path.basename(c):
            continue
        if os.path.basename(os.path.realpath(c + "/device")).lower().startswith(out.lower()):
            return c
    return None
card = our_card() or "/sys/class/drm/card0"
def first(pats):
    for p in pats:
        g = sorted(glob.glob(p))
        if g:
            return g[0]
    return None
f_clk = first([card + "/device/hwmon/hwmon*/freq1_input"])
f_pow = first([card + "/device/hwmon/hwmon*/power1_input", card + "/device/hwmon/hwmon*/power1_average"])
f_busy = first([card + "/device/gpu_busy_percent"])
rd = lambda f: float(open(f).read().split()[0]) if f else float("nan")
clk, pw, busy, stop = [], [], [], False
def sample():
    while not stop:
        try:
            clk.append(rd(f_clk) / 1e6); pw.append(rd(f_pow) / 1e6); busy.append(rd(f_busy))
        except Exception:
            pass
        time.sleep(0.02)
t = threading.Thread(target=sample); t.start()
rc = subprocess.call(cmd)
stop = True; t.join()
def st(v):
    v = sorted(x for x in v if x == x)
    return {"min": v[0], "median": statistics.median(v), "p90": v[int(0.9 * (len(v) - 1))], "max": v[-1]} if v else None
hot = [k for k in range(min(len(clk), len(busy))) if busy[k] >= 90]
print(json.dumps({"label": label, "rc": rc, "samples": len(clk), "files": [f_clk, f_pow, f_busy], "sclk_mhz": st(clk), "power_w": st(pw), "busy_pct": st(busy),
                  "while_busy_ge_90": {"samples": len(hot), "sclk_mhz": st([clk[k] for k in hot]), "power_w": st([pw[k] for k in hot])}}))
