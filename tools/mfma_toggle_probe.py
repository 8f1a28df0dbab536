"""Runs tools/_probe/mfma_toggle_probe <variant> for a few seconds each while sampling the card's board power / reported clock (sysfs hwmon):
what the matrix pipe sustains under the 1400 W cap when its operands change every instruction (profiles/r5_mfma_toggle.md)."""
import glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "_probe", "mfma_toggle_probe")
pci = subprocess.run([exe, "9", "0"], capture_output=True, text=True).stdout.split("pci ")[-1].split()[0].lower()      # the HIP device's PCI address
hw = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(os.path.join(h, "power1_input"))
      and os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))).lower() == pci]
def rd(h, n):
    try: return int(open(os.path.join(h, n)).read())
    except Exception: return None
for v in (0, 1, 2, 3, 1):
    samples, stop = [], threading.Event()
    def sampler():
        while not stop.is_set():
            if hw: samples.append((time.time(), (rd(hw[0], "power1_input") or 0) / 1e6, (rd(hw[0], "freq1_input") or 0) / 1e6))
            time.sleep(0.01)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time()
    out = subprocess.run([exe, str(v), "4"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    stop.set(); th.join()
    late = [(w, f) for (t, w, f) in samples if t - t0 > 2.0]
    pw = sum(w for w, _ in late) / max(len(late), 1); fr = sum(f for _, f in late) / max(len(late), 1)
    print(f"{out} | board power {pw:.0f} W (cap {(rd(hw[0], 'power1_cap') or 0) / 1e6:.0f} W), reported sclk {fr:.0f} MHz" if hw else out, flush=True)
