"""What clock and power does the card run at while the encoder GEMMs execute?  Loops one GEMM shape for ~3 s per case and samples the
amdgpu sysfs files (pp_dpm_sclk current level, hwmon freq1_input / power1_average / power1_cap) from a thread; prints the median and
range.  MI355X_MICROARCH.md prices the bf16 MFMA peak (2.5 PFLOP/s) at 2.4 GHz: the peak at the clock a kernel actually gets is
2.5 PF x clock / 2.4 GHz.  usage: python tools/clock_probe.py"""
import glob, os, statistics, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops, _lib as L


def find(pattern):
    for c in sorted(glob.glob(pattern)):
        try:
            open(c).read()
            return c
        except OSError:
            continue
    return None


# the box shows every card of the host under /sys/class/drm; ours is the one whose PCI address torch reports for cuda:0
props = torch.cuda.get_device_properties(0)
bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
mine = [c for c in sorted(glob.glob("/sys/class/drm/card*")) if os.path.basename(os.path.realpath(c + "/device")) == bdf]
print("cuda:0 is PCI", bdf, "->", mine, flush=True)
card = mine[0] if mine else "/sys/class/drm/card*"
freq = find(card + "/device/hwmon/hwmon*/freq1_input")
power = find(card + "/device/hwmon/hwmon*/power1_average") or find(card + "/device/hwmon/hwmon*/power1_input")
cap = find(card + "/device/hwmon/hwmon*/power1_cap")
sclk = find(card + "/device/pp_dpm_sclk")
print("sysfs:", freq, power, cap, sclk, flush=True)
if cap:
    print("power cap W:", int(open(cap).read()) / 1e6)
stop = False
samples = []


def sampler():
    while not stop:
        row = {}
        try:
            if freq: row["mhz"] = int(open(freq).read()) / 1e6
            if power: row["w"] = int(open(power).read()) / 1e6
            if sclk:
                cur = [l for l in open(sclk).read().splitlines() if l.strip().endswith("*")]
                if cur: row["sclk"] = cur[0].strip()
        except OSError as e:
            row["err"] = repr(e)
        samples.append(row)
        time.sleep(0.02)


dev, dt = "cuda:0", torch.bfloat16
M = 48000
for name, n, k, act in (("fc1 plain", 5120, 1280, L.ACT_NONE), ("fc2", 1280, 5120, L.ACT_NONE), ("hipBLASLt fc1", 5120, 1280, None)):
    a = torch.randn(M, k, device=dev, dtype=dt)
    w = torch.randn(n, k, device=dev, dtype=dt) * 0.03
    bias = torch.randn(n, device=dev, dtype=dt)
    out = torch.empty(M, n, device=dev, dtype=dt)
    run = (lambda: ops.gemm(a, w, bias=bias, act=act, out=out)) if act is not None else (lambda: torch.matmul(a, w.t(), out=out))
    for _ in range(5): run()
    torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); it = 0
    e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(50): run()
        torch.cuda.synchronize(); it += 50
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
    ms = e0.elapsed_time(e1) / it
    mh = [s["mhz"] for s in samples[len(samples) // 4:] if "mhz" in s]
    pw = [s["w"] for s in samples[len(samples) // 4:] if "w" in s]
    sc = sorted({s.get("sclk", "") for s in samples})
    tf = 2.0 * M * n * k / ms / 1e9
    line = f"{name}: {ms:.3f} ms {tf:.0f} TFLOP/s"
    if mh:
        med = statistics.median(mh)
        line += f" | clock MHz median {med:.0f} range {min(mh):.0f}-{max(mh):.0f} -> bf16 peak at that clock {2500 * med / 2400:.0f} TF, frac {tf / (2500 * med / 2400):.3f}"
    if pw: line += f" | power W median {statistics.median(pw):.0f} max {max(pw):.0f}"
    print(line, "| sclk levels seen:", sc[:4], flush=True)
