"""HBM ceilings of this box by access mix (what an HBM-bound kernel can be asked to reach): write-only (fill), read-only (sum),
copy (read + write), on 1 GiB tensors; rocm-smi clock / power beside them."""
import re, subprocess, sys, threading, time
import torch
n = 1 << 28   # fp32 elements = 1 GiB
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda"); xb = torch.empty(2 * n, dtype=torch.bfloat16, device="cuda")
x.normal_(); xb.normal_()
def rate(name, f, nbytes, secs=1.5):
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            m, p = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out), re.search(r"Power \(W\): ([\d.]+)", out)
            if m and p: samples.append((int(m.group(1)), float(p.group(1))))
            time.sleep(0.1)
    for _ in range(3): f()
    torch.cuda.synchronize(); th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); k = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(10): f()
        k += 10; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    us = e0.elapsed_time(e1) / k * 1e3
    s = samples[1:] or samples
    print(f"{name:34s} {nbytes / us / 1e6:6.2f} TB/s  ({us:7.1f} us)  sclk {sum(a for a,_ in s)/max(len(s),1):5.0f} MHz  power {sum(b for _,b in s)/max(len(s),1):5.0f} W", flush=True)
rate("write only  (fill 1 GiB fp32)", lambda: y.fill_(1.5), 4.0 * n)
rate("read only   (sum 1 GiB fp32)", lambda: x.sum(), 4.0 * n)
rate("copy        (1 GiB -> 1 GiB)", lambda: y.copy_(x), 8.0 * n)
rate("read 1 : write 8 (bf16 -> 8x fp32?)", lambda: torch.add(x, 1.0, out=y), 8.0 * n)
rate("bf16 read only (sum 1 GiB bf16)", lambda: xb.float().sum() if False else xb.sum(dtype=torch.float32), 2.0 * 2 * n)
