"""The bf16 MFMA rate the 1400 W package limit allows, by LDS operand traffic per MFMA (scripts/probes/power_roof.hip):
what a conv kernel with a bigger register tile could be asked to reach on this box.  -> profiles/r03_power_roof.txt"""
import ctypes, os, re, subprocess, sys, threading, time
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "probes", "power_roof.so"))
lib.roof_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
out = torch.zeros(1 << 20, device="cuda"); seed = torch.randint(0, 2 ** 31 - 1, (4096,), device="cuda", dtype=torch.int32)
NAMES = ["64x64 wave tile, operands in registers (no LDS)", "64x64 wave tile   1.00 KB LDS / MFMA  (today's kernels)",
         "128x64 wave tile  0.75 KB LDS / MFMA", "128x128 wave tile 0.50 KB LDS / MFMA (1 wave / SIMD)",
         "128x128 wave tile, operands in registers (no LDS)",
         "64x64   1.00 KB LDS + 0.125 KB L2->LDS DMA / MFMA", "64x64   1.00 KB LDS + 0.25 KB L2->LDS DMA / MFMA (k_conv_halo)",
         "128x64  0.75 KB LDS + 0.125 KB L2->LDS DMA / MFMA", "128x64  0.75 KB LDS + 0.25 KB L2->LDS DMA / MFMA",
         "64x64   no LDS reads, 0.25 KB L2->LDS DMA / MFMA", "64x64   0.50 KB LDS + 0.50 KB L2->VGPR weight fragments / MFMA", "64x64   0.50 KB LDS + 0.50 KB L2->VGPR + 0.125 KB L2->LDS DMA / MFMA",
         "64x64   1.00 KB LDS + 0.25 KB HBM->LDS DMA / MFMA (256 MiB source)"]
src_small = torch.randint(0, 2 ** 31 - 1, (1 << 18,), device="cuda", dtype=torch.int32)     # 1 MiB: L2 resident (weights)
src_big = torch.randint(0, 2 ** 31 - 1, (1 << 26,), device="cuda", dtype=torch.int32)       # 256 MiB: streams from HBM
print(f"# MFMA 32x32x16 bf16 back to back, {secs} s per row; rocm-smi sclk / socket power averaged over the run\n# operands: " + ("bf16 N(0, 1) (random exponents: what a conv sees)" if len(sys.argv) > 2 and sys.argv[2] == "normal" else "random sign and mantissa, fixed exponent"))
RAW = len(sys.argv) > 2 and sys.argv[2] == "normal"
# the DMA sources are what k_roof_wt's fragments read after the first segment: the same kind of data as the LDS fill
if RAW:
    src_small = torch.randn(1 << 19, device="cuda").bfloat16().view(torch.int32).contiguous()
    src_big = torch.randn(1 << 27, device="cuda").bfloat16().view(torch.int32).contiguous()
else:
    src_small = (src_small & 0x007f807f) | 0x3f003f00
    src_big = (src_big & 0x007f807f) | 0x3f003f00
if RAW:   # operands ~ N(0, 1) in bf16 (random exponents) instead of fixed-exponent random mantissas
    seed = torch.randn(8192, device="cuda").bfloat16().view(torch.int32).contiguous()
    lib.roof_set_raw(1)
for row in (range(13) if (len(sys.argv) < 3 or RAW) else ()):
    v = 6 if row == 12 else row; src = src_big if row == 12 else src_small
    per_iter = lib.roof_mfmas_per_iter(v); grid = 256 if v in (3, 4) else 512; iters = 15984 // per_iter // 3 * 3
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.roof_launch(v, out.data_ptr(), seed.data_ptr(), iters, grid, st, src.data_ptr(), src.numel() * 4)
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            m, p = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o), re.search(r"Power \(W\): ([\d.]+)", o)
            if m and p: samples.append((int(m.group(1)), float(p.group(1))))
            time.sleep(0.15)
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize(); th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(20): f()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    flops = grid * 4 * iters * per_iter * 32768.0
    s = samples[2:] if len(samples) > 4 else samples
    sclk = sum(a for a, _ in s) / max(len(s), 1)
    print(f"{NAMES[row]:70s} {us:8.1f} us  {flops / us / 1e6:6.0f} TF  sclk {sclk:5.0f} MHz  power {sum(b for _, b in s) / max(len(s), 1):5.0f} W"
          f"  (MFMA pipe busy {flops / us / 1e6 / (2500.0 * sclk / 2400.0):.2f} of the clock's peak)", flush=True)

lib.roof_launch_wt.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint]
WT_ROWS = ((0, "128x64 wave-tile conv schedule: 1 barrier / 64 MFMAs, halo from HBM + weights from L2 (0.133 KB / MFMA)", src_big),
                        (1, "  same with a barrier per tap (16 MFMAs)", src_big),
                        (0, "  1 barrier / 64 MFMAs, halo from an L2-resident source too", src_small),
                        (2, "  1 barrier / 64 MFMAs, HBM halo, accumulators restarted every 16 segments (a new output tile)", src_big))
for bar, name, hsrc in WT_ROWS:
    if bar == 2:
        if not RAW: continue
        lib.roof_set_raw(2); bar = 0
    iters = 256; grid = 256; st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.roof_launch_wt(bar, out.data_ptr(), seed.data_ptr(), iters, grid, st, src_small.data_ptr(), src_small.numel() * 4, hsrc.data_ptr(), hsrc.numel() * 4)
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            m, p = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o), re.search(r"Power \(W\): ([\d.]+)", o)
            if m and p: samples.append((int(m.group(1)), float(p.group(1))))
            time.sleep(0.15)
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize(); th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(20): f()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    flops = grid * 8 * iters * 64 * 32768.0
    s = samples[2:] if len(samples) > 4 else samples
    sclk = sum(a for a, _ in s) / max(len(s), 1)
    print(f"{name:70s} {us:8.1f} us  {flops / us / 1e6:6.0f} TF  sclk {sclk:5.0f} MHz  power {sum(b for _, b in s) / max(len(s), 1):5.0f} W"
          f"  (MFMA pipe busy {flops / us / 1e6 / (2500.0 * sclk / 2400.0):.2f} of the clock's peak)", flush=True)
