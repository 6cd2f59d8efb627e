"""Resident workgroups per CU of every kernel, from the compiler's metadata (hipcc -save-temps .s files in a directory):
registers (512 per SIMD lane, allocated in blocks of 8), LDS (160 KB) and wave slots (8 per SIMD).  Compare with what each
launcher assumes when it sizes a persistent grid."""
import glob, re, subprocess, sys
d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/occ"
rows = []
for f in sorted(glob.glob(d + "/*-hip-amdgcn-amd-amdhsa-gfx950.s")):
    txt = open(f).read()
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.max_flat_workgroup_size:\s+(\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
        agpr, lds, threads, name, vgpr = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5))
        waves = (threads + 63) // 64
        regs = (vgpr + 7) // 8 * 8          # unified VGPR + AGPR file: vgpr_count is the total
        wps = min(8, 512 // max(regs, 1))   # waves per SIMD by registers
        by_reg = wps * 4 // waves if waves <= wps * 4 else 0
        by_lds = (160 * 1024) // lds if lds else 99
        try:
            dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception:
            dn = name
        rows.append((dn[:86], threads, vgpr, lds, wps, by_reg, by_lds, min(by_reg, by_lds, 32 // waves)))
print(f"{'kernel':86s} thr  vgpr    lds  w/SIMD  WG/CU(reg) WG/CU(lds) WG/CU")
for r in rows:
    print(f"{r[0]:86s} {r[1]:4d} {r[2]:4d} {r[3]:7d} {r[4]:4d} {r[5]:8d} {r[6]:10d} {r[7]:6d}")
