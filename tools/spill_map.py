#!/usr/bin/env python3
"""Where the (20, 2) solver kernel touches scratch: per role copy (s_setprio marks the start of a role's loop) and per
barrier-to-barrier segment, instruction count and scratch loads / stores.   python tools/spill_map.py <frp_ipm_lds.hip.o> [kernel substring]"""
import os, subprocess, sys, tempfile
L = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "nmpc_ipm_lds_kernelILi20ELi2ELb1ELi3"
with tempfile.TemporaryDirectory() as d:
    subprocess.run([f"{L}/llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fb.bin", obj, f"{d}/copy.o"], check=True)
    subprocess.run([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fb.bin",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/dev.co"], check=True)
    lines = subprocess.run([f"{L}/llvm-objdump", "-d", f"{d}/dev.co"], capture_output=True, text=True).stdout.split("\n")
start = [i for i, l in enumerate(lines) if want in l and l.endswith(">:")][0]
end = ([i for i, l in enumerate(lines) if i > start and l.endswith(">:")] + [len(lines)])[0]
n = ld = st = 0; tot = 0
for l in lines[start + 1:end]:
    t = l.strip().split()
    if not t: continue
    n += 1
    ld += t[0].startswith("scratch_load"); st += t[0].startswith("scratch_store")
    if t[0] in ("s_barrier", "s_setprio", "s_swappc_b64", "s_endpgm"):
        tag = t[0] + (" " + t[1] if t[0] == "s_setprio" else "")
        if t[0] == "s_setprio": print("  ---- role loop starts")
        if ld or st or n > 100: print(f"  {tag:14s} {n:5d} instructions  scratch loads {ld:3d} stores {st:3d}")
        tot += ld + st; n = ld = st = 0
print("scratch instructions in the kernel body:", tot)
