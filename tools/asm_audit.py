#!/usr/bin/env python3
"""What the compiler did to the loads and shuffles of every kernel of one translation unit.

   python tools/asm_audit.py kokkos-kernels_amd/csrc/kk_spgemm.hip [name filter]

Compiles the file to gfx950 assembly (hipcc -S --cuda-device-only, the library's flags) and prints per kernel:
   serial   global loads that have `s_waitcnt vmcnt(0)` among the three instructions before them -- a load the source guards with a
            condition is compiled as a branch around it, and at the join every load in flight is awaited before the next one is issued
   loads    global / buffer loads in all
   xlane    ds_bpermute / ds_permute / ds_swizzle: every __shfl* is one, and it occupies the LDS unit
   lds      other LDS instructions
   valu     vector-ALU instructions
Static counts over the whole kernel (unrolled code counts as often as it is unrolled): read them next to the SQ counters
(tools/gpu_r4.sh sq), not instead of them."""
import os, re, subprocess, sys, tempfile

def main():
    src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=off",
                           "-S", "--cuda-device-only", "-c", src, "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    cur = None; res = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m: cur = m.group(1); res[cur] = [0, 0, 0, 0, 0]
        if cur is None: continue
        r = res[cur]
        if re.search(r"\b(global_load|buffer_load)", l):
            r[1] += 1
            k = i - 1; seen = 0
            while k > 0 and seen < 3:
                t = lines[k].strip()
                if t and not t.startswith(";") and not t.startswith("."):
                    seen += 1
                    if "s_waitcnt vmcnt(0)" in t: r[0] += 1; break
                    if "global_load" in t: break
                k -= 1
        elif re.search(r"\bds_(bpermute|permute|swizzle)", l): r[2] += 1
        elif re.search(r"^\s+ds_", l): r[3] += 1
        elif re.search(r"^\s+v_", l): r[4] += 1
    names = subprocess.run(["c++filt"], input="\n".join(res.keys()), capture_output=True, text=True).stdout.split("\n")
    print("%6s %6s %6s %6s %7s  kernel" % ("serial", "loads", "xlane", "lds", "valu"))
    for (k, r), n in sorted(zip(res.items(), names), key=lambda x: -(x[0][1][0] * 1000 + x[0][1][2])):
        if r[1] and flt in n: print("%6d %6d %6d %6d %7d  %s" % (r[0], r[1], r[2], r[3], r[4], n[:150]))

if __name__ == "__main__":
    main()
