#!/usr/bin/env python3
"""Print a table of per-kernel register / LDS / occupancy figures for HIP sources (gfx950).
usage: tools/kres.py file.hip [...]"""
import re, subprocess, sys, os
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
def demangle(n):
    try:
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                out = subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
                if out and out != n:
                    return out
            except Exception:
                pass
        return n
    except Exception:
        return n
for src in sys.argv[1:]:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fvisibility=hidden",
                          "-I" + os.path.join(root, "include"), "-Rpass-analysis=kernel-resource-usage", "-x", "hip",
                          "-c", os.path.abspath(src), "-o", "/dev/null"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src))).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
        if not m: continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:") or txt.startswith("Name:"):
            cur = {"name": demangle(txt.split(":",1)[1].strip())}; rows.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.split(":",1); cur[k.strip()] = v.strip()
    print(f"== {src}")
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["name"])[:70]
        print(f"  {name:70s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs','?'):>3} "
              f"spill={r.get('VGPRs Spill','?')}/{r.get('SGPRs Spill','?')} scratch={r.get('ScratchSize [bytes/lane]','?')} "
              f"lds={r.get('LDS Size [bytes/block]','?'):>6} occ={r.get('Occupancy [waves/SIMD]','?')}")
